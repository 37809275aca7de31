"""Random graphs through the LDS-tiled CSR score / hop kernels (option CSR_TILED = 3) against the per-edge CSR kernels (CSR_TILED = 0)
of the same library: sizes 8..1024, sparse / dense / hub rows / isolated rows, float32 and bf16 storage (test infrastructure).
   python tools/exp/fuzz_csr_tiled.py [count] [seed]"""
import os, sys, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.graphml import CsrStructure, gat_forward_rows_csr
dev = torch.device("cuda:0")
count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(count):
    N = rng.choice([8, 9, 31, 64, 100, 129, 255, 256, 300, 511, 512, 513, 777, 1000, 1023, 1024])
    B = rng.choice([1, 2, 3]) if N > 300 else rng.choice([1, 2, 5, 9])
    G = rng.choice([64, 128])
    K, P = rng.choice([2, 3]), rng.choice([1, 2, 4])
    bf16 = rng.random() < 0.5
    kind = rng.choice(["sparse", "dense", "hubs", "empty"])
    g = torch.Generator().manual_seed(500 + it)
    dens = {"sparse": 5.0 / N, "dense": min(0.5, 40.0 / N), "hubs": 3.0 / N, "empty": 1.0 / N}[kind]
    S = (torch.rand(B, N, N, generator=g) < dens).float()
    if kind == "hubs":
        S[:, rng.randrange(N), :] = 1.0          # a row with N out-edges
        S[:, :, rng.randrange(N)] = 1.0          # a column with N in-edges
    if kind == "empty":
        S[:, : N // 2, :] = 0.0
    S = S.to(dev)
    torch.manual_seed(900 + it)
    layer = GraphFilterBatchAttentional(G, G, K, P, attentionMode="KeyQuery").to(dev).eval()
    X = torch.randn(B, N, G, device=dev) * 0.5
    if bf16:
        X = X.to(torch.bfloat16)
    st = CsrStructure().build(S.clone(), 0)
    nnz = st.ready(dev)
    csc = (st.cscptr, st.csc[0], st.csc[1])
    outs = []
    for tiled in (3, 0):
        nat.set_option("CSR_TILED", tiled)
        out = torch.empty(B * N, P * G, dtype=X.dtype, device=dev)
        gat_forward_rows_csr(X, st.rowptr, st.colidx, nnz, layer, out=out, csc=csc)
        torch.cuda.synchronize()
        outs.append(out.float())
    nat.set_option("CSR_TILED", 3)
    a, b = outs
    scale = float(b.abs().max()) + 1e-6
    err = float((a - b).abs().max()) / scale
    tol = 2e-2 if bf16 else 2e-5
    ok = err <= tol and not bool(torch.isnan(a).any())
    bad += 0 if ok else 1
    print("%s B=%d N=%d G=%d K=%d P=%d %s %s nnz/row %.1f  rel err %.2e" % ("ok  " if ok else "FAIL", B, N, G, K, P, "bf16" if bf16 else "f32", kind, nnz / (B * N), err), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
