"""Random dense GSOs through magat_gso_csr_build against the host construction of tests/test_gpu_config5.py (test infrastructure).
   python tools/exp/fuzz_csr_build.py [count] [seed]"""
import os, sys, random
R = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from test_gpu_config5 import _legacy_structure
from magat_pathplanning_amd.graphml import CsrStructure
dev = torch.device("cuda:0")
count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(count):
    N = rng.choice([8, 9, 63, 64, 65, 100, 128, 129, 333, 512, 640, 1000, 1023, 1024])
    B = rng.choice([1, 2, 3, 7])
    rule = rng.choice([0, 1, 2])
    dens = rng.choice([0.0, 1.0 / N, 5.0 / N, 0.02, 0.1, 0.6 if N <= 128 else 0.05])
    dt = rng.choice([torch.float32, torch.float64])
    g = torch.Generator().manual_seed(it)
    S = ((torch.rand(B, N, N, generator=g) < dens).to(dt) * (torch.rand(B, N, N, generator=g).to(dt) - 0.3))
    if rng.random() < 0.3:
        S[:, rng.randrange(N), :] = 1.0
        S[:, :, rng.randrange(N)] = -2.0
    st = CsrStructure().build(S.clone().to(dev), rule, scrub_nan=1, gso_mode=0)
    rowptr, colidx, cscptr, cscsrc, cscpos, nnz = _legacy_structure(S, rule, dev)
    ok = st.exact_nnz() == nnz and torch.equal(st.rowptr.cpu().long(), rowptr) and torch.equal(st.cscptr.cpu().long(), cscptr) and \
        torch.equal(st.colidx[:nnz].cpu().long(), colidx) and torch.equal(st.csc[0][:nnz].cpu().long(), cscsrc) and \
        torch.equal(st.csc[1][:nnz].cpu().long(), cscpos)
    bad += 0 if ok else 1
    print("%s B=%d N=%d rule=%d dens=%.4f %s nnz %d (%.1f per instance)" % ("ok  " if ok else "FAIL", B, N, rule, dens, str(dt)[6:], nnz, nnz / B), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
