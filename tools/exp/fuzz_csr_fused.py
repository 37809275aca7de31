"""Random graphs through the FUSED bf16-storage CSR layer (csrc/gat_csr_fused.hip, option CSR_FUSED = 1) against (a) the oracle's
emulation of exactly that order (oracle.gat_layer_forward_bf16_fused; small N only: it is dense) and (b) the split form of the same
library (CSR_FUSED = 0): sizes 1 .. 1024, sparse / dense / hub rows and columns / empty rows / directed, P in {1, 2, 4}, bf16 or
float32 result rows, attention on / off (test infrastructure).
   python tools/exp/fuzz_csr_fused.py [count] [seed]"""
import os
import random
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.graphml import CsrStructure, gat_forward_rows_csr
from oracle import magat_oracle as orc

dev = torch.device("cuda:0")
count = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(count):
    N = rng.choice([1, 2, 7, 31, 32, 33, 64, 100, 129, 255, 256, 300, 511, 513, 777, 1000, 1023, 1024])
    B = rng.choice([1, 2, 3]) if N > 300 else rng.choice([1, 2, 5, 9, 17])
    P = rng.choice([1, 2, 4])
    kind = rng.choice(["sparse", "dense", "hubs", "empty", "directed"])
    f32out = rng.random() < 0.4
    want_att = rng.random() < 0.3
    g = torch.Generator().manual_seed(700 + it)
    dens = {"sparse": 5.0 / N, "dense": min(0.5, 40.0 / N), "hubs": 3.0 / N, "empty": 1.0 / N, "directed": 8.0 / N}[kind]
    S = (torch.rand(B, N, N, generator=g) < dens).float()
    if kind == "hubs":
        S[:, rng.randrange(N), :] = 1.0
        S[:, :, rng.randrange(N)] = 1.0
    if kind == "empty":
        S[:, : N // 2, :] = 0.0
    if kind != "directed":
        S = ((S + S.transpose(1, 2)) > 0).float() if kind in ("sparse", "dense") else S
    torch.manual_seed(900 + it)
    layer = GraphFilterBatchAttentional(128, 128, 2, P, attentionMode="KeyQuery")
    with torch.no_grad():
        layer.bias.uniform_(-0.1, 0.1)
    params = {k: v.detach().clone() for k, v in layer.state_dict().items()}
    layer = layer.to(dev).eval()
    Xf = torch.randn(B, N, 128, generator=g) * 0.5
    X = Xf.to(dev).to(torch.bfloat16)
    Sd = S.to(dev)
    st = CsrStructure().build(Sd.clone(), 0)
    nnz = st.ready(dev)
    csc = (st.cscptr, st.csc[0], st.csc[1])
    outs, atts = [], []
    for fused in (1, 0):
        nat.set_option("CSR_FUSED", fused)
        out = (torch.full((B * N, P * 128 + 4), -3.0, dtype=torch.float32, device=dev) if f32out
               else torch.empty(B * N, P * 128, dtype=torch.bfloat16, device=dev))
        _, att = gat_forward_rows_csr(X, st.rowptr, st.colidx, nnz, layer, out=out, csc=csc, want_attention=want_att)
        torch.cuda.synchronize()
        outs.append(out[:, :P * 128].float().cpu())
        atts.append(att[:, :nnz].cpu() if want_att and nnz else None)
    nat.set_option("CSR_FUSED", 1)
    a, b = outs
    scale = float(b.abs().max()) + 1e-6
    err = float((a - b).abs().max()) / scale
    ok = err <= 2e-2 and not bool(torch.isnan(a).any())
    if atts[0] is not None:
        ok = ok and float((atts[0] - atts[1]).abs().max()) <= 2e-2
    msg = "vs split %.2e" % err
    if N <= 300:            # the dense emulation of the fused order
        y_emul, _ = orc.gat_layer_forward_bf16_fused(X.float().cpu().permute(0, 2, 1).contiguous(), S.unsqueeze(1), params)
        ye = y_emul.permute(0, 2, 1).reshape(B * N, P * 128)
        e2 = float((a - ye).abs().max()) / (float(ye.abs().max()) + 1e-6)
        ok = ok and e2 <= 2.0 ** -7
        msg += "  vs emulation %.2e" % e2
    bad += 0 if ok else 1
    print("%s B=%d N=%d P=%d %s %s att=%d nnz/row %.1f  %s" % ("ok  " if ok else "FAIL", B, N, P, kind, "f32out" if f32out else "bf16out",
                                                             want_att, nnz / (B * N), msg), flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
