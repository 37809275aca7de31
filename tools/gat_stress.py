"""Randomised cross-check of the persistent dense GAT kernel against the independent CSR kernels (same layer, different
code path) over many shapes / densities / modes - a race or prefetch-ordering bug in the persistent kernel shows up as a
mismatch.  GPU only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, GraphFilterBatchAttentional_Origin
from magat_pathplanning_amd.graphml import dense_gso_to_csr, gat_forward_rows, gat_forward_rows_csr
from magat_pathplanning_amd.synthetic import comm_gso, random_gso

dev = torch.device("cuda:0")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 40
worst = 0.0
for it in range(trials):
    mode = ["KeyQuery", "GAT_modified", "GAT_origin"][int(rng.integers(0, 3))]
    G = int(rng.choice([64, 128]))
    N = int(rng.integers(40, 129)) if G == 128 else int(rng.integers(40, 129))
    K = int(rng.integers(1, 5))
    P = int(rng.choice([1, 2, 4]))
    B = int(rng.choice([8, 260, 300, 520]))
    concat = bool(rng.integers(0, 2))
    cls = GraphFilterBatchAttentional_Origin if mode == "GAT_origin" else GraphFilterBatchAttentional
    layer = cls(G, G, K, P, concatenate=concat, attentionMode=mode).to(dev).eval()
    X = torch.randn(B, N, G, device=dev) * 0.6
    if rng.integers(0, 2):
        S = comm_gso(B, N, int(6 * N ** 0.5), seed=int(rng.integers(1 << 30)))
    else:
        S = random_gso(B, N, float(rng.choice([0.03, 0.1, 0.4, 1.0])), seed=int(rng.integers(1 << 30)))
    S = S.to(dev)
    with torch.no_grad():
        ya = gat_forward_rows(X, S, layer)[0].clone()
        yb = gat_forward_rows(X, S, layer)[0].clone()                       # run-to-run determinism
        rowptr, colidx, nnz = dense_gso_to_csr(S.contiguous(), self_loops=mode == "GAT_origin")
        yc = gat_forward_rows_csr(X, rowptr, colidx, nnz, layer)[0]
    torch.cuda.synchronize()
    assert torch.equal(ya, yb), ("non-deterministic", it, mode, B, N, G, K, P)
    err = float((ya - yc).abs().max())
    worst = max(worst, err)
    assert err <= 5e-5 * max(1.0, float(yc.abs().max())), ("mismatch", it, mode, B, N, G, K, P, concat, err)
print("ok: %d random cases, worst |dense - csr| = %.2e" % (trials, worst))
