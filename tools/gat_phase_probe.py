"""Per-phase cycle breakdown of gat_dense_kernel at a benchmark shape (instrumentation; GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.graphml import gat_forward_rows
from magat_pathplanning_amd.synthetic import comm_gso

B, N, K, P, G = int(sys.argv[1]), int(sys.argv[2]), 3, 4, 128
dev = torch.device("cuda:0")
layer = GraphFilterBatchAttentional(G, G, K, P, attentionMode="KeyQuery").to(dev).eval()
X = torch.randn(B, N, G, device=dev)
S = comm_gso(B, N, {100: 50, 20: 28, 10: 20}.get(N, 50), seed=1).to(dev)
lib = nat.lib()
grid = (B + 7) // 8 * 8 * P
buf = torch.zeros(grid, 8, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        gat_forward_rows(X, S, layer)
    lib.magat_gat_set_debug_buffer(buf.data_ptr())
    gat_forward_rows(X, S, layer)
    torch.cuda.synchronize()
    lib.magat_gat_set_debug_buffer(None)
d = buf.cpu().double()
d = d[d[:, 0] > 0]
names = ["stage(loads+LDS)", "scores+softmax", "U->LDS", "hop1", "hop2(last)+store"]
seg = [d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], None, None, None]
if K == 3:
    seg[2] = torch.zeros_like(seg[0])
    seg[3] = d[:, 3] - d[:, 2]
    seg[4] = d[:, 6] - d[:, 3]
tot = d[:, 6] - d[:, 0]
print("blocks", d.shape[0], "total cycles mean %.0f" % tot.mean().item())
for n, s_ in zip(names, seg):
    if s_ is not None:
        print("  %-18s mean %8.0f  p10 %8.0f  p90 %8.0f" % (n, s_.mean().item(), s_.quantile(0.1).item(), s_.quantile(0.9).item()))
if d[:, 5].max() > 0:
    print("  instance prologue (masks, x_i, first tile wait): mean %.0f cycles per instance, %.1f instances per workgroup"
          % ((d[:, 4] / d[:, 5].clamp_min(1)).mean().item(), d[:, 5].mean().item()))
wall = d[:, 7]
print("kernel span (100MHz wall clock): %.1f us" % ((wall.max() - wall.min()).item() / 100.0))
first = d[:, 0].min()
order = torch.argsort(d[:, 0])
starts = (d[order, 0] - first)
print("block start cycles quantiles:", [int(starts.quantile(q).item()) for q in (0.1, 0.25, 0.5, 0.75, 0.9, 1.0)])
