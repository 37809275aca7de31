"""Is the GAT graph kernel's fixed ~40 us per launch (198 us at B=512 vs 356 us at B=1024) a load-imbalance tail?
Times the layer on the bench GSOs, on B copies of ONE instance (perfect balance), and on instances sorted by edge count."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.graphml import gat_forward_rows
from magat_pathplanning_amd.synthetic import comm_gso
B, N = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0"); lib = nat.lib()
X = torch.randn(B, N, 128, device=dev)
S0 = comm_gso(B, N, 50, seed=1).to(dev)
nnz = (S0 != 0).sum(dim=(1, 2)).float()
print("edges per instance: mean %.0f  std %.0f  min %.0f  max %.0f" % (nnz.mean(), nnz.std(), nnz.min(), nnz.max()))
order = torch.argsort(nnz, descending=True)
W = 256
snake = torch.empty_like(order)
for r in range(B // W):          # round r of workgroup w takes sorted[r*W + w] (even r) or sorted[r*W + W-1-w] (odd r)
    blk = order[r * W:(r + 1) * W]
    snake[r * W:(r + 1) * W] = blk if r % 2 == 0 else blk.flip(0)
layer = GraphFilterBatchAttentional(128, 128, 3, 4, concatenate=True, attentionMode="KeyQuery").to(dev).eval()
for name, S in (("bench GSOs", S0), ("one instance replicated", S0[int(order[B // 2])].unsqueeze(0).expand(B, N, N).contiguous()),
                ("sorted by edges (desc)", S0[order].contiguous()), ("snake order", S0[snake].contiguous())):
    with torch.no_grad():
        for _ in range(3):
            gat_forward_rows(X, S, layer)
        lib.magat_profile_reserve(256); lib.magat_profile_reset(); lib.magat_profile_enable(1)
        for _ in range(10):
            gat_forward_rows(X, S, layer)
        torch.cuda.synchronize(); lib.magat_profile_enable(0); lib.magat_profile_collect()
    c, t = ctypes.c_longlong(0), ctypes.c_double(0)
    lib.magat_profile_read(11, ctypes.byref(c), ctypes.byref(t))
    print("%-26s gat_graph %.1f us" % (name, t.value * 1e3 / 10))
