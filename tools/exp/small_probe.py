"""gat_small_kernel (published widths, graphs of at most 32 agents): time of a layer call at the published F-32-P4 shape.
python tools/exp/small_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso

dev = torch.device("cuda:0")
for B, N, G, K, concat in ((512, 100, 32, 2, False), (512, 100, 32, 3, True), (512, 50, 32, 2, False), (1, 100, 32, 2, False), (1024, 20, 32, 2, False), (1024, 10, 32, 2, False), (4096, 20, 32, 2, False), (1024, 20, 64, 3, True), (1, 10, 32, 2, False), (1, 20, 32, 2, False)):
    torch.manual_seed(0)
    layer = GraphFilterBatchAttentional(G, G, K, 4, attentionMode="KeyQuery", concatenate=concat).to(dev).eval()
    x = (torch.randn(B, G, N) * 0.5).to(dev)
    S = comm_gso(B, N, 50 if N > 32 else 28, seed=1, dtype=torch.float64).to(dev)
    layer.addGSO(S.unsqueeze(1))
    with torch.no_grad():
        for _ in range(5):
            layer(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(50):
                layer(x)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
        lib = nat.lib()
        lib.magat_profile_reset(); lib.magat_profile_enable(1)
        for _ in range(50):
            layer(x)
        torch.cuda.synchronize(); lib.magat_profile_enable(0); lib.magat_profile_collect()
        c, ms = ctypes.c_longlong(0), ctypes.c_double(0)
        lib.magat_profile_read(19, ctypes.byref(c), ctypes.byref(ms))
        lib.magat_profile_reset()
    print("B %4d N %2d G %3d K %d %-6s: layer call %.1f us, kernel (profiler tag, %d launches) %.2f us" % (B, N, G, K, "concat" if concat else "mean", best, c.value, ms.value * 1e3 / max(1, c.value)), flush=True)
