"""Batch-1 step with head MEAN at the default width (128 features, K = 3, P = 4, skip-concat) - what the reference's main.py runs unless
--AttentionConcat is given.  python tools/exp/lat_mean_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config

dev = torch.device("cuda:0")
for concat in (True, False):
    for N, m in ((10, 20), (100, 50)):
        cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", AttentionConcat=concat,
                          device=str(dev))
        torch.manual_seed(0)
        net = DecentralPlannerGATNet(cfg).to(dev).eval()
        x, S = fov_states(1, N, seed=17).to(dev), comm_gso(1, N, m, seed=18, dtype=torch.float64).to(dev)
        with torch.no_grad():
            for _ in range(40):
                net.addGSO(S); net(x).cpu()
            ts = []
            for _ in range(300):
                t0 = time.perf_counter()
                net.addGSO(S); net(x).cpu()
                ts.append((time.perf_counter() - t0) * 1e6)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                net.addGSO(S); net(x)
            e1.record(); torch.cuda.synchronize()
        ts.sort()
        print("heads %-6s N %3d: median %.1f us/step, device back to back %.1f us" % ("concat" if concat else "mean", N, ts[150], e0.elapsed_time(e1) * 5), flush=True)
