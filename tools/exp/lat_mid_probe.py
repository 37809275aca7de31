"""Step time of mid-size calls (a few hundred agents) with the one-agent-per-workgroup encoder against the eight-agent-group kernels:
where LAT_AGENTS should end.  python tools/exp/lat_mid_probe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
for (B, N, mw) in ((64, 10, 20), (3, 100, 50), (4, 100, 50), (5, 100, 50), (6, 100, 50), (8, 100, 50), (10, 100, 50), (16, 100, 50), (100, 10, 20), (200, 10, 20)):
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    net = DecentralPlannerGATNet(cfg).to(dev).eval()
    x, S = fov_states(B, N).to(dev), comm_gso(B, N, mw, dtype=torch.float64).to(dev)
    res = []
    for lat in (256, 4096):
        nat.set_option("MAGAT_LAT_AGENTS", lat)
        with torch.no_grad():
            for _ in range(30):
                net.addGSO(S); net(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            big = torch.randn(4096, 4096, device=dev)
            for _ in range(3): big @ big
            e0.record()
            for _ in range(100):
                net.addGSO(S); net(x)
            e1.record(); torch.cuda.synchronize()
            dev_us = e0.elapsed_time(e1) * 10
            ts = []
            for _ in range(200):
                t0 = time.perf_counter()
                net.addGSO(S); net(x).cpu()
                ts.append((time.perf_counter() - t0) * 1e6)
        res.append((sorted(ts)[100], dev_us))
    nat.reset_option("MAGAT_LAT_AGENTS")
    print("B=%3d N=%3d (%4d agents): eight-agent kernels median %.1f us (device %.1f) | one agent per workgroup %.1f us (device %.1f)" % (
        B, N, B * N, res[0][0], res[0][1], res[1][0], res[1][1]), flush=True)
