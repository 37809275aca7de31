"""Batch-1 step of the published F-32-P4 shape (bottleneck 32, K = 2, P = 4, head-mean) at N = 10 .. 100: median step, device time, kernel tags."""
import os, sys, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
lib = nat.lib()
for N in (10, 30, 50, 100):
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=4, bottleneckFeature=32, AttentionConcat=False, bottleneckMode="BottomNeck_only")
    net = DecentralPlannerGATNet(cfg).to(dev).eval()
    x, S = fov_states(1, N).to(dev), comm_gso(1, N, 20 if N <= 20 else 50, dtype=torch.float64).to(dev)
    with torch.no_grad():
        for _ in range(30):
            net.addGSO(S); net(x).cpu()
        ts = []
        for _ in range(300):
            t0 = time.perf_counter(); net.addGSO(S); net(x).cpu(); ts.append((time.perf_counter() - t0) * 1e6)
        big = torch.randn(4096, 4096, device=dev)
        for _ in range(3): big @ big
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            net.addGSO(S); net(x)
        e1.record(); torch.cuda.synchronize()
        lib.magat_profile_reset(); lib.magat_profile_enable(1)
        for _ in range(20):
            net.addGSO(S); net(x)
        torch.cuda.synchronize(); lib.magat_profile_enable(0); lib.magat_profile_collect()
        tags = {}
        for tag, name in nat.TAGS.items():
            c, ms = ctypes.c_longlong(0), ctypes.c_double(0)
            lib.magat_profile_read(tag, ctypes.byref(c), ctypes.byref(ms))
            if c.value: tags[name] = "%dx %.1f us" % (c.value // 20, ms.value * 1e3 / c.value)
        lib.magat_profile_reset()
    print("N=%3d median %.1f us device %.1f us  %s" % (N, sorted(ts)[150], e0.elapsed_time(e1) * 10, tags), flush=True)
