"""Where the batch-1 step's HOST time goes: enqueue time of addGSO / forward (no synchronisation), the copy, the step
with a synchronisation per step, and a cProfile of the loop.  python tools/lat_host_probe.py [N]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
net = DecentralPlannerGATNet(cfg).to(dev).eval()
x, S = fov_states(1, N).to(dev), comm_gso(1, N, 20 if N <= 20 else 50, dtype=torch.float64).to(dev)
med = lambda v: sorted(v)[len(v) // 2]
with torch.no_grad():
    for _ in range(50):
        net.addGSO(S); net(x).cpu()
    torch.cuda.synchronize()
    a, f, c, tot = [], [], [], []
    for _ in range(400):
        t0 = time.perf_counter()
        net.addGSO(S)
        t1 = time.perf_counter()
        y = net(x)
        t2 = time.perf_counter()
        y.cpu()
        t3 = time.perf_counter()
        a.append((t1 - t0) * 1e6); f.append((t2 - t1) * 1e6); c.append((t3 - t2) * 1e6); tot.append((t3 - t0) * 1e6)
    print("N=%d median us: addGSO enqueue %.1f  forward enqueue %.1f  .cpu() %.1f  step %.1f" % (N, med(a), med(f), med(c), med(tot)))
    # device-only: the same launches with the host far ahead (events around 200 steps enqueued after a long kernel)
    big = torch.randn(8192, 8192, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        (big @ big)
    torch.cuda.synchronize()
    for _ in range(6):
        (big @ big)           # ~tens of ms of device work: the host enqueues the 100 steps behind it
    e0.record()
    for _ in range(100):
        net.addGSO(S); y = net(x)
    e1.record()
    torch.cuda.synchronize()
    print("N=%d device-bound step (host ahead): %.1f us" % (N, e0.elapsed_time(e1) * 1000 / 100))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300):
        net.addGSO(S); net(x).cpu()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
