"""Host time of every C-ABI call of the batch-1 step (ctypes call duration = HIP launch overhead + the library's host code) against
the Python around them.  python tools/lat_host_calls.py [N]"""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
net = DecentralPlannerGATNet(cfg).to(dev).eval()
x, S = fov_states(1, N).to(dev), comm_gso(1, N, 20 if N <= 20 else 50, dtype=torch.float64).to(dev)
real = nat.lib()
acc = collections.defaultdict(lambda: [0, 0.0])


class Proxy:
    def __getattr__(self, name):
        f = getattr(real, name)

        def w(*a):
            t0 = time.perf_counter()
            r = f(*a)
            e = acc[name]
            e[0] += 1
            e[1] += time.perf_counter() - t0
            return r
        return w


with torch.no_grad():
    for _ in range(50):
        net.addGSO(S); net(x).cpu()
    torch.cuda.synchronize()
    px = Proxy()
    nat_lib = nat.lib
    nat.lib = lambda: px
    import magat_pathplanning_amd.graphml as gm
    import magat_pathplanning_amd.planner as pl
    steps = 400
    acc.clear()
    t0 = time.perf_counter()
    for _ in range(steps):
        net.addGSO(S); y = net(x); y.cpu()
    tot = (time.perf_counter() - t0) / steps * 1e6
    nat.lib = nat_lib
ctot = 0.0
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("  %-40s %5.2f calls/step  %7.2f us/step" % (k, c / steps, t / steps * 1e6))
    ctot += t / steps * 1e6
print("N=%d step %.1f us (with the timing proxy), C-ABI calls %.1f us, everything else (Python, torch, copy + sync) %.1f us" % (N, tot, ctot, tot - ctot))
