"""Where do the one-off 50-90 ms host stalls of a closed loop come from?  Replays tools/latency_probe.py's sequence and prints
every step slower than 2 ms with its position, with the garbage collector's activity beside it."""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
gc_events = []
def cb(phase, info):
    if phase == "start":
        cb.t0 = time.perf_counter()
    else:
        gc_events.append((info["generation"], (time.perf_counter() - cb.t0) * 1e3, info.get("collected", 0)))
gc.callbacks.append(cb)
for (B, N, mw) in ((1, 10, 20), (1, 100, 50), (8, 100, 50)):
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    net = DecentralPlannerGATNet(cfg).to(dev).eval()
    x, S = fov_states(B, N).to(dev), comm_gso(B, N, mw, dtype=torch.float64).to(dev)
    with torch.no_grad():
        for mode in ("eager", "graph"):
            net.enable_hip_graph(mode == "graph")
            for _ in range(20):
                net.addGSO(S); y = net(x)
            torch.cuda.synchronize()
            for it in range(200):
                n0 = len(gc_events)
                t0 = time.perf_counter()
                net.addGSO(S); y = net(x); y.cpu()
                dt = (time.perf_counter() - t0) * 1e3
                if dt > 2.0:
                    print("B=%d N=%d %s step %d: %.1f ms   gc during the step: %s" % (B, N, mode, it, dt, gc_events[n0:]))
        net.enable_hip_graph(False)
print("gc generation-2 collections in total:", [e for e in gc_events if e[0] == 2])
