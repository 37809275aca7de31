"""Closed-loop-style latency: one planning instance per step (the reference's test_batch_size=1 usage)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
for (B, N, mw) in ((1, 10, 20), (1, 100, 50), (8, 100, 50)):
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    net = DecentralPlannerGATNet(cfg).to(dev).eval()
    x, S = fov_states(B, N).to(dev), comm_gso(B, N, mw, dtype=torch.float64).to(dev)
    with torch.no_grad():
        for _ in range(20):
            net.addGSO(S); y = net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            net.addGSO(S); y = net(x); y.cpu()
        dt = (time.perf_counter() - t0) / 200
        # the product's own graph mode: enable_hip_graph() (capture on the first call of a shape, then replay)
        net.enable_hip_graph(True)
        try:
            for _ in range(5):
                net.addGSO(S); sy = net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                net.addGSO(S); sy = net(x); sy.cpu()
            dg = (time.perf_counter() - t0) / 200
            ok = torch.equal(sy, y)
        except Exception as e:
            dg, ok = float("nan"), repr(e)[:120]
        net.enable_hip_graph(False)
    print("B=%d N=%3d  eager %.1f us/step   hipGraph replay %.1f us/step   same=%s" % (B, N, dt * 1e6, dg * 1e6, ok))
