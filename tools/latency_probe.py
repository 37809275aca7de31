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
        # graph replay
        g = torch.cuda.CUDAGraph()
        sx, sS = x.clone(), S.clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                net.addGSO(sS); net(sx)
        torch.cuda.current_stream().wait_stream(s)
        try:
            with torch.cuda.graph(g):
                net.addGSO(sS); sy = net(sx)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                sx.copy_(x); sS.copy_(S); g.replay(); sy.cpu()
            dg = (time.perf_counter() - t0) / 200
            ok = torch.equal(sy, y)
        except Exception as e:
            dg, ok = float("nan"), repr(e)[:120]
    print("B=%d N=%3d  eager %.1f us/step   graph replay %.1f us/step   same=%s" % (B, N, dt * 1e6, dg * 1e6, ok))
