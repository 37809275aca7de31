"""Closed-loop-style latency: one planning instance per step (the reference's test_batch_size=1 usage,
agents/decentralplannerlocal_OnlineExpert_GAT.py:1030-1055).  Per-step wall times of addGSO + forward + copy of the logits to the
host; median AND mean.  The warm-up loop performs the same `.cpu()` as the timed loop: the first host copy of a new result
size costs ~90 ms in the runtime (a pinned staging buffer; tools/stall_probe.py) - that one-off was the "580 us mean against
191 us median" of profiles/r04j/latency.txt (90 ms / 200 steps = 450 us), not a tail of the step."""
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
            net.addGSO(S); y = net(x); y.cpu()
        torch.cuda.synchronize()
        ts = []
        for _ in range(400):
            t0 = time.perf_counter()
            net.addGSO(S); y = net(x); y.cpu()
            ts.append((time.perf_counter() - t0) * 1e6)
        # device time of the same step (hipEvent pair around 200 back-to-back steps, no host copy): what the launches cost
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            net.addGSO(S); y = net(x)
        e1.record()
        torch.cuda.synchronize()
        # ... and the step exactly as the reference's loop drives it (agents/decentralplannerlocal_OnlineExpert_GAT.py:1032-1046): the state
        # tensor and the float64 GSO arrive as HOST tensors every step and are moved with .to(device) before addGSO / forward
        xh, Sh = x.cpu(), S.cpu()
        for _ in range(20):
            net.addGSO(Sh.to(dev)); net(xh.to(dev)).cpu()
        th = []
        for _ in range(300):
            t0 = time.perf_counter()
            net.addGSO(Sh.to(dev)); y = net(xh.to(dev)); y.cpu()
            th.append((time.perf_counter() - t0) * 1e6)
    srt = sorted(ts)
    print("B=%d N=%3d  with host-resident inputs (two .to(device) copies per step, as the reference's loop): median %.1f us/step" % (B, N, sorted(th)[150]))
    print("B=%d N=%3d  median %.1f us/step  mean %.1f  p90 %.1f  p99 %.1f  max %.1f   back-to-back (no host copy) %.1f us/step" % (
        B, N, srt[200], sum(ts) / len(ts), srt[360], srt[396], srt[-1], e0.elapsed_time(e1) * 1000 / 200))
