"""Where the one-off 60-90 ms stall in a fresh model's first steps comes from: per-step wall times under variations."""
import os, sys, time, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
B, N = 1024, 20
def run(label, act_scale="1", nogc=False, pre_sleep=0.0, steps=40):
    os.environ["MAGAT_ACT_SCALE"] = act_scale
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    net = DecentralPlannerGATNet(cfg).to(dev).eval()
    x, S = fov_states(B, N).to(dev), comm_gso(B, N, 28).to(dev)
    if nogc:
        gc.collect(); gc.disable()
    ts = []
    with torch.no_grad():
        net.addGSO(S); net(x); torch.cuda.synchronize()
        if pre_sleep: time.sleep(pre_sleep)
        t00 = time.perf_counter()
        for i in range(steps):
            t0 = time.perf_counter()
            net.addGSO(S); net(x)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            ts.append((t0 - t00, (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
    if nogc: gc.enable()
    bad = [(i, "%.1f ms after start: enqueue %.2f ms, sync %.2f ms" % (t[0] * 1e3, t[1], t[2])) for i, t in enumerate(ts) if t[1] + t[2] > 5]
    print(label, "stalls:", bad if bad else "none", flush=True)
run("default            ")
run("no calibration     ", act_scale="0")
run("gc disabled        ", nogc=True)
run("sleep 0.3 s first  ", pre_sleep=0.3)
run("default again      ")
