"""Step time of mid-size batches with the encoder head as split-K float32 (option HEAD_SPLITK = 5120, the default) against the long-K
f16x3 head at every size (HEAD_SPLITK = 0: the arithmetic of large batches - logits then never depend on the batch size)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
for (B, N, mw) in ((6, 100, 50), (8, 100, 50), (64, 10, 20), (12, 100, 50), (20, 100, 50), (32, 100, 50), (40, 100, 50), (51, 100, 50), (256, 20, 28)):
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
    net = DecentralPlannerGATNet(cfg).to(dev).eval()
    x, S = fov_states(B, N).to(dev), comm_gso(B, N, mw).to(dev)
    res = []
    for sk in (5120, 0):
        nat.set_option("MAGAT_HEAD_SPLITK", sk)
        with torch.no_grad():
            for _ in range(30):
                net.addGSO(S); net(x)
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    net.addGSO(S); net(x)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 10)
        res.append(best)
    nat.reset_option("MAGAT_HEAD_SPLITK")
    print("B=%3d N=%3d (%5d agents): split-K head %.1f us/step | long-K head %.1f us/step (%+.1f %%)" % (B, N, B * N, res[0], res[1], (res[1] / res[0] - 1) * 100), flush=True)
