import sys, os, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
B, N, mw = 1, 100, 50
cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
net = DecentralPlannerGATNet(cfg).to(dev).eval()
x, S = fov_states(B, N).to(dev), comm_gso(B, N, mw, dtype=torch.float64).to(dev)
with torch.no_grad():
    for _ in range(20):
        net.addGSO(S); y = net(x)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            net.addGSO(S); y = net(x); y.cpu()
        print("eager %.1f us/step" % ((time.perf_counter() - t0) / 200 * 1e6))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200):
        net.addGSO(S); y = net(x); y.cpu()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
