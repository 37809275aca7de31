"""Random DecentralPlannerNet (GNN baseline) configurations: HIP inference against the CPU oracle (test infrastructure)."""
import os, sys, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from magat_pathplanning_amd import DecentralPlannerNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
from oracle import magat_oracle as orc
dev = torch.device("cuda:0")
count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(count):
    N = rng.choice([1, 2, 5, 10, 20, 33, 64, 100, 128, 129, 200])
    B = rng.choice([1, 2, 3])
    K = rng.choice([1, 2, 3, 4])
    cnn = rng.choice(["ResNetLarge_withMLP", "ResNetSlim_withMLP", "ResNetLarge", "ResNetSlim", "Default"])
    f64 = rng.choice([True, False])
    tag = "B=%d N=%d K=%d %s f64=%s" % (B, N, K, cnn, f64)
    print("try ", tag, flush=True)
    try:
        cfg = make_config(device="cuda:0", num_agents=N, nGraphFilterTaps=K, CNN_mode=cnn)
        torch.manual_seed(300 + it)
        net = DecentralPlannerNet(cfg).eval()
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        x = fov_states(B, N, seed=it)
        S = comm_gso(B, N, 20 if N <= 20 else 50, seed=it + 1, dtype=torch.float64 if f64 else torch.float32)
        ref = orc.planner_gnn_forward(x, S.clone(), sd, cfg).numpy()
        net = net.to(dev)
        with torch.no_grad():
            net.addGSO(S.clone().to(dev))
            got = net(x.to(dev)).cpu().numpy()
        err = float(np.abs(got - ref).max())
        ok = err <= 1e-4 * max(1.0, float(np.abs(ref).max()))
        bad += 0 if ok else 1
        print("%s err %.2e %s" % ("ok  " if ok else "BAD ", err, tag), flush=True)
    except Exception as e:
        bad += 1
        print("RAISE %s -> %s" % (tag, repr(e)[:160]), flush=True)
print("failures:", bad, "of", count)
