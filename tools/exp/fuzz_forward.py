"""Random model configurations through the HIP inference path against the CPU oracle (test infrastructure: a sweep, not a product
path).   python tools/exp/fuzz_forward.py [count] [seed] [fov]

With `fov` the field of view is drawn from 5 .. 15 as well.  Only CNN_mode = Default sizes itself from the FOV; the ResNet modes
carry the reference's fixed 1152-wide compress layer (graphs/models/decentralplanner_GAT_bottleneck_SkipConcatGNN.py:96-118), so
at FOV != 9 they RAISE torch's "mat1 and mat2 shapes cannot be multiplied" exactly as the reference's module does, and Default at
FOV 5 raises torch's "Output size is too small" from its third pooling: those lines are the reference's error behaviour, counted
as failures by the summary line but not defects (seed 78: 46 ok, 34 such raises, nothing else)."""
import os, sys, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
from oracle import magat_oracle as orc
dev = torch.device("cuda:0")
count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(count):
    N = rng.choice([1, 2, 3, 5, 8, 10, 17, 31, 32, 33, 50, 64, 100, 102, 103, 106, 117, 128, 129, 150])
    B = rng.choice([1, 2, 3, 5]) if N < 100 else rng.choice([1, 2])
    G = rng.choice([16, 32, 64, 128])
    K = rng.choice([1, 2, 3, 4])
    P = rng.choice([1, 2, 4])
    att = rng.choice(["KeyQuery", "GAT_modified", "GAT_origin"])
    skip = rng.choice(["BottomNeck_only", "BottomNeck_skipConcat", "BottomNeck_skipConcatGNN", "BottomNeck_skipAddGNN", ""])
    cnn = rng.choice(["ResNetLarge_withMLP", "ResNetSlim_withMLP", "ResNetLarge", "ResNetSlim", "Default"])
    if skip == "BottomNeck_skipAddGNN" and cnn.endswith("_withMLP"):
        cnn = "Default"            # (that reference file has no *_withMLP branch: the product maps it to Default, the oracle's init does not)
    concat = rng.choice([True, False]) if skip != "BottomNeck_skipAddGNN" else False
    f64 = rng.choice([True, False])
    fov = rng.choice([9, 9, 9, 5, 7, 11, 13, 15]) if len(sys.argv) > 3 and sys.argv[3] == "fov" else 9
    kw = dict(FOV=fov, num_agents=N, nGraphFilterTaps=K, nAttentionHeads=P, bottleneckFeature=G, bottleneckMode=skip, CNN_mode=cnn,
              attentionMode=att, AttentionConcat=concat)
    print("try  B=%d N=%d G=%d K=%d P=%d %s %s %s concat=%s f64=%s" % (B, N, G, K, P, att, skip or "legacy", cnn, concat, f64) + " fov=%d" % fov, flush=True)
    try:
        cfg = make_config(device="cuda:0", **kw)
        sd = orc.init_state_dict(cfg, seed=100 + it)
        x = fov_states(B, N, seed=it, fov=fov)
        S = comm_gso(B, N, 20 if N <= 20 else 50, seed=it + 1, dtype=torch.float64 if f64 else torch.float32)
        ref = orc.planner_forward(x, S.clone(), sd, cfg).numpy()
        net = DecentralPlannerGATNet(cfg)
        net.load_state_dict(sd)
        net = net.to(dev).eval()
        with torch.no_grad():
            net.addGSO(S.clone().to(dev))
            first = net(x.to(dev))
            net.addGSO(S.clone().to(dev))
            second = net(x.to(dev))              # (round 6: the step plan of the host side takes over from the second forward on)
            got = first.cpu().numpy()
        err = float(np.abs(got - ref).max())
        ok = err <= 1e-4 * max(1.0, float(np.abs(ref).max())) and bool(torch.equal(first, second))
        if not ok:
            bad += 1
        print("%s err %.2e  B=%d N=%d G=%d K=%d P=%d %s %s %s concat=%s f64=%s" % ("ok  " if ok else "BAD ", err, B, N, G, K, P, att, skip or "legacy", cnn, concat, f64), flush=True)
    except Exception as e:
        bad += 1
        print("RAISE B=%d N=%d G=%d K=%d P=%d %s %s %s concat=%s -> %s" % (B, N, G, K, P, att, skip or "legacy", cnn, concat, repr(e)[:160]))
print("failures:", bad, "of", count)
