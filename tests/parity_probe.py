"""Max |logits_hip - logits_oracle| at a benchmark-like shape, with and without the bf16x6 layer-3 path."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
from oracle import magat_oracle as orc
dev = torch.device("cuda:0")
cfg = make_config(num_agents=100, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
for seed in (1, 2, 3):
    sd = orc.init_state_dict(cfg, seed=seed)
    x, S = fov_states(6, 100, seed=seed), comm_gso(6, 100, 50, seed=seed + 10)
    ref = orc.planner_forward(x, S.clone(), sd, cfg).double()
    net = DecentralPlannerGATNet(cfg); net.load_state_dict(sd); net = net.to(dev).eval()
    with torch.no_grad():
        net.addGSO(S.to(dev)); got = net(x.to(dev)).cpu().double()
    print("seed %d: max|dlogits| = %.3e   (|logits| max %.3f)  argmax agreement %.4f" % (
        seed, (got - ref).abs().max().item(), ref.abs().max().item(), (got.argmax(1) == ref.argmax(1)).float().mean().item()))
