"""Batch-1 step under rocprofv3 --kernel-trace: which launches a closed-loop step is made of and what each costs.
   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt -o t -- python $REPO/tools/lat_trace.py 10"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
net = DecentralPlannerGATNet(cfg).to(dev).eval()
x, S = fov_states(1, N).to(dev), comm_gso(1, N, 20 if N <= 20 else 50, dtype=torch.float64).to(dev)
with torch.no_grad():
    for _ in range(300):
        net.addGSO(S)
        net(x).cpu()
torch.cuda.synchronize()
