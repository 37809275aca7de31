"""Runs only the GAT layer (maps GEMM + graph kernel) a few times; target for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional
from magat_pathplanning_amd.graphml import gat_forward_rows
from magat_pathplanning_amd.synthetic import comm_gso
B, N = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
layer = GraphFilterBatchAttentional(128, 128, 3, 4, attentionMode="KeyQuery").to(dev).eval()
X = torch.randn(B, N, 128, device=dev)
S = comm_gso(B, N, {100: 50, 20: 28, 10: 20}.get(N, 50), seed=1).to(dev)
with torch.no_grad():
    for _ in range(reps):
        gat_forward_rows(X, S, layer)
torch.cuda.synchronize()
