"""Run-to-run determinism and batch-composition independence of the encoder kernels (a race in the LDS hand-overs of the chain
kernels, a stale prefetch in their persistent loops or a dependence on a neighbour group would show here):
  * the same batch twice -> bit-identical logits, for many batch sizes (incl. sizes that leave a partial 8-agent group and
    fewer / more groups than CUs),
  * an instance's logits do not depend on what else is in the batch (its rows inside a big batch == alone).
PYTHONPATH=. python tools/encoder_stress.py [rounds]"""
import sys

import torch

from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
bad = 0
for N in (10, 13, 100):
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device="cuda:0")
    torch.manual_seed(N)
    net = DecentralPlannerGATNet(cfg).to(dev).eval()
    for B in (1, 2, 3, 7, 33, 64, 205, 512, 700):
        if B * N > 80000:
            continue
        x, S = fov_states(B, N, seed=B).to(dev), comm_gso(B, N, max(8, N // 2), seed=B + 1).to(dev)
        with torch.no_grad():
            net.addGSO(S)
            ref = net(x).clone()
            for r in range(rounds):
                net.addGSO(S)
                y = net(x)
                if not torch.equal(y, ref):
                    bad += 1
                    print("NOT DETERMINISTIC N %d B %d round %d: max diff %.3g" % (N, B, r, float((y - ref).abs().max())))
            if B >= 3:
                for b in (0, B // 2, B - 1):
                    net.addGSO(S[b:b + 1].contiguous())
                    y1 = net(x[b:b + 1].contiguous())
                    d = float((y1 - ref[b * N:(b + 1) * N]).abs().max())
                    if d > 2e-5:          # (the head changes its summation form with the agent count: float32 rounding)
                        bad += 1
                        print("BATCH DEPENDENCE N %d B %d instance %d: %.3g" % (N, B, b, d))
    print("N %d done" % N)
print("problems:", bad)
