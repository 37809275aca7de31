"""Hunts the rare mismatch of the attention layer's HIP backward (seen once in ~40 suite runs on
test_gat_training_backward_matches_autograd[GAT_modified-False-40-128-3-2]): repeats the test body for all seven parameter
sets in pytest's order, many times, and on a mismatch prints which tensor, which rows, and whether a second backward on the
same inputs reproduces it.   PYTHONPATH=. python tools/train_bwd_hunt.py [rounds]"""
import sys

import torch

from magat_pathplanning_amd import GraphFilterBatchAttentional, GraphFilterBatchAttentional_Origin
from magat_pathplanning_amd.graphml import _composite
from magat_pathplanning_amd.synthetic import comm_gso

CASES = [("KeyQuery", True, 12, 64, 3, 2), ("KeyQuery", False, 20, 128, 2, 4), ("GAT_modified", True, 9, 32, 4, 3),
         ("KeyQuery", True, 30, 16, 1, 2), ("GAT_modified", False, 40, 128, 3, 2), ("GAT_origin", True, 14, 32, 3, 4),
         ("GAT_origin", False, 25, 64, 2, 2)]
dev = torch.device("cuda:0")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
refs = {}
bad = 0
for rnd in range(rounds):
    for case in CASES:
        mode, concat, N, G, K, P = case
        B = 3
        origin = mode == "GAT_origin"
        cls = GraphFilterBatchAttentional_Origin if origin else GraphFilterBatchAttentional
        if case not in refs:
            g = torch.Generator().manual_seed(N * 7 + G)
            ref = cls(G, G, K, P, concatenate=concat, attentionMode=mode).double()
            with torch.no_grad():
                if not origin:
                    ref.weight_bias.uniform_(-0.3, 0.3, generator=g)
            x = (torch.randn(B, G, N, generator=g) * 0.6).double().requires_grad_(True)
            S = comm_gso(B, N, max(6, int(4 * N ** 0.5)), seed=N, dtype=torch.float64)
            S[0, 2, :] = 0
            S[1, 3, 5], S[1, 5, 3] = 0.7, 0.0
            wgt = torch.randn(B, P * G if concat else G, N, generator=g).double()
            y_ref, _ = _composite(ref, x, S.unsqueeze(1))
            (y_ref * wgt).sum().backward()
            refs[case] = (ref, x, S, wgt, y_ref.detach())
        ref, x, S, wgt, y_ref = refs[case]
        layer = cls(G, G, K, P, concatenate=concat, attentionMode=mode)
        layer.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
        layer = layer.to(dev).train()
        xg = x.detach().float().to(dev).requires_grad_(True)
        layer.addGSO(S.unsqueeze(1).to(dev))
        y = layer(xg)
        (y * wgt.float().to(dev)).sum().backward()
        torch.cuda.synchronize()
        pairs = [("y", y, y_ref), ("dx", xg.grad, x.grad)] + [(n, p.grad, getattr(ref, n).grad) for n, p in layer.named_parameters()
                                                               if p.grad is not None and getattr(ref, n).grad is not None]
        for name, a, b in pairs:
            a, b = a.detach().cpu().double(), b.detach().double()
            scale = max(1.0, float(b.abs().max()))
            err = (a - b).abs()
            if float(err.max()) > 2e-4 * scale:
                bad += 1
                idx = (err > 2e-4 * scale).nonzero()
                print("MISMATCH round %d case %s tensor %s max %.3g at %d elements; first %s" % (rnd, case, name, float(err.max()), len(idx), idx[:6].tolist()))
                # again, same module and inputs
                layer.zero_grad()
                xg2 = x.detach().float().to(dev).requires_grad_(True)
                y2 = layer(xg2)
                (y2 * wgt.float().to(dev)).sum().backward()
                torch.cuda.synchronize()
                print("   repeat on the same module: dx max err %.3g" % float((xg2.grad.cpu().double() - x.grad).abs().max()))
                break
print("rounds %d x %d cases, mismatches: %d" % (rounds, len(CASES), bad))
