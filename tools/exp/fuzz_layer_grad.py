"""Random GraphFilterBatchAttentional(_Origin) configurations: HIP training forward / backward against float64 autograd through the
oracle restatement (test infrastructure).   python tools/exp/fuzz_layer_grad.py [count] [seed]"""
import os, sys, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, GraphFilterBatchAttentional_Origin
from magat_pathplanning_amd.synthetic import directed_gso
from oracle import magat_oracle as orc
dev = torch.device("cuda:0")
count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ACT = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "identity": lambda t: t}
bad = 0
for it in range(count):
    N = rng.choice([1, 2, 5, 10, 20, 33, 64, 100, 129, 150])
    B = rng.choice([1, 2, 3])
    G = rng.choice([16, 32, 64, 128])
    K = rng.choice([1, 2, 3, 4])
    P = rng.choice([1, 2, 4])
    E = rng.choice([1, 1, 2, 3])
    mode = rng.choice(["KeyQuery", "GAT_modified", "GAT_origin"])
    act = rng.choice(list(ACT))
    concat = rng.choice([True, False])
    tag = "B=%d N=%d G=%d K=%d P=%d E=%d %s %s concat=%s" % (B, N, G, K, P, E, mode, act, concat)
    print("try ", tag, flush=True)
    try:
        torch.manual_seed(2000 + it)
        cls = GraphFilterBatchAttentional_Origin if mode == "GAT_origin" else GraphFilterBatchAttentional
        layer = cls(G, G, K, P, E, True, nonlinearity=ACT[act], concatenate=concat, attentionMode=mode)
        with torch.no_grad():
            if mode != "GAT_origin":
                layer.weight_bias.uniform_(-0.3, 0.3)
        p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
        x = torch.randn(B, G, N) * 0.6
        S = torch.stack([directed_gso(B, N, 0.3 if N <= 32 else 0.08, seed=it + e, dtype=torch.float64) for e in range(E)], dim=1)
        wgt = torch.randn(B, P * G if concat else G, N)
        p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
        x64 = x.double().requires_grad_(True)
        y64, _ = orc.gat_layer_forward(x64, S, p64, mode, concat, nonlinearity=ACT[act])
        (y64 * wgt.double()).sum().backward()
        layer = layer.to(dev).train()
        xd = x.to(dev).requires_grad_(True)
        layer.addGSO(S.to(dev))
        y = layer(xd)
        (y * wgt.to(dev)).sum().backward()
        scale = max(1.0, float(y64.abs().max()))
        e_y = float((y.detach().cpu().double() - y64.detach()).abs().max()) / scale
        gs = max(1e-6, float(x64.grad.abs().max()))
        e_x = float((xd.grad.cpu().double() - x64.grad).abs().max()) / gs
        e_p = 0.0
        for k, v in layer.named_parameters():
            want = p64[k].grad if p64[k].grad is not None else torch.zeros_like(p64[k])
            got = v.grad.cpu().double() if v.grad is not None else torch.zeros_like(want)
            e_p = max(e_p, float((got - want).abs().max()) / max(1.0, float(want.abs().max())))
        ok = e_y < 1e-4 and e_x < 1e-3 and e_p < 1e-3
        bad += 0 if ok else 1
        print("%s y %.1e dx %.1e dp %.1e  %s" % ("ok  " if ok else "BAD ", e_y, e_x, e_p, tag), flush=True)
    except Exception as e:
        bad += 1
        print("RAISE %s -> %s" % (tag, repr(e)[:160]), flush=True)
print("failures:", bad, "of", count)
