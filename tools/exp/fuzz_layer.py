"""Random GraphFilterBatchAttentional(_Origin) configurations through the HIP layer against the CPU oracle (test infrastructure).
   python tools/exp/fuzz_layer.py [count] [seed]"""
import os, sys, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, GraphFilterBatchAttentional_Origin
from magat_pathplanning_amd.synthetic import directed_gso, comm_gso
from oracle import magat_oracle as orc
dev = torch.device("cuda:0")
count = int(sys.argv[1]) if len(sys.argv) > 1 else 80
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(count):
    N = rng.choice([1, 2, 3, 7, 10, 20, 31, 32, 33, 64, 100, 102, 103, 105, 106, 113, 127, 128, 129, 200, 300])
    B = rng.choice([1, 2, 3, 4]) if N <= 128 else rng.choice([1, 2])
    G = rng.choice([16, 32, 64, 128, 256]) if N <= 128 else rng.choice([16, 32, 64, 128])
    F = G if rng.random() < 0.8 else rng.choice([16, 32, 64, 128])
    K = rng.choice([1, 2, 3, 4])
    P = rng.choice([1, 2, 3, 4])
    mode = rng.choice(["KeyQuery", "GAT_modified", "GAT_origin"])
    if mode == "GAT_origin":
        F = G
    concat = rng.choice([True, False])
    want_att = rng.random() < 0.35
    f64 = rng.choice([True, False])
    nin = N if rng.random() < 0.7 or N < 3 else rng.randint(1, N - 1)
    tag = "B=%d N=%d nin=%d G=%d F=%d K=%d P=%d %s concat=%s att=%s f64=%s" % (B, N, nin, G, F, K, P, mode, concat, want_att, f64)
    print("try ", tag, flush=True)
    try:
        torch.manual_seed(1000 + it)
        cls = GraphFilterBatchAttentional_Origin if mode == "GAT_origin" else GraphFilterBatchAttentional
        layer = cls(G, F, K, P, 1, True, concatenate=concat, attentionMode=mode)
        with torch.no_grad():
            if mode != "GAT_origin":
                layer.weight_bias.uniform_(-0.3, 0.3)
        p = {k: v.detach().clone() for k, v in layer.state_dict().items()}
        x = torch.randn(B, G, nin) * 0.7
        S = directed_gso(B, N, 0.3 if N <= 32 else 0.08, seed=it, dtype=torch.float64 if f64 else torch.float32).unsqueeze(1)
        ref, aref = orc.gat_layer_forward(x, S, p, mode, concat)
        layer = layer.to(dev).eval()
        layer.return_attention = want_att
        layer.addGSO(S.to(dev))
        with torch.no_grad():
            got = layer(x.to(dev)).cpu()
        err = float((got - ref).abs().max())
        ok = tuple(got.shape) == tuple(ref.shape) and err <= 1e-4 * max(1.0, float(ref.abs().max()))
        if want_att and ok:
            ea = float((layer.aij.cpu() - aref).abs().max())
            ok = ea <= 1e-5
        if not ok:
            bad += 1
        print("%s err %.2e  %s" % ("ok  " if ok else "BAD ", err, tag), flush=True)
    except Exception as e:
        msg = repr(e)[:140]
        soft = "NotImplementedError" in msg or "unsupported" in msg.lower()
        bad += 0 if soft else 1
        print("%s %s -> %s" % ("decl" if soft else "RAISE", tag, msg), flush=True)
print("failures:", bad, "of", count)
