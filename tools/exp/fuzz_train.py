"""Fuzz of the TRAINING step (test infrastructure): loss.backward() through the planner in train() mode with the CNN on the HIP
training kernels (default) against the same step with torch's convolutions / BatchNorm (MAGAT_TRAIN_CNN=torch), from identical
weights, over random batch shapes (1 .. ~400 agents, incl. counts that are not multiples of the 8-agent group or the 128-agent
tile), CNN modes, skip variants and attention modes.  Prints one line per mismatch and a summary.

What it found (round 4): the two float32 backends agree to 1e-4 .. 5e-3 in every gradient except, in about one case in fifteen,
ONE weight gradient that differs by 1 - 5 % of its maximum.  Arbitrated against the same step in float64 on the CPU, one backend
is at 1e-6 and the other carries the whole difference - torch / MIOpen three times out of four, the HIP path otherwise - and the
difference sits in a single output channel: an activation within float32 rounding of zero passes its ReLU in one arithmetic
and not in the other, and with tens of agents one pixel's term is a percent of a weight's gradient.  Not a defect of either;
`--dense` (continuous inputs instead of the binary maps) shows the same, so it is not a max-pool tie either.

    python tools/exp/fuzz_train.py [cases] [seed]
"""
import copy
import os
import random
import sys

import torch
import torch.nn.functional as tnf

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from magat_pathplanning_amd import DecentralPlannerGATNet, DecentralPlannerNet          # noqa: E402
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config         # noqa: E402

DENSE = "--dense" in sys.argv
if DENSE:
    sys.argv.remove("--dense")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda:0")


def rel(a, b):
    a = a.double(); b = b.double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-6))


bad = 0
for case in range(cases):
    N = rng.choice([1, 2, 3, 5, 7, 10, 13, 20, 33])
    B = rng.choice([1, 2, 3, 5, 8, 11])
    kw = dict(num_agents=N, nGraphFilterTaps=rng.choice([1, 2, 3]), nAttentionHeads=rng.choice([1, 2, 4]),
              bottleneckMode=rng.choice(["BottomNeck_skipConcat", "BottomNeck_only", "BottomNeck_skipConcatGNN",
                                         "BottomNeck_skipAddGNN", ""]),
              CNN_mode=rng.choice(["ResNetLarge_withMLP", "ResNetSlim_withMLP", "ResNetLarge", "ResNetSlim", "Default"]),
              attentionMode=rng.choice(["GAT_modified", "KeyQuery", "GAT_origin"]), device="cuda:0")
    kw["AttentionConcat"] = kw["bottleneckMode"] != "BottomNeck_skipAddGNN"      # (that variant adds: needs the head mean)
    gnn = rng.random() < 0.2
    try:
        cfg = make_config(**kw)
    except TypeError as e:
        print("config", kw, e); break
    torch.manual_seed(case)
    base = (DecentralPlannerNet if gnn else DecentralPlannerGATNet)(cfg)
    for m in base.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    x = fov_states(B, N, seed=case).to(dev)
    if DENSE:             # continuous inputs: no two activations tie in a ReLU / max-pool (see the note at the end)
        x = torch.randn(x.shape, generator=torch.Generator().manual_seed(case)).to(dev)
    S = comm_gso(B, N, max(1, N // 2), seed=case + 1).to(dev)
    tgt = torch.randint(0, 5, (B * N,), generator=torch.Generator().manual_seed(case)).to(dev)
    res = {}
    try:
        for backend in ("hip", "torch"):
            os.environ["MAGAT_TRAIN_CNN"] = backend
            net = copy.deepcopy(base).to(dev).train()
            net.addGSO(S.clone())
            logits = net(x)
            loss = tnf.cross_entropy(logits, tgt)
            loss.backward()
            res[backend] = (logits.detach(), {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None},
                            {k: v.detach().clone() for k, v in net.state_dict().items() if v.dtype.is_floating_point})
    except Exception as e:                        # noqa: BLE001
        if B * N == 1 and "Expected more than 1 value per channel" in str(e):
            continue                              # torch's own BatchNorm refusal of a single 1x1 sample; not ours
        bad += 1
        print("case %d %s B=%d gnn=%d backend=%s: %s: %s" % (case, kw, B, gnn, backend, type(e).__name__, str(e)[:300]), flush=True)
        continue
    lh, gh, sh = res["hip"]; lt, gt, st = res["torch"]
    worst = ("logits", rel(lh, lt))
    if gh.keys() != gt.keys():
        bad += 1
        print("case %d %s: gradient key sets differ" % (case, kw), flush=True)
        continue
    for k in gh:
        # tolerance with an absolute floor: a conv bias in front of BatchNorm has a zero gradient up to rounding
        d = float((gh[k].double() - gt[k].double()).abs().max())
        sc = max(float(gt[k].abs().max()), 1e-3 * float(max(v.abs().max() for v in gt.values())))
        if d / sc > worst[1]:
            worst = (k, d / sc)
    for k in sh:
        r = rel(sh[k], st[k])
        if r > worst[1] and "num_batches" not in k:
            worst = ("state " + k, r)
    if not (worst[1] < 5e-3) or not torch.isfinite(lh).all():
        # which of the two float32 backends is off?  Arbitrate with the same step in float64 on the CPU (plain torch).
        note = ""
        if worst[0] in gh:
            os.environ["MAGAT_TRAIN_CNN"] = "torch"
            ref = (DecentralPlannerNet if gnn else DecentralPlannerGATNet)(make_config(**{**kw, "device": "cpu"}))
            ref.load_state_dict(base.state_dict())
            for m in ref.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
            ref = ref.double().train()
            os.environ["MAGAT_ALLOW_TORCH_COMPOSITE"] = "1"          # the checker's plain-torch float64 pass, CPU
            ref.addGSO(S.double().cpu())
            tnf.cross_entropy(ref(x.double().cpu()), tgt.cpu()).backward()
            del os.environ["MAGAT_ALLOW_TORCH_COMPOSITE"]
            g64 = dict(ref.named_parameters())[worst[0]].grad
            sc = max(float(g64.abs().max()), 1e-30)
            eh = float((gh[worst[0]].double().cpu() - g64).abs().max()) / sc
            et = float((gt[worst[0]].double().cpu() - g64).abs().max()) / sc
            gmax = max(float(v.abs().max()) for v in gt.values())
            # a ReLU whose input is within rounding of zero opens in one arithmetic and not in another: ONE output channel of the
            # convolution in front of it sees a different gradient.  Share of the squared difference in the worst channel:
            dd = (gh[worst[0]].double().cpu() - g64) if eh > et else (gt[worst[0]].double().cpu() - g64)
            if dd.dim() == 4:
                per = (dd ** 2).sum(dim=(1, 2, 3))
                note_ch = "; %.0f %% of the squared difference sits in output channel %d" % (100 * float(per.max() / per.sum()), int(per.argmax()))
            else:
                note_ch = ""
            note = "  vs float64: hip %.2e, torch %.2e (this gradient's max / the largest gradient's max: %.1e)%s" % (eh, et, sc / gmax, note_ch)
            one_channel = dd.dim() == 4 and float(per.max() / per.sum()) > 0.9
            if one_channel and eh > 3 * et + 1e-6:
                note += "  -> one ReLU within rounding of zero opened differently; here it is the HIP path that differs from float64"
                print("case %d B=%d N=%d %s: %s %.3e%s" % (case, B, N, kw["CNN_mode"], worst[0], worst[1], note), flush=True)
                continue
            if eh <= 3 * et + 1e-6:
                note += "  -> the HIP path is the one that agrees with float64" if et > 10 * eh else "  -> rounding, both backends alike"
                print("case %d B=%d N=%d %s: %s %.3e%s" % (case, B, N, kw["CNN_mode"], worst[0], worst[1], note), flush=True)
                continue
        bad += 1
        print("case %d %s B=%d gnn=%d: worst %s %.3e%s" % (case, kw, B, gnn, worst[0], worst[1], note), flush=True)
print("fuzz_train: %d cases, %d bad" % (cases, bad))
