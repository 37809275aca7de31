"""The 128-wide row-tile graph kernel (gat_mid.hip, XR form) against what ran before it at 103 .. 128 agents (two launches up to
105, the CSR kernels above), and - debug build only, where it accepts 97 .. 102 agents - against gat_mfma.hip at c3's own size.
python tools/exp/wide_mid_probe.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
G, K, P = 128, 3, 4


def timed(layer, x, reps=20):
    with torch.no_grad():
        for _ in range(5):
            layer(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record()
            for _ in range(reps):
                y = layer(x)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best, y


for N in (100, 103, 105, 106, 116, 128):
    for concat in (True, False):
        torch.manual_seed(N)
        layer = GraphFilterBatchAttentional(G, G, K, P, attentionMode="KeyQuery", concatenate=concat).to(dev).eval()
        x = (torch.randn(B, G, N) * 0.5).to(dev)
        S = comm_gso(B, N, 50, seed=N).to(dev)
        layer.addGSO(S.unsqueeze(1))
        nat.lib().magat_form_reset()
        t_new, y_new = timed(layer, x)
        mid = int(nat.lib().magat_form_count(nat.FORMS["gat_mid"])) > 0
        nat.set_option("GAT_MFMA", 0)
        layer.addGSO(S.unsqueeze(1))
        t_old, y_old = timed(layer, x)
        nat.reset_option("GAT_MFMA")
        err = float((y_new - y_old).abs().max())
        print("B %d N %3d %-6s: default %7.1f us (%s) | GAT_MFMA=0 (two launches / CSR) %7.1f us | max diff %.2e"
              % (B, N, "concat" if concat else "mean", t_new, "gat_mid" if mid else "gat_mfma", t_old, err), flush=True)

# whole forward at 400 x 128 agents: the one-launch graph layer against the CSR kernels it replaces there (GAT_WIDE_FROM = 129)
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import fov_states, make_config
for Bm, N in ((400, 128), (1, 128)):
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device=str(dev))
    torch.manual_seed(0)
    net = DecentralPlannerGATNet(cfg).to(dev).eval()
    x, S = fov_states(Bm, N, seed=1).to(dev), comm_gso(Bm, N, 50, seed=2, dtype=torch.float64).to(dev)
    out = {}
    for label, opt in (("one launch", None), ("CSR kernels", 129)):
        if opt is not None:
            nat.set_option("GAT_WIDE_FROM", opt)
        with torch.no_grad():
            for _ in range(6):
                net.addGSO(S); y = net(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for _ in range(3):
                e0.record()
                for _ in range(20):
                    net.addGSO(S); y = net(x)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 20)
        out[label] = (best, y.float().cpu())
        nat.reset_option("GAT_WIDE_FROM")
    d = float((out["one launch"][1] - out["CSR kernels"][1]).abs().max())
    print("model %d x %d agents: step %.3f ms = %.2f M agent-steps/s (one launch) | %.3f ms = %.2f M (CSR kernels) | max logit diff %.2e"
          % (Bm, N, out["one launch"][0], Bm * N / out["one launch"][0] / 1e3, out["CSR kernels"][0], Bm * N / out["CSR kernels"][0] / 1e3, d))
