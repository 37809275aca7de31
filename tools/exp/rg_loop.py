"""Stress of the range-guard path of the one-launch GAT layer: the pytest sequence (other shapes first), fresh layers per iteration."""
import sys, os, ctypes
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_gat_mfma import _layer_and_inputs
from magat_pathplanning_amd import _native as nat
from oracle import magat_oracle as orc
dev = torch.device("cuda:0")
lib = nat.lib()
bad = 0
ref = {}
for it in range(150):
    # something else first, like the test file does
    l2, S2, x2 = _layer_and_inputs(700 if it % 2 else 5, 20 if it % 2 else 64, 3, 4, True, seed=3)
    l2 = l2.to(dev).eval(); l2.addGSO(S2.unsqueeze(1).to(dev))
    with torch.no_grad():
        l2(x2.to(dev))
    B, N, K, P = 4, 100, 3, 4
    layer, S, x = _layer_and_inputs(B, N, K, P, True, seed=21 + it)
    x[1, 5, 7] = 9.0e4
    with torch.no_grad():
        for prm in layer.parameters():
            prm.mul_(0.05)
        layer.weight.mul_(1e-6)
    y_ref, _ = orc.gat_layer_forward(x, S.unsqueeze(1), {k: v.detach() for k, v in layer.state_dict().items()}, "KeyQuery", True)
    layer = layer.to(dev).eval()
    layer.addGSO(S.unsqueeze(1).to(dev))
    scale = float(y_ref.abs().max())
    with torch.no_grad():
        y = layer(x.to(dev)).cpu()
    st = (ctypes.c_int32 * 2)()
    lib.magat_gat_read_status(nat.ptr(layer._scratch.workspace), st, nat.current_stream(dev))
    err = float((y - y_ref).abs().max())
    if not (err < 2e-6 * max(scale, 1.0)):
        bad += 1
        d = (y - y_ref).abs()
        idx = torch.nonzero(~(d < 2e-6 * scale))
        print("iter", it, "err", err, "scale", scale, "nbad", idx.shape[0], "first", idx[:3].tolist(), "nan", int(torch.isnan(y).sum()),
              "status", st[0], st[1], flush=True)
print("bad", bad, "of 150")
