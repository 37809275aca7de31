"""Cycles per phase of gat_mid_kernel (the row-tile one-launch graph kernel) - DEBUG build only (build_native --debug;
MAGAT_ALLOW_EXPERIMENT_BUILD=1).  python tools/exp/mid_phase_probe.py B N G K [wide_from]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso

B, N, G, K = (int(a) for a in sys.argv[1:5])
P = 4
dev = torch.device("cuda:0")
lib = nat.lib()
fn = lib.magat_gat_mid_set_debug_buffer      # (debug build: MAGAT_LIB_PATH=.../libmagat_hip_debug.so)
fn.argtypes = [ctypes.c_void_p]
fn.restype = ctypes.c_int
torch.manual_seed(0)
layer = GraphFilterBatchAttentional(G, G, K, P, attentionMode="KeyQuery", concatenate=True).to(dev).eval()
x = (torch.randn(B, G, N) * 0.5).to(dev)
S = comm_gso(B, N, 50, seed=1).to(dev)
layer.addGSO(S.unsqueeze(1))
buf = torch.zeros(4096, 4, 16, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        layer(x)
    lib.magat_form_reset()
    fn(buf.data_ptr())
    layer(x)
    torch.cuda.synchronize()
    fn(None)
print("forms:", {k: int(lib.magat_form_count(v)) for k, v in nat.FORMS.items() if lib.magat_form_count(v)})
d = buf.cpu().double()
d = d[d[:, 0, 0] > 0]          # workgroups that ran
nt = (N + 31) // 32
d = d[:, :nt]
names = ["G1 + Q planes", "barrier", "G2 products", "softmax", "A planes", "G3 (first tap | all)", "barrier", "hops (+U^T planes, taps)", "epilogue", "barrier"]
print("workgroups %d, waves %d; cycles per phase of the last head (mean over workgroups and waves | wave 0 | last wave)" % (d.shape[0], nt))
tot = 0.0
for i, n in enumerate(names):
    seg = d[:, :, i + 1] - d[:, :, i]
    tot += seg.mean().item()
    print("  %-26s %8.0f | %8.0f | %8.0f" % (n, seg.mean().item(), seg[:, 0].mean().item(), seg[:, nt - 1].mean().item()))
print("  head total                 %8.0f" % tot)
