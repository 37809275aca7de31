"""Per-phase cycle breakdown of gat_mfma_kernel (debug-hooks build; GPU only):
  MAGAT_LIB_PATH=magat_pathplanning_amd/lib/libmagat_hip_debug.so python tools/gat_mfma_probe.py [B N K P]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.graphml import gat_forward_rows
from magat_pathplanning_amd.synthetic import comm_gso

B, N, K, P = (int(a) for a in (sys.argv[1:5] + ["512", "100", "3", "4"][len(sys.argv) - 1:]))
G = 128
dev = torch.device("cuda:0")
layer = GraphFilterBatchAttentional(G, G, K, P, attentionMode="KeyQuery").to(dev).eval()
X = torch.randn(B, N, G, device=dev)
S = comm_gso(B, N, {100: 50, 20: 28, 10: 20}.get(N, 50), seed=1).to(dev)
raw = ctypes.CDLL(nat.LIB_PATH)
grid = min(B, 256)
buf = torch.zeros(grid, 4, 16, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3):
        gat_forward_rows(X, S, layer)
    raw.magat_gat_mfma_set_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
    gat_forward_rows(X, S, layer)
    torch.cuda.synchronize()
    raw.magat_gat_mfma_set_debug_buffer(ctypes.c_void_p(0))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        gat_forward_rows(X, S, layer)
    t1.record(); torch.cuda.synchronize()
    print("layer: %.1f us per call" % (t0.elapsed_time(t1) * 1000 / 20))
d = buf.cpu().double()
names = ["W_p loads issued", "barrier (prev head)", "G1 + Q planes", "barrier", "G2 + softmax + A planes", "G3 (K taps)",
         "barrier", "hop 1 (U^T planes + product)", "hop 2", None, "epilogue"]
idx = [0, 1, 2, 3, 4, 5, 6, 7] + ([8] if K == 3 else []) + [10]
for wv in range(4):
    dd = d[:, wv, :]
    dd = dd[dd[:, 0] > 0]
    tot = dd[:, 10] - dd[:, 0]
    print("wave %d: last head total %.0f cycles" % (wv, tot.mean().item()))
    for a, b_ in zip(idx[:-1], idx[1:]):
        seg = dd[:, b_] - dd[:, a]
        print("   %-30s mean %8.0f  p10 %8.0f  p90 %8.0f" % (names[b_], seg.mean().item(), seg.quantile(0.1).item(), seg.quantile(0.9).item()))
    if wv == 0:
        print("   instance prologue (X planes, masks, barrier): %.0f cycles" % (dd[:, 15] - dd[:, 14]).mean().item())
        print("   (G1 product %.0f | Q planes %.0f;  G2 product %.0f | softmax %.0f | A planes %.0f)" % (
            (dd[:, 11] - dd[:, 1]).mean().item(), (dd[:, 2] - dd[:, 11]).mean().item(), (dd[:, 12] - dd[:, 3]).mean().item(),
            (dd[:, 13] - dd[:, 12]).mean().item(), (dd[:, 4] - dd[:, 13]).mean().item()))
