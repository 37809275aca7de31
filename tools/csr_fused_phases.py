"""Phase cycles of the fused CSR kernels (csrc/gat_csr_fused.hip) at BASELINE config 5's shape, from a -DFUSED_STAMPS build:
  tools/build_variant.sh stamps gat_csr_fused.hip -DFUSED_STAMPS
  MAGAT_LIB_PATH=.../libmagat_hip_stamps.so python tools/csr_fused_phases.py
Per wave, summed over its steps: cycles between stamps (s_memtime), mean / max over the waves that worked."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.graphml import CsrStructure, gat_forward_rows_csr
from magat_pathplanning_amd.synthetic import comm_gso

B, N, G, K, P = 128, 1000, 128, 2, 4
dev = torch.device("cuda:0")
torch.manual_seed(0)
layer = GraphFilterBatchAttentional(G, G, K, P, attentionMode="KeyQuery").to(dev).eval()
X = (torch.randn(B, N, G, device=dev) * 0.5).to(torch.bfloat16)
S = comm_gso(B, N, 160, seed=2).to(dev)
st = CsrStructure().build(S, 0)
nnz = st.ready(dev)
csc = (st.cscptr, st.csc[0], st.csc[1])
out = torch.empty(B * N, P * G, dtype=torch.bfloat16, device=dev)
lib = nat.lib()
WAVES = 8
dbg = torch.zeros(2 * 8 * WAVES * 4096, dtype=torch.int64, device=dev)
for _ in range(3):
    gat_forward_rows_csr(X, st.rowptr, st.colidx, nnz, layer, out=out, csc=csc)
torch.cuda.synchronize()
lib.magat_csr_fused_set_debug.argtypes = [ctypes.c_void_p]
lib.magat_csr_fused_set_debug(dbg.data_ptr())
gat_forward_rows_csr(X, st.rowptr, st.colidx, nnz, layer, out=out, csc=csc)
torch.cuda.synchronize()
lib.magat_csr_fused_set_debug(None)
d = dbg.cpu().view(2, 4096, WAVES, 8).double()
names = {0: ["prologue (order, pointers, own row, first gather)", "q' MFMA + pack", "edge loop", "normalise", "", "", "weights -> LDS", "kernel total"],
         1: ["prologue (order, pointers, first indices / values)", "edge loop", "own row", "tap phase: rest (z pack, bias requests)",
             "tap phase: matrix chains (16 MFMAs each)", "tap phase: epilogues (bias, relu, bf16, LDS stage, stores)", "weights -> LDS",
             "kernel total"]}
for k, title in ((0, "score kernel"), (1, "hop + tap kernel")):
    w = d[k].reshape(-1, 8)
    w = w[w[:, 7] > 0]
    print("== %s: %d waves" % (title, w.shape[0]))
    for i, nm in enumerate(names[k]):
        if nm:
            print("   %-52s mean %9.0f  max %9.0f cycles" % (nm, w[:, i].mean(), w[:, i].max()))
