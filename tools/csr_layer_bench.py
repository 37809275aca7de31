"""The bf16-storage graph layer at BASELINE config 5's shape (128 x 1000 agents, K=2, P=4, G=F=128) on the CSR kernels: per-tag
kernel times (maps GEMM, scores + hop).  MAGAT_LIB_PATH picks a build (tools/build_variant.sh ... -DCSR_WHATIF_*)."""
import sys, torch
import os
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
out = torch.empty(B * N, P * G, dtype=torch.float32, device=dev)
lib = nat.lib()
for _ in range(3):
    gat_forward_rows_csr(X, st.rowptr, st.colidx, nnz, layer, out=out, csc=csc)
torch.cuda.synchronize()
lib.magat_profile_reset(); lib.magat_profile_enable(1)
for _ in range(10):
    gat_forward_rows_csr(X, st.rowptr, st.colidx, nnz, layer, out=out, csc=csc)
torch.cuda.synchronize()
lib.magat_profile_collect()
lib.magat_profile_enable(0)
import ctypes
for tag, name in ((10, "maps"), (11, "graph (scores, hop)")):
    ms, cnt = ctypes.c_double(), ctypes.c_longlong()
    lib.magat_profile_read(tag, ctypes.byref(cnt), ctypes.byref(ms))
    print("%-22s %8.1f us per launch  x%d   nnz/row %.1f" % (name, ms.value * 1e3 / max(cnt.value, 1), cnt.value, nnz / (B * N)))
