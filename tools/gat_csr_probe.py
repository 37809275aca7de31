import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.graphml import gat_forward_rows, gat_forward_rows_csr, dense_gso_to_csr
from magat_pathplanning_amd.synthetic import comm_gso
B, N = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
layer = GraphFilterBatchAttentional(128, 128, 3, 4, attentionMode="KeyQuery").to(dev).eval()
X = torch.randn(B, N, 128, device=dev)
S = comm_gso(B, N, {100: 50, 20: 28, 10: 20}.get(N, 50), seed=1).to(dev)
rowptr, colidx, nnz = dense_gso_to_csr(S)
print("nnz per instance", nnz / B)
lib = nat.lib()
def run(fn, reps=10):
    for _ in range(3): fn()
    lib.magat_profile_reset(); lib.magat_profile_enable(1)
    for _ in range(reps): fn()
    torch.cuda.synchronize(); lib.magat_profile_enable(0); lib.magat_profile_collect()
    import ctypes
    for tag in (10, 11, 14):
        c, t = ctypes.c_longlong(0), ctypes.c_double(0)
        lib.magat_profile_read(tag, ctypes.byref(c), ctypes.byref(t))
        if c.value: print("   tag", nat.TAGS[tag], "launches", c.value, "us per call-set %.1f" % (t.value * 1e3 / reps))
with torch.no_grad():
    print("dense kernel"); run(lambda: gat_forward_rows(X, S, layer))
    print("csr kernels"); run(lambda: gat_forward_rows_csr(X, rowptr, colidx, nnz, layer))
