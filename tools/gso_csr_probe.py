"""Times the GSO -> CSR structure kernels at config 5's shape (B=128, N=1000) on the device."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import _native as nat
from magat_pathplanning_amd.graphml import CsrStructure, dense_gso_to_csr
from magat_pathplanning_amd.synthetic import comm_gso

dev = torch.device("cuda:0")
B, N = 128, 1000
S = comm_gso(B, N, 160, seed=1).to(dev)
lib = nat.lib()
stream = nat.current_stream(dev)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


st = CsrStructure()
print("csr_build (mask + totals + structure): %.1f us" % timeit(lambda: st.build(S, 0, scrub_nan=1)))
deg = torch.empty(B * N, dtype=torch.int32, device=dev)
print("old row_degrees: %.1f us" % timeit(lambda: lib.magat_gso_row_degrees(nat.ptr(S), 0, 0, nat.ptr(deg), B, N, stream)))
print("old gso_prepare (scrub): %.1f us" % timeit(lambda: lib.magat_gso_prepare(nat.ptr(S), 0, S.numel(), 1, 0, stream)))
print("torch S.abs().sum(): %.1f us" % timeit(lambda: S.abs().sum()))
print("torch copy: %.1f us" % timeit(lambda: S.clone()))
