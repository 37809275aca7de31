"""The action head as its own launch (skinny_gemm_kernel<5>: 51 200 rows x (128 + 512) inputs -> 5): device time per launch.
A/B of SKINNY_BATCH through variant libraries (MAGAT_LIB_PATH).  python tools/exp/skinny_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from magat_pathplanning_amd import _native as nat

dev = torch.device("cuda:0")
lib = nat.lib()
for M, Cin, C2 in ((51200, 128, 512), (51200, 128, 128), (100, 128, 512)):
    a, b = torch.randn(M, Cin, device=dev), torch.randn(M, C2, device=dev)
    w, bias, out = torch.randn(5, Cin + C2, device=dev) * 0.05, torch.randn(5, device=dev), torch.empty(M, 5, device=dev)
    d = nat.ConvGemmDesc()
    d.inp, d.Cin, d.lda = a.data_ptr(), Cin, Cin
    d.in2, d.C2, d.lda2, d.W2, d.stride2 = b.data_ptr(), C2, C2, 1, 1
    d.wt, d.bias, d.out = w.data_ptr(), bias.data_ptr(), out.data_ptr()
    d.M, d.Hin, d.Win, d.kH, d.kW, d.stride, d.pad, d.Hout, d.Wout = M, 1, 1, 1, 1, 1, 0, 1, 1
    d.Cout, d.ldc, d.relu, d.tag = 5, 5, 0, nat.TAG_ACTIONS
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(5):
        nat.check(lib.magat_conv_gemm_f32(ctypes.byref(d), st), "gemm")
    torch.cuda.synchronize()
    best = 1e9
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        e0.record()
        for _ in range(50):
            lib.magat_conv_gemm_f32(ctypes.byref(d), st)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    ref = torch.cat([a, b], 1) @ w.t() + bias
    print("M %6d K %3d + %3d: %.2f us per launch (%.0f GB/s), max err %.1e" % (M, Cin, C2, best, M * (Cin + C2) * 4 / best / 1e3, float((out - ref).abs().max())), flush=True)
