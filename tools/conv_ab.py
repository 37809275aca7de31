"""Interleaved A/B timing of the f16x3 direct conv kernel under different runtime switches (GPU only).
usage: python tools/conv_ab.py "MAGAT_CONV_PRIO=0" "MAGAT_CONV_PRIO=1" ...   (each arg: comma-separated VAR=VALUE list)
Every round times each configuration once (median of 3 launches) - box drift hits all configurations alike."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import _native as nat
from magat_pathplanning_amd.encoder import split_f16x2
M = 51200
dev = torch.device("cuda:0"); lib = nat.lib()
cfgs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
shapes = (("l1.conv2+ds", 32, 32, 32, 6), ("l2.conv2+ds", 64, 64, 32, 6), ("l3.conv1", 64, 128, 0, 6), ("l3.conv2+ds", 128, 128, 64, 6))
def taps(n):
    one = sum(sum(1 for t in range(3) if 0 <= o - 1 + t < n) for o in range(n)); return one * one
for name, cin, cout, c2, hw in shapes:
    npix = hw * hw
    x = torch.relu(torch.randn(npix, M, cin, device=dev)); x2 = torch.relu(torch.randn(npix, M, max(c2, 8), device=dev))
    w = torch.randn(cout, 9 * cin + c2, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    ws = split_f16x2(w)[0].to(dev); out = torch.empty(npix, M, cout, device=dev)
    d = nat.ConvGemmDesc()
    d.inp, d.wt, d.bias, d.out = x.data_ptr(), ws.data_ptr(), b.data_ptr(), out.data_ptr()
    d.in_pix_stride, d.out_pix_stride = M * cin, M * cout
    d.M, d.Cin, d.lda, d.Hin, d.Win, d.kH, d.kW, d.stride, d.pad = M, cin, cin, hw, hw, 3, 3, 1, 1
    d.Hout, d.Wout, d.Cout, d.ldc, d.relu, d.in_fmt, d.in_gl, d.out_gl = hw, hw, cout, cout, 1, 4, 1, 1
    if c2:
        d.in2, d.in2_pix_stride, d.C2, d.lda2, d.W2, d.stride2 = x2.data_ptr(), M * c2, c2, c2, hw, 1
    st = nat.current_stream(dev)
    res = [[] for _ in cfgs]; ref = None; same = True
    for rnd in range(10):
        for ci, cfg in enumerate(cfgs):
            for k, v in cfg.items(): os.environ[k] = v
            ts = []
            for r in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); nat.check(lib.magat_conv_gemm_f32(ctypes.byref(d), st), name); e1.record()
                torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort(); 
            if rnd >= 2: res[ci].append(ts[1])
            if ref is None: ref = out.clone()
            elif rnd == 0: same = same and bool(torch.equal(ref, out))
            for k in cfg: os.environ.pop(k, None)
    fl = 2.0 * M * (taps(hw) * cin * cout + npix * c2 * cout)
    print(name, " identical outputs:", same)
    for cfg, r in zip(cfgs, res):
        r.sort(); med = r[len(r) // 2]
        print("   %-44s median %8.1f us  min %8.1f  max %8.1f   %.1f TF-equiv" % (",".join("%s=%s" % kv for kv in cfg.items()) or "(default)", med, r[0], r[-1], fl / med / 1e6))
