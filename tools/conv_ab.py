"""Interleaved A/B timing of the f16x3 direct conv kernel under different runtime switches (GPU only).
usage: python tools/conv_ab.py "LAYOUT=2,MAGAT_CONV_TM=1" "LAYOUT=2,MAGAT_CONV_TM=2" ...   (each arg: comma-separated VAR=VALUE list)
Every round times each configuration once (median of 3 launches) - box drift hits all configurations alike."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import _native as nat
from magat_pathplanning_amd.encoder import split_f16x2
M = 51200
dev = torch.device("cuda:0"); lib = nat.lib()
cfgs = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
# pseudo-switch LAYOUT=0 row-major float32, 1 float32 granules (default), 2 f16 plane granules (+ K-permuted weights)
PERM = torch.tensor([16 * (q >> 4) + 8 * ((q & 7) >> 2) + 4 * ((q >> 3) & 1) + (q & 3) for q in range(32)])
def to_gl(t):        # [npix][M][C] -> float32 granules [npix][M/128][C/4][128][4]
    return t.view(t.shape[0], M // 128, 128, t.shape[-1] // 4, 4).permute(0, 1, 3, 2, 4).contiguous()
def from_gl(t, c):
    return t.view(t.shape[0], M // 128, c // 4, 128, 4).permute(0, 1, 3, 2, 4).reshape(t.shape[0], M, c)
def to_pl(t):        # [npix][M][C] float32 -> f16 plane granules [npix][M/128][2][C/8][128][8]
    c = t.shape[-1]; t = t.clamp(-65504.0, 65504.0)
    idx = (torch.arange(c) // 32) * 32 + PERM.repeat(c // 32)
    tp = t[..., idx.to(t.device)]                     # operand order within every 32-channel tile
    h1 = tp.half(); h2 = (tp - h1.float()).half()
    pl = torch.stack((h1, h2), dim=1)                  # [npix][2][M][C]
    return pl.view(t.shape[0], 2, M // 128, 128, c // 8, 8).permute(0, 2, 1, 4, 3, 5).contiguous()
def from_pl(buf, npix, c):   # inverse of to_pl on a float32-typed buffer of the same bytes
    pl = buf.view(torch.float16).view(npix, M // 128, 2, c // 8, 128, 8).permute(0, 2, 1, 4, 3, 5).reshape(npix, 2, M, c)
    v = pl[:, 0].float() + pl[:, 1].float()
    idx = (torch.arange(c) // 32) * 32 + PERM.repeat(c // 32)
    out = torch.empty_like(v); out[..., idx.to(v.device)] = v
    return out

shapes = (("gat_maps-like (1 pixel, N=2048)", 128, 2048, 0, 1), ("l1.conv2+ds", 32, 32, 32, 6), ("l2.conv1", 32, 64, 0, 6), ("l2.conv2+ds", 64, 64, 32, 6), ("l3.conv1", 64, 128, 0, 6), ("l3.conv2+ds", 128, 128, 64, 6))
def taps(n):
    one = sum(sum(1 for t in range(3) if 0 <= o - 1 + t < n) for o in range(n)); return one * one
for name, cin, cout, c2, hw in shapes:
    npix = hw * hw
    x = torch.relu(torch.randn(npix, M, cin, device=dev)); x2 = torch.relu(torch.randn(npix, M, max(c2, 32), device=dev))
    w = torch.randn(cout, 9 * cin + c2, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    ws = split_f16x2(w)[0].to(dev); out = torch.empty(npix, M, cout, device=dev)
    kidx = (torch.arange(w.shape[1]) // 32) * 32 + PERM.repeat(w.shape[1] // 32)
    wsp = split_f16x2(w[:, kidx.to(dev)])[0].to(dev)
    ins = {0: (x, x2), 1: (to_gl(x), to_gl(x2)), 2: (to_pl(x), to_pl(x2))}
    d = nat.ConvGemmDesc()
    d.inp, d.wt, d.bias, d.out = x.data_ptr(), ws.data_ptr(), b.data_ptr(), out.data_ptr()
    d.in_pix_stride, d.out_pix_stride = M * cin, M * cout
    d.M, d.Cin, d.lda, d.Hin, d.Win, d.kH, d.kW, d.stride, d.pad = M, cin, cin, hw, hw, 3, 3, 1, 1
    d.Hout, d.Wout, d.Cout, d.ldc, d.relu, d.in_fmt = hw, hw, cout, cout, 1, 4
    if c2:
        d.in2_pix_stride, d.C2, d.lda2, d.W2, d.stride2 = M * c2, c2, c2, hw, 1
    st = nat.current_stream(dev)
    res = [[] for _ in cfgs]; ref = None; same = True
    for rnd in range(10):
        for ci, cfg in enumerate(cfgs):
            for k, v in cfg.items(): os.environ[k] = v
            lay = int(cfg.get("LAYOUT", 1)); d.in_gl = d.out_gl = lay
            d.inp = ins[lay][0].data_ptr(); d.in2 = ins[lay][1].data_ptr() if c2 else None
            d.wt = (wsp if lay == 2 else ws).data_ptr()
            ts = []
            for r in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); nat.check(lib.magat_conv_gemm_f32(ctypes.byref(d), st), name); e1.record()
                torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort(); 
            if rnd >= 2: res[ci].append(ts[1])
            if rnd == 0:
                got = out if lay == 0 else (from_gl(out, cout) if lay == 1 else from_pl(out, npix, cout))
                if ref is None: ref = got.clone()
                else:
                    # the plane layout rounds the OUTPUT to two f16 planes (2^-22 relative) and feeds the matrix core its
                    # 16 products per k step in another order (different fp32 rounding inside the MFMA)
                    tol = 0.0 if lay < 2 else 3e-6 * float(ref.abs().max())
                    same = same and bool((ref - got).abs().max() <= tol)
                    if lay == 2: print("      plane layout max|diff| %.3e (tol %.3e, max|ref| %.2f)" % ((ref - got).abs().max().item(), tol, ref.abs().max().item()))
            for k in cfg: os.environ.pop(k, None)
    fl = 2.0 * M * (taps(hw) * cin * cout + npix * c2 * cout)
    print(name, " outputs agree (float32 layouts bit-identical, plane layout to 3e-6 relative):", same)
    for cfg, r in zip(cfgs, res):
        r.sort(); med = r[len(r) // 2]
        print("   %-44s median %8.1f us  min %8.1f  max %8.1f   %.1f TF-equiv" % (",".join("%s=%s" % kv for kv in cfg.items()) or "(default)", med, r[0], r[-1], fl / med / 1e6))
