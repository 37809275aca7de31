"""Head of the ResNet encoder (AvgPool2d(2) + fc + Flatten + Linear = one GEMM over the 6x6x128 map): the float32-MFMA
kernel that sum-pools on load (K = 9*128) against the f16x3 direct kernel on the un-pooled map with the cell weights
repeated for the four pixels of a cell (K = 36*128, 4x the multiplies, but 16-bit matrix-core rate).  GPU only."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import _native as nat
from magat_pathplanning_amd.encoder import split_f16x2
M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
dev = torch.device("cuda:0"); lib = nat.lib()
PERM = torch.tensor([16 * (q >> 4) + 8 * ((q & 7) >> 2) + 4 * ((q >> 3) & 1) + (q & 3) for q in range(32)])
Mp = (M + 127) // 128 * 128
def to_pl(t):        # [npix][Mp][C] float32 -> f16 plane granules [npix][Mp/128][2][C/8][128][8]
    c = t.shape[-1]; t = t.clamp(-65504.0, 65504.0)
    idx = (torch.arange(c) // 32) * 32 + PERM.repeat(c // 32)
    tp = t[..., idx.to(t.device)]
    h1 = tp.half(); h2 = (tp - h1.float()).half()
    pl = torch.stack((h1, h2), dim=1)
    return pl.view(t.shape[0], 2, Mp // 128, 128, c // 8, 8).permute(0, 2, 1, 4, 3, 5).contiguous()
torch.manual_seed(0)
x = torch.relu(torch.randn(36, Mp, 128, device=dev))
wc = torch.randn(128, 9 * 128, device=dev) * 0.02; b = torch.randn(128, device=dev)
# (a) pooled float32 kernel
outa = torch.empty(M, 128, device=dev)
da = nat.ConvGemmDesc()
da.inp, da.wt, da.bias, da.out = x.data_ptr(), wc.data_ptr(), b.data_ptr(), outa.data_ptr()
da.in_pix_stride = Mp * 128
da.M, da.Cin, da.lda, da.Hin, da.Win, da.kH, da.kW, da.stride, da.pad = M, 128, 128, 3, 3, 3, 3, 1, 0
da.Hout, da.Wout, da.Cout, da.ldc, da.relu, da.pool, da.pool_w = 1, 1, 128, 128, 0, 1, 6
# (b) direct f16x3 on the 6x6 map, weights of cell (iy/2, ix/2) for pixel (iy, ix)
w6 = torch.empty(128, 36, 128, device=dev)
for iy in range(6):
    for ix in range(6):
        cell = (iy // 2) * 3 + ix // 2
        w6[:, iy * 6 + ix] = wc[:, cell * 128:(cell + 1) * 128]
w6 = w6.reshape(128, 36 * 128)
kidx = (torch.arange(w6.shape[1]) // 32) * 32 + PERM.repeat(w6.shape[1] // 32)
wsp = split_f16x2(w6[:, kidx.to(dev)])[0].to(dev)
xp = to_pl(x)
outb = torch.empty(M, 128, device=dev)
db = nat.ConvGemmDesc()
db.inp, db.wt, db.bias, db.out = xp.data_ptr(), wsp.data_ptr(), b.data_ptr(), outb.data_ptr()
db.in_pix_stride = Mp * 128
db.M, db.Cin, db.lda, db.Hin, db.Win, db.kH, db.kW, db.stride, db.pad = M, 128, 128, 6, 6, 6, 6, 1, 0
db.Hout, db.Wout, db.Cout, db.ldc, db.relu, db.in_fmt, db.in_gl, db.out_gl = 1, 1, 128, 128, 0, 4, 2, 0
st = nat.current_stream(dev)
def t(d, n=12):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); nat.check(lib.magat_conv_gemm_f32(ctypes.byref(d), st), "head"); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort(); return ts[len(ts) // 2], ts[0]
ref = (x[:, :M].view(3, 2, 3, 2, M, 128).sum(dim=(1, 3)).permute(2, 0, 1, 3).reshape(M, 9 * 128).double() @ wc.double().t() + b.double())
for tm in ("2", "1"):
    os.environ["MAGAT_CONV_TM"] = tm
    ta = t(da); tb = t(db)
    print("M=%d TM=%s  pooled f32 kernel %.1f us (min %.1f)   direct f16x3 6x6 %.1f us (min %.1f)   max|a-ref| %.2e  max|b-ref| %.2e  (max|ref| %.1f)" % (
        M, tm, ta[0], ta[1], tb[0], tb[1], (outa.double() - ref).abs().max().item(), (outb.double() - ref).abs().max().item(), ref.abs().max().item()))
