"""bf16x6 split-MFMA vs fp32-MFMA conv on the layer-3 shapes (GPU only)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import _native as nat
from magat_pathplanning_amd.encoder import split_bf16x3, split_f16x2
M = int(sys.argv[1]) if len(sys.argv) > 1 else 51200
dev = torch.device("cuda:0"); lib = nat.lib()
def taps():
    one = sum(sum(1 for t in range(3) if 0 <= o - 1 + t < 6) for o in range(6)); return one * one
for name, cin, cout, c2 in (("l3.conv1", 64, 128, 0), ("l3.conv2+ds", 128, 128, 64)):
    x = torch.relu(torch.randn(36, M, cin, device=dev)); x2 = torch.relu(torch.randn(36, M, max(c2, 8), device=dev))
    w = torch.randn(cout, 9 * cin + c2, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    res = {}
    def f16_planes(t):
        t = t.clamp(-65504.0, 65504.0)
        h1 = t.half()
        return torch.stack((h1, (t - h1.float()).half())).contiguous()
    def to_gl(t):     # [36][M][C] row-major -> granule-major agent tiles [36][M/128][C/4][128][4]
        return t.view(36, M // 128, 128, t.shape[-1] // 4, 4).permute(0, 1, 3, 2, 4).contiguous()
    def from_gl(t, c):
        return t.view(36, M // 128, c // 4, 128, 4).permute(0, 1, 3, 2, 4).reshape(36, M, c)
    for fmt in (0, 1, 2, 4, 5, 6, 7, 8):
        d = nat.ConvGemmDesc()
        gl = fmt >= 6
        gli, glo = fmt in (6, 7), fmt in (6, 8)
        ofmt = fmt
        if gl: fmt = 4
        xs = split_bf16x3(x) if fmt == 1 else (f16_planes(x) if fmt == 5 else x); x2s = split_bf16x3(x2) if fmt == 1 else (f16_planes(x2) if fmt == 5 else x2)
        if gli: xs, x2s, d.in_gl = to_gl(x), to_gl(x2), 1
        if glo: d.out_gl = 1
        ws = w if fmt == 0 else (split_f16x2(w)[0].to(dev) if fmt >= 4 else split_bf16x3(w))
        out = torch.empty(36, M, cout, device=dev)
        d.inp, d.wt, d.bias, d.out = xs.data_ptr(), ws.data_ptr(), b.data_ptr(), out.data_ptr()
        d.in_pix_stride, d.out_pix_stride, d.in_plane_stride = M * cin, M * cout, 36 * M * cin
        d.M, d.Cin, d.lda, d.Hin, d.Win, d.kH, d.kW, d.stride, d.pad = M, cin, cin, 6, 6, 3, 3, 1, 1
        d.Hout, d.Wout, d.Cout, d.ldc, d.relu, d.in_fmt = 6, 6, cout, cout, 1, fmt
        if c2:
            d.in2, d.in2_pix_stride, d.C2, d.lda2, d.W2, d.stride2 = x2s.data_ptr(), M * c2, c2, c2, 6, 1
            d.in2_plane_stride = 36 * M * c2
        st = nat.current_stream(dev); ts = []
        for r in range(9):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); nat.check(lib.magat_conv_gemm_f32(ctypes.byref(d), st), name); e1.record()
            torch.cuda.synchronize()
            if r >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort(); res[ofmt] = (ts[len(ts) // 2], from_gl(out, cout).clone() if glo else out.clone())
    fl = 2.0 * M * (taps() * cin * cout + 36 * c2 * cout)
    ref64 = None
    print("%-12s f16x3 direct: gl-in only %8.1f us (%s)   gl-out only %8.1f us (%s)" % (name, res[7][0], bool(torch.equal(res[4][1], res[7][1])), res[8][0], bool(torch.equal(res[4][1], res[8][1]))))
    print("%-12s f16x3 granule-major tiles %8.1f us (%.1f TF-equiv), identical to row-major: %s" % (name, res[6][0], 2.0 * M * (taps() * cin * cout + 36 * c2 * cout) / res[6][0] / 1e6, bool(torch.equal(res[4][1], res[6][1]))))
    print("%-12s fp32 %8.1f us (%.1f TF)   bf16x6 planes %8.1f us (%.2fx)   bf16x6 split-on-load %8.1f us (%.1f TF-equiv, %.2fx)   f16x3 %8.1f us (%.1f TF-equiv, %.2fx)   f16x3 planes-in %8.1f us (%.2fx, identical output: %s)   max|diff vs fp32 kernel| %.2e %.2e %.2e  (out scale %.2f)" % (
        name, res[0][0], fl / res[0][0] / 1e6, res[1][0], res[0][0] / res[1][0], res[2][0], fl / res[2][0] / 1e6,
        res[0][0] / res[2][0], res[4][0], fl / res[4][0] / 1e6, res[0][0] / res[4][0],
        res[5][0], res[0][0] / res[5][0], bool(torch.equal(res[4][1], res[5][1])),
        (res[0][1] - res[1][1]).abs().max().item(), (res[0][1] - res[2][1]).abs().max().item(),
        (res[0][1] - res[4][1]).abs().max().item(), res[0][1].abs().max().item()))
    # accuracy against float64 on a slice of agents (pixel (2,2): all nine taps valid)
    xs64 = x[:, :256].double().cpu(); w64 = w.double().cpu(); acc = torch.zeros(256, cout, dtype=torch.float64)
    for ty in range(3):
        for tx in range(3):
            pix = (2 + ty - 1) * 6 + (2 + tx - 1)
            acc += xs64[pix] @ w64[:, (ty * 3 + tx) * cin:(ty * 3 + tx + 1) * cin].T
    if c2:
        acc += x2[2 * 6 + 2, :256].double().cpu() @ w64[:, 9 * cin:].T
    acc = torch.relu(acc + b.double().cpu())
    for fmt, nm in ((0, "fp32 MFMA"), (2, "bf16x6"), (4, "f16x3")):
        got = res[fmt][1][2 * 6 + 2, :256].double().cpu()
        print("      %-10s vs float64: max|err| %.2e  rms %.2e" % (nm, (got - acc).abs().max().item(), (got - acc).pow(2).mean().sqrt().item()))
