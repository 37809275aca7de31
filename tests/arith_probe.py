"""NOT a test (no test_ prefix): sizes two cheaper split-arithmetic options against the 1e-4 logits gate, on the CPU oracle.
The f16x3 GEMM computes x.w as h1.g1 + h1.g2 + h2.g1 (x = h1 + h2, w = g1 + g2 in f16).  Options (DESIGN.md "what comes next" (6)):
  fp8corr : the two correction products with BOTH operands rounded to fp8 e4m3 (per-tensor power-of-two scale) - the form the
            gfx950 f8f6f4 MFMA could run at twice the f16 rate;
  mxfp8 / mxfp6_e2m3 / mxfp6_e3m2 / mxfp4 : the same with OCP MX block-scaled operands (blocks of 32 input channels, shared
            power-of-two scale) - v_mfma_scale_f32_32x32x64_f8f6f4 runs fp8 at 2x and fp6 / fp4 at 4x the f16 rate, and both
            corrections fit ONE K = 64 instruction per 32-channel slab ([h1|h2] against [g2;g1], one scale per K block);
  drop    : h2.g1 dropped (two products instead of three).
Applied to the 3x3 convolutions whose input has >= `cmin` channels; everything else stays float32.
usage: python tests/arith_probe.py            (prints max |dlogit| per option / layer set)"""
import os
import sys

import torch
import torch.nn.functional as tnf

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import magat_oracle as orc                                   # noqa: E402
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config   # noqa: E402

real_conv2d = tnf.conv2d


def split16(t):
    h1 = t.half().float()
    return h1, (t - h1).half().float()


def q8(t):
    m = float(t.abs().max())
    if m == 0.0:
        return t
    s = 2.0 ** torch.floor(torch.log2(torch.tensor(224.0 / m))).item()
    return (t * s).to(torch.float8_e4m3fn).float() / s


def _grid(ebits, mbits, bias, finite_max=None):
    vals = [0.0]
    for e in range(2 ** ebits):
        for m in range(2 ** mbits):
            v = (m / 2 ** mbits) * 2.0 ** (1 - bias) if e == 0 else (1 + m / 2 ** mbits) * 2.0 ** (e - bias)
            vals.append(v)
    g = torch.tensor(sorted(set(vals)), dtype=torch.float64)
    return g[g <= finite_max] if finite_max else g


# OCP MX element formats: (grid of non-negative values, exponent of the largest power of two)
MX = {"mxfp8": (_grid(4, 3, 7, 448.0), 8), "mxfp6_e2m3": (_grid(2, 3, 1), 2), "mxfp6_e3m2": (_grid(3, 2, 3), 4),
      "mxfp4": (_grid(2, 1, 1), 2)}


def qmx(t, fmt, dim):
    """Block-scaled rounding along `dim` in blocks of 32 (shared power-of-two scale per block, OCP MX)."""
    grid, emax = MX[fmt]
    t = t.double().movedim(dim, -1)
    shp = t.shape
    assert shp[-1] % 32 == 0
    b = t.reshape(-1, 32)
    amax = b.abs().amax(dim=1, keepdim=True).clamp_min(1e-300)
    scale = 2.0 ** (torch.floor(torch.log2(amax)) - emax)
    v = (b / scale).abs().clamp_max(float(grid[-1]))
    idx = torch.searchsorted(grid, v.contiguous()).clamp(1, len(grid) - 1)
    lo, hi = grid[idx - 1], grid[idx]
    q = torch.where(v - lo <= hi - v, lo, hi) * torch.sign(b) * scale
    return q.reshape(shp).movedim(-1, dim).float()


def make_conv(mode, cmin):
    def conv(x, w, b=None, stride=1, padding=0):
        if w.shape[-1] != 3 or w.shape[1] < cmin or mode == "f32":
            return real_conv2d(x, w, b, stride, padding)
        sc = 2.0 ** torch.floor(torch.log2(16384.0 / w.abs().max())).item()      # weight scale of the pack (encoder.py)
        h1, h2 = split16(x.float())
        g1, g2 = split16(w.float() * sc)
        c = lambda a, k: real_conv2d(a.double(), k.double(), None, stride, padding)
        y = c(h1, g1)
        if mode == "f16x3":
            y = y + c(h1, g2) + c(h2, g1)
        elif mode == "fp8corr":
            y = y + c(q8(h1), q8(g2)) + c(q8(h2), q8(g1))
        elif mode == "drop":
            y = y + c(h1, g2)
        elif mode == "fp8fixed":  # plain e4m3 with FIXED power-of-two scales (no per-block maxima): h1, h2*2^11, g1*2^-6, g2*2^5
            f8 = lambda t, k: (t * 2.0 ** k).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * 2.0 ** -k
            y = y + c(f8(h1, 0), f8(g2, 5)) + c(f8(h2, 11), f8(g1, -6))
        elif mode in MX:          # corrections in a block-scaled MX format (blocks of 32 input channels)
            y = y + c(qmx(h1, mode, 1), qmx(g2, mode, 1)) + c(qmx(h2, mode, 1), qmx(g1, mode, 1))
        y = (y / sc).to(x.dtype)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return conv


def main():
    torch.manual_seed(0)
    for (B, N) in ((8, 10), (2, 100)):
        cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
        sd = orc.init_state_dict(cfg, seed=7)
        x, S = fov_states(B, N, seed=1), comm_gso(B, N, 20 if N == 10 else 50, seed=2)
        ref = orc.planner_forward(x, S.clone(), sd, cfg)
        print("B=%d N=%d   |logit| max %.3f, mean %.3f" % (B, N, ref.abs().max(), ref.abs().mean()))
        for mode in (sys.argv[1:] or ("f16x3", "fp8corr", "fp8fixed", "mxfp8", "mxfp6_e2m3", "mxfp6_e3m2", "mxfp4", "drop")):
            for cmin, what in ((32, "every BasicBlock 3x3 conv"), (64, "layer2.conv2 + layer3"), (128, "layer3.conv2 only")):
                orc.tnf.conv2d = make_conv(mode, cmin)
                try:
                    got = orc.planner_forward(x, S.clone(), sd, cfg)
                finally:
                    orc.tnf.conv2d = real_conv2d
                d = (got - ref).abs()
                print("   %-8s on %-28s max|dlogit| %.2e   mean %.2e" % (mode, what, d.max(), d.mean()))


if __name__ == "__main__":
    main()
