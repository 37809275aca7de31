"""Stem stage in isolation (magat_encoder_stem_block_f32): the eight-agent-group kernel (form 2) against the 64-agent row-band
kernel (form 1) and against a float64 torch evaluation of the same two layers, value by value (plane pairs decoded), plus the
stage's time at the benchmark size.   python tools/stem8_check.py [--time]"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat          # noqa: E402
from magat_pathplanning_amd import encoder as enc                                   # noqa: E402
from magat_pathplanning_amd.synthetic import fov_states, make_config                # noqa: E402
from oracle import magat_oracle as orc                                              # noqa: E402


def decode(buf, M):
    """[tiles][36][hi: chunk 4 x 128 agents x 8 halves | lo: same] -> float64 (M, 36, 32) in true channel order"""
    T = (M + 127) // 128
    h = buf.view(torch.float16).view(T, 36, 2, 4, 128, 8).double()
    v = (h[:, :, 0] + h[:, :, 1])                          # (T, 36, chunk, agent, 8)
    out = torch.zeros(T * 128, 36, 32, dtype=torch.float64)
    for chunk in range(4):
        ks, fh = chunk >> 1, chunk & 1
        for i in range(8):
            out[:, :, enc.chain_channel(ks, fh, i)] = v[:, :, chunk, :, i].permute(0, 2, 1).reshape(T * 128, 36)
    return out[:M]


def main():
    dev = torch.device("cuda:0")
    cfg = make_config(num_agents=10, device="cuda:0")
    net = DecentralPlannerGATNet(cfg)
    sd = orc.init_state_dict(cfg, seed=3)
    net.load_state_dict(sd)
    net = net.to(dev).eval()
    rt = net._refresh(dev)
    lib = nat.lib()
    # float64 reference of the two layers (BN folded as encoder.fold_resnet does)
    pk = rt.pack_host.double()
    o = rt.pack_offs
    w0, b0 = pk[o[0]:o[0] + 864].view(32, 3, 3, 3), pk[o[1]:o[1] + 32]
    w1 = pk[o[2]:o[2] + 32 * 288].view(32, 3, 3, 32).permute(0, 3, 1, 2)
    b1 = pk[o[3]:o[3] + 32]
    for M in (8, 16, 20, 4, 300):
        x = fov_states(1, M, seed=M)[0] + 0.01 * torch.randn(M, 3, 11, 11)
        stem = torch.relu(torch.nn.functional.conv2d(x.double(), w0, b0, padding=1))
        c1 = torch.relu(torch.nn.functional.conv2d(stem, w1, b1, stride=2, padding=1))
        want_out = c1.permute(0, 2, 3, 1).reshape(M, 36, 32)
        want_ctr = stem[:, :, ::2, ::2].permute(0, 2, 3, 1).reshape(M, 36, 32)
        xd = x.to(dev).contiguous()
        T = (M + 127) // 128
        res = {}
        for form in (1, 2):
            out = torch.zeros(T * 36 * 16384, dtype=torch.uint8, device=dev)
            ctr = torch.zeros_like(out)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            rc = lib.magat_encoder_stem_block_f32(ctypes.byref(rt.desc), nat.ptr(xd), nat.ptr(out), nat.ptr(ctr), M, form,
                                                  nat.ptr(flag), nat.current_stream(dev))
            torch.cuda.synchronize()
            assert rc == 0, rc
            res[form] = (decode(out.cpu(), M), decode(ctr.cpu(), M), int(flag.item()))
        for form in (1, 2):
            eo = (res[form][0] - want_out).abs()
            ec = (res[form][1] - want_ctr).abs()
            print("M=%d form %d: out err %.3g ctr err %.3g flag %d" % (M, form, eo.max(), ec.max(), res[form][2]))
            if eo.max() > 1e-4 or ec.max() > 1e-4:
                for name, e in (("out", eo), ("ctr", ec)):
                    bad = (e > 1e-4).nonzero()
                    print("  ", name, "bad entries", len(bad), "agents", sorted(set(bad[:, 0].tolist()))[:12], "pixels",
                          sorted(set(bad[:, 1].tolist()))[:40], "channels", sorted(set(bad[:, 2].tolist()))[:34])
    if "--time" in sys.argv:
      ms = [int(a) for a in sys.argv[sys.argv.index("--time") + 1:] if a.isdigit()] or [51200]
      for M in ms:                                   # (--time 51200 65536 ...: the stage's time by agent count, multiples of 128)
        xd = fov_states(M // 100 + 1, 100, seed=1).view(-1, 3, 11, 11)[:M].to(dev).contiguous()
        T = M // 128
        out = torch.zeros(T * 36 * 16384, dtype=torch.uint8, device=dev)
        ctr = torch.zeros_like(out)
        for form in ((1, 2, 1, 2) if len(ms) == 1 else (2, 2)):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for it in range(3):
                lib.magat_encoder_stem_block_f32(ctypes.byref(rt.desc), nat.ptr(xd), nat.ptr(out), nat.ptr(ctr), M, form, None,
                                                 nat.current_stream(dev))
            ev[0].record()
            for it in range(20):
                lib.magat_encoder_stem_block_f32(ctypes.byref(rt.desc), nat.ptr(xd), nat.ptr(out), nat.ptr(ctr), M, form, None,
                                                 nat.current_stream(dev))
            ev[1].record()
            torch.cuda.synchronize()
            print("form %d: %.1f us per launch at M = %d" % (form, ev[0].elapsed_time(ev[1]) / 20 * 1e3, M))


if __name__ == "__main__":
    main()
