"""Per-wave phase cycles of stem8_kernel (debug build: python -m magat_pathplanning_amd.build_native --debug; run with
MAGAT_LIB_PATH=magat_pathplanning_amd/lib/libmagat_hip_debug.so): the stamps of the LAST group of every workgroup."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import fov_states, make_config
dev = torch.device("cuda:0")
cfg = make_config(num_agents=100, device="cuda:0")
net = DecentralPlannerGATNet(cfg).to(dev).eval()
rt = net._refresh(dev)
lib = nat.lib()
M = 51200
xd = fov_states(512, 100, seed=1).view(M, 3, 11, 11).to(dev).contiguous()
out = torch.zeros((M // 128) * 36 * 16384, dtype=torch.uint8, device=dev)
ctr = torch.zeros_like(out)
buf = torch.zeros(256, 8, 16, dtype=torch.int64, device=dev)
h = ctypes.CDLL(nat.LIB_PATH)
h.magat_stem8_set_debug_buffer.argtypes = [ctypes.c_void_p]
run = lambda: lib.magat_encoder_stem_block_f32(ctypes.byref(rt.desc), nat.ptr(xd), nat.ptr(out), nat.ptr(ctr), M, 2, None,
                                               nat.current_stream(dev))
for _ in range(3):
    run()
h.magat_stem8_set_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
h.magat_stem8_set_debug_buffer(None)
t = buf.cpu().numpy().astype(np.float64)
f = lambda a, b, ws: "".join("%8.0f" % (t[:, w, a] - t[:, w, b]).mean() for w in ws)
print("stem8 kernel, cycles per 8-agent group (last group of each workgroup); total %.0f" % (t[:, :, 5] - t[:, :, 0]).mean())
print("  %-36s" % "barrier (top), waves 0-7" + f(1, 0, range(8)))
print("  %-36s" % "stem walk, waves 0-7" + f(2, 1, range(8)))
print("  %-36s" % "barrier" + f(3, 2, range(8)))
print("  %-36s" % "conv1 walk, waves 0-3" + f(4, 3, range(4)))
print("  %-36s" % "conv1 epilogue + stores" + f(5, 4, range(4)))
print("  %-36s" % "next maps -> planes, waves 4-7" + " " * 32 + f(4, 3, range(4, 8)))
print("  %-36s" % "stride-2 copy" + " " * 32 + f(5, 4, range(4, 8)))
