"""Phase cycles of the one-launch encoder of the batch-1 step (block_lat_kernel<HEAD, STEM>; debug build:
python -m magat_pathplanning_amd.build_native --debug; MAGAT_LIB_PATH=magat_pathplanning_amd/lib/libmagat_hip_debug.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
h = ctypes.CDLL(nat.LIB_PATH)
h.magat_block_lat_set_debug_buffer.argtypes = [ctypes.c_void_p]
names = ["weights of the stem, zero fill, barrier", "state maps -> planes", "stem (9 MFMAs) + split", "layer1.conv1 | stride-2 copy",
         "A: layer1.conv2 + downsample", "B: layer2.conv1", "C: layer2.conv2 + downsample", "layer3.conv1", "layer3.conv2 walks",
         "pool + head planes + barrier", "head walk (K = 1152)", "head epilogue + planes + barrier", "compressMLP", "guard + book"]
for N in (10, 100):
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device="cuda:0")
    net = DecentralPlannerGATNet(cfg).to(dev).eval()
    x, S = fov_states(1, N).to(dev), comm_gso(1, N, 20 if N <= 20 else 50).to(dev)
    buf = torch.zeros(N, 4, 16, dtype=torch.int64, device=dev)
    with torch.no_grad():
        for _ in range(5):
            net.addGSO(S); net(x)
        acc = None
        for _ in range(20):
            buf.zero_()
            h.magat_block_lat_set_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
            net.addGSO(S); net(x)
            torch.cuda.synchronize()
            h.magat_block_lat_set_debug_buffer(None)
            t = buf.cpu().numpy().astype(np.float64)
            acc = t if acc is None else acc + t
    t = acc / 20
    d = t[:, :, 1:15] - t[:, :, 0:14]          # [agent][wave][phase]
    tot = (t[:, :, 14] - t[:, :, 0]).mean()
    print("N = %d: %.0f cycles per workgroup (mean over agents and waves); phases, cycles by wave (mean over agents)" % (N, tot))
    for i, n in enumerate(names):
        print("  %-42s" % n + "".join("%8.0f" % d[:, w, i].mean() for w in range(4)) + "   %5.1f %%" % (100 * d[:, :, i].mean() / tot))
