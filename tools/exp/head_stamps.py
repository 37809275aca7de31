"""Where the fused head kernel's K walk spends its cycles, per wave, summed over the slabs (experiment build of conv_gemm_bf16x6.hip with
g_head_dbg counters: DESIGN.md section 10).  python tools/exp/head_stamps.py B N"""
import sys, os, ctypes, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
B, N = int(sys.argv[1]), int(sys.argv[2])
cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device="cuda:0")
net = DecentralPlannerGATNet(cfg).to(dev).eval()
x, S = fov_states(B, N).to(dev), comm_gso(B, N, 50).to(dev)
with torch.no_grad():
    for _ in range(3):
        net.addGSO(S); net(x)
torch.cuda.synchronize()
h = ctypes.CDLL(nat.LIB_PATH)
buf = (ctypes.c_longlong * (8192 * 8))()
assert h.magat_head_debug_read(buf, 8192 * 8) == 0
t = np.array(buf[:], dtype=np.float64).reshape(8192, 8)
t = t[t[:, 4] > 0]
print("waves stamped", len(t), "| cycles per wave over the K walk, mean (p90): MFMA bodies + load issue %.0f (%.0f) | split %.0f (%.0f) | wait for the loads %.0f (%.0f) | barrier %.0f (%.0f)"
      % tuple(v for i in range(4) for v in (t[:, i].mean(), np.percentile(t[:, i], 90))))
