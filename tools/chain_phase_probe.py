"""Phase timing of the BasicBlock chain kernel (debug build: python -m magat_pathplanning_amd.build_native --debug;
run with MAGAT_LIB_PATH=magat_pathplanning_amd/lib/libmagat_hip_debug.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
B, N = 512, 100
cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device="cuda:0")
net = DecentralPlannerGATNet(cfg).to(dev).eval()
x, S = fov_states(B, N).to(dev), comm_gso(B, N, 50).to(dev)
groups = 256          # persistent grid: one workgroup per CU; the stamps are those of its LAST agent group
buf = torch.zeros(groups, 8, dtype=torch.int64, device=dev)
h = ctypes.CDLL(nat.LIB_PATH)
h.magat_chain_set_debug_buffer.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    for _ in range(3):
        net.addGSO(S); net(x)
    h.magat_chain_set_debug_buffer(ctypes.c_void_p(buf.data_ptr()))
    net.addGSO(S); net(x)
    torch.cuda.synchronize()
    h.magat_chain_set_debug_buffer(None)
t = buf.cpu().numpy().astype(np.float64)
if t.any():          # (not launched when the chain kernels run as one launch, option BLOCK_FUSED = 2)
    names = ["prologue (barrier + LDS write of the prefetched inputs)", "stage A", "barrier A", "stage B", "barrier B", "stage C (+ stores)"]
    d = t[:, 1:7] - t[:, 0:6]
    tot = t[:, 6] - t[:, 0]
    print("chain kernel, per-workgroup cycles (wave 0, last group), mean / p90:  total %.0f / %.0f" % (tot.mean(), np.percentile(tot, 90)))
    for i, n in enumerate(names):
        print("  %-30s %8.0f / %8.0f   (%.1f %%)" % (n, d[:, i].mean(), np.percentile(d[:, i], 90), 100 * d[:, i].mean() / tot.mean()))

# layer3 kernel (one workgroup per agent group): per-WAVE stamps before and after every barrier
g3 = 256               # persistent: one workgroup per CU, the stamps are those of its LAST agent group
buf3 = torch.zeros(g3, 8, 16, dtype=torch.int64, device=dev)
h.magat_block3_set_debug_buffer.argtypes = [ctypes.c_void_p]
with torch.no_grad():
    h.magat_block3_set_debug_buffer(ctypes.c_void_p(buf3.data_ptr()))
    net.addGSO(S); net(x)
    torch.cuda.synchronize()
    h.magat_block3_set_debug_buffer(None)
t = buf3.cpu().numpy().astype(np.float64)
full = nat.get_option("BLOCK_FUSED") >= 2
names = ["prologue (input DMA + barrier)", "conv1 half 0", "  barrier", "conv2 half 0", "  barrier", "conv1 half 1", "  barrier",
         "conv2 half 1 + residual", "  barrier", "relu + pool + store"]
d = t[:, :, 1:11] - t[:, :, 0:10]                 # [wg][wave][phase]
tot = (t[:, :4, 10] - t[:, :4, 0]).mean()
if not full:
    print("layer3 kernel, cycles per agent group (mean over workgroups), by wave; total %.0f" % tot)
nw = 8 if t[:, 4:, 10].any() else 4          # the four-wave form of the kernel leaves waves 4..7 unstamped
print("  %-32s" % "phase" + "".join("   wave%d" % w for w in range(nw)))
for i, n in enumerate(names):
    if not full:
        print("  %-32s" % n + "".join("%8.0f" % d[:, w, i].mean() for w in range(nw)))
if not full and t[:, 0, 13].any():
    o = t[:, :4, :]
    print("  output phase: S write 0 %.0f | pool+store 0 %.0f | S write 1 %.0f | pool+store 1 + zero fill %.0f" % (
        (o[:, :, 13] - o[:, :, 9]).mean(), (o[:, :, 14] - o[:, :, 13]).mean(), (o[:, :, 15] - o[:, :, 14]).mean(),
        (o[:, :, 10] - o[:, :, 15]).mean()))

# the merged kernel (option BLOCK_FUSED = 2, default): stamps 0..10 of block_full_p_kernel, per wave, LAST group of a workgroup
if full:
    names = ["wait inputs + barrier", "stage A (layer1.conv2)", "stage B (layer2.conv1)", "stage C (layer2.conv2)",
             "layer3.conv1 half 0", "layer3.conv2 half 0", "layer3.conv1 half 1", "layer3.conv2 half 1 + residual",
             "pooled epilogue pass 0", "pass 1 + zero fill"]
    d = t[:, :4, 1:11] - t[:, :4, 0:10]
    print("merged kernel, cycles per agent group incl. the barrier behind each phase; total %.0f" % (t[:, :4, 10] - t[:, :4, 0]).mean())
    for i, n in enumerate(names):
        print("  %-34s" % n + "".join("%8.0f" % d[:, w, i].mean() for w in range(4)))
    if t[:, 4, 0].any() and t[:, 4, 3].any():         # FULL_CLOCKS: core-clock cycles and 100 MHz ticks of every workgroup, start to end
        cyc, ticks = t[:, 4, 2] - t[:, 4, 0], t[:, 4, 3] - t[:, 4, 1]
        print("  workgroups: %.0f core cycles in %.1f us each: the kernel ran at %.0f MHz" % (cyc.mean(), ticks.mean() / 100.0, (cyc / ticks).mean() * 100.0))
    if t[:, 0, 11].any():
        print("  layer3.conv1 half 0 by wave: walk " + " ".join("%.0f" % (t[:, w, 11] - t[:, w, 4]).mean() for w in range(4)) +
              " | epilogue " + " ".join("%.0f" % (t[:, w, 12] - t[:, w, 11]).mean() for w in range(4)) +
              " | barrier wait " + " ".join("%.0f" % (t[:, w, 5] - t[:, w, 12]).mean() for w in range(4)))
    if t[:, 0, 13].any() and t[:, 0, 14].any() and not t[:, 0, 15].any():      # block_full_p_kernel: stamps 13 / 14 = half 1's walk / epilogue
        print("  layer3.conv1 half 1 by wave: walk " + " ".join("%.0f" % (t[:, w, 13] - t[:, w, 6]).mean() for w in range(4)) +
              " | epilogue " + " ".join("%.0f" % (t[:, w, 14] - t[:, w, 13]).mean() for w in range(4)) +
              " | barrier wait " + " ".join("%.0f" % (t[:, w, 7] - t[:, w, 14]).mean() for w in range(4)))
    elif t[:, 0, 13].any():
        f = lambda a, b: " ".join("%.0f" % (t[:, w, a] - t[:, w, b]).mean() for w in range(4))
        print("  pooled epilogue pass 0 by wave: S write (waves 0-1) / DMA requests (2-3) " + f(13, 8) + " | barrier " + f(14, 13) +
              " | pool + store " + f(15, 14) + " | barrier " + f(9, 15))
