"""How long the GPU takes to reach its steady clocks after an idle period: per-step device times (hipEvent pairs around whole
steps) of the c3 forward, started (a) cold - the device idle for a few seconds, as behind bench.py's CPU-baseline leg -, (b) right
behind ~150 ms of registers-only MFMA work (magat_mfma_sustained_f16_ex).  python tools/ramp_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
B, N = 512, 100
cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device="cuda:0")
net = DecentralPlannerGATNet(cfg).to(dev).eval()
x, S = fov_states(B, N, seed=1).to(dev), comm_gso(B, N, 50, seed=2).to(dev)
lib = nat.lib()
cus = torch.cuda.get_device_properties(dev).multi_processor_count
scratch = torch.empty(260 * cus, dtype=torch.float32, device=dev)


def preheat(ms_total):
    v, mhz, per = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_double(0)
    for _ in range(max(1, ms_total // 10)):
        nat.check(lib.magat_mfma_sustained_f16_ex(ctypes.byref(v), ctypes.byref(mhz), ctypes.byref(per), nat.ptr(scratch), 10,
                                                  nat.current_stream(dev)), "sustained")
    return v.value, mhz.value


def stream_hbm(ms_total):
    """~ms_total of plain HBM streaming (device-to-device copies of 1 GB): wakes the memory clocks the MFMA probe does not touch"""
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.copy_(a); e1.record(); torch.cuda.synchronize()
    per = max(e0.elapsed_time(e1), 0.05)
    for _ in range(max(1, int(ms_total / per))):
        b.copy_(a)
    torch.cuda.synchronize()
    return 2 * a.numel() * 4 / per / 1e6      # GB/s of the first copy


def run(nsteps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)]
    with torch.no_grad():
        ev[0].record()
        for i in range(nsteps):
            net.addGSO(S); net(x)
            ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(nsteps)]


with torch.no_grad():
    for _ in range(3):
        net.addGSO(S); net(x)
torch.cuda.synchronize()
IDLE = float(os.environ.get("RAMP_IDLE_S", "5"))
for label, heat in (("cold (%g s idle)" % IDLE, 0), ("behind 150 ms of MFMA work", 150), ("cold again", 0), ("behind 400 ms of MFMA work", 400)):
    time.sleep(IDLE)
    info = preheat(heat) if heat else None
    if heat and os.environ.get("RAMP_HBM", "0") == "1":
        stream_hbm(heat)
    t = run(80)
    grp = [sum(t[i:i + 10]) / 10 for i in range(0, 80, 10)]
    print("%-28s steps 1-5: %s | mean of steps 1-10, 11-20, ..: %s%s" % (
        label, " ".join("%.3f" % v for v in t[:5]), " ".join("%.3f" % v for v in grp),
        "" if info is None else "  (preheat: %.0f TFLOP/s at %.0f MHz)" % info))

# second question: blocks of 40 steps, alternately without and with the library's per-launch hipEvent pairs (magat_profile_enable):
# does the instrumentation change the time of a step?
lib.magat_profile_reserve(40 * 41)
preheat(150)
run(20)
for rep in range(3):
    for prof in (0, 1):
        lib.magat_profile_reset(); lib.magat_profile_enable(prof)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e0.record()
        with torch.no_grad():
            for _ in range(40):
                net.addGSO(S); net(x)
        e1.record(); torch.cuda.synchronize()
        print("block of 40 steps, per-launch events %s: device %.4f ms/step, host wall %.4f" % ("ON " if prof else "off", e0.elapsed_time(e1) / 40, (time.perf_counter() - t0) * 1e3 / 40))
        lib.magat_profile_enable(0)
