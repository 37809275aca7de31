#!/usr/bin/env python
"""bench.py -- agent-steps/s of the batched MAGAT GAT forward on MI355X (contract: see task brief).

A "step" is one addGSO(S) + forward(x) of DecentralPlannerGATNet over one batch of synthetic planning
instances already resident in HBM.  Workload at N GPUs = BASELINE.json configs[2]/[3]: 100 agents,
50x50 map, K=3, P=4, F=128, bottleneck + SkipConcat, KeyQuery, head concat, batch 512 PER GPU (weak
scaling: 8 GPUs = the 4096-instance config).  Instances are independent, so ranks never communicate
inside a step; the only collectives are the barrier, the MAX of the elapsed time and a sum of ones (`ranks_seen`).

`python bench.py --gpus N` launches itself under torch.distributed.run when it is not already running under it.

Rank 0 prints ONE JSON line.  `value` is measured with the library's DEFAULT arithmetic, which is fp32-class
(f16x3 split products, see `dtype`); extra objects:
  roofline       dominant kernel of the step: ALGORITHMIC flops (or bytes) / hipEvent time / the dense peak of the
                 data type its MFMAs are issued in; `issued_*` = the matrix-core flops actually issued (3x for f16x3).
                 The hipEvent times come from a second, instrumented pass of the same K steps right behind the timed
                 region (`instrumented_ms_per_step`): the event pairs cost 2-3 % of a step and are kept out of `value`
  roofline_gat   the hand-written graph kernel the north star names, against HBM
  kernels        every kernel tag (hipEvents on the launch stream through the library's profiling hooks)
  north_star_b1024  the same model at the north-star shape N=100, batch 1024 (a-s/s, graph-kernel GB/s and fraction)
  c2, c5         BASELINE configs[1] and [4] (N=20 batch 1024; N=1000 CSR bf16 batch 128) as extra legs: a-s/s, ms/step and the
                 graph layer's kernels against their roofs
  per_rank_ms, ranks   every rank's own elapsed time per step and the device it ran on (`ms_per_step` is their MAX)
  published_f32p4  the published model setting (N=10, K=2, P=4, G=32, head-mean, batch 1024) as its own leg
  f32_strict     the headline workload with every product on the float32 matrix cores (no split planes): a-s/s and the dominant
                 kernel against the 157.3 TF float32 MFMA roof
  ms_per_step_device  hipEvent pair around the K timed steps on the launch stream (ms_per_step is host wall-clock over barriers)
  step_ms_min_med_max device time of a timed step (one event between the steps): a sporadic 60-90 ms stall of the box shows up
                      as max >> median (value / ms_per_step stay the whole region's, as the contract says)
  cpu_baseline   the pinned CPU oracle (kind "port") on this box's host cores: processes x threads sweep over the physical
                 cores, median of three runs of the best split, bounded sample; runs BEHIND every GPU leg (round 5)
  roofline.legs  (round 6) the numbers of the extra legs once more in ONE compact object inside `roofline` (the object a
                 driver keeps whole): c2 / c5 a-s/s and graph-layer fractions, batch 1024, f32_strict, the c3 graph kernel against
                 both roofs, the published widths at N = 10 and N = 100, the batch-1 latency
  first_steps_ms the first five steps of a device that idled for two seconds, each timed on its own (not part of `value`):
                 what a cold caller sees before the clocks are back; `config.preheat` says what runs in front of the timed region
  latency_b1     one planning instance per step (the reference's test_batch_size = 1 loop): median step incl. the host copy
"""
import argparse
import ctypes
import gc
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_F32_TFLOPS = 157.3    # fp32 MFMA (v_mfma_f32_32x32x2_f32) = fp32 vector peak
PEAK_16_TFLOPS = 2500.0    # dense 16-bit MFMA (v_mfma_f32_32x32x16_{f16,bf16})

WORKLOADS = {
    # name: (B per GPU, N, map_w, K, P, G, bottleneckMode, CNN_mode, concat)
    "c3": (512, 100, 50, 3, 4, 128, "BottomNeck_skipConcat", "ResNetLarge_withMLP", True),
    "c2": (1024, 20, 28, 3, 4, 128, "BottomNeck_only", "ResNetLarge_withMLP", True),
    "c1": (64, 10, 20, 2, 1, 128, "BottomNeck_only", "ResNetLarge_withMLP", True),
    "n100b1024": (1024, 100, 50, 3, 4, 128, "BottomNeck_skipConcat", "ResNetLarge_withMLP", True),
    # config 5: large sparse graph -> CSR kernels, bf16 storage inside the GAT layer; c5f32 = same shape, fp32 storage
    "c5": (128, 1000, 160, 2, 4, 128, "BottomNeck_only", "ResNetLarge_withMLP", True),
    "c5f32": (128, 1000, 160, 2, 4, 128, "BottomNeck_only", "ResNetLarge_withMLP", True),
    # the PUBLISHED MAGAT setting (scripts/train_DMap.sh:42, "MAGAT F-32-P4"; the released checkpoints, README.md:385-390): 10 agents,
    # 20x20 map, K=2, 4 heads of 32 features, head MEAN (no --AttentionConcat), BottomNeck_only; batch 1024 planning instances
    "published_f32p4": (1024, 10, 20, 2, 4, 32, "BottomNeck_only", "ResNetLarge_withMLP", False),
    # ... and the same checkpoint shape on the README's 100-robot generalisation set (README.md:372-390), 50x50 map
    "published_f32p4_n100": (512, 100, 50, 2, 4, 32, "BottomNeck_only", "ResNetLarge_withMLP", False),
}
GAT_STORAGE = {"c5": "bf16"}

# issued matrix-core flops per algorithmic fp32 flop, the data type they are issued in and that type's dense peak
ARITH = {
    "f32": (1.0, "f32 MFMA (v_mfma_f32_32x32x2_f32)", PEAK_F32_TFLOPS),
    "f16x3": (3.0, "f16x3 split: 2 f16 planes per value (22 significand bits), 3 f16 MFMAs per fp32 multiply-add, "
                   "f32 accumulate (fp32-class: measured error = the fp32 MFMA kernel's)", PEAK_16_TFLOPS),
    "bf16": (1.0, "bf16 MFMA, f32 accumulate (bf16 storage variant)", PEAK_16_TFLOPS),
}


def valid_taps(hin, hout, stride, k=3, pad=1):
    """sum over output pixels of the number of 3x3 taps that fall inside the input (1-D count squared)."""
    one = sum(sum(1 for t in range(k) if 0 <= o * stride - pad + t < hin) for o in range(hout))
    return one * one


def lib_opt(nat, name):
    return nat.get_option(name)


def conv_arith(nat, cfg, layer):
    """Arithmetic form of BasicBlock `layer` (0..2) with the library's current options (csrc/encoder_f32.hip)."""
    if not cfg.CNN_mode.startswith("ResNet") or not (lib_opt(nat, "CONV_SPLIT") >> layer & 1):
        return "f32"
    return "f16x3"


def kernel_work(nat, cfg, N, S_bytes, deg=None, planned=False, fused_stem=True, pooled_head=False, csr_fused=False):
    """ALGORITHMIC work per AGENT-STEP for each kernel tag name: dict(flops=fp32 multiply-add flops, bytes=HBM bytes every
    kernel must move if it kept nothing it does not have to, arith=key of ARITH or None).  SURVEY.md section 8(d).  deg: mean
    out-degree, given when the layer runs on the CSR kernels (N > 128 or bf16 storage)."""
    G, K, P = cfg.bottleneckFeature, cfg.nGraphFilterTaps, cfg.nAttentionHeads
    F = G
    nfm = cfg.numInputFeatures
    NC = P * G + P * K * F
    t11, t6 = valid_taps(11, 6, 2), valid_taps(6, 6, 1)
    w = {}
    chans = [(32, 32), (32, 64), (64, 128)]
    conv = {}
    for l, (ci, co) in enumerate(chans):
        taps = t11 if l == 0 else t6
        hw_in = 121 if l == 0 else 36
        conv[(l, 1)] = (2 * taps * ci * co, 4 * (hw_in * ci + 36 * co))
        # conv2's residual 1x1 branch reads the block input at the 36 stride-s pixels only; the two-kernel stem path hands it
        # the full 121-pixel map, the fused stem kernel writes those 36 pixels as their own map
        hw2 = hw_in if (l == 0 and not fused_stem) else 36
        conv[(l, 2)] = (2 * (t6 * co * co + 36 * ci * co), 4 * (36 * co + hw2 * ci + 36 * co))
    stem = (2 * 121 * 27 * 32, 4 * (363 + 121 * 32))
    w["conv_first"] = dict(flops=stem[0], bytes=stem[1], arith="f32")
    # stem + layer1.conv1 in one kernel: the (3,H,W) input in, layer1.conv1's map + the stem's stride-2 pixels out
    w["conv_first+layer1.conv1 (fused)"] = dict(flops=stem[0] + conv[(0, 1)][0], bytes=4 * (363 + 2 * 36 * 32), arith="f16x3")
    for l in range(3):
        a = conv_arith(nat, cfg, l)
        w["layer%d.conv1" % (l + 1)] = dict(flops=conv[(l, 1)][0], bytes=conv[(l, 1)][1], arith=a)
        w["layer%d.conv2+ds" % (l + 1)] = dict(flops=conv[(l, 2)][0], bytes=conv[(l, 2)][1], arith=a)
    # BasicBlock chain kernels (csrc/block_fused.hip): conv1 -> conv2 (+downsample) with the maps in LDS
    w["layer1.conv2+layer2 (fused)"] = dict(flops=conv[(0, 2)][0] + conv[(1, 1)][0] + conv[(1, 2)][0],
                                            bytes=4 * (2 * 36 * 32 + 36 * 64), arith="f16x3")
    w["layer2 (fused)"] = dict(flops=conv[(1, 1)][0] + conv[(1, 2)][0], bytes=4 * (36 * 32 + 36 * 64), arith="f16x3")
    # layer3 + ReLU + 2x2 pool in one launch: layer2's map in, the 9 pooled cells out
    w["layer3 (fused, pooled)"] = dict(flops=conv[(2, 1)][0] + conv[(2, 2)][0], bytes=4 * (36 * 64 + 9 * 128), arith="f16x3")
    # both chain kernels as one launch: layer1.conv1's map + the stem's stride-2 pixels in, the 9 pooled cells out
    w["layer1.conv2+layer2+layer3 (fused, pooled)"] = dict(
        flops=conv[(0, 2)][0] + conv[(1, 1)][0] + conv[(1, 2)][0] + conv[(2, 1)][0] + conv[(2, 2)][0],
        bytes=4 * (2 * 36 * 32 + 9 * 128), arith="f16x3")
    head_a = "f16x3" if (pooled_head and lib_opt(nat, "HEAD_F16") and conv_arith(nat, cfg, 2) == "f16x3") else "f32"
    w["head(avgpool+fc+linear)"] = dict(flops=2 * 9 * 128 * nfm, bytes=4 * ((9 if pooled_head else 36) * 128 + nfm), arith=head_a)
    w["compressMLP"] = dict(flops=2 * nfm * G, bytes=4 * (nfm + G), arith=head_a)     # (follows the head's arithmetic)
    gat_a = "f32"
    if lib_opt(nat, "GAT_SPLIT") and NC % 32 == 0 and G % 32 == 0:
        gat_a = "f16x3"
    if getattr(cfg, "gat_storage", "fp32") == "bf16":
        gat_a = "bf16"
    w["gat_maps_gemm"] = dict(flops=2 * G * NC, bytes=4 * (G + NC), arith=gat_a)
    yw = P * F if cfg.AttentionConcat else F          # head-mean: one merged [N][F] row block is written
    # graph kernel: SURVEY 8(d) "kernel (ii)" bytes per instance / N (Q, U cross HBM once)
    w["gat_graph"] = dict(flops=0, arith=None, bytes=(4 * (N * G + P * N * G + P * K * N * F + N * yw) +
                                                      (16 * N if planned else S_bytes * N * N)) / N)
    # the whole layer as one launch of matrix-core products (gat_mfma.hip): the LAYER-level bytes of SURVEY 8(d) - X, S, Y -
    # and, per agent-step, the maps (2 G NC) + the dense score product (2 N G per head) + K - 1 dense hops (2 N F per head)
    w["gat_layer (one launch)"] = dict(flops=2 * G * NC + 2 * N * G * P + 2 * (K - 1) * N * F * P, arith="f16x3",
                                       bytes=(4 * (N * G + N * yw) + S_bytes * N * N) / N)
    if planned:
        w["gat_prepare"] = dict(flops=0, arith=None, bytes=S_bytes * N + 16 + 8.0 / N)
    if deg is not None:
        # CSR kernels: X, the hoisted maps Z and Y in the storage type (4 or 2 bytes), CSR + CSC index arrays, and the
        # attention values written by the score kernel and read once per hop
        es = 2 if getattr(cfg, "gat_storage", "fp32") == "bf16" else 4
        w["gat_graph"] = dict(flops=0, arith=None,
                              bytes=es * (G + P * G + P * K * F + yw) + 4 * (2 + 3 * deg) + 4 * P * deg * K)
        w["gat_maps_gemm"]["bytes"] = es * (G + NC)
        if csr_fused:
            # maps inside the score / hop kernels (csrc/gat_csr_fused.hip): the tag's launches ARE the layer - SURVEY 8(d)'s
            # LAYER-level bytes (X in, Y out, one CSR index + row pointer per edge / row: 2 G + 2 P F + 4 (1 + deg) in bf16) and
            # the layer's flops (q' = W^T x and the tap contraction on the matrix cores + the edge products)
            w["gat_graph"] = dict(flops=2 * G * P * G + 2 * P * K * G * F + 4 * P * deg * G, arith="bf16",
                                  bytes=es * (G + yw) + 4 * (1 + deg))
    width = yw + (nfm if cfg.bottleneckMode == "BottomNeck_skipConcat" else 0)
    # the action head (width -> 5) runs as streamed float32 dot products (vector FMAs, option SKINNY): bound by its bytes; with
    # bf16 storage in the graph layer it reads that layer's rows as bf16
    skinny = bool(lib_opt(nat, "SKINNY"))
    gbytes = 2 if (skinny and getattr(cfg, "gat_storage", "fp32") == "bf16" and yw % 8 == 0) else 4
    w["actionsMLP"] = dict(flops=2 * width * 5, bytes=gbytes * yw + 4 * (width - yw + 5), arith=None if skinny else "f32")
    return w


def load_pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/<round>/summary_*.json, made by
    tools/profile_round.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction).  PMC counters
    cannot be collected from inside this process, so the latest committed summary of the same workload is used."""
    import glob
    # the latest round by directory name (r02h > r02f > r01h): file times do not survive a checkout
    # (summaries of the other workloads - summary_<tag>_c2.json, ..._c5.json, tools/profile_round.sh <tag> c5 - are not c3's)
    paths = sorted((q for q in glob.glob(os.path.join(ROOT, "profiles", "*", "summary_*.json"))
                    if not os.path.basename(q)[:-5].endswith(("_c2", "_c5"))),
                   key=lambda q: os.path.basename(os.path.dirname(q)))
    if not paths:
        return {}
    try:
        d = json.load(open(paths[-1]))
    except Exception:
        return {}
    out = {}
    for k, v in d.get("layers", {}).items():
        if "hbm_bytes_per_launch" in v:
            out[k] = {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "source": os.path.relpath(paths[-1], ROOT),
                      "collected": d.get("collected", "a gpurun MI355X box of the round named by the directory (builder's run)")}
    return out


def build_model(cfg, device, seed=1337):
    import torch
    from magat_pathplanning_amd import DecentralPlannerGATNet
    torch.manual_seed(seed)
    net = DecentralPlannerGATNet(cfg)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():       # random-init weights of the named architecture; BN stats perturbed
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2, generator=g)
                m.running_var.uniform_(0.5, 1.5, generator=g)
                m.bias.normal_(0, 0.1, generator=g)
    return net.to(device).eval()


def sustained_mfma_tflops(lib, dev, reps=3, ms=5):
    import torch
    from magat_pathplanning_amd import _native as nat
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    scratch = torch.empty(256 * cus + 4 * cus, dtype=torch.float32, device=dev)      # results + the kernel's own clock stamps
    best = (0.0, 0.0, 0.0)
    for _ in range(reps):
        v, mhz, per = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        nat.check(lib.magat_mfma_sustained_f16_ex(ctypes.byref(v), ctypes.byref(mhz), ctypes.byref(per), nat.ptr(scratch), ms,
                                                  nat.current_stream(dev)), "magat_mfma_sustained_f16_ex")
        best = max(best, (v.value, mhz.value, per.value))
    sustained_mfma_tflops.clock_mhz, sustained_mfma_tflops.per_clk = best[1], best[2]
    return best[0]


def train_step_leg(dev, steps=10):
    import torch
    import torch.nn.functional as tnf
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    out = {"what": "forward + cross-entropy + backward + SGD step, train mode (batch-statistics BatchNorm), K=3, P=4, "
                   "BottomNeck_skipConcat; ms per step", "steps": steps}
    prev = os.environ.get("MAGAT_TRAIN_CNN")
    try:
        for Bt, Nt in ((64, 100), (64, 32), (64, 10)):
            cfgt = make_config(num_agents=Nt, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat",
                               device=str(dev))
            xt = fov_states(Bt, Nt, seed=21).to(dev)
            St = comm_gso(Bt, Nt, 20 if Nt <= 20 else 50, seed=22).to(dev)
            tgt = torch.randint(0, 5, (Bt * Nt,), generator=torch.Generator().manual_seed(23)).to(dev)
            row = {}
            for backend in ("hip", "torch", "auto"):
                os.environ["MAGAT_TRAIN_CNN"] = backend
                torch.manual_seed(24)
                net = DecentralPlannerGATNet(cfgt).to(dev).train()
                opt = torch.optim.SGD(net.parameters(), lr=0.01)
                for it in range(3 + steps):
                    if it == 3:
                        torch.cuda.synchronize(dev)
                        t0 = time.perf_counter()
                    net.addGSO(St)
                    loss = tnf.cross_entropy(net(xt), tgt)
                    opt.zero_grad()
                    loss.backward()
                    opt.step()
                torch.cuda.synchronize(dev)
                # (`auto` is the module's default: torch's convolutions below train_cnn.TRAIN_HIP_MIN_AGENTS agents, the HIP ones above)
                row[{"hip": "cnn_on_hip_kernels_ms", "torch": "cnn_on_torch_miopen_ms", "auto": "default_auto_ms"}[backend]] = round(
                    (time.perf_counter() - t0) / steps * 1e3, 3)
                del net, opt
            out["%dx%d_agents" % (Bt, Nt)] = row
    finally:
        if prev is None:
            os.environ.pop("MAGAT_TRAIN_CNN", None)
        else:
            os.environ["MAGAT_TRAIN_CNN"] = prev
        torch.cuda.empty_cache()
    return out


def physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        return os.cpu_count()


def _cpu_worker(conn, cfg_dict, sd_np, N, map_w, Bc):
    """One process of the CPU baseline: runs the pinned oracle's forward on its own Bc instances for a given number of
    seconds with a given thread count, on command.  Never touches the GPU."""
    import types
    import torch
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states
    from oracle import magat_oracle as orc
    cfg = types.SimpleNamespace(**cfg_dict)
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    x, S = fov_states(Bc, N, seed=99), comm_gso(Bc, N, map_w, seed=98)
    with torch.no_grad():
        torch.set_num_threads(1)
        conn.send("ready")
        while True:
            msg = conn.recv()
            if msg is None:
                return
            threads, seconds, start_at = msg
            torch.set_num_threads(threads)
            orc.planner_forward(x, S.clone(), sd, cfg)          # warm-up at this thread count
            while time.time() < start_at:                       # all active workers start together
                time.sleep(0.001)
            t0 = time.perf_counter()
            reps = 0
            while True:
                orc.planner_forward(x, S.clone(), sd, cfg)
                reps += 1
                el = time.perf_counter() - t0
                if el >= seconds:
                    break
            conn.send((reps, el))


def cpu_baseline(cfg, sd, N, map_w, budget_s=30.0):
    """Pinned CPU oracle (oracle/magat_oracle.py, the reference's dense op sequence in torch-CPU) on a bounded sample of the
    same workload, using the WHOLE host: planning instances are as independent on the CPU as on the GPU, so the fair
    all-cores figure is processes x threads.  A pool of worker processes (each with its own Bc-instance batch) is swept over
    (processes, threads) splits of the physical cores; the best split is then measured three times and the MEDIAN reported."""
    import multiprocessing as mp
    import numpy as np
    import torch
    t_begin = time.perf_counter()
    logical, phys = os.cpu_count() or 1, physical_cores() or 1
    Bc = 8
    splits = []
    for th in (1, 2, 4, 8, 16, 32):
        pr = max(1, phys // th)
        if pr * th <= logical and pr <= 256:        # (one process per physical core is part of the sweep)
            splits.append((pr, th))
    splits.append((1, min(phys, logical)))
    splits = sorted(set(splits))
    nproc = max(p for p, _ in splits)
    ctx = mp.get_context("spawn")
    sd_np = {k: v.detach().cpu().numpy() for k, v in sd.items()}
    cfg_dict = dict(vars(cfg), device="cpu")
    workers = []
    for _ in range(nproc):
        a, b = ctx.Pipe()
        pr = ctx.Process(target=_cpu_worker, args=(b, cfg_dict, sd_np, N, map_w, Bc), daemon=True)
        pr.start()
        workers.append((pr, a))
    for _, a in workers:
        assert a.recv() == "ready"

    def measure(procs, threads, seconds):
        start_at = time.time() + 0.3 + 0.02 * procs
        for _, a in workers[:procs]:
            a.send((threads, seconds, start_at))
        got = [a.recv() for _, a in workers[:procs]]
        return sum(Bc * N * r / el for r, el in got)        # every worker's own rate over its own interval, summed

    sweep = {}
    for pr, th in splits:
        sweep[(pr, th)] = measure(pr, th, 1.5)
    best = max(sweep, key=sweep.get)
    left = budget_s - (time.perf_counter() - t_begin)
    per = max(1.5, min(4.0, left / 3.5))
    finals = sorted(measure(best[0], best[1], per) for _ in range(3))
    for pr, a in workers:
        a.send(None)
    for pr, a in workers:
        pr.join(timeout=10)
    return {"value": round(finals[1], 1), "unit": "agent-steps/s", "cores": best[0] * best[1], "kind": "port",
            "processes": best[0], "threads_per_process": best[1], "repeats": [round(v, 1) for v in finals],
            "physical_cores": phys, "logical_cores": logical,
            "split_sweep": {"%dx%d" % k: round(v, 1) for k, v in sweep.items()},
            "sample": "oracle.planner_forward, %d instances x N=%d per process (same model/config), %d processes x %d threads "
                      "(best of the swept splits of the %d physical cores), median of 3 runs of %.1f s, torch-CPU %s.  Port against "
                      "the TRUE reference (importable in the build container only), same inputs, 8 threads, 3 shapes x 4 runs: "
                      "time ratio port / reference 0.81-1.30, median 1.05 (tools/cpu_equivalence.py, "
                      "profiles/r06a/cpu_equivalence.txt): the port understates the reference's CPU rate by up to that factor"
                      % (Bc, N, best[0], best[1], phys, per, torch.__version__),
            "port_over_reference_time_ratio": [0.81, 1.30]}


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` outside torch.distributed.run: start N ranks of this script on this node."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed steps after >= 10 warm-ups
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the north_star_b1024 / c2 / c5 / published / f32_strict / train_step legs")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL of the N > 1 launch on a box with fewer GPUs than ranks: every rank drives cuda:0, the "
                         "collectives (sum of ones, barrier, gather of the elapsed times) go over gloo.  Exercises bench.py's own "
                         "rank logic (ranks_seen, per_rank_ms, MAX over ranks); its value is NOT a scaling measurement and the "
                         "line says so (config.rehearsal)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        relaunch_under_torchrun(args.gpus)        # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    ranks_seen = 1
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)          # "nccl" IS RCCL on ROCm
        ones = torch.ones(1, device="cpu" if args.share_gpu else dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N, map_w, K, P, G, bmode, cnn, concat = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    cfg = make_config(num_agents=N, nGraphFilterTaps=K, nAttentionHeads=P, bottleneckFeature=G,
                      bottleneckMode=bmode, CNN_mode=cnn, AttentionConcat=concat, device=str(dev),
                      gat_storage=GAT_STORAGE.get(args.workload, "fp32"))
    net = build_model(cfg, dev)
    lib = nat.lib()

    # The CPU leg (rank 0, N=1 only) runs LAST since round 5 (see the end of main): 128 worker processes at full load for ~40 s in
    # FRONT of the GPU legs left the node hot and the host busy reaping them - same box, same command: 2.40 / 3.51 (a 23 ms host
    # stall inside the timed region; the device took 2.35) / 2.46 ms per step with the CPU leg first, 2.34 without it
    # (profiles/r05g/driver_style_runs.txt)
    cpu = None

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def run_leg(x, S, steps, warmup, timing, net=net, repeats=1):
        """`warmup` untimed steps, then EXACTLY `steps` timed ones between barrier + synchronize on both sides.
        Returns (MAX over ranks of the elapsed seconds, per-tag kernel times of THIS rank, every rank's own elapsed ms).
        repeats > 1 (the EXTRA legs only, never the headline): the timed region is run that often and the fastest one reported -
        the GPU boxes show a sporadic 60-90 ms device stall every few seconds (tools/exp/stall_probe2.py: not tied to the model,
        the step or the collector), which triples a 20-step region of a 1 ms workload when it lands inside it."""
        def step():
            net.addGSO(S)
            return net(x)
        with torch.no_grad():
            # a full collection of the interpreter's heap (torch + numpy: ~1e6 objects) inside the step loop showed up as one-off
            # 50-90 ms host stalls: collect now, move the survivors out of the collector's reach; every step still runs in full.
            # IN FRONT of the warm-up steps since round 5: behind them it left the device idle for those 50-90 ms, and a device
            # that idles that long drops its clocks and needs ~5-10 steps to come back (tools/ramp_probe.py) - the timed region
            # started inside that ramp (r04, driver settings: timed steps 4.5 % slower than the instrumented pass behind them)
            gc.collect()
            gc.freeze()
            for _ in range(warmup):
                out = step()
            elapsed = None
            for _rep in range(max(1, repeats)):
                barrier()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                # one event per step INSIDE the region (~1 us each, no synchronisation): the per-step device times say whether
                # a stall of the box landed in the region - `step_ms_min_med_max` in the line; `value` is the whole region's
                marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
                t0 = time.perf_counter()
                ev0.record()                     # (torch's current stream IS the stream the library launches on)
                marks[0].record()
                for i_ in range(steps):
                    out = step()
                    marks[i_ + 1].record()
                ev1.record()
                barrier()
                el = time.perf_counter() - t0
                if elapsed is None or el < elapsed:
                    elapsed = el
                    run_leg.device_ms = ev0.elapsed_time(ev1) / steps
                    per = sorted(marks[i_].elapsed_time(marks[i_ + 1]) for i_ in range(steps))
                    run_leg.step_ms = [round(per[0], 4), round(per[len(per) // 2], 4), round(per[-1], 4)]
            # The per-kernel times come from a SECOND pass of the same `steps` steps, right behind the timed region, with the
            # library's hipEvent pairs around every launch: ~14 pairs per step cost 2-3 % of a c3 step (same box: 2.47 ms
            # against 2.53), and the timed region is the workload, not the instrumentation.  Its own elapsed time is reported
            # next to the table (`instrumented_ms_per_step`).
            if timing:
                lib.magat_profile_reserve(40 * (steps + 1))      # event pairs created outside the instrumented pass
                lib.magat_profile_reset()
                lib.magat_profile_enable(1)
                barrier()
                t1 = time.perf_counter()
                for _ in range(steps):
                    out = step()
                barrier()
                run_leg.instrumented_ms = (time.perf_counter() - t1) / steps * 1e3
            gc.unfreeze()
        kern = {}
        if timing:
            lib.magat_profile_enable(0)
            lib.magat_profile_collect()
            for tag, name in nat.TAGS.items():
                cnt, tot = ctypes.c_longlong(0), ctypes.c_double(0.0)
                lib.magat_profile_read(tag, ctypes.byref(cnt), ctypes.byref(tot))
                if cnt.value:
                    kern[name] = (cnt.value, tot.value)
        assert out.shape == (x.shape[0] * x.shape[1], 5) and bool(torch.isfinite(out).all())
        per_rank = [elapsed * 1e3]
        if dist is not None:
            mine = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.share_gpu else dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank = [float(t.item()) * 1e3 for t in every]
            elapsed = max(per_rank) / 1e3
        return elapsed, kern, per_rank

    def kernel_table(kern, steps, Bk, Nk, S, pmc, cfg=cfg):
        """per-kernel entries: time, algorithmic rate against the roof that binds, issued rate."""
        csr = Nk > 128 or cfg.gat_storage == "bf16"
        deg = float((S != 0).sum().item()) / (Bk * Nk) if csr else None
        work = kernel_work(nat, cfg, Nk, 4, deg, planned="gat_prepare" in kern and not csr,
                           fused_stem="layer1.conv1" not in kern, pooled_head=("layer3 (fused, pooled)" in kern or "layer1.conv2+layer2+layer3 (fused, pooled)" in kern),
                           csr_fused=csr and cfg.gat_storage == "bf16" and "gat_graph" in kern and "gat_maps_gemm" not in kern)
        if "conv_first" in kern and "layer1.conv1" not in kern and cfg.CNN_mode.startswith("ResNet"):
            kern = {("conv_first+layer1.conv1 (fused)" if k == "conv_first" else k): v for k, v in kern.items()}
        # compressMLP in the head's epilogue (option HEAD_COMPRESS; round 5): one launch carries both layers' work - the pooled
        # map in, feat AND comp out
        hk = "head(avgpool+fc+linear)"
        if hk in kern and "compressMLP" not in kern and "compressMLP" in work and cfg.CNN_mode.startswith("ResNet"):
            kern = {("head+compressMLP (one launch)" if k == hk else k): v for k, v in kern.items()}
            hw_, cw_ = work[hk], work["compressMLP"]
            work["head+compressMLP (one launch)"] = dict(flops=hw_["flops"] + cw_["flops"], arith=hw_["arith"],
                                                         bytes=hw_["bytes"] + 4 * cfg.bottleneckFeature)
        agent_steps = Bk * Nk * steps
        table = {}
        for name, (cnt, tot_ms) in kern.items():
            sec = tot_ms / 1e3
            ent = {"launches": cnt, "avg_us": round(tot_ms * 1e3 / cnt, 2), "ms_per_step": round(tot_ms / steps, 4)}
            wk = work.get(name)
            if wk and sec > 0:
                nprod, dt_name, peak = ARITH[wk["arith"]] if wk["arith"] else (0.0, None, None)
                alg_tf = wk["flops"] * agent_steps / sec / 1e12
                alg_gbs = wk["bytes"] * agent_steps / sec / 1e9
                tr = pmc.get(name)
                hbm_bytes = tr["hbm_bytes_per_launch"] if tr else wk["bytes"] * Bk * Nk      # per launch
                t_hbm = hbm_bytes / (PEAK_HBM_GBS * 1e9)
                t_mfma = (wk["flops"] * nprod * Bk * Nk / (peak * 1e12)) if peak else 0.0
                ent["flops_per_agent_step"] = wk["flops"]
                ent["bytes_per_agent_step"] = round(wk["bytes"], 1)
                ent["algorithmic_gbs"] = round(alg_gbs, 1)
                if peak:
                    ent["algorithmic_tflops"] = round(alg_tf, 2)
                    ent["issued_tflops"] = round(alg_tf * nprod, 2)
                    ent["issued_frac"] = round(alg_tf * nprod / peak, 4)
                    ent["mfma_dtype"] = dt_name
                if tr:
                    ent["hbm_traffic_gbs"] = round(tr["hbm_bytes_per_launch"] * cnt / sec / 1e9, 1)
                # the roof that binds: the one the kernel would need longer for at its peak (PMC bytes when committed)
                if t_hbm >= t_mfma:
                    ent.update(bound="hbm", achieved=round(alg_gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                               frac=round(alg_gbs / PEAK_HBM_GBS, 4))
                else:
                    ent.update(bound="mfma", achieved=round(alg_tf, 2), peak=peak, unit="TFLOP/s", frac=round(alg_tf / peak, 4))
            table[name] = ent
        return table

    def roof(table, name, Bk, Nk, pmc):
        e = table[name]
        tr = pmc.get(name)
        r = {"kernel": name, "bound": e["bound"], "achieved": e["achieved"], "peak": e["peak"], "unit": e["unit"],
             "frac": e["frac"], "traffic": None if tr is None else round(tr["hbm_bytes_per_launch"]),
             "traffic_source": None if tr is None else tr["source"],
             "traffic_collected": None if tr is None else tr["collected"],
             # PMC counters cannot be read from inside this process: `traffic` is the committed rocprofv3 --pmc pass of the same
             # command on ANOTHER box, bytes per launch (a property of the kernel and the workload, not of the clock) - it is
             # NOT a measurement of this run, and hbm_traffic_gbs divides it by THIS run's time
             "traffic_measured_in_this_run": False,
             "algorithmic_per_launch": round((e["bytes_per_agent_step"] if e["bound"] == "hbm" else
                                              e["flops_per_agent_step"]) * Bk * Nk),
             "avg_us": e["avg_us"], "launches": e["launches"]}
        for k in ("mfma_dtype", "issued_tflops", "issued_frac", "algorithmic_tflops", "algorithmic_gbs", "hbm_traffic_gbs"):
            if k in e:
                r[k] = e[k]
        return r

    x = fov_states(B, N, seed=1337 + rank).to(dev)
    S = comm_gso(B, N, map_w, seed=4242 + rank).to(dev)       # float32, as the dataloader hands it over
    # the profiling hooks run on EVERY rank or on none (a MAX over ranks of differently instrumented processes would
    # measure the instrumentation); rank 0's table is the one reported
    timing = not args.no_kernel_timing
    # The device has been IDLE while the CPU-baseline leg ran (tens of seconds): its clocks take ~10 steps (25 ms) to come back
    # (tools/ramp_probe.py, profiles/r05g/ramp_probe.txt: per-step device times 3.14, 3.00, 2.80, 2.64, 2.56 ms .. then 2.37-2.39
    # steady; behind matrix work 2.75, 2.46, 2.40, 2.39 ..).  With the driver's `--warmup 5 --steps 20` the timed region used to
    # start inside that ramp (r04: the instrumented pass right behind it was 4.5 % FASTER than the timed steps).  The
    # registers-only MFMA measurement this line reports anyway (`roofline.sustained_*`) now runs HERE, in front of the W warm-up
    # steps, on every rank: ~150 ms of matrix work that is not a step - W and K are untouched, the timed region is steady state.
    # What a COLD caller sees: two untimed steps take the one-off work (weight packs, the calibration
    # pass), the device then idles for two seconds, and the next five steps are timed one by one (synchronised each: +~20 us of
    # host latency per step).  Not part of `value`; the pre-heat below and the W warm-up steps come after it.
    first_steps_ms = []
    with torch.no_grad():
        for _ in range(2):
            net.addGSO(S)
            net(x)
        torch.cuda.synchronize(dev)
        time.sleep(2.0)
        for _ in range(5):
            t0_ = time.perf_counter()
            net.addGSO(S)
            net(x)
            torch.cuda.synchronize(dev)
            first_steps_ms.append(round((time.perf_counter() - t0_) * 1e3, 4))
    sus_pre = None
    try:
        sus_pre = sustained_mfma_tflops(lib, dev, reps=15, ms=10)
        sus_pre = (sus_pre, sustained_mfma_tflops.clock_mhz, sustained_mfma_tflops.per_clk)
    except Exception:
        pass
    elapsed, kern, per_rank_ms = run_leg(x, S, args.steps, args.warmup, timing)
    instr_ms = getattr(run_leg, "instrumented_ms", 0.0)
    dev_ms = getattr(run_leg, "device_ms", 0.0)      # hipEvent pair around the K timed steps on the launch stream (this rank)
    step_ms = getattr(run_leg, "step_ms", None)      # [min, median, max] device time of a timed step (events between the steps)
    # what each rank ran on: the 0.9-scaling target is decided by the slowest die, so the line names them
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "device": props.name, "cus": props.multi_processor_count,
            "clock_mhz": round(getattr(props, "clock_rate", 0) / 1e3), "host": socket.gethostname(), "local_rank": local_rank}
    rank_info = [mine]
    if dist is not None:
        rank_info = [None] * world
        dist.all_gather_object(rank_info, mine)

    res = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = B * N * world * args.steps / elapsed
        ariths = sorted({conv_arith(nat, cfg, l) for l in range(3)})
        res = {"metric": "agent-steps/s (batched GAT forward)", "value": round(value, 1), "unit": "agent-steps/s",
               "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 4), "ms_per_step_device": round(dev_ms, 4), "step_ms_min_med_max": step_ms,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "per_rank_ms": [round(v / args.steps, 4) for v in per_rank_ms], "ranks": rank_info,
               "dtype": "f32 in/out, fp32-class arithmetic: convolutions + GAT maps as %s split products on the 16-bit matrix "
                        "cores with f32 accumulation (the encoder head too, on the layer3 kernel's pooled map; with GAT_MFMA the attention "
                        "scores and the K-hop aggregation as well), MLPs on f32 MFMA%s"
                        % ("/".join(ariths), "" if cfg.gat_storage == "fp32" else "; bf16 STORAGE inside the GAT layer"),
               "data": "synthetic (seeded binary FOV states + comm-radius GSO; random-init weights, BN stats perturbed)",
               "config": {"workload": "%s: N=%d agents, %dx%d map, K=%d, P=%d, F=%d, %s, %s, KeyQuery, %s; batch %d per GPU "
                                      "(global %d); resident inputs, addGSO+forward per step"
                                      % (args.workload, N, map_w, map_w, K, P, G, bmode, cnn,
                                         "head-concat" if concat else "head-mean", B, B * world),
                          "arithmetic": {a: ARITH[a][1] for a in ariths},
                          "parity": "logits within 1e-4 of the reference (gate; observed ~1e-6 with this arithmetic)",
                          "options": {k: lib_opt(nat, k) for k in ("CONV_SPLIT", "RANGE_GUARD", "BLOCK_FUSED", "HEAD_F16",
                                                                    "HEAD_COMPRESS", "GAT_MFMA", "GAT_PACK")},
                          "global_batch": B * world, "agents": N, "parallelism": "instance-sharded x%d" % world,
                          "preheat": "in front of the W warm-up steps: gc.collect + freeze, and ~150 ms of registers-only MFMA work "
                                     "(the roofline.sustained_* measurement) so that the timed region starts at the clocks the "
                                     "chip holds under load; a cold caller's first steps: first_steps_ms"},
               "first_steps_ms": first_steps_ms}
        if args.share_gpu:
            res["config"]["rehearsal"] = ("--share-gpu: %d ranks drive ONE device over gloo - the launch / rank logic of the "
                                          "N > 1 run, not a scaling measurement" % world)
        if timing:
            pmc = load_pmc_traffic() if args.workload == "c3" and not args.batch else {}
            table = kernel_table(kern, args.steps, B, N, S, pmc)
            res["kernels"] = table
            bounded = [k for k, v in table.items() if "bound" in v and k != "gat_prepare"]
            if bounded:
                dom = max(bounded, key=lambda k: table[k]["ms_per_step"])
                res["roofline"] = roof(table, dom, B, N, pmc)
            for gk in ("gat_graph", "gat_layer (one launch)"):
                if gk in table:
                    res["roofline_gat"] = roof(table, gk, B, N, pmc)
            # what THIS box sustains on the f16 matrix cores from registers alone (the clock it holds under matrix load
            # included: the chip is power-limited well below its 2.4 GHz peak clock): measured here, right behind the timed
            # region, best of three ~5 ms launches; `peak` / `frac` above stay the guide's nominal 2.5 PFLOP/s figure
            try:
                # (measured in front of the timed region - see above - and once more behind it: the better of the two)
                sus = sustained_mfma_tflops(lib, dev)
                if sus_pre and sus_pre[0] > sus:
                    sus, sustained_mfma_tflops.clock_mhz, sustained_mfma_tflops.per_clk = sus_pre
                for rk in ("roofline", "roofline_gat"):
                    r_ = res.get(rk)
                    if r_ and r_.get("bound") == "mfma" and "issued_tflops" in r_:
                        r_["sustained_f16_mfma_tflops_measured"] = round(sus, 1)
                        # the measurement's own evidence: the core clock the chip held inside that launch (read by the kernel)
                        # and the flop per clock and SIMD it amounts to - 1024 = a matrix pipe that never idles
                        r_["sustained_clock_mhz"] = round(sustained_mfma_tflops.clock_mhz, 1)
                        r_["sustained_flop_per_clk_per_simd"] = round(sustained_mfma_tflops.per_clk, 1)
                        r_["sustained_frac_of_issue_limit"] = round(sustained_mfma_tflops.per_clk / 1024.0, 4)
                        r_["issued_frac_of_sustained"] = round(r_["issued_tflops"] / sus, 4)
                        r_["frac_of_sustained"] = round(r_["achieved"] / sus, 4)
                        r_["sustained_note"] = ("magat_mfma_sustained_f16: v_mfma_f32_32x32x16_f16 from registers only, one wave "
                                                "per SIMD on every CU, operand bits toggling - the matrix-core rate this box holds "
                                                "at the clock its power limit allows; issued = 3 f16 MFMAs per fp32 multiply-add")
            except Exception as e_:
                res["sustained_mfma_error"] = repr(e_)[:160]
            res["kernel_time_ms_per_step"] = round(sum(v["ms_per_step"] for k, v in table.items() if k != "gat_prepare"), 4)
            res["instrumented_ms_per_step"] = round(instr_ms, 4)
            res["kernel_timing"] = ("hipEvent pairs around every library launch, on the launch stream, in a second pass of the "
                                    "same %d steps right behind the timed region (the pairs cost 2-3 %% of a step: the timed "
                                    "region runs without them)" % args.steps)
    # ---- extra legs (N=1 only; never `value`) -----------------------------------------------------------------------------
    if world == 1 and rank == 0 and not args.no_extra_legs and args.workload == "c3" and not args.batch:
        esteps, ewarm = max(5, min(args.steps, 20)), 3
        # (a) the north-star shape: N=100, batch 1024, default arithmetic
        Bn = 1024
        xn = fov_states(Bn, N, seed=7).to(dev)
        Sn = comm_gso(Bn, N, map_w, seed=8).to(dev)
        el, kn, _ = run_leg(xn, Sn, esteps, ewarm, timing, repeats=2)
        ns = {"workload": "N=100, K=3, P=4, batch 1024 (north-star target shape), same model", "steps": esteps,
              "value": round(Bn * N * esteps / el, 1), "unit": "agent-steps/s", "ms_per_step": round(el / esteps * 1e3, 4)}
        if timing:
            tn = kernel_table(kn, esteps, Bn, N, Sn, {})
            for gk in ("gat_graph", "gat_layer (one launch)"):
                if gk in tn:
                    ns["gat_kernel"] = {k: tn[gk][k] for k in ("avg_us", "bound", "achieved", "peak", "unit", "frac",
                                                                "bytes_per_agent_step") if k in tn[gk]}
                    ns["gat_kernel"]["kernel"] = gk
        res["north_star_b1024"] = ns
        del xn, Sn
        # (b) the other single-GPU configs of BASELINE.json as their own (small) legs: c2 (N=20, batch 1024) and c5 (N=1000,
        # CSR GSO, bf16 storage inside the graph layer), each with its own model; graph-layer kernels against their roofs
        for wl in ("c2", "c5"):
            Bw, Nw, mw, Kw, Pw, Gw, bmw, cnw, ccw = WORKLOADS[wl]
            cfgw = make_config(num_agents=Nw, nGraphFilterTaps=Kw, nAttentionHeads=Pw, bottleneckFeature=Gw,
                               bottleneckMode=bmw, CNN_mode=cnw, AttentionConcat=ccw, device=str(dev),
                               gat_storage=GAT_STORAGE.get(wl, "fp32"))
            netw = build_model(cfgw, dev)
            xw = fov_states(Bw, Nw, seed=11).to(dev)
            Sw = comm_gso(Bw, Nw, mw, seed=12).to(dev)
            wsteps = esteps if wl == "c2" else max(3, esteps // 2)
            el, kw, _ = run_leg(xw, Sw, wsteps, 6, timing, net=netw, repeats=2)
            leg = {"workload": "%s: N=%d, %dx%d map, K=%d, P=%d, F=%d, batch %d%s" % (
                       wl, Nw, mw, mw, Kw, Pw, Gw, Bw, ", CSR GSO, bf16 storage in the graph layer" if wl == "c5" else ""),
                   "steps": wsteps, "value": round(Bw * Nw * wsteps / el, 1), "unit": "agent-steps/s",
                   "ms_per_step": round(el / wsteps * 1e3, 4), "ms_per_step_device": round(getattr(run_leg, "device_ms", 0.0), 4)}
            if timing:
                tw = kernel_table(kw, wsteps, Bw, Nw, Sw, {}, cfg=cfgw)
                keep = ("avg_us", "ms_per_step", "launches", "bound", "achieved", "peak", "unit", "frac", "bytes_per_agent_step")
                leg["kernels"] = {k: {q: v[q] for q in keep if q in v} for k, v in tw.items()
                                  if k.startswith("gat_") or k in ("gso_to_csr", "range_guard", "head_mean")}
                leg["kernel_time_ms_per_step"] = round(sum(v["ms_per_step"] for v in tw.values()), 4)
                if wl == "c5":
                    # the graph LAYER against SURVEY 8(d)'s layer-level bytes (2 G + 2 P F + 4 (1 + deg) per agent-step in bf16):
                    # every launch between compressMLP's rows and the action head - row cast, (maps GEMM,) degree ranking,
                    # score and hop kernels; `us_with_structure_build` adds the GSO -> CSR + CSC pass of addGSO
                    degw = float((Sw != 0).sum().item()) / (Bw * Nw)
                    lb = 2 * Gw + 2 * Pw * Gw + 4 * (1 + degw)
                    lus = 1e3 * sum(v["ms_per_step"] for k, v in tw.items() if k in ("gat_graph", "gat_maps_gemm", "gat_cast"))
                    sus_ = lus + 1e3 * tw.get("gso_to_csr", {}).get("ms_per_step", 0.0)
                    leg["gat_layer"] = {"us": round(lus, 1), "us_with_structure_build": round(sus_, 1),
                                        "bytes_per_agent_step": round(lb, 1), "edges_per_agent": round(degw, 2),
                                        "frac": round(lb * Bw * Nw / (lus * 1e-6) / (PEAK_HBM_GBS * 1e9), 4),
                                        "frac_with_structure_build": round(lb * Bw * Nw / (sus_ * 1e-6) / (PEAK_HBM_GBS * 1e9), 4),
                                        "maps_in_memory": "gat_maps_gemm" in tw}
            res[wl] = leg
            del netw, xw, Sw
            torch.cuda.empty_cache()
        # (c) the published model setting (N=10, K=2, 4 heads x 32 features, head-mean) as its own leg
        Bw, Nw, mw, Kw, Pw, Gw, bmw, cnw, ccw = WORKLOADS["published_f32p4"]
        cfgw = make_config(num_agents=Nw, nGraphFilterTaps=Kw, nAttentionHeads=Pw, bottleneckFeature=Gw, bottleneckMode=bmw,
                           CNN_mode=cnw, AttentionConcat=ccw, device=str(dev), gat_storage="fp32")
        netw = build_model(cfgw, dev)
        xw, Sw = fov_states(Bw, Nw, seed=13).to(dev), comm_gso(Bw, Nw, mw, seed=14).to(dev)
        el, kw, _ = run_leg(xw, Sw, esteps, 24, timing, net=netw, repeats=2)      # (a fresh model's first ~15 steps can carry a one-off 60-90 ms runtime stall: tools/exp/pub_probe.py)
        leg = {"workload": "published MAGAT F-32-P4 (scripts/train_DMap.sh:42): N=%d, %dx%d map, K=%d, P=%d, G=F=%d, head-mean, "
                           "BottomNeck_only, batch %d" % (Nw, mw, mw, Kw, Pw, Gw, Bw),
               "steps": esteps, "value": round(Bw * Nw * esteps / el, 1), "unit": "agent-steps/s",
               "ms_per_step": round(el / esteps * 1e3, 4), "ms_per_step_device": round(getattr(run_leg, "device_ms", 0.0), 4),
               "graph_layer_one_launch": bool(lib.magat_gat_one_launch_supported(Nw, Gw, Gw, Kw, 0, 0))}
        if timing:
            tw = kernel_table(kw, esteps, Bw, Nw, Sw, {}, cfg=cfgw)
            keep = ("avg_us", "ms_per_step", "launches", "bound", "achieved", "peak", "unit", "frac", "bytes_per_agent_step")
            leg["kernels"] = {k: {q: v[q] for q in keep if q in v} for k, v in tw.items()}
        res["published_f32p4"] = leg
        del netw, xw, Sw
        torch.cuda.empty_cache()
        # (c2) the same published widths on the README's 100-robot generalisation set (README.md:372-390)
        Bw, Nw, mw, Kw, Pw, Gw, bmw, cnw, ccw = WORKLOADS["published_f32p4_n100"]
        cfgw = make_config(num_agents=Nw, nGraphFilterTaps=Kw, nAttentionHeads=Pw, bottleneckFeature=Gw, bottleneckMode=bmw,
                           CNN_mode=cnw, AttentionConcat=ccw, device=str(dev), gat_storage="fp32")
        netw = build_model(cfgw, dev)
        xw, Sw = fov_states(Bw, Nw, seed=15).to(dev), comm_gso(Bw, Nw, mw, seed=16).to(dev)
        el, kw, _ = run_leg(xw, Sw, esteps, 24, timing, net=netw, repeats=2)      # (a fresh model's first ~15 steps can carry a one-off 60-90 ms runtime stall: tools/exp/pub_probe.py)
        leg = {"workload": "published MAGAT F-32-P4 on the 100-robot set (README.md:372-390): N=%d, %dx%d map, K=%d, P=%d, G=F=%d, "
                           "head-mean, batch %d" % (Nw, mw, mw, Kw, Pw, Gw, Bw),
               "steps": esteps, "value": round(Bw * Nw * esteps / el, 1), "unit": "agent-steps/s",
               "ms_per_step": round(el / esteps * 1e3, 4),
               "graph_layer_one_launch": bool(lib.magat_gat_one_launch_supported(Nw, Gw, Gw, Kw, 0, 0))}
        if timing:
            tw = kernel_table(kw, esteps, Bw, Nw, Sw, {}, cfg=cfgw)
            keep = ("avg_us", "ms_per_step", "launches", "bound", "achieved", "peak", "unit", "frac", "bytes_per_agent_step")
            leg["kernels"] = {k: {q: v[q] for q in keep if q in v} for k, v in tw.items() if k.startswith("gat_") or k == "head_mean"}
        res["published_f32p4_n100"] = leg
        del netw, xw, Sw
        torch.cuda.empty_cache()
        # (c3) the reference's OWN inference loop: one planning instance per step ("test_batch_size": 1,
        # configs/dcpGAT_OE_Random.json:58; agents/decentralplannerlocal_OnlineExpert_GAT.py:1030-1055): addGSO (float64 GSO, as the
        # simulator hands it) + forward + copy of the logits to the host, wall time per step; and the device time of the same
        # step back to back without the copy
        try:
            lat = {"what": "batch 1: addGSO(float64 S) + forward + logits.cpu() per step, wall clock; K=3, P=4, BottomNeck_skipConcat",
                   "timed_steps": 300}
            # (+ the published MAGAT F-32-P4 checkpoint shape - bottleneck 32, K = 2, head-mean, BottomNeck_only - in the same loop)
            for key, Nl, ml, kw_ in (("N10", 10, 20, {}), ("N100", 100, 50, {}),
                                     ("published_f32p4_N10", 10, 20, dict(nGraphFilterTaps=2, bottleneckFeature=32, AttentionConcat=False,
                                                                          bottleneckMode="BottomNeck_only")),
                                     ("published_f32p4_N100", 100, 50, dict(nGraphFilterTaps=2, bottleneckFeature=32, AttentionConcat=False,
                                                                            bottleneckMode="BottomNeck_only")),
                                     # (head MEAN at the default width: what main.py runs unless --AttentionConcat is given,
                                     #  main.py:115, utils/config.py:122)
                                     ("headmean_N10", 10, 20, dict(AttentionConcat=False)),
                                     ("headmean_N100", 100, 50, dict(AttentionConcat=False))):
                cfgl = make_config(**dict(dict(num_agents=Nl, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat",
                                               device=str(dev)), **kw_))
                netl = build_model(cfgl, dev)
                xl, Sl = fov_states(1, Nl, seed=17).to(dev), comm_gso(1, Nl, ml, seed=18, dtype=torch.float64).to(dev)
                with torch.no_grad():
                    for _ in range(30):
                        netl.addGSO(Sl)
                        netl(xl).cpu()
                    torch.cuda.synchronize(dev)
                    ts = []
                    for _ in range(300):
                        t0_ = time.perf_counter()
                        netl.addGSO(Sl)
                        netl(xl).cpu()
                        ts.append((time.perf_counter() - t0_) * 1e6)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(200):
                        netl.addGSO(Sl)
                        netl(xl)
                    e1.record()
                    torch.cuda.synchronize(dev)
                    # which forms one step takes (the one-agent-per-workgroup encoder with stem, head and guard inside: one launch)
                    lib.magat_form_reset()
                    netl.addGSO(Sl)
                    netl(xl)
                    forms_l = {k: int(lib.magat_form_count(v)) for k, v in nat.FORMS.items() if lib.magat_form_count(v)}
                ts.sort()
                lat[key] = {"median_us": round(ts[150], 1), "mean_us": round(sum(ts) / len(ts), 1), "p90_us": round(ts[270], 1),
                                   "device_back_to_back_us": round(e0.elapsed_time(e1) * 1e3 / 200, 1), "forms": forms_l}
                del netl, xl, Sl
            res["latency_b1"] = lat
        except Exception as e:
            res["latency_b1"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
        # (c4) the CLOSED LOOP on the device (multiRobotSimNew's getGSO / getCurrentState / move around the
        # forward, utils/new_simulator.py:279-321, 334-549, 745-806): per step GSO from the positions (with lambda_max) -> FOV state
        # tensors -> addGSO + forward -> action decode + collision shielding + position update; c3's model, 512 x 100 agents on a
        # 50 x 50 map, and one instance of 100 agents (the reference's own loop shape).  Not part of `value`.
        try:
            import numpy as np
            from magat_pathplanning_amd.simulator import batched_fov_states, batched_gso, batched_move
            cl = {"what": "batched_gso(normalize) -> batched_fov_states -> addGSO + forward -> batched_move, positions on the device"}
            rngc = np.random.default_rng(3)
            mapc = (rngc.random((50, 50)) < 0.08).astype(np.uint8)
            freec = np.argwhere(mapc == 0)
            for key, Bc, steps_c in (("b512_n100", 512, 20), ("b1_n100", 1, 200)):
                posc = np.stack([freec[rngc.permutation(len(freec))[:100]] for _ in range(Bc)]).astype(np.int32)
                goalc = np.stack([freec[rngc.permutation(len(freec))[:100]] for _ in range(Bc)]).astype(np.int32)
                dm, dpos, dgoal = torch.from_numpy(mapc).to(dev), torch.from_numpy(posc).to(dev).contiguous(), torch.from_numpy(goalc).to(dev)
                best = None
                with torch.no_grad():
                    for rep in range(3):
                        for _ in range(4 if rep == 0 else 0):
                            net.addGSO(batched_gso(dpos, 7.0))
                            batched_move(dm, dpos, logits=net(batched_fov_states(dm, dpos, dgoal, 9)), goal=dgoal)
                        torch.cuda.synchronize(dev)
                        t0_ = time.perf_counter()
                        for _ in range(steps_c):
                            net.addGSO(batched_gso(dpos, 7.0))
                            batched_move(dm, dpos, logits=net(batched_fov_states(dm, dpos, dgoal, 9)), goal=dgoal)
                        torch.cuda.synchronize(dev)
                        el_c = (time.perf_counter() - t0_) / steps_c
                        best = el_c if best is None or el_c < best else best
                cl[key] = {"ms_per_step": round(best * 1e3, 4), "value": round(Bc * 100 / best, 1), "unit": "agent-steps/s", "timed_steps": steps_c}
            res["closed_loop"] = cl
        except Exception as e:
            res["closed_loop"] = {"error": repr(e)[:200]}
        # (c4b) c3 with head MEAN (AttentionConcat = False: what main.py runs unless --AttentionConcat is given, main.py:115,
        # utils/config.py:122): same batch, the graph layer's heads averaged, the action head over 256 instead of 640 inputs
        try:
            cfgm = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", AttentionConcat=False,
                               device=str(dev))
            netm = build_model(cfgm, dev)
            elm, _, _ = run_leg(x, S, esteps, ewarm, False, net=netm, repeats=2)
            res["c3_headmean"] = {"workload": "c3 with AttentionConcat = False (head mean)", "value": round(B * N * esteps / elm, 1),
                                  "unit": "agent-steps/s", "ms_per_step": round(elm / esteps * 1e3, 4)}
            del netm
        except Exception as e:
            res["c3_headmean"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
        # (c5) the default width on more than 105 agents (400 x 128): the graph layer as one launch (gat_mid.hip, X fragments in
        # registers; before round 6g: the CSR kernels)
        try:
            cfgw = make_config(num_agents=128, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device=str(dev))
            netw = build_model(cfgw, dev)
            xw, Sw = fov_states(400, 128, seed=21).to(dev), comm_gso(400, 128, 50, seed=22).to(dev)
            elw, _, _ = run_leg(xw, Sw, esteps, ewarm, False, net=netw, repeats=2)
            res["n128"] = {"workload": "400 x 128 agents, K=3, P=4, BottomNeck_skipConcat", "value": round(400 * 128 * esteps / elw, 1),
                           "unit": "agent-steps/s", "ms_per_step": round(elw / esteps * 1e3, 4),
                           "graph_layer_one_launch": bool(lib.magat_gat_one_launch_supported(128, 128, 128, 3, 0, 1))}
            del netw, xw, Sw
        except Exception as e:
            res["n128"] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
        # (d) STRICT float32: every product on the float32 matrix cores (v_mfma_f32_32x32x2_f32, 157.3 TF peak) - no split
        # planes anywhere (CONV_SPLIT=0, HEAD_F16=0, GAT_SPLIT=0, GAT_MFMA=0).  The headline's f16x3 arithmetic is fp32-CLASS
        # (22-bit products, fp32 accumulation, measured error = this form's); this leg is the number a reader who wants
        # IEEE float32 products gets from the same library
        strict = {"CONV_SPLIT": 0, "HEAD_F16": 0, "GAT_SPLIT": 0, "GAT_MFMA": 0}
        for k_, v_ in strict.items():
            nat.set_option(k_, v_)
        nets = build_model(cfg, dev)
        el, ks_, _ = run_leg(x, S, esteps, ewarm, timing, net=nets, repeats=2)
        f32 = {"options": strict, "steps": esteps, "value": round(B * N * esteps / el, 1), "unit": "agent-steps/s",
               "ms_per_step": round(el / esteps * 1e3, 4), "ms_per_step_device": round(getattr(run_leg, "device_ms", 0.0), 4),
               "arithmetic": ARITH["f32"][1]}
        if timing:
            ts = kernel_table(ks_, esteps, B, N, S, {})
            bounded = [k for k, v in ts.items() if "bound" in v and k != "gat_prepare"]
            if bounded:
                dom = max(bounded, key=lambda k: ts[k]["ms_per_step"])
                f32["roofline"] = roof(ts, dom, B, N, {})
            f32["kernels_ms_per_step"] = {k: v["ms_per_step"] for k, v in ts.items()}
        res["f32_strict"] = f32
        for k_ in strict:
            nat.reset_option(k_)
        del nets
        torch.cuda.empty_cache()
        # (f) a TRAINING step (forward + cross-entropy + backward + SGD, train mode; agents/..._GAT.py:556-567) with the
        # CNN's convolutions and BatchNorm on this library's kernels (train_cnn.py) and on torch's (MIOpen / ATen); the graph
        # layer on its HIP forward / backward either way.  Not part of `value`.
        try:
            res["train_step"] = train_step_leg(dev)
        except Exception as e:          # (a reported extra: never takes the bench line down)
            res["train_step"] = {"error": repr(e)[:200]}
        # (g) the extra legs once more, compact, INSIDE `roofline` - the object a driver keeps whole
        if "roofline" in res:
            def _g(d, *ks):
                for k_ in ks:
                    d = d.get(k_, {}) if isinstance(d, dict) else {}
                return d if d != {} else None
            c2k, c5k = _g(res, "c2", "kernels") or {}, _g(res, "c5", "kernels") or {}
            g3 = res.get("roofline_gat", {})
            res["roofline"]["legs"] = {
                "c2": {"value": _g(res, "c2", "value"), "ms_per_step": _g(res, "c2", "ms_per_step"),
                       "gat_us": _g(c2k, "gat_layer (one launch)", "avg_us"), "gat_frac": _g(c2k, "gat_layer (one launch)", "frac"),
                       "gat_bound": _g(c2k, "gat_layer (one launch)", "bound")},
                "c5": {"value": _g(res, "c5", "value"), "ms_per_step": _g(res, "c5", "ms_per_step"),
                       "gat_graph_us_per_step": None if "gat_graph" not in c5k else round(1e3 * c5k["gat_graph"]["ms_per_step"], 1),
                       "gat_graph_frac": _g(c5k, "gat_graph", "frac"), "gat_layer_us": _g(res, "c5", "gat_layer", "us"),
                       "gat_layer_frac": _g(res, "c5", "gat_layer", "frac"),
                       "gat_layer_bytes_per_agent_step": _g(res, "c5", "gat_layer", "bytes_per_agent_step"),
                       "maps_in_memory": _g(res, "c5", "gat_layer", "maps_in_memory")},
                "b1024": {"value": _g(res, "north_star_b1024", "value"), "gat_frac": _g(res, "north_star_b1024", "gat_kernel", "frac")},
                "f32_strict": {"value": _g(res, "f32_strict", "value"), "frac": _g(res, "f32_strict", "roofline", "frac"),
                               "kernel_us": _g(res, "f32_strict", "roofline", "avg_us")},
                "gat_c3": {"us": g3.get("avg_us"), "bound": g3.get("bound"), "frac": g3.get("frac"),
                           "frac_hbm": None if "algorithmic_gbs" not in g3 else round(g3["algorithmic_gbs"] / PEAK_HBM_GBS, 4),
                           "frac_mfma": None if "algorithmic_tflops" not in g3 else round(g3["algorithmic_tflops"] / PEAK_16_TFLOPS, 4)},
                "published_f32p4": {"value": _g(res, "published_f32p4", "value"),
                                    "one_launch": _g(res, "published_f32p4", "graph_layer_one_launch")},
                "published_f32p4_n100": {"value": _g(res, "published_f32p4_n100", "value"),
                                         "one_launch": _g(res, "published_f32p4_n100", "graph_layer_one_launch")},
                "closed_loop": {"b512_n100_value": _g(res, "closed_loop", "b512_n100", "value"),
                                "b512_n100_ms": _g(res, "closed_loop", "b512_n100", "ms_per_step"),
                                "b1_n100_ms": _g(res, "closed_loop", "b1_n100", "ms_per_step")},
                "c3_headmean": {"value": _g(res, "c3_headmean", "value"), "ms_per_step": _g(res, "c3_headmean", "ms_per_step")},
                "n128": {"value": _g(res, "n128", "value"), "ms_per_step": _g(res, "n128", "ms_per_step"),
                         "one_launch": _g(res, "n128", "graph_layer_one_launch")},
                "latency_b1_us": {"N10": _g(res, "latency_b1", "N10", "median_us"), "N100": _g(res, "latency_b1", "N100", "median_us"),
                                  "published_f32p4_N10": _g(res, "latency_b1", "published_f32p4_N10", "median_us"),
                                  "published_f32p4_N100": _g(res, "latency_b1", "published_f32p4_N100", "median_us"),
                                  "headmean_N10": _g(res, "latency_b1", "headmean_N10", "median_us"),
                                  "headmean_N100": _g(res, "latency_b1", "headmean_N100", "median_us"),
                                  "N10_device": _g(res, "latency_b1", "N10", "device_back_to_back_us"),
                                  "N100_device": _g(res, "latency_b1", "N100", "device_back_to_back_us")}}
    # ---- the CPU baseline: the pinned oracle on this box's host cores, behind every GPU leg (nothing of it is inside a timed region)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        torch.cuda.synchronize(dev)
        try:
            cpu = cpu_baseline(cfg, {k: v.detach().cpu() for k, v in net.state_dict().items()}, N, map_w)
            res["cpu_baseline"] = cpu
            if "north_star_b1024" in res:
                res["north_star_b1024"]["vs_cpu_baseline"] = round(res["north_star_b1024"]["value"] / cpu["value"], 1)
        except Exception as e:          # (the GPU results above are measured: a failing host leg must not lose the line)
            res["cpu_baseline"] = {"value": None, "unit": "agent-steps/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
