#!/usr/bin/env python
"""bench.py -- agent-steps/s of the batched MAGAT GAT forward on MI355X (contract: see task brief).

A "step" is one addGSO(S) + forward(x) of DecentralPlannerGATNet over one batch of synthetic planning
instances already resident in HBM.  Workload at N GPUs = BASELINE.json configs[2]/[3]: 100 agents,
50x50 map, K=3, P=4, F=128, bottleneck + SkipConcat, KeyQuery, head concat, batch 512 PER GPU (weak
scaling: 8 GPUs = the 4096-instance config).  Instances are independent, so ranks never communicate
inside a step; the only collectives are the barrier and the MAX of the elapsed time.

Rank 0 prints ONE JSON line.  Extra objects: `roofline` (dominant kernel of the timed region, timed with
hipEvents on the launch stream through the library's profiling hooks), `roofline_gat` (the hand-written
graph kernel the north star names, against HBM), `kernels` (every kernel tag), `cpu_baseline` (the pinned
CPU oracle, kind "port", timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_F32_TFLOPS = 157.3    # fp32 MFMA (v_mfma_f32_32x32x2_f32) = fp32 vector peak
PEAK_FP8_TFLOPS = 5000.0   # dense MXFP8 (v_mfma_scale_f32_32x32x64_f8f6f4), MI355X_MICROARCH.md
PEAK_BF16_TFLOPS = 2500.0  # dense 16-bit MFMA (v_mfma_f32_32x32x16_{f16,bf16}); f16x3 issues 3, bf16x6 6 such flops per fp32 flop

WORKLOADS = {
    # name: (B per GPU, N, map_w, K, P, G, bottleneckMode, CNN_mode, concat)
    "c3": (512, 100, 50, 3, 4, 128, "BottomNeck_skipConcat", "ResNetLarge_withMLP", True),
    "c2": (1024, 20, 28, 3, 4, 128, "BottomNeck_only", "ResNetLarge_withMLP", True),
    "c1": (64, 10, 20, 2, 1, 128, "BottomNeck_only", "ResNetLarge_withMLP", True),
    "n100b1024": (1024, 100, 50, 3, 4, 128, "BottomNeck_skipConcat", "ResNetLarge_withMLP", True),
    # config 5: large sparse graph -> CSR kernels, bf16 storage inside the GAT layer; c5f32 = same shape, fp32 storage
    "c5": (128, 1000, 160, 2, 4, 128, "BottomNeck_only", "ResNetLarge_withMLP", True),
    "c5f32": (128, 1000, 160, 2, 4, 128, "BottomNeck_only", "ResNetLarge_withMLP", True),
}
GAT_STORAGE = {"c5": "bf16"}


def valid_taps(hin, hout, stride, k=3, pad=1):
    """sum over output pixels of the number of 3x3 taps that fall inside the input (1-D count squared)."""
    one = sum(sum(1 for t in range(k) if 0 <= o * stride - pad + t < hin) for o in range(hout))
    return one * one


def split_tags(cfg):
    """Kernel tags that run on the bf16x6 split-MFMA kernel with the library's defaults (MAGAT_CONV_SPLIT mask,
    MAGAT_GAT_SPLIT), mirroring csrc/encoder_f32.hip::enc_split_mask and csrc/gat_f32.hip::gat_maps_gemm."""
    mask = int(os.environ.get("MAGAT_CONV_SPLIT", "7"))
    tags = set()
    for l in range(3):
        if mask >> l & 1:
            tags |= {2 + 2 * l, 3 + 2 * l}
    G, K, P = cfg.bottleneckFeature, cfg.nGraphFilterTaps, cfg.nAttentionHeads
    nc = P * G + P * K * G if cfg.attentionMode == "KeyQuery" else (P * K * G + 2 * P + 31) // 32 * 32
    if int(os.environ.get("MAGAT_GAT_SPLIT", "1")) and nc % 32 == 0 and G % 32 == 0:
        tags.add(10)
    return tags


def mx_tags(cfg):
    """Kernel tags whose f16x3 GEMM runs as "f16 + MX correction" (encoder_f32.hip: every conv whose input is a block output,
    i.e. layer2.* and layer3.* of the ResNet encoders, plane-granule chain on, MAGAT_CONV_MX != 0)."""
    if not cfg.CNN_mode.startswith("ResNet") or not int(os.environ.get("MAGAT_CONV_MX", "1")):
        return set()
    if not (int(os.environ.get("MAGAT_CONV_PCHAIN", "1")) and int(os.environ.get("MAGAT_CONV_DIRECT", "1"))):
        return set()
    return {4, 5, 6, 7}


def per_agent_work(cfg, N, S_bytes, deg=None, planned=False):
    """Algorithmic work per AGENT-STEP for each kernel tag: (flops, hbm_bytes, bound).  deg: mean out-degree, given
    when the layer runs on the CSR kernels (N > 128 or bf16 storage).  planned: the GSO plan (magat_gat_gso_plan, made
    at addGSO on a side stream) read S instead of the graph kernel, which then reads 16 B of edge mask per agent -
    the S bytes are priced on the plan kernel (tag 16), not on the graph kernel."""
    G, K, P = cfg.bottleneckFeature, cfg.nGraphFilterTaps, cfg.nAttentionHeads
    F = G
    nfm = cfg.numInputFeatures
    NC = P * G + P * K * F
    t11, t6 = valid_taps(11, 6, 2), valid_taps(6, 6, 1)
    w = {}
    w[1] = (2 * 121 * 27 * 32, 4 * (363 + 121 * 32), "hbm")
    chans = [(32, 32), (32, 64), (64, 128)]
    for l, (ci, co) in enumerate(chans):
        taps = t11 if l == 0 else t6
        hw_in = 121 if l == 0 else 36
        w[2 + 2 * l] = (2 * taps * ci * co, 4 * (hw_in * ci + 36 * co), "mfma")
        w[3 + 2 * l] = (2 * (t6 * co * co + 36 * ci * co), 4 * (36 * co + hw_in * ci + 36 * co), "mfma")
    w[8] = (2 * 9 * 128 * nfm, 4 * (36 * 128 + nfm), "mfma")   # pooled on load: K = 9*128
    w[9] = (2 * nfm * G, 4 * (nfm + G), "mfma")
    w[10] = (2 * G * NC, 4 * (G + NC), "mfma")
    # graph kernel: SURVEY 8(d) "kernel (ii)" bytes per instance / N
    yw = P * F if cfg.AttentionConcat else F          # head-mean: one merged [N][F] row block is written
    w[11] = (0, (4 * (N * G + P * N * G + P * K * N * F + N * yw) + (16 * N if planned else S_bytes * N * N)) / N, "hbm")
    if planned:
        w[16] = (0, S_bytes * N + 16 + 8.0 / N, "hbm")
    if deg is not None:
        # CSR kernels: X, the hoisted maps Z and Y in the storage type (4 or 2 bytes), CSR + CSC index arrays, and the
        # attention values written by the score kernel and read once per hop
        es = 2 if getattr(cfg, "gat_storage", "fp32") == "bf16" else 4
        w[11] = (0, es * (G + P * G + P * K * F + yw) + 4 * (2 + 3 * deg) + 4 * P * deg * K, "hbm")
    width = yw + (nfm if cfg.bottleneckMode == "BottomNeck_skipConcat" else 0)
    w[12] = (2 * width * 5, 4 * (width + 5), "hbm")
    return w


def load_pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/<round>/summary_*.json, made by
    tools/profile_round.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 read correction).  PMC counters
    cannot be collected from inside this process, so the latest committed summary of the same workload is used."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "summary_*.json")))
    if not paths:
        return {}
    try:
        d = json.load(open(paths[-1]))
    except Exception:
        return {}
    out = {}
    for k, v in d.get("layers", {}).items():
        if "hbm_bytes_per_launch" in v:
            out[k] = {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "source": os.path.relpath(paths[-1], ROOT)}
    return out


def build_model(cfg, device, seed=1337):
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


def cpu_baseline(cfg, sd, N, map_w, budget_s=12.0):
    """Pinned CPU oracle (oracle/magat_oracle.py, the reference's dense op sequence in torch-CPU) on a
    bounded sample of the same workload: B=32 instances per forward, repeated for ~budget_s seconds."""
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states
    from oracle import magat_oracle as orc
    Bc = 32
    x = fov_states(Bc, N, seed=99)
    S = comm_gso(Bc, N, map_w, seed=98)
    sd_cpu = {k: v.detach().cpu() for k, v in sd.items()}
    threads = torch.get_num_threads()
    with torch.no_grad():
        orc.planner_forward(x, S.clone(), sd_cpu, cfg)      # warm-up
        t0 = time.perf_counter()
        reps = 0
        while True:
            orc.planner_forward(x, S.clone(), sd_cpu, cfg)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget_s or reps >= 200:
                break
    return {"value": round(Bc * N * reps / el, 1), "unit": "agent-steps/s", "cores": threads, "kind": "port",
            "sample": "oracle.planner_forward, B=%d N=%d (same model/config), %d forwards in %.1f s, torch-CPU %d threads"
                      % (Bc, N, reps, el, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed steps after >= 10 warm-ups
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override per-GPU batch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
    B, N, map_w, K, P, G, bmode, cnn, concat = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    cfg = make_config(num_agents=N, nGraphFilterTaps=K, nAttentionHeads=P, bottleneckFeature=G,
                      bottleneckMode=bmode, CNN_mode=cnn, AttentionConcat=concat, device=str(dev),
                      gat_storage=GAT_STORAGE.get(args.workload, "fp32"))
    net = build_model(cfg, dev)
    x = fov_states(B, N, seed=1337 + rank).to(dev)
    S = comm_gso(B, N, map_w, seed=4242 + rank).to(dev)       # float32, as the dataloader hands it over
    lib = nat.lib()

    def step():
        net.addGSO(S)
        return net(x)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.no_grad():
        for _ in range(args.warmup):
            out = step()
        timing = rank == 0 and not args.no_kernel_timing
        if timing:
            lib.magat_profile_reserve(32 * (args.steps + 1))      # event pairs created outside the timed region
            lib.magat_profile_reset()
            lib.magat_profile_enable(1)
        # One-off 50-90 ms host stalls were seen in about one run in fifteen on these boxes (GPU idle meanwhile): a full
        # collection of the interpreter's heap (torch + numpy put ~1e6 objects there) triggered by the step loop's small
        # allocations fits.  Collect now and move the survivors out of the collector's reach; every step still runs in full.
        gc.collect()
        gc.freeze()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = step()
        barrier()
        elapsed = time.perf_counter() - t0
        gc.unfreeze()
    if timing:
        lib.magat_profile_enable(0)
        lib.magat_profile_collect()
    assert out.shape == (B * N, 5) and bool(torch.isfinite(out).all())
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = B * N * world * args.steps / elapsed
        res = {"metric": "agent-steps/s (batched GAT forward)", "value": round(value, 1), "unit": "agent-steps/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if cfg.gat_storage == "fp32" else "f32 arithmetic, bf16 storage inside the GAT layer",
               "data": "synthetic (seeded binary FOV states + comm-radius GSO; random-init weights, BN stats perturbed)",
               "config": {"precision": "float32 in / float32 out, logits within 1e-4 of the reference (observed 5e-6..1e-5; 1e-6 with "
                                       "MAGAT_CONV_MX=0); dense maps on fp32 MFMA or split products with fp32 accumulate: f16x3 (2 f16 "
                                       "planes per value = 22 significand bits, 3 f16 MFMAs per product) and, for the layer2/layer3 "
                                       "convolutions, the main product h1*g1 on f16 MFMAs + both 2^-11-sized correction products in one "
                                       "block-scaled fp8 (e4m3) MFMA per slab",
                          "workload": "%s: N=%d agents, %dx%d map, K=%d, P=%d, F=%d, %s, %s, KeyQuery, %s; batch %d per GPU "
                                      "(global %d); resident inputs, addGSO+forward per step"
                                      % (args.workload, N, map_w, map_w, K, P, G, bmode, cnn,
                                         "head-concat" if concat else "head-mean", B, B * world),
                          "global_batch": B * world, "agents": N, "parallelism": "instance-sharded x%d" % world}}
        if timing:
            csr = N > 128 or cfg.gat_storage == "bf16"
            cnt16, tot16 = ctypes.c_longlong(0), ctypes.c_double(0.0)
            lib.magat_profile_read(16, ctypes.byref(cnt16), ctypes.byref(tot16))
            work = per_agent_work(cfg, N, 4, float((S != 0).sum().item()) / (B * N) if csr else None,
                                  planned=cnt16.value > 0 and not csr)
            TAG_OF = {v: k for k, v in nat.TAGS.items()}
            splits = split_tags(cfg)
            kernels, dom = {}, None
            agent_steps = B * N * args.steps
            for tag, name in nat.TAGS.items():
                cnt, tot = ctypes.c_longlong(0), ctypes.c_double(0.0)
                lib.magat_profile_read(tag, ctypes.byref(cnt), ctypes.byref(tot))
                if cnt.value == 0:
                    continue
                sec = tot.value / 1e3
                ent = {"launches": cnt.value, "avg_us": round(tot.value * 1e3 / cnt.value, 2),
                       "ms_per_step": round(tot.value / args.steps, 4)}
                if tag in work and sec > 0:
                    fl, by, bound = work[tag]
                    if bound == "mfma" and tag in splits:
                        # split-MFMA kernels: every fp32 multiply-add is issued as three f16 (f16x3, the default) or
                        # six bf16 (bf16x6, MAGAT_CONV_F16=0) matrix-core multiply-adds; price the kernel against the
                        # 16-bit matrix peak with the flops it actually issues, and keep the fp32-equivalent rate
                        nprod = 3 if int(os.environ.get("MAGAT_CONV_F16", "1")) else 6
                        ach = nprod * fl * agent_steps / sec / 1e12
                        peak, dt = PEAK_BF16_TFLOPS, ("f16 (f16x3 split, f32 accumulate)" if nprod == 3 else
                                                      "bf16 (bf16x6 split, f32 accumulate)")
                        if nprod == 3 and tag in mx_tags(cfg):
                            # "f16 + MX correction" (layer2 / layer3 convs): of the three products one is issued as f16 MFMAs
                            # and two inside a block-scaled fp8 MFMA at twice the f16 rate - the peak for the same three
                            # products is 3 / (1/2500 + 2/5000) TFLOP/s
                            peak = round(3.0 / (1.0 / PEAK_BF16_TFLOPS + 2.0 / PEAK_FP8_TFLOPS), 1)
                            dt = "f16 main product + 2 correction products in MXFP8 (e4m3, block-scaled MFMA), f32 accumulate"
                        ent.update(bound="mfma", mfma_dtype=dt, achieved=round(ach, 1),
                                   peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                                   f32_equiv_tflops=round(ach / nprod, 2), flops_per_agent_step=fl)
                    elif bound == "mfma":
                        ach = fl * agent_steps / sec / 1e12
                        ent.update(bound="mfma", mfma_dtype="f32", achieved=round(ach, 2), peak=PEAK_F32_TFLOPS,
                                   unit="TFLOP/s", frac=round(ach / PEAK_F32_TFLOPS, 4), flops_per_agent_step=fl)
                    else:
                        ach = by * agent_steps / sec / 1e9
                        ent.update(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                                   frac=round(ach / PEAK_HBM_GBS, 4), bytes_per_agent_step=round(by, 1))
                kernels[name] = ent
                if "bound" in ent and (dom is None or tot.value > dom[1]):
                    dom = (name, tot.value)
            if "conv_first" in kernels and "layer1.conv1" not in kernels and 2 in work:
                # plane chain (default): the stem and layer1.conv1 are ONE kernel (csrc/layer1_fused.hip) under the stem's
                # tag.  Algorithmic bytes per agent-step: the (3,H,W) input + the layer1.conv1 output + the stem's
                # stride-2 pixels for the residual branch; 205 issued flop/B, below the 312 flop/B ridge: priced on HBM.
                e = kernels.pop("conv_first")
                by = 4 * (3 * 121 + 2 * 36 * 32)
                sec = e["ms_per_step"] * args.steps / 1e3
                ach = by * agent_steps / sec / 1e9
                e.update(bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4),
                         bytes_per_agent_step=by, flops_per_agent_step=work[1][0] + work[2][0],
                         note="stem + layer1.conv1 fused: the 15.5 KB/agent stem output never reaches HBM")
                kernels = {"conv_first+layer1.conv1 (fused)": e, **kernels}
                if dom and dom[0] == "conv_first":
                    dom = None
                    for k, v in kernels.items():
                        if "bound" in v and (dom is None or v["ms_per_step"] > kernels[dom[0]]["ms_per_step"]):
                            dom = (k, v["ms_per_step"])
            res["kernels"] = kernels

            pmc = load_pmc_traffic() if args.workload == "c3" and not args.batch else {}

            def roof(name):
                e = kernels[name]
                tr = pmc.get(name)
                return {"kernel": name, "bound": e["bound"], "achieved": e["achieved"], "peak": e["peak"],
                        "unit": e["unit"], "frac": e["frac"],
                        "traffic": None if tr is None else round(tr["hbm_bytes_per_launch"]),
                        "traffic_source": None if tr is None else tr["source"],
                        "algorithmic_per_launch": round((e["bytes_per_agent_step"] if e["bound"] == "hbm" else
                                                         e["flops_per_agent_step"]) * B * N),
                        "mfma_dtype": e.get("mfma_dtype"), "f32_equiv_tflops": e.get("f32_equiv_tflops"),
                        "avg_us": e["avg_us"], "launches": e["launches"]}
            if dom:
                res["roofline"] = roof(dom[0])
            if "gat_graph" in kernels:
                res["roofline_gat"] = roof("gat_graph")
            if "gat_prepare" in kernels:
                kernels["gat_prepare"]["note"] = ("GSO plan, made at addGSO on a side stream: runs under the encoder, "
                                                  "not part of the main-stream kernel time")
            res["kernel_time_ms_per_step"] = round(sum(k["ms_per_step"] for n_, k in kernels.items()
                                                       if n_ != "gat_prepare"), 4)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg, net.state_dict(), N, map_w)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
