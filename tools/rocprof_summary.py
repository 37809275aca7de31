"""Folds rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE PMC passes) into small, committable
summaries:  gpurun_out/prof_<tag>/summary_<tag>.{txt,json}.  HBM bytes follow MI355X_MICROARCH.md section HBM:
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of a wide coalesced streaming read,
so read bytes = 2 * FETCH_SIZE * 1024 (WRITE_SIZE is taken as reported: uncalibrated, see the guide)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    import re
    m = re.search(r"conv_gemm_f16x3_direct_kernel<(\d+), (\d+), (0|1)(?:, \(bool\))?(?:, )?(true|false|0|1)?>", name)
    if m:                                       # f16x3, activations straight into registers (FUSE2: + a second layer in the epilogue)
        inf = {"0": "f32-in", "1": "planes-in"}[m.group(3)]
        fuse = m.group(4) in ("true", "1")
        return "conv_f16x3_direct<%s,TM%s,%s%s>" % (m.group(1), m.group(2), inf, ",+layer2" if fuse else "")
    m = re.search(r"conv_gemm_bf16x6_kernel<true, (\d+), \d+, \d+, 2>", name)
    if m:                                       # NPL = 2: the f16x3 flavour of the split kernel
        return "conv_gemm_f16x3<f32-in,%s>" % m.group(1)
    for key, s in (("conv_gemm_bf16x6_kernel<true, 128", "conv_gemm_bf16x6<f32-in,128>"),
                   ("conv_gemm_bf16x6_kernel<true, 64", "conv_gemm_bf16x6<f32-in,64>"),
                   ("conv_gemm_bf16x6_kernel<true, 32", "conv_gemm_bf16x6<f32-in,32>"),
                   ("conv_gemm_bf16x6_kernel<false", "conv_gemm_bf16x6<planes>"),
                   ("conv_gemm_kernel<64, 128, 2, 2, true", "conv_gemm_f32<64,128,pool>"),
                   ("conv_gemm_kernel<128, 128, 2, 2, true", "conv_gemm_f32<128,128,pool>"),
                   ("conv_gemm_kernel<64, 128", "conv_gemm_f32<64,128>"),
                   ("conv_gemm_kernel<128, 128", "conv_gemm_f32<128,128>"), ("conv_gemm_kernel<128, 64", "conv_gemm_f32<128,64>"),
                   ("conv_gemm_kernel<128, 32", "conv_gemm_f32<128,32>"), ("conv_first_kernel", "conv_first"),
                   ("layer1_fused_kernel", "layer1_fused(stem+layer1.conv1)"),
                   ("stem8_kernel", "stem8(stem+layer1.conv1, eight-agent groups)"),
                   
                   ("block_chain_w4_kernel", "block_chain_w4(layer1.conv2+layer2)"), ("block3_w4_kernel", "block3_w4(layer3+pool)"),
                   
                   ("block_full_p_kernel", "block_full_p(layer1.conv2+layer2+layer3+pool, pooling in registers)"),
                   ("gat_guard_count_kernel", "guard_count(gat)"), ("guard_count_kernel", "guard_count(encoder)"),
                   ("gat_mfma_kernel", "gat_mfma(one-launch KeyQuery layer)"), ("gat_dense_kernel", "gat_dense_kernel"), ("head_mean_relu", "head_mean_relu"),
                   ("gat_mid_kernel", "gat_mid(one-launch layer, G = 32 | 64, N = 33..128)"), ("gat_small_kernel", "gat_small(one-launch layer, N <= 32)"),
                   ("csr_fused_scores_kernel", "csr_fused_scores(q' on MFMA + edge scores + softmax)"),
                   ("csr_fused_hop_kernel", "csr_fused_hop(hop on X + taps on MFMA)"), ("csr_rank_kernel", "csr_rank(degree ranking)"),
                   ("csr_fused_pack_kernel", "csr_fused_pack"), ("gso_mask_x4_kernel", "gso_mask_x4(GSO -> bit matrix)"),
                   ("gso_mask_kernel", "gso_mask(GSO -> bit matrix)"), ("gso_structure_kernel", "gso_structure(CSR + CSC)"),
                   ("gso_totals_kernel", "gso_totals"), ("cast_f32_bf16_kernel", "cast_f32_bf16"),
                   ("pack_kernel", "gat_pack"), ("gso_prepare", "gso_prepare")):
        if key in name:
            return s
    return name[:60]


def find(root, pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


def timed_launch_stats(trace_csv, steps, warmup):
    """Per kernel, over the TIMED steps only: the bench command runs `warmup` untimed steps first (cold caches, first-touch
    page faults, lazy module loads), and rocprofv3's own --stats averages them in - which is how a profile's avg_ns came to
    sit 10 % above the bench line's hipEvent time.  Launches are taken in start order, the first warmup / (steps + warmup)
    of every kernel's launches dropped, min / median / mean of the rest reported."""
    by = defaultdict(list)
    for r in csv.DictReader(open(trace_csv)):
        by[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    for v in by.values():
        v.sort()
    # where the timed region starts: among the kernels launched exactly once per step, the one that comes first in a step;
    # its launch number `warmup` opens the first timed step (everything before it - warm-up steps, the one-off float32
    # calibration pass of the activation scales, weight packing - is dropped for EVERY kernel)
    # (bench.py also runs untimed steps in FRONT of the warm-up since round 6 - two set-up steps + the five `first_steps_ms` - so a
    #  once-per-step kernel has steps + warmup + 7 launches: the timed steps are its LAST `steps` launches)
    once = [v for v in by.values() if len(v) in (steps + warmup, steps + warmup + 7)]
    cut = min(once, key=lambda v: v[0][0])[-steps][0] if once else 0
    stats = {}
    for k, v in by.items():
        d = sorted(x[1] for x in v if x[0] >= cut)
        if d:
            stats[k] = dict(timed_launches=len(d), dropped_warmup_launches=len(v) - len(d), min_us=d[0], median_us=d[len(d) // 2],
                            mean_us=sum(d) / len(d), total_us=sum(d))
    return stats


def main():
    out, tag = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3].isdigit() else 5
    warmup = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].isdigit() else 2
    # optional: the bench line of the SAME command run untraced on the same box (tools/profile_round.sh writes it): its hipEvent
    # table is printed beside the traced durations (rocprofv3 stretches the long kernels by up to 10 %)
    bench_json = next((a for a in sys.argv[3:] if a.endswith(".json")), os.path.join(out, "bench_untraced.json"))
    res = {"tag": tag, "kernels": {}}
    lines = []
    tr0 = find(os.path.join(out, "trace"), "*kernel_trace.csv")
    if tr0:
        stats = timed_launch_stats(tr0, steps, warmup)
        tot = sum(v["total_us"] for v in stats.values()) or 1.0
        lines.append("== rocprofv3 --kernel-trace, TIMED steps only (%d steps after %d warm-up steps dropped): us per launch" % (steps, warmup))
        lines.append("   (compare `median_us` with the bench line's hipEvent `avg_us`; under rocprofv3 the long kernels run a few per cent")
        lines.append("    slower than in an untraced run: the tool serialises dispatches and the clocks follow the lower duty cycle)")
        for k, v in sorted(stats.items(), key=lambda kv: -kv[1]["total_us"]):
            lines.append("%-44s n=%5d min=%10.2f median=%10.2f mean=%10.2f  %5.1f %%" % (
                k, v["timed_launches"], v["min_us"], v["median_us"], v["mean_us"], 100.0 * v["total_us"] / tot))
            res["kernels"].setdefault(k, {}).update(v, pct_timed=100.0 * v["total_us"] / tot)
        res["timed_kernel_us_per_step"] = tot / steps
        lines.append("sum over kernels: %.1f us per step   (TRACED durations)" % (tot / steps))
    if os.path.exists(bench_json):
        try:
            bd = json.loads(open(bench_json).read().strip().splitlines()[-1])
            lines.append("== the same command UNTRACED on the same box (bench.py, hipEvent pairs around every launch; %d timed steps): us per launch"
                         % bd.get("steps", 0))
            for k, v in bd.get("kernels", {}).items():
                lines.append("%-48s launches=%4d avg_us=%10.2f  per step %8.4f ms%s" % (
                    k, v["launches"], v["avg_us"], v["ms_per_step"], ("  frac %.4f (%s)" % (v["frac"], v["bound"])) if "frac" in v else ""))
            lines.append("untraced: %.4f ms per step (host wall), %.4f (device), kernels %.4f; value %.0f agent-steps/s" % (
                bd["ms_per_step"], bd.get("ms_per_step_device", 0.0), bd.get("kernel_time_ms_per_step", 0.0), bd["value"]))
            res["untraced"] = {"ms_per_step": bd["ms_per_step"], "value": bd["value"],
                               "kernels": {k: {"avg_us": v["avg_us"], "launches": v["launches"]} for k, v in bd.get("kernels", {}).items()}}
        except Exception as e:
            lines.append("(untraced bench line unreadable: %r)" % (e,))
    st = find(os.path.join(out, "trace"), "*kernel_stats.csv")
    if st:
        lines.append("== rocprofv3 --kernel-trace --stats as the tool prints it (ALL launches, warm-up steps included): durations (ns)")
        with open(st) as f:
            for row in csv.DictReader(f):
                n = short(row["Name"])
                lines.append("%-28s calls=%6s avg_ns=%12s total_ns=%14s pct=%6s" % (
                    n, row["Calls"], row["AverageNs"], row["TotalDurationNs"], row["Percentage"]))
                res["kernels"].setdefault(n, {}).update(calls=int(row["Calls"]), avg_us_all_launches=float(row["AverageNs"]) / 1e3,
                                                        pct=float(row["Percentage"]))
    for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        cc = find(os.path.join(out, sub), "*counter_collection.csv")
        if not cc:
            continue
        acc = defaultdict(lambda: [0, 0.0])
        with open(cc) as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != ctr:
                    continue
                k = short(row["Kernel_Name"])
                acc[k][0] += 1
                acc[k][1] += float(row["Counter_Value"])
        lines.append("== rocprofv3 --pmc %s: per-launch average (KiB as reported)" % ctr)
        for k, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            lines.append("%-28s launches=%5d avg_KiB=%14.1f" % (k, n, tot / n))
            res["kernels"].setdefault(k, {})[ctr + "_KiB_per_launch"] = tot / n
    # per-layer view, keyed like the bench line's kernel table: every kernel NAME that belongs to a tag, its launches taken
    # in dispatch order and the LAST steps * (launches per step) of them kept - the warm-up steps in front also hold the
    # one-off float32 calibration pass and the weight packing.  (A positional "launch i of the step" mapping, as rounds 1-2
    # had it, silently mis-assigns everything behind the first launch the step gains or loses.)
    stem_kernel = "stem8_kernel"      # (option L1_FUSED = 2, the default since round 4; the row-band form: layer1_fused_kernel)
    try:
        if not any("stem8_kernel" in r["Kernel_Name"] for r in csv.DictReader(open(find(os.path.join(out, "trace"), "*kernel_trace.csv")))):
            stem_kernel = "layer1_fused_kernel"
    except Exception:
        pass
    TAGS = [("conv_first+layer1.conv1 (fused)", stem_kernel, 1, 0),
            ("layer1.conv2+layer2+layer3 (fused, pooled)", "block_full_", 1, 0),
            ("head+compressMLP (one launch)", "+layer2>", 1, 0),      # (matched on the SHORT name: see per_tag)
            ("gat_layer (one launch)", "gat_mfma_kernel", 1, 0),
            ("actionsMLP", "skinny_gemm_kernel", 1, 0)]
    SEQ = [t[0] for t in TAGS]
    layers = defaultdict(dict)

    def per_tag(rows, order_key, value, nsteps):
        """rows of one csv -> {tag: [values of the last nsteps steps]}"""
        got = {}
        for tag, needle, per_step, which in TAGS:
            mine = [r for r in rows if needle in r["Kernel_Name"] or needle in short(r["Kernel_Name"])]
            mine.sort(key=order_key)
            mine = mine[-nsteps * per_step:]
            if len(mine) == nsteps * per_step:
                got[tag] = [value(r) for i, r in enumerate(mine) if i % per_step == which]
        return got

    tr = find(os.path.join(out, "trace"), "*kernel_trace.csv")
    if tr:
        rows = list(csv.DictReader(open(tr)))
        for k, v in per_tag(rows, lambda r: int(r["Start_Timestamp"]),
                            lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, steps).items():
            layers[k]["avg_us"] = sum(v) / len(v)
            layers[k]["launches"] = len(v)
    for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        cc = find(os.path.join(out, sub), "*counter_collection.csv")
        if not cc:
            continue
        rows = [r for r in csv.DictReader(open(cc)) if r.get("Counter_Name") == ctr]
        key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
        pmc_steps = 2                                    # (tools/profile_round.sh: the PMC passes time 2 steps after 1 warm-up)
        for k, v in per_tag(rows, (lambda r: int(r[key])) if key else (lambda r: 0), lambda r: float(r["Counter_Value"]),
                            pmc_steps).items():
            layers[k][ctr + "_KiB_per_launch"] = sum(v) / len(v)
    lines.append("== per-layer view (dispatch order within a step): avg_us | HBM read MB (2*FETCH) | write MB")
    for k in SEQ:
        v = layers.get(k)
        if not v:
            continue
        rd = 2.0 * v.get("FETCH_SIZE_KiB_per_launch", 0.0) * 1024.0
        wr = v.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024.0
        v["hbm_read_bytes_per_launch"], v["hbm_write_bytes_per_launch"], v["hbm_bytes_per_launch"] = rd, wr, rd + wr
        lines.append("%-26s avg_us=%10.2f read=%9.1f MB write=%9.1f MB" % (k, v.get("avg_us", float("nan")), rd / 1e6, wr / 1e6))
    res["layers"] = layers
    for k, v in res["kernels"].items():
        if "FETCH_SIZE_KiB_per_launch" in v:
            rd = 2.0 * v["FETCH_SIZE_KiB_per_launch"] * 1024.0
            wr = v.get("WRITE_SIZE_KiB_per_launch", 0.0) * 1024.0
            v["hbm_bytes_per_launch"] = rd + wr
            v["hbm_read_bytes_per_launch"], v["hbm_write_bytes_per_launch"] = rd, wr
    lines.append("== corrected HBM bytes per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 rule)")
    for k, v in res["kernels"].items():
        if "hbm_bytes_per_launch" in v:
            lines.append("%-28s read=%.1f MB write=%.1f MB" % (k, v["hbm_read_bytes_per_launch"] / 1e6,
                                                              v["hbm_write_bytes_per_launch"] / 1e6))
    fused = [k for k in res["kernels"] if k.startswith(("csr_fused_scores", "csr_fused_hop", "csr_rank"))]
    if fused and all("hbm_bytes_per_launch" in res["kernels"][k] for k in fused):
        tb = sum(res["kernels"][k]["hbm_bytes_per_launch"] for k in fused)
        tu = sum(res["kernels"][k].get("median_us", 0.0) for k in fused)
        lines.append("== bf16-storage graph LAYER (config 5; degree ranking + score kernel + hop / tap kernel): %.1f MB of HBM traffic per step, "
                     "%.1f us (traced medians)" % (tb / 1e6, tu))
        res["csr_fused_layer"] = {"hbm_bytes_per_step": tb, "traced_us": tu}
    with open(os.path.join(out, "summary_%s.txt" % tag), "w") as f:
        f.write("\n".join(lines) + "\n")
    import datetime
    import socket
    # provenance of the PMC bytes bench.py quotes as `roofline.traffic` (they are read from THIS file on other boxes)
    res["collected"] = "rocprofv3 --pmc passes of tools/profile_round.sh on gpurun box %s, %s UTC" % (
        socket.gethostname(), datetime.datetime.utcnow().strftime("%Y-%m-%d %H:%M"))
    with open(os.path.join(out, "summary_%s.json" % tag), "w") as f:
        json.dump(res, f, indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
