#!/bin/bash
# Arbitrary SQ/TCC counter pass over bench.py (c3), folded per layer by dispatch order.
# Usage: tools/pmc_bench.sh <tag> "<CTR1 CTR2 ...>" ["<second pass>"...]
R=$PWD; TAG=$1; shift; OUT=$R/gpurun_out/pmcb_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for P in "$@"; do i=$((i+1)); rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/p$i.log 2>&1; done
cd $R
python - <<PY
import csv, glob, collections
SEQ = ["conv_first", "l1.conv1", "l1.conv2+ds", "l2.conv1", "l2.conv2+ds", "l3.conv1", "l3.conv2+ds", "head", "compress", "gat_maps", "gat_graph", "actions"]
ours = ("conv_gemm_kernel", "conv_gemm_bf16x6_kernel", "conv_first_kernel", "gat_dense_kernel")
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if any(o in r["Kernel_Name"] for o in ours)]
    ctrs = sorted(set(r["Counter_Name"] for r in rows))
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    byd = collections.defaultdict(dict)
    for r in rows:
        byd[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for k, did in enumerate(sorted(byd)):
        for c, v in byd[did].items():
            per[SEQ[k % len(SEQ)]][c].append(v)
    print("%-14s " % "layer" + " ".join("%18s" % c[-18:] for c in ctrs))
    for name in SEQ:
        print("%-14s " % name + " ".join("%18.0f" % (sum(per[name][c]) / max(1, len(per[name][c]))) for c in ctrs))
PY
