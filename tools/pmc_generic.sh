#!/bin/bash
# Usage: tools/pmc_generic.sh <tag> <kernel-substr> "<cmd>" "<ctr set 1>" ["<ctr set 2>" ...]
R=$PWD; TAG=$1; KSUB=$2; CMD=$3; shift 3; OUT=$R/gpurun_out/pmcg_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
i=0
for P in "$@"; do i=$((i+1)); [ -d $OUT/p$i ] || rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o g -- $CMD > $OUT/p$i.log 2>&1; done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-34s avg %16.0f  last %16.0f (n=%d)" % (k, sum(v) / len(v), v[-1], len(v)))
PY
