#!/bin/bash
# SQ counter passes for the GAT graph kernels (instrumentation).  Usage: tools/pmc_gat.sh <tag> [B N]
R=$PWD; OUT=$R/gpurun_out/pmc_gat_${1:-x}; B=${2:-512}; N=${3:-100}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for P in "$P1" "$P2"; do i=$((i+1)); rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o g -- python $R/tools/gat_only.py $B $N 2 > $OUT/p$i.log 2>&1; done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        for key in ("gat_dense", "gat_list", "gat_struct"):
            if key in r["Kernel_Name"]:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key in acc:
        for k, v in acc[key].items():
            print("%-12s %-30s %16.0f  (n=%d)" % (key, k, sum(v) / len(v), len(v)))
PY
