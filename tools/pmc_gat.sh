#!/bin/bash
# SQ counter passes for gat_dense_kernel (instrumentation).  Usage: tools/pmc_gat.sh <tag>
R=$PWD; OUT=$R/gpurun_out/pmc_gat_${1:-x}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_WAVE32_LDS"
i=0
for P in "$P1" "$P2"; do i=$((i+1)); rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/p$i -o g -- python $R/tools/gat_only.py 512 100 2 > $OUT/p$i.log 2>&1; done
cd $R
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gat_dense" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-34s %16.0f  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
