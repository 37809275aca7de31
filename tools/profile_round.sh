#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run via gpurun from the repo root):
#   kernel-trace + stats of the default bench command, then separate PMC passes (FETCH_SIZE, WRITE_SIZE)
#   as MI355X_MICROARCH.md prescribes (TCC slots do not fit both; never combined with other trace domains).
# Usage: tools/profile_round.sh r01 [workload: c3 (default) | c2 | c5]
set -u
TAG=${1:-r00}
WL=${2:-c3}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
if [ "$WL" != c3 ]; then OUT=$R/gpurun_out/prof_${TAG}_$WL; TAG=${TAG}_$WL; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-extra-legs"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o c3 -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o c3 -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-legs > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o c3 -- python $R/bench.py --workload $WL --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-legs > $OUT/pmc_write.log 2>&1
cd $R
# the same command once more WITHOUT the tracer (its hipEvent table goes into the summary beside the traced durations)
python $R/bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $OUT/bench_untraced.json 2>/dev/null
python tools/rocprof_summary.py $OUT $TAG
