#!/bin/bash
# rocprofv3 kernel stats of one bench command: tools/kstats.sh <tag> <bench args...> -> gpurun_out/kstats_<tag>.txt
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/kstats_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $R/bench.py --no-cpu-baseline --no-kernel-timing --no-extra-legs "$@" > $OUT/log.txt 2>&1
cd $R
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python - "$F" <<'PY' > gpurun_out/kstats_$TAG.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:28]:
    print("%-90s calls=%6s avg_us=%10.2f pct=%6s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cat gpurun_out/kstats_$TAG.txt
