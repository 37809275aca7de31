#!/bin/bash
# per-kernel durations of the batch-1 step (N = 10 and N = 100): tools/lat_trace.sh
R=$PWD; cd /tmp; export TMPDIR=/tmp
for N in 10 100; do
  rm -rf /tmp/lt$N
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lt$N -o t -- python $R/tools/lat_trace.py $N > /tmp/lt$N.log 2>&1
  echo "== batch 1, N = $N: kernels of a step (300 steps + warm-up; avg us, calls)"
  python - /tmp/lt$N <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/t_kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
tot = 0.0
for r in rows:
    c = int(r["Calls"])
    if c >= 250:
        per = float(r["TotalDurationNs"]) / 300.0 / 1e3
        tot += per
        print("   %-74s calls %5d avg %7.2f us  per step %7.2f us" % (r["Name"][:74], c, float(r["AverageNs"]) / 1e3, per))
print("   sum per step: %.1f us" % tot)
PY
done
