#!/bin/bash
# Per-kernel times (rocprofv3 kernel trace) of the bf16-storage graph layer at config 5's shape for a list of library builds:
#   tools/csr_fused_ab.sh "" coal ...      ("" = the release library; others: lib/libmagat_hip_<name>.so, tools/build_variant.sh)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  L=$R/magat_pathplanning_amd/lib/libmagat_hip${n:+_$n}.so
  D=/tmp/csrab_${n:-release}
  rm -rf $D
  MAGAT_ALLOW_EXPERIMENT_BUILD=1 MAGAT_LIB_PATH=$L rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python $R/tools/csr_layer_bench.py > $D.log 2>&1
  echo "== ${n:-release}"
  tail -3 $D.log
  python - $D <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("csr_", "gso_", "conv_gemm", "cast_")):
        print("   %-70s calls %4s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
