#!/bin/bash
# A/B of one library option on the headline workload, alternating runs in one box session:
#   tools/ab_option.sh MAGAT_RANGE_GUARD 0 1 [extra bench args]   -> gpurun_out/ab_<name>.txt
NAME=$1; A=$2; B=$3; shift 3
OUT=gpurun_out/ab_${NAME}.txt
: > $OUT
for rep in 1 2 3; do
  for v in $A $B; do
    env $NAME=$v python bench.py --no-cpu-baseline --no-extra-legs --steps 40 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$NAME=$v', d['value'], d['ms_per_step'], d.get('kernel_time_ms_per_step'), {k:v['ms_per_step'] for k,v in d.get('kernels',{}).items() if k in ('range_guard',)})" >> $OUT
  done
done
cat $OUT
