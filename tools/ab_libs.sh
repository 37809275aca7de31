#!/bin/bash
# same-box A/B of library builds on the headline workload: tools/ab_libs.sh "<env assignments>" name1 name2 ...  (name "" = the release lib)
ENVS=$1; shift
for rep in 1 2; do
  for n in "$@"; do
    L=$PWD/magat_pathplanning_amd/lib/libmagat_hip${n:+_$n}.so
    env $ENVS MAGAT_LIB_PATH=$L python bench.py --no-cpu-baseline --no-extra-legs --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('%-10s step %.4f ms | chain %.1f us | stem %.1f | gat %.1f | head %.1f | comp %.1f | actions %.1f' % ('${n:-release}', d['ms_per_step'], k['layer1.conv2+layer2+layer3 (fused, pooled)']['avg_us'], k['conv_first+layer1.conv1 (fused)']['avg_us'], k.get('gat_layer (one launch)',{}).get('avg_us',0), ([v for n_, v in k.items() if n_.startswith('head')] or [{}])[0].get('avg_us', 0), k.get('compressMLP',{}).get('avg_us',0), k.get('actionsMLP',{}).get('avg_us',0)))"
  done
done
