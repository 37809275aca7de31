#!/bin/bash
# same-box A/B of environment settings: tools/ab_env.sh "A=1 B=2" "A=0" ...  (each argument = one setting; WORKLOADS="c3 c2" picks
# the bench workloads, default the headline one)
for rep in 1 2; do
  for w in ${WORKLOADS:-c3}; do
  for e in "$@"; do
    env $e python bench.py --workload $w --no-cpu-baseline --no-extra-legs --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
g=lambda n: k.get(n,{}).get('avg_us',0)
print('%-16s %-28s step %.4f ms | chain %.1f | stem %.1f | head %.1f | comp %.1f | gat %.1f | guard/step %.1f' % ('$w', '$e', d['ms_per_step'], g('layer1.conv2+layer2+layer3 (fused, pooled)'), g('conv_first+layer1.conv1 (fused)'), g('head(avgpool+fc+linear)'), g('compressMLP'), g('gat_layer (one launch)'), 1e3*k.get('range_guard',{}).get('ms_per_step',0)))"
  done
  done
done
