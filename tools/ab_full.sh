#!/bin/bash
# A/B of BLOCK_FULL 1 vs 2 (same library) and the base library
for rep in 1 2 3; do
  for v in 1 2; do
    MAGAT_BLOCK_FULL=$v python bench.py --no-cpu-baseline --no-extra-legs --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']['layer1.conv2+layer2+layer3 (fused, pooled)']
print('BLOCK_FULL=$v', d['ms_per_step'], 'chain us', k['avg_us'])"
  done
done
MAGAT_LIB_PATH=$PWD/magat_pathplanning_amd/lib/libmagat_hip_base.so python bench.py --no-cpu-baseline --no-extra-legs --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']['layer1.conv2+layer2+layer3 (fused, pooled)']
print('base lib', d['ms_per_step'], 'chain us', k['avg_us'])"
