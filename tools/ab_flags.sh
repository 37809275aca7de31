#!/bin/bash
# Same-box A/B of compile-time variants of one source file: for every flag set (| separated in FLAG_LIST, the empty set first
# and last), rebuild, run the model parity tests' quick subset and print the bench step time + the kernel table line(s) asked for.
#   SRC=block_fused.hip KERNELS="layer1.conv2+layer2" FLAG_LIST="-DX" bash tools/ab_flags.sh
cd $(dirname $0)/..
SRC=${SRC:-block_fused.hip}
IFS="|" read -ra VARS <<< "|${FLAG_LIST}|x"; unset "VARS[${#VARS[@]}-1]"; VARS+=("")
for F in "${VARS[@]}"; do
  bash tools/build_variant.sh ab $SRC "$F" > /dev/null 2>&1 || { echo "build failed: $F"; continue; }
  MAGAT_LIB_PATH=magat_pathplanning_amd/lib/libmagat_hip_ab.so python bench.py --no-cpu-baseline --no-extra-legs --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
ks = [k for k in d['kernels'] if any(s in k for s in '''${KERNELS:-layer}'''.split('|'))]
print('flags [%s]: %.4f ms/step  ' % ('''$F''', d['ms_per_step']) + '  '.join('%s %.1f us' % (k, d['kernels'][k]['avg_us']) for k in ks))"
done
rm -f magat_pathplanning_amd/lib/libmagat_hip_ab.so
