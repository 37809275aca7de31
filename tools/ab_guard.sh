for rep in 1 2; do for n in "" oldguard; do
L=$PWD/magat_pathplanning_amd/lib/libmagat_hip${n:+_$n}.so
MAGAT_LIB_PATH=$L python bench.py --no-cpu-baseline --no-extra-legs --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; g=k['range_guard']
print('%-10s step %.4f ms guard avg %.2f us launches %s per-step %s' % ('${n:-release}', d['ms_per_step'], g['avg_us'], g.get('launches'), g.get('ms_per_step')))"
done; done
