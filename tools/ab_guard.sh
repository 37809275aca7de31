# same-box A/B of library builds on the guard's cost: tools/ab_guard.sh name1 name2 ...  ("" = the release lib); step time without timing hooks
for rep in 1 2; do for n in "$@"; do
L=$PWD/magat_pathplanning_amd/lib/libmagat_hip${n:+_$n}.so
MAGAT_LIB_PATH=$L python bench.py --no-cpu-baseline --no-extra-legs --no-kernel-timing --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s step %.4f ms' % ('${n:-release}', d['ms_per_step']))"
done; done
