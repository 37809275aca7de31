set -u
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02l > gpurun_out/prof_r02l.log 2>&1
O=gpurun_out/prof_r02l
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
for W in c1 c2 c5; do python bench.py --workload $W > $O/bench_$W.json 2> $O/bench_$W.err; done
python bench.py --workload c5f32 > $O/bench_c5f32.json 2> $O/bench_c5f32.err
python tools/closed_loop_bench.py > $O/closed_loop.txt 2>&1
python tools/latency_probe.py > $O/latency_probe.txt 2>&1
MAGAT_LIB_PATH=magat_pathplanning_amd/lib/libmagat_hip_debug.so python tools/gat_mfma_probe.py > $O/gat_mfma_phase_cycles.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu tools/exp/mfma_valu.hip 2>/dev/null && /tmp/mfma_valu > $O/mfma_valu.txt 2>&1
tail -3 gpurun_out/prof_r02l.log; for f in $O/bench_*.json; do python -c "
import sys,json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'))"; done
