# Round-end evidence from ONE box: full GPU suite, rocprofv3 trace + PMC passes, every bench leg, latency.  tools/exp/collect_round.sh <tag>
set -u
TAG=${1:-r00}
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$TAG
mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/pytest_full.txt
bash tools/profile_round.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
for W in c2 c5; do python bench.py --workload $W --no-cpu-baseline > $O/bench_$W.json 2> $O/bench_$W.err; done
python tools/latency_probe.py 2>&1 | grep "^B=" > $O/latency.txt
bash tools/lat_trace.sh > $O/lat_trace.txt 2>&1
python tools/exp/lat_pub_probe.py 2>&1 | grep "^N=" > $O/latency_published_shape.txt
tail -3 gpurun_out/prof_$TAG.log; cat $O/pytest_full.txt; cat $O/latency.txt
for f in $O/bench_*.json; do python -c "
import sys,json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('traffic'))"; done
