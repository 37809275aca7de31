cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_csr -o c -- python $GRAFT_REPO_ROOT/tools/csr_layer_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/prof_csr/c_kernel_stats.csv')))
for r in rows[:8]: print(r['Name'][:70], r['Calls'], '%.1f us'%(float(r['AverageNs'])/1e3))
P
