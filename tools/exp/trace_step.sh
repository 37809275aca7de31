# per-launch durations of ONE step of a bench workload, in launch order: tools/exp/trace_step.sh <workload>
W=${1:-c5}; R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$W -o t -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-extra-legs > /dev/null 2>&1
cd $R
python - <<P
import csv
rows=list(csv.DictReader(open('gpurun_out/trace_$W/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'stem8' in r['Kernel_Name']]
# the last step: from the stem launch that follows the last skinny/actions kernel
last=[i for i in idx]
a=last[-2] if len(last)>=2 and last[-1]-last[-2] < 40 else last[-1]
t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%9.1f +%8.1f  %s'%((s-t0)/1e3,(e-s)/1e3,r['Kernel_Name'][:90]))
P
