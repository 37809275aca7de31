cd /tmp && export TMPDIR=/tmp
cat > /tmp/b1.py <<'P'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
B, N, mw = 1, 100, 50
cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
net = DecentralPlannerGATNet(cfg).to(dev).eval()
x, S = fov_states(B, N).to(dev), comm_gso(B, N, mw, dtype=torch.float64).to(dev)
with torch.no_grad():
    for _ in range(30):
        net.addGSO(S); y = net(x); y.cpu()
P
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_b1 -o b1 -- python /tmp/b1.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/tl_b1/b1_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find the last layer1_fused kernel
idx=[i for i,r in enumerate(rows) if 'stem8' in r['Kernel_Name'] or 'layer1_fused' in r['Kernel_Name']]
a=idx[-1]-3; t0=int(rows[a]['Start_Timestamp'])
for r in rows[a:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print('%8.1f +%7.1f  %s'%((s-t0)/1e3,(e-s)/1e3,r['Kernel_Name'][:80]))
P
