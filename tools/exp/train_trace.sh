# kernel-time breakdown of a training step with the HIP convolution backend: tools/exp/train_trace.sh <B> <N>
B=${1:-64}; N=${2:-10}; R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
cat > /tmp/tt.py <<P
import os, sys
sys.path.insert(0, "$R")
os.environ["MAGAT_TRAIN_CNN"] = "hip"
import torch, torch.nn.functional as tnf
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
dev = torch.device("cuda:0")
B, N = $B, $N
cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device="cuda:0")
x = fov_states(B, N, seed=5).to(dev); S = comm_gso(B, N, 20 if N <= 20 else 50, seed=6).to(dev); tgt = torch.randint(0, 5, (B * N,)).to(dev)
net = DecentralPlannerGATNet(cfg).to(dev).train(); opt = torch.optim.SGD(net.parameters(), lr=0.01)
for _ in range(12):
    net.addGSO(S); loss = tnf.cross_entropy(net(x), tgt); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
P
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ttout -o t -- python /tmp/tt.py > /dev/null 2>&1
python - <<P
import csv
rows=list(csv.DictReader(open("/tmp/ttout/t_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step: %.1f us over 12 steps" % (tot/12e3))
for r in rows[:25]:
    print("%-80s calls/step %6.1f avg_us %8.1f us/step %8.1f" % (r["Name"][:80], int(r["Calls"])/12, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/12e3))
P
