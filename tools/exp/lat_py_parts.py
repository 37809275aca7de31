import os, sys, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from magat_pathplanning_amd import DecentralPlannerGATNet, _native as nat
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config
import magat_pathplanning_amd.graphml as gm
N = 10
dev = torch.device("cuda:0")
cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat")
net = DecentralPlannerGATNet(cfg).to(dev).eval()
x, S = fov_states(1, N).to(dev), comm_gso(1, N, 20, dtype=torch.float64).to(dev)
def bench(name, fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print("%-44s %7.2f us" % (name, dt))
with torch.no_grad():
    for _ in range(20):
        net.addGSO(S); net(x).cpu()
    torch.cuda.synchronize()
    bench("_weights_key", lambda: net._weights_key(dev))
    bench("_refresh", lambda: net._refresh(dev))
    bench("torch.cuda.device ctx", lambda: torch.cuda.device(dev).__enter__())
    def ctx():
        with torch.cuda.device(dev): pass
    bench("with torch.cuda.device", ctx)
    bench("nat.current_stream", lambda: nat.current_stream(dev))
    bench("torch.empty(10,5)", lambda: torch.empty(10, 5, dtype=torch.float32, device=dev))
    bench("x.reshape.to", lambda: x.reshape(10, 3, 11, 11).to(dev))
    bench("x.contiguous().float()", lambda: x.contiguous().float())
    bench("torch.device(str)", lambda: torch.device(net.config.device))
    layer = net.GFL[0]
    bench("layer.bias chain", lambda: layer.bias.detach().to(dev, torch.float32).reshape(-1).contiguous())
    bench("ConvGemmDesc()", lambda: nat.ConvGemmDesc())
    bench("nat.ptr(x)", lambda: nat.ptr(x))
    bench("is_grad_enabled", lambda: torch.is_grad_enabled())
    bench("addGSO", lambda: net.addGSO(S), 500)
    rt = net._refresh(dev)
    st = nat.current_stream(dev)
    bench("_run_encoder (enqueue)", lambda: net._run_encoder(rt, x.reshape(10,3,11,11), 10, dev, st), 300)
    comp = net._buf("comp", (10, 128), dev); gat = net._buf("gat", (10, net.gat_width), dev); feat = net._buf("feat", (10, 128), dev)
    layer.addGSO(net.S)
    bench("gat_forward_rows (enqueue)", lambda: gm.gat_forward_rows(comp.view(1, 10, 128), net.S, layer, out=gat, csr=rt.csr), 300)
    bench("_run_actions (enqueue)", lambda: net._run_actions(rt, feat, comp, gat, gat, 10, dev, st), 300)
    bench("layer.addGSO", lambda: layer.addGSO(net.S))
    bench("forward (enqueue)", lambda: net(x), 300)
    bench("packed_weights", lambda: gm._packed_weights(layer, dev, st, 128, 128, 3, 4, 0))
