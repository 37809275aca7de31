"""Training-step time of DecentralPlannerGATNet (forward + cross-entropy + backward + SGD step, train mode) with the HIP
convolution backend (train_cnn.py: magat_conv_gemm_f32 forward / input gradient, magat_conv_wgrad_f32 weight gradient) and with
torch's own convolutions (MAGAT_TRAIN_CNN=torch: MIOpen); the graph layer trains on the HIP kernels either way.
   python tools/train_step_bench.py [B N] ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as tnf
from magat_pathplanning_amd import DecentralPlannerGATNet
from magat_pathplanning_amd.synthetic import comm_gso, fov_states, make_config

dev = torch.device("cuda:0")
shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or [(64, 10), (64, 100)]
for B, N in shapes:
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat", device="cuda:0")
    x = fov_states(B, N, seed=5).to(dev)
    S = comm_gso(B, N, 20 if N <= 20 else 50, seed=6).to(dev)
    tgt = torch.randint(0, 5, (B * N,)).to(dev)
    for backend in ("hip", "torch", "hip", "torch"):
        os.environ["MAGAT_TRAIN_CNN"] = backend
        torch.manual_seed(1)
        net = DecentralPlannerGATNet(cfg).to(dev).train()
        opt = torch.optim.SGD(net.parameters(), lr=0.01)

        def step():
            net.addGSO(S)
            loss = tnf.cross_entropy(net(x), tgt)
            opt.zero_grad()
            loss.backward()
            opt.step()
            return loss
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("B=%d N=%d (%d agents)  convolutions on %-5s  %.2f ms / training step  (%.0f agent-steps/s), loss %.4f" % (
            B, N, B * N, backend, dt * 1e3, B * N / dt, float(loss)))
