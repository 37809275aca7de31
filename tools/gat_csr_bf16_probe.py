"""GAT layer alone at BASELINE config 5's shape: CSR kernels with fp32 vs bf16 storage (per-tag HIP-event times)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, _native as nat
from magat_pathplanning_amd.graphml import gat_forward_rows_csr, dense_gso_to_csr
from magat_pathplanning_amd.synthetic import comm_gso
B, N, K, P, G = (int(a) for a in (sys.argv[1:6] + ["128", "1000", "2", "4", "128"][len(sys.argv) - 1:]))
dev = torch.device("cuda:0")
layer = GraphFilterBatchAttentional(G, G, K, P, attentionMode="KeyQuery").to(dev).eval()
X = torch.randn(B, N, G, device=dev) * 0.5
S = comm_gso(B, N, 160 if N >= 1000 else 50, seed=1).to(dev)
rowptr, colidx, nnz = dense_gso_to_csr(S)
del S
deg = nnz / (B * N)
print("B %d N %d K %d P %d G %d  mean degree %.2f" % (B, N, K, P, G, deg))
lib = nat.lib()
X16 = X.to(torch.bfloat16)


def run(x, reps=10):
    for _ in range(3):
        gat_forward_rows_csr(x, rowptr, colidx, nnz, layer)
    lib.magat_profile_reset(); lib.magat_profile_enable(1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gat_forward_rows_csr(x, rowptr, colidx, nnz, layer)
    e1.record()
    torch.cuda.synchronize(); lib.magat_profile_enable(0); lib.magat_profile_collect()
    tot = e0.elapsed_time(e1) / reps * 1e3
    es = 2 if x.dtype == torch.bfloat16 else 4
    alg = B * N * (es * G + es * P * G + 4 * (1 + deg))           # X + Y + CSR per agent-step
    print("   layer %.1f us  -> %.2f M agent-steps/s, algorithmic %.1f MB -> %.0f GB/s (%.1f %% of 8 TB/s)"
          % (tot, B * N / tot, alg / 1e6, alg / tot / 1e3, alg / tot / 1e3 / 80))
    for tag in (10, 11, 14):
        c, t = ctypes.c_longlong(0), ctypes.c_double(0)
        lib.magat_profile_read(tag, ctypes.byref(c), ctypes.byref(t))
        if c.value:
            print("   tag %-10s launches/step %d  us/step %.1f" % (nat.TAGS[tag], c.value // reps, t.value * 1e3 / reps))


with torch.no_grad():
    print("fp32 storage"); run(X)
    print("bf16 storage"); run(X16)
    y32, _ = gat_forward_rows_csr(X, rowptr, colidx, nnz, layer)
    y16, _ = gat_forward_rows_csr(X16, rowptr, colidx, nnz, layer)
    d = (y16.float() - y32).abs()
    print("bf16 vs fp32 output: max|d| %.3e  mean|d| %.3e  scale %.3f" % (d.max().item(), d.mean().item(), y32.abs().max().item()))
