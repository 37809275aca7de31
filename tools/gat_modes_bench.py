"""GAT layer (maps GEMM + graph kernel) per attention mode at a benchmark shape: per-tag HIP-event times."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from magat_pathplanning_amd import GraphFilterBatchAttentional, GraphFilterBatchAttentional_Origin, _native as nat
from magat_pathplanning_amd.graphml import gat_forward_rows
from magat_pathplanning_amd.synthetic import comm_gso
B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 100)
dev = torch.device("cuda:0")
lib = nat.lib()
X = torch.randn(B, N, 128, device=dev)
S = comm_gso(B, N, {100: 50, 20: 28, 10: 20}.get(N, 50), seed=1).to(dev)
for mode in ("KeyQuery", "GAT_modified", "GAT_origin"):
    for concat in (True, False):
        cls = GraphFilterBatchAttentional_Origin if mode == "GAT_origin" else GraphFilterBatchAttentional
        layer = cls(128, 128, 3, 4, concatenate=concat, attentionMode=mode).to(dev).eval()
        with torch.no_grad():
            for _ in range(3):
                gat_forward_rows(X, S, layer)
            lib.magat_profile_reserve(256); lib.magat_profile_reset(); lib.magat_profile_enable(1)
            for _ in range(10):
                gat_forward_rows(X, S, layer)
            torch.cuda.synchronize(); lib.magat_profile_enable(0); lib.magat_profile_collect()
        out = []
        for tag in (10, 11, 13, 19):
            c, t = ctypes.c_longlong(0), ctypes.c_double(0)
            lib.magat_profile_read(tag, ctypes.byref(c), ctypes.byref(t))
            if c.value:
                out.append("%s %.1f us" % (nat.TAGS[tag], t.value * 1e3 / 10))
        print("%-13s %-6s %s" % (mode, "concat" if concat else "mean", "   ".join(out)))
