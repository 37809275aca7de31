"""Times the batched on-device simulator front-end (GSO construction + FOV state tensors) at a benchmark shape, next to
the per-instance numpy restatement of the reference's host loops (oracle/sim_oracle.py) on a few instances."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from magat_pathplanning_amd.simulator import batched_fov_states, batched_gso
from oracle import sim_oracle as so

B, N, size = (int(a) for a in (sys.argv[1:4] + ["512", "100", "50"][len(sys.argv) - 1:]))
rng = np.random.default_rng(1)
m = (rng.random((size, size)) < 0.08).astype(np.uint8)
free = np.argwhere(m == 0)
pos = np.stack([free[rng.permutation(len(free))[:N]] for _ in range(B)]).astype(np.int32)
goal = np.stack([free[rng.permutation(len(free))[:N]] for _ in range(B)]).astype(np.int32)
dev = torch.device("cuda:0")
dm, dp, dg = torch.from_numpy(m).to(dev), torch.from_numpy(pos).to(dev), torch.from_numpy(goal).to(dev)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t_gso = timed(lambda: batched_gso(dp, 7.0))
t_mask = timed(lambda: batched_gso(dp, 7.0, normalize=False, dtype=torch.float32))
t_fov = timed(lambda: batched_fov_states(dm, dp, dg, 9))
k = min(B, 8)
t0 = time.perf_counter()
for b in range(k):
    so.gso_from_positions(pos[b], 7.0)
c_gso = (time.perf_counter() - t0) / k * 1e3
t0 = time.perf_counter()
for b in range(k):
    so.fov_states(m, pos[b], goal[b], 9)
c_fov = (time.perf_counter() - t0) / k * 1e3
print("B %d N %d map %dx%d" % (B, N, size, size))
print("  GSO  (W, lambda_max, S f64): %.3f ms / batch  (0/1 adjacency only: %.3f ms)   host restatement %.2f ms / instance -> %.0f ms / batch"
      % (t_gso, t_mask, c_gso, c_gso * B))
print("  FOV state tensors          : %.3f ms / batch   host restatement %.2f ms / instance -> %.0f ms / batch"
      % (t_fov, c_fov, c_fov * B))
