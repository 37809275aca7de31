"""Device time of the simulator front / back end pieces at batch 1 (the closed loop around the batch-1 forward)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch
from magat_pathplanning_amd.simulator import batched_fov_states, batched_gso, batched_move
dev = torch.device("cuda:0")
for (B, N, size) in ((1, 10, 20), (1, 100, 50), (8, 100, 50)):
    rng = np.random.default_rng(3)
    m = (rng.random((size, size)) < 0.08).astype(np.uint8)
    free = np.argwhere(m == 0)
    pos = np.stack([free[rng.permutation(len(free))[:N]] for _ in range(B)]).astype(np.int32)
    goal = np.stack([free[rng.permutation(len(free))[:N]] for _ in range(B)]).astype(np.int32)
    dm, dpos, dgoal = torch.from_numpy(m).to(dev), torch.from_numpy(pos).to(dev).contiguous(), torch.from_numpy(goal).to(dev)
    logits = torch.randn(B * N, 5, device=dev)
    big = torch.randn(4096, 4096, device=dev)
    def dev_time(fn, n=200):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        for _ in range(4): big @ big
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): fn()
        host = (time.perf_counter() - t0) / n * 1e6
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n, host
    p2 = dpos.clone()
    print("B=%d N=%d: gso(normalize) %.1f us (host %.1f) | gso(0/1) %.1f (%.1f) | fov_states %.1f (%.1f) | move %.1f (%.1f)" % (
        (B, N) + dev_time(lambda: batched_gso(dpos, 7.0)) + dev_time(lambda: batched_gso(dpos, 7.0, normalize=False)) +
        dev_time(lambda: batched_fov_states(dm, dpos, dgoal, 9)) + dev_time(lambda: batched_move(dm, p2, logits=logits, goal=dgoal))))
