"""Random episodes through the on-device simulator (BatchedEpisode: FOV states, GSO, action policies, collision shielding,
bookkeeping) against oracle/sim_oracle.py, EVERY instance, every step (test infrastructure).  python tools/exp/fuzz_sim.py [count] [seed]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from oracle import sim_oracle as so
from magat_pathplanning_amd.simulator import BatchedEpisode, POLICIES
dev = torch.device("cuda:0")
count = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bad = 0
for it in range(count):
    N = int(rng.choice([1, 2, 3, 8, 10, 20, 33, 64, 100]))
    size = int(rng.choice([6, 10, 20, 28, 50]))
    while size * size < 3 * N:
        size += 4
    B = int(rng.choice([1, 2, 5]))
    T = int(rng.choice([3, 8, 15]))
    maxstep = int(rng.choice([4, 10, 40]))
    dens = float(rng.choice([0.0, 0.05, 0.15]))
    pol_name = str(rng.choice(sorted(POLICIES)))
    radius = float(rng.choice([2.0, 5.0, 7.0]))
    tag = "B=%d N=%d map=%d T=%d maxstep=%d dens=%.2f %s r=%.0f" % (B, N, size, T, maxstep, dens, pol_name, radius)
    print("try ", tag, flush=True)
    try:
        m = (rng.random((size, size)) < dens).astype(np.uint8)
        free = np.argwhere(m == 0)
        pos = np.stack([free[rng.permutation(len(free))[:N]] for _ in range(B)]).astype(np.int32)
        goal = np.stack([free[rng.permutation(len(free))[:N]] for _ in range(B)]).astype(np.int32)
        logits = rng.normal(size=(T, B, N, 5)).astype(np.float32) * 2
        if pol_name == "sum_multinorm":      # (logits / sum(logits) is a distribution for positive logits only: the reference's
            logits = np.abs(logits) + 1e-3    #  torch.multinomial refuses anything else, the device flags it - test_gpu_sim.py)
        uni = rng.random((T, B, N))
        ep = BatchedEpisode(dv(m), dv(pos), dv(goal), maxstep, comm_radius=radius, action_select=pol_name)
        ok = True
        ep.gso()                  # (step 0: fixes the instance's radius, as the reference's first getGSO does)
        x0 = ep.states().cpu().numpy()
        for b in range(B):
            ok &= np.array_equal(x0[b], so.fov_states(m, pos[b], goal[b]))
        hist = []
        for t in range(T):
            ep.step(logits=dv(logits[t]), uniforms=dv(uni[t]))
            hist.append((ep.pos.cpu().numpy().copy(), ep.actions.cpu().numpy().copy(), ep.flags.cpu().numpy().copy()))
        for b in range(B):
            st = so.EpisodeState(m, pos[b], goal[b], maxstep)
            for t in range(T):
                _, pc, keys = so.episode_step(st, logits[t, b], t, POLICIES[pol_name], uni[t, b])
                ok &= np.array_equal(hist[t][0][b], st.pos)
                if keys is not None:
                    ok &= np.array_equal(hist[t][1][b], keys)
                ok &= bool(hist[t][2][b] & 15) == pc
            ok &= np.array_equal(ep.reach_goal.cpu().numpy()[b], st.reach_goal)
            ok &= np.array_equal(ep.end_step.cpu().numpy()[b], st.end_step)
            ok &= int(ep.flowtime[b]) == st.flowtime and int(ep.makespan[b]) == st.makespan
            W = so.gso_from_positions(st.pos, float(ep.radii.cpu().numpy()[b]))
            ok &= np.allclose(ep.gso().cpu().numpy()[b], W, atol=1e-12)
        bad += 0 if ok else 1
        print("%s %s" % ("ok  " if ok else "BAD ", tag), flush=True)
    except Exception as e:
        bad += 1
        print("RAISE %s -> %s" % (tag, repr(e)[:200]), flush=True)
print("failures:", bad, "of", count)
