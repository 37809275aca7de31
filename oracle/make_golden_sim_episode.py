"""Generates tests/golden/simepisode_*.npz and simradius_*.npz with the REAL reference simulator (build container only):
multiRobotSimNew.move (utils/new_simulator.py:471-549) stepped over whole episodes with each of the three action policies
(:863-883), and the step-0 branch of computeAdjacencyMatrix (:759-781: communication radius grown by 10 % until the graph
is connected).  TEST INFRASTRUCTURE.      python oracle/make_golden_sim_episode.py

Two sources of randomness in the reference are pinned so that the run is a deterministic target:
  * random.choice (cell conflicts, :416) := first element (the documented rule of the device kernel: lowest index wins);
  * torch.multinomial(w, 1) := inverse CDF with a recorded uniform u:  first k with  w_0 + .. + w_k > u * sum(w)
    (float64 running sum over the float32 weights the reference passes in).  The uniforms are part of the fixture.
Fixtures hold data only: maps, coordinates, per-step logits and uniforms, and the state the reference had after each step."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.make_golden_sim import load_reference_frontend, scenario  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
POLICY = {"soft_max": 0, "sum_multinorm": 1, "exp_multinorm": 2}


class Uniforms:
    def __init__(self):
        self.queue = []

    def multinomial(self, w, n, *a, **k):
        assert n == 1
        u = self.queue.pop(0)
        c = np.cumsum(w.detach().reshape(-1).numpy().astype(np.float64))
        hit = np.nonzero(c > u * c[-1])[0]
        return torch.tensor([[int(hit[0])]])


def make_sim(Sim, N, m, pos, goal, maxstep, policy):
    cfg = types.SimpleNamespace(num_agents=N, batch_numAgent=False)
    s = types.SimpleNamespace(config=cfg, size_map=m.shape, maxstep=maxstep,
                              up=np.array([-1, 0]), down=np.array([1, 0]), left=np.array([0, -1]), right=np.array([0, 1]),
                              stop=np.array([0, 0]), up_keyValue=0, down_keyValue=2, left_keyValue=1, right_keyValue=3,
                              stop_keyValue=4, wall_dict={tuple(p): i for i, p in enumerate(np.argwhere(m != 0))},
                              current_positions=pos.astype(np.float64).copy(), goal_positions=goal.astype(np.float64).copy(),
                              reach_goal=np.zeros(N), first_move=np.zeros(N), end_step=np.zeros(N), path_list=[],
                              fun_Softmax=torch.nn.LogSoftmax(dim=-1), makespanPredict=maxstep, flowtimePredict=maxstep * N)
    s.normalize = lambda x: Sim.normalize(s, x)
    s.check_collision = lambda p, mv: Sim.check_collision(s, p, mv)
    fn = {0: Sim.convectToActionKey_softmax, 1: Sim.convectToActionKey_sum_multinorm, 2: Sim.convectToActionKey_exp_multinorm}[policy]
    s.convectToActionKey = lambda v: fn(s, v)
    return s


def step_logits(rng, pos, goal, policy, sharp):
    """Synthetic policy output: noise plus a push towards the goal (so that episodes do end), as probabilities for the
    sum policy (its weights must be non-negative), raw scores otherwise."""
    N = len(pos)
    z = rng.normal(size=(N, 5)).astype(np.float32)
    d = goal - pos
    for i in range(N):
        if d[i, 0] < 0: z[i, 0] += sharp
        if d[i, 0] > 0: z[i, 2] += sharp
        if d[i, 1] < 0: z[i, 1] += sharp
        if d[i, 1] > 0: z[i, 3] += sharp
        if not d[i].any(): z[i, 4] += 2 * sharp
    if policy == 1:
        z = torch.softmax(torch.from_numpy(z), dim=-1).numpy()
    return z.astype(np.float32)


def episodes(Sim, simmod):
    rng = np.random.default_rng(77001)
    uni = Uniforms()
    simmod.random.choice = lambda l: l[0]
    real_multinomial = torch.multinomial
    torch.multinomial = uni.multinomial
    try:
        cases = [("n4_map8_softmax", 4, 8, 0.0, 6, 0, 24, 6.0), ("n4_map8_sum", 4, 8, 0.0, 6, 1, 30, 6.0),
                 ("n8_map10_softmax", 8, 10, 0.03, 4, 0, 30, 6.0), ("n8_map10_exp", 8, 10, 0.03, 4, 2, 40, 5.0),
                 ("n8_map10_sum", 8, 10, 0.03, 4, 1, 40, 6.0), ("n30_map20_exp", 30, 20, 0.04, 3, 2, 70, 5.0),
                 ("n30_map14_crowded_sum", 30, 14, 0.10, 3, 1, 30, 3.0), ("n30_map14_crowded_timeout_exp", 30, 14, 0.10, 2, 2, 6, 1.0)]
        for name, N, size, density, B, policy, maxstep, sharp in cases:
            T = maxstep + 3                       # calls past the end: the finalisation branch runs (repeatedly)
            rec = {k: [] for k in ("map", "pos0", "goal", "logits", "uniforms", "pos", "reach", "first_move", "end_step", "done",
                                   "predict_collision", "key", "flowtime", "makespan")}
            for b in range(B):
                m, pos, goal = scenario(rng, N, size, density, far_goals=True)
                if b == 1:
                    goal[0] = pos[0]                  # starts on its goal
                s = make_sim(Sim, N, m, pos, goal, maxstep, policy)
                per = {k: [] for k in ("logits", "uniforms", "pos", "reach", "first_move", "end_step", "done", "predict_collision",
                                       "key", "flowtime", "makespan")}
                for t in range(T):
                    lg = step_logits(rng, s.current_positions.astype(np.int64), goal, policy, sharp)
                    u = rng.random(N)
                    uni.queue = list(u) if policy else []
                    keys = []
                    orig = s.convectToActionKey

                    def recording(v, o=orig):
                        keys.append(int(o(v)))
                        return torch.tensor(keys[-1])

                    s.convectToActionKey = recording
                    done, _, pc = Sim.move(s, [torch.from_numpy(lg[i:i + 1]) for i in range(N)], t)
                    s.convectToActionKey = orig
                    per["logits"].append(lg); per["uniforms"].append(u)
                    per["pos"].append(s.current_positions.astype(np.int32).copy())
                    per["reach"].append(s.reach_goal.astype(np.uint8).copy())
                    per["first_move"].append(s.first_move.astype(np.int32).copy())
                    per["end_step"].append(s.end_step.astype(np.int32).copy())
                    per["done"].append(int(done)); per["predict_collision"].append(int(pc))
                    per["key"].append(np.array(keys, np.int32) if keys else np.full(N, -1, np.int32))
                    per["flowtime"].append(int(s.flowtimePredict)); per["makespan"].append(int(s.makespanPredict))
                rec["map"].append(m.astype(np.uint8)); rec["pos0"].append(pos.astype(np.int32)); rec["goal"].append(goal.astype(np.int32))
                for k, v in per.items():
                    rec[k].append(np.stack(v) if isinstance(v[0], np.ndarray) else np.array(v, np.int32))
            path = os.path.join(OUT, "simepisode_%s.npz" % name)
            np.savez_compressed(path, policy=np.int32(policy), maxstep=np.int32(maxstep), **{k: np.stack(v) for k, v in rec.items()})
            done = np.stack(rec["done"])
            print("wrote", path, os.path.getsize(path) // 1024, "KB", "episodes finished by arrival:", int(done[:, -1].sum()), "/", B)
    finally:
        torch.multinomial = real_multinomial


def radii(Sim):
    rng = np.random.default_rng(77002)
    for name, N, size, density, B, commR in (("n10_map20_r2", 10, 20, 0.05, 8, 2.0), ("n20_map28_r3", 20, 28, 0.05, 6, 3.0),
                                             ("n100_map50_r4", 100, 50, 0.05, 3, 4.0), ("n12_map10_r7", 12, 10, 0.1, 3, 7.0)):
        poss, rs, Ss, Ssym = [], [], [], []
        for b in range(B):
            _, pos, _ = scenario(rng, N, size, density, far_goals=False)
            out = []
            for sym in (False, True):
                cfg = types.SimpleNamespace(num_agents=N, symmetric_norm=sym, commR=commR)
                fake = types.SimpleNamespace(config=cfg, communicationRadius=commR, zeroTolerance=1e-9)
                fake.get_maxEigenValue = lambda mat, f=fake: Sim.get_maxEigenValue(f, mat)
                S, r, conn = Sim.computeAdjacencyMatrix(fake, 0, pos[None].astype(np.float64), commR)
                assert conn
                out.append((S[0], r))
            assert out[0][1] == out[1][1]
            poss.append(pos.astype(np.int32)); rs.append(out[0][1]); Ss.append(out[0][0]); Ssym.append(out[1][0])
        path = os.path.join(OUT, "simradius_%s.npz" % name)
        np.savez_compressed(path, pos=np.stack(poss), commR=np.float64(commR), radius=np.array(rs, np.float64), S=np.stack(Ss),
                            S_symnorm=np.stack(Ssym))
        print("wrote", path, os.path.getsize(path) // 1024, "KB", "radii:", np.round(rs, 3))


def main():
    _, Sim = load_reference_frontend()
    episodes(Sim, sys.modules["utils.new_simulator"])
    radii(Sim)


if __name__ == "__main__":
    main()
