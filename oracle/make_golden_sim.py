"""Generates tests/golden/sim_*.npz by running the REAL reference simulator front-end (imported from /root/reference,
build container only): AgentState.toInputTensor ('Project_G') and multiRobotSimNew.computeAdjacencyMatrix.
TEST INFRASTRUCTURE.      python oracle/make_golden_sim.py
Fixtures hold data only: obstacle maps, agent / goal coordinates, the state tensors and GSOs the reference produced."""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle._ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def load_reference_frontend():
    import_reference()
    for pk in ("dataloader", "offlineExpert"):
        if pk not in sys.modules:
            m = types.ModuleType(pk)
            m.__path__ = ["/root/reference/" + pk]
            sys.modules[pk] = m
    st = importlib.import_module("dataloader.statetransformer_Guidance")
    sim = importlib.import_module("utils.new_simulator")
    return st.AgentState, sim.multiRobotSimNew


def scenario(rng, N, size, density, far_goals):
    m = (rng.random((size, size)) < density).astype(np.int64)
    free = np.argwhere(m == 0)
    idx = rng.permutation(len(free))
    pos = free[idx[:N]]
    goal = free[idx[N:2 * N]] if far_goals else free[rng.permutation(len(free))[:N]]
    return m, pos.astype(np.int64), goal.astype(np.int64)


def main():
    AgentState, Sim = load_reference_frontend()
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20260928)
    cases = [("n10_map20", 10, 20, 0.10, 6, 7.0), ("n20_map28", 20, 28, 0.10, 4, 7.0), ("n100_map50", 100, 50, 0.08, 3, 7.0),
             ("n12_map10_dense", 12, 10, 0.20, 4, 3.0), ("n30_map40_r77", 30, 40, 0.05, 3, 7.7)]
    for name, N, size, density, B, commR in cases:
        cfg = types.SimpleNamespace(num_agents=N, FOV=9, guidance="Project_G", symmetric_norm=False, commR=commR)
        maps, poss, goals, xs, Ss, Ssym = [], [], [], [], [], []
        for b in range(B):
            m, pos, goal = scenario(rng, N, size, density, far_goals=(b % 2 == 0))
            if b == 1:                                   # a few agents ON their goals, and goals just outside the FOV
                goal[0] = pos[0]
                goal[1] = np.clip(pos[1] + np.array([5, 0]), 0, size - 1)
                goal[2] = np.clip(pos[2] + np.array([-5, 5]), 0, size - 1)
            st = AgentState(cfg)
            st.setmap(m)
            x = st.toInputTensor(goal.astype(np.float64), pos.astype(np.float64)).numpy()
            assert x.shape == (N, 3, 11, 11) and set(np.unique(x)) <= {0.0, 1.0}
            fake = types.SimpleNamespace(config=cfg, communicationRadius=commR, zeroTolerance=1e-9)
            fake.get_maxEigenValue = lambda mat: Sim.get_maxEigenValue(fake, mat)
            S, _, _ = Sim.computeAdjacencyMatrix(fake, 5, pos[None].astype(np.float64), commR)
            cfg.symmetric_norm = True
            S2, _, _ = Sim.computeAdjacencyMatrix(fake, 5, pos[None].astype(np.float64), commR)
            cfg.symmetric_norm = False
            maps.append(m); poss.append(pos); goals.append(goal); xs.append(x.astype(np.uint8)); Ss.append(S[0]); Ssym.append(S2[0])
        path = os.path.join(OUT, "sim_%s.npz" % name)
        np.savez_compressed(path, map=np.stack(maps).astype(np.uint8), pos=np.stack(poss).astype(np.int32),
                            goal=np.stack(goals).astype(np.int32), x=np.stack(xs), S=np.stack(Ss), S_symnorm=np.stack(Ssym),
                            commR=np.float64(commR), FOV=np.int64(9))
        print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
