"""Generates tests/golden/simstep_*.npz with the REAL reference collision shielding (multiRobotSimNew.check_collision,
utils/new_simulator.py:334-454; build container only).  TEST INFRASTRUCTURE.    python oracle/make_golden_sim_step.py
Each scenario is run twice, with random.choice replaced by "first" and by "last": when both runs agree the reference's
random tie-break was irrelevant and the result is a deterministic target (`det` = 1); otherwise both outcomes are kept
(the on-device rule, lowest index wins, must equal the "first"... only by accident, so those cases are used for the
validity invariants only)."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.make_golden_sim import load_reference_frontend, scenario  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
MOVES = np.array([[-1, 0], [0, -1], [1, 0], [0, 1], [0, 0]])


def main():
    _, Sim = load_reference_frontend()
    simmod = sys.modules["utils.new_simulator"]
    rng = np.random.default_rng(424242)
    cases = [("n10_map20", 10, 20, 0.10, 24), ("n30_map12_crowded", 30, 12, 0.12, 24), ("n100_map50", 100, 50, 0.08, 8),
             ("n60_map16_crowded", 60, 16, 0.10, 16)]
    for name, N, size, density, B in cases:
        maps, poss, acts, mv_first, mv_last, det = [], [], [], [], [], []
        for b in range(B):
            m, pos, _ = scenario(rng, N, size, density, far_goals=False)
            act = rng.integers(0, 5, size=N)
            if b % 3 == 0:                        # bias towards motion: more conflicts
                act = rng.integers(0, 4, size=N)
            fake = types.SimpleNamespace(config=types.SimpleNamespace(num_agents=N), size_map=m.shape, stop=np.array([0, 0]),
                                         wall_dict={tuple(p): i for i, p in enumerate(np.argwhere(m != 0))})
            res = []
            for pick in (lambda l: l[0], lambda l: l[-1]):
                simmod.random.choice = pick
                out = Sim.check_collision(fake, pos.astype(np.float64), MOVES[act].astype(np.float64))
                res.append(np.asarray(out[0]).astype(np.int8))
            maps.append(m.astype(np.uint8)); poss.append(pos.astype(np.int32)); acts.append(act.astype(np.int32))
            mv_first.append(res[0]); mv_last.append(res[1]); det.append(int((res[0] == res[1]).all()))
        path = os.path.join(OUT, "simstep_%s.npz" % name)
        np.savez_compressed(path, map=np.stack(maps), pos=np.stack(poss), action=np.stack(acts), move_first=np.stack(mv_first),
                            move_last=np.stack(mv_last), det=np.array(det, np.int8))
        print("wrote", path, os.path.getsize(path) // 1024, "KB", "deterministic cases:", int(np.sum(det)), "/", B)


if __name__ == "__main__":
    main()
