"""oracle/sim_oracle.py (CPU restatement of the simulator front-end) against fixtures produced by the
reference's own AgentState / multiRobotSimNew code (oracle/make_golden_sim.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import sim_oracle as so

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIM = sorted(glob.glob(os.path.join(GOLDEN, "sim_*.npz")))


def test_sim_fixture_inventory():
    assert len(SIM) == 5


@pytest.mark.parametrize("path", SIM, ids=[os.path.basename(p)[:-4] for p in SIM])
def test_fov_states_bit_exact(path):
    z = np.load(path)
    for b in range(z["x"].shape[0]):
        got = so.fov_states(z["map"][b], z["pos"][b], z["goal"][b], int(z["FOV"]))
        np.testing.assert_array_equal(got, z["x"][b])


@pytest.mark.parametrize("path", SIM, ids=[os.path.basename(p)[:-4] for p in SIM])
def test_gso_matches_reference(path):
    z = np.load(path)
    for b in range(z["S"].shape[0]):
        for key, sym in (("S", False), ("S_symnorm", True)):
            got = so.gso_from_positions(z["pos"][b], float(z["commR"]), symmetric_norm=sym)
            np.testing.assert_array_equal(got != 0, z[key][b] != 0)          # edge structure exact
            np.testing.assert_allclose(got, z[key][b], rtol=1e-12, atol=0)


def test_projected_goal_octants():
    """every direction class of projectedgoal (statetransformer_Guidance.py:103-124), incl. the exact diagonals and
    round-half-to-even ties; expected values were produced by the reference's AgentState.projectedgoal"""
    want = {(10, 0): (10, 5), (-10, 0): (0, 5), (0, 10): (5, 10), (0, -10): (5, 0), (10, 10): (10, 10), (-10, 10): (0, 10),
            (-10, -10): (0, 0), (10, -10): (10, 0), (10, 20): (7, 10), (6, -20): (7, 0), (20, 7): (10, 7), (-20, -3): (0, 4),
            (20, 10): (10, 7), (20, 6): (10, 7), (3, -10): (7, 0), (-7, 10): (1, 10), (10, -3): (10, 3), (-10, 5): (0, 7)}
    for (dx, dy), rc in want.items():
        assert so.projected_goal(9, 20, 20, 20 + dx, 20 + dy) == rc, (dx, dy)


STEP = sorted(glob.glob(os.path.join(GOLDEN, "simstep_*.npz")))


def test_simstep_fixture_inventory():
    assert len(STEP) == 4


@pytest.mark.parametrize("path", STEP, ids=[os.path.basename(p)[:-4] for p in STEP])
def test_shielding_matches_reference(path):
    """oracle.shield_moves against multiRobotSimNew.check_collision: identical wherever the reference's random tie-break
    is irrelevant (`det`), and identical to the reference run with random.choice := "first claimant" everywhere (the
    claim lists are in agent order, so "first" IS the lowest-index rule of the deterministic restatement)."""
    z = np.load(path)
    for b in range(z["pos"].shape[0]):
        mv, _ = so.shield_moves(z["map"][b], z["pos"][b], so.MOVES[z["action"][b]])
        np.testing.assert_array_equal(mv, z["move_first"][b])
        if z["det"][b]:
            np.testing.assert_array_equal(mv, z["move_last"][b])
        new = z["pos"][b] + mv
        assert len({tuple(p) for p in new}) == len(new)                      # no two agents in one cell
        assert (z["map"][b][new[:, 0], new[:, 1]] == 0).all()                # nobody inside an obstacle
