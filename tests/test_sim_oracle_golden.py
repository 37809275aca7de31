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


EPISODE = sorted(glob.glob(os.path.join(GOLDEN, "simepisode_*.npz")))
RADIUS = sorted(glob.glob(os.path.join(GOLDEN, "simradius_*.npz")))


def test_episode_fixture_inventory():
    assert len(EPISODE) == 8 and len(RADIUS) == 4
    policies = {int(np.load(p)["policy"]) for p in EPISODE}
    assert policies == {0, 1, 2}
    done = np.concatenate([np.load(p)["done"][:, -1] for p in EPISODE])
    assert done.any() and not done.all()            # both endings: everybody arrived / step budget spent


@pytest.mark.parametrize("path", EPISODE, ids=[os.path.basename(p)[:-4] for p in EPISODE])
def test_episode_matches_reference(path):
    """oracle.episode_step against multiRobotSimNew.move stepped by the reference itself (random.choice := first,
    torch.multinomial := inverse CDF over recorded uniforms): sampled keys, positions and all bookkeeping after EVERY
    call, including the calls past the end of the episode."""
    z = np.load(path)
    policy, maxstep = int(z["policy"]), int(z["maxstep"])
    for b in range(z["pos0"].shape[0]):
        st = so.EpisodeState(z["map"][b], z["pos0"][b], z["goal"][b], maxstep)
        for t in range(z["logits"].shape[1]):
            done, pc, keys = so.episode_step(st, z["logits"][b, t], t, policy, z["uniforms"][b, t])
            assert int(done) == z["done"][b, t] and int(pc) == z["predict_collision"][b, t], (b, t)
            if keys is None:
                assert (z["key"][b, t] == -1).all()
            else:
                np.testing.assert_array_equal(keys, z["key"][b, t])
            np.testing.assert_array_equal(st.pos, z["pos"][b, t])
            np.testing.assert_array_equal(st.reach_goal, z["reach"][b, t])
            np.testing.assert_array_equal(st.first_move, z["first_move"][b, t])
            np.testing.assert_array_equal(st.end_step, z["end_step"][b, t])
            assert st.flowtime == z["flowtime"][b, t] and st.makespan == z["makespan"][b, t], (b, t)


@pytest.mark.parametrize("path", RADIUS, ids=[os.path.basename(p)[:-4] for p in RADIUS])
def test_step0_radius_matches_reference(path):
    """connect_radius + gso_from_positions against computeAdjacencyMatrix(step=0): the grown radius bit for bit (same
    float64 products), then the GSO at that radius."""
    z = np.load(path)
    grew = 0
    for b in range(z["pos"].shape[0]):
        r, steps = so.connect_radius(z["pos"][b], float(z["commR"]))
        assert r == z["radius"][b]
        grew += steps > 1
        for key, sym in (("S", False), ("S_symnorm", True)):
            got = so.gso_from_positions(z["pos"][b], r, symmetric_norm=sym)
            np.testing.assert_array_equal(got != 0, z[key][b] != 0)
            np.testing.assert_allclose(got, z[key][b], rtol=1e-12, atol=0)
    if "r7" not in path:
        assert grew > 0
