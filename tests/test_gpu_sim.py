"""GPU parity of the on-device simulator front-end through the C ABI: against fixtures made by the
reference's AgentState / multiRobotSimNew, and against the oracle at benchmark sizes."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIM = sorted(glob.glob(os.path.join(GOLDEN, "sim_*.npz")))


@pytest.mark.parametrize("path", SIM, ids=[os.path.basename(p)[:-4] for p in SIM])
def test_fov_states_bit_exact_vs_reference(gpu_device, path):
    from magat_pathplanning_amd.simulator import batched_fov_states
    z = np.load(path)
    x = batched_fov_states(torch.from_numpy(z["map"]).to(gpu_device), torch.from_numpy(z["pos"]).to(gpu_device),
                           torch.from_numpy(z["goal"]).to(gpu_device), int(z["FOV"]))
    np.testing.assert_array_equal(x.cpu().numpy(), z["x"].astype(np.float32))


@pytest.mark.parametrize("path", SIM, ids=[os.path.basename(p)[:-4] for p in SIM])
def test_gso_vs_reference(gpu_device, path):
    from magat_pathplanning_amd.simulator import batched_gso
    z = np.load(path)
    pos = torch.from_numpy(z["pos"]).to(gpu_device)
    for key, sym in (("S", False), ("S_symnorm", True)):
        S = batched_gso(pos, float(z["commR"]), symmetric_norm=sym).cpu().numpy()
        np.testing.assert_array_equal(S != 0, z[key] != 0)                    # edge structure: exact
        np.testing.assert_allclose(S, z[key], rtol=1e-9, atol=0)              # lambda_max: power iteration vs eigvalsh
    S32 = batched_gso(pos, float(z["commR"]), dtype=torch.float32).cpu().numpy()
    np.testing.assert_allclose(S32, z["S"].astype(np.float32), rtol=1e-6, atol=0)
    W = batched_gso(pos, float(z["commR"]), normalize=False).cpu().numpy()
    np.testing.assert_array_equal(W, (z["S"] != 0).astype(np.float64))


def test_front_end_at_benchmark_size_and_into_the_model(gpu_device):
    """c3-sized batch: every instance against the oracle's edge structure / state tensors on a sample, an edgeless
    instance, a shared (unbatched) map, and the result fed straight into the model (closed loop stays on device)."""
    from oracle import sim_oracle as so
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.simulator import batched_fov_states, batched_gso
    from magat_pathplanning_amd.synthetic import make_config
    rng = np.random.default_rng(7)
    B, N, size = 64, 100, 50
    m = (rng.random((size, size)) < 0.08).astype(np.uint8)
    free = np.argwhere(m == 0)
    pos = np.stack([free[rng.permutation(len(free))[:N]] for _ in range(B)]).astype(np.int32)
    goal = np.stack([free[rng.permutation(len(free))[:N]] for _ in range(B)]).astype(np.int32)
    pos[3] = np.stack([np.arange(N) % 50, (np.arange(N) // 50) * 40], 1)[:N] * np.array([1, 1])   # a line: sparse graph
    pos[5, :, 0] = np.arange(N) // 10 * 5
    pos[5, :, 1] = np.arange(N) % 10 * 5 + 2            # 5-cell lattice ...
    dpos, dgoal = torch.from_numpy(pos).to(gpu_device), torch.from_numpy(goal).to(gpu_device)
    S, lam = batched_gso(dpos, 7.0, return_lambda=True)
    Sfar = batched_gso(dpos, 0.5)                        # radius below the lattice pitch: no edges anywhere
    assert float(Sfar.abs().max()) == 0.0
    x = batched_fov_states(torch.from_numpy(m).to(gpu_device), dpos, dgoal, 9)
    for b in (0, 3, 5, B - 1):
        ref = so.gso_from_positions(pos[b], 7.0)
        np.testing.assert_array_equal(S[b].cpu().numpy() != 0, ref != 0)
        np.testing.assert_allclose(S[b].cpu().numpy(), ref, rtol=1e-9, atol=0)
        np.testing.assert_array_equal(x[b].cpu().numpy(), so.fov_states(m, pos[b], goal[b], 9).astype(np.float32))
    cfg = make_config(num_agents=N, nGraphFilterTaps=3, nAttentionHeads=4, bottleneckMode="BottomNeck_skipConcat",
                      device=str(gpu_device))
    net = DecentralPlannerGATNet(cfg).to(gpu_device).eval()
    with torch.no_grad():
        net.addGSO(S)
        logits = net(x)
    assert logits.shape == (B * N, 5) and bool(torch.isfinite(logits).all())


def test_front_end_rejects_cpu_tensors():
    from magat_pathplanning_amd import _native as nat
    from magat_pathplanning_amd.simulator import batched_gso
    with pytest.raises(nat.MagatNativeError):
        batched_gso(torch.zeros(1, 4, 2, dtype=torch.int32), 7.0)


STEP = sorted(glob.glob(os.path.join(GOLDEN, "simstep_*.npz")))


@pytest.mark.parametrize("path", STEP, ids=[os.path.basename(p)[:-4] for p in STEP])
def test_move_shielding_vs_reference(gpu_device, path):
    """magat_sim_move against multiRobotSimNew.check_collision outputs: identical to the reference run with
    random.choice := first claimant on every scenario, hence identical to the reference wherever its random tie-break
    does not matter (`det`)."""
    from magat_pathplanning_amd.simulator import batched_move
    z = np.load(path)
    pos = torch.from_numpy(z["pos"]).to(gpu_device).contiguous()
    before = pos.clone()
    out = batched_move(torch.from_numpy(z["map"]).to(gpu_device), pos, actions=torch.from_numpy(z["action"]).to(gpu_device))
    mv = out["moves"].cpu().numpy()
    np.testing.assert_array_equal(mv, z["move_first"])
    det = z["det"].astype(bool)
    np.testing.assert_array_equal(mv[det], z["move_last"][det])
    np.testing.assert_array_equal((pos - before).cpu().numpy(), mv.astype(np.int32))


def test_closed_loop_step_on_device(gpu_device):
    """front-end -> model -> decode/shield/advance without leaving the device; decode and shielding checked against
    the oracle on every instance, invariants (distinct cells, no obstacle, inside the arena) on the new positions."""
    from oracle import sim_oracle as so
    from magat_pathplanning_amd import DecentralPlannerGATNet
    from magat_pathplanning_amd.simulator import batched_fov_states, batched_gso, batched_move
    from magat_pathplanning_amd.synthetic import make_config
    rng = np.random.default_rng(11)
    B, N, size = 12, 40, 18
    m = (rng.random((size, size)) < 0.1).astype(np.uint8)
    free = np.argwhere(m == 0)
    pos = np.stack([free[rng.permutation(len(free))[:N]] for _ in range(B)]).astype(np.int32)
    goal = np.stack([free[rng.permutation(len(free))[:N]] for _ in range(B)]).astype(np.int32)
    dm = torch.from_numpy(m).to(gpu_device)
    dpos, dgoal = torch.from_numpy(pos).to(gpu_device), torch.from_numpy(goal).to(gpu_device)
    cfg = make_config(num_agents=N, nGraphFilterTaps=2, nAttentionHeads=2, device=str(gpu_device))
    torch.manual_seed(3)
    net = DecentralPlannerGATNet(cfg).to(gpu_device).eval()
    for step in range(3):
        cur = dpos.cpu().numpy().copy()
        with torch.no_grad():
            net.addGSO(batched_gso(dpos, 5.0))
            logits = net(batched_fov_states(dm, dpos, dgoal, 9))
        out = batched_move(dm, dpos, logits=logits, goal=dgoal)
        lg = logits.view(B, N, 5).cpu().numpy()
        new = dpos.cpu().numpy()
        for b in range(B):
            want_pos, want_act, want_reached = so.move_step(m, cur[b], goal[b], lg[b])
            np.testing.assert_array_equal(out["actions"][b].cpu().numpy(), want_act)
            np.testing.assert_array_equal(new[b], want_pos)
            np.testing.assert_array_equal(out["reached"][b].cpu().numpy(), want_reached)
            assert len({tuple(p) for p in new[b]}) == N and (m[new[b][:, 0], new[b][:, 1]] == 0).all()
            assert new[b].min() >= 0 and new[b].max() < size


def test_gso_large_instances_vs_oracle(gpu_device):
    """N above the workgroup size (strided row loops) and a disconnected graph (two far-apart clusters): lambda_max is
    the maximum over components."""
    from oracle import sim_oracle as so
    from magat_pathplanning_amd.simulator import batched_gso
    rng = np.random.default_rng(5)
    N = 300
    pos = rng.integers(0, 90, size=(3, N, 2)).astype(np.int32)
    pos[1, : N // 2] = rng.integers(0, 30, size=(N // 2, 2))
    pos[1, N // 2:] = rng.integers(200, 215, size=(N - N // 2, 2))        # two clusters, no edge between them
    S, lam = batched_gso(torch.from_numpy(pos).to(gpu_device), 7.0, return_lambda=True)
    for b in range(3):
        ref = so.gso_from_positions(pos[b], 7.0)
        got = S[b].cpu().numpy()
        np.testing.assert_array_equal(got != 0, ref != 0)
        np.testing.assert_allclose(got, ref, rtol=1e-9, atol=0)
