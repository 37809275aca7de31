"""Seeded synthetic inputs of SURVEY.md section 8(d): binary 3-channel FOV state tensors shaped like
AgentState.toInputTensor's output (dataloader/statetransformer_Guidance.py:185-239) and comm-radius
GSOs built like multiRobotSimNew.computeAdjacencyMatrix_fixedCommRadius (utils/new_simulator.py:816-846)."""
import types

import numpy as np
import torch


def make_config(**kw):
    base = dict(num_agents=10, FOV=9, bottleneckFeature=128, numInputFeatures=128, nGraphFilterTaps=2,
                nAttentionHeads=1, use_dropout=False, CNN_mode="ResNetLarge_withMLP", attentionMode="KeyQuery",
                AttentionConcat=True, GSO_mode="dist_GSO", device="cuda:0", bottleneckMode="BottomNeck_only",
                batch_numAgent=True)
    base.update(kw)
    return types.SimpleNamespace(**base)


def fov_states(B, N, seed=1337, fov=9):
    g = torch.Generator().manual_seed(seed)
    w = fov + 2
    x = torch.zeros(B, N, 3, w, w)
    x[:, :, 0, 1:-1, 1:-1] = (torch.rand(B, N, fov, fov, generator=g) < 0.1).float()
    gi = torch.randint(0, fov * fov, (B, N), generator=g)
    goal = torch.zeros(B, N, w * w)
    goal.scatter_(2, ((gi // fov + 1) * w + gi % fov + 1).unsqueeze(-1), 1.0)
    x[:, :, 1] = goal.view(B, N, w, w)
    x[:, :, 2, 1:-1, 1:-1] = (torch.rand(B, N, fov, fov, generator=g) < min(0.5, N / 400.0)).float()
    x[:, :, 2, w // 2, w // 2] = 1.0
    return x


def calibration_states(fov=9, agents=2048, seed=0):
    """The CANONICAL calibration batch of the split arithmetic's activation scales (planner._calibrate): `agents` state
    tensors (agents, 3, fov+2, fov+2) that depend on nothing but (fov, agents, seed) - every process that holds the same
    weights measures the same layer magnitudes from it, whatever batch or shard it is given first.  Same three binary
    channels as fov_states (obstacles / goal / neighbouring agents, AgentState.toInputTensor 'Project_G':
    dataloader/statetransformer_Guidance.py:185-239), with the obstacle and agent densities swept per agent over the range
    the maps of SURVEY.md section 8 span (0 .. 0.3 obstacles, 0 .. 0.5 agents) and the goal either inside the field of
    view or projected onto its border ring."""
    g = torch.Generator().manual_seed(seed)
    w = fov + 2
    x = torch.zeros(agents, 3, w, w)
    d_obs = torch.linspace(0.0, 0.3, agents)[torch.randperm(agents, generator=g)].view(-1, 1, 1)
    d_agt = torch.linspace(0.0, 0.5, agents)[torch.randperm(agents, generator=g)].view(-1, 1, 1)
    x[:, 0, 1:-1, 1:-1] = (torch.rand(agents, fov, fov, generator=g) < d_obs).float()
    x[:, 2, 1:-1, 1:-1] = (torch.rand(agents, fov, fov, generator=g) < d_agt).float()
    x[:, 2, w // 2, w // 2] = 1.0
    gy = torch.randint(0, w, (agents,), generator=g)
    gx = torch.randint(0, w, (agents,), generator=g)
    x[torch.arange(agents), 1, gy, gx] = 1.0
    return x


def comm_gso(B, N, map_w, comm_radius=7.0, seed=1337, dtype=torch.float32, normalize=True):
    """Uniform integer positions on a map_w x map_w grid, W = (dist < R) with zero diagonal,
    S = W / lambda_max(W) (symmetric, so eigvalsh)."""
    rng = np.random.default_rng(seed)
    pos = rng.integers(0, map_w, size=(B, N, 2)).astype(np.float64)
    d = np.sqrt(((pos[:, :, None, :] - pos[:, None, :, :]) ** 2).sum(-1))
    Wm = (d < comm_radius).astype(np.float64)
    idx = np.arange(N)
    Wm[:, idx, idx] = 0.0
    if normalize:
        lam = np.linalg.eigvalsh(Wm)[:, -1]
        lam[lam <= 0] = 1.0
        Wm = Wm / lam[:, None, None]
    return torch.from_numpy(Wm).to(dtype)


def random_gso(B, N, density, seed=1337, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    Wm = (torch.rand(B, N, N, generator=g) < density).float()
    Wm = torch.triu(Wm, 1)
    return (Wm + Wm.transpose(1, 2)).to(dtype)


def directed_gso(B, N, density, seed=1337, dtype=torch.float64):
    """DIRECTED random GSO (M != M^T: every ordered pair drawn independently) carrying the entries the reference's mask rule
    `|S| > 1e-9` (graphML.py:1274-1276) has to get right: values at 5e-10 (no edge) and -3e-9 (an edge), one NaN (no edge),
    an isolated node, a node whose ONLY incident edge is one-way INTO it (its softmax row is empty, its column is not:
    the row-softmax / column-aggregate orientation of graphML.py:1757), and one that only sends; float64 entries are
    1 / lambda_max-scaled like new_simulator.py:820-846 hands them over."""
    g = torch.Generator().manual_seed(seed)
    W = (torch.rand(B, N, N, generator=g) < density).double()
    idx = torch.arange(N)
    W[:, idx, idx] = 0.0
    for b in range(B):
        if N >= 6:
            iso, sink, src = (3 + b) % N, (4 + b) % N, (5 + b) % N
            peer = (iso + 3) % N
            for n in (iso, sink, src):
                W[b, n, :] = 0
                W[b, :, n] = 0
            if peer not in (iso, sink, src):
                W[b, peer, sink] = 1.0       # edge peer -> sink only: row `sink` has no edges, column `sink` has one
                W[b, src, peer] = 1.0        # edge src -> peer only: column `src` has none
        lam = float(np.abs(np.linalg.eigvals(W[b].numpy())).max()) or 1.0
        W[b] = W[b] / max(lam, 1e-6)
        if N >= 12:
            k, l = (8 + b) % N, (10 + b) % N
            W[b, k, l] = 5e-10
            W[b, l, k] = -3e-9
    if N >= 2:
        W[0, 0, N - 1] = float("nan")
    return W.to(dtype)
