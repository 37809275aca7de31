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
