"""Batched on-device simulator front-end: what the reference does per instance, per
step, in numpy on the host right before the model call -

    multiRobotSimNew.getGSO / computeAdjacencyMatrix   (utils/new_simulator.py:301-321, 745-806)
    multiRobotSimNew.getCurrentState -> AgentState.toInputTensor, guidance 'Project_G'
                                                        (utils/new_simulator.py:279-296; dataloader/statetransformer_Guidance.py:136-239)

- for B independent planning instances at once, on the GPU, so that a batched closed loop never leaves the device:

    S = batched_gso(pos, config.commR)                   # (B,N,N), what model.addGSO() takes
    x = batched_fov_states(obstacle_map, pos, goal, 9)   # (B,N,3,11,11), what model.forward() takes

HIP only (csrc/sim_frontend.hip): CPU tensors raise MagatNativeError."""
import torch

from . import _native as nat


def _dev_i32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise nat.MagatNativeError("%s must be a device tensor (no CPU fallback)" % name)
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    return t.contiguous()


def batched_gso(pos, comm_radius, symmetric_norm=False, normalize=True, dtype=torch.float64, return_lambda=False):
    """pos (B,N,2) integer agent coordinates (row, col) on the device -> S (B,N,N) `dtype` (float64 like the simulator
    hands it over, or float32 like the dataloader).  Fixed communication radius (every step after the first; the
    step-0 radius growth until the graph is connected stays with the caller)."""
    pos = _dev_i32(pos, "pos")
    assert pos.dim() == 3 and pos.shape[2] == 2, "pos must be (B,N,2)"
    assert dtype in (torch.float32, torch.float64)
    B, N, _ = pos.shape
    S = torch.empty(B, N, N, dtype=dtype, device=pos.device)
    lam = torch.empty(B, dtype=torch.float64, device=pos.device)
    with torch.cuda.device(pos.device):
        nat.check(nat.lib().magat_sim_gso(nat.ptr(pos), float(comm_radius), 1 if symmetric_norm else 0,
                                          1 if normalize else 0, nat.ptr(S), int(dtype == torch.float64), nat.ptr(lam),
                                          B, N, nat.current_stream(pos.device)), "magat_sim_gso")
    return (S, lam) if return_lambda else S


def batched_fov_states(obstacle_map, pos, goal, FOV=9):
    """obstacle_map (H,W) or (B,H,W) uint8/bool device tensor (non-zero = obstacle), pos / goal (B,N,2) integer
    (row, col) -> x (B,N,3,FOV+2,FOV+2) float32, identical to stacking AgentState.toInputTensor over the instances."""
    pos, goal = _dev_i32(pos, "pos"), _dev_i32(goal, "goal")
    if not isinstance(obstacle_map, torch.Tensor) or not obstacle_map.is_cuda:
        raise nat.MagatNativeError("obstacle_map must be a device tensor (no CPU fallback)")
    m = obstacle_map.to(torch.uint8).contiguous()
    B, N, _ = pos.shape
    assert goal.shape == pos.shape
    batched = m.dim() == 3
    assert m.dim() in (2, 3) and (not batched or m.shape[0] == B)
    H, W = m.shape[-2], m.shape[-1]
    x = torch.empty(B, N, 3, FOV + 2, FOV + 2, dtype=torch.float32, device=pos.device)
    with torch.cuda.device(pos.device):
        nat.check(nat.lib().magat_sim_fov_states(nat.ptr(m), 1 if batched else 0, H, W, nat.ptr(pos), nat.ptr(goal),
                                                 nat.ptr(x), FOV, B, N, nat.current_stream(pos.device)),
                  "magat_sim_fov_states")
    return x
