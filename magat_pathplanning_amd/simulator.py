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


def batched_move(obstacle_map, pos, logits=None, actions=None, goal=None):
    """One closed-loop step for B instances on the device (multiRobotSimNew.move, utils/new_simulator.py:471-520):
    decode the action keys from the model's logits (B*N,5) (or take `actions` (B,N)), shield the proposed moves exactly
    like check_collision (:334-454; lowest index instead of random.choice among moving claimants) and advance `pos`
    IN PLACE.  Returns dict(actions (B,N) int32, moves (B,N,2) int8, reached (B,N) bool or None, flags (B,) int32)."""
    if pos.dtype != torch.int32 or not pos.is_cuda or not pos.is_contiguous():
        raise nat.MagatNativeError("pos must be a contiguous int32 device tensor (it is updated in place)")
    assert (logits is None) != (actions is None), "give logits or actions"
    B, N, _ = pos.shape
    dev = pos.device
    m = obstacle_map.to(torch.uint8).contiguous()
    if not m.is_cuda:
        raise nat.MagatNativeError("obstacle_map must be a device tensor (no CPU fallback)")
    batched = m.dim() == 3
    H, W = m.shape[-2], m.shape[-1]
    lg = None if logits is None else logits.reshape(B * N, 5).contiguous().float()
    ac = None if actions is None else _dev_i32(actions, "actions").reshape(B * N)
    gl = None if goal is None else _dev_i32(goal, "goal")
    a_out = torch.empty(B, N, dtype=torch.int32, device=dev)
    mv = torch.empty(B, N, 2, dtype=torch.int8, device=dev)
    reached = torch.empty(B, N, dtype=torch.uint8, device=dev) if gl is not None else None
    flags = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        nat.check(nat.lib().magat_sim_move(nat.ptr(lg), nat.ptr(ac), nat.ptr(m), 1 if batched else 0, H, W, nat.ptr(pos),
                                           nat.ptr(gl), nat.ptr(a_out), nat.ptr(mv), nat.ptr(reached), nat.ptr(flags), B, N,
                                           nat.current_stream(dev)), "magat_sim_move")
    return dict(actions=a_out, moves=mv, reached=None if reached is None else reached.bool(), flags=flags)
