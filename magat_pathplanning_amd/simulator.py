"""Batched on-device simulator front-end: what the reference does per instance, per
step, in numpy on the host right before the model call -

    multiRobotSimNew.getGSO / computeAdjacencyMatrix   (utils/new_simulator.py:301-321, 745-806)
    multiRobotSimNew.getCurrentState -> AgentState.toInputTensor, guidance 'Project_G'
                                                        (utils/new_simulator.py:279-296; dataloader/statetransformer_Guidance.py:136-239)

- for B independent planning instances at once, on the GPU, so that a batched closed loop never leaves the device:

    S = batched_gso(pos, config.commR)                   # (B,N,N), what model.addGSO() takes
    x = batched_fov_states(obstacle_map, pos, goal, 9)   # (B,N,3,11,11), what model.forward() takes

and the step behind it (multiRobotSimNew.move, :471-549) with the episode state on the device:

    ep = BatchedEpisode(obstacle_map, pos, goal, maxstep, comm_radius=config.commR, action_select='exp_multinorm')
    S = ep.gso(); x = ep.states(); done = ep.step(logits)        # radius grown at step 0 like the reference

HIP only (csrc/sim_frontend.hip): CPU tensors raise MagatNativeError."""
import torch

from . import _native as nat


def _dev_i32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise nat.MagatNativeError("%s must be a device tensor (no CPU fallback)" % name)
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    return t.contiguous()


def batched_connect_radius(pos, comm_radius, max_steps=64, return_steps=False):
    """Step-0 branch of computeAdjacencyMatrix (utils/new_simulator.py:759-768) for B instances: r = R / 1.1, then
    r *= 1.1 until the graph (distance < r) is connected.  Returns radii (B,) float64 on the device (bit-equal to the
    reference's), to be passed to batched_gso for the rest of the episode."""
    pos = _dev_i32(pos, "pos")
    assert pos.dim() == 3 and pos.shape[2] == 2, "pos must be (B,N,2)"
    B, N, _ = pos.shape
    radii = torch.empty(B, dtype=torch.float64, device=pos.device)
    steps = torch.empty(B, dtype=torch.int32, device=pos.device)
    with torch.cuda.device(pos.device):
        nat.check(nat.lib().magat_sim_connect_radius(nat.ptr(pos), float(comm_radius), nat.ptr(radii), nat.ptr(steps), B, N,
                                                     int(max_steps), nat.current_stream(pos.device)),
                  "magat_sim_connect_radius")
    return (radii, steps) if return_steps else radii


def batched_gso(pos, comm_radius, symmetric_norm=False, normalize=True, dtype=torch.float64, return_lambda=False):
    """pos (B,N,2) integer agent coordinates (row, col) on the device -> S (B,N,N) `dtype` (float64 like the simulator
    hands it over, or float32 like the dataloader).  `comm_radius`: one number, or a (B,) float64 device tensor of
    per-instance radii (batched_connect_radius: the radius each instance grew to at step 0)."""
    pos = _dev_i32(pos, "pos")
    assert pos.dim() == 3 and pos.shape[2] == 2, "pos must be (B,N,2)"
    if dtype not in (torch.float32, torch.float64):      # (the kernels write 4- or 8-byte entries: nothing else fits S)
        raise TypeError("batched_gso: dtype must be torch.float32 or torch.float64, got %s" % (dtype,))
    if isinstance(comm_radius, torch.Tensor):
        B, N, _ = pos.shape
        if not comm_radius.is_cuda or comm_radius.dtype != torch.float64 or comm_radius.numel() != B:
            raise nat.MagatNativeError("per-instance radii must be a (B,) float64 device tensor")
        radii = comm_radius.contiguous()
        S = torch.empty(B, N, N, dtype=dtype, device=pos.device)
        lam = torch.empty(B, dtype=torch.float64, device=pos.device)
        with torch.cuda.device(pos.device):
            nat.check(nat.lib().magat_sim_gso_radii(nat.ptr(pos), nat.ptr(radii), 1 if symmetric_norm else 0,
                                                    1 if normalize else 0, nat.ptr(S), int(dtype == torch.float64),
                                                    nat.ptr(lam), B, N, nat.current_stream(pos.device)),
                      "magat_sim_gso_radii")
        return (S, lam) if return_lambda else S
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


POLICIES = {"soft_max": 0, "sum_multinorm": 1, "exp_multinorm": 2}


class BatchedEpisode:
    """B planning cases stepped together with ALL simulator state on the device - the batched counterpart of one
    multiRobotSimNew between setup() and the end of the episode (utils/new_simulator.py:108-221, 471-549):
    current positions, reach_goal / first_move / end_step, the step-0 communication radius, makespan and flowtime.

    action_select: 'soft_max' | 'sum_multinorm' | 'exp_multinorm' (config.action_select; the reference's default outside
    'test_trainingSet' mode is exp_multinorm, :134-145).  The multinomial policies draw from `generator` (a device
    torch.Generator; seeded runs repeat exactly) - one float64 uniform per agent and step, inverse CDF on the device."""

    def __init__(self, obstacle_map, pos, goal, maxstep, comm_radius, action_select="soft_max", symmetric_norm=False,
                 FOV=9, generator=None):
        if action_select not in POLICIES:
            raise ValueError("action_select must be one of %s" % sorted(POLICIES))
        self.pos = _dev_i32(pos, "pos").clone()
        self.goal = _dev_i32(goal, "goal")
        if not isinstance(obstacle_map, torch.Tensor) or not obstacle_map.is_cuda:
            raise nat.MagatNativeError("obstacle_map must be a device tensor (no CPU fallback)")
        self.map = obstacle_map.to(torch.uint8).contiguous()
        B, N, _ = self.pos.shape
        dev = self.pos.device
        self.B, self.N, self.dev = B, N, dev
        self.policy = POLICIES[action_select]
        self.maxstep = int(maxstep)
        self.comm_radius, self.symmetric_norm, self.FOV = float(comm_radius), bool(symmetric_norm), int(FOV)
        self.generator = generator
        self.currentstep = 0
        self.radii = None
        self.reach_goal = torch.zeros(B, N, dtype=torch.uint8, device=dev)
        self.first_move = torch.zeros(B, N, dtype=torch.int32, device=dev)
        self.end_step = torch.zeros(B, N, dtype=torch.int32, device=dev)
        self.done = torch.zeros(B, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(B, dtype=torch.int32, device=dev)
        self.actions = torch.empty(B, N, dtype=torch.int32, device=dev)
        # makespanPredict / flowtimePredict start at maxstep and maxstep * N (:220-221)
        self.makespan = torch.full((B,), self.maxstep, dtype=torch.int32, device=dev)
        self.flowtime = torch.full((B,), self.maxstep * N, dtype=torch.int32, device=dev)

    def gso(self, dtype=torch.float64):
        """getGSO(step): at the first call the radius grows until each instance's graph is connected and is kept."""
        if self.radii is None:
            self.radii, steps = batched_connect_radius(self.pos, self.comm_radius, return_steps=True)
            # once per episode (the reference grows the radius until the graph IS connected, new_simulator.py:759-768):
            # an instance still disconnected after the kernel's step limit must not be used silently
            if bool((steps < 0).any().item()):
                bad = torch.nonzero(steps < 0).flatten().tolist()
                raise nat.MagatNativeError("communication graph still disconnected after the radius growth limit in "
                                           "instances %s" % bad[:8])
        return batched_gso(self.pos, self.radii, symmetric_norm=self.symmetric_norm, dtype=dtype)

    def states(self):
        return batched_fov_states(self.map, self.pos, self.goal, self.FOV)

    def step(self, logits=None, actions=None, uniforms=None):
        """move(actionVec, currentstep) for every instance; returns `done` (B,) int32 = allReachGoal as the reference
        returns it (evaluated before the move).  No host synchronisation."""
        B, N, dev = self.B, self.N, self.dev
        assert (logits is None) != (actions is None), "give logits or actions"
        lg = None if logits is None else logits.reshape(B * N, 5).contiguous().float()
        ac = None if actions is None else _dev_i32(actions, "actions").reshape(B * N)
        policy = self.policy if lg is not None else 0
        if policy and uniforms is None:
            uniforms = torch.rand(B, N, dtype=torch.float64, device=dev, generator=self.generator)
        if uniforms is not None:
            if not uniforms.is_cuda or uniforms.dtype != torch.float64 or uniforms.numel() != B * N:
                raise nat.MagatNativeError("uniforms must be a (B,N) float64 device tensor")
            uniforms = uniforms.contiguous()
        d = nat.SimStepDesc()
        d.logits, d.actions_in, d.map = nat.ptr(lg), nat.ptr(ac), nat.ptr(self.map)
        d.map_batched = 1 if self.map.dim() == 3 else 0
        d.H, d.W, d.B, d.N = self.map.shape[-2], self.map.shape[-1], B, N
        d.policy, d.uniforms = policy, nat.ptr(uniforms)
        d.pos, d.goal = nat.ptr(self.pos), nat.ptr(self.goal)
        d.reach_goal, d.first_move, d.end_step = nat.ptr(self.reach_goal), nat.ptr(self.first_move), nat.ptr(self.end_step)
        d.currentstep, d.maxstep = self.currentstep, self.maxstep
        d.actions_out, d.moves_out, d.flags_out = nat.ptr(self.actions), None, nat.ptr(self.flags)
        d.done_out, d.flowtime_out, d.makespan_out = nat.ptr(self.done), nat.ptr(self.flowtime), nat.ptr(self.makespan)
        import ctypes
        with torch.cuda.device(dev):
            nat.check(nat.lib().magat_sim_step(ctypes.byref(d), nat.current_stream(dev)), "magat_sim_step")
        self.currentstep += 1
        return self.done
