"""CPU restatement of the simulator front-end that feeds the hot path.
TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

  gso_from_positions  - multiRobotSimNew.computeAdjacencyMatrix, fixed-radius branch
                        (utils/new_simulator.py:745-806; get_maxEigenValue :808-818)
  fov_states          - AgentState.toInputTensor with guidance 'Project_G'
                        (dataloader/statetransformer_Guidance.py:185-239, projectedgoal :103-124,
                        setPosAgents :88-101, setmap :63-66)

Pinned against outputs of the reference classes themselves: oracle/make_golden_sim.py -> tests/golden/sim_*.npz.
Plain numpy loops, small cases only.
"""
import numpy as np

ZERO_TOLERANCE = 1e-9


def gso_from_positions(pos, comm_radius, symmetric_norm=False):
    """pos (N,2) agent coordinates -> (N,N) float64 GSO:  W = (euclidean distance < R), zero diagonal; if W has any
    edge: optional D^-1/2 W D^-1/2, then W / lambda_max(W); an edgeless W stays zero (new_simulator.py:783-804)."""
    pos = np.asarray(pos, np.float64)
    N = pos.shape[0]
    d = np.sqrt(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1))       # squareform(pdist(., 'euclidean'))
    W = (d < comm_radius).astype(np.float64)
    W[np.arange(N), np.arange(N)] = 0.0
    if not W.any():
        return W
    if symmetric_norm:
        deg = W.sum(axis=1)
        zero = np.abs(deg) < ZERO_TOLERANCE
        deg[zero] = 1.0
        inv = np.sqrt(1.0 / deg)
        inv[zero] = 0.0
        W = inv[:, None] * W * inv[None, :]
    lam = np.max(np.linalg.eigvalsh(W))                                   # W is symmetric -> eigvalsh branch (:810-812)
    return W / lam


def projected_goal(fov, cx, cy, gx, gy):
    """(row, col) of the goal marker on the (FOV+2)^2 tensor when the goal lies outside the FOV
    (statetransformer_Guidance.py:103-124)."""
    W = fov + 2
    dist = W // 2
    dy, dx = float(gy - cy), float(gx - cx)
    angle = np.arctan2(dy, dx)
    if (np.pi / 4 <= angle <= np.pi * 3 / 4) or (-np.pi * (3 / 4) <= angle <= -np.pi / 4):
        col = int(dist * (np.sign(dy) + 1))
        row = int(dist + np.round(dist * dx / np.abs(dy)))
    else:
        row = int(dist * (np.sign(dx) + 1))
        col = int(dist + np.round(dist * dy / np.abs(dx)))
    return row, col


def fov_states(obstacle_map, pos, goal, fov=9):
    """obstacle_map (H,W) {0,1}; pos, goal (N,2) integer (row, col) -> (N,3,fov+2,fov+2) uint8:
    channel 0 obstacles (outside the map = obstacle), 1 goal or projected goal, 2 agents (incl. self);
    1-pixel zero border around the fov x fov window except for a projected goal, which lands on it."""
    obstacle_map = np.asarray(obstacle_map)
    H, Wm = obstacle_map.shape
    N = len(pos)
    half, Wt = fov // 2, fov + 2
    agents = np.zeros((H, Wm), np.int64)
    for n in range(N):
        agents[int(pos[n][0]), int(pos[n][1])] = 1
    out = np.zeros((N, 3, Wt, Wt), np.uint8)
    for n in range(N):
        cx, cy = int(pos[n][0]), int(pos[n][1])
        gx, gy = int(goal[n][0]), int(goal[n][1])
        goal_in = False
        for a in range(fov):
            for b_ in range(fov):
                x, y = cx - half + a, cy - half + b_
                inside = 0 <= x < H and 0 <= y < Wm
                out[n, 0, a + 1, b_ + 1] = obstacle_map[x, y] if inside else 1
                out[n, 2, a + 1, b_ + 1] = agents[x, y] if inside else 0
                if inside and x == gx and y == gy:
                    out[n, 1, a + 1, b_ + 1] = 1
                    goal_in = True
        if not goal_in:
            r, c = projected_goal(fov, cx, cy, gx, gy)
            out[n, 1, r, c] = 1
    return out
