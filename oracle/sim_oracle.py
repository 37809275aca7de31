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


# ------------------------------------------------------------------ action decode + shielding
# move vectors by action key (utils/new_simulator.py:56-65): up 0, left 1, down 2, right 3, stop 4
MOVES = np.array([[-1, 0], [0, -1], [1, 0], [0, 1], [0, 0]], np.int64)


def decode_actions(logits):
    """convectToActionKey_softmax (new_simulator.py:863-869): argmax of softmax(logits) = argmax of the logits
    (first maximum wins, like torch.max).  logits (N,5) -> (N,) int64 action keys."""
    return np.argmax(np.asarray(logits), axis=1).astype(np.int64)


def shield_moves(obstacle_map, pos, move):
    """multiRobotSimNew.check_collision (new_simulator.py:334-454) with ONE documented deviation: where the reference
    lets `random.choice` pick the agent that may enter a cell claimed by several MOVING agents (:416), the lowest agent
    index wins here (a batched kernel needs a deterministic rule; a stationary claimant always wins, as in the reference).
    Everything else follows the reference step by step: out-of-arena moves stop (:354-357), face-to-face swaps stop both
    (:361-375), moves into obstacles stop (:392-403), cell conflicts (:407-423), then the backward cascade: whoever
    moves into the cell of an agent that was forced to stay stops as well (:424-446).
    obstacle_map (H,W); pos (N,2) int; move (N,2) int in {-1,0,1}.  Returns (new_move (N,2), flags dict)."""
    obstacle_map = np.asarray(obstacle_map)
    H, W = obstacle_map.shape
    pos = np.asarray(pos, np.int64)
    move = np.array(move, np.int64)
    N = len(pos)
    new = pos + move
    out = (new[:, 0] >= H) | (new[:, 1] >= W) | (new[:, 0] < 0) | (new[:, 1] < 0)
    move[out] = 0
    # face-to-face: same half-step position
    half = {}
    swap = []
    newh = pos * 2 + move                       # doubled coordinates: exact half steps
    for i in range(N):
        key = (int(newh[i, 0]), int(newh[i, 1]))
        if key in half:
            j = half[key]
            swap += [i, j]
            move[i] = 0
            move[j] = 0
        half[key] = i
    need_reverse = []
    claims = {}
    wall = []
    for i in range(N):
        tgt = pos[i] + move[i]
        if obstacle_map[tgt[0], tgt[1]] != 0:
            wall.append(i)
            move[i] = 0
            need_reverse.append(tuple(int(v) for v in pos[i]))
            tgt = pos[i]
        claims.setdefault((int(tgt[0]), int(tgt[1])), []).append(i)
    collide = []
    for cell, lst in claims.items():
        if len(lst) > 1:
            selected = min(lst)                                   # deviation: lowest index instead of random.choice
            for i in lst:
                if (move[i] == 0).all():
                    selected = i
            for i in lst:
                if i != selected:
                    move[i] = 0
                    need_reverse.append(tuple(int(v) for v in pos[i]))
            collide += lst
    new = pos + move
    into = {}
    for i in range(N):
        into.setdefault((int(new[i, 0]), int(new[i, 1])), []).append((i, (int(pos[i, 0]), int(pos[i, 1]))))
    while need_reverse:
        p = need_reverse.pop(0)
        for agent, cur in into.get(p, []):
            if cur != p:
                need_reverse.append(cur)
            move[agent] = 0
    return move, dict(out_boundary=out, swap=sorted(set(swap)), wall=wall, collide=sorted(set(collide)))


def move_step(obstacle_map, pos, goal, logits):
    """One simulator step for one instance: multiRobotSimNew.move (new_simulator.py:471-520) without the bookkeeping of
    makespan / flowtime: decode, propose, shield, advance, reach-goal test.  Returns (new_pos, actions, reached)."""
    actions = decode_actions(logits)
    new_move, _ = shield_moves(obstacle_map, pos, MOVES[actions])
    new_pos = np.asarray(pos, np.int64) + new_move
    reached = np.abs(new_pos - np.asarray(goal, np.int64)).sum(axis=1) == 0
    return new_pos, actions, reached


# ------------------------------------------------------------------ policies, episode
# bookkeeping, step-0 communication radius
def sample_actions(logits, policy, uniforms):
    """The three action policies (new_simulator.py:863-883).  policy 0: convectToActionKey_softmax (argmax); 1:
    convectToActionKey_sum_multinorm, one draw with weights normalize(x) = x / sum(x) (float32, :857-861); 2:
    convectToActionKey_exp_multinorm, weights exp(x) (float32).  The draw of torch.multinomial is pinned to the inverse
    CDF with the given uniforms: first k with  w_0 + .. + w_k > u * sum(w), float64 running sum over the float32 weights
    (the reference run with torch.multinomial patched to this rule: oracle/make_golden_sim_episode.py)."""
    logits = np.asarray(logits, np.float32)
    if policy == 0:
        return decode_actions(logits)
    keys = np.zeros(len(logits), np.int64)
    for i, l in enumerate(logits):
        if policy == 1:
            s = np.float32(0)
            for q in range(5):
                s = np.float32(s + l[q])
            w = (l / s).astype(np.float32)
        else:
            w = np.exp(l.astype(np.float64)).astype(np.float32)
        c = np.cumsum(w.astype(np.float64))
        keys[i] = int(np.nonzero(c > uniforms[i] * c[-1])[0][0])
    return keys


class EpisodeState:
    """The per-case state multiRobotSimNew keeps across move() calls (new_simulator.py:191-221)."""

    def __init__(self, obstacle_map, pos, goal, maxstep):
        n = len(pos)
        self.map = np.asarray(obstacle_map)
        self.pos = np.array(pos, np.int64)
        self.goal = np.asarray(goal, np.int64)
        self.maxstep = int(maxstep)
        self.reach_goal = np.zeros(n, np.uint8)
        self.first_move = np.zeros(n, np.int64)
        self.end_step = np.zeros(n, np.int64)
        self.flowtime = self.maxstep * n
        self.makespan = self.maxstep


def episode_step(st, logits, currentstep, policy=0, uniforms=None):
    """multiRobotSimNew.move (new_simulator.py:471-549), complete: returns (allReachGoal, check_predictCollsion, keys)."""
    n = len(st.pos)
    all_reached = int(np.count_nonzero(st.reach_goal)) == n
    predict_collision = False
    keys = None
    if (not all_reached) and currentstep < st.maxstep:
        keys = sample_actions(logits, policy, uniforms)
        st.first_move[(keys != 4) & (st.first_move == 0)] = currentstep
        new_move, fl = shield_moves(st.map, st.pos, MOVES[keys])
        predict_collision = bool(fl["out_boundary"].any() or fl["wall"] or fl["swap"] or fl["collide"])
        st.pos = st.pos + new_move
        at = np.abs(st.pos - st.goal).sum(axis=1) == 0
        st.reach_goal[at] = 1
        st.end_step[at & (st.end_step == 0)] = currentstep
    if all_reached or currentstep >= st.maxstep:
        st.end_step[st.end_step == 0] = currentstep - 1
        st.flowtime = int(np.sum(st.end_step - st.first_move + 1))
        st.makespan = int(st.end_step.max() - st.first_move.min() + 1)
    return all_reached, predict_collision, keys


def is_connected(W):
    """graphTools.isConnected (utils/graphUtils/graphTools.py:562-589) decides by the multiplicity of the Laplacian's zero
    eigenvalue (== 1); for an undirected graph that is reachability of every node from node 0, tested here exactly."""
    n = len(W)
    seen = np.zeros(n, bool)
    seen[0] = True
    front = [0]
    while front:
        i = front.pop()
        for j in np.nonzero(W[i])[0]:
            if not seen[j]:
                seen[j] = True
                front.append(int(j))
    return bool(seen.all())


def connect_radius(pos, comm_radius):
    """Step-0 branch of computeAdjacencyMatrix (new_simulator.py:759-768): r = R / 1.1; repeat r *= 1.1 until the graph
    (distance < r, zero diagonal) is connected.  Returns (r, growth steps)."""
    pos = np.asarray(pos, np.float64)
    d = np.sqrt(((pos[:, None, :] - pos[None, :, :]) ** 2).sum(-1))
    r = comm_radius / 1.1
    steps = 0
    while True:
        r = r * 1.1
        steps += 1
        W = (d < r).astype(np.float64)
        np.fill_diagonal(W, 0.0)
        if is_connected(W):
            return r, steps
