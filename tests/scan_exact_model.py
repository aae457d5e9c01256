"""numpy model of the time-parallel rollout with the reference's rounding points
(csrc/rollout_scan_exact_kernel.h: k_rollout_scan_exact).

TEST INFRASTRUCTURE: a statement of the ALGORITHM the HIP kernel implements -- which quantities are
walked (heading, position, cost: float32-rounded after every addition) and which are computed for
all steps side by side, how goal breaks and stops in zero-traction cells are resolved per chunk of
4 steps (first event wins), how a stopped rollout keeps paying through ordinary records, where the
terminal cost comes from, the vote on the constant-traction assumption -- checked on the CPU against
the oracle (tests/test_scan_exact_model.py).  Elementary operations follow the oracle's float64
expressions (oracle/mppi_oracle.c: unicycle_step, dist2_to_goal, control_cost_term, terminal_cost).
"""
import numpy as np

from scan_model import _cell_index

f32 = np.float32
f64 = np.float64
CHL = 4  # steps per lane = steps per chunk; a chunk wave owns two chunks (8 steps)


def scan_exact_rollout(p, lin_grid, ang_grid, obs, unk, noise, u):
    """p: oracle.OracleParams.  Returns (costs float32 [N], failed bool [tiles of 32 rollouts])."""
    noise = np.asarray(noise, dtype=f32)
    u = np.asarray(u, dtype=f32)
    N, T = noise.shape[:2]
    lin, ang = np.asarray(lin_grid)[0], np.asarray(ang_grid)[0]
    rows, cols = obs.shape
    W = -(-T // 8)
    Tp, K = 8 * W, 2 * W
    dt, res = f32(p.dt), f32(p.res)
    xlo, ylo = f32(p.xlo), f32(p.ylo)
    x0, y0, th0 = (f32(v) for v in p.x0)
    xg, yg = (f32(v) for v in p.xgoal)
    gt2 = f64(f32(p.goal_tolerance) * f32(p.goal_tolerance))
    oc, uc = f32(p.obs_cost), f32(p.unknown_cost)
    vden = f64(f32(p.v_post_rollout)) + 1e-6
    xi0, yi0 = _cell_index(x0, xlo, res, cols), _cell_index(y0, ylo, res, rows)
    ref_lin, ref_ang = int(lin[yi0, xi0]), int(ang[yi0, xi0])
    vtr0 = p.lin_lo + p.lin_ratio * ref_lin          # the assumption: the start cell's traction everywhere
    wtr0 = p.ang_lo + p.ang_ratio * ref_ang
    zero_byte = next((b for b in range(128) if p.lin_lo + p.lin_ratio * b == 0.0), None)

    # ---- A: controls of all steps side by side (steps past the horizon: zero noise, zero controls)
    e = np.zeros((N, Tp, 2), dtype=f32)
    e[:, :T] = noise
    uu = np.zeros((Tp, 2), dtype=f32)
    uu[:T] = u
    v = np.clip(uu[None, :, 0] + e[:, :, 0], f32(p.vrange[0]), f32(p.vrange[1])).astype(f32)
    w = np.clip(uu[None, :, 1] + e[:, :, 1], f32(p.wrange[0]), f32(p.wrange[1])).astype(f32)
    # ---- the heading walk: row t = the heading before step t
    th = np.empty((N, Tp + 1), dtype=f32)
    th[:, 0] = th0
    for t in range(Tp):
        th[:, t + 1] = (th[:, t].astype(f64) + f64(dt) * wtr0 * w[:, t].astype(f64)).astype(f32)
    # ---- B: sin / cos of every (already rounded) heading, side by side; then the position walks
    cs, sn = np.cos(th[:, :Tp].astype(f64)), np.sin(th[:, :Tp].astype(f64))
    x = np.empty((N, Tp + 1), dtype=f32)
    y = np.empty((N, Tp + 1), dtype=f32)
    x[:, 0], y[:, 0] = x0, y0
    for t in range(Tp):
        x[:, t + 1] = (x[:, t].astype(f64) + f64(dt) * vtr0 * v[:, t].astype(f64) * cs[:, t]).astype(f32)
        y[:, t + 1] = (y[:, t].astype(f64) + f64(dt) * vtr0 * v[:, t].astype(f64) * sn[:, t]).astype(f32)
    # ---- C: lookups at the position a step STARTS in, distances after it, stage costs; all steps side by side
    xi, yi = _cell_index(x[:, :Tp], xlo, res, cols), _cell_index(y[:, :Tp], ylo, res, rows)
    cl, ca = lin[yi, xi].astype(np.int64), ang[yi, xi].astype(np.int64)
    po = np.where(obs[yi, xi] != 0, oc, f32(0)).astype(f32)
    pu = np.where(unk[yi, xi] != 0, uc, f32(0)).astype(f32)
    zero = (cl == zero_byte) if zero_byte is not None else np.zeros_like(cl, dtype=bool)
    mism = (cl != ref_lin) | (ca != ref_ang)

    def d2_of(px, py):
        dx, dy = (xg - px).astype(f32).astype(f64), (yg - py).astype(f32).astype(f64)
        return dx * dx + dy * dy

    n2 = d2_of(x[:, 1:], y[:, 1:])                              # [N, Tp]: after step t
    sg = f64(dt) + f64(p.dist_weight) * np.sqrt(n2)
    valid = (np.arange(Tp) < T)[None, :]
    # ---- per chunk of 4 steps: the first step that starts in a zero-traction cell (s), the first goal hit before it
    ch = lambda a: a.reshape(N, K, CHL)
    idx = np.arange(CHL)[None, None, :]
    zb, hb, mm = ch(zero & valid), ch((n2 <= gt2) & valid), ch(mism)
    s_idx = np.where(zb, idx, CHL).min(axis=2)
    h_idx = np.where(hb & (idx < s_idx[:, :, None]), idx, CHL).min(axis=2)
    n_valid = np.clip(T - np.arange(K) * CHL, 0, CHL)[None, :]
    is_hit = h_idx < CHL
    froze = ~is_hit & (s_idx < n_valid)
    n_act = np.where(is_hit, h_idx + 1, np.minimum(s_idx, n_valid))
    bad = (mm & (idx < n_act[:, :, None])).any(axis=2)
    ev = np.where(is_hit, 1, np.where(froze, 2, 0))
    # what a rollout that stops in this chunk goes on paying (its slot): from the place where it stands
    sc = np.minimum(s_idx, CHL - 1)
    take = lambda a: np.take_along_axis(ch(a), sc[:, :, None], axis=2)[:, :, 0]
    f_d2 = d2_of(take(x[:, :Tp]), take(y[:, :Tp]))
    f_k = f64(dt) + f64(p.dist_weight) * np.sqrt(f_d2)
    f_po, f_pu, f_hit = take(po), take(pu), f_d2 <= gt2
    # ---- the first event of a rollout wins; later chunks are dead
    first = np.where(ev != 0, np.arange(K)[None, :], K).min(axis=1)     # [N]
    has = first < K
    fc = np.minimum(first, K - 1)
    rn = np.arange(N)
    dead = has[:, None] & (np.arange(K)[None, :] > first[:, None])
    n_act = np.where(dead, 0, n_act)
    failed_lane = (bad & ~dead).any(axis=1)
    stopped = has & (ev[rn, fc] == 2)
    f_begin = np.where(stopped, fc * CHL + s_idx[rn, fc], 0)
    f_end = np.where(stopped, np.where(f_hit[rn, fc], f_begin + 1, T), 0)
    # terminal cost: zero after a goal hit (or a stop inside the goal circle), from where a stopped rollout
    # stands, else from the position after the last step
    term = np.where(has, np.where(stopped & ~f_hit[rn, fc], np.sqrt(f_d2[rn, fc]) / vden, 0.0),
                    np.sqrt(d2_of(x[:, T], y[:, T])) / vden)
    # ---- the records: a chunk's own steps, then -- a stopped rollout -- the place where it stands
    t_all = np.arange(Tp)[None, :]
    own = (idx < n_act[:, :, None]).reshape(N, Tp)
    standing = (t_all >= f_begin[:, None]) & (t_all < f_end[:, None])
    r_sg = np.where(own, sg, np.where(standing, f_k[rn, fc][:, None], 0.0))
    r_po = np.where(own, po, np.where(standing, f_po[rn, fc][:, None], f32(0))).astype(f32)
    r_pu = np.where(own, pu, np.where(standing, f_pu[rn, fc][:, None], f32(0))).astype(f32)
    # ---- the cost walk (mppi.py:994-998 per step), the terminal cost, the T control-cost terms (1005-1009)
    cost = np.zeros(N, dtype=f32)
    for t in range(Tp):
        cost = (cost.astype(f64) + r_sg[:, t]).astype(f32)
        cost = (cost + r_po[:, t]).astype(f32)
        cost = (cost + r_pu[:, t]).astype(f32)
    cost = (cost.astype(f64) + term).astype(f32)
    s0 = f64(f32(p.u_std[0])) * f64(f32(p.u_std[0]))
    s1 = f64(f32(p.u_std[1])) * f64(f32(p.u_std[1]))
    for t in range(T):
        cc = f64(f32(p.lambda_weight)) * ((f64(u[t, 0]) / s0) * noise[:, t, 0].astype(f64) +
                                           (f64(u[t, 1]) / s1) * noise[:, t, 1].astype(f64))
        cost = (cost.astype(f64) + cc).astype(f32)
    tiles = -(-N // 32)
    failed = np.array([failed_lane[k * 32:(k + 1) * 32].any() for k in range(tiles)])
    return cost, failed
