"""numpy model of the time-parallel rollout (csrc/rollout_scan_kernel.h: k_rollout_scan).

TEST INFRASTRUCTURE: a statement of the ALGORITHM the HIP kernel implements, checked on the CPU
against the oracle (tests/test_scan_model.py) so that the event logic -- goal break, rollouts
frozen in a zero-traction cell, the vote on the constant-traction assumption, chunk hand-offs --
is pinned before it meets the GPU.  The kernel follows this file operation for operation, except
for the hardware sin / cos / sqrt (here: numpy's, rounded to float32).

Under the assumption that every visited cell carries the traction of the start cell
(vtr0, wtr0), rollout_det_dyn_numba (mppi.py:916-1009) is two prefix sums over the horizon:
    theta_t = theta_0 + wtr0*dt * sum_{k<t} w_k
    x_t     = x_0 + vtr0*dt * sum_{k<t} v_k cos(theta_k)        (y likewise)
The horizon is cut into chunks of CH steps, one wave per (tile of 64 rollouts, chunk).
"""
import numpy as np

f32 = np.float32


def _cell_index(pos, lo, res, n):
    q = np.floor_divide(f32(pos - lo), f32(res))  # float32 floor division of the reference (mppi.py:971)
    return np.clip(q.astype(np.int64), 0, n - 1)


def scan_rollout(p, lin_grid, ang_grid, obs, unk, noise, u, ch=8, chain64=True):
    """p: oracle.OracleParams.  Returns (costs float32 [N], failed bool [N/64 tiles]).
    chain64: the stage addends and their additions in float64, rounded to float32 after every
    step as the reference's CPU path does (mppi.py:994 under the simulator); False: float32
    addends, float32 additions."""
    real = np.float64 if chain64 else f32
    noise = np.asarray(noise, dtype=f32)
    u = np.asarray(u, dtype=f32)
    N, T = noise.shape[:2]
    lin, ang = np.asarray(lin_grid)[0], np.asarray(ang_grid)[0]
    rows, cols = obs.shape
    W = -(-T // ch)
    Tp = W * ch
    dt, res = f32(p.dt), f32(p.res)
    xlo, ylo = f32(p.xlo), f32(p.ylo)
    x0, y0, th0 = (f32(v) for v in p.x0)
    xg, yg = (f32(v) for v in p.xgoal)
    gt2 = f32(p.goal_tolerance) * f32(p.goal_tolerance)
    lam = f32(p.lambda_weight)
    oc, uc = f32(p.obs_cost), f32(p.unknown_cost)
    dw = f32(p.dist_weight)
    vden = f32(np.float64(f32(p.v_post_rollout)) + 1e-6)
    # the assumption: traction bytes of the start cell
    xi0, yi0 = _cell_index(x0, xlo, res, cols), _cell_index(y0, ylo, res, rows)
    ref_lin, ref_ang = int(lin[yi0, xi0]), int(ang[yi0, xi0])
    vtr0 = p.lin_lo + p.lin_ratio * ref_lin
    wtr0 = p.ang_lo + p.ang_ratio * ref_ang
    zero_byte = None
    for b in range(128):
        if p.lin_lo + p.lin_ratio * b == 0.0:
            zero_byte = b
            break
    # ---- phase A: controls, control-cost terms (padded steps: zero noise, zero controls)
    e = np.zeros((N, Tp, 2), dtype=f32)
    e[:, :T] = noise
    uu = np.zeros((Tp, 2), dtype=f32)
    uu[:T] = u
    v = np.clip(uu[None, :, 0] + e[:, :, 0], f32(p.vrange[0]), f32(p.vrange[1])).astype(f32)
    w = np.clip(uu[None, :, 1] + e[:, :, 1], f32(p.wrange[0]), f32(p.wrange[1])).astype(f32)
    s0sq = f32(np.float64(f32(p.u_std[0])) ** 2)
    s1sq = f32(np.float64(f32(p.u_std[1])) ** 2)
    r0, r1 = (uu[:, 0] / s0sq).astype(f32), (uu[:, 1] / s1sq).astype(f32)
    cc = (lam * (r0[None] * e[:, :, 0] + r1[None] * e[:, :, 1])).astype(f32)      # [N, Tp]
    # ---- phase B: heading = float64 prefix sum
    kth = np.float64(wtr0) * np.float64(dt)
    dth = kth * w.astype(np.float64)
    th_pre = np.float64(th0) + np.concatenate([np.zeros((N, 1)), np.cumsum(dth, axis=1)[:, :-1]], axis=1)
    turns = th_pre * (1.0 / (2.0 * np.pi))
    frac = (turns - np.floor(turns)).astype(f32)  # v_fract_f64, then v_sin_f32 / v_cos_f32 take turns
    c = np.cos(2.0 * np.pi * frac.astype(np.float64)).astype(f32)
    s = np.sin(2.0 * np.pi * frac.astype(np.float64)).astype(f32)
    # ---- phase C: position = float32 prefix inside the chunk, float64 across chunks
    q = (dt * v).astype(f32)
    vtr0f = f32(vtr0)
    dx = (vtr0f * (q * c).astype(f32)).astype(f32).reshape(N, W, ch)
    dy = (vtr0f * (q * s).astype(f32)).astype(f32).reshape(N, W, ch)
    lx, ly = np.cumsum(dx, axis=2, dtype=f32), np.cumsum(dy, axis=2, dtype=f32)   # inclusive, per chunk
    bx = np.float64(x0) + np.concatenate([np.zeros((N, 1)), np.cumsum(lx[:, :, -1].astype(np.float64), axis=1)[:, :-1]], axis=1)
    by = np.float64(y0) + np.concatenate([np.zeros((N, 1)), np.cumsum(ly[:, :, -1].astype(np.float64), axis=1)[:, :-1]], axis=1)
    bxf, byf = bx.astype(f32)[:, :, None], by.astype(f32)[:, :, None]
    x_post, y_post = (bxf + lx).astype(f32), (byf + ly).astype(f32)
    x_pre = np.concatenate([np.broadcast_to(bxf, (N, W, 1)), x_post[:, :, :-1]], axis=2)
    y_pre = np.concatenate([np.broadcast_to(byf, (N, W, 1)), y_post[:, :, :-1]], axis=2)
    # ---- phase D: lookups at the pre-step position, stage costs
    xi, yi = _cell_index(x_pre, xlo, res, cols), _cell_index(y_pre, ylo, res, rows)
    cl, ca = lin[yi, xi].astype(np.int64), ang[yi, xi].astype(np.int64)
    pen = (obs[yi, xi].astype(f32) * oc + unk[yi, xi].astype(f32) * uc).astype(f32)
    zero = (cl == zero_byte) if zero_byte is not None else np.zeros_like(cl, dtype=bool)
    mismatch = (cl != ref_lin) | (ca != ref_ang)

    def d2(x, y):
        ddx, ddy = (xg - x).astype(f32).astype(real), (yg - y).astype(f32).astype(real)
        return (ddx * ddx + ddy * ddy).astype(real)

    n2 = d2(x_post, y_post)
    n2_pre = np.concatenate([d2(x_pre[:, :, :1], y_pre[:, :, :1]), n2[:, :, :-1]], axis=2)
    root, root_pre = np.sqrt(n2), np.sqrt(n2_pre)
    # ---- per chunk: walk its CH steps until the first event; what each step ADDS to the cost
    #      (stage addend, penalty addend), assuming the rollout is alive when the chunk starts
    sg = (real(dt) + real(dw) * root).astype(real)
    sg_frozen = (real(dt) + real(dw) * root_pre).astype(real)   # a rollout frozen at the pre-step position
    add_sg = np.zeros((N, W, ch), dtype=real)
    add_pen = np.zeros((N, W, ch), dtype=f32)
    event = np.zeros((N, W), dtype=np.int64)   # 0 none, 1 goal reached, 2 frozen (not at the goal)
    bad = np.zeros((N, W), dtype=bool)
    frozen_sg = np.zeros((N, W), dtype=real)
    frozen_pen = np.zeros((N, W), dtype=f32)
    n2_end = np.full((N, W), 1e9, dtype=real)   # squared goal distance where the chunk leaves the rollout
    for cidx in range(W):
        alive = np.ones(N, dtype=bool)
        frozen = np.zeros(N, dtype=bool)
        for j in range(ch):
            t = cidx * ch + j
            if t >= T:
                break
            z = alive & ~frozen & zero[:, cidx, j]
            frozen_sg[:, cidx] = np.where(z, sg_frozen[:, cidx, j], frozen_sg[:, cidx])
            frozen_pen[:, cidx] = np.where(z, pen[:, cidx, j], frozen_pen[:, cidx])
            n2_end[:, cidx] = np.where(z, n2_pre[:, cidx, j], n2_end[:, cidx])
            frozen |= z
            bad[:, cidx] |= alive & ~frozen & mismatch[:, cidx, j]
            add_sg[:, cidx, j] = np.where(alive, np.where(frozen, frozen_sg[:, cidx], sg[:, cidx, j]), 0)
            add_pen[:, cidx, j] = np.where(alive, np.where(frozen, frozen_pen[:, cidx], pen[:, cidx, j]), 0)
            n2_now = np.where(frozen, n2_end[:, cidx], n2[:, cidx, j])
            n2_end[:, cidx] = np.where(alive, n2_now, n2_end[:, cidx])
            hit = alive & (n2_now <= gt2)
            event[:, cidx] = np.where(hit, 1, event[:, cidx])
            alive &= ~hit
        event[:, cidx] = np.where((event[:, cidx] == 0) & frozen, 2, event[:, cidx])
    # ---- across chunks: the first chunk with an event decides what the later ones add
    #      (goal reached: nothing; frozen: the frozen addends, every step to the end of the horizon)
    first = np.full(N, W, dtype=np.int64)
    for cidx in range(W - 1, -1, -1):
        first = np.where(event[:, cidx] != 0, cidx, first)
    rows_n = np.arange(N)
    has = first < W
    fc = np.minimum(first, W - 1)
    kind = np.where(has, event[rows_n, fc], 0)
    for cidx in range(W):
        later = has & (cidx > first)
        valid = (cidx * ch + np.arange(ch)) < T
        fs = np.where(kind == 2, frozen_sg[rows_n, fc], real(0))
        fp = np.where(kind == 2, frozen_pen[rows_n, fc], f32(0))
        add_sg[:, cidx] = np.where(later[:, None], fs[:, None] * valid[None], add_sg[:, cidx])
        add_pen[:, cidx] = np.where(later[:, None], fp[:, None] * valid[None], add_pen[:, cidx])
    failed_lane = np.zeros(N, dtype=bool)
    for cidx in range(W):
        failed_lane |= bad[:, cidx] & ~(has & (cidx > first))
    reached = kind == 1
    n2_final = np.where(has, n2_end[rows_n, fc], n2_end[:, W - 1])
    term = np.where(reached, 0.0, np.sqrt(n2_final.astype(np.float64)) / (np.float64(f32(p.v_post_rollout)) + 1e-6))
    # ---- the accumulation in the reference's order, float32-rounded (mppi.py:994-1009): the sums the
    #      chunks produce side by side cannot reproduce T sequential roundings, so ONE wave walks them
    acc = np.zeros(N, dtype=f32)
    for t in range(T):
        cidx, j = divmod(t, ch)
        acc = ((acc.astype(real) + add_sg[:, cidx, j]).astype(f32) + add_pen[:, cidx, j]).astype(f32)
    acc = (acc.astype(np.float64) + term).astype(f32)
    for t in range(T):
        acc = (acc + cc[:, t]).astype(f32)
    costs = acc
    tiles = -(-N // 64)
    failed = np.array([failed_lane[k * 64:(k + 1) * 64].any() for k in range(tiles)])
    return costs, failed
