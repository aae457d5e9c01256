"""numpy model of the time-parallel rollout (csrc/rollout_scan_kernel.h: k_rollout_scan).

TEST INFRASTRUCTURE: a statement of the ALGORITHM the HIP kernel implements, checked on the CPU
against the oracle (tests/test_scan_model.py) so that the event logic -- goal break, rollouts
frozen in a zero-traction cell, the vote on the constant-traction assumption, chunk hand-offs,
the in-order cost accumulation -- is pinned before it meets the GPU.  The kernel follows this
file operation for operation, except for the hardware sin / cos / sqrt (here: numpy's, rounded to
float32).

Under the assumption that every visited cell carries the traction of the start cell
(vtr0, wtr0), rollout_det_dyn_numba (mppi.py:916-1009) is two prefix sums over the horizon:
    theta_t = theta_0 + wtr0*dt * sum_{k<t} w_k
    x_t     = x_0 + vtr0*dt * sum_{k<t} v_k cos(theta_k)        (y likewise)
The horizon is cut into chunks of `ch` steps, one lane per (rollout, chunk).
"""
import numpy as np

f32 = np.float32


def _cell_index(pos, lo, res, n):
    q = np.floor_divide(f32(pos - lo), f32(res))  # float32 floor division of the reference (mppi.py:971)
    return np.clip(q.astype(np.int64), 0, n - 1)


def frozen_block(acc, k64, pen, count):
    """`count` further steps of a rollout that stands still: each adds the same stage cost k64
    (float64) and penalty pen to the float32 cost, rounded after every addition as the reference
    does (mppi.py:994-998).  While the cost stays inside one binade the additions are exact
    multiples of its ulp -- acc + m * round_to_ulp(k64) -- so whole runs of steps are taken at
    once and only the steps that cross into the next binade are added one by one."""
    acc = acc.astype(f32).copy()
    count = count.astype(np.int64).copy()
    k64 = k64.astype(np.float64)
    pen = pen.astype(f32)
    slow = (pen != 0) & (count > 0)          # (a zero-traction cell that is also an obstacle: step by step)
    while slow.any():
        acc = np.where(slow, ((acc.astype(np.float64) + k64).astype(f32) + pen).astype(f32), acc)
        count = np.where(slow, count - 1, count)
        slow &= count > 0
    while (count > 0).any():
        go = count > 0
        acc = np.where(go, (acc.astype(np.float64) + k64).astype(f32), acc)     # one exact step
        count = np.where(go, count - 1, count)
        go = count > 0
        a64 = np.abs(acc.astype(np.float64))
        e = np.floor(np.log2(np.maximum(a64, 1e-300)))
        ulp = np.exp2(e - 23)
        top = np.exp2(e + 1)
        q = np.rint(k64 / ulp) * ulp
        ok = go & (q > 0) & (acc > 0)
        m = np.where(ok, np.floor((top - a64) / np.where(q > 0, q, 1.0)), 0).astype(np.int64)
        m = np.where(a64 + m * q >= top, m - 1, m)
        m = np.clip(np.minimum(m, count), 0, None)
        acc = np.where(ok, (acc.astype(np.float64) + m * q).astype(f32), acc)
        count = np.where(ok, count - m, count)
    return acc


def scan_rollout(p, lin_grid, ang_grid, obs, unk, noise, u, ch=8):
    """p: oracle.OracleParams.  Returns (costs float32 [N], failed bool [N/64 tiles])."""
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
    oc, uc = f32(p.obs_cost), f32(p.unknown_cost)
    dw = f32(p.dist_weight)
    vden = np.float64(f32(p.v_post_rollout)) + 1e-6
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
    k0 = f32(np.float64(f32(p.lambda_weight)) / np.float64(f32(p.u_std[0])) ** 2)
    k1 = f32(np.float64(f32(p.lambda_weight)) / np.float64(f32(p.u_std[1])) ** 2)
    cc = ((k0 * uu[:, 0])[None] * e[:, :, 0] + ((k1 * uu[:, 1])[None] * e[:, :, 1]).astype(f32)).astype(f32)  # [N, Tp]
    # ---- phase B: heading in TURNS: float32 prefix inside the chunk, float64 across chunks
    kturn = f32(np.float64(wtr0) * np.float64(dt) * 0.15915494309189535)
    dturn = (kturn * w).astype(f32).reshape(N, W, ch)
    lt = np.cumsum(dturn, axis=2, dtype=f32)                                   # inclusive, per chunk
    base_t = np.float64(th0) * 0.15915494309189535 + np.concatenate(
        [np.zeros((N, 1)), np.cumsum(lt[:, :, -1].astype(np.float64), axis=1)[:, :-1]], axis=1)
    base_fr = (base_t - np.floor(base_t)).astype(f32)[:, :, None]              # v_fract_f64
    fr = (base_fr + np.concatenate([np.zeros((N, W, 1), dtype=f32), lt[:, :, :-1]], axis=2)).astype(f32)
    c = np.cos(2.0 * np.pi * fr.astype(np.float64)).astype(f32)              # v_cos_f32 / v_sin_f32 take turns
    s = np.sin(2.0 * np.pi * fr.astype(np.float64)).astype(f32)
    # ---- phase C: position = float32 prefix inside the chunk, float64 across chunks
    kv = f32(f32(vtr0) * dt)
    q = (kv * v).astype(f32).reshape(N, W, ch)
    dx, dy = (q * c).astype(f32), (q * s).astype(f32)
    lx, ly = np.cumsum(dx, axis=2, dtype=f32), np.cumsum(dy, axis=2, dtype=f32)   # inclusive, per chunk
    bx = np.float64(x0) + np.concatenate([np.zeros((N, 1)), np.cumsum(lx[:, :, -1].astype(np.float64), axis=1)[:, :-1]], axis=1)
    by = np.float64(y0) + np.concatenate([np.zeros((N, 1)), np.cumsum(ly[:, :, -1].astype(np.float64), axis=1)[:, :-1]], axis=1)
    bxf, byf = bx.astype(f32)[:, :, None], by.astype(f32)[:, :, None]
    x_post, y_post = (bxf + lx).astype(f32), (byf + ly).astype(f32)
    x_pre = np.concatenate([np.broadcast_to(bxf, (N, W, 1)), x_post[:, :, :-1]], axis=2)
    y_pre = np.concatenate([np.broadcast_to(byf, (N, W, 1)), y_post[:, :, :-1]], axis=2)
    # ---- phase D: lookups at the pre-step position, stage costs (float32)
    xi, yi = _cell_index(x_pre, xlo, res, cols), _cell_index(y_pre, ylo, res, rows)
    cl, ca = lin[yi, xi].astype(np.int64), ang[yi, xi].astype(np.int64)
    pen = (obs[yi, xi].astype(f32) * oc + unk[yi, xi].astype(f32) * uc).astype(f32)
    zero = (cl == zero_byte) if zero_byte is not None else np.zeros_like(cl, dtype=bool)
    mismatch = (cl != ref_lin) | (ca != ref_ang)
    ddx, ddy = (xg - x_post).astype(f32), (yg - y_post).astype(f32)
    n2 = (ddx * ddx + ddy * ddy).astype(f32)
    sg = (dt + dw * np.sqrt(n2)).astype(f32)
    valid = (np.arange(Tp) < T).reshape(1, W, ch)
    hitb = (n2 <= gt2) & valid
    zerob = zero & valid
    # ---- per chunk: the first step that freezes (s) and the first goal hit before it (h)
    idx = np.arange(ch)[None, None, :]
    s_idx = np.where(zerob, idx, ch).min(axis=2)                                # [N, W]
    h_idx = np.where(hitb & (idx < s_idx[:, :, None]), idx, ch).min(axis=2)
    n_valid = np.clip(T - np.arange(W) * ch, 0, ch)[None, :]
    froze = (h_idx == ch) & (s_idx < n_valid)
    n_act = np.where(h_idx < ch, h_idx + 1, np.minimum(s_idx, n_valid))         # steps the chunk itself adds
    bad = (mismatch & (idx < n_act[:, :, None])).any(axis=2)
    # a rollout frozen at step s stands at the pre-step position of s for the rest of the horizon:
    # what it pays per step, in float64 as the reference's CPU path computes it
    sc = np.minimum(s_idx, ch - 1)
    take = lambda a: np.take_along_axis(a, sc[:, :, None], axis=2)[:, :, 0]
    fx, fy, f_pen = take(x_pre), take(y_pre), take(pen)
    fdx, fdy = (xg - fx).astype(f32).astype(np.float64), (yg - fy).astype(f32).astype(np.float64)
    f_d2 = fdx * fdx + fdy * fdy
    f_k = np.float64(dt) + np.float64(p.dist_weight) * np.sqrt(f_d2)
    f_hit = f_d2 <= np.float64(gt2)
    event = np.where(h_idx < ch, 1, np.where(froze, 2, 0))                      # 0 none, 1 goal reached, 2 frozen
    # ---- across chunks: the first chunk with an event ends the rollout; later chunks add nothing
    first = np.full(N, W, dtype=np.int64)
    for cidx in range(W - 1, -1, -1):
        first = np.where(event[:, cidx] != 0, cidx, first)
    has = first < W
    fc = np.minimum(first, W - 1)
    rows_n = np.arange(N)
    dead = has[:, None] & (np.arange(W)[None] > first[:, None])
    n_act = np.where(dead, 0, n_act)
    failed_lane = (bad & ~dead).any(axis=1)
    kind = np.where(has, event[rows_n, fc], 0)
    fz_count = np.where(kind == 2, np.where(f_hit[rows_n, fc], 1, T - (fc * ch + s_idx[rows_n, fc])), 0)
    reached = (kind == 1) | ((kind == 2) & f_hit[rows_n, fc])
    last_j = (T - 1) - (W - 1) * ch
    d2_final = np.where(kind == 2, f_d2[rows_n, fc], n2[:, W - 1, last_j].astype(np.float64))
    term = np.where(reached, 0.0, np.sqrt(d2_final) / vden)
    # ---- the accumulation in the reference's order, float32-rounded after every addition
    #      (mppi.py:994-1009): ONE wave walks the steps
    acc = np.zeros(N, dtype=f32)
    for t in range(T):
        cidx, j = divmod(t, ch)
        on = j < n_act[:, cidx]
        acc = ((acc + np.where(on, sg[:, cidx, j], f32(0))).astype(f32) + np.where(on, pen[:, cidx, j], f32(0))).astype(f32)
    acc = frozen_block(acc, f_k[rows_n, fc], f_pen[rows_n, fc], fz_count)
    acc = (acc.astype(np.float64) + term).astype(f32)
    for t in range(T):
        acc = (acc + cc[:, t]).astype(f32)
    tiles = -(-N // 64)
    failed = np.array([failed_lane[k * 64:(k + 1) * 64].any() for k in range(tiles)])
    return acc, failed
