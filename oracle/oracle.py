"""ctypes front-end of oracle/liboracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "mppi_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


class OracleParams(C.Structure):
    _fields_ = [
        ("res", C.c_float), ("xlo", C.c_float), ("ylo", C.c_float),
        ("vrange", C.c_float * 2), ("wrange", C.c_float * 2), ("xgoal", C.c_float * 2),
        ("x0", C.c_float * 3), ("u_std", C.c_float * 2),
        ("dt", C.c_float), ("goal_tolerance", C.c_float), ("v_post_rollout", C.c_float),
        ("lambda_weight", C.c_float), ("cvar_alpha", C.c_float),
        ("obs_cost", C.c_float), ("unknown_cost", C.c_float), ("_pad", C.c_float),
        ("dist_weight", C.c_double),
        ("lin_lo", C.c_double), ("lin_ratio", C.c_double),
        ("ang_lo", C.c_double), ("ang_ratio", C.c_double),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        assert _lib.oracle_params_size() == C.sizeof(OracleParams)
        _lib.oracle_xoroshiro_normal.restype = C.c_double
        _lib.oracle_xoroshiro_uniform.restype = C.c_float
    return _lib


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i8(a):
    return np.ascontiguousarray(a, dtype=np.int8)


def traction_scale(bounds):
    """(lo, ratio) as mppi.py:674-675 evaluates them: lo = bounds[0],
    ratio = 0.01*(bounds[1]-bounds[0]) with the subtraction in the array's own
    dtype and the product in float64."""
    b = np.asarray(bounds)
    if b.dtype not in (np.float32, np.float64):
        b = b.astype(np.float64)
    return float(b[0]), float(0.01 * float(b[1] - b[0]))


def bin_table(bin_values, bounds):
    """np.int8(100.*(bin_values[b]-bounds[0])/(bounds[1]-bounds[0])) per bin, in
    the dtypes held on the device (terrain.py:689); truncating cast."""
    bv = np.asarray(bin_values)
    bd = np.asarray(bounds)
    rng_ = bd[1] - bd[0]
    out = np.zeros(len(bv), dtype=np.int8)
    for b in range(len(bv)):
        out[b] = np.int8(int(100.0 * float(bv[b] - bd[0]) / float(rng_)))
    return out


DEFAULT_UNKNOWN_COST = 1e2  # mppi.py:32
DEFAULT_OBS_COST = 1e5      # mppi.py:33
DEFAULT_DIST_WEIGHT = 1.0   # mppi.py:36


def make_params(params, res, padded_xlimits, padded_ylimits, lin_bounds, ang_bounds,
                default_obs_cost=DEFAULT_OBS_COST, default_dist_weight=DEFAULT_DIST_WEIGHT):
    """Pack a reference-style params dict the way
    move_mppi_task_vars_to_device (mppi.py:214-234) does."""
    p = OracleParams()
    p.res = np.float32(res)
    p.xlo = np.float32(np.asarray(padded_xlimits, dtype=np.float64).astype(np.float32)[0])
    p.ylo = np.float32(np.asarray(padded_ylimits, dtype=np.float64).astype(np.float32)[0])
    for name in ("vrange", "wrange", "xgoal", "x0", "u_std"):
        arr = np.asarray(params[name], dtype=np.float64).astype(np.float32)
        for i in range(len(arr)):
            getattr(p, name)[i] = arr[i]
    p.dt = np.float32(params["dt"])
    p.goal_tolerance = np.float32(params["goal_tolerance"])
    p.v_post_rollout = np.float32(params.get("v_post_rollout", 0.0))
    p.lambda_weight = np.float32(params["lambda_weight"])
    p.cvar_alpha = np.float32(params.get("cvar_alpha", 1.0))
    p.obs_cost = np.float32(params.get("obs_penalty", default_obs_cost))
    p.unknown_cost = np.float32(params.get("unknown_penalty", DEFAULT_UNKNOWN_COST))
    p.dist_weight = float(params.get("dist_weight", default_dist_weight))
    p.lin_lo, p.lin_ratio = traction_scale(lin_bounds)
    p.ang_lo, p.ang_ratio = traction_scale(ang_bounds)
    return p


def rollout_det(p, lin_grid, ang_grid, obs, unk, noise, u, risk=None):
    """lin_grid/ang_grid: (G, rows, stride) int8 (sample 0 is used)."""
    lin_grid, ang_grid, obs, unk = _i8(lin_grid), _i8(ang_grid), _i8(obs), _i8(unk)
    noise, u = _f32(noise), _f32(u)
    n, t = noise.shape[0], noise.shape[1]
    rp, cp = obs.shape
    costs = np.zeros(n, dtype=np.float32)
    riskp = None
    if risk is not None:
        risk = _i8(risk).reshape(rp, cp)
        riskp = _p(risk, C.c_int8)
    lib().oracle_rollout_det(
        C.byref(p), _p(lin_grid, C.c_int8), _p(ang_grid, C.c_int8),
        C.c_int(lin_grid.shape[-2]), C.c_int(lin_grid.shape[-1]),
        _p(obs, C.c_int8), _p(unk, C.c_int8), riskp, C.c_int(rp), C.c_int(cp),
        _p(noise, C.c_float), _p(u, C.c_float), C.c_int(n), C.c_int(t), _p(costs, C.c_float))
    return costs


def rollout_tdm(p, lin_grid, ang_grid, obs, unk, noise, u, want_per_sample=False):
    lin_grid, ang_grid, obs, unk = _i8(lin_grid), _i8(ang_grid), _i8(obs), _i8(unk)
    noise, u = _f32(noise), _f32(u)
    n, t = noise.shape[0], noise.shape[1]
    m = lin_grid.shape[0]
    rp, cp = obs.shape
    costs = np.zeros(n, dtype=np.float32)
    per = np.zeros((n, m), dtype=np.float32) if want_per_sample else None
    lib().oracle_rollout_tdm(
        C.byref(p), _p(lin_grid, C.c_int8), _p(ang_grid, C.c_int8), C.c_int(m),
        C.c_int(lin_grid.shape[1]), C.c_int(lin_grid.shape[2]),
        _p(obs, C.c_int8), _p(unk, C.c_int8), C.c_int(rp), C.c_int(cp),
        _p(noise, C.c_float), _p(u, C.c_float), C.c_int(n), C.c_int(t),
        _p(costs, C.c_float), _p(per, C.c_float) if per is not None else None)
    return (costs, per) if want_per_sample else costs


def rollout_barebone(p, obs_pos, obs_r, noise, u):
    obs_pos, obs_r = _f32(obs_pos).reshape(-1, 2), _f32(obs_r).reshape(-1)
    noise, u = _f32(noise), _f32(u)
    n, t = noise.shape[0], noise.shape[1]
    costs = np.zeros(n, dtype=np.float32)
    lib().oracle_rollout_barebone(
        C.byref(p), _p(obs_pos, C.c_float), _p(obs_r, C.c_float), C.c_int(len(obs_r)),
        _p(noise, C.c_float), _p(u, C.c_float), C.c_int(n), C.c_int(t), _p(costs, C.c_float))
    return costs


def update_useq(lambda_weight, costs, noise, vrange, wrange, u, num_threads=32):
    """Returns (weights, u_out, clobbered_costs)."""
    costs = _f32(costs).copy()
    noise = _f32(noise)
    u = _f32(u).copy()
    n, t = noise.shape[0], noise.shape[1]
    weights = np.zeros(n, dtype=np.float32)
    vr = np.asarray(vrange, dtype=np.float64).astype(np.float32)
    wr = np.asarray(wrange, dtype=np.float64).astype(np.float32)
    lib().oracle_update_useq(
        C.c_float(np.float32(lambda_weight)), _p(costs, C.c_float), _p(noise, C.c_float),
        _p(weights, C.c_float), _p(vr, C.c_float), _p(wr, C.c_float), _p(u, C.c_float),
        C.c_int(n), C.c_int(t), C.c_int(num_threads))
    return weights, u, costs


def xoroshiro_init(n, seed):
    states = np.zeros((n, 2), dtype=np.uint64)
    lib().oracle_xoroshiro_init(_p(states, C.c_uint64), C.c_long(n), C.c_uint64(seed))
    return states


def xoroshiro_normal(states, index):
    return float(lib().oracle_xoroshiro_normal(_p(states, C.c_uint64), C.c_long(index)))


def xoroshiro_uniform(states, index):
    return np.float32(lib().oracle_xoroshiro_uniform(_p(states, C.c_uint64), C.c_long(index)))


def sample_noise(states, u_std, n, t):
    """Advances `states` in place (4 draws per stream)."""
    us = np.asarray(u_std, dtype=np.float64).astype(np.float32)
    noise = np.zeros((n, t, 2), dtype=np.float32)
    lib().oracle_sample_noise(_p(states, C.c_uint64), _p(us, C.c_float), C.c_int(n), C.c_int(t),
                              _p(noise, C.c_float))
    return noise


def sample_grids(pmf_padded, states, n_grids, thread_dim, table, alpha_dyn, out):
    """Writes out[:, :Rp, :Cp] in place; advances `states`."""
    pmf = _i8(pmf_padded)
    b, rp, cp = pmf.shape
    assert out.dtype == np.int8 and out.flags.c_contiguous and out.shape[0] == n_grids
    table = _i8(table)
    lib().oracle_sample_grids(
        _p(pmf, C.c_int8), C.c_int(b), C.c_int(rp), C.c_int(cp), _p(states, C.c_uint64),
        C.c_int(n_grids), C.c_int(thread_dim[0]), C.c_int(thread_dim[1]), _p(table, C.c_int8),
        C.c_double(alpha_dyn), _p(out, C.c_int8), C.c_int(out.shape[1]), C.c_int(out.shape[2]))
    return out


def state_rollout_noise(p, lin_grid, ang_grid, noise, u_prev, u_cur, n_vis):
    lin_grid, ang_grid = _i8(lin_grid), _i8(ang_grid)
    noise, u_prev, u_cur = _f32(noise), _f32(u_prev), _f32(u_cur)
    t = u_cur.shape[0]
    out = np.zeros((n_vis, t + 1, 3), dtype=np.float32)
    lib().oracle_state_rollout_noise(
        C.byref(p), _p(lin_grid, C.c_int8), _p(ang_grid, C.c_int8),
        C.c_int(lin_grid.shape[-2]), C.c_int(lin_grid.shape[-1]), _p(noise, C.c_float),
        _p(u_prev, C.c_float), _p(u_cur, C.c_float), C.c_int(n_vis), C.c_int(t),
        _p(out, C.c_float))
    return out


def state_rollout_envs(p, lin_grid, ang_grid, u_cur, n_vis):
    lin_grid, ang_grid = _i8(lin_grid), _i8(ang_grid)
    u_cur = _f32(u_cur)
    t = u_cur.shape[0]
    out = np.zeros((n_vis, t + 1, 3), dtype=np.float32)
    lib().oracle_state_rollout_envs(
        C.byref(p), _p(lin_grid, C.c_int8), _p(ang_grid, C.c_int8),
        C.c_int(lin_grid.shape[1]), C.c_int(lin_grid.shape[2]), _p(u_cur, C.c_float),
        C.c_int(n_vis), C.c_int(t), _p(out, C.c_float))
    return out


def state_rollout_barebone(p, noise, u_prev, u_cur, n_vis):
    noise, u_prev, u_cur = _f32(noise), _f32(u_prev), _f32(u_cur)
    t = u_cur.shape[0]
    out = np.zeros((n_vis, t + 1, 3), dtype=np.float32)
    lib().oracle_state_rollout_barebone(
        C.byref(p), _p(noise, C.c_float), _p(u_prev, C.c_float), _p(u_cur, C.c_float),
        C.c_int(n_vis), C.c_int(t), _p(out, C.c_float))
    return out
