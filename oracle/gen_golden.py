"""Generate tests/golden/*.npz by running the reference's OWN code on its CPU
path (NUMBA_ENABLE_CUDASIM=1, see oracle/cudasim_shim.py).

TEST INFRASTRUCTURE ONLY.  Run in this container (needs /root/reference):

    /opt/conda/bin/python3.9 oracle/gen_golden.py [--only NAME] [--out DIR]

Every fixture stores the exact inputs the reference saw (params, padded maps,
sampled grids, noise, u before the update) and what its kernels produced
(costs, weights, u after the update), so that the C restatement
(oracle/mppi_oracle.c) and the HIP path can be checked stage by stage
(parity level L1) and, through the xoroshiro128+
compatibility generator, end to end from the seed (level L3).

Reference entry points exercised (file:line in /root/reference/mppi_numba):
  mppi.py:186 solve -> 256-303 / 329-372 / 402-448 / 480-528
  mppi.py:534 shift_and_update, mppi.py:545 get_state_rollout
  terrain.py:380 set_TDM_from_PMF_grid, terrain.py:610 sample_grids
  barebone_mppi_numba.ipynb cells 2-3 (Config, MPPI_Numba without maps)
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cudasim_shim  # noqa: E402  (must precede reference imports)

import numpy as np  # noqa: E402
from numba.cuda.random import (  # noqa: E402
    create_xoroshiro128p_states,
    xoroshiro128p_normal_float32,
    xoroshiro128p_uniform_float32,
)

import mppi_numba.config as ref_config  # noqa: E402
from mppi_numba.config import Config  # noqa: E402
from mppi_numba.mppi import MPPI_Numba  # noqa: E402
from mppi_numba.terrain import TDM_Numba  # noqa: E402

OUT_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# --------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------
class _UpdateRecorder:
    """Stands in for planner.update_useq_numba: records what goes into and
    comes out of the reference's update kernel, then runs the real kernel.

    The simulator runs the 32 threads of update_useq_numba[1, 32] as Python
    threads.  The kernel reads `beta = weights_d[0]` (mppi.py:1150) and then
    thread 0 overwrites weights_d[0] (mppi.py:1153-1154) with no barrier in
    between -- harmless on a GPU, where the 32 threads are one warp in
    lockstep, but a real race under the simulator: threads that read beta late
    see exp(...) instead of the minimum cost and produce zero weights.  To get
    fixtures with the kernel's GPU semantics we (a) also launch the SAME kernel
    as [1, 1] on copies of the inputs (no race, deterministic order), and
    (b) re-launch [1, 32] from restored inputs until its weights agree with
    the serial launch to 1e-4 relative, i.e. until the race did not fire.
    """

    MAX_TRIES = 400

    def __init__(self, planner, real_kernel):
        self.planner = planner
        self.real = real_kernel
        self.records = []
        self.tries = []

    def __getitem__(self, launch_cfg):
        from numba import cuda

        def run(lam, costs_d, noise_d, weights_d, vrange_d, wrange_d, u_d):
            noise = noise_d.copy_to_host().copy()
            u_in = u_d.copy_to_host().copy()
            costs = costs_d.copy_to_host().copy()
            rec = dict(noise=noise, u_in=u_in, costs=costs)
            # (a) serial launch of the same kernel on copies
            c1, w1, u1 = cuda.to_device(costs.copy()), cuda.to_device(np.zeros_like(costs)), \
                cuda.to_device(u_in.copy())
            self.real[1, 1](lam, c1, noise_d, w1, vrange_d, wrange_d, u1)
            rec["serial_weights"] = w1.copy_to_host().copy()
            rec["serial_u_out"] = u1.copy_to_host().copy()
            # (b) the reference's own launch configuration, race-free run
            ok = False
            for attempt in range(self.MAX_TRIES):
                costs_d.copy_to_device(costs)
                u_d.copy_to_device(u_in)
                self.real[launch_cfg](lam, costs_d, noise_d, weights_d, vrange_d, wrange_d, u_d)
                w = weights_d.copy_to_host()
                if np.allclose(w, rec["serial_weights"], rtol=1e-4, atol=1e-37):
                    ok = True
                    break
            if not ok:
                raise RuntimeError("simulator race fired %d times in a row" % self.MAX_TRIES)
            self.tries.append(attempt + 1)
            rec["weights"] = weights_d.copy_to_host().copy()
            rec["u_out"] = u_d.copy_to_host().copy()
            self.records.append(rec)

        return run


def _flatten_records(records, prefix="it"):
    out = {}
    for k, rec in enumerate(records):
        for name, arr in rec.items():
            out["%s%d_%s" % (prefix, k, name)] = arr
    out["num_iterations"] = np.int64(len(records))
    return out


# tests/test_oracle_live.py regenerates a few fixtures with other worlds (different PMFs, masks,
# angular maps) to check the oracle beyond the committed ones; 0 = the committed fixtures
SEED_OFFSET = int(os.environ.get("GOLDEN_SEED_OFFSET", "0"))


def _random_pmf(rng, bins, rows, cols):
    """int8 PMF grid whose bins sum to 100 in every cell."""
    raw = rng.dirichlet(np.ones(bins), size=(rows, cols))  # (rows, cols, bins)
    pmf = np.floor(raw * 100).astype(np.int64)
    pmf[..., -1] += 100 - pmf.sum(axis=-1)
    return np.ascontiguousarray(np.moveaxis(pmf, -1, 0)).astype(np.int8)


def _params(x0, xgoal, dt, **over):
    p = dict(
        x0=np.asarray(x0, dtype=float),
        xgoal=np.asarray(xgoal, dtype=float),
        dt=dt,
        goal_tolerance=0.5,
        v_post_rollout=0.01,
        lambda_weight=1.0,
        cvar_alpha=1.0,
        num_opt=1,
        u_std=np.array([2.0, 3.0]),
        vrange=np.array([0.0, 3.0]),
        wrange=np.array([-np.pi, np.pi]),
    )
    p.update(over)
    return p


def _params_arrays(p):
    """Store the params dict as plain arrays (no pickles in the fixtures)."""
    out = {}
    for k, v in p.items():
        out["param_" + k] = np.asarray(v, dtype=np.float64)
    return out


def _tdm_arrays(tag, tdm):
    rp, cp = tdm.pmf_grid_d.shape[1:]
    out = {
        tag + "_pmf_grid_unpadded": np.asarray(tdm.pmf_grid),
        tag + "_pmf_grid_padded": tdm.pmf_grid_d.copy_to_host(),
        tag + "_bin_values": np.asarray(tdm.bin_values_d.copy_to_host()),
        tag + "_bin_values_bounds": np.asarray(tdm.bin_values_bounds_d.copy_to_host()),
        tag + "_obstacle_map_padded": tdm.obstacle_map_d.copy_to_host(),
        tag + "_unknown_map_padded": tdm.unknown_map_d.copy_to_host(),
        tag + "_padded_xlimits": np.asarray(tdm.padded_xlimits, dtype=np.float64),
        tag + "_padded_ylimits": np.asarray(tdm.padded_ylimits, dtype=np.float64),
        tag + "_pad_cells": np.int64(tdm.pad_cells),
        tag + "_res": np.float64(tdm.res),
        # outside [:, :rp, :cp] the simulator array is uninitialised
        tag + "_sample_grid": tdm.sample_grid_batch_d.copy_to_host()[:, :rp, :cp].copy(),
    }
    if tdm.risk_traction_map_d is not None:
        out[tag + "_risk_traction_map_padded"] = tdm.risk_traction_map_d.copy_to_host()
    return out


def _make_cfg(n_rollouts, **kw):
    cfg = Config(**kw)
    # config.py:72-79 clamps to [100, 15000]; fixtures use smaller N for speed
    cfg.num_control_rollouts = n_rollouts
    cfg.num_vis_state_rollouts = max(1, min(cfg.num_vis_state_rollouts, n_rollouts,
                                            cfg.num_grid_samples))
    return cfg


def _world(rng, bins=6, rows=14, cols=18, res=0.5):
    pmf = _random_pmf(rng, bins, rows, cols)
    obstacle = (rng.random((rows, cols)) < 0.08).astype(np.int8)
    unknown = (rng.random((rows, cols)) < 0.08).astype(np.int8)
    tdm_dict = dict(
        xlimits=(0.0, cols * res),
        ylimits=(0.0, rows * res),
        res=res,
        bin_values=np.linspace(0.0, 1.0, bins),
        bin_values_bounds=(0.0, 1.0),
        det_dynamics_cvar_alpha=0.4,
    )
    return pmf, obstacle, unknown, tdm_dict


def _world_arrays(pmf, obstacle, unknown, tdm_dict):
    out = dict(in_pmf_grid=pmf, in_obstacle_map=obstacle, in_unknown_map=unknown)
    for k, v in tdm_dict.items():
        out["tdm_" + k] = np.asarray(v, dtype=np.float64)
    return out


def _run_closed_loop(planner, lin, ang, params, n_solves, recorder, world_step=None):
    """solve / shift_and_update a few times the way test.ipynb:398-429 does,
    with a trivial 'world' (full traction) so that no extra reference classes
    are needed."""
    outs = {}
    x = np.asarray(params["x0"], dtype=float).copy()
    dt = params["dt"]
    for s in range(n_solves):
        first = len(recorder.records)
        useq = planner.solve()
        outs["solve%d_useq" % s] = useq.copy()
        outs["solve%d_first_iteration" % s] = np.int64(first)
        outs["solve%d_x0" % s] = x.copy()
        outs["solve%d_lin_sample_grid" % s] = _window(lin)
        outs["solve%d_ang_sample_grid" % s] = _window(ang)
        if s == 0:
            outs["state_rollout_after_solve0"] = planner.get_state_rollout().copy()
        u0 = useq[0]
        x = x + dt * np.array([u0[0] * np.cos(x[2]), u0[0] * np.sin(x[2]), u0[1]])
        planner.shift_and_update(x, useq, num_shifts=1)
    return outs


def _window(tdm):
    rp, cp = tdm.pmf_grid_d.shape[1:]
    return tdm.sample_grid_batch_d.copy_to_host()[:, :rp, :cp].copy()


def _cfg_arrays(cfg):
    return dict(
        cfg_T=np.float64(cfg.T), cfg_dt=np.float64(cfg.dt),
        cfg_num_steps=np.int64(cfg.num_steps),
        cfg_num_control_rollouts=np.int64(cfg.num_control_rollouts),
        cfg_num_grid_samples=np.int64(cfg.num_grid_samples),
        cfg_num_vis_state_rollouts=np.int64(cfg.num_vis_state_rollouts),
        cfg_max_map_dim=np.asarray(cfg.max_map_dim, dtype=np.int64),
        cfg_max_speed_padding=np.float64(cfg.max_speed_padding),
        cfg_tdm_sample_thread_dim=np.asarray(cfg.tdm_sample_thread_dim, dtype=np.int64),
        cfg_seed=np.int64(cfg.seed),
    )


# --------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------
def gen_rng():
    """Known-answer vectors for numba.cuda.random (xoroshiro128+, version
    0.54.1 as installed here; the reference leaves numba unpinned)."""
    n_streams, seed = 6, 1
    out = dict(seed=np.int64(seed))
    st = create_xoroshiro128p_states(n_streams, seed=seed)
    host = st.copy_to_host()
    out["initial_s0"] = host["s0"].astype(np.uint64)
    out["initial_s1"] = host["s1"].astype(np.uint64)
    normals = np.zeros((n_streams, 5), dtype=np.float64)
    for s in range(n_streams):
        for k in range(5):
            normals[s, k] = xoroshiro128p_normal_float32(st, s)
    out["normals_f64"] = normals  # the simulator returns python floats here
    st2 = create_xoroshiro128p_states(n_streams, seed=seed)
    uniforms = np.zeros((n_streams, 7), dtype=np.float32)
    for s in range(n_streams):
        for k in range(7):
            uniforms[s, k] = xoroshiro128p_uniform_float32(st2, s)
    out["uniforms_f32"] = uniforms
    st3 = create_xoroshiro128p_states(3, seed=12345)
    h3 = st3.copy_to_host()
    out["seed12345_s0"] = h3["s0"].astype(np.uint64)
    out["seed12345_s1"] = h3["s1"].astype(np.uint64)
    return out


def _det_like(mode_kw, n_rollouts=48, seed_world=3, alpha=0.4, num_opt=2,
              goal=(2.8, 4.2), n_solves=2, extra_params=None, res=0.5, dt=0.1, horizon=2.0,
              x0=(1.6, 2.1, 0.3), bounds=None, world_kw=None, max_map_dim=(30, 32)):
    rng = np.random.default_rng(seed_world + SEED_OFFSET)
    pmf, obstacle, unknown, tdm_dict = _world(rng, res=res, **(world_kw or {}))
    tdm_dict["det_dynamics_cvar_alpha"] = alpha
    if bounds is not None:
        tdm_dict["bin_values_bounds"] = bounds
    cfg = _make_cfg(n_rollouts, T=horizon, dt=dt, num_grid_samples=8,
                    max_speed_padding=3.0, tdm_sample_thread_dim=(4, 4),
                    num_vis_state_rollouts=5, max_map_dim=max_map_dim, seed=1, **mode_kw)
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    lin.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
    ang_pmf = _random_pmf(rng, pmf.shape[0], pmf.shape[1], pmf.shape[2])
    ang.set_TDM_from_PMF_grid(ang_pmf, tdm_dict, obstacle, unknown)
    planner = MPPI_Numba(cfg)
    params = _params(x0=x0, xgoal=goal, dt=cfg.dt, num_opt=num_opt,
                     dist_weight=1.5, obs_penalty=1e5, unknown_penalty=1e2)
    if extra_params:
        params.update(extra_params)
    planner.setup(params, lin, ang)
    rec = _UpdateRecorder(planner, MPPI_Numba.update_useq_numba)
    planner.update_useq_numba = rec
    out = {}
    out.update(_cfg_arrays(cfg))
    out.update(_world_arrays(pmf, obstacle, unknown, tdm_dict))
    out["in_ang_pmf_grid"] = ang_pmf
    out.update(_params_arrays(params))
    loop = _run_closed_loop(planner, lin, ang, params, n_solves, rec)
    out.update(loop)
    out.update(_tdm_arrays("lin", lin))
    out.update(_tdm_arrays("ang", ang))
    out.update(_flatten_records(rec.records))
    return out


def gen_det():
    return _det_like(dict(use_det_dynamics=True))


def gen_det_mean():
    # det_dynamics_cvar_alpha == 1.0 takes the separate branch terrain.py:426-433
    return _det_like(dict(use_det_dynamics=True), alpha=1.0, num_opt=1, n_solves=1,
                     goal=(7.5, 5.5), seed_world=5)


def gen_speedmap():
    # lambda_weight = 30 keeps the weights spread over many rollouts
    return _det_like(dict(use_nom_dynamics_with_speed_map=True), seed_world=7,
                     extra_params=dict(lambda_weight=30.0))


def gen_speedmap_mean():
    return _det_like(dict(use_nom_dynamics_with_speed_map=True), alpha=1.0,
                     num_opt=1, n_solves=1, seed_world=8)


def gen_speedmap_mean_bounds():
    # mean risk map (terrain.py:476-478 scales it as (100*(mean-lo))/range, the CVaR branch as
    # 100*((cvar-lo)/range)) with traction bounds away from (0, 1) on a map large enough for the
    # two operation orders to truncate differently in some cells
    return _det_like(dict(use_nom_dynamics_with_speed_map=True), alpha=1.0, num_opt=1, n_solves=1,
                     seed_world=9, bounds=(0.0, 1.7), world_kw=dict(rows=44, cols=52), max_map_dim=(60, 68),
                     x0=(9.6, 8.1, 0.3), goal=(14.0, 12.5))


# Units and ranges away from the notebooks' defaults: a resolution that is not a power of two
# (cell borders are not exact in float32), dt = 0.05, reverse driving, a heading of several
# turns, traction bounds above 1, small penalties and tolerances.
ODD = dict(res=0.3, dt=0.05, horizon=0.65, x0=(1.31, 1.97, 37.3), goal=(1.9, 2.6), bounds=(0.0, 1.2),
           extra_params=dict(vrange=np.array([-1.0, 2.5]), wrange=np.array([-2.0, 2.0]),
                             u_std=np.array([1.5, 2.5]), goal_tolerance=0.2, v_post_rollout=0.3,
                             dist_weight=0.3, obs_penalty=37.5, unknown_penalty=3.25, lambda_weight=4.0))


def gen_det_odd_units():
    return _det_like(dict(use_det_dynamics=True), n_rollouts=40, seed_world=31, alpha=0.7, num_opt=1,
                     n_solves=2, **ODD)


def gen_speedmap_odd_units():
    return _det_like(dict(use_nom_dynamics_with_speed_map=True), n_rollouts=40, seed_world=32, alpha=0.25,
                     num_opt=1, n_solves=1, **ODD)


def _tdm_like(n_rollouts, m_samples, cvar_alpha, alpha_dyn, num_opt=1, n_solves=2,
              seed_world=11, force_oversized=False, thread_dim=(4, 4), lambda_weight=1.0,
              goal=(2.6, 3.4), wall_ahead=False, res=0.5, dt=0.1, horizon=1.5, x0=(1.4, 1.7, 0.5),
              bounds=None, extra_params=None):
    saved = ref_config.max_threads_per_block
    if force_oversized:
        ref_config.max_threads_per_block = 4
    try:
        rng = np.random.default_rng(seed_world + SEED_OFFSET)
        pmf, obstacle, unknown, tdm_dict = _world(rng, bins=5, rows=10, cols=12, res=res)
        if bounds is not None:
            tdm_dict["bin_values_bounds"] = bounds
        if wall_ahead:
            # obstacle / unknown cells right in front of x0 = (1.4, 1.7, 0.5)
            obstacle[4, 4:6] = 1
            unknown[3, 4:6] = 1
        cfg = _make_cfg(n_rollouts, T=horizon, dt=dt, num_grid_samples=m_samples,
                        max_speed_padding=3.0, tdm_sample_thread_dim=thread_dim,
                        num_vis_state_rollouts=4, max_map_dim=(24, 26), seed=1,
                        use_tdm=True)
        lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
        lin.set_TDM_from_PMF_grid(pmf, tdm_dict, obstacle, unknown)
        ang_pmf = _random_pmf(rng, pmf.shape[0], pmf.shape[1], pmf.shape[2])
        ang.set_TDM_from_PMF_grid(ang_pmf, tdm_dict, obstacle, unknown)
        planner = MPPI_Numba(cfg)
        params = _params(x0=x0, xgoal=goal, dt=cfg.dt,
                         num_opt=num_opt, cvar_alpha=cvar_alpha, alpha_dyn=alpha_dyn,
                         dist_weight=1.0, lambda_weight=lambda_weight)
        if extra_params:
            params.update(extra_params)
        planner.setup(params, lin, ang)
        rec = _UpdateRecorder(planner, MPPI_Numba.update_useq_numba)
        planner.update_useq_numba = rec
        out = {}
        out.update(_cfg_arrays(cfg))
        out["cfg_max_threads_per_block"] = np.int64(cfg.max_threads_per_block)
        out.update(_world_arrays(pmf, obstacle, unknown, tdm_dict))
        out["in_ang_pmf_grid"] = ang_pmf
        out.update(_params_arrays(params))
        out.update(_run_closed_loop(planner, lin, ang, params, n_solves, rec))
        out.update(_tdm_arrays("lin", lin))
        out.update(_tdm_arrays("ang", ang))
        out.update(_flatten_records(rec.records))
        return out
    finally:
        ref_config.max_threads_per_block = saved


def gen_tdm_odd_units():
    odd = dict(ODD, x0=(1.11, 1.37, -21.9), goal=(1.6, 1.9))
    return _tdm_like(20, 6, cvar_alpha=0.34, alpha_dyn=0.8, num_opt=1, n_solves=2, seed_world=33, **odd)


def gen_tdm_cvar():
    return _tdm_like(24, 8, cvar_alpha=0.5, alpha_dyn=1.0, num_opt=2, wall_ahead=True)


def gen_tdm_mean_alpha_dyn():
    # cvar_alpha == 1 (mean over samples), tail-restricted sampling alpha_dyn < 1,
    # M not a power of two, thread tile that does not divide the grid
    return _tdm_like(20, 6, cvar_alpha=1.0, alpha_dyn=0.6, n_solves=1, thread_dim=(3, 5),
                     goal=(1.9, 2.3))


def gen_tdm_cvar_odd():
    # numel = ceil(M * float32(alpha)) with float32 rounding of alpha: 10*0.3f -> 4
    return _tdm_like(16, 10, cvar_alpha=0.3, alpha_dyn=1.0, n_solves=1, lambda_weight=12.0)


def gen_tdm_oversized_mean():
    # M > max_threads_per_block path (mppi.py:760-913); only cvar_alpha == 1 is
    # meaningful there (the 'sort' swaps unconditionally)
    return _tdm_like(12, 8, cvar_alpha=1.0, alpha_dyn=1.0, n_solves=1, force_oversized=True)


def gen_barebone(with_obstacles):
    ns = cudasim_shim.load_barebone_namespace()
    BConfig, BPlanner = ns["Config"], ns["MPPI_Numba"]
    cfg = BConfig(T=3.0, dt=0.1, num_control_rollouts=100, num_vis_state_rollouts=6, seed=1)
    cfg.num_control_rollouts = 64  # BASELINE config 1: N=64, T=30, below the clamp
    cfg.num_vis_state_rollouts = 6
    params = dict(
        dt=cfg.dt, x0=np.array([0, 0, np.pi / 4]),
        xgoal=np.array([2.0, 1.5]) if with_obstacles else np.array([0.9, 0.7]),
        goal_tolerance=0.5, dist_weight=10, lambda_weight=1.0, num_opt=1,
        u_std=np.array([1.0, 1.0]), vrange=np.array([0.0, 2.0]),
        wrange=np.array([-np.pi, np.pi]),
    )
    if with_obstacles:
        params["obstacle_positions"] = np.array([[1.2, 1.4], [0.6, 0.1]])
        params["obstacle_radius"] = np.array([0.5, 0.3])
        params["obs_penalty"] = 1e6
    planner = BPlanner(cfg)
    planner.setup(params)
    rec = _UpdateRecorder(planner, BPlanner.update_useq_numba)
    planner.update_useq_numba = rec
    out = dict(cfg_T=np.float64(cfg.T), cfg_dt=np.float64(cfg.dt),
               cfg_num_steps=np.int64(cfg.num_steps),
               cfg_num_control_rollouts=np.int64(cfg.num_control_rollouts),
               cfg_num_vis_state_rollouts=np.int64(cfg.num_vis_state_rollouts),
               cfg_seed=np.int64(cfg.seed))
    out.update(_params_arrays(params))
    x = params["x0"].astype(float).copy()
    for s in range(2):
        useq = planner.solve()
        out["solve%d_useq" % s] = useq.copy()
        out["solve%d_x0" % s] = x.copy()
        if s == 0:
            out["state_rollout_after_solve0"] = planner.get_state_rollout().copy()
        u0 = useq[0]
        x = x + cfg.dt * np.array([u0[0] * np.cos(x[2]), u0[0] * np.sin(x[2]), u0[1]])
        planner.shift_and_update(x, useq, num_shifts=1)
    out.update(_flatten_records(rec.records))
    return out


def gen_semantic(mode_kw, alpha, tag_seed):
    """set_TDM_from_semantic_grid (terrain.py:183-342): a 9x11 grid of three terrain types
    with made-up (values, pmf) pairs; plain strings stand in for the Terrain objects (the
    TDM only uses them as dictionary keys)."""
    rng = np.random.default_rng(tag_seed)
    rows, cols, res, bins = 9, 11, 1.0, 7
    sg = rng.integers(0, 3, size=(rows, cols))
    values = np.array([0.0, 0.1, 0.3, 0.35, 0.7, 0.9, 1.0])
    id2name = {0: "dirt", 1: "grass", 2: "mud"}
    name2terrain = {k: "TERRAIN_" + k for k in id2name.values()}
    terrain2pmf = {}
    pmfs = {}
    for name, conc in (("dirt", [1, 1, 2, 4, 8, 8, 3]), ("grass", [1, 3, 5, 5, 3, 2, 1]), ("mud", [6, 6, 4, 2, 1, 1, 1])):
        pmf = rng.dirichlet(np.asarray(conc, dtype=float))
        terrain2pmf[name2terrain[name]] = (values, pmf)
        pmfs[name] = pmf
    obstacle = (rng.random((rows, cols)) < 0.1).astype(np.int8)
    unknown = (rng.random((rows, cols)) < 0.1).astype(np.int8)
    use_tdm = bool(mode_kw.get("use_tdm"))
    cfg = _make_cfg(24, T=1.0, dt=0.1, num_grid_samples=6 if use_tdm else 4, max_speed_padding=5.0,
                    tdm_sample_thread_dim=(4, 4), num_vis_state_rollouts=3, max_map_dim=(14, 16), seed=1,
                    **mode_kw)
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    for tdm in (lin, ang):
        tdm.set_TDM_from_semantic_grid(sg, res, bins, values, np.array([0.0, 1.0]), (0.0, cols * res),
                                       (0.0, rows * res), id2name, name2terrain, terrain2pmf,
                                       det_dynamics_cvar_alpha=alpha, obstacle_map=obstacle, unknown_map=unknown)
    planner = MPPI_Numba(cfg)
    params = _params(x0=(3.2, 2.6, 0.4), xgoal=(8.5, 6.5), dt=cfg.dt, num_opt=1, cvar_alpha=0.5,
                     alpha_dyn=1.0, dist_weight=1.0)
    planner.setup(params, lin, ang)
    rec = _UpdateRecorder(planner, MPPI_Numba.update_useq_numba)
    planner.update_useq_numba = rec
    out = {}
    out.update(_cfg_arrays(cfg))
    out.update(dict(in_semantic_grid=sg, in_values=values, in_obstacle_map=obstacle, in_unknown_map=unknown,
                    in_pmf_dirt=pmfs["dirt"], in_pmf_grass=pmfs["grass"], in_pmf_mud=pmfs["mud"],
                    in_alpha=np.float64(-1.0 if alpha is None else alpha), in_res=np.float64(res)))
    out.update(_params_arrays(params))
    out.update(_run_closed_loop(planner, lin, ang, params, 1, rec))
    out.update(_tdm_arrays("lin", lin))
    out["lin_semantic_grid_after"] = np.asarray(lin.semantic_grid)
    out.update(_flatten_records(rec.records))
    return out


FIXTURES = {
    "rng_xoroshiro": gen_rng,
    "det_cvar": gen_det,
    "det_mean": gen_det_mean,
    "speedmap_cvar": gen_speedmap,
    "speedmap_mean": gen_speedmap_mean,
    "speedmap_mean_bounds": gen_speedmap_mean_bounds,
    "det_odd_units": gen_det_odd_units,
    "speedmap_odd_units": gen_speedmap_odd_units,
    "tdm_cvar": gen_tdm_cvar,
    "tdm_odd_units": gen_tdm_odd_units,
    "tdm_mean_alpha_dyn": gen_tdm_mean_alpha_dyn,
    "tdm_cvar_odd": gen_tdm_cvar_odd,
    "tdm_oversized_mean": gen_tdm_oversized_mean,
    "semantic_tdm": lambda: gen_semantic(dict(use_tdm=True), None, 21),
    "semantic_det": lambda: gen_semantic(dict(use_det_dynamics=True), 0.3, 22),
    "semantic_det_mean": lambda: gen_semantic(dict(use_det_dynamics=True), 1.0, 23),
    "semantic_speedmap": lambda: gen_semantic(dict(use_nom_dynamics_with_speed_map=True), 0.3, 24),
    "barebone_flat": lambda: gen_barebone(False),
    "barebone_obstacles": lambda: gen_barebone(True),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", action="append", default=None)
    ap.add_argument("--out", default=OUT_DIR)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    names = args.only or list(FIXTURES)
    for name in names:
        t0 = time.time()
        data = FIXTURES[name]()
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, **data)
        print("[gen_golden] %-24s %3d arrays  %6.1f s  -> %s"
              % (name, len(data), time.time() - t0, os.path.relpath(path)))


if __name__ == "__main__":
    main()
