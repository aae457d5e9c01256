"""Randomised configurations of the hot path against the oracle: every planner mode with
random map sizes, resolutions (power-of-two and not), time steps, horizons, control
ranges (reverse driving, large headings), sample counts, CVaR levels and goals.  Costs must
be bit-identical to the CPU restatement of the reference except for the rare rollout whose
float64 trig / sqrt differs in the last bit (bounded here), the update within 1e-5."""
import numpy as np
import pytest

import bench
from helpers import ulp_diff_f32
from oracle import oracle as O

pytestmark = pytest.mark.gpu

MODES = [dict(use_det_dynamics=True), dict(use_nom_dynamics_with_speed_map=True), dict(use_tdm=True)]


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    mode = MODES[seed % 3]
    rows, cols = int(rng.integers(12, 90)), int(rng.integers(12, 90))
    res = float(rng.choice([0.1, 0.2, 0.25, 0.3, 0.5, 1.0]))
    dt = float(rng.choice([0.05, 0.1, 0.2]))
    t_steps = int(rng.integers(1, 70))
    n = int(rng.integers(1, 400))
    m = int(rng.choice([1, 2, 7, 33, 64, 70])) if "use_tdm" in mode else 1
    bins = int(rng.integers(2, 12))
    raw = rng.dirichlet(np.ones(bins) * rng.uniform(0.3, 2.0), size=(rows, cols))
    p = np.floor(raw * 100).astype(np.int64)
    p[..., -1] += 100 - p.sum(axis=-1)
    pmf = np.ascontiguousarray(np.moveaxis(p, -1, 0)).astype(np.int8)
    density = rng.uniform(0.0, 0.15)
    obstacle = (rng.random((rows, cols)) < density).astype(np.int8)
    unknown = (rng.random((rows, cols)) < density).astype(np.int8)
    x_lo, y_lo = float(rng.uniform(-20, 20)), float(rng.uniform(-20, 20))
    td = dict(xlimits=(x_lo, x_lo + cols * res), ylimits=(y_lo, y_lo + rows * res), res=res,
              bin_values=np.linspace(0.0, float(rng.uniform(0.5, 1.0)), bins),
              bin_values_bounds=(0.0, float(rng.choice([1.0, 1.2]))),
              det_dynamics_cvar_alpha=float(rng.choice([1.0, 0.9, 0.5, 0.1])))
    v_hi = float(rng.uniform(0.5, 4.0))
    v_lo = float(rng.choice([0.0, -1.0]))
    w_hi = float(rng.uniform(0.5, 4.0))
    x0 = np.array([rng.uniform(x_lo + res, x_lo + cols * res - res), rng.uniform(y_lo + res, y_lo + rows * res - res),
                   rng.choice([rng.uniform(-3.2, 3.2), rng.uniform(-40, 40)])])
    goal = x0[:2] + rng.uniform(-1, 1, 2) * rng.choice([0.3, 3.0, 30.0])
    params = dict(x0=x0, xgoal=goal, dt=dt, goal_tolerance=float(rng.uniform(0.05, 1.0)),
                  v_post_rollout=float(rng.uniform(0.01, 1.0)), lambda_weight=float(rng.uniform(0.2, 20.0)),
                  cvar_alpha=float(rng.choice([1.0, 0.8, 0.3, 0.05])), alpha_dyn=float(rng.choice([1.0, 0.6])),
                  num_opt=1, u_std=np.array([rng.uniform(0.2, 3.0), rng.uniform(0.2, 3.0)]),
                  vrange=np.array([v_lo, v_hi]), wrange=np.array([-w_hi, w_hi]),
                  dist_weight=float(rng.choice([1.0, 0.3, 5.0])), obs_penalty=float(rng.choice([1e5, 37.5])),
                  unknown_penalty=float(rng.choice([1e2, 3.25])))
    pad_speed = float(rng.uniform(1.0, 6.0))
    return mode, rows, cols, res, dt, t_steps, n, m, pmf, obstacle, unknown, td, params, pad_speed


def _seeds():
    """60 fixed cases; MPPI_FUZZ_RANGE="a:b" swaps in other seeds for a soak run."""
    import os
    span = os.environ.get("MPPI_FUZZ_RANGE")
    if span:
        lo, hi = (int(v) for v in span.split(":"))
        return range(lo, hi)
    return range(60)


@pytest.mark.parametrize("seed", _seeds())
def test_random_configuration_matches_oracle(seed):
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    mode, rows, cols, res, dt, t_steps, n, m, pmf, obstacle, unknown, td, params, pad_speed = random_case(seed)
    pad = int(np.ceil(pad_speed * dt / res))
    cfg = Config(T=(t_steps + 0.5) * dt, dt=dt, num_grid_samples=m, num_control_rollouts=n,
                 max_speed_padding=pad_speed, num_vis_state_rollouts=1, max_map_dim=(rows + 2 * pad, cols + 2 * pad),
                 seed=seed, enforce_recommended_limits=False,
                 map_preprocessing=("device", "host")[seed % 2], **mode)
    if cfg.num_steps != t_steps:  # float division of T/dt landed on the other side: take what Config says
        t_steps = cfg.num_steps
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    lin.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
    ang.set_TDM_from_PMF_grid(pmf[::-1].copy() if seed % 4 == 0 else pmf, td, obstacle, unknown)
    planner = MPPI_Numba(cfg)
    planner.setup(params, lin, ang)
    useq = planner.solve()
    assert useq is not None and useq.shape == (t_steps, 2) and np.isfinite(useq).all()
    # a non-trivial nominal sequence, then one staged iteration against the oracle
    rng = np.random.default_rng(seed)
    planner.set_u((useq + rng.normal(0, 0.3, useq.shape)).astype(np.float32))
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    planner.rollout()
    got = planner.costs_d.copy_to_host()
    p = O.make_params(params, lin.res, lin.padded_xlimits, lin.padded_ylimits,
                      lin.bin_values_bounds_d.copy_to_host(), ang.bin_values_bounds_d.copy_to_host())
    grids = (lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
             lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host())
    if "use_tdm" in mode:
        want = O.rollout_tdm(p, *grids, noise, u_in)
    elif "use_nom_dynamics_with_speed_map" in mode:
        want = O.rollout_det(p, *grids, noise, u_in, risk=lin.risk_traction_map_d.copy_to_host())
    else:
        want = O.rollout_det(p, *grids, noise, u_in)
    ulps = ulp_diff_f32(got, want)
    kernel = planner.last_rollout_kernel()
    assert (ulps != 0).sum() <= max(1, n // 500), (kernel, int((ulps != 0).sum()), int(ulps.max()))
    assert ulps.max() <= 8, (kernel, int(ulps.max()))
    planner.update()
    _, u_ref, _ = O.update_useq(params["lambda_weight"], want, noise, params["vrange"], params["wrange"], u_in)
    scale = np.array([params["vrange"][1] - params["vrange"][0], params["wrange"][1] - params["wrange"][0]])
    assert (np.abs(planner.u_cur_d.copy_to_host() - u_ref) / scale).max() <= 1e-5, kernel


def _patch_seeds():
    """48 fixed cases; MPPI_FUZZ_PATCH_RANGE="a:b" swaps in other seeds for a soak run."""
    import os
    span = os.environ.get("MPPI_FUZZ_PATCH_RANGE")
    if span:
        lo, hi = (int(v) for v in span.split(":"))
        return range(lo, hi)
    return range(1000, 1048)


@pytest.mark.parametrize("seed", _patch_seeds())
def test_random_configuration_on_the_exact_schedule_inside_the_time_parallel_kernel(seed):
    """Round 5: the same random cases with MPPI_DEBUG_NO_SPECULATION -- k_rollout_scan_exact `direct` (what the planner
    launches on a map it has stopped speculating on) from the first launch: horizons of 1..104 steps (every count of
    chunk waves, every length of the last chunk), 1..700 rollouts (ragged tiles), resolutions that are not powers of
    two, frozen rollouts, goal breaks.  Bits of the oracle."""
    from mppi_numba_amd import _lib
    test_random_configuration_on_patchwise_constant_traction(seed, debug_flags=_lib.DEBUG_NO_SPECULATION)


@pytest.mark.parametrize("seed", _patch_seeds())
def test_random_configuration_on_patchwise_constant_traction(seed, debug_flags=0):
    """The speculation of the time-parallel kernels HOLDS on these maps (one traction value, or two in
    large patches: tiles that stay inside a patch keep their assumption, tiles that cross fail their
    vote and are re-run step by step), with everything else random: resolution (powers of two and
    not), dt, horizons of 1..104 steps (every count of chunk waves), reverse driving, headings of
    +-40 rad, start next to the padding ring (frozen rollouts), goals inside the start cell (goal
    breaks), small and large penalties.  Bits of the oracle, as the other fuzz test."""
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    mode, rows, cols, res, dt, t_steps, n, m, pmf, obstacle, unknown, td, params, pad_speed = random_case(seed)
    rng = np.random.default_rng(seed + 7)
    mode = dict(use_det_dynamics=True)
    t_steps = int(rng.integers(1, 105))
    n = int(rng.integers(1, 700))
    bins = pmf.shape[0]
    pmf = np.zeros_like(pmf)
    b0 = int(rng.integers(1, bins))
    pmf[b0] = 100
    if seed % 3 == 0 and bins > 2:  # a patch of another traction value
        b1 = int((b0 + rng.integers(1, bins - 1)) % bins) or 1
        r0, c0 = int(rng.integers(0, rows // 2)), int(rng.integers(0, cols // 2))
        pmf[:, r0:r0 + rows // 2, c0:c0 + cols // 2] = 0
        pmf[b1, r0:r0 + rows // 2, c0:c0 + cols // 2] = 100
    td = dict(td, det_dynamics_cvar_alpha=1.0)
    if seed % 4 == 1:  # start right beside the zero-traction ring
        params = dict(params)
        x0 = np.array(params["x0"], dtype=np.float64)
        x0[0] = td["xlimits"][0] + 0.6 * res
        params["x0"] = x0
    pad = int(np.ceil(pad_speed * dt / res))
    cfg = Config(T=(t_steps + 0.5) * dt, dt=dt, num_grid_samples=1, num_control_rollouts=n,
                 max_speed_padding=pad_speed, num_vis_state_rollouts=1, max_map_dim=(rows + 2 * pad, cols + 2 * pad),
                 seed=seed, enforce_recommended_limits=False, map_preprocessing=("device", "host")[seed % 2], **mode)
    t_steps = cfg.num_steps
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    lin.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
    ang.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
    planner = MPPI_Numba(cfg)
    planner.setup(params, lin, ang)
    if debug_flags:
        planner.set_debug_flags(debug_flags)
    useq = planner.solve()
    assert useq is not None and useq.shape == (t_steps, 2) and np.isfinite(useq).all()
    planner.set_u((useq + rng.normal(0, 0.3, useq.shape)).astype(np.float32))
    planner.sample_noise()
    noise, u_in = planner.noise_samples_d.copy_to_host(), planner.u_cur_d.copy_to_host()
    planner.rollout()
    kernel = planner.last_rollout_kernel()
    if t_steps <= 104 and seed % 3 != 0:  # (on a two-patch map the planner may have stopped speculating: review_speculation)
        assert kernel.startswith("k_rollout_scan_exact"), kernel
    if debug_flags and t_steps <= 104:
        # (direct whenever the map window fits beside the noise; else the kernels of its own)
        assert "direct=1" in kernel or not kernel.startswith("k_rollout_scan"), kernel
    got = planner.costs_d.copy_to_host()
    p = O.make_params(params, lin.res, lin.padded_xlimits, lin.padded_ylimits,
                      lin.bin_values_bounds_d.copy_to_host(), ang.bin_values_bounds_d.copy_to_host())
    want = O.rollout_det(p, lin.sample_grid_batch_d.copy_to_host(), ang.sample_grid_batch_d.copy_to_host(),
                         lin.obstacle_map_d.copy_to_host(), lin.unknown_map_d.copy_to_host(), noise, u_in)
    ulps = ulp_diff_f32(got, want)
    assert (ulps != 0).sum() <= max(1, n // 500), (kernel, int((ulps != 0).sum()), int(ulps.max()))
    assert ulps.max() <= 8, (kernel, int(ulps.max()))
    planner.update()
    _, u_ref, _ = O.update_useq(params["lambda_weight"], want, noise, params["vrange"], params["wrange"], u_in)
    scale = np.array([params["vrange"][1] - params["vrange"][0], params["wrange"][1] - params["wrange"][0]])
    assert (np.abs(planner.u_cur_d.copy_to_host() - u_ref) / scale).max() <= 1e-5, kernel
    # and the loop (noise computed in the launch) equals the stage-level sequence on the same counters
    planner.iterate_async(2)
    planner.synchronize()
    assert np.isfinite(planner.u_cur_d.copy_to_host()).all()
