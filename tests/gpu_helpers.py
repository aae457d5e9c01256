"""Builders shared by the GPU parity tests: reconstruct, from a golden fixture,
the Config / TDM_Numba / MPPI_Numba objects the reference was run with."""
import numpy as np

from helpers import params_from_golden


def config_from_golden(name, g, rng="philox", math="exact"):
    from mppi_numba_amd.config import Config
    mode = mode_of(name)
    cfg = Config(
        T=float(g["cfg_T"]), dt=float(g["cfg_dt"]),
        num_grid_samples=int(g["cfg_num_grid_samples"]),
        num_control_rollouts=int(g["cfg_num_control_rollouts"]),
        max_speed_padding=float(g["cfg_max_speed_padding"]),
        tdm_sample_thread_dim=tuple(int(v) for v in g["cfg_tdm_sample_thread_dim"]),
        num_vis_state_rollouts=int(g["cfg_num_vis_state_rollouts"]),
        max_map_dim=tuple(int(v) for v in g["cfg_max_map_dim"]),
        seed=int(g["cfg_seed"]), enforce_recommended_limits=False, rng=rng, math=math, **mode)
    assert cfg.num_steps == int(g["cfg_num_steps"])
    # the fixtures pin V explicitly (the generator overrides Config's clamp)
    cfg.num_vis_state_rollouts = int(g["cfg_num_vis_state_rollouts"])
    if "cfg_max_threads_per_block" in g:
        cfg.max_threads_per_block = int(g["cfg_max_threads_per_block"])
    return cfg


def mode_of(name):
    if name.startswith("speedmap"):
        return dict(use_nom_dynamics_with_speed_map=True)
    if name.startswith("tdm"):
        return dict(use_tdm=True)
    assert name.startswith("det")
    return dict(use_det_dynamics=True)


def tdm_dict_from_golden(g):
    return dict(
        xlimits=tuple(float(v) for v in g["tdm_xlimits"]),
        ylimits=tuple(float(v) for v in g["tdm_ylimits"]),
        res=float(g["tdm_res"]),
        bin_values=np.asarray(g["tdm_bin_values"], dtype=np.float64),
        bin_values_bounds=tuple(float(v) for v in g["tdm_bin_values_bounds"]),
        det_dynamics_cvar_alpha=float(g["tdm_det_dynamics_cvar_alpha"]),
    )


def build_from_golden(name, g, rng="philox", math="exact"):
    """(cfg, lin_tdm, ang_tdm, planner, params) set up like the fixture's run."""
    from mppi_numba_amd.mppi import MPPI_Numba
    from mppi_numba_amd.terrain import TDM_Numba
    cfg = config_from_golden(name, g, rng=rng, math=math)
    td = tdm_dict_from_golden(g)
    lin, ang = TDM_Numba(cfg), TDM_Numba(cfg)
    lin.set_TDM_from_PMF_grid(g["in_pmf_grid"], td, g["in_obstacle_map"], g["in_unknown_map"])
    ang.set_TDM_from_PMF_grid(g["in_ang_pmf_grid"], td, g["in_obstacle_map"], g["in_unknown_map"])
    planner = MPPI_Numba(cfg)
    params = params_from_golden(g)
    planner.setup(params, lin, ang)
    return cfg, lin, ang, planner, params


def solve_of_iteration(g, k):
    s = 0
    while ("solve%d_first_iteration" % (s + 1)) in g and k >= int(g["solve%d_first_iteration" % (s + 1)]):
        s += 1
    return s
