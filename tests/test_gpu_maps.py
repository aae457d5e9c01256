"""Map preprocessing on the device against the host path, which is the
numpy code of the reference (terrain.py:408-495, 511-583) and is itself pinned by the golden
fixtures: bit-identical padded PMF, masks and risk map for every planner mode, then the same
rollout costs end to end on a reference fixture."""
import numpy as np
import pytest

from gpu_helpers import config_from_golden, tdm_dict_from_golden
from helpers import golden

pytestmark = pytest.mark.gpu

MODES = {
    "det": dict(use_det_dynamics=True),
    "speed": dict(use_nom_dynamics_with_speed_map=True),
    "tdm": dict(use_tdm=True),
}


def random_pmf(rng, bins, rows, cols, spoil=0):
    raw = rng.dirichlet(np.ones(bins) * 0.7, size=(rows, cols))
    p = np.floor(raw * 100).astype(np.int64)
    p[..., -1] += 100 - p.sum(axis=-1)
    # a few degenerate columns: all mass in one bin, in the first bin, in the last bin
    p[0, 0] = 0; p[0, 0, bins // 2] = 100
    p[1, 1] = 0; p[1, 1, 0] = 100
    p[2, 2] = 0; p[2, 2, -1] = 100
    for k in range(spoil):  # columns that do not sum to 100
        p[3 + k, 4, 1] += 7
    return np.ascontiguousarray(np.moveaxis(p, -1, 0)).astype(np.int8)


def build_pair(mode, alpha, bins, rows, cols, max_dim, res=0.25, spoil=0, masks=True, m=1):
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.terrain import TDM_Numba
    rng = np.random.default_rng(bins * 1000 + rows + int(alpha * 100))
    pmf = random_pmf(rng, bins, rows, cols, spoil)
    obstacle = (rng.random((rows, cols)) < 0.05).astype(np.int8) if masks else None
    unknown = (rng.random((rows, cols)) < 0.05).astype(np.int8) if masks else None
    td = dict(xlimits=(-3.0, -3.0 + cols * res), ylimits=(2.0, 2.0 + rows * res), res=res,
              bin_values=np.linspace(0.0, 0.9, bins), bin_values_bounds=(0.0, 1.2),
              det_dynamics_cvar_alpha=alpha)
    out = []
    for where in ("host", "device"):
        cfg = Config(T=2.0, dt=0.1, num_grid_samples=m if mode == "tdm" else 1, num_control_rollouts=128,
                     max_speed_padding=4.0, num_vis_state_rollouts=1, max_map_dim=max_dim, seed=1,
                     enforce_recommended_limits=False, map_preprocessing=where, **MODES[mode])
        tdm = TDM_Numba(cfg)
        tdm.set_TDM_from_PMF_grid(pmf, td, obstacle, unknown)
        out.append(tdm)
    return out


@pytest.mark.parametrize("mode", ["det", "speed", "tdm"])
@pytest.mark.parametrize("alpha", [1.0, 0.5, 0.07])
def test_device_preprocessing_is_bit_identical_to_the_host_path(mode, alpha):
    host, dev = build_pair(mode, alpha, bins=13, rows=90, cols=70, max_dim=(120, 120))
    assert np.array_equal(host.pmf_grid_d.copy_to_host(), dev.pmf_grid_d.copy_to_host())
    assert np.array_equal(host.obstacle_map_d.copy_to_host(), dev.obstacle_map_d.copy_to_host())
    assert np.array_equal(host.unknown_map_d.copy_to_host(), dev.unknown_map_d.copy_to_host())
    assert np.array_equal(host.pmf_grid, dev.pmf_grid)
    assert np.array_equal(host.padded_xlimits, dev.padded_xlimits)
    assert np.array_equal(host.padded_ylimits, dev.padded_ylimits)
    if mode == "speed":
        assert np.array_equal(host.risk_traction_map_d.copy_to_host(), dev.risk_traction_map_d.copy_to_host())
    # and the grids sampled from them (det modes: the one-hot PMF samples to itself)
    if mode != "tdm":
        assert np.array_equal(host.sample_grids().copy_to_host(), dev.sample_grids().copy_to_host())


def test_cropping_and_missing_masks_and_bad_columns(capsys):
    host, dev = build_pair("det", 0.3, bins=5, rows=100, cols=130, max_dim=(70, 64), spoil=3, masks=False)
    text = capsys.readouterr().out
    assert "cropped" in text and "sum up to 100" in text
    assert dev.pmf_grid_d.shape == host.pmf_grid_d.shape
    assert np.array_equal(host.pmf_grid_d.copy_to_host(), dev.pmf_grid_d.copy_to_host())
    assert not dev.obstacle_map_d.copy_to_host().any()
    assert np.array_equal(host.pmf_grid[:, :dev.pmf_grid.shape[1], :dev.pmf_grid.shape[2]], dev.pmf_grid)


@pytest.mark.parametrize("name", ["det_cvar", "det_mean", "speedmap_cvar", "speedmap_mean", "speedmap_mean_bounds", "tdm_cvar",
                                  "det_odd_units", "speedmap_odd_units", "tdm_odd_units"])
def test_device_preprocessing_reproduces_the_reference_fixtures(name):
    """The maps the REFERENCE built from the fixture's raw PMF (its padded PMF, masks, risk
    traction map, limits), rebuilt by the HIP kernel."""
    from mppi_numba_amd.terrain import TDM_Numba
    g = golden(name)
    cfg = config_from_golden(name, g)
    cfg.map_preprocessing = "device"
    td = tdm_dict_from_golden(g)
    for which, raw in (("lin", g["in_pmf_grid"]), ("ang", g["in_ang_pmf_grid"])):
        tdm = TDM_Numba(cfg)
        tdm.set_TDM_from_PMF_grid(raw, td, g["in_obstacle_map"], g["in_unknown_map"])
        assert np.array_equal(tdm.pmf_grid_d.copy_to_host(), g[which + "_pmf_grid_padded"])
        assert np.array_equal(tdm.pmf_grid, g[which + "_pmf_grid_unpadded"])
        assert np.array_equal(tdm.obstacle_map_d.copy_to_host(), g[which + "_obstacle_map_padded"])
        assert np.array_equal(tdm.unknown_map_d.copy_to_host(), g[which + "_unknown_map_padded"])
        assert np.array_equal(np.asarray(tdm.padded_xlimits), g[which + "_padded_xlimits"])
        if name.startswith("speedmap"):
            assert np.array_equal(tdm.risk_traction_map_d.copy_to_host(), g[which + "_risk_traction_map_padded"])
