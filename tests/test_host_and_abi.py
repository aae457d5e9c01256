"""CPU-only checks: the C ABI library loads and exports every symbol declared
in include/mppi_hip.h (no compute calls without a GPU), the ctypes structs match
the C structs, and the pure-numpy host preprocessing reproduces the reference's
maps (golden fixtures)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import golden
from mppi_numba_amd import _lib, tdm_host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mppi_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mppi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 40
    for name in names:
        assert hasattr(lib, name), "libmppi_hip.so does not export %s" % name
    # and the binding covers the header (mppi_last_error is bound separately)
    assert set(names) - {"mppi_last_error"} == set(_lib.SIGNATURES)
    assert lib.mppi_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof/offsetof of the ctypes mirrors against a C program compiled from the header."""
    src = tmp_path / "layout.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "%s"\n'
        "int main(void) {\n"
        '  printf("%%zu %%zu %%zu %%zu\\n", sizeof(mppi_params), sizeof(mppi_planner_cfg), sizeof(mppi_tdm_cfg), sizeof(mppi_device_props));\n'
        '  printf("%%zu %%zu %%zu %%zu\\n", offsetof(mppi_params, dist_weight), offsetof(mppi_params, num_opt), offsetof(mppi_planner_cfg, seed), offsetof(mppi_tdm_cfg, seed));\n'
        "  return 0; }\n" % HEADER)
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes = [int(v) for v in out]
    assert sizes[:4] == [C.sizeof(_lib.Params), C.sizeof(_lib.PlannerCfg), C.sizeof(_lib.TdmCfg),
                         C.sizeof(_lib.DeviceProps)]
    assert sizes[4:] == [_lib.Params.dist_weight.offset, _lib.Params.num_opt.offset,
                         _lib.PlannerCfg.seed.offset, _lib.TdmCfg.seed.offset]


def test_errors_are_reported_not_swallowed():
    """Without a GPU every compute entry point must fail loudly (no CPU fallback)."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    assert _lib.device_count() == 0
    cfg = _lib.PlannerCfg(device=0, mode=0, num_control_rollouts=64, num_steps=10, num_grid_samples=1,
                          num_vis_state_rollouts=1, rng=0, math=0, rank=0, world_size=1, seed=1)
    handle = C.c_void_p()
    with pytest.raises(_lib.MppiError) as err:
        _lib.call("mppi_planner_create", C.byref(cfg), C.byref(handle))
    assert err.value.code == -4  # MPPI_ERR_NO_DEVICE
    from mppi_numba_amd.config import Config
    from mppi_numba_amd.mppi import MPPI_Numba
    with pytest.raises(_lib.MppiError):
        MPPI_Numba(Config(use_det_dynamics=True))
    from mppi_numba_amd.batch import MPPI_Batch
    with pytest.raises(_lib.MppiError):
        MPPI_Batch(Config(use_det_dynamics=True, num_control_rollouts=128), 4)
    # argument checks come before the device probe
    bad = _lib.PlannerCfg(device=0, mode=0, num_control_rollouts=64, num_steps=10, num_grid_samples=1,
                          num_vis_state_rollouts=1, rng=0, math=0, rank=0, world_size=1, num_instances=-2, seed=1)
    with pytest.raises(_lib.MppiError) as err:
        _lib.call("mppi_planner_create", C.byref(bad), C.byref(handle))
    assert err.value.code in (-1, -4)


def test_config_mirrors_reference_clamps(capsys):
    from mppi_numba_amd.config import Config
    c = Config(T=10, dt=0.1, num_control_rollouts=50, use_det_dynamics=True)
    assert c.num_control_rollouts == 100 and c.num_steps == 100
    c = Config(num_control_rollouts=70000, use_tdm=True)
    assert c.num_control_rollouts == 15000 and c.num_vis_state_rollouts == 20
    c = Config(num_control_rollouts=65536, use_det_dynamics=True, enforce_recommended_limits=False)
    assert c.num_control_rollouts == 65536
    c = Config(num_grid_samples=0, use_tdm=True)
    assert c.num_grid_samples == 1 and c.num_vis_state_rollouts == 1
    with pytest.raises(AssertionError):
        Config(use_tdm=True, use_det_dynamics=True)
    with pytest.raises(AssertionError):
        Config()
    with pytest.raises(AssertionError):
        Config(use_tdm=True, map_preprocessing="somewhere")
    assert Config(use_tdm=True).map_preprocessing == "device"
    import copy
    import pickle
    c2 = pickle.loads(pickle.dumps(copy.deepcopy(c)))
    assert c2.num_grid_samples == 1
    capsys.readouterr()


MODES = {"det_cvar": "det", "det_mean": "det", "speedmap_cvar": "speed", "speedmap_mean": "speed", "speedmap_mean_bounds": "speed",
         "tdm_cvar": "tdm", "det_odd_units": "det", "speedmap_odd_units": "speed", "tdm_odd_units": "tdm"}


@pytest.mark.parametrize("name", sorted(MODES))
def test_host_preprocessing_vs_reference(name):
    g = golden(name)
    alpha = float(g["tdm_det_dynamics_cvar_alpha"])
    bin_values = np.asarray(g["tdm_bin_values"]).astype(np.float32)
    bounds = np.asarray(g["tdm_bin_values_bounds"]).astype(np.float32)
    pmf_in = g["in_pmf_grid"]
    if MODES[name] == "det":
        pmf = tdm_host.one_hot_cvar_pmf(pmf_in, bin_values, alpha)
    elif MODES[name] == "speed":
        pmf = np.zeros_like(pmf_in)
        pmf[-1] = 100
    else:
        pmf = pmf_in
    assert (pmf == g["lin_pmf_grid_unpadded"]).all()
    max_map_dim = tuple(int(v) for v in g["cfg_max_map_dim"])
    vr, vc, pad, _, _ = tdm_host.padding_info(pmf.shape, max_map_dim, float(g["cfg_max_speed_padding"]),
                                              float(g["cfg_dt"]), float(g["tdm_res"]))
    assert pad == int(g["lin_pad_cells"])
    assert (tdm_host.pad_pmf(pmf, vr, vc, pad) == g["lin_pmf_grid_padded"]).all()
    assert (tdm_host.pad_mask(g["in_obstacle_map"], vr, vc, pad) == g["lin_obstacle_map_padded"]).all()
    px, py = tdm_host.padded_limits(g["tdm_xlimits"], g["tdm_ylimits"], vr, vc, pad, float(g["tdm_res"]))
    assert np.array_equal(px, g["lin_padded_xlimits"]) and np.array_equal(py, g["lin_padded_ylimits"])
    if MODES[name] == "speed":
        risk = tdm_host.risk_traction_map(pmf_in, bin_values, bounds, alpha)
        assert (tdm_host.pad_layer(risk, vr, vc, pad) == g["lin_risk_traction_map_padded"]).all()
    # the sampled det-mode grid is the one-hot bin's table value
    table = tdm_host.bin_table(g["lin_bin_values"], g["lin_bin_values_bounds"])
    if MODES[name] != "tdm":
        padded = g["lin_pmf_grid_padded"]
        expect = table[np.argmax(padded == 100, axis=0)]
        assert (expect == g["solve0_lin_sample_grid"][0]).all()


def test_padding_crops_to_max_map_dim():
    pmf = np.zeros((2, 30, 40), dtype=np.int8)
    pmf[1] = 100
    vr, vc, pad, mr, mc = tdm_host.padding_info(pmf.shape, (20, 24), 5.0, 0.1, 0.25)
    assert pad == 2 and (vr, vc) == (16, 20) and (mr, mc) == (16, 20)
    out = tdm_host.pad_pmf(pmf, vr, vc, pad)
    assert out.shape == (2, 20, 24)
    assert (out[0, :pad] == 100).all() and (out[1, pad:-pad, pad:-pad] == 100).all()
    assert (out.sum(axis=0) == 100).all()


def test_bin_table_truncates_in_device_dtype():
    """0.35 -> 35 with float64 bin values, 34 with float32."""
    b64 = tdm_host.bin_table(np.array([0.0, 0.35, 1.0]), np.array([0.0, 1.0]))
    b32 = tdm_host.bin_table(np.array([0.0, 0.35, 1.0], dtype=np.float32), np.array([0.0, 1.0], dtype=np.float32))
    assert list(b64) == [0, 35, 100] and list(b32) == [0, 34, 100]


SEMANTIC = {"semantic_tdm": "tdm", "semantic_det": "det", "semantic_det_mean": "det", "semantic_speedmap": "speed"}


def semantic_inputs(g):
    values = g["in_values"]
    id2name = {0: "dirt", 1: "grass", 2: "mud"}
    name2terrain = {k: "TERRAIN_" + k for k in id2name.values()}
    terrain2pmf = {name2terrain[k]: (values, g["in_pmf_" + k]) for k in id2name.values()}
    alpha = float(g["in_alpha"])
    return values, id2name, name2terrain, terrain2pmf, (None if alpha < 0 else alpha)


@pytest.mark.parametrize("name", sorted(SEMANTIC))
def test_semantic_grid_preprocessing_vs_reference(name):
    """set_TDM_from_semantic_grid's host part (terrain.py:183-342)."""
    g = golden(name)
    values, id2name, name2terrain, terrain2pmf, alpha = semantic_inputs(g)
    bounds = np.array([0.0, 1.0], dtype=np.float32)
    pmf, risk = tdm_host.semantic_pmf_grid(g["in_semantic_grid"], lambda sid: name2terrain[id2name[sid]],
                                           terrain2pmf, len(values), bounds, SEMANTIC[name], alpha)
    assert (pmf == g["lin_pmf_grid_unpadded"]).all()
    max_map_dim = tuple(int(v) for v in g["cfg_max_map_dim"])
    vr, vc, pad, _, _ = tdm_host.padding_info(pmf.shape, max_map_dim, float(g["cfg_max_speed_padding"]),
                                              float(g["cfg_dt"]), float(g["in_res"]))
    assert (tdm_host.pad_pmf(pmf, vr, vc, pad) == g["lin_pmf_grid_padded"]).all()
    if risk is not None:
        assert (tdm_host.pad_layer(risk, vr, vc, pad) == g["lin_risk_traction_map_padded"]).all()
    # this entry point keeps the caller's float64 bin values on the device (terrain.py:332):
    # 0.35 truncates to 35 here, to 34 through set_TDM_from_PMF_grid's float32 copy
    assert g["lin_bin_values"].dtype == np.float64
    table = tdm_host.bin_table(tdm_host.as_device_float(values), tdm_host.as_device_float(np.array([0.0, 1.0])))
    assert table[3] == 35
    assert set(np.unique(g["lin_sample_grid"])) <= set(table.tolist())


def test_mean_risk_map_scales_like_the_reference_for_any_bounds():
    """terrain.py:476-478 scales the MEAN risk traction as (100*(mean - lo))/range, terrain.py:488-490
    the CVaR one as 100*((cvar - lo)/range): after the int8 truncation the two orders differ in
    a few cells per thousand once the bounds are not (0, 1).  One- and two-hot PMFs, decimal
    upper bounds (where 100*mean/hi lands on integers)."""
    from mppi_numba_amd import tdm_host
    rng = np.random.default_rng(17)
    bins, rows, cols = 6, 120, 200
    bin_values = np.linspace(0.0, 1.0, bins).astype(np.float32)
    differing = 0
    for trial in range(8):
        pmf = np.zeros((bins, rows * cols), dtype=np.int8)
        first = rng.integers(0, bins, rows * cols)
        second = rng.integers(0, bins, rows * cols)
        share = rng.integers(0, 101, rows * cols)
        np.add.at(pmf, (first, np.arange(rows * cols)), share.astype(np.int8))
        np.add.at(pmf, (second, np.arange(rows * cols)), (100 - share).astype(np.int8))
        pmf = pmf.reshape(bins, rows, cols)
        bounds = np.array([0.0, (0.3, 0.5, 0.8, 1.2, 1.7, 2.0, 2.5, 3.0)[trial]], dtype=np.float32)
        # the reference's expressions, literally (float64 cumsums, float32 bounds)
        weighted = np.cumsum(0.01 * pmf.astype(float) * bin_values.reshape((-1, 1, 1)), axis=0)
        traction_range = bounds[1] - bounds[0]
        want_mean = np.reshape(100 * (weighted[-1] - bounds[0]) / traction_range, (1, rows, cols)).astype(np.int8)
        got_mean = tdm_host.risk_traction_map(pmf, bin_values, bounds, 1.0)
        assert np.array_equal(got_mean, want_mean)
        other_order = np.reshape(100 * np.asarray((weighted[-1] - bounds[0]) / traction_range), (1, rows, cols)).astype(np.int8)
        differing += int((other_order != want_mean).sum())
    assert differing > 0  # the sweep does exercise the distinction


def test_roctx_ranges_are_opt_in():
    """SURVEY.md section 5 tracing hook: MPPI_ROCTX=1 turns the roctx ranges on (the marker library
    ships with ROCm), nothing is loaded otherwise."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from mppi_numba_amd import _lib; "
            "print(_lib.load().mppi_trace_ranges_enabled())" % ROOT)
    env = dict(os.environ)
    env.pop("MPPI_ROCTX", None)
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout.strip() == "0"
    env["MPPI_ROCTX"] = "1"
    assert subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout.strip() == "1"
