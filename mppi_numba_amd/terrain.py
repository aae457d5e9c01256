#!/usr/bin/env python3
"""Traction distribution maps on MI355X: `TDM_Numba`, plus the small host-side
helpers `Terrain` and `TractionGrid`.

Host-side mirror of /root/reference/mppi_numba/terrain.py for the planner's hot
path.  `TDM_Numba` keeps the reference's method names, arguments and public
attributes; what used to be numba device arrays and the
`sample_grids_numba` kernel (terrain.py:633-695) now lives behind the C ABI
(include/mppi_hip.h: mppi_tdm_*).

PMF grids are int8 (num_bins, rows, cols) whose bins sum to 100 per cell.  The map
preprocessing the reference does with numpy on the host,
  * use_tdm: keep the PMF as given;
  * use_det_dynamics: all mass in the bin closest above the CVaR_alpha traction
    (terrain.py:408-452);
  * use_nom_dynamics_with_speed_map: nominal traction for the dynamics plus an
    int8 map of CVaR_alpha traction that scales the time cost (terrain.py:455-495);
then a ring of zero-traction cells, ceil(max_speed_padding*dt/res) wide, so that rollouts
can never leave the allocated map (terrain.py:511-583),
runs by default on the DEVICE (`Config(map_preprocessing="device")`, one HIP kernel,
csrc/map_kernels.h, bit-identical results); `Config(map_preprocessing="host")` selects the
numpy restatement in tdm_host.py, which is also what `set_TDM_from_semantic_grid` uses.
"""
import copy
import ctypes as C
import time

import numpy as np

from . import _lib, tdm_host
from .device_array import DeviceArray, HostMirror


class TDM_Numba(object):

    """
    Traction Distribution Map backed by device memory of libmppi_hip.so.

    Typical workflow (same as the reference):
        1. tdm = TDM_Numba(cfg)           allocate the sampled-grid batch once
        2. tdm.reset()
        3. tdm.set_TDM_from_semantic_grid(...) or tdm.set_TDM_from_PMF_grid(...)
        4. hand it to MPPI_Numba.setup(params, lin_tdm, ang_tdm)
        5. repeat from 2 when the map changes
    """

    def __init__(self, cfg, sample_shard=None):
        self.cfg = cfg
        for name in ("T", "dt", "num_steps", "num_grid_samples", "num_control_rollouts",
                     "max_speed_padding", "tdm_sample_thread_dim", "num_vis_state_rollouts",
                     "max_map_dim", "seed", "use_tdm", "use_det_dynamics",
                     "use_nom_dynamics_with_speed_map", "use_costmap"):
            setattr(self, name, getattr(cfg, name))
        self.det_dyn = self.use_det_dynamics or self.use_nom_dynamics_with_speed_map or self.use_costmap
        # multi-GPU extension (CVaR mode): this object draws samples [rank*M/count, (rank+1)*M/count)
        # of the M = cfg.num_grid_samples traction maps -- the very draws an unsharded TDM makes
        self.sample_shard = (0, 1) if sample_shard is None else (int(sample_shard[0]), int(sample_shard[1]))
        assert 0 <= self.sample_shard[0] < self.sample_shard[1]
        if self.sample_shard[1] > 1:
            assert not self.det_dyn, "only use_tdm has samples to shard"
            assert self.num_grid_samples % (2 * self.sample_shard[1]) == 0, \
                "num_grid_samples must be a multiple of 2 * shard count"

        self.thread_dim = self.tdm_sample_thread_dim
        self.block_dim = (1, self.num_grid_samples)
        self.total_threads = self.num_grid_samples * self.thread_dim[0] * self.thread_dim[1]

        self._handle = None
        self.sample_grid_batch_d = None
        self.risk_traction_map_d = None
        self.obstacle_map_d = None
        self.unknown_map_d = None
        self.rng_states_d = None

        self.device_var_initialized = False
        self.reset()

    # ------------------------------------------------------------------ life cycle
    def __del__(self):
        handle, self._handle = getattr(self, "_handle", None), None
        if handle is not None:
            try:
                _lib.load().mppi_tdm_destroy(handle)
            except Exception:
                pass

    def __deepcopy__(self, memo):
        raise TypeError("TDM_Numba owns device memory and cannot be deep-copied")

    def reset(self):
        # semantic-grid bookkeeping (simulation benchmarks only)
        self.semantic_grid = None
        self.semantic_grid_initialized = False
        self.id2name = None
        self.name2terrain = None
        self.id2terrain_fn = None
        self.terrain2pmf = None

        # PMF grid and its metadata
        self.pmf_grid = None
        self.bin_values = None
        self.bin_values_bounds = None
        self.pmf_grid_d = None
        self.bin_values_d = None
        self.bin_values_bounds_d = None
        self.num_pmf_bins = None
        self.xlimits = None
        self.ylimits = None
        self.padded_xlimits = None
        self.padded_ylimits = None
        self.pad_cells = None
        self.res = None
        self.pmf_grid_initialized = False

        self.risk_traction_map_d = None
        self.obstacle_map = None
        self.obstacle_map_d = None
        self.unknown_map = None
        self.unknown_map_d = None

        # visualisation (read by the reference's TDM_Visualizer)
        self.cell_dimensions = None
        self.figsize = None

        self.init_device_vars_before_sampling()

    def init_device_vars_before_sampling(self):
        """One-time allocation of the sampled-grid batch (terrain.py:164-180):
        (M, rows, cols) int8 for use_tdm, (1, rows, cols) otherwise."""
        if self.device_var_initialized:
            return
        t0 = time.time()
        rows, cols = self.max_map_dim
        self._num_grids = 1 if self.det_dyn else self.num_grid_samples // self.sample_shard[1]
        cfg = _lib.TdmCfg(
            device=getattr(self.cfg, "device", 0), num_grids=self._num_grids,
            max_rows=int(rows), max_cols=int(cols),
            thread_dim_x=int(self.thread_dim[0]), thread_dim_y=int(self.thread_dim[1]),
            rng=_lib.RNG_XOROSHIRO if getattr(self.cfg, "rng", "philox") == "xoroshiro" else _lib.RNG_PHILOX,
            seed=int(self.seed))
        handle = C.c_void_p()
        _lib.call("mppi_tdm_create", C.byref(cfg), C.byref(handle))
        self._handle = handle
        if self.sample_shard[1] > 1:
            _lib.call("mppi_tdm_set_sample_shard", handle, self.sample_shard[0] * self._num_grids)
        self.sample_grid_batch_d = DeviceArray((self._num_grids, rows, cols), np.int8,
                                               self._fetch_sampled_grids)
        self.rng_states_d = DeviceArray((self._rng_state_count(), 2), np.uint64, self._fetch_rng_states)
        self.device_var_initialized = True
        print("TDM has initialized GPU memory after {} s".format(time.time() - t0))

    # ------------------------------------------------------------------ device access
    def _fetch_sampled_grids(self):
        out = np.empty(self.sample_grid_batch_d.shape, dtype=np.int8)
        _lib.call("mppi_tdm_get_sampled_grids", self._handle, _lib.ptr(out, C.c_int8))
        return out

    def _rng_state_count(self):
        n = C.c_long(0)
        _lib.call("mppi_tdm_rng_states", self._handle, None, 0, C.byref(n))
        return int(n.value)

    def _fetch_rng_states(self):
        n = self._rng_state_count()
        out = np.zeros((n, 2), dtype=np.uint64)
        if n:
            cnt = C.c_long(0)
            _lib.call("mppi_tdm_rng_states", self._handle, _lib.ptr(out, C.c_uint64), n, C.byref(cnt))
        return out

    def set_sampled_grids(self, grids):
        """Test hook (not in the reference): overwrite sample_grid_batch_d[:, :r, :c]
        with a host array of shape (num_grids, r, c)."""
        g = np.ascontiguousarray(grids, dtype=np.int8)
        assert g.ndim == 3 and g.shape[0] == self._num_grids
        _lib.call("mppi_tdm_set_sampled_grids", self._handle, _lib.ptr(g, C.c_int8),
                  int(g.shape[1]), int(g.shape[2]))

    # ------------------------------------------------------------------ map setup
    def set_TDM_from_semantic_grid(self, sg, res, num_pmf_bins, bin_values, bin_values_bounds,
                                   xlimits, ylimits, id2name, name2terrain, terrain2pmf,
                                   det_dynamics_cvar_alpha=None,
                                   obstacle_map=None,
                                   unknown_map=None):
        """Build the PMF grid from a grid of semantic ids whose terrains have
        known (values, pmf) pairs (terrain.py:183-342); simulation benchmarks."""
        if det_dynamics_cvar_alpha is None:
            assert self.use_tdm or self.use_costmap
        else:
            assert 0 < det_dynamics_cvar_alpha <= 1.0

        self.semantic_grid = sg.copy()
        self.id2name = id2name
        self.name2terrain = name2terrain
        self.id2terrain_fn = lambda semantic_id: self.name2terrain[self.id2name[semantic_id]]
        self.terrain2pmf = terrain2pmf
        self.semantic_grid_initialized = True
        self.cell_dimensions = (res, res)
        self.xlimits = xlimits
        self.ylimits = ylimits
        num_rows, num_cols = sg.shape
        self.num_pmf_bins = num_pmf_bins
        self.bin_values = np.asarray(bin_values).astype(np.float32)
        self.bin_values_bounds = np.asarray(bin_values_bounds).astype(np.float32)
        self.res = res

        assert bin_values[0] == 0, "Assume minimum bin value is 0 for now"
        assert bin_values_bounds[0] == 0, "Assume minimum traction is 0 for now"

        if self.use_det_dynamics:
            mode = "det"
        elif self.use_nom_dynamics_with_speed_map:
            mode = "speed"
        elif self.use_tdm:
            mode = "tdm"
        else:
            assert False, "TDM cannot be set up"
        self.pmf_grid, risk = tdm_host.semantic_pmf_grid(
            self.semantic_grid, self.id2terrain_fn, self.terrain2pmf, num_pmf_bins,
            self.bin_values_bounds, mode, det_dynamics_cvar_alpha)
        risk_padded = None
        if risk is not None:
            risk_padded, _, _ = self.set_padding_risk_traction(risk, self.max_speed_padding, self.dt,
                                                               res, xlimits, ylimits)

        padded_pmf_grid, self.padded_xlimits, self.padded_ylimits = self.set_padding(
            self.pmf_grid, self.max_speed_padding, self.dt, res, xlimits, ylimits)
        # this entry point keeps the caller's dtype for the device copies (terrain.py:332-333)
        self._upload(padded_pmf_grid, np.asarray(bin_values), np.asarray(bin_values_bounds),
                     obstacle_map, unknown_map, num_rows, num_cols, res, risk_padded)

        rows_p, cols_p = self.pmf_grid_d.shape[1:]
        original = copy.deepcopy(self.semantic_grid)
        self.semantic_grid = original[:rows_p - 2 * self.pad_cells, :cols_p - 2 * self.pad_cells]
        self.pmf_grid_initialized = True

    def get_padded_grid_xy_dim(self):
        if self.pmf_grid_initialized:
            return self.pmf_grid_d.shape[1:]
        print("Padded grid has not been initialized yet.")
        return None

    def prepare_obstacle_and_unknown_map(self, obstacle_map, unknown_map, num_rows, num_cols, res):
        """int8 copies of the masks (zeros when absent) and their padded versions
        (terrain.py:353-371).  Returns (padded_obstacle, padded_unknown)."""
        if obstacle_map is not None:
            assert obstacle_map.shape == (num_rows, num_cols), "obstacle_map does not have the same XY dim as pmf grid."
            self.obstacle_map = np.asarray(obstacle_map).astype(np.int8).reshape(num_rows, num_cols)
        else:
            self.obstacle_map = np.zeros((num_rows, num_cols), dtype=np.int8)
        if unknown_map is not None:
            assert unknown_map.shape == (num_rows, num_cols), "unknown_map does not have the same XY dim as pmf grid."
            self.unknown_map = np.asarray(unknown_map).astype(np.int8).reshape(num_rows, num_cols)
        else:
            self.unknown_map = np.zeros((num_rows, num_cols), dtype=np.int8)
        padded_obstacle = self.set_padding_2d(self.obstacle_map, self.max_speed_padding, self.dt, res)
        padded_unknown = self.set_padding_2d(self.unknown_map, self.max_speed_padding, self.dt, res)
        self.obstacle_map_d = HostMirror(padded_obstacle)
        self.unknown_map_d = HostMirror(padded_unknown)
        return padded_obstacle, padded_unknown

    def print_bin_values_bounds(self, obj_name):
        if self.bin_values_bounds_d is None:
            print("{}: Bin value is None".format(obj_name))
        else:
            print("{}: bin values bounds are ".format(obj_name), self.bin_values_bounds_d.copy_to_host())

    def set_TDM_from_PMF_grid(self, pmf_grid, tdm_dict, obstacle_map=None, unknown_map=None):
        """Initialise from an int8 PMF grid (num_bins, rows, cols) and a dict with
        xlimits, ylimits, res, bin_values, bin_values_bounds,
        det_dynamics_cvar_alpha (terrain.py:380-508)."""
        alpha = tdm_dict["det_dynamics_cvar_alpha"]
        if not (0 < alpha <= 1.0):
            print("WARNING: TDM cannot be setup since alpha is not in (0,1]")
        assert alpha > 0
        assert alpha <= 1.0
        assert len(pmf_grid.shape) == 3, "PMF grid must have 3 dimensions"
        self.num_pmf_bins, num_rows, num_cols = pmf_grid.shape
        self.res = res = tdm_dict["res"]
        self.cell_dimensions = (res, res)
        self.xlimits = tdm_dict["xlimits"]
        self.ylimits = tdm_dict["ylimits"]
        self.bin_values = np.asarray(tdm_dict["bin_values"]).astype(np.float32)
        self.bin_values_bounds = np.asarray(tdm_dict["bin_values_bounds"]).astype(np.float32)
        assert self.bin_values[0] == 0, "Assume minimum bin value is 0 for now"
        assert self.bin_values_bounds[0] == 0, "Assume minimum traction is 0 for now"

        if getattr(self.cfg, "map_preprocessing", "device") == "device":
            return self._set_TDM_from_PMF_grid_on_device(pmf_grid, alpha, obstacle_map, unknown_map,
                                                         num_rows, num_cols, res)

        risk_padded = None
        if self.use_det_dynamics or self.use_nom_dynamics_with_speed_map:
            col_sums = np.sum(pmf_grid, axis=0)
            if (col_sums != 100).any():
                # (the reference's message at terrain.py:409-411 raises before printing)
                print("WARNING: the provided PMF has columns that don't sum up to 100: {}".format(
                    np.argwhere(col_sums != 100)))

        if self.use_det_dynamics:
            self.pmf_grid = tdm_host.one_hot_cvar_pmf(pmf_grid, self.bin_values, alpha)
            if (np.sum(self.pmf_grid, axis=0) != 100).any():
                print("WARNING: pmf_grid not properly set in set_TDM_from_PMF_grid. Values don't' sum to 100")
        elif self.use_nom_dynamics_with_speed_map:
            self.pmf_grid = np.zeros((self.num_pmf_bins, num_rows, num_cols), dtype=np.int8)
            self.pmf_grid[-1] = np.int8(100)  # nominal dynamics: traction == last bin
            risk = tdm_host.risk_traction_map(pmf_grid, self.bin_values, self.bin_values_bounds, alpha)
            risk_padded, _, _ = self.set_padding_risk_traction(risk, self.max_speed_padding, self.dt, res,
                                                               self.xlimits, self.ylimits)
        else:
            self.pmf_grid = np.asarray(pmf_grid).astype(np.int8)

        if (np.sum(self.pmf_grid, axis=0) != 100).any():
            print("WARNING: some PMF columns do not sum to 100: {}".format(
                np.argwhere(np.sum(self.pmf_grid, axis=0) != 100)))

        padded_pmf_grid, self.padded_xlimits, self.padded_ylimits = self.set_padding(
            self.pmf_grid, self.max_speed_padding, self.dt, res, self.xlimits, self.ylimits)
        # this entry point holds float32 copies on the device (terrain.py:401-406)
        self._upload(padded_pmf_grid, self.bin_values, self.bin_values_bounds,
                     obstacle_map, unknown_map, num_rows, num_cols, res, risk_padded)
        self.pmf_grid_initialized = True

    @property
    def pmf_grid(self):
        """Processed, unpadded PMF grid (bins, rows, cols) int8.  With device-side map
        preprocessing it lives on the GPU and is copied back on first access."""
        if self._pmf_grid is None and self._pmf_grid_fetch is not None:
            self._pmf_grid = self._pmf_grid_fetch()
        return self._pmf_grid

    @pmf_grid.setter
    def pmf_grid(self, value):
        self._pmf_grid = value
        self._pmf_grid_fetch = None

    def _set_TDM_from_PMF_grid_on_device(self, pmf_grid, alpha, obstacle_map, unknown_map,
                                         num_rows, num_cols, res):
        """Config(map_preprocessing="device"): CVaR bin / risk map / cropping / padding by
        one HIP kernel (csrc/map_kernels.h) instead of numpy; the host only ships the raw
        int8 PMF and masks.  Same results bit for bit (tests/test_gpu_maps.py); the host-side
        attributes (pmf_grid, *_d mirrors) are fetched from the device on demand."""
        vr, vc, pad = self.get_padding_info(pmf_grid.shape, self.max_speed_padding, self.dt, res)
        self.pad_cells = pad
        self.padded_xlimits, self.padded_ylimits = tdm_host.padded_limits(self.xlimits, self.ylimits, vr, vc, pad, res)
        kind = (_lib.PREP_DET if self.use_det_dynamics else
                _lib.PREP_SPEED if self.use_nom_dynamics_with_speed_map else _lib.PREP_TDM)
        raw = np.ascontiguousarray(pmf_grid, dtype=np.int8)
        masks = []
        for name, m in (("obstacle_map", obstacle_map), ("unknown_map", unknown_map)):
            if m is not None:
                assert m.shape == (num_rows, num_cols), "%s does not have the same XY dim as pmf grid." % name
                m = np.ascontiguousarray(np.asarray(m).astype(np.int8).reshape(num_rows, num_cols))
                setattr(self, name, m)
            else:
                setattr(self, name, np.zeros((num_rows, num_cols), dtype=np.int8))
            masks.append(m)
        bv = tdm_host.as_device_float(self.bin_values)
        bd = tdm_host.as_device_float(self.bin_values_bounds)
        table = tdm_host.bin_table(bv, bd)
        lo, ratio = tdm_host.traction_scale(bd)
        bv32 = np.ascontiguousarray(self.bin_values, dtype=np.float32)
        bd32 = np.ascontiguousarray(self.bin_values_bounds, dtype=np.float32)
        bad = C.c_int(0)
        _lib.call("mppi_tdm_set_maps_from_pmf", self._handle, kind, _lib.ptr(raw, C.c_int8), self.num_pmf_bins,
                  num_rows, num_cols, vr, vc, pad, _lib.ptr(bv32, C.c_float), _lib.ptr(bd32, C.c_float),
                  float(alpha), _lib.ptr(table, C.c_int8), lo, ratio,
                  None if masks[0] is None else _lib.ptr(masks[0], C.c_int8),
                  None if masks[1] is None else _lib.ptr(masks[1], C.c_int8), C.byref(bad))
        if bad.value or vr < num_rows or vc < num_cols:
            # rare path: the same warnings, under the same mode conditions and with the same cell
            # indices, as the host path above prints (the kernel only counts the offending columns
            # inside the crop)
            col_sums = np.sum(pmf_grid, axis=0)
            if (col_sums != 100).any():
                if self.use_det_dynamics or self.use_nom_dynamics_with_speed_map:
                    print("WARNING: the provided PMF has columns that don't sum up to 100: {}".format(
                        np.argwhere(col_sums != 100)))
                else:
                    print("WARNING: some PMF columns do not sum to 100: {}".format(np.argwhere(col_sums != 100)))
        bins, rows_p, cols_p = self.num_pmf_bins, vr + 2 * pad, vc + 2 * pad
        handle = self._handle

        def fetch(which):
            out = np.empty((bins, rows_p, cols_p) if which == 0 else (rows_p, cols_p), dtype=np.int8)
            args = [None, None, None, None]
            args[which] = _lib.ptr(out, C.c_int8)
            _lib.call("mppi_tdm_get_maps", handle, *args)
            return out

        self.pmf_grid_d = DeviceArray((bins, rows_p, cols_p), np.int8, lambda: fetch(0))
        self.obstacle_map_d = DeviceArray((rows_p, cols_p), np.int8, lambda: fetch(1))
        self.unknown_map_d = DeviceArray((rows_p, cols_p), np.int8, lambda: fetch(2))
        if kind == _lib.PREP_SPEED:
            self.risk_traction_map_d = DeviceArray((1, rows_p, cols_p), np.int8, lambda: fetch(3))
        # unpadded processed PMF, which the reference keeps on the host: fetched if anyone asks
        self.pmf_grid = None
        self._pmf_grid_fetch = lambda: fetch(0)[:, pad:pad + vr, pad:pad + vc].copy()
        self.bin_values_d = HostMirror(bv)
        self.bin_values_bounds_d = HostMirror(bd)
        self.bin_to_int8 = table
        self.traction_lo, self.traction_ratio = lo, ratio
        self.pmf_grid_initialized = True

    def _upload(self, padded_pmf_grid, bin_values_dev, bounds_dev, obstacle_map, unknown_map,
                num_rows, num_cols, res, risk_padded):
        """Everything the reference moves with cuda.to_device for one map
        (terrain.py:331-333, 370-371, 405-406, 495, 506) in one C call."""
        padded_obs, padded_unk = self.prepare_obstacle_and_unknown_map(obstacle_map, unknown_map,
                                                                       num_rows, num_cols, res)
        pmf = np.ascontiguousarray(padded_pmf_grid, dtype=np.int8)
        bins, rows_p, cols_p = pmf.shape
        assert padded_obs.shape == (rows_p, cols_p)
        bv = tdm_host.as_device_float(bin_values_dev)
        bd = tdm_host.as_device_float(bounds_dev)
        table = tdm_host.bin_table(bv, bd)
        lo, ratio = tdm_host.traction_scale(bd)
        risk_ptr = None
        if risk_padded is not None:
            risk2d = np.ascontiguousarray(risk_padded, dtype=np.int8).reshape(rows_p, cols_p)
            risk_ptr = _lib.ptr(risk2d, C.c_int8)
        obs_c = np.ascontiguousarray(padded_obs, dtype=np.int8)
        unk_c = np.ascontiguousarray(padded_unk, dtype=np.int8)
        _lib.call("mppi_tdm_set_maps", self._handle, _lib.ptr(pmf, C.c_int8), bins, rows_p, cols_p,
                  _lib.ptr(table, C.c_int8), lo, ratio, _lib.ptr(obs_c, C.c_int8),
                  _lib.ptr(unk_c, C.c_int8), risk_ptr)
        self.pmf_grid_d = HostMirror(pmf)
        self.bin_values_d = HostMirror(bv)
        self.bin_values_bounds_d = HostMirror(bd)
        self.bin_to_int8 = table
        self.traction_lo, self.traction_ratio = lo, ratio
        if risk_padded is not None:
            self.risk_traction_map_d = HostMirror(np.asarray(risk_padded, dtype=np.int8))

    # ------------------------------------------------------------------ padding
    def get_padding_info(self, grid_shape, max_speed_padding, dt, res):
        """(valid_rows, valid_cols, pad_cells): how much of the incoming grid fits
        the allocation once a ring of ceil(max_speed_padding*dt/res) cells is added."""
        rows, cols = grid_shape[-2], grid_shape[-1]
        valid_rows, valid_cols, pad_cells, max_rows, max_cols = tdm_host.padding_info(
            grid_shape, self.max_map_dim, max_speed_padding, dt, res)
        if max_rows < 1 or max_cols < 1:
            print("While padding the TDM, the max_allowed rows {} or cols {} are below 1.\nAllocated GPU array size: {}".format(
                max_rows, max_cols,
                [1 if self.det_dyn else self.num_grid_samples, self.max_map_dim[0], self.max_map_dim[1]]))
            assert False
        if valid_rows < rows or valid_cols < cols:
            print("WARNING: While padding the TDM, original PMF is cropped from ({}, {}) to ({}, {})to fit in allocated GPU memory.".format(
                rows, cols, valid_rows, valid_cols))
        return valid_rows, valid_cols, pad_cells

    def set_padding(self, pmf_grid, max_speed_padding, dt, res, xlimits, ylimits):
        """Surround the PMF grid with cells whose whole mass sits in bin 0 (zero
        traction), cropping from the bottom-left origin if needed (terrain.py:525-543)."""
        vr, vc, pad = self.get_padding_info(pmf_grid.shape, max_speed_padding, dt, res)
        self.pad_cells = pad
        px, py = tdm_host.padded_limits(xlimits, ylimits, vr, vc, pad, res)
        return tdm_host.pad_pmf(pmf_grid, vr, vc, pad), px, py

    def set_padding_risk_traction(self, grid, max_speed_padding, dt, res, xlimits, ylimits):
        """Same ring for the (1, rows, cols) risk traction map, filled with 0."""
        vr, vc, pad = self.get_padding_info(grid.shape, max_speed_padding, dt, res)
        self.pad_cells = pad
        px, py = tdm_host.padded_limits(xlimits, ylimits, vr, vc, pad, res)
        return tdm_host.pad_layer(grid, vr, vc, pad), px, py

    def set_padding_2d(self, map, max_speed_padding, dt, res, pad_val=0):
        """Same ring for a 2-D int8 mask."""
        vr, vc, pad = self.get_padding_info(map.shape, max_speed_padding, dt, res)
        self.pad_cells = pad
        return tdm_host.pad_mask(map, vr, vc, pad, pad_val)

    # ------------------------------------------------------------------ sampling
    def sample_grids_true_dist(self):
        """One traction realisation per cell from the terrains' TRUE densities
        (not the PMF); needs the semantic grid (terrain.py:586-608)."""
        flat = self.semantic_grid.flatten()
        ids, counts = np.unique(flat, return_counts=True)
        lins = np.zeros_like(self.semantic_grid, dtype=float)
        angs = np.zeros_like(self.semantic_grid, dtype=float)
        for sid, num in zip(ids, counts):
            lin_s, ang_s = self.id2terrain_fn(sid).sample_traction(int(num))
            mask = self.semantic_grid == sid
            lins[mask] = lin_s
            angs[mask] = ang_s
        return TractionGrid(lins, angs)

    def sample_grids_true_dist_on_device(self, seed=None, device=None):
        """sample_grids_true_dist with the draw made on the GPU: every cell picks, with a
        Philox counter keyed by `seed`, one entry of the sample pool its terrain type keeps
        (Terrain.lin_saved_samples / ang_saved_samples), linear and angular independently.
        Returns a TractionGrid whose arrays are the device draw and whose `.device_world`
        (DeviceWorld) is what MPPI_Numba.closed_loop steps through without host round trips."""
        sg = np.asarray(self.semantic_grid)
        ids = np.unique(sg)
        pools_lin, pools_ang = [], []
        for sid in ids:
            terrain = self.id2terrain_fn(sid)
            pools_lin.append(np.asarray(terrain.lin_saved_samples, dtype=np.float64).ravel())
            pools_ang.append(np.asarray(terrain.ang_saved_samples, dtype=np.float64).ravel())
        pool_len = min(min(len(a) for a in pools_lin), min(len(a) for a in pools_ang))
        assert pool_len > 0, "a terrain type has no saved samples"
        lin_pool = np.ascontiguousarray(np.stack([a[:pool_len] for a in pools_lin]))
        ang_pool = np.ascontiguousarray(np.stack([a[:pool_len] for a in pools_ang]))
        terrain_of_cell = np.ascontiguousarray(np.searchsorted(ids, sg).astype(np.int32))
        dev = getattr(self.cfg, "device", 0) if device is None else device
        world = DeviceWorld(sg.shape[0], sg.shape[1], res=1.0, device=dev)
        world.sample_true_dist(terrain_of_cell, lin_pool, ang_pool,
                               seed=getattr(self.cfg, "seed", 1) if seed is None else seed)
        lins, angs = world.get_grids()
        grid = TractionGrid(lins, angs)
        grid.device_world = world
        return grid

    def sample_grids(self, alpha_dyn=1.0):
        """Draw the traction grids from the PMF on the GPU (terrain.py:610-622);
        returns the (M or 1, rows, cols) int8 device batch."""
        _lib.call("mppi_tdm_sample_grids", self._handle, float(alpha_dyn))
        return self.sample_grid_batch_d

    def int8_grid_to_float32(self, int8grid):
        ratio = np.asarray(int8grid.copy()).astype(np.float32) / 100.
        return ratio * (self.bin_values_bounds[1] - self.bin_values_bounds[0]) + self.bin_values_bounds[0]


class TractionGrid(object):

    """A deterministic grid of traction coefficients (e.g. one sampled world)."""

    def __init__(self, lin_traction, ang_traction, res=1.0, use_int8=False, xlimits=None, ylimits=None):
        if use_int8:
            self.lin_traction = (100 * lin_traction).astype(np.int8)
            self.ang_traction = (100 * ang_traction).astype(np.int8)
        else:
            self.lin_traction = lin_traction
            self.ang_traction = ang_traction
        self.res = res
        self.height, self.width = self.lin_traction.shape
        self.xlimits = (0, self.res * self.width) if xlimits is None else xlimits
        self.ylimits = (0, self.res * self.height) if ylimits is None else ylimits

    def get(self, x, y):
        """(lin, ang) traction at a world position; (0, 0) outside the grid."""
        xi = int((x - self.xlimits[0]) // self.res)
        yi = int((y - self.ylimits[0]) // self.res)
        if xi < 0 or xi >= self.width or yi < 0 or yi >= self.height:
            return 0, 0
        return self.lin_traction[yi, xi], self.ang_traction[yi, xi]

    def get_grids(self):
        return self.lin_traction, self.ang_traction


class DeviceWorld(object):

    """TractionGrid on the GPU (include/mppi_hip.h: mppi_world_*): float64 (rows, cols) grids of
    linear / angular traction, `get` for batches of points with the reference's cell rule
    (terrain.py:776-782), the true-distribution draw (terrain.py:586-608) and the world the
    planner's device-side closed loop steps through (MPPI_Numba.closed_loop)."""

    def __init__(self, rows, cols, res=1.0, xlimits=None, ylimits=None, lin=None, ang=None, device=0):
        self.rows, self.cols, self.res = int(rows), int(cols), float(res)
        self.xlimits = (0, self.res * self.cols) if xlimits is None else tuple(xlimits)
        self.ylimits = (0, self.res * self.rows) if ylimits is None else tuple(ylimits)
        self._handle = C.c_void_p()
        lin_p = ang_p = None
        if lin is not None:
            lin = np.ascontiguousarray(lin, dtype=np.float64).reshape(self.rows, self.cols)
            ang = np.ascontiguousarray(ang, dtype=np.float64).reshape(self.rows, self.cols)
            lin_p, ang_p = _lib.ptr(lin, C.c_double), _lib.ptr(ang, C.c_double)
        _lib.call("mppi_world_create", int(device), self.rows, self.cols, self.res, float(self.xlimits[0]),
                  float(self.ylimits[0]), lin_p, ang_p, C.byref(self._handle))

    @classmethod
    def from_traction_grid(cls, grid, device=0):
        """The device twin of a host TractionGrid (same cells, limits and resolution)."""
        return cls(grid.height, grid.width, res=grid.res, xlimits=grid.xlimits, ylimits=grid.ylimits,
                   lin=np.asarray(grid.lin_traction, dtype=np.float64),
                   ang=np.asarray(grid.ang_traction, dtype=np.float64), device=device)

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                _lib.load().mppi_world_destroy(h)
            except Exception:
                pass
            self._handle = None

    def get(self, x, y):
        """(lin, ang) at world positions; scalars in -> scalars out, arrays -> arrays."""
        xs, ys = np.broadcast_arrays(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))
        xy = np.ascontiguousarray(np.stack([xs.ravel(), ys.ravel()], axis=1))
        lin = np.empty(xy.shape[0], dtype=np.float64)
        ang = np.empty(xy.shape[0], dtype=np.float64)
        _lib.call("mppi_world_get", self._handle, _lib.ptr(xy, C.c_double), int(xy.shape[0]),
                  _lib.ptr(lin, C.c_double), _lib.ptr(ang, C.c_double))
        if xs.ndim == 0:
            return lin[0], ang[0]
        return lin.reshape(xs.shape), ang.reshape(xs.shape)

    def get_grids(self):
        lin = np.empty((self.rows, self.cols), dtype=np.float64)
        ang = np.empty((self.rows, self.cols), dtype=np.float64)
        _lib.call("mppi_world_get_grids", self._handle, _lib.ptr(lin, C.c_double), _lib.ptr(ang, C.c_double))
        return lin, ang

    def sample_true_dist(self, terrain_of_cell, lin_pool, ang_pool, seed=1):
        terrain_of_cell = np.ascontiguousarray(terrain_of_cell, dtype=np.int32).reshape(self.rows * self.cols)
        lin_pool = np.ascontiguousarray(lin_pool, dtype=np.float64)
        ang_pool = np.ascontiguousarray(ang_pool, dtype=np.float64)
        assert lin_pool.ndim == 2 and lin_pool.shape == ang_pool.shape
        _lib.call("mppi_world_sample_true_dist", self._handle, _lib.ptr(terrain_of_cell, C.c_int32),
                  int(lin_pool.shape[0]), _lib.ptr(lin_pool, C.c_double), _lib.ptr(ang_pool, C.c_double),
                  int(lin_pool.shape[1]), int(seed) & 0xFFFFFFFFFFFFFFFF)


class Terrain(object):

    """A semantic terrain type with densities for linear and angular traction
    (any object with sample(n), mean(samples), var(samples), cvar(alpha, samples=, front=))."""

    def __init__(self, name, rgb, lin_density, ang_density, cvar_alpha=0.1, cvar_front=True, num_saved_samples=1e4):
        self.name = name
        self.rgb = rgb
        self.lin_density = lin_density
        self.ang_density = ang_density
        self.num_saved_samples = num_saved_samples
        self.lin_saved_samples = lin_density.sample(num_saved_samples)
        self.ang_saved_samples = ang_density.sample(num_saved_samples)
        self.cvar_alpha = cvar_alpha
        self.cvar_front = cvar_front
        for tag, dens, saved in (("lin", lin_density, self.lin_saved_samples),
                                 ("ang", ang_density, self.ang_saved_samples)):
            setattr(self, tag + "_mean", dens.mean(saved))
            setattr(self, tag + "_var", dens.var(saved))
            setattr(self, tag + "_std", np.sqrt(getattr(self, tag + "_var")))
        self.update_cvar_alpha(cvar_alpha)

    def update_cvar_alpha(self, alpha):
        assert 0 < alpha <= 1.0
        self.cvar_alpha = alpha
        self.lin_cvar, self.lin_cvar_thres = self.lin_density.cvar(
            alpha, samples=self.lin_saved_samples, front=self.cvar_front)
        self.ang_cvar, self.ang_cvar_thres = self.ang_density.cvar(
            alpha, samples=self.ang_saved_samples, front=self.cvar_front)

    def sample_traction(self, num_samples):
        return self.lin_density.sample(num_samples), self.ang_density.sample(num_samples)

    def __repr__(self):
        return ("Terrain {}: mean=({:.2f}, {:.2f}), std=({:.2f}, {:.2f}), cvar({:.2f})=({:.2f}, {:.2f}) "
                "from {} saved samples").format(self.name, self.lin_mean, self.ang_mean, self.lin_std,
                                                self.ang_std, self.cvar_alpha, self.lin_cvar, self.ang_cvar,
                                                self.num_saved_samples)
