#!/usr/bin/env python3
"""Map-free MPPI for a nominal unicycle with disc obstacles: the classes that
/root/reference/barebone_mppi_numba.ipynb defines inline (cell 2 `Config`,
cell 3 `MPPI_Numba`), on MI355X.  Same constructor keywords, methods and params
keys ('obstacle_positions', 'obstacle_radius', 'obs_penalty', 'dist_weight').

Stage cost dist_weight*d^2, terminal cost (1-reached)*d^2, obstacle penalty
when the POST-step position lies inside a disc (notebook cell 3).
"""
import copy
import ctypes as C
import time

import numpy as np

from . import _lib
from . import config as _config
from .device_array import DeviceArray

DEFAULT_OBS_COST = 1e3
DEFAULT_DIST_WEIGHT = 10

rec_max_control_rollouts = int(1e6)  # the notebook raises the cap of config.py
rec_min_control_rollouts = 100


class Config:

    """ Configurations that are typically fixed throughout execution. """

    def __init__(self, T=10, dt=0.1, num_control_rollouts=1024, num_vis_state_rollouts=20, seed=1,
                 enforce_recommended_limits=True, rng="philox", math="exact", device=0):
        assert T > 0
        assert dt > 0
        assert T > dt
        self.seed = seed
        self.T = T
        self.dt = dt
        self.num_steps = int(T / dt)
        assert self.num_steps > 0
        self.max_threads_per_block = _config.max_threads_per_block
        self.rng, self.math, self.device = rng, math, device

        self.num_control_rollouts = num_control_rollouts
        if enforce_recommended_limits:
            if self.num_control_rollouts > rec_max_control_rollouts:
                self.num_control_rollouts = rec_max_control_rollouts
                print("MPPI Config: Clip num_control_rollouts to be recommended max number of {}. (Max={})".format(
                    rec_max_control_rollouts, _config.max_blocks))
            elif self.num_control_rollouts < rec_min_control_rollouts:
                self.num_control_rollouts = rec_min_control_rollouts
                print("MPPI Config: Clip num_control_rollouts to be recommended min number of {}. (Recommended max={})".format(
                    rec_min_control_rollouts, rec_max_control_rollouts))
        self.num_vis_state_rollouts = max(1, min(num_vis_state_rollouts, self.num_control_rollouts))


def _f32(values):
    return np.asarray(values, dtype=np.float64).astype(np.float32)


class MPPI_Numba(object):

    """Information-theoretic MPPI (Williams et al., Alg. 2) without maps.
    Workflow: MPPI_Numba(cfg) -> setup(params) -> solve() -> get_state_rollout()
    -> shift_and_update(next_state, useq)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.T = cfg.T
        self.dt = cfg.dt
        self.num_steps = cfg.num_steps
        self.num_control_rollouts = cfg.num_control_rollouts
        self.num_vis_state_rollouts = cfg.num_vis_state_rollouts
        self.seed = cfg.seed
        self.max_threads_per_block = cfg.max_threads_per_block
        self._handle = None
        self.noise_samples_d = None
        self.u_cur_d = None
        self.u_prev_d = None
        self.costs_d = None
        self.weights_d = None
        self.rng_states_d = None
        self.state_rollout_batch_d = None
        self.device_var_initialized = False
        self._discs_key = None  # what the device's disc arrays hold (None: nothing handed over yet)
        self.reset()

    def __del__(self):
        handle, self._handle = getattr(self, "_handle", None), None
        if handle is not None:
            try:
                _lib.load().mppi_planner_destroy(handle)
            except Exception:
                pass

    def reset(self):
        self.u_seq0 = np.zeros((self.num_steps, 2), dtype=np.float32)
        self.params = None
        self.params_set = False
        self.u_prev_d = None
        self.init_device_vars_before_solving()

    def init_device_vars_before_solving(self):
        if self.device_var_initialized:
            return
        t0 = time.time()
        cfg = _lib.PlannerCfg(
            device=getattr(self.cfg, "device", 0), mode=_lib.MODE_BAREBONE,
            num_control_rollouts=int(self.num_control_rollouts), num_steps=int(self.num_steps),
            num_grid_samples=1, num_vis_state_rollouts=int(self.num_vis_state_rollouts),
            rng=_lib.RNG_XOROSHIRO if getattr(self.cfg, "rng", "philox") == "xoroshiro" else _lib.RNG_PHILOX,
            math=_lib.MATH_FAST if getattr(self.cfg, "math", "exact") == "fast" else _lib.MATH_EXACT,
            rank=0, world_size=1, seed=int(self.seed))
        handle = C.c_void_p()
        _lib.call("mppi_planner_create", C.byref(cfg), C.byref(handle))
        self._handle = handle
        n, t, v = self.num_control_rollouts, self.num_steps, self.num_vis_state_rollouts
        self.noise_samples_d = DeviceArray((n, t, 2), np.float32, lambda: self._fetch("mppi_planner_get_noise", (n, t, 2)))
        self.u_cur_d = DeviceArray((t, 2), np.float32, lambda: self._fetch("mppi_planner_get_u", (t, 2)))
        self._u_prev_view = DeviceArray((t, 2), np.float32, lambda: self._fetch("mppi_planner_get_u_prev", (t, 2)))
        self.u_prev_d = self._u_prev_view
        self.costs_d = DeviceArray((n,), np.float32, lambda: self._fetch("mppi_planner_get_costs", (n,)))
        self.weights_d = DeviceArray((n,), np.float32, lambda: self._fetch("mppi_planner_get_weights", (n,)))
        self._last_state_rollout = np.zeros((v, t + 1, 3), dtype=np.float32)
        self.state_rollout_batch_d = DeviceArray((v, t + 1, 3), np.float32, lambda: self._last_state_rollout.copy())
        self.device_var_initialized = True
        print("MPPI planner has initialized GPU memory after {} s".format(time.time() - t0))

    def _fetch(self, fn, shape):
        out = np.empty(shape, dtype=np.float32)
        _lib.call(fn, self._handle, _lib.ptr(out, C.c_float))
        return out

    def setup(self, params):
        self.set_params(params)

    def set_params(self, params):
        self.params = copy.deepcopy(params)
        self.params_set = True

    def check_solve_conditions(self):
        if not self.params_set:
            print("MPPI parameters are not set. Cannot solve")
            return False
        if not self.device_var_initialized:
            print("Device variables not initialized. Cannot solve.")
            return False
        return True

    def move_mppi_task_vars_to_device(self):
        """Pack the task description as notebook cell 3 casts it (np.float32 everywhere except dist_weight) and hand it to
        the library.  (Assigning a Python / numpy scalar to a c_float field rounds float64 -> float32 to nearest, which is
        what the np.float32(...) casts do: no numpy temporaries on the control path -- the notebook times solve() as a
        whole, `bench.py --workload bb` likewise.)"""
        p = self.params
        c = _lib.Params()
        for name, count in (("x0", 3), ("xgoal", 2), ("vrange", 2), ("wrange", 2), ("u_std", 2)):
            src, dst = p[name], getattr(c, name)
            for i in range(count):
                dst[i] = float(src[i])
        c.dt = float(p['dt'])
        c.goal_tolerance = float(p['goal_tolerance'])
        c.v_post_rollout = 0.0
        c.lambda_weight = float(p['lambda_weight'])
        c.cvar_alpha = 1.0
        c.obs_cost = float(DEFAULT_OBS_COST if 'obs_penalty' not in p else p['obs_penalty'])
        c.unknown_cost = 0.0
        c.res, c.xlo, c.ylo = 1.0, 0.0, 0.0
        c.dist_weight = float(DEFAULT_DIST_WEIGHT if 'dist_weight' not in p else p['dist_weight'])
        c.alpha_dyn = 1.0
        c.num_opt = int(p['num_opt'])
        _lib.call("mppi_planner_set_params", self._handle, C.byref(c))
        # the discs: handed over when they have changed (the notebook uploads them with every solve)
        if "obstacle_positions" in p and "obstacle_radius" in p:
            op, orad = np.asarray(p['obstacle_positions']), np.asarray(p['obstacle_radius'])
            key = (op.dtype.str, op.shape, op.tobytes(), orad.dtype.str, orad.shape, orad.tobytes())
            if key != self._discs_key:
                pos = np.ascontiguousarray(_f32(op).reshape(-1, 2))
                rad = np.ascontiguousarray(_f32(orad).reshape(-1))
                assert len(pos) == len(rad)
                _lib.call("mppi_planner_set_disc_obstacles", self._handle, _lib.ptr(pos, C.c_float),
                          _lib.ptr(rad, C.c_float), len(rad))
                self._discs_key = key
        elif self._discs_key != ():
            _lib.call("mppi_planner_set_disc_obstacles", self._handle, None, None, 0)
            self._discs_key = ()

    def solve(self):
        if not self.check_solve_conditions():
            print("MPPI solve condition not met. Cannot solve. Return")
            return
        return self.solve_with_nominal_dynamics()

    def solve_with_nominal_dynamics(self):
        self.move_mppi_task_vars_to_device()
        useq = np.empty((self.num_steps, 2), dtype=np.float32)
        _lib.call("mppi_planner_solve", self._handle, None, None, _lib.ptr(useq, C.c_float))
        self.u_prev_d = self._u_prev_view
        return useq

    def shift_and_update(self, new_x0, u_cur, num_shifts=1):
        self.params["x0"] = new_x0.copy()
        self.shift_optimal_control_sequence(u_cur, num_shifts)

    def shift_optimal_control_sequence(self, u_cur, num_shifts=1):
        shifted = u_cur.copy()
        shifted[:-num_shifts] = shifted[num_shifts:]
        shifted = np.ascontiguousarray(shifted.astype(np.float32))
        _lib.call("mppi_planner_set_u", self._handle, _lib.ptr(shifted, C.c_float))

    def get_state_rollout(self):
        assert self.params_set, "MPPI parameters are not set"
        if not self.device_var_initialized:
            print("Device variables not initialized. Cannot run mppi.")
            return
        self.move_mppi_task_vars_to_device()
        out = np.empty((self.num_vis_state_rollouts, self.num_steps + 1, 3), dtype=np.float32)
        _lib.call("mppi_planner_get_state_rollout", self._handle, None, None, _lib.ptr(out, C.c_float))
        self._last_state_rollout = out
        return out.copy()

    # --- stage-level hooks for parity tests (not in the notebook) ---
    def set_u(self, u):
        u = np.ascontiguousarray(u, dtype=np.float32).reshape(self.num_steps, 2)
        _lib.call("mppi_planner_set_u", self._handle, _lib.ptr(u, C.c_float))

    def set_noise(self, noise):
        noise = np.ascontiguousarray(noise, dtype=np.float32).reshape(self.num_control_rollouts, self.num_steps, 2)
        _lib.call("mppi_planner_set_noise", self._handle, _lib.ptr(noise, C.c_float))

    def sample_noise(self):
        self.move_mppi_task_vars_to_device()
        _lib.call("mppi_planner_sample_noise", self._handle)

    def rollout(self):
        self.move_mppi_task_vars_to_device()
        _lib.call("mppi_planner_rollout", self._handle, None, None)

    def set_costs(self, costs):
        costs = np.ascontiguousarray(costs, dtype=np.float32).reshape(self.num_control_rollouts)
        _lib.call("mppi_planner_set_costs", self._handle, _lib.ptr(costs, C.c_float))

    def update(self):
        self.move_mppi_task_vars_to_device()
        _lib.call("mppi_planner_update", self._handle)
        self.u_prev_d = self._u_prev_view

    def last_rollout_kernel(self):
        """Which kernel variant the last rollout launch used (diagnostic string)."""
        buf = C.create_string_buffer(512)
        _lib.call("mppi_planner_describe_last_rollout", self._handle, buf, 512)
        return buf.value.decode()

    def set_debug_flags(self, flags):
        """Developer switches (_lib.DEBUG_*): which rollout kernel variant runs; never the costs."""
        _lib.call("mppi_planner_set_debug_flags", self._handle, int(flags))
