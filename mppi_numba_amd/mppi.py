#!/usr/bin/env python3
"""`MPPI_Numba` on MI355X: Model Predictive Path Integral control for a unicycle
with terrain-dependent traction.

Host-side mirror of /root/reference/mppi_numba/mppi.py:39-608: same class name,
same methods (reset / setup / set_params / set_tdm / solve / shift_and_update /
get_state_rollout), same public attributes, same print-and-return-None error
behaviour.  The eight @cuda.jit kernels of the reference (mppi.py:611-1370) are
hand-written HIP kernels behind the C ABI (include/mppi_hip.h: mppi_planner_*);
one solve() is ONE library call instead of nine host-to-device copies and
2 + 3*num_opt kernel launches driven from Python.

Extensions (not in the reference) are grouped at the end of the class:
stage-level hooks for parity tests, device-side shift, multi-GPU sharding.
"""
import copy
import ctypes as C
import time

import numpy as np

from . import _lib
from .device_array import DeviceArray

# penalties used when params lacks 'obs_penalty' / 'unknown_penalty' (mppi.py:32-33)
DEFAULT_UNKNOWN_COST = float(1e2)
DEFAULT_OBS_COST = float(1e5)
# weight of the distance-to-goal term in the stage cost (mppi.py:36)
DEFAULT_DIST_WEIGHT = 1.0


def _f32(values):
    return np.asarray(values, dtype=np.float64).astype(np.float32)


class MPPI_Numba(object):

    """
    Planner object; device memory is allocated once at construction.

    Typical workflow (same as the reference):
      1. planner = MPPI_Numba(cfg)
      2. planner.reset()
      3. planner.setup(mppi_params, linear_tdm, angular_tdm)
      4. useq = planner.solve()
      5. planner.get_state_rollout()            (visualisation)
      6. planner.shift_and_update(next_state, useq, num_shifts=1)
      7. repeat from 2 when the traction maps change
    """

    def __init__(self, cfg, rank=0, world_size=1, sample_shard=None):
        self.cfg = cfg
        for name in ("T", "dt", "num_steps", "num_grid_samples", "num_control_rollouts",
                     "max_speed_padding", "tdm_sample_thread_dim", "num_vis_state_rollouts",
                     "max_map_dim", "seed", "use_tdm", "use_det_dynamics",
                     "use_nom_dynamics_with_speed_map", "use_costmap"):
            setattr(self, name, getattr(cfg, name))
        self.det_dyn = self.use_det_dynamics or self.use_nom_dynamics_with_speed_map or self.use_costmap
        self.max_threads_per_block = cfg.max_threads_per_block

        # multi-GPU extension: this object owns rollouts [rank*N/world, (rank+1)*N/world)
        self.rank = int(rank)
        self.world_size = int(world_size)
        # ... or, in the CVaR mode, all N rollouts over traction samples [r*M/G, (r+1)*M/G): sample_shard
        # = (r, G), with TDM_Numba(cfg, sample_shard=(r, G)) maps; one all-gather of the (N, M/G)
        # per-sample costs per iteration, every rank then holds all costs and updates locally
        self.sample_shard = (0, 1) if sample_shard is None else (int(sample_shard[0]), int(sample_shard[1]))
        if self.sample_shard[1] > 1:
            assert cfg.use_tdm and self.world_size == 1, "sample shards: use_tdm, and no control-sample shards"
            assert cfg.num_grid_samples % (2 * self.sample_shard[1]) == 0
        # batched multi-query extension (batch.py): problems solved by this handle
        self.num_instances = int(getattr(self, "num_instances", 1))

        self._handle = None
        self.noise_samples_d = None
        self.u_cur_d = None
        self.u_prev_d = None
        self.costs_d = None
        self.weights_d = None
        self.rng_states_d = None
        self.state_rollout_batch_d = None

        self.device_var_initialized = False
        self.reset()

    def __del__(self):
        handle, self._handle = getattr(self, "_handle", None), None
        if handle is not None:
            try:
                _lib.load().mppi_planner_destroy(handle)
            except Exception:
                pass

    def __deepcopy__(self, memo):
        raise TypeError("MPPI_Numba owns device memory and cannot be deep-copied")

    def reset(self):
        self.u_seq0 = np.zeros((self.num_steps, 2), dtype=np.float32)
        self.params = None
        self.params_set = False

        self.lin_tdm = None
        self.ang_tdm = None
        self.tdm_set = False

        # the reference drops its handle on the previous control sequence here but keeps
        # u_cur_d (allocation is guarded below), so the last solution warm-starts the next task
        self.u_prev_d = None
        self.init_device_vars_before_solving()

    def _mode(self):
        if self.use_det_dynamics:
            return _lib.MODE_DET
        if self.use_nom_dynamics_with_speed_map:
            return _lib.MODE_SPEED_MAP
        if self.use_tdm:
            return _lib.MODE_TDM
        print("None of the planner options are selected.")
        assert False

    def init_device_vars_before_solving(self):
        """One-time allocation of noise (N,T,2), u_cur/u_prev (T,2), costs (N),
        weights (N), state rollouts (V,T+1,3) (mppi.py:108-127).  With the default
        Philox generator no RNG state is stored at all."""
        if self.device_var_initialized:
            return
        t0 = time.time()
        cfg = _lib.PlannerCfg(
            device=getattr(self.cfg, "device", 0), mode=self._mode(),
            num_control_rollouts=int(self.num_control_rollouts), num_steps=int(self.num_steps),
            num_grid_samples=int(self.num_grid_samples) // self.sample_shard[1] if self.use_tdm else 1,
            num_vis_state_rollouts=int(self.num_vis_state_rollouts),
            rng=_lib.RNG_XOROSHIRO if getattr(self.cfg, "rng", "philox") == "xoroshiro" else _lib.RNG_PHILOX,
            math=_lib.MATH_FAST if getattr(self.cfg, "math", "exact") == "fast" else _lib.MATH_EXACT,
            rank=self.rank, world_size=self.world_size, num_instances=self.num_instances,
            seed=int(self.seed))
        handle = C.c_void_p()
        _lib.call("mppi_planner_create", C.byref(cfg), C.byref(handle))
        self._handle = handle
        if self.sample_shard[1] > 1:
            _lib.call("mppi_planner_set_sample_sharding", handle, self.sample_shard[0], self.sample_shard[1])
        # per-GPU rollouts; a batched handle stacks its problems: (B*n_per_problem, ...)
        self.num_local_rollouts = self.num_instances * (self.num_control_rollouts // self.world_size)
        n, t, v = self.num_local_rollouts, self.num_steps, self.num_vis_state_rollouts
        lead = () if self.num_instances == 1 else (self.num_instances,)
        ushape = lead + (t, 2)
        cshape = (n,) if self.num_instances == 1 else (self.num_instances, n // self.num_instances)
        self.noise_samples_d = DeviceArray((n, t, 2), np.float32, lambda: self._fetch("mppi_planner_get_noise", (n, t, 2)))
        self.u_cur_d = DeviceArray(ushape, np.float32, lambda: self._fetch("mppi_planner_get_u", ushape))
        self._u_prev_view = DeviceArray(ushape, np.float32, lambda: self._fetch("mppi_planner_get_u_prev", ushape))
        self.u_prev_d = self._u_prev_view
        self.costs_d = DeviceArray(cshape, np.float32, lambda: self._fetch("mppi_planner_get_costs", cshape))
        self.weights_d = DeviceArray(cshape, np.float32, lambda: self._fetch("mppi_planner_get_weights", cshape))
        self.rng_states_d = DeviceArray((self._rng_state_count(), 2), np.uint64, self._fetch_rng_states)
        self.state_rollout_batch_d = DeviceArray((v, t + 1, 3), np.float32, lambda: self._last_state_rollout.copy())
        self._last_state_rollout = np.zeros((v, t + 1, 3), dtype=np.float32)
        self.device_var_initialized = True
        print("MPPI planner has initialized GPU memory after {} s".format(time.time() - t0))

    # ------------------------------------------------------------------ device access
    def _fetch(self, fn, shape):
        out = np.empty(shape, dtype=np.float32)
        _lib.call(fn, self._handle, _lib.ptr(out, C.c_float))
        return out

    def _rng_state_count(self):
        n = C.c_long(0)
        _lib.call("mppi_planner_rng_states", self._handle, None, 0, C.byref(n))
        return int(n.value)

    def _fetch_rng_states(self):
        n = self._rng_state_count()
        out = np.zeros((n, 2), dtype=np.uint64)
        if n:
            cnt = C.c_long(0)
            _lib.call("mppi_planner_rng_states", self._handle, _lib.ptr(out, C.c_uint64), n, C.byref(cnt))
        return out

    # ------------------------------------------------------------------ reference API
    def setup(self, params, lin_tdm, ang_tdm):
        self.set_tdm(lin_tdm, ang_tdm)
        self.set_params(params)

    def is_within_bound(self, v, vbounds):
        return v >= vbounds[0] and v <= vbounds[1]

    def set_params(self, params):
        if not self.is_within_bound(params['x0'][0], self.lin_tdm.xlimits):
            print("ERROR: When setting mppi params, x0[0] is not within xlimits!")
            assert False
        if not self.is_within_bound(params['x0'][1], self.lin_tdm.ylimits):
            print("ERROR: When setting mppi params, x0[1] is not within ylimits!")
            assert False
        self.params = copy.deepcopy(params)
        self.params_set = True

    def set_tdm(self, lin_tdm, ang_tdm):
        self.lin_tdm = lin_tdm
        self.ang_tdm = ang_tdm
        self.tdm_set = True

    def check_solve_conditions(self):
        if not self.params_set:
            print("MPPI parameters are not set. Cannot solve")
            return False
        if not self.tdm_set:
            print("MPPI has not received TDMs. Cannot solve")
            return False
        if not self.device_var_initialized:
            print("Device variables not initialized. Cannot solve.")
            return False
        if not self.lin_tdm.pmf_grid_initialized:
            print("Linear TDM's PMF not initialized. Cannot solve.")
            return False
        if not self.ang_tdm.pmf_grid_initialized:
            print("Angular TDM's PMF not initialized. Cannot solve.")
            return False
        if not self.is_within_bound(self.params["x0"][0], self.lin_tdm.padded_xlimits):
            print("Robot initial condition not within padded xlimits.")
            return False
        if not self.is_within_bound(self.params["x0"][1], self.lin_tdm.padded_ylimits):
            print("Robot initial condition not within padded ylimits.")
            return False
        return True

    def move_mppi_task_vars_to_device(self):
        """Pack the task description the way mppi.py:214-234 casts it (np.float32
        everywhere except dist_weight / alpha_dyn) and hand it to the library."""
        p = self.params
        c = _lib.Params()
        # (assigning a Python / numpy scalar to a c_float field rounds float64 -> float32 to
        # nearest, which is what the reference's np.float32(...) casts do; no numpy temporaries
        # on the control path)
        for name, count in (("x0", 3), ("xgoal", 2), ("vrange", 2), ("wrange", 2), ("u_std", 2)):
            src, dst = p[name], getattr(c, name)
            for i in range(count):
                dst[i] = float(src[i])
        c.dt = float(p['dt'])
        c.goal_tolerance = float(p['goal_tolerance'])
        c.v_post_rollout = float(p['v_post_rollout'])
        c.lambda_weight = float(p['lambda_weight'])
        c.cvar_alpha = float(p['cvar_alpha'])
        c.obs_cost = float(p.get('obs_penalty', DEFAULT_OBS_COST))
        c.unknown_cost = float(p.get('unknown_penalty', DEFAULT_UNKNOWN_COST))
        c.res = float(np.float32(self.lin_tdm.res))
        c.xlo, c.ylo = self._padded_origin()
        c.dist_weight = float(p.get('dist_weight', DEFAULT_DIST_WEIGHT))
        c.alpha_dyn = float(p.get('alpha_dyn', 1.0))
        c.num_opt = int(p['num_opt'])
        _lib.call("mppi_planner_set_params", self._handle, C.byref(c))
        return c

    def _padded_origin(self):
        """float32 lower corner of the padded map (mppi.py:226-227), cached per map."""
        tdm = self.lin_tdm
        key = (id(tdm), float(tdm.padded_xlimits[0]), float(tdm.padded_ylimits[0]))
        if getattr(self, "_origin_key", None) != key:
            self._origin_key = key
            self._origin = (float(_f32(tdm.padded_xlimits)[0]), float(_f32(tdm.padded_ylimits)[0]))
        return self._origin

    def solve(self):
        """Entry point: sample the traction grids once, run params['num_opt']
        iterations of {sample noise, rollout, update}, return the (T,2) float32
        control sequence (None if the preconditions are not met, as the reference)."""
        if not self.check_solve_conditions():
            print("MPPI solve condition not met. Cannot solve. Return")
            return
        if self.use_tdm and self.cfg.num_grid_samples > self.cfg.max_threads_per_block:
            return self.solve_stochastic_oversized()
        return self._solve()

    # the reference has one method per variant; they differ only in the rollout kernel
    def solve_det_dyn(self):
        return self._solve()

    def solve_nom_dyn_w_speed_map(self):
        return self._solve()

    def solve_stochastic(self):
        return self._solve()

    def solve_stochastic_oversized(self):
        # M > max_threads_per_block: same kernel, lanes stride over the samples.  The
        # reference's variant (mppi.py:760-913) 'sorts' by swapping without comparing, so
        # its CVaR is only meaningful for cvar_alpha == 1; this one sorts properly.
        return self._solve()

    def _solve(self):
        self.move_mppi_task_vars_to_device()
        lead = () if self.num_instances == 1 else (self.num_instances,)
        useq = np.empty(lead + (self.num_steps, 2), dtype=np.float32)
        _lib.call("mppi_planner_solve", self._handle, self.lin_tdm._handle, self.ang_tdm._handle,
                  _lib.ptr(useq, C.c_float))
        self.u_prev_d = self._u_prev_view  # the reference aliases u_prev_d to u_cur_d in the loop
        return useq

    def shift_and_update(self, new_x0, u_cur, num_shifts=1):
        self.params["x0"] = new_x0.copy()
        self.shift_optimal_control_sequence(u_cur, num_shifts)

    def shift_optimal_control_sequence(self, u_cur, num_shifts=1):
        """u[:-k] = u[k:] on the caller's copy (the tail is kept, not zeroed) and a
        fresh upload, exactly as mppi.py:539-542."""
        shifted = u_cur.copy()
        shifted[:-num_shifts] = shifted[num_shifts:]
        shifted = np.ascontiguousarray(shifted.astype(np.float32))
        _lib.call("mppi_planner_set_u", self._handle, _lib.ptr(shifted, C.c_float))

    def get_state_rollout(self):
        """(V, T+1, 3) float32 state sequences for plotting (mppi.py:545-608):
        use_tdm -> the optimal controls over V sampled traction grids; otherwise
        row 0 = optimal controls, rows 1.. = noisy controls of the last iteration."""
        assert self.params_set, "MPPI parameters are not set"
        assert self.tdm_set, "MPPI has not received TDMs"
        if not self.device_var_initialized:
            print("Device variables not initialized. Cannot run mppi.")
            return
        self.move_mppi_task_vars_to_device()
        out = np.empty((self.num_vis_state_rollouts, self.num_steps + 1, 3), dtype=np.float32)
        _lib.call("mppi_planner_get_state_rollout", self._handle, self.lin_tdm._handle,
                  self.ang_tdm._handle, _lib.ptr(out, C.c_float))
        self._last_state_rollout = out
        return out.copy()

    # ------------------------------------------------------------------ extensions
    def shift_and_update_on_device(self, new_x0, num_shifts=1):
        """Like shift_and_update(new_x0, last_solution) without the host round trip."""
        self.params["x0"] = np.asarray(new_x0).copy()
        _lib.call("mppi_planner_shift_u", self._handle, int(num_shifts))

    def closed_loop(self, world, max_steps, goal_tolerance=None, x_init=None):
        """The notebooks' control loop (test.ipynb cell 4) on the device: max_steps times
        {solve, world.get at the current state, float64 Euler step, shift_and_update, goal
        check}, without a host round trip per control step.  `world` is a DeviceWorld, or a
        TractionGrid (its device twin is created and cached on it).  Returns
        (xhist (max_steps+1, 3) float64, uhist (max_steps, 2) float32, steps_taken) --
        with a leading problem axis for a batched handle; rows never reached are NaN, as in
        the notebook.  Afterwards params['x0'] and the device's control sequence are where
        the loop left them (the last solution, shifted once: the host looks at the goal flags only
        every 16 steps, the solves that ran meanwhile have advanced the generators' counters and
        nothing else -- a finished problem's controls are put aside at its goal and restored)."""
        from .terrain import DeviceWorld
        if not self.check_solve_conditions():
            print("MPPI solve condition not met. Cannot solve. Return")
            return
        if not isinstance(world, DeviceWorld):
            twin = getattr(world, "device_world", None)
            if twin is None:
                twin = DeviceWorld.from_traction_grid(world, device=getattr(self.cfg, "device", 0))
                world.device_world = twin
            world = twin
        B, single = self.num_instances, self.num_instances == 1
        self.move_mppi_task_vars_to_device()
        if single:
            x0 = np.asarray(self.params["x0"], dtype=np.float64).reshape(1, 3) if x_init is None \
                else np.asarray(x_init, dtype=np.float64).reshape(1, 3)
            x0_f32 = np.ascontiguousarray(x0.astype(np.float32))
            goal = np.ascontiguousarray(np.asarray(self.params["xgoal"], dtype=np.float64).astype(np.float32)).reshape(1, 2)
            _lib.call("mppi_planner_set_instances", self._handle, 1, _lib.ptr(x0_f32, C.c_float),
                      _lib.ptr(goal, C.c_float))
            x_init = x0
        tol = float(self.params["goal_tolerance"] if goal_tolerance is None else goal_tolerance)
        xhist = np.empty((B, max_steps + 1, 3), dtype=np.float64)
        uhist = np.empty((B, max_steps, 2), dtype=np.float32)
        steps = np.zeros(B, dtype=np.int32)
        xi = None
        if x_init is not None:
            xi = np.ascontiguousarray(np.asarray(x_init, dtype=np.float64).reshape(B, 3))
        try:
            _lib.call("mppi_planner_closed_loop", self._handle, self.lin_tdm._handle, self.ang_tdm._handle,
                      world._handle, int(max_steps), float(self.cfg.dt), tol, None if xi is None else _lib.ptr(xi, C.c_double),
                      _lib.ptr(xhist, C.c_double), _lib.ptr(uhist, C.c_float), _lib.ptr(steps, C.c_int))
        finally:
            if single:
                _lib.call("mppi_planner_set_instances", self._handle, 0, None, None)
        last = np.stack([xhist[b, steps[b]] for b in range(B)])
        if single:
            self.params["x0"] = last[0].copy()
            return xhist[0], uhist[0], int(steps[0])
        self.x0s = np.ascontiguousarray(last.astype(np.float32))
        self.params["x0"] = last[0].copy()
        return xhist, uhist, steps

    def set_u(self, u):
        u = np.ascontiguousarray(u, dtype=np.float32).reshape(self.num_instances * self.num_steps, 2)
        _lib.call("mppi_planner_set_u", self._handle, _lib.ptr(u, C.c_float))

    def sample_noise(self):
        self.move_mppi_task_vars_to_device()
        _lib.call("mppi_planner_sample_noise", self._handle)

    def set_noise(self, noise):
        noise = np.ascontiguousarray(noise, dtype=np.float32).reshape(self.num_local_rollouts, self.num_steps, 2)
        _lib.call("mppi_planner_set_noise", self._handle, _lib.ptr(noise, C.c_float))

    def rollout(self):
        """One rollout pass over the CURRENT noise, u and sampled grids -> costs_d."""
        self.move_mppi_task_vars_to_device()
        _lib.call("mppi_planner_rollout", self._handle, self.lin_tdm._handle, self.ang_tdm._handle)

    def set_costs(self, costs):
        costs = np.ascontiguousarray(costs, dtype=np.float32).reshape(self.num_local_rollouts)
        _lib.call("mppi_planner_set_costs", self._handle, _lib.ptr(costs, C.c_float))

    def update(self):
        """The control update from the CURRENT costs and noise -> weights_d, u_cur_d."""
        self.move_mppi_task_vars_to_device()
        _lib.call("mppi_planner_update", self._handle)
        self.u_prev_d = self._u_prev_view

    def record_sample_costs(self):
        _lib.call("mppi_planner_get_sample_costs", self._handle, None)

    def sample_costs(self):
        # (sample shards: all M costs of every control sample, gathered from the shards)
        out = np.empty((self.num_local_rollouts, self.num_grid_samples), dtype=np.float32)
        _lib.call("mppi_planner_get_sample_costs", self._handle, _lib.ptr(out, C.c_float))
        return out

    def time_kernels(self, reps=200):
        """(us per rollout launch, us per update launch) over `reps` ordinary iterations: the
        dispatch's own begin / end timestamps of every launch (include/mppi_hip.h)."""
        a, b = C.c_float(0), C.c_float(0)
        _lib.call("mppi_planner_time_kernels", self._handle, self.lin_tdm._handle, self.ang_tdm._handle,
                  int(reps), C.byref(a), C.byref(b))
        return float(a.value), float(b.value)

    def iterate_async(self, iterations):
        """Enqueue `iterations` x {noise, rollout, update} without sampling the TDMs
        and without copying u back; pair with synchronize()."""
        _lib.call("mppi_planner_iterate_async", self._handle, self.lin_tdm._handle,
                  self.ang_tdm._handle, int(iterations))

    def synchronize(self):
        _lib.call("mppi_planner_synchronize", self._handle)

    def last_rollout_kernel(self):
        """Which kernel variant the last rollout launch used (diagnostic string)."""
        buf = C.create_string_buffer(512)
        _lib.call("mppi_planner_describe_last_rollout", self._handle, buf, 512)
        return buf.value.decode()

    def set_debug_flags(self, flags):
        """Developer switches (_lib.DEBUG_*): which rollout kernel variant runs.  math="exact": never the
        costs (u to float32 resolution: the update's float64 summation tree follows the rollout kernel's
        tiles, 32 or 64 rollouts); math="fast": variants agree to float32 tolerance (include/mppi_hip.h)."""
        _lib.call("mppi_planner_set_debug_flags", self._handle, int(flags))

    def set_fold_poll_limit(self, polls):
        """Test hook: how long a workgroup of a rollout launch waits for the controls its siblings publish inside the
        launch before it gives the launch up (include/mppi_hip.h, MPPI_ERR_BUSY)."""
        _lib.call("mppi_planner_set_fold_poll_limit", self._handle, int(polls))

    def fold_state(self):
        """(folding, faults): whether this handle still folds its updates into the next rollout launch, and how many
        launches have given that hand-over up so far."""
        folding, faults = C.c_int(0), C.c_long(0)
        _lib.call("mppi_planner_fold_state", self._handle, C.byref(folding), C.byref(faults))
        return bool(folding.value), int(faults.value)

    def set_graph_replay(self, enabled=True, iterations_per_graph=2):
        """hipGraph replay of the iteration loop (include/mppi_hip.h); same results."""
        _lib.call("mppi_planner_set_graph_replay", self._handle, int(iterations_per_graph) if enabled else 0)

    def graph_stats(self):
        captures, replays = C.c_long(0), C.c_long(0)
        _lib.call("mppi_planner_graph_stats", self._handle, C.byref(captures), C.byref(replays))
        return dict(captures=int(captures.value), replays=int(replays.value))

    def set_profiling(self, enabled):
        _lib.call("mppi_planner_set_profiling", self._handle, int(bool(enabled)))

    def stage_times_ms(self):
        ms = (C.c_float * 4)()
        _lib.call("mppi_planner_stage_times", self._handle, ms)
        return dict(noise=ms[0], rollout=ms[1], update=ms[2], collective=ms[3])

    def last_elapsed_ms(self):
        ms = C.c_float(0)
        _lib.call("mppi_planner_last_elapsed_ms", self._handle, C.byref(ms))
        return float(ms.value)

    # multi-GPU: N sharded over ranks, one all-gather of 2T+2 doubles per iteration
    def comm_init(self, unique_id):
        assert len(unique_id) == _lib.COMM_ID_BYTES
        _lib.call("mppi_planner_comm_init", self._handle, C.c_char_p(bytes(unique_id)))

    def comm_count(self):
        """Ranks RCCL itself reports for this handle's communicator (0: none)."""
        n = C.c_int(0)
        _lib.call("mppi_planner_comm_count", self._handle, C.byref(n))
        return int(n.value)

    # multi-GPU without a collective: the peer exchange (include/mppi_hip.h)
    def p2p_export(self):
        """This rank's inbox as an opaque handle (bytes) for the other ranks' p2p_connect."""
        buf = C.create_string_buffer(_lib.P2P_HANDLE_BYTES)
        _lib.call("mppi_planner_p2p_export", self._handle, buf)
        return bytes(buf.raw)

    def p2p_connect(self, handles):
        """The inbox handles of ALL ranks in rank order (p2p_export of each; the own one is ignored)."""
        raw = b"".join(bytes(h) for h in handles)
        assert len(raw) == _lib.P2P_HANDLE_BYTES * len(handles)
        _lib.call("mppi_planner_p2p_connect", self._handle, C.c_char_p(raw), len(handles))

    def p2p_ping(self, token, timeout_ms=200):
        """All ranks together after p2p_connect (+ a host barrier): ranks whose token reached this rank's running
        kernel within the timeout (== world_size: the exchange works)."""
        heard = C.c_int(0)
        _lib.call("mppi_planner_p2p_ping", self._handle, C.c_ulonglong(int(token)), int(timeout_ms), C.byref(heard))
        return int(heard.value)

    def p2p_enable(self, enabled=True):
        """Switch a connected peer exchange off / on (off: the communicator's all-gather); all ranks alike."""
        _lib.call("mppi_planner_p2p_set_enabled", self._handle, int(bool(enabled)))

    def p2p_stats(self):
        on, count, kind = C.c_int(0), C.c_long(0), C.create_string_buffer(32)
        _lib.call("mppi_planner_p2p_stats", self._handle, C.byref(on), C.byref(count), kind, 32)
        return dict(connected=bool(on.value), exchanges=int(count.value), inbox=kind.value.decode())

    def update_local(self):
        n = C.c_int(0)
        _lib.call("mppi_planner_packet_len", self._handle, C.byref(n))
        packet = np.zeros(n.value, dtype=np.float64)
        self.move_mppi_task_vars_to_device()
        _lib.call("mppi_planner_update_local", self._handle, _lib.ptr(packet, C.c_double))
        return packet

    # multi-GPU, CVaR mode: M sharded over ranks (sample_shard), host-staged form of the exchange
    def sample_costs_local(self):
        """This shard's (N, M/G) per-sample costs of the last rollout()."""
        out = np.empty((self.num_local_rollouts, self.num_grid_samples // self.sample_shard[1]), dtype=np.float32)
        _lib.call("mppi_planner_sample_costs_local", self._handle, _lib.ptr(out, C.c_float))
        return out

    def sample_costs_apply(self, slabs):
        """The slabs of all G shards in rank order (G, N, M/G) -> costs_d (CVaR over all M)."""
        slabs = np.ascontiguousarray(slabs, dtype=np.float32)
        assert slabs.shape == (self.sample_shard[1], self.num_local_rollouts,
                               self.num_grid_samples // self.sample_shard[1])
        _lib.call("mppi_planner_sample_costs_apply", self._handle, _lib.ptr(slabs, C.c_float), int(slabs.shape[0]))

    def update_apply(self, packets):
        packets = np.ascontiguousarray(packets, dtype=np.float64)
        _lib.call("mppi_planner_update_apply", self._handle, _lib.ptr(packets, C.c_double),
                  int(packets.shape[0]))
        self.u_prev_d = self._u_prev_view

    def update_apply_and_rollout(self, packets):
        """update_apply(packets) and the next iteration's rollout() in one call (sample_noise() first):
        a time-parallel rollout launch applies the update itself, no k_apply launch in between."""
        packets = np.ascontiguousarray(packets, dtype=np.float64)
        self.move_mppi_task_vars_to_device()
        _lib.call("mppi_planner_update_apply_and_rollout", self._handle, _lib.ptr(packets, C.c_double),
                  int(packets.shape[0]), self.lin_tdm._handle, self.ang_tdm._handle)
        self.u_prev_d = self._u_prev_view


def comm_unique_id():
    """128-byte RCCL id: create on rank 0, broadcast to the other ranks by any
    means (file, socket, torch.distributed object broadcast), then comm_init()."""
    buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
    _lib.call("mppi_comm_unique_id", buf)
    return bytes(buf.raw)


class MPPI_Group(object):
    """One process, several GPUs: G shard planners (rank g on device g) driven by one control
    thread (include/mppi_hip.h: mppi_group_comm_init / mppi_group_iterate_async).  Every device
    holds its own copy of the maps (its own pair of TDMs).  The sharded problem is the one a
    single MPPI_Numba with N control samples solves: the noise is keyed by the global sample
    index and the packets are combined in rank order, so u is the same on every device.

        cfgs = [copy of cfg with .device = g for g in range(G)]          # cfg.num_control_rollouts = N (global)
        group = MPPI_Group(cfgs); group.setup(params, lin_tdms, ang_tdms); u = group.solve()
    """

    def __init__(self, cfgs):
        world = len(cfgs)
        self.planners = [MPPI_Numba(cfg, rank=g, world_size=world) for g, cfg in enumerate(cfgs)]
        self.world_size = world
        self._comm = False

    def setup(self, mppi_params, lin_tdms, ang_tdms):
        for p, lin, ang in zip(self.planners, lin_tdms, ang_tdms):
            p.setup(mppi_params, lin, ang)
        if not self._comm:
            handles = (C.c_void_p * self.world_size)(*[p._handle for p in self.planners])
            _lib.call("mppi_group_comm_init", handles, self.world_size)
            self._comm = True

    def connect_peers(self):
        """The peer exchange between the group's devices (peer access; include/mppi_hip.h): iterations of the
        time-parallel exact kernel then need neither RCCL nor a launch of their own for the update."""
        handles = (C.c_void_p * self.world_size)(*[p._handle for p in self.planners])
        _lib.call("mppi_group_p2p_connect", handles, self.world_size)

    def _arrays(self):
        mk = lambda hs: (C.c_void_p * self.world_size)(*hs)
        return (mk([p._handle for p in self.planners]), mk([p.lin_tdm._handle for p in self.planners]),
                mk([p.ang_tdm._handle for p in self.planners]))

    def iterate_async(self, iterations):
        for p in self.planners:
            p.move_mppi_task_vars_to_device()
        ps, lins, angs = self._arrays()
        _lib.call("mppi_group_iterate_async", ps, lins, angs, self.world_size, int(iterations))

    def synchronize(self):
        for p in self.planners:
            p.synchronize()

    def solve(self):
        """Sample the traction grids on every device, run params['num_opt'] iterations, return u."""
        alpha = self.planners[0].params.get("alpha_dyn", 1.0) if self.planners[0].use_tdm else 1.0
        for p in self.planners:
            p.lin_tdm.sample_grids(alpha)
            p.ang_tdm.sample_grids(alpha)
        self.iterate_async(int(self.planners[0].params["num_opt"]))
        self.synchronize()
        return self.planners[0].u_cur_d.copy_to_host()

    def shift_and_update(self, new_x0, u_cur, num_shifts=1):
        for p in self.planners:
            p.shift_and_update(new_x0, u_cur, num_shifts=num_shifts)
