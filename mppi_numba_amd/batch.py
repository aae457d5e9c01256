#!/usr/bin/env python3
"""`MPPI_Batch`: B independent MPPI problems in one handle (batched multi-query).

Not in the reference (BASELINE.json configs[4]).  A single
unicycle problem at the reference's sizes (N ~ 1e3..1e4) leaves most of an
MI355X idle: the rollout is a sequential integrator, so one problem keeps at
most N/64 wavefronts busy.  A planner that answers many queries against the same
traction maps (several robots, several candidate goals, a receding-horizon
sweep) fills the machine by stacking them: every kernel launch covers
(problem, rollout), each problem keeps its own control sequence, minimum cost and
normaliser, and its result is bit-identical to a single-problem `MPPI_Numba`
fed the same noise.

    batch = MPPI_Batch(cfg, num_instances=64)
    batch.setup(params, lin_tdm, ang_tdm, x0s, goals)   # params as for MPPI_Numba
    useqs = batch.solve()                               # (B, T, 2) float32
    batch.shift_and_update(new_x0s, useqs, num_shifts=1)

Everything in `params` except 'x0' and 'xgoal' is shared by the problems.
"""
import ctypes as C

import numpy as np

from . import _lib
from .mppi import MPPI_Numba


class MPPI_Batch(MPPI_Numba):

    def __init__(self, cfg, num_instances, rank=0, world_size=1):
        num_instances = int(num_instances)
        assert num_instances >= 1, "num_instances must be >= 1"
        assert not getattr(cfg, "use_costmap", False)
        self.num_instances = num_instances
        self.x0s = None
        self.goals = None
        super().__init__(cfg, rank=rank, world_size=world_size)

    def reset(self):
        super().reset()
        self.u_seq0 = np.zeros((self.num_instances, self.num_steps, 2), dtype=np.float32)

    # ------------------------------------------------------------------ task set-up
    def setup(self, params, lin_tdm, ang_tdm, x0s=None, goals=None):
        self.set_tdm(lin_tdm, ang_tdm)
        if x0s is None:
            x0s = np.tile(np.asarray(params["x0"], dtype=np.float32), (self.num_instances, 1))
        if goals is None:
            goals = np.tile(np.asarray(params["xgoal"], dtype=np.float32), (self.num_instances, 1))
        params = dict(params)
        params["x0"] = np.asarray(x0s[0]).copy()
        params["xgoal"] = np.asarray(goals[0]).copy()
        self.set_params(params)
        self.set_instances(x0s, goals)

    def set_instances(self, x0s, goals=None):
        """(B,3) start states and (B,2) goals (goals=None keeps the current ones)."""
        x0s = np.ascontiguousarray(np.asarray(x0s, dtype=np.float64).astype(np.float32)).reshape(self.num_instances, 3)
        if goals is None:
            goals = self.goals
        goals = np.ascontiguousarray(np.asarray(goals, dtype=np.float64).astype(np.float32)).reshape(self.num_instances, 2)
        for b in range(self.num_instances):
            if not (self.is_within_bound(x0s[b, 0], self.lin_tdm.xlimits)
                    and self.is_within_bound(x0s[b, 1], self.lin_tdm.ylimits)):
                print("ERROR: instance {}: x0 is not within the map limits!".format(b))
                assert False
        self.x0s, self.goals = x0s, goals
        _lib.call("mppi_planner_set_instances", self._handle, self.num_instances,
                  _lib.ptr(x0s, C.c_float), _lib.ptr(goals, C.c_float))

    def check_solve_conditions(self):
        if self.x0s is None:
            print("Batch instances are not set. Cannot solve")
            return False
        return super().check_solve_conditions()

    # ------------------------------------------------------------------ control loop
    def shift_and_update(self, new_x0s, u_cur, num_shifts=1):
        """Per problem: x0 <- new_x0s[b]; u[b, :-k] = u[b, k:] (tail kept), uploaded."""
        self.set_instances(new_x0s)
        self.params["x0"] = np.asarray(new_x0s[0]).copy()
        shifted = np.array(u_cur, dtype=np.float32).reshape(self.num_instances, self.num_steps, 2)
        shifted[:, :-num_shifts] = shifted[:, num_shifts:].copy()
        self.set_u(shifted)

    def shift_and_update_on_device(self, new_x0s, num_shifts=1):
        self.set_instances(new_x0s)
        self.params["x0"] = np.asarray(new_x0s[0]).copy()
        _lib.call("mppi_planner_shift_u", self._handle, int(num_shifts))

    def shift_optimal_control_sequence(self, u_cur, num_shifts=1):
        shifted = np.array(u_cur, dtype=np.float32).reshape(self.num_instances, self.num_steps, 2)
        shifted[:, :-num_shifts] = shifted[:, num_shifts:].copy()
        self.set_u(shifted)

    def get_state_rollout(self, instance=0):
        """(V, T+1, 3) state sequences of one problem (see MPPI_Numba.get_state_rollout)."""
        self.move_mppi_task_vars_to_device()
        out = np.empty((self.num_vis_state_rollouts, self.num_steps + 1, 3), dtype=np.float32)
        _lib.call("mppi_planner_get_instance_state_rollout", self._handle, self.lin_tdm._handle,
                  self.ang_tdm._handle, int(instance), _lib.ptr(out, C.c_float))
        self._last_state_rollout = out
        return out.copy()
