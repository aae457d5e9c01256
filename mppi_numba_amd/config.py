#!/usr/bin/env python3
"""Config for MPPI_Numba / TDM_Numba on MI355X.

Mirrors /root/reference/mppi_numba/config.py (module globals config.py:9-14,
class Config config.py:16-100): same constructor keywords, same attributes,
same validation, clamping and messages, so `copy.deepcopy`, attribute
mutation and pickling of Config objects keep working.

Difference: the reference queries the GPU at import time (config.py:9).  Here
the query goes through the C ABI (mppi_device_props_get) and only happens if
a device is present; on a box without a GPU the gfx950 architectural limits
are used, so that importing this module and building a Config never needs a
GPU.  The values only steer clamps and messages, never arithmetic.
"""

# gfx950 (MI355X) launch limits; refreshed from the device when one is present
_GFX950_LIMITS = dict(max_threads_per_block=1024, max_block_dim_x=1024,
                      max_grid_dim_x=2 ** 31 - 1)


def _query_limits():
    limits = dict(_GFX950_LIMITS)
    try:
        from . import _lib
        if _lib.device_count() > 0:
            pr = _lib.device_props(0)
            limits = dict(max_threads_per_block=pr.max_threads_per_block,
                          max_block_dim_x=pr.max_block_dim_x,
                          max_grid_dim_x=pr.max_grid_dim_x)
    except (ImportError, OSError, AttributeError, RuntimeError):
        # Config-only use: library not built (ImportError / OSError), a stale build that lacks an
        # entry point (AttributeError from the symbol lookup), or a device query that fails
        # (MppiError, a RuntimeError).  The limits only steer clamps and messages.
        pass
    return limits


_limits = _query_limits()
max_threads_per_block = _limits["max_threads_per_block"]
max_square_block_dim = (int(_limits["max_block_dim_x"] ** 0.5), int(_limits["max_block_dim_x"] ** 0.5))
max_blocks = _limits["max_grid_dim_x"]
max_rec_blocks = rec_max_control_rollouts = 15000
rec_min_control_rollouts = 100


class Config:

    """ Configurations that are typically fixed throughout execution. """

    def __init__(self,
                 T=10,  # Horizon (s)
                 dt=0.1,  # Length of each step (s)
                 num_grid_samples=1024,  # Number of grid samples when sampling dynamics
                 num_control_rollouts=1024,  # Number of control sequences
                 max_speed_padding=5.0,  # Maximum assumed speed for padding the perimeter of grid
                 tdm_sample_thread_dim=(16, 16),  # Only shapes the xoroshiro-compatible sampler
                 num_vis_state_rollouts=20,  # Number of visualization rollouts
                 max_map_dim=(250, 250),  # Largest padded map (cells); anything bigger is cropped
                 seed=1,
                 use_tdm=False,
                 use_det_dynamics=False,
                 use_nom_dynamics_with_speed_map=False,
                 use_costmap=False,
                 # --- extensions (not in the reference) ---
                 enforce_recommended_limits=True,  # False lifts the [100, 15000] rollout clamp
                 rng="philox",  # "philox" (rocRAND) | "xoroshiro" (numba-compatible streams)
                 math="exact",  # "exact" (the reference CPU path's rounding points: bit-identical costs) | "fast" (tolerance mode: 99.9 % of the costs within 1e-6, faster everywhere)
                 device=0,
                 map_preprocessing="device",  # "device" (HIP kernel, csrc/map_kernels.h) | "host" (numpy, as the reference)
                 ):

        # planner variant: exactly one, and the costmap interface does not exist upstream either
        variants = dict(use_tdm=use_tdm, use_det_dynamics=use_det_dynamics,
                        use_nom_dynamics_with_speed_map=use_nom_dynamics_with_speed_map, use_costmap=use_costmap)
        for name, flag in variants.items():
            setattr(self, name, flag)
        assert T > 0 and dt > 0 and T > dt
        assert sum(bool(v) for v in variants.values()) == 1, \
            "MPPI Config Error: Only one of the {} can be true.".format(", ".join(variants))
        assert not use_costmap, "Interface with costmap2d is not yet implemented."

        self.seed = seed
        self.T, self.dt = T, dt
        self.num_steps = int(T / dt)
        assert self.num_steps > 0
        self.max_speed_padding = max_speed_padding
        self.max_map_dim = max_map_dim
        self.max_threads_per_block = max_threads_per_block

        # extensions
        assert rng in ("philox", "xoroshiro") and math in ("exact", "fast") and map_preprocessing in ("host", "device")
        self.enforce_recommended_limits = enforce_recommended_limits
        self.rng, self.math, self.device, self.map_preprocessing = rng, math, device, map_preprocessing

        self.num_grid_samples = self._grid_samples(num_grid_samples, enforce_recommended_limits)
        self.num_control_rollouts = self._control_rollouts(num_control_rollouts, enforce_recommended_limits)
        self.tdm_sample_thread_dim = self._sampler_block(tdm_sample_thread_dim)
        # rollouts kept for plotting: no more than there are control samples or sampled maps
        self.num_vis_state_rollouts = max(1, min(num_vis_state_rollouts, self.num_control_rollouts,
                                                 self.num_grid_samples))

    @staticmethod
    def _say(text, *values):
        print("MPPI Config: " + text.format(*values))

    @classmethod
    def _grid_samples(cls, wanted, clamp):
        """M: at least one map; the reference caps it at its recommended block count."""
        if wanted > max_threads_per_block:
            print("WARNING: slow-down expected since each thread needs to handle multiple grid samples due to "
                  "num_grid_samples({})>max_threads_per_block({})".format(wanted, max_threads_per_block))
        if clamp and wanted > max_rec_blocks:
            cls._say("Limit num_grid_samples by recommended max block number (<={}). But this can be overwritten if needed.",
                     max_rec_blocks)
            return max_rec_blocks
        if wanted < 1:
            cls._say("Set num_grid_samples from {} -> 1. Need at least 1 map to work with", wanted)
            return 1
        return wanted

    @classmethod
    def _control_rollouts(cls, wanted, clamp):
        """N: clipped to the reference's recommended [100, 15000] unless the clamp is lifted."""
        if clamp and wanted > rec_max_control_rollouts:
            cls._say("Clip num_control_rollouts to be recommended max number of {}. (Max={})",
                     rec_max_control_rollouts, max_blocks)
            wanted = rec_max_control_rollouts
        elif clamp and wanted < rec_min_control_rollouts:
            cls._say("Clip num_control_rollouts to be recommended min number of {}. (Recommended max={})",
                     rec_min_control_rollouts, rec_max_control_rollouts)
            wanted = rec_min_control_rollouts
        assert wanted >= 1
        return wanted

    @classmethod
    def _sampler_block(cls, shape):
        """2-D block of the grid sampler (only shapes the xoroshiro-compatible sampler here)."""
        shape = tuple(shape)
        assert len(shape) == 2 and shape[0] > 0 and shape[1] > 0
        threads = shape[0] * shape[1]
        if threads >= max_threads_per_block:
            cls._say("Requested {} threads per block (more than max {}) for sampling tdm. Change tdm_sample_thread_dim to {}",
                     threads, max_threads_per_block, max_square_block_dim)
            return max_square_block_dim
        return shape
