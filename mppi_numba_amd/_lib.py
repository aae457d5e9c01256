"""ctypes binding of libmppi_hip.so (C ABI: include/mppi_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call
fails, the caller gets an exception.  Build with
`python -c "import __graft_entry__ as g; g.build()"` or
`make -C mppi_numba_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MPPI_HIP_LIB lets a developer point at an experimental build of the SAME library
LIB_PATH = os.environ.get("MPPI_HIP_LIB") or os.path.join(_HERE, "libmppi_hip.so")

MPPI_OK = 0
MODE_DET, MODE_SPEED_MAP, MODE_TDM, MODE_BAREBONE = 0, 1, 2, 3
RNG_PHILOX, RNG_XOROSHIRO = 0, 1
MATH_EXACT, MATH_FAST = 0, 1
COMM_ID_BYTES = 128
PREP_TDM, PREP_DET, PREP_SPEED = 0, 1, 2
DEBUG_NO_SPEC_KERNEL, DEBUG_NO_SPECULATION, DEBUG_NO_DEEP_KERNEL, DEBUG_CC_GLOBAL = 1, 2, 4, 8
DEBUG_KEEP_SPECULATING = 16
# MPPI_MATH_FAST, the time-parallel rollout kernel (include/mppi_hip.h)
DEBUG_NO_SCAN_KERNEL, DEBUG_SCAN_READ_NOISE, DEBUG_SCAN_FULL_TILES = 32, 64, 128
DEBUG_NO_FOLDED_APPLY = 256
DEBUG_NO_REDUCE_FOLD = 512
DEBUG_NO_SCAN_DIRECT = 1024
DEBUG_DROP_NOISE_FLAG = 2048
ABI_VERSION = 1
P2P_HANDLE_BYTES = 64


class MppiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libmppi_hip error %d: %s" % (code, message))
        self.code = code


class DeviceProps(C.Structure):
    _fields_ = [
        ("max_threads_per_block", C.c_int), ("max_block_dim_x", C.c_int),
        ("max_grid_dim_x", C.c_int), ("wavefront_size", C.c_int),
        ("compute_units", C.c_int), ("lds_bytes_per_cu", C.c_int),
        ("gcn_arch", C.c_char * 64), ("name", C.c_char * 128),
    ]


class TdmCfg(C.Structure):
    _fields_ = [
        ("device", C.c_int), ("num_grids", C.c_int), ("max_rows", C.c_int), ("max_cols", C.c_int),
        ("thread_dim_x", C.c_int), ("thread_dim_y", C.c_int), ("rng", C.c_int),
        ("_reserved", C.c_int), ("seed", C.c_uint64),
    ]


class PlannerCfg(C.Structure):
    _fields_ = [
        ("device", C.c_int), ("mode", C.c_int), ("num_control_rollouts", C.c_int),
        ("num_steps", C.c_int), ("num_grid_samples", C.c_int),
        ("num_vis_state_rollouts", C.c_int), ("rng", C.c_int), ("math", C.c_int),
        ("rank", C.c_int), ("world_size", C.c_int), ("num_instances", C.c_int),
        ("seed", C.c_uint64),
    ]


class Params(C.Structure):
    _fields_ = [
        ("x0", C.c_float * 3), ("xgoal", C.c_float * 2), ("vrange", C.c_float * 2),
        ("wrange", C.c_float * 2), ("u_std", C.c_float * 2),
        ("dt", C.c_float), ("goal_tolerance", C.c_float), ("v_post_rollout", C.c_float),
        ("lambda_weight", C.c_float), ("cvar_alpha", C.c_float), ("obs_cost", C.c_float),
        ("unknown_cost", C.c_float), ("res", C.c_float), ("xlo", C.c_float), ("ylo", C.c_float),
        ("dist_weight", C.c_double), ("alpha_dyn", C.c_double),
        ("num_opt", C.c_int), ("_reserved", C.c_int),
    ]


_i8p = C.POINTER(C.c_int8)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p

# name -> argtypes; every entry point of include/mppi_hip.h (restype int unless noted)
SIGNATURES = {
    "mppi_abi_version": [],
    "mppi_trace_ranges_enabled": [],
    "mppi_device_count": [C.POINTER(C.c_int)],
    "mppi_device_props_get": [C.c_int, C.POINTER(DeviceProps)],
    "mppi_tdm_create": [C.POINTER(TdmCfg), C.POINTER(_vp)],
    "mppi_tdm_destroy": [_vp],
    "mppi_tdm_set_maps": [_vp, _i8p, C.c_int, C.c_int, C.c_int, _i8p, C.c_double, C.c_double,
                          _i8p, _i8p, _i8p],
    "mppi_tdm_set_maps_from_pmf": [_vp, C.c_int, _i8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   _f32p, _f32p, C.c_double, _i8p, C.c_double, C.c_double, _i8p, _i8p,
                                   C.POINTER(C.c_int)],
    "mppi_tdm_get_maps": [_vp, _i8p, _i8p, _i8p, _i8p],
    "mppi_tdm_sample_grids": [_vp, C.c_double],
    "mppi_tdm_set_sampled_grids": [_vp, _i8p, C.c_int, C.c_int],
    "mppi_tdm_get_sampled_grids": [_vp, _i8p],
    "mppi_tdm_rng_states": [_vp, _u64p, C.c_long, C.POINTER(C.c_long)],
    "mppi_planner_create": [C.POINTER(PlannerCfg), C.POINTER(_vp)],
    "mppi_planner_destroy": [_vp],
    "mppi_planner_set_params": [_vp, C.POINTER(Params)],
    "mppi_planner_set_disc_obstacles": [_vp, _f32p, _f32p, C.c_int],
    "mppi_planner_set_u": [_vp, _f32p],
    "mppi_planner_get_u": [_vp, _f32p],
    "mppi_planner_get_u_prev": [_vp, _f32p],
    "mppi_planner_shift_u": [_vp, C.c_int],
    "mppi_planner_solve": [_vp, _vp, _vp, _f32p],
    "mppi_planner_iterate_async": [_vp, _vp, _vp, C.c_int],
    "mppi_planner_synchronize": [_vp],
    "mppi_planner_sample_noise": [_vp],
    "mppi_planner_set_noise": [_vp, _f32p],
    "mppi_planner_get_noise": [_vp, _f32p],
    "mppi_planner_rollout": [_vp, _vp, _vp],
    "mppi_planner_set_costs": [_vp, _f32p],
    "mppi_planner_get_costs": [_vp, _f32p],
    "mppi_planner_get_sample_costs": [_vp, _f32p],
    "mppi_planner_update": [_vp],
    "mppi_planner_get_weights": [_vp, _f32p],
    "mppi_planner_get_state_rollout": [_vp, _vp, _vp, _f32p],
    "mppi_planner_get_instance_state_rollout": [_vp, _vp, _vp, C.c_int, _f32p],
    "mppi_planner_set_instances": [_vp, C.c_int, _f32p, _f32p],
    "mppi_planner_rng_states": [_vp, _u64p, C.c_long, C.POINTER(C.c_long)],
    "mppi_planner_set_profiling": [_vp, C.c_int],
    "mppi_planner_stage_times": [_vp, _f32p],
    "mppi_planner_last_elapsed_ms": [_vp, _f32p],
    "mppi_planner_time_kernels": [_vp, _vp, _vp, C.c_int, _f32p, _f32p],
    "mppi_planner_describe_last_rollout": [_vp, C.c_char_p, C.c_int],
    "mppi_planner_set_debug_flags": [_vp, C.c_int],
    "mppi_planner_set_fold_poll_limit": [_vp, C.c_int],
    "mppi_planner_fold_state": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_long)],
    "mppi_debug_occupy_cus": [C.c_int, C.c_int, C.c_int],
    "mppi_selftest_philox": [C.c_int, C.POINTER(C.c_int)],
    "mppi_debug_read_stamps": [C.POINTER(C.c_ulonglong), C.c_int, C.c_int],
    "mppi_planner_graph_probe": [_vp, _vp, _vp, C.c_int, C.c_int, _f32p, _f32p],
    "mppi_planner_set_graph_replay": [_vp, C.c_int],
    "mppi_planner_graph_stats": [_vp, C.POINTER(C.c_long), C.POINTER(C.c_long)],
    "mppi_planner_p2p_export": [_vp, C.c_char_p],
    "mppi_planner_p2p_connect": [_vp, C.c_char_p, C.c_int],
    "mppi_group_p2p_connect": [C.POINTER(_vp), C.c_int],
    "mppi_planner_p2p_stats": [_vp, C.POINTER(C.c_int), C.POINTER(C.c_long), C.c_char_p, C.c_int],
    "mppi_planner_p2p_set_enabled": [_vp, C.c_int],
    "mppi_planner_p2p_ping": [_vp, C.c_ulonglong, C.c_int, C.POINTER(C.c_int)],
    "mppi_world_create": [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _f64p, _f64p, C.POINTER(_vp)],
    "mppi_world_destroy": [_vp],
    "mppi_world_get": [_vp, _f64p, C.c_int, _f64p, _f64p],
    "mppi_world_get_grids": [_vp, _f64p, _f64p],
    "mppi_world_sample_true_dist": [_vp, C.POINTER(C.c_int32), C.c_int, _f64p, _f64p, C.c_int, C.c_uint64],
    "mppi_planner_closed_loop": [_vp, _vp, _vp, _vp, C.c_int, C.c_double, C.c_double, _f64p, _f64p, _f32p, C.POINTER(C.c_int)],
    "mppi_tdm_set_sample_shard": [_vp, C.c_int],
    "mppi_planner_set_sample_sharding": [_vp, C.c_int, C.c_int],
    "mppi_planner_sample_costs_local": [_vp, _f32p],
    "mppi_planner_sample_costs_apply": [_vp, _f32p, C.c_int],
    "mppi_comm_unique_id": [C.c_char_p],
    "mppi_planner_comm_init": [_vp, C.c_char_p],
    "mppi_planner_comm_count": [_vp, C.POINTER(C.c_int)],
    "mppi_group_comm_init": [C.POINTER(_vp), C.c_int],
    "mppi_group_iterate_async": [C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.c_int, C.c_int],
    "mppi_planner_packet_len": [_vp, C.POINTER(C.c_int)],
    "mppi_planner_update_local": [_vp, _f64p],
    "mppi_planner_update_apply": [_vp, _f64p, C.c_int],
    "mppi_planner_update_apply_and_rollout": [_vp, _f64p, C.c_int, _vp, _vp],
}

_lib = None


def load():
    """Load libmppi_hip.so once; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: the HIP engine is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` at the repo root "
            "(or `make -C mppi_numba_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.mppi_last_error.restype = C.c_char_p
    lib.mppi_last_error.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.mppi_abi_version() != ABI_VERSION:
        raise ImportError("libmppi_hip.so ABI version %d != binding %d; rebuild"
                          % (lib.mppi_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc):
    if rc != MPPI_OK:
        raise MppiError(rc, load().mppi_last_error().decode("utf-8", "replace"))


def call(name, *args):
    check(getattr(load(), name)(*args))


def device_count():
    """Number of visible HIP devices; 0 (not an exception) on a box without GPUs."""
    n = C.c_int(0)
    rc = load().mppi_device_count(C.byref(n))
    return n.value if rc == MPPI_OK else 0


def device_props(device=0):
    pr = DeviceProps()
    call("mppi_device_props_get", device, C.byref(pr))
    return pr


def ptr(array, ctype):
    return array.ctypes.data_as(C.POINTER(ctype))
