/*
 * mppi_hip.h -- C ABI of libmppi_hip.so, the MI355X (gfx950) MPPI rollout engine.
 *
 * The reference (mit-acl/mppi_numba) has no FFI: its device boundary is the
 * set of numba calls made by mppi_numba/mppi.py, terrain.py and config.py
 * (cuda.device_array / cuda.to_device / copy_to_host / kernel[grid, block](...)).
 * Each entry point below replaces one such group of calls; the reference
 * file:line it stands in for is given next to it (paths relative to
 * /root/reference/mppi_numba).  The Python host code binds this header with
 * ctypes (mppi_numba_amd/_lib.py); INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 (MPPI_OK) or a negative mppi_status; the text of
 *     the last failure on the calling thread is mppi_last_error();
 *   - handles are opaque and not thread-safe; one host thread drives a handle,
 *     as in the reference (single Python thread, default stream);
 *   - host arrays are owned by the caller, device memory by the library;
 *   - every call is synchronous from the caller's point of view unless its
 *     name ends in _async;
 *   - host-side array layouts are the reference's: noise (N,T,2) f32, u (T,2)
 *     f32, costs/weights (N) f32, sampled grids (G,R,C) int8, maps (Rp,Cp) int8,
 *     pmf (B,Rp,Cp) int8, state rollouts (V,T+1,3) f32, all C-contiguous.
 *     Device-side layouts are private to the library.
 */
#ifndef MPPI_HIP_H
#define MPPI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPPI_HIP_ABI_VERSION 1

typedef enum mppi_status {
  MPPI_OK = 0,
  MPPI_ERR_INVALID = -1,   /* bad argument / shape mismatch                   */
  MPPI_ERR_HIP = -2,       /* a HIP runtime call failed                        */
  MPPI_ERR_STATE = -3,     /* call sequence error (maps or params not set ...) */
  MPPI_ERR_NO_DEVICE = -4, /* no usable gfx950 device                          */
  MPPI_ERR_COMM = -5,      /* RCCL failure, or a peer of the peer exchange that did not deliver */
  MPPI_ERR_BUSY = -6       /* the device did not run all workgroups of a rollout launch side by side (another tenant, a
                              CU mask): the controls handed over inside that launch did not arrive in time; the control
                              sequence of the call is not valid, later calls update through a launch of their own */
} mppi_status;

/* which of the reference's solve_* variants the planner runs (mppi.py:193-211) */
typedef enum mppi_mode {
  MPPI_MODE_DET = 0,       /* use_det_dynamics                 mppi.py:308-375 */
  MPPI_MODE_SPEED_MAP = 1, /* use_nom_dynamics_with_speed_map  mppi.py:237-305 */
  MPPI_MODE_TDM = 2,       /* use_tdm (CVaR over M samples)    mppi.py:376-531 */
  MPPI_MODE_BAREBONE = 3   /* barebone_mppi_numba.ipynb: no maps, disc obstacles */
} mppi_mode;

typedef enum mppi_rng_kind {
  MPPI_RNG_PHILOX = 0,    /* rocRAND Philox4x32-10, counter keyed by the GLOBAL
                             (rollout, step) index: results do not depend on the
                             number of GPUs                                     */
  MPPI_RNG_XOROSHIRO = 1  /* bit-compatible with numba.cuda.random
                             (xoroshiro128+, stream per thread) for seed -> u
                             known-answer tests against the reference           */
} mppi_rng_kind;

typedef enum mppi_math_kind {
  MPPI_MATH_EXACT = 0, /* float64 trig/sqrt where the reference's CPU path has
                          float64: costs bit-identical to it (default)          */
  MPPI_MATH_FAST = 1   /* float32 sincosf/sqrtf in the general kernel: ~1 ulp cost
                          differences.  For numerical comparison; NOT faster -- the
                          pipelined / fused kernels exist for the exact path only */
} mppi_math_kind;

const char* mppi_last_error(void);
int mppi_abi_version(void);

/* ---- device query: replaces the import-time query of config.py:9-12 ------- */
typedef struct mppi_device_props {
  int max_threads_per_block;
  int max_block_dim_x;
  int max_grid_dim_x;
  int wavefront_size;
  int compute_units;
  int lds_bytes_per_cu;
  char gcn_arch[64];
  char name[128];
} mppi_device_props;

int mppi_device_count(int* count);
int mppi_device_props_get(int device, mppi_device_props* out);

/* ---- traction distribution map (TDM_Numba) -------------------------------- */
typedef struct mppi_tdm mppi_tdm;

typedef struct mppi_tdm_cfg {
  int device;
  int num_grids;        /* M for use_tdm, 1 for the deterministic modes
                           (terrain.py:171-177)                                 */
  int max_rows;         /* cfg.max_map_dim: allocation of sample_grid_batch     */
  int max_cols;
  int thread_dim_x;     /* cfg.tdm_sample_thread_dim (only shapes the xoroshiro
                           compatible stream->cell mapping, terrain.py:645-668) */
  int thread_dim_y;
  int rng;              /* mppi_rng_kind */
  int _reserved;
  uint64_t seed;
} mppi_tdm_cfg;

/* terrain.py:164-180 init_device_vars_before_sampling */
int mppi_tdm_create(const mppi_tdm_cfg* cfg, mppi_tdm** out);
int mppi_tdm_destroy(mppi_tdm* tdm);

/* terrain.py:331-333,370-371,405-406,495,506: upload of the padded PMF grid,
 * masks and (speed-map mode) padded risk traction map.  bin_to_int8[b] is
 * np.int8(100.*(bin_values[b]-lo)/(hi-lo)) (terrain.py:689) evaluated by the
 * host in the dtypes the reference would hold; traction_lo / traction_ratio are
 * bin_values_bounds[0] and 0.01*(bounds[1]-bounds[0]) (mppi.py:674-675).
 * risk may be NULL. */
int mppi_tdm_set_maps(mppi_tdm* tdm, const int8_t* pmf, int bins, int rows, int cols,
                      const int8_t* bin_to_int8, double traction_lo, double traction_ratio,
                      const int8_t* obstacle, const int8_t* unknown, const int8_t* risk);

/* Map preprocessing on the device (replaces the numpy code of
 * terrain.py:408-495 and the padding of terrain.py:511-583): the RAW PMF grid
 * (bins, src_rows, src_cols) and masks (src_rows, src_cols; NULL = zeros) in, on the
 * device: crop to (valid_rows, valid_cols) from the origin, ring of pad_cells
 * zero-traction cells, and per kind
 *   MPPI_PREP_TDM    the PMF itself                             (use_tdm)
 *   MPPI_PREP_DET    one-hot PMF at the CVaR_alpha bin          (use_det_dynamics)
 *   MPPI_PREP_SPEED  nominal PMF + int8 risk traction map       (use_nom_dynamics_with_speed_map)
 * bin_values / bounds are the float32 copies the reference holds (terrain.py:393-394);
 * the remaining arguments are those of mppi_tdm_set_maps.  *bad_columns (may be NULL)
 * receives the number of raw PMF columns that do not sum to 100.  Bit-identical to the
 * host path. */
typedef enum mppi_prep_kind { MPPI_PREP_TDM = 0, MPPI_PREP_DET = 1, MPPI_PREP_SPEED = 2 } mppi_prep_kind;
int mppi_tdm_set_maps_from_pmf(mppi_tdm* tdm, int kind, const int8_t* pmf, int bins, int src_rows,
                               int src_cols, int valid_rows, int valid_cols, int pad_cells,
                               const float* bin_values, const float bounds[2], double alpha,
                               const int8_t* bin_to_int8, double traction_lo, double traction_ratio,
                               const int8_t* obstacle, const int8_t* unknown, int* bad_columns);
/* the maps as held on the device; any pointer may be NULL.  pmf (bins, rows, cols),
 * obstacle / unknown / risk (rows, cols) with rows, cols the PADDED size */
int mppi_tdm_get_maps(mppi_tdm* tdm, int8_t* pmf, int8_t* obstacle, int8_t* unknown, int8_t* risk);

/* terrain.py:610-622 sample_grids (kernel terrain.py:633-695) */
int mppi_tdm_sample_grids(mppi_tdm* tdm, double alpha_dyn);

/* test/visualisation access to sample_grid_batch_d, host layout
 * (num_grids, max_rows, max_cols); set writes the [:, :rows, :cols] window */
int mppi_tdm_set_sampled_grids(mppi_tdm* tdm, const int8_t* grids, int rows, int cols);
int mppi_tdm_get_sampled_grids(mppi_tdm* tdm, int8_t* out);

/* xoroshiro-compatible generator only: copy the (streams, 2) uint64 states */
int mppi_tdm_rng_states(mppi_tdm* tdm, uint64_t* out, long capacity, long* count);

/* ---- planner (MPPI_Numba) -------------------------------------------------- */
typedef struct mppi_planner mppi_planner;

typedef struct mppi_planner_cfg {
  int device;
  int mode;                   /* mppi_mode */
  int num_control_rollouts;   /* N over ALL ranks                               */
  int num_steps;              /* T = int(cfg.T / cfg.dt)                        */
  int num_grid_samples;       /* M (1 unless MPPI_MODE_TDM)                     */
  int num_vis_state_rollouts; /* V                                              */
  int rng;                    /* mppi_rng_kind                                  */
  int math;                   /* mppi_math_kind                                 */
  int rank;                   /* this handle owns rollouts
                                 [rank*N/world, (rank+1)*N/world)               */
  int world_size;
  int num_instances;          /* batched multi-query extension (below): B
                                 problems in one handle; 0 or 1 = one problem   */
  uint64_t seed;
} mppi_planner_cfg;

/* mppi.py:214-234 move_mppi_task_vars_to_device: one struct instead of nine
 * to_device calls.  Fields the reference casts to np.float32 are float. */
typedef struct mppi_params {
  float x0[3];
  float xgoal[2];
  float vrange[2];
  float wrange[2];
  float u_std[2];
  float dt;
  float goal_tolerance;
  float v_post_rollout;
  float lambda_weight;
  float cvar_alpha;
  float obs_cost;
  float unknown_cost;
  float res;          /* lin_tdm.res                                  */
  float xlo;          /* float32(lin_tdm.padded_xlimits[0])           */
  float ylo;          /* float32(lin_tdm.padded_ylimits[0])           */
  double dist_weight; /* passed to the kernels as a Python number     */
  double alpha_dyn;   /* params['alpha_dyn'] (use_tdm), else 1.0      */
  int num_opt;
  int _reserved;
} mppi_params;

/* mppi.py:108-127 init_device_vars_before_solving */
int mppi_planner_create(const mppi_planner_cfg* cfg, mppi_planner** out);
int mppi_planner_destroy(mppi_planner* p);

int mppi_planner_set_params(mppi_planner* p, const mppi_params* params);

/* barebone notebook: params['obstacle_positions'] (K,2), ['obstacle_radius'] (K) */
int mppi_planner_set_disc_obstacles(mppi_planner* p, const float* positions, const float* radii,
                                    int count);

/* ---- batched multi-query (not in the reference; BASELINE.json configs[4]) ------
 * One handle solves B = cfg.num_instances independent MPPI problems per launch.
 * They share the maps and every field of mppi_params except the start state and
 * the goal; each has its own N = num_control_rollouts samples, control sequence,
 * minimum cost and normaliser.  Instance b is bit-identical to a single-instance
 * handle given the same noise.  Array shapes with B > 1: u (B,T,2), costs and
 * weights (B,N_local), noise (B*N_local,T,2), packets B*(2T+2) doubles per rank.
 * Needs N/world_size to be a multiple of 64; not available in MPPI_MODE_BAREBONE.
 * x0: (count,3) float32, xgoal: (count,2) float32; count must equal B.  A handle with B = 1
 * also takes count = 0 (x0, xgoal ignored): back to the start / goal of mppi_params. */
int mppi_planner_set_instances(mppi_planner* p, int count, const float* x0, const float* xgoal);

/* mppi.py:539-542 shift_optimal_control_sequence / mppi.py:305,375 copy_to_host */
int mppi_planner_set_u(mppi_planner* p, const float* u);
int mppi_planner_get_u(mppi_planner* p, float* u);
int mppi_planner_get_u_prev(mppi_planner* p, float* u);
/* device-side u[:-k] = u[k:] (tail kept), no host round trip */
int mppi_planner_shift_u(mppi_planner* p, int num_shifts);

/* mppi.py:186-531 solve(): samples both TDMs once, then num_opt x
 * {sample noise, rollout, update}; writes the (T,2) control sequence.
 * lin/ang are NULL in MPPI_MODE_BAREBONE. */
int mppi_planner_solve(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, float* u_out);

/* the same without sampling the TDMs and without the final copy: `iterations`
 * back-to-back {noise, rollout, update} on the planner's stream */
int mppi_planner_iterate_async(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, int iterations);
int mppi_planner_synchronize(mppi_planner* p);

/* stage-level entry points (the individual kernel launches of mppi.py:259-303),
 * used by the parity tests to inject identical noise / grids / costs */
int mppi_planner_sample_noise(mppi_planner* p);                              /* mppi.py:1354 */
int mppi_planner_set_noise(mppi_planner* p, const float* noise);             /* (N_local,T,2) */
int mppi_planner_get_noise(mppi_planner* p, float* noise);
int mppi_planner_rollout(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang);     /* mppi.py:613-1111 */
int mppi_planner_set_costs(mppi_planner* p, const float* costs);             /* (N_local) */
int mppi_planner_get_costs(mppi_planner* p, float* costs);
int mppi_planner_get_sample_costs(mppi_planner* p, float* costs);            /* (N_local,M), TDM mode */
int mppi_planner_update(mppi_planner* p);                                    /* mppi.py:1113-1191 */
int mppi_planner_get_weights(mppi_planner* p, float* weights);               /* normalised, (N_local) */

/* mppi.py:545-608 get_state_rollout -> (V, T+1, 3) */
int mppi_planner_get_state_rollout(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, float* out);
/* the same for problem `instance` of a batched handle (get_state_rollout = instance 0) */
int mppi_planner_get_instance_state_rollout(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, int instance,
                                            float* out);

/* xoroshiro-compatible generator only: copy the (N_local*T, 2) uint64 states */
int mppi_planner_rng_states(mppi_planner* p, uint64_t* out, long capacity, long* count);

/* hipEvent timing (planner's stream) of the stages of the LAST iteration of the
 * last solve/iterate call, in ms; recorded only while profiling is enabled:
 * [0] noise  [1] rollout  [2] update (weights + weighted sum + apply)  [3] collective */
int mppi_planner_set_profiling(mppi_planner* p, int enabled);
int mppi_planner_stage_times(mppi_planner* p, float ms[4]);
/* (new; measurement) average duration in microseconds of the rollout launch and of the update
 * launch over `reps` ordinary iterations of the loop (iterate_async): every launch carries its
 * own start / stop events, filled by the runtime with the dispatch's begin / end timestamps --
 * the durations rocprofv3 --kernel-trace reports -- with nothing inserted between the kernels
 * (bench.py's roofline figures). */
int mppi_planner_time_kernels(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, int reps, float* us_rollout,
                              float* us_update);
/* GPU time (ms, hipEvents on the planner's stream) of the last iterate_async call (of the
 * last solve too while profiling is enabled; solve() does not time itself otherwise) */
int mppi_planner_last_elapsed_ms(mppi_planner* p, float* ms);

/* tracing hook: with MPPI_ROCTX=1 in the environment the library brackets solve / sample_grids /
 * noise / rollout / exchange / update / closed_loop with roctx ranges (names "mppi:...") for
 * `rocprofv3 --marker-trace --kernel-trace`; returns 1 when the ranges are live, else 0 */
int mppi_trace_ranges_enabled(void);

/* diagnostic: which rollout kernel variant (and its launch geometry) the last rollout used */
int mppi_planner_describe_last_rollout(mppi_planner* p, char* buf, int capacity);

/* diagnostic: compares the library's single-block Philox4x32-10 with rocRAND's engine on
 * 65536 (seed, subsequence, offset) triples; *mismatches must come back 0 */
/* developer switches for the parity tests, which pin every rollout kernel variant by name
 * (mppi_planner_describe_last_rollout).  With MPPI_MATH_EXACT the costs never depend on them (every
 * variant has the reference's bits); u agrees to float32 resolution between kernel families (the
 * update's float64 summation tree follows the rollout kernel's tiles of 32 or 64 rollouts: both sit
 * far inside the 1e-5 the reference's own unordered float32 atomics allow).  With MPPI_MATH_FAST they select between kernels that agree to
 * float32 tolerance only -- and so does whatever else decides which kernel runs there: N, the map
 * (a failed traction vote re-runs a tile sequentially; a map that keeps failing switches the
 * speculative kernels off until it changes). */
#define MPPI_DEBUG_NO_SPEC_KERNEL 1  /* retired (round 6: k_rollout_spec removed, profiles/r06_families.md); accepted, no effect */
#define MPPI_DEBUG_NO_SPECULATION 2  /* the speculative kernels on their exact schedule from the first step */
#define MPPI_DEBUG_NO_DEEP_KERNEL 4  /* retired (round 6: k_rollout_deep removed); accepted, no effect */
#define MPPI_DEBUG_CC_GLOBAL 8       /* control-cost products in the global scratch array even when LDS has room */
#define MPPI_DEBUG_KEEP_SPECULATING 16 /* keep the speculative kernels on a map where their tiles keep falling back */
/* MPPI_MATH_FAST, the time-parallel rollout (k_rollout_scan): */
#define MPPI_DEBUG_NO_SCAN_KERNEL 32     /* not the time-parallel kernels: k_rollout_pipe / k_rollout_fused / k_rollout_map instead */
#define MPPI_DEBUG_SCAN_READ_NOISE 64    /* the iteration loop stores its noise and the kernel reads it (as the stage-level calls do) */
#define MPPI_DEBUG_SCAN_FULL_TILES 128   /* workgroups of 64 rollouts (one lane per rollout) instead of 32 (two) */
#define MPPI_DEBUG_NO_FOLDED_APPLY 256   /* sharded handle: every iteration's update by its own k_apply launch, never by the next rollout launch */
#define MPPI_DEBUG_NO_REDUCE_FOLD 512    /* one GPU: every iteration's update by its own k_combine_tiles launch, never reduced and applied by the next rollout launch */
#define MPPI_DEBUG_NO_SCAN_DIRECT 1024   /* a map the planner has stopped speculating on: k_rollout_pipe + k_update_rows (round 4) instead of k_rollout_scan_exact on its exact schedule (one launch per iteration) */
#define MPPI_DEBUG_DROP_NOISE_FLAG 2048  /* test hook: the noise generator on the second stream does not announce itself -- the launch that waits for it gives up after ~60 ms, the next draining call returns MPPI_ERR_BUSY and the handle orders its streams with events from then on */
int mppi_planner_set_debug_flags(mppi_planner* p, int flags);
/* How long a workgroup of a rollout launch polls for the controls its sibling workgroups publish inside the launch
 * (update folded into the next rollout launch: rollout_scan*_kernel.h) before it gives the launch up: `polls` of
 * ~0.3-1 us each (default 2^20, about a second; test hook).  After a give-up the next call that waits for the handle's
 * stream -- synchronize, solve, and every stage-level call that returns data (get_u, get_costs, rollout, update ...) --
 * returns MPPI_ERR_BUSY and the handle stops folding (mppi_planner_fold_state: folding, faults so far).  Calling
 * mppi_planner_set_fold_poll_limit again re-arms the handle: it folds again (and gives up again if the device is still
 * shared). */
int mppi_planner_set_fold_poll_limit(mppi_planner* p, int polls);
int mppi_planner_fold_state(mppi_planner* p, int* folding, long* faults);
/* developer / test hook: occupy `workgroups` compute units of `device` (one workgroup each: 100 KiB of LDS) for
 * `milliseconds` on a stream of its own, so that a rollout launch beside it cannot have all its workgroups resident */
int mppi_debug_occupy_cus(int device, int workgroups, int milliseconds);
int mppi_selftest_philox(int device, int* mismatches);
/* developer instrumentation: in-kernel clock stamps of a -DMPPI_STAMPS build
 * (csrc/Makefile target `stamps`, tools/stamp_timeline.py); MPPI_ERR_STATE otherwise */
int mppi_debug_read_stamps(unsigned long long* out, int count, int clear);
/* hipGraph replay of the iteration loop (off by default).  A sharded handle needs its RCCL
 * communicator first: the all-gather is captured with the kernels.
 * iterations_per_graph: 0 = off, else an even number (the noise double buffer must come
 * back to where it was).  When on, every iteration also produces the noise of its
 * successor, that many iterations are captured into a graph the first time and replayed
 * for as long as nothing a kernel argument carries has changed (parameters, maps, start
 * state of a single-problem handle ...); the Philox call epoch then lives in device
 * memory.  Results are identical to the direct loop.  Worth it for loops of many
 * iterations (params.num_opt, iterate_async); a handle whose start state changes every
 * call re-captures every call unless it is a batched handle (set_instances), whose
 * start / goal live in device memory. */
int mppi_planner_set_graph_replay(mppi_planner* p, int iterations_per_graph);
int mppi_planner_graph_stats(mppi_planner* p, long* captures, long* replays);
/* developer measurement: wall microseconds per iteration of `iterations` (even) x {noise,
 * rollout, update}, launched directly vs replayed from a captured hipGraph, `replays`
 * times each.  Replays reuse the captured arguments: a measurement, not a way to plan. */
int mppi_planner_graph_probe(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, int iterations, int replays,
                             float* us_direct, float* us_graph);

/* ---- the simulated world of the closed-loop demo, on the device ------
 * mppi_world replaces TractionGrid (terrain.py:750-785): float64 (rows, cols) grids of linear and
 * angular traction, resolution and lower limits as in its constructor (terrain.py:754-773).
 *   create            TractionGrid.__init__; lin / ang may be NULL (zeros) when the grids are drawn
 *                     on the device afterwards
 *   get               TractionGrid.get (terrain.py:776-782) for `count` points xy[count][2]:
 *                     the cell's (lin, ang), (0, 0) outside the grid; Python's float // rule
 *   get_grids         TractionGrid.get_grids (terrain.py:784-785)
 *   sample_true_dist  TDM_Numba.sample_grids_true_dist (terrain.py:586-608): every cell draws from
 *                     the densities of its terrain type, linear and angular independently.  The
 *                     densities are host objects; the device draws uniformly (Philox4x32-10 keyed by
 *                     seed, counter = cell index and call number) from the pools of samples every
 *                     Terrain keeps (terrain.py:44-46): lin_pool / ang_pool [n_terrains][pool_len],
 *                     terrain_of_cell [rows*cols] = row of the pools for each cell.               */
typedef struct mppi_world mppi_world;
int mppi_world_create(int device, int rows, int cols, double res, double xlo, double ylo, const double* lin,
                      const double* ang, mppi_world** out);
int mppi_world_destroy(mppi_world* w);
int mppi_world_get(mppi_world* w, const double* xy, int count, double* lin_out, double* ang_out);
int mppi_world_get_grids(mppi_world* w, double* lin, double* ang);
int mppi_world_sample_true_dist(mppi_world* w, const int32_t* terrain_of_cell, int n_terrains, const double* lin_pool,
                                const double* ang_pool, int pool_len, uint64_t seed);
/* The notebooks' closed loop (test.ipynb cell 4: solve -> TractionGrid.get -> float64 Euler step ->
 * shift_and_update(x_new, useq, 1) -> goal check) for every problem of a batched handle
 * (set_instances, count >= 1), max_steps times or until every problem is within goal_tolerance of
 * its goal, without a host round trip per control step: the new start state, the LDS window
 * origin and the shifted control sequence are written by a kernel on the planner's stream.
 *   dt     the world's Euler step (cfg.dt as the notebook uses it, float64); <= 0: params.dt
 *   x_init [B][3] float64 start states (NULL: the instances' float32 start states)
 *   xhist  [B][max_steps+1][3] float64, row 0 = start, rows never reached = NaN
 *   uhist  [B][max_steps][2]   float32, the control applied at every step (NaN likewise)
 *   steps_taken [B]            steps until the goal test passed (or the number of steps run)
 * Afterwards the handle is where the notebook's loop would have left it (start states, shifted u). */
int mppi_planner_closed_loop(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, mppi_world* w, int max_steps,
                             double dt, double goal_tolerance, const double* x_init, double* xhist, float* uhist,
                             int* steps_taken);

/* ---- multi-GPU: N sharded over ranks, one RCCL all-gather of (2T+2) floats
 *      per iteration (not in the reference) ------------------------------------ */
#define MPPI_COMM_ID_BYTES 128
int mppi_comm_unique_id(char id[MPPI_COMM_ID_BYTES]);
int mppi_planner_comm_init(mppi_planner* p, const char id[MPPI_COMM_ID_BYTES]);
/* ranks RCCL itself reports for the handle's communicator (ncclCommCount); 0 without one */
int mppi_planner_comm_count(mppi_planner* p, int* ranks);
/* The peer exchange: the same sharding without a collective on the iteration's path.  Every rank owns an inbox
 * in fine-grained device memory; the workgroup that has combined the local tiles for step t of an update writes the
 * rank's four numbers for that step straight into every rank's inbox (peer access inside one process, IPC-mapped
 * memory across processes: xGMI on a multi-GPU node) and waits for the others' in its own -- inside the next
 * rollout launch, or inside the update launch that closes a loop.  A sharded iteration of the time-parallel exact
 * kernel (deterministic dynamics, one round of tiles: BASELINE configs[1] per GPU) is then ONE launch; u has the bits
 * of the all-gather + k_apply path.  Handles the kernels cannot serve keep using the communicator.
 *   p2p_export   this rank's inbox as an opaque handle (hipIpcMemHandle_t) for the other processes
 *   p2p_connect  the handles of all `count` = world_size ranks, in rank order (the own one is ignored)
 *   group_p2p_connect  one process driving `count` devices: planners[g] is rank g
 *   p2p_stats    connected?, exchanges performed so far, how the inbox was allocated
 * A rank whose peers stop sending (a dead process) does not hang: after a few seconds of polling (MPPI_P2P_MAX_POLLS) its kernels raise
 * a fault word and run on without waiting again; mppi_planner_solve / mppi_planner_synchronize then return
 * MPPI_ERR_COMM (the control sequence of that call is not valid; reconnect before using the exchange again). */
#define MPPI_P2P_HANDLE_BYTES 64
int mppi_planner_p2p_export(mppi_planner* p, char handle[MPPI_P2P_HANDLE_BYTES]);
int mppi_planner_p2p_connect(mppi_planner* p, const char* handles, int count);
int mppi_group_p2p_connect(mppi_planner** planners, int count);
int mppi_planner_p2p_stats(mppi_planner* p, int* connected, long* exchanges, char* kind, int capacity);
/* all ranks together, after connecting: every rank writes `token` (neither 0 nor all ones, the same on all ranks, new for every
 * call) into every inbox and waits up to ~timeout_ms for the others' -- without trapping.  *heard == world_size: peer
 * stores reach this rank's running kernels, the exchange can be trusted; otherwise switch it off on ALL ranks. */
int mppi_planner_p2p_ping(mppi_planner* p, unsigned long long token, int timeout_ms, int* heard);
/* a connected exchange switched off and on again (all ranks alike; off: the handle's communicator is used) */
int mppi_planner_p2p_set_enabled(mppi_planner* p, int enabled);
/* One process driving `count` devices: planners[g] is rank g of `count` on its own device.
 * comm_init creates all communicators inside one RCCL group; iterate_async runs `iterations` x
 * {per device: noise, rollout, shard packet | one group of all-gathers | per device: apply} and
 * returns without waiting (mppi_planner_synchronize each handle).  Replaces, for a multi-GPU
 * planner, the single-device numba context of the reference (config.py:9, mppi.py:186-303). */
int mppi_group_comm_init(mppi_planner** planners, int count);
int mppi_group_iterate_async(mppi_planner** planners, mppi_tdm** lins, mppi_tdm** angs, int count, int iterations);
/* host-staged alternative to RCCL (the exchange itself is then done by the
 * caller, e.g. over gloo): packet = {beta, den, num[T][2]} of the local shard,
 * 2T+2 doubles; update_apply takes the packets of all ranks in rank order */
int mppi_planner_packet_len(mppi_planner* p, int* doubles);
/* ---- multi-GPU, CVaR mode: the M traction-map samples sharded over ranks ------
 * Rank r of G rolls ALL N control samples (rollout_numba, mppi.py:613-755) over traction samples
 * [r*M/G, (r+1)*M/G): its TDMs draw exactly those samples of the unsharded set
 * (mppi_tdm_set_sample_shard: first sample, even; Philox generator), its planner is created with
 * num_grid_samples = M/G, world_size = 1 and told its place (mppi_planner_set_sample_sharding,
 * before comm_init).  Per iteration ONE RCCL all-gather of the (N, M/G) float32 per-sample costs
 * (N*M*4 bytes in total); every rank then sorts / averages all M costs of every control sample
 * (the reference's order, mppi.py:716-755: costs have the bits of the unsharded launch) and runs
 * the control update locally -- all ranks hold the same u without a second collective.
 * sample_costs_local / sample_costs_apply are the host-staged form of the exchange (after
 * mppi_planner_rollout; slabs: (G, N, M/G) in rank order), followed by mppi_planner_update. */
int mppi_tdm_set_sample_shard(mppi_tdm* t, int first_sample);
int mppi_planner_set_sample_sharding(mppi_planner* p, int rank, int count);
int mppi_planner_sample_costs_local(mppi_planner* p, float* slab);
int mppi_planner_sample_costs_apply(mppi_planner* p, const float* slabs, int count);
int mppi_planner_update_local(mppi_planner* p, double* packet);
int mppi_planner_update_apply(mppi_planner* p, const double* packets, int count);
/* update_apply followed by the NEXT iteration's rollout (its noise already sampled) as one call: when
 * that rollout is one of the time-parallel kernels its launch forms u from the packets itself (every
 * wave the 8 steps it owns, k_apply's expressions and order: same bits) and no k_apply is launched --
 * what mppi_planner_iterate_async does between the iterations of a sharded handle (a sharded iteration
 * is then rollout, rank packet, all-gather: three launches instead of four).  Otherwise exactly
 * update_apply + rollout. */
int mppi_planner_update_apply_and_rollout(mppi_planner* p, const double* packets, int count, mppi_tdm* lin,
                                          mppi_tdm* ang);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_HIP_H */
