// rollout_kernels.h -- the integrator kernels (gfx950, wave64).
//
// Replaces (reference paths relative to /root/reference/mppi_numba):
//   rollout_det_dyn_numba              mppi.py:916-1009   -> k_rollout_pipe (latency regime),
//                                                             k_rollout_fused (throughput regime),
//                                                             k_rollout_map<DET> (general)
//   rollout_det_dyn_w_speed_map_numba  mppi.py:1013-1111  -> k_rollout_fused<.., SPEED>, k_rollout_map<SPEED>
//   rollout_numba / rollout_oversized  mppi.py:613-913    -> k_rollout_tdm_fast, k_rollout_tdm
//   barebone rollout_numba             barebone_mppi_numba.ipynb cell 3 -> k_rollout_barebone
//   get_state_rollout_* kernels        mppi.py:1194-1351  -> k_state_rollout<..>
// (which one runs: launch_rollout_t in mppi_api.hip)
//
// Device data layout (private to the library):
//   noise  [N/64][T][64] float2   tile-major (device_math.h tile_index): lane n of a
//                             wave reads 8 contiguous bytes next to its neighbours
//                             at every step, consecutive steps are adjacent
//                             512-byte rows, and the weighted sum over n of the
//                             update still streams 64-element runs
//   cells  [Rp*Cp] uint32     one word per map cell: lin | ang<<8 | obs<<16 |
//                             unk<<24 (int8 each) -> ONE gather per step
//   cellsM [Rp*Cp][M] uint32  the same per traction sample, sample index
//                             fastest: the M lanes that evaluate one control
//                             sequence sit in the same or neighbouring cells, so
//                             a wave's gather touches a handful of 256-byte
//                             runs instead of 64 cache lines (M,R,C layout)
//   u [T] float2, costs [N] float
//
// One rollout (or one (rollout, sample) pair) per lane; the cost stays in a
// register and is written once (the reference does three global RMWs per step).
#pragma once
#include "device_math.h"
#include "rng_kernels.h"
#include "update_kernels.h"

namespace mppi {

enum MapKind { MAP_DET = 0, MAP_SPEED = 1 };

struct DevParams {
  float x0, y0, th0;
  float xg, yg;
  float v_lo, v_hi, w_lo, w_hi;
  float dt, gt2, lambda;
  float obs_cost, unk_cost;
  float res, inv_res, xlo, ylo;
  float cvar_alpha;
  int numel;          // ceil(M * float32(alpha)) evaluated in float64 (mppi.py:743)
  double dist_weight;
  double v_post_den;  // float64(v_post_rollout) + 1e-6            (mppi.py:28)
  double lin_lo, lin_ratio, ang_lo, ang_ratio;
  double s0sq, s1sq;  // u_std**2 in float64                       (mppi.py:709)
  int n_local;        // rollouts on this GPU
  int n_steps;        // T
  int n_grids;        // M
  int rows, cols;     // padded map dims Rp, Cp
  int n_obstacles;    // barebone only
  // k_rollout_fused, exact mode (round 6): the noise of the first `stash_steps` steps (a multiple of 8) is kept in the
  // LDS the window leaves free, `stash_offset` bytes into the dynamic LDS, [wave][step][64] float2, for the control-cost
  // pass -- which otherwise reads all of it from memory again, every wave at the same time (0: nothing is kept)
  int stash_steps, stash_offset;
  // LDS-resident window of the 16-bit cell map (deterministic modes)
  int win_r0, win_c0;      // first row / column of the window (column multiple of 8)
  int win_rows, win_cols;  // window size; win_cols is a multiple of 8
  int pitch16;             // row pitch of the global 16-bit cell array (multiple of 8)
  int lin_max_byte, ang_max_byte;  // host-side bounds only (largest traction byte in the grids)
  // (host-side window planning; rounds 2-5: a kernel that copied the window in bands of rows) cells a rollout can
  // cover per step (dt * max|v| * max traction / res); win_progressive = 0: everything up front
  float win_step_cells;
  int win_progressive;
  // the traction byte b with lin_lo + lin_ratio*b == 0.0 (a rollout that enters such a cell never
  // moves again), or -1 when there is none
  int lin_zero_byte;
  // batched multi-query handle: per-problem start / goal / window origin, else nullptr
  const struct BatchInst* inst;
  int inst_tiles;  // tiles of 64 rollouts per problem
  int n_inst;      // rollouts per problem on this GPU
  // speculative kernels: tiles whose assumption about the traction failed (host-mapped counter the
  // host looks at when it synchronises anyway: a map on which speculation does not pay gets the
  // exact pipelined kernel from then on), or nullptr
  unsigned int* spec_failures;
  // k_rollout_scan: host-side quotients (float64 divisions off the device)
  float cc_k0, cc_k1;              // lambda / u_std^2
  double inv_v_post_den;           // 1 / (v_post_rollout + 1e-6)
  double neg_log2e_over_lambda;    // -log2(e) / lambda
  // mppi_planner_time_kernels only (else nullptr): when every wave of this launch entered ([0, ktime_waves)) and left
  // ([ktime_waves, 2 ktime_waves)) on the device's constant 100 MHz clock -- for the kernels of the throughput regime,
  // whose loop runs on two streams and cannot be timed with events (see there)
  unsigned long long* ktime;
  int ktime_waves;
  // Round 6: this launch's noise was generated on the planner's second stream.  Rounds 1-5 ordered the two with a
  // cross-stream wait in front of the launch -- a barrier packet that cost the loop ~6 us per iteration at N = 65536
  // although the generator had finished long before (profiles/r06_ns_notes.md).  Instead a one-thread kernel behind
  // the generator stores its sequence number here, and the rollout's waves look at it before their first noise load:
  // in the steady state one load.  nullptr: ordered by the stream (or by the event wait, launch_plan.h: settle_noise_wait).
  const unsigned long long* noise_flag;
  unsigned long long noise_flag_expect;
  // ... and the other direction: the generator of the NEXT iteration's noise may start once this launch has (everything
  // in front of it on the stream -- the previous update, the last reader of the buffer the generator overwrites -- is
  // complete then).  The first wave of the launch stores `progress_value` here; a one-wave gate kernel in front of the
  // generator on the second stream waits for it (k_wait_progress).  Rounds 1-5: an event recorded in front of this
  // launch -- a marker that held the launch back by ~6 us (profiles/r06_ns_notes.md: timeline).
  unsigned long long* progress;
  unsigned long long progress_value;
  // Both waits are bounded and fail soft: a wave that has polled for ~60 ms (the rollout's) / ~4 s (the gate's) raises this
  // host-mapped word and the launch ends without its results; the call that waits for the stream next returns
  // MPPI_ERR_BUSY and the handle orders its streams with events from then on.  (What it takes: a tool that runs one kernel
  // at a time -- rocprofv3 --pmc does -- so that the generator cannot run beside the launch that waits for it.)
  unsigned int* flag_fault;
};

// What differs between the problems of a batched handle (mppi_planner_set_instances).
struct BatchInst {
  float x0, y0, th0;
  float xg, yg;
  int win_r0, win_c0;  // the LDS window is planned around each start state
  int pad;
};

// every wave of a workgroup, before its first load of the noise (see DevParams::noise_flag).  Every wave looks once; when
// the generator has not finished yet ONE wave per workgroup polls -- thousands of waves on one word would be a hot spot of
// their own -- and the others wait at the barrier.  false: the generator did not finish within the bound (flag_fault raised):
// the caller leaves the kernel.
__device__ __forceinline__ bool wait_for_noise(const DevParams& P) {
  if (P.noise_flag == nullptr) return true;  // (uniform over the launch)
  __shared__ int gave_up;
  const bool ready = __builtin_amdgcn_readfirstlane(
      (int)(__hip_atomic_load(P.noise_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P.noise_flag_expect)) != 0;
  if (threadIdx.x < 64) {
    bool there = ready;
    for (unsigned int polls = 0; !there && polls < (1u << 16); ++polls) {
      __builtin_amdgcn_s_sleep(32);
      there = __builtin_amdgcn_readfirstlane(
                  (int)(__hip_atomic_load(P.noise_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= P.noise_flag_expect)) != 0;
    }
    if (threadIdx.x == 0) {
      gave_up = there ? 0 : 1;
      if (!there && P.flag_fault) __hip_atomic_store(P.flag_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  __syncthreads();
  if (gave_up) return false;
  // the generator finished while this kernel was running: nothing of its output may be served from this kernel's caches
  // (a wave that saw the flag at its first look needs nothing: the kernel's own start made those stores visible)
  if (!ready) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return true;
}

__global__ void k_set_noise_flag(unsigned long long* flag, unsigned long long value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// first thing in a launch that carries DevParams::progress
__device__ __forceinline__ void signal_progress(const DevParams& P) {
  if (P.progress && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(P.progress, P.progress_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave on the second stream, in front of the generator: returns when the main stream's launch number `value` has
// started.  Bounded (a main stream that never gets there is an error the host reports elsewhere; the generator then
// simply runs).
__global__ __launch_bounds__(64) void k_wait_progress(const unsigned long long* progress, unsigned long long value,
                                                      unsigned int* flag_fault) {
  for (unsigned int polls = 0; polls < (1u << 24); ++polls) {
    if (__builtin_amdgcn_readfirstlane((int)(__hip_atomic_load(progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= value))) return;
    __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0 && flag_fault) __hip_atomic_store(flag_fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (one slot per wave, plain stores: atomics of a few thousand waves on one address took longer than the kernel)
__device__ __forceinline__ void ktime_begin(const DevParams& P) {
  if (P.ktime && (threadIdx.x & 63) == 0)
    P.ktime[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (unsigned long long)wall_clock64();
}
__device__ __forceinline__ void ktime_end(const DevParams& P) {
  if (P.ktime && (threadIdx.x & 63) == 0)
    P.ktime[P.ktime_waves + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = (unsigned long long)wall_clock64();
}

// Problem b of a batched handle: patch the by-value parameters and return its control
// sequence.  b is uniform over the workgroup, so these are scalar loads; a single-problem
// handle (inst == nullptr) pays one uniform branch.
__device__ __forceinline__ const float2* select_instance(DevParams& P, const float2* __restrict__ u, int b) {
  if (P.inst == nullptr) return u;
  const BatchInst I = P.inst[b];
  P.x0 = I.x0; P.y0 = I.y0; P.th0 = I.th0;
  P.xg = I.xg; P.yg = I.yg;
  P.win_r0 = I.win_r0; P.win_c0 = I.win_c0;
  return u + (size_t)b * P.n_steps;
}

// u[t]/std^2 (float64) for the control-cost term, staged in LDS once per block
__device__ __forceinline__ void stage_control_ratios(const DevParams& P, const float2* __restrict__ u,
                                                     double2* uos) {
  for (int t = threadIdx.x; t < P.n_steps; t += blockDim.x) {
    float2 ut = u[t];
    uos[t] = make_double2((double)ut.x / P.s0sq, (double)ut.y / P.s1sq);
  }
  __syncthreads();
}

// lambda*((u0/s0^2)*e0 + (u1/s1^2)*e1)                           (mppi.py:708-710)
__device__ __forceinline__ double control_cost(const DevParams& P, double2 r, float2 e) {
  return (double)P.lambda * fma(r.x, (double)e.x, r.y * (double)e.y);
}

struct StepOut {
  float x, y, th;
  double d2;
};

// One Euler step with traction (mppi.py:977-992): float64 products, float32 stores.
template <bool EXACT, bool BOUNDED = false>
__device__ __forceinline__ StepOut unicycle_step(const DevParams& P, float x, float y, float th, float v,
                                                 float w, int lin, int ang) {
  StepOut o;
  if (EXACT) {
    double s, c;
    sincos_f64<BOUNDED>((double)th, s, c);
    double q = (double)P.dt * (double)v;  // exact: two float32 factors
    double vtr = fma(P.lin_ratio, (double)lin, P.lin_lo);
    double wtr = fma(P.ang_ratio, (double)ang, P.ang_lo);
    o.x = (float)fma(vtr, q * c, (double)x);
    o.y = (float)fma(vtr, q * s, (double)y);
    o.th = (float)fma(wtr, (double)P.dt * (double)w, (double)th);
    double dx = (double)(P.xg - o.x), dy = (double)(P.yg - o.y);
    o.d2 = fma(dx, dx, dy * dy);
  } else {
    float s, c;
    sincosf(th, &s, &c);
    float vtr = fmaf((float)P.lin_ratio, (float)lin, (float)P.lin_lo);
    float wtr = fmaf((float)P.ang_ratio, (float)ang, (float)P.ang_lo);
    float q = P.dt * v * vtr;
    o.x = fmaf(q, c, x);
    o.y = fmaf(q, s, y);
    o.th = fmaf(P.dt * wtr, w, th);
    float dx = P.xg - o.x, dy = P.yg - o.y;
    o.d2 = (double)fmaf(dx, dx, dy * dy);
  }
  return o;
}

// cost += stage_cost(d2, step_time, dist_weight)  (mppi.py:20-22, 994): the sum is
// float64, the store float32.
template <bool EXACT>
__device__ __forceinline__ float add_stage_cost(const DevParams& P, float cost, double d2, double step_time) {
  if (EXACT) return (float)((double)cost + fma(P.dist_weight, sqrt_newton_f64(d2), step_time));
  return cost + fmaf((float)P.dist_weight, sqrtf((float)d2), (float)step_time);
}

__device__ __forceinline__ int clamp_index(int i, int n) { return min(max(i, 0), n - 1); }

// -------------------------------------------------------------------------
// Deterministic-dynamics rollouts: one control sample per lane, 64-lane
// workgroups so that small N still spreads over the CUs.
// Cost order (mppi.py:994-1009): per step stage, obstacle, unknown; then the
// terminal cost; then the control cost of all T steps.
// -------------------------------------------------------------------------
// steps whose noise is fetched together: kNoiseBatch independent 8-byte loads per
// lane are in flight while the previous batch is integrated, and the batch body is
// straight-line code (no per-step branch), so that the compiler can hoist the LDS
// reads of u[t] and interleave independent work of neighbouring steps.
constexpr int kNoiseBatch = 8;

struct RolloutState {
  float x, y, th, cost;
  double d2;
  bool done, reached;
};

// 16-bit cell: lin (7 bits) | ang (7 bits) << 7 | obstacle << 14 | unknown << 15.
// Usable when both masks are 0/1 and every traction byte is in [0, 127] (the
// reference's own maps always are: tractions 0..100, indicator masks).
template <int KIND, bool EXACT, bool BOUNDED, bool LDSMAP>
__device__ __forceinline__ void map_step(const DevParams& P, const uint32_t* __restrict__ cells,
                                         const int8_t* __restrict__ risk, const uint16_t* lds_map,
                                         float2 ut, float2 e, RolloutState& st) {
  int xi = clamp_index(floordiv_to_int(st.x - P.xlo, P.res, P.inv_res), P.cols);
  int yi = clamp_index(floordiv_to_int(st.y - P.ylo, P.res, P.inv_res), P.rows);
  int lin, ang, obs, unk, rk = 0;
  if (LDSMAP) {
    // the window covers every cell reachable within the horizon (host-checked);
    // the clamp only guards memory safety
    int wr = clamp_index(yi - P.win_r0, P.win_rows), wc = clamp_index(xi - P.win_c0, P.win_cols);
    uint32_t c16 = lds_map[wr * P.win_cols + wc];
    lin = (int)(c16 & 127u);
    ang = (int)((c16 >> 7) & 127u);
    obs = (int)((c16 >> 14) & 1u);
    unk = (int)(c16 >> 15);
  } else {
    int ci = yi * P.cols + xi;
    uint32_t cell = cells[ci];
    if (KIND == MAP_SPEED) rk = (int)risk[ci];
    lin = (int)(int8_t)(cell & 0xff);
    ang = (int)(int8_t)((cell >> 8) & 0xff);
    obs = (int)(int8_t)((cell >> 16) & 0xff);
    unk = (int)(int8_t)(cell >> 24);
  }
  float v = clip_f32(ut.x + e.x, P.v_lo, P.v_hi);
  float w = clip_f32(ut.y + e.y, P.w_lo, P.w_hi);
  StepOut o = unicycle_step<EXACT, BOUNDED>(P, st.x, st.y, st.th, v, w, lin, ang);
  double step_time = (double)P.dt;
  if (KIND == MAP_SPEED) {
    // dt / (effective_speed + 1e-6), effective speed from the risk map (mppi.py:1095-1096)
    double eff = fma(P.lin_ratio, (double)rk, P.lin_lo);
    step_time = (double)P.dt / (eff + 1e-6);
  }
  float c1 = add_stage_cost<EXACT>(P, st.cost, o.d2, step_time);
  c1 = c1 + (float)obs * P.obs_cost;
  c1 = c1 + (float)unk * P.unk_cost;
  bool hit = o.d2 <= (double)P.gt2;
  bool act = !st.done;
  st.x = act ? o.x : st.x;
  st.y = act ? o.y : st.y;
  st.th = act ? o.th : st.th;
  st.d2 = act ? o.d2 : st.d2;
  st.cost = act ? c1 : st.cost;
  st.reached = st.reached || (act && hit);
  st.done = st.done || hit;
}

// Copies the map window into LDS with the threads [first_thread, first_thread + n_threads)
// of the workgroup.  Window columns [win_c0, win_c0 + win_cols) are multiples of 8 cells =
// 16-byte vectors.  Batches of eight loads per lane, no branch inside a batch: indices past
// the end are clamped to the last vector, which is then simply written again.
// (CELL_BYTES = 4: the 32-bit cells of the speed-map mode, same geometry in cells)
template <int CELL_BYTES = 2, int BATCH = 8>
__device__ __forceinline__ void copy_window_to_lds(const DevParams& P, const uint16_t* __restrict__ cells16,
                                                   uint16_t* lds_map, int first_thread, int n_threads) {
  const int tid = (int)threadIdx.x - first_thread;
  if (tid < 0 || tid >= n_threads) return;
  constexpr int kPerVec = 16 / CELL_BYTES;  // cells per 16-byte vector
  const int vec_per_row = P.win_cols / kPerVec;
  const int total = P.win_rows * vec_per_row;
  const int src_pitch = P.pitch16 / kPerVec;
  const uint4* src = reinterpret_cast<const uint4*>(cells16) + ((size_t)P.win_r0 * P.pitch16 + P.win_c0) / kPerVec;
  uint4* dst = reinterpret_cast<uint4*>(lds_map);
  const bool flat = (vec_per_row == src_pitch);  // full-width window: one contiguous run
  const int step = BATCH * n_threads;
  for (int i0 = tid; i0 < total; i0 += step) {
    uint4 v[BATCH];
    int idx[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      int i = min(i0 + k * n_threads, total - 1);
      idx[k] = i;
      int r = flat ? 0 : i / vec_per_row;
      v[k] = src[flat ? (size_t)i : (size_t)r * src_pitch + (i - r * vec_per_row)];
    }
#pragma unroll
    for (int k = 0; k < BATCH; ++k) dst[idx[k]] = v[k];
  }
}

// The same copy without the trip through registers (round 5): global_load_lds_dwordx4 writes 64 lanes x 16 bytes
// straight into LDS at M0 + lane * 16, so a wave keeps ALL its vectors in flight (the register version above has 8 per
// lane and pays the memory latency once per batch: 2 556 vectors at C2 = three round trips of ~2.3k cycles for two
// waves, 7k cycles between kernel entry and the first barrier; profiles/r05_pipe_notes.md).  Vector v of the window
// (row-major, win_cols / 8 per row) lands at lds_map + 16 v; the row of v by a float reciprocal (exact: v < 2^16,
// rows < 2^10).  The caller waits (s_waitcnt vmcnt(0)) before the barrier that publishes the window.
__device__ __forceinline__ void copy_window_to_lds_direct(const DevParams& P, const uint16_t* __restrict__ cells16,
                                                          uint16_t* lds_map, int first_thread, int n_threads) {
  const int tid = (int)threadIdx.x - first_thread;
  if (tid < 0 || tid >= n_threads) return;  // (whole waves: both bounds are multiples of 64)
  const int vec_per_row = P.win_cols >> 3;
  const int total = P.win_rows * vec_per_row;
  const int src_pitch = P.pitch16 >> 3;
  const uint4* src = reinterpret_cast<const uint4*>(cells16) + ((size_t)P.win_r0 * P.pitch16 + P.win_c0) / 8;
  const float inv_vpr = 1.0f / (float)vec_per_row;
  const int lane = tid & 63;
  const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
  if (vec_per_row == src_pitch) {  // full-width window (the whole map at long horizons): one contiguous run
    for (int base = wave_base; base < total; base += n_threads) {
      const int v = base + lane;
      if (v < total)
        __builtin_amdgcn_global_load_lds(src + v, (__attribute__((address_space(3))) void*)(reinterpret_cast<uint4*>(lds_map) + base),
                                         16, 0, 0);
    }
    return;
  }
  for (int base = wave_base; base < total; base += n_threads) {
    const int v = base + lane;
    const int r = (int)(((float)v + 0.5f) * inv_vpr);
    const uint4* g = src + (size_t)r * src_pitch + (v - r * vec_per_row);
    if (v < total)
      __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(reinterpret_cast<uint4*>(lds_map) + base),
                                       16, 0, 0);
  }
}

// LDS: [T] double2 control ratios | [T] float2 u | (LDSMAP) window of 16-bit cells
template <int KIND, bool EXACT, bool BOUNDED, bool LDSMAP>
__global__ void k_rollout_map(DevParams P, const uint32_t* __restrict__ cells,
                              const uint16_t* __restrict__ cells16, const int8_t* __restrict__ risk,
                              const float2* __restrict__ noise, const float2* __restrict__ u,
                              float* __restrict__ costs) {
  extern __shared__ double2 uos[];
  ktime_begin(P);
  // batched handle: the waves of a workgroup belong to one problem (host: blockDim/64 divides inst_tiles)
  u = select_instance(P, u, P.inst ? (int)(blockIdx.x * (blockDim.x >> 6)) / P.inst_tiles : 0);
  float2* us = reinterpret_cast<float2*>(uos + P.n_steps);  // u[t] staged next to the ratios
  // 16-byte aligned start of the map window
  uint16_t* lds_map = reinterpret_cast<uint16_t*>(uos + P.n_steps + (P.n_steps + 1) / 2);
  if (LDSMAP) {
    copy_window_to_lds(P, cells16, lds_map, 0, (int)blockDim.x);
  }
  for (int t = threadIdx.x; t < P.n_steps; t += blockDim.x) us[t] = u[t];
  stage_control_ratios(P, u, uos);
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = n < P.n_local;
  const int nn = live ? n : P.n_local - 1;
  const int T = P.n_steps;
  const float2* col = noise + tile_index(0, nn, T);  // this lane's column; rows are 64 apart

  RolloutState st = {P.x0, P.y0, P.th0, 0.0f, 1e9, false, false};
  float2 e_cur[kNoiseBatch], e_nxt[kNoiseBatch];
  // rows past the horizon are clamped to the last row: every load is unconditional
#pragma unroll
  for (int j = 0; j < kNoiseBatch; ++j) e_cur[j] = col[(size_t)min(j, T - 1) * 64];
  int t0 = 0;
  for (; t0 + kNoiseBatch <= T; t0 += kNoiseBatch) {
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) e_nxt[j] = col[(size_t)min(t0 + kNoiseBatch + j, T - 1) * 64];
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j)
      map_step<KIND, EXACT, BOUNDED, LDSMAP>(P, cells, risk, lds_map, us[t0 + j], e_cur[j], st);
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) e_cur[j] = e_nxt[j];
    if (__all(st.done)) break;
  }
  if (!__all(st.done))
    for (int t = t0; t < T; ++t) {  // (batch registers shifted down, not indexed: see k_rollout_fused)
      map_step<KIND, EXACT, BOUNDED, LDSMAP>(P, cells, risk, lds_map, us[t], e_cur[0], st);
#pragma unroll
      for (int j = 0; j + 1 < kNoiseBatch; ++j) e_cur[j] = e_cur[j + 1];
    }

  float cost = st.cost;
  // terminal cost (mppi.py:26-28, 1005)
  double term = (st.reached ? 0.0 : 1.0) * sqrt(st.d2) / P.v_post_den;
  cost = EXACT ? (float)((double)cost + term) : cost + (float)term;
  // control cost over ALL steps, also after an early goal break (mppi.py:1007-1009);
  // the float32-rounded accumulation is sequential, loads and products are batched
  for (t0 = 0; t0 + kNoiseBatch <= T; t0 += kNoiseBatch) {
    double cc[kNoiseBatch];
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) cc[j] = control_cost(P, uos[t0 + j], col[(size_t)(t0 + j) * 64]);
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) cost = EXACT ? (float)((double)cost + cc[j]) : cost + (float)cc[j];
  }
  for (int t = t0; t < T; ++t) {
    double c1 = control_cost(P, uos[t], col[(size_t)t * 64]);
    cost = EXACT ? (float)((double)cost + c1) : cost + (float)c1;
  }
  if (live) costs[n] = cost;
  ktime_end(P);
}

// -------------------------------------------------------------------------
// Throughput variant of the deterministic rollout (MPPI_MODE_DET, exact math, LDS
// window, bounded heading increment): one wave per 64 rollouts, 4..16 waves per CU.
// When there are more tiles than a single round of the pipelined kernel below can
// take, latency no longer matters and the SIMDs are bound by instruction issue; this
// kernel is k_rollout_map<DET, true, true, true> with the per-step instruction count
// cut from ~190 to ~120 by the same exact rewrites the pipelined kernel uses:
//   * (cos, sin) rotated by the exact increment of the float32-rounded heading instead
//     of a full sincos (|increment| <= 0.36 rad, T <= 2000: host-checked),
//   * branch-free Newton sqrt, 24-bit window index, floor instead of the exact floor
//     division when the resolution is a power of two.
// Rounding points are those of the reference (mppi.py:977-1009): float64 products,
// float32 stores of x, y, theta and of the running cost.  After the goal is reached the
// state keeps integrating (harmless: only the cost is frozen), as in the pipelined kernel.
// LDS: [T] double2 control ratios | [T] float2 u | window of 16-bit cells.
//
// ONEPASS (MPPI_MATH_FAST): state AND stage costs keep the reference's rounding points -- on a map
// whose traction changes from cell to cell a trajectory one ulp off reads another cell once in ~10^5
// steps, and a rollout stopped in a zero-traction cell adds the SAME stage cost for the rest of the
// horizon, so a float32 addend that rounds the other way does so on every one of those steps (measured:
// 0.4 % of the costs up to 50 ulp off, tests/test_gpu_fast_throughput.py) -- but the control cost is ONE
// float32 accumulator filled while the noise goes by and added after the terminal cost: the second
// pass over the noise (the reference's order, mppi.py:1005-1009) disappears.  The price is the
// reference's own rounding noise: its T float32-rounded additions of control-cost terms are not
// reproduced (sqrt(T / 12) ulp rms: 99.9 % of the costs within 7e-7 at T = 100, 1.2e-6 at T = 200;
// the floor of ANY single-pass design, tests/test_cost_order_noise.py).
// LDS (ONEPASS): [T] float2 control ratios in the double2 area.
// -------------------------------------------------------------------------
// LONE (round 6; exact det mode, workgroups of at most four waves: one or two waves per SIMD, N = 65536 on one GPU):
// nobody hides this wave's trips to memory, and the control-cost pass made one per eight steps (2k cycles each) and one
// per step of the horizon's last, shorter batch.  Here everything that pass still reads from memory is requested at
// once, in front of the steps it reads from the noise kept in LDS (DevParams::stash_steps); 256 registers per lane
// (launch bound of four waves) hold it.
template <bool POW2RES, bool SPEED = false, bool ONEPASS = false, bool LONE = false>
__global__ __launch_bounds__(LONE ? 256 : 1024) void k_rollout_fused(DevParams P, const uint16_t* __restrict__ cells16,
                                const float2* __restrict__ noise, const float2* __restrict__ u,
                                float* __restrict__ costs, float* __restrict__ w_rel,
                                float* __restrict__ tile_beta) {
  extern __shared__ double2 uos[];
  signal_progress(P);
  ktime_begin(P);
  MPPI_STAMP(blockIdx.x == 5 && threadIdx.x == 0, 710);  // (stamps build: tools/fused_stamps.py)
  // batched handle: the waves of a workgroup belong to one problem (host: blockDim/64 divides inst_tiles)
  u = select_instance(P, u, P.inst ? (int)(blockIdx.x * (blockDim.x >> 6)) / P.inst_tiles : 0);
  const int T = P.n_steps, N = P.n_local;
  float2* us = reinterpret_cast<float2*>(uos + T);
  uint16_t* lds_map = reinterpret_cast<uint16_t*>(uos + T + (T + 1) / 2);
  copy_window_to_lds<SPEED ? 4 : 2>(P, cells16, lds_map, 0, (int)blockDim.x);
  for (int t = threadIdx.x; t < T; t += blockDim.x) us[t] = u[t];
  float2* ratio32 = reinterpret_cast<float2*>(uos);
  if constexpr (ONEPASS) {
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
      const float2 ut = u[t];
      ratio32[t] = make_float2((float)((double)ut.x / P.s0sq), (float)((double)ut.y / P.s1sq));
    }
    __syncthreads();
  } else {
    stage_control_ratios(P, u, uos);  // ends with a barrier
  }
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = n < N;
  const int nn = live ? n : N - 1;
  const float2* col = noise + tile_index(0, nn, T);  // this lane's column; rows are 64 apart
  // (exact mode) this wave's rows of the noise stash: see DevParams::stash_steps
  float2* stash = reinterpret_cast<float2*>(reinterpret_cast<char*>(uos) + P.stash_offset) +
                  ((size_t)(threadIdx.x >> 6) * (size_t)P.stash_steps * 64 + (threadIdx.x & 63));
  [[maybe_unused]] int stashed = 0;  // steps [0, stashed) of this wave's noise are in LDS
  if (!wait_for_noise(P)) {  // (workgroup-uniform; reported by the host: DevParams::flag_fault)
    ktime_end(P);
    return;
  }

  MPPI_STAMP(blockIdx.x == 5 && threadIdx.x == 0, 711);  // (stamps build: tools/fused_stamps.py)
  float x = P.x0, y = P.y0, th = P.th0, cost = 0.0f;
  [[maybe_unused]] float cc32 = 0.0f;
  double x64 = (double)x, y64 = (double)y, th64 = (double)th, d2 = 1e9;
  double s, c;
  sincos_f64<false>(th64, s, c);
  bool done = false, reached = false;
  const double dt64 = (double)P.dt, gt2 = (double)P.gt2;
  const float win_c0f = (float)P.win_c0, win_r0f = (float)P.win_r0;
  const float win_last_col = (float)(P.win_cols - 1), win_last_row = (float)(P.win_rows - 1);
  const int win_pitch_bytes = (SPEED ? 4 : 2) * P.win_cols;
  const char* lds_bytes = reinterpret_cast<const char*>(lds_map);

  // the cell of (px, py) in the LDS window: the request only -- the value is looked at one step later
  auto lookup = [&](float px, float py) {
    int xi, yi;
    if (POW2RES) {  // res is a power of two: see cell_coord_pow2
      xi = cell_coord_pow2(px, P.xlo, P.inv_res, win_c0f, win_last_col);
      yi = cell_coord_pow2(py, P.ylo, P.inv_res, win_r0f, win_last_row);
    } else {
      xi = clamp_index(floordiv_to_int(px - P.xlo, P.res, P.inv_res) - P.win_c0, P.win_cols);
      yi = clamp_index(floordiv_to_int(py - P.ylo, P.res, P.inv_res) - P.win_r0, P.win_rows);
    }
    // 32-bit cell (speed-map mode): the 16 bits below + the risk traction byte
    if constexpr (SPEED) {
      return *reinterpret_cast<const uint32_t*>(lds_bytes + (__mul24(yi, win_pitch_bytes) + (xi << 2)));
    } else {
      // (handed on as a 16-bit FLOAT: a 16-bit integer is zero-extended next to the load -- v_and_b32 0xffff behind an
      //  s_waitcnt, i.e. the wait for the gather right where it was issued, in front of the barrier below: the ISA of
      //  round 6's first version -- while a half is just its bits in the low half of the register; pipe_cell_arrived.
      //  Measured: nothing at ns / C4 / C5, where the generator's or the other rollout waves fill the slots either way)
      return __builtin_bit_cast(_Float16, *reinterpret_cast<const uint16_t*>(lds_bytes + (__mul24(yi, win_pitch_bytes) + (xi << 1))));
    }
  };
  // Round 6: the chain of a step is  cell -> traction -> x, y -> next cell, and its LDS gather (64 lanes, bank conflicts:
  // ~100+ cycles) was waited for a handful of instructions after it was issued -- with ONE wave per SIMD at N = 65536
  // nothing else covers it.  As in the pipelined kernels' state role the lookup of step t + 1 is requested the moment
  // x, y of step t exist, and the rotation and the whole cost side of step t run in its shadow.
  auto cell_now = lookup(x, y);
  auto step = [&](float2 ut, float2 e, [[maybe_unused]] int t) {
    // the cell this step STARTS in: traction codes and penalty bits (the compiler's wait for the lookup lands here)
    uint32_t lin_code, ang_code, pen_bits;
    double step_time = dt64;
    if constexpr (SPEED) {
      const uint32_t c32 = cell_now;
      lin_code = c32 & 127u;
      ang_code = (c32 >> 7) & 127u;
      pen_bits = c32 & 0xc000u;
      // time per step = dt over the risk-aware effective speed (mppi.py:1095-1096)
      const double eff = fma(P.lin_ratio, (double)(int)(int8_t)(c32 >> 16), P.lin_lo);
      step_time = dt64 / (eff + 1e-6);
    } else {
      asm volatile("v_and_b32 %0, 0x7f, %3\n\tv_bfe_u32 %1, %3, 7, 7\n\tv_and_b32 %2, 0xc000, %3"
                   : "=&v"(lin_code), "=&v"(ang_code), "=v"(pen_bits)
                   : "v"(cell_now));
    }
    const double vtr = fma(P.lin_ratio, (double)(int)lin_code, P.lin_lo);
    const double wtr = fma(P.ang_ratio, (double)(int)ang_code, P.ang_lo);
    const double qv = dt64 * (double)clip_f32(ut.x + e.x, P.v_lo, P.v_hi);  // exact: float32 factors
    const double qw = dt64 * (double)clip_f32(ut.y + e.y, P.w_lo, P.w_hi);
    x = (float)fma(vtr, qv * c, x64);
    y = (float)fma(vtr, qv * s, y64);
    th = (float)fma(wtr, qw, th64);
    cell_now = lookup(x, y);
    __builtin_amdgcn_sched_barrier(0);  // ---- everything below runs while the lookup is in flight
    x64 = (double)x;
    y64 = (double)y;
    const double th_new = (double)th;
    rotate_sincos_f64(th_new - th64, s, c);  // exact increment of the ROUNDED heading
    th64 = th_new;
    if constexpr (ONEPASS) {
      const float2 rt = ratio32[t];  // every step's term, also past the goal (mppi.py:1007-1009)
      cc32 = fmaf(rt.x, e.x, fmaf(rt.y, e.y, cc32));
    }
    {
      const double dx = (double)(P.xg - x), dy = (double)(P.yg - y);
      const double nd2 = fma(dx, dx, dy * dy);
      float c1 = (float)((double)cost + fma(P.dist_weight, sqrt_newton_f64(nd2), step_time));
      c1 = c1 + ((pen_bits & 0x4000u) ? P.obs_cost : 0.0f);  // cell the step STARTED in (mppi.py:971-998)
      c1 = c1 + ((pen_bits & 0x8000u) ? P.unk_cost : 0.0f);
      const bool hit = nd2 <= gt2, act = !done;
      cost = act ? c1 : cost;
      d2 = act ? nd2 : d2;
      reached = reached || (act && hit);
      done = done || hit;
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  float2 e_cur[kNoiseBatch], e_nxt[kNoiseBatch];
#pragma unroll
  for (int j = 0; j < kNoiseBatch; ++j) e_cur[j] = col[(size_t)min(j, T - 1) * 64];
  int t0 = 0;
  for (; t0 + kNoiseBatch <= T; t0 += kNoiseBatch) {
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) e_nxt[j] = col[(size_t)min(t0 + kNoiseBatch + j, T - 1) * 64];
    if (!ONEPASS && t0 < P.stash_steps) {  // (uniform; stash_steps is a multiple of the batch)
#pragma unroll
      for (int j = 0; j < kNoiseBatch; ++j) stash[(size_t)(t0 + j) * 64] = e_cur[j];
      stashed = t0 + kNoiseBatch;
    }
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) step(us[t0 + j], e_cur[j], t0 + j);
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) e_cur[j] = e_nxt[j];
    if (!ONEPASS && __all(done)) break;  // (ONEPASS: the control cost needs every step's noise)
  }
  // (the horizon's last, shorter batch: the batch registers shifted down one per step -- indexing them with t - t0
  //  put both batches into scratch memory, seven scratch accesses per batch in the loop above as well: round 6)
  if (ONEPASS || !__all(done))
    for (int t = t0; t < T; ++t) {
      step(us[t], e_cur[0], t);
#pragma unroll
      for (int j = 0; j + 1 < kNoiseBatch; ++j) e_cur[j] = e_cur[j + 1];
    }

  MPPI_STAMP(blockIdx.x == 5 && threadIdx.x == 0, 712);  // (stamps build: tools/fused_stamps.py)
  // terminal cost, then the control cost of all T steps (mppi.py:1005-1009): the float32-rounded
  // accumulation is sequential, loads and products are batched
  cost = (float)((double)cost + (reached ? 0.0 : 1.0) * sqrt(d2) / P.v_post_den);
  if constexpr (ONEPASS) {
    cost = (float)fma((double)P.lambda, (double)cc32, (double)cost);
    if (live) costs[n] = cost;
    if ((n & ~63) < N) emit_tile_weights(cost, live, P.lambda, n, n >> 6, w_rel, tile_beta);
    ktime_end(P);
    return;
  }
  // (round 6: the steps whose noise this wave kept in LDS come from there -- with one or two waves per SIMD every batch
  //  of this pass was a trip to memory nobody hid: profiles/r06_ns_notes.md, section 5)
  if constexpr (LONE) {
    static_assert(!LONE || (!SPEED && !ONEPASS), "LONE serves the exact deterministic mode");
    // the batches behind the kept steps: up to kPre of them and the horizon's last, shorter batch, all in flight at once
    constexpr int kPre = 6;
    float2 g[kPre][kNoiseBatch], gt[kNoiseBatch - 1];
    const int full = (T / kNoiseBatch) * kNoiseBatch;  // steps in whole batches
#pragma unroll
    for (int b = 0; b < kPre; ++b)
#pragma unroll
      for (int j = 0; j < kNoiseBatch; ++j) g[b][j] = col[(size_t)min(stashed + b * kNoiseBatch + j, T - 1) * 64];
#pragma unroll
    for (int j = 0; j < kNoiseBatch - 1; ++j) gt[j] = col[(size_t)min(full + j, T - 1) * 64];
    for (t0 = 0; t0 < stashed; t0 += kNoiseBatch) {  // from LDS
      double cc[kNoiseBatch];
#pragma unroll
      for (int j = 0; j < kNoiseBatch; ++j) cc[j] = control_cost(P, uos[t0 + j], stash[(size_t)(t0 + j) * 64]);
#pragma unroll
      for (int j = 0; j < kNoiseBatch; ++j) cost = (float)((double)cost + cc[j]);
    }
#pragma unroll
    for (int b = 0; b < kPre; ++b) {  // requested above
      if (t0 + kNoiseBatch <= T) {  // (uniform)
        double cc[kNoiseBatch];
#pragma unroll
        for (int j = 0; j < kNoiseBatch; ++j) cc[j] = control_cost(P, uos[t0 + j], g[b][j]);
#pragma unroll
        for (int j = 0; j < kNoiseBatch; ++j) cost = (float)((double)cost + cc[j]);
        t0 += kNoiseBatch;
      }
    }
    for (; t0 + kNoiseBatch <= T; t0 += kNoiseBatch) {  // a longer horizon: one trip per batch, as before
      double cc[kNoiseBatch];
#pragma unroll
      for (int j = 0; j < kNoiseBatch; ++j) cc[j] = control_cost(P, uos[t0 + j], col[(size_t)(t0 + j) * 64]);
#pragma unroll
      for (int j = 0; j < kNoiseBatch; ++j) cost = (float)((double)cost + cc[j]);
    }
#pragma unroll
    for (int j = 0; j < kNoiseBatch - 1; ++j)
      if (full + j < T) cost = (float)((double)cost + control_cost(P, uos[full + j], gt[j]));
    t0 = T;
  } else
  for (t0 = 0; t0 + kNoiseBatch <= T; t0 += kNoiseBatch) {
    double cc[kNoiseBatch];
    if (t0 + kNoiseBatch <= stashed) {
#pragma unroll
      for (int j = 0; j < kNoiseBatch; ++j) cc[j] = control_cost(P, uos[t0 + j], stash[(size_t)(t0 + j) * 64]);
    } else {
#pragma unroll
      for (int j = 0; j < kNoiseBatch; ++j) cc[j] = control_cost(P, uos[t0 + j], col[(size_t)(t0 + j) * 64]);
    }
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) cost = (float)((double)cost + cc[j]);
  }
  for (int t = t0; t < T; ++t) cost = (float)((double)cost + control_cost(P, uos[t], col[(size_t)t * 64]));
  MPPI_STAMP(blockIdx.x == 5 && threadIdx.x == 0, 713);  // (stamps build: tools/fused_stamps.py)
  if (live) costs[n] = cost;
  // first half of the control update (update_kernels.h): weights relative to the tile's minimum
  if ((n & ~63) < N) emit_tile_weights(cost, live, P.lambda, n, n >> 6, w_rel, tile_beta);
  ktime_end(P);
}

// -------------------------------------------------------------------------
// Pipelined deterministic rollout (MPPI_MODE_DET, exact math, LDS map).
//
// At N = 8192 a GPU has 8x more SIMDs than there are waves, and a lone wave
// issues one float64 instruction per ~5.4 cycles (one float32 per ~2.9): the fused
// kernel above is bound by its ~190 instructions per step, not by memory.  Here
// two waves cooperate on every 64 rollouts and only the unavoidable chain stays
// on the critical wave:
//   state wave  cell lookup in the LDS map, traction, float64 state update,
//               rotation of (cos, sin) by the exact heading increment; reads the
//               clipped controls from a ring and publishes x, y (float32);
//   cost wave   streams the noise: clipped controls for the NEXT chunk, control-
//               cost products (to a scratch array for the final accumulation);
//               one chunk BEHIND the state wave: obstacle / unknown bits of the
//               visited cell, distance, sqrt, stage cost, goal test, the
//               float32-rounded accumulation; then terminal and control costs.
// They meet at one workgroup barrier per chunk of C steps (double-buffered rings).
// Every rollout sees the same operations with the same rounding points as in
// k_rollout_map; only the trig is evaluated incrementally.
// LDS: [T] double2 ratios | [T] float2 u | map window | per pair:
//      xy[2][C][64] float2, vw[2][C][64] float2.
// -------------------------------------------------------------------------
template <int C>
struct PipeRing {
  static constexpr int kHalf = C * 64;  // entries per buffer
  // xy[2][C][64] float2 + qd[2][C][64] double2 {dt*v, dt*w} + cell[2][C][64] uint16 (the 16-bit cell a step started in)
  static constexpr int kBytesPerPair = 2 * kHalf * (int)sizeof(float2) + 2 * kHalf * (int)sizeof(double2) + 2 * kHalf * 2;
};

// ---- the state role of the exact schedule: one chunk of C steps (round 5) --------------------------------------
// Per step and rollout the chain is  cell of (x, y) -> LDS lookup -> traction -> x, y, theta (float64 products, one
// float32 rounding each: mppi.py:977-990) -> next cell.  A lone wave issues one instruction per ~5 cycles whatever its
// type, so a step costs what the wave issues plus whatever latency nothing covers.  Round 1-4 issued the lookup of
// step t and waited for it (~100 cycles of an LDS gather with bank conflicts) with four instructions behind it; here
// the lookup of step t + 1 is issued the moment x_{t+1}, y_{t+1} exist, and the 24 instructions that do not need it
// (float64 copies of the new state, the rotation of (cos, sin) by the exact heading increment, the ring stores) run in
// its shadow (sched_barrier keeps the compiler from moving them back in front).  The instruction count per step is
// cut as well: both cell coordinates in one packed float32 subtract and one packed fma; no clamp where the host has
// proved it idle (POW2RES variant: a map no rollout can leave -- launch_plan.h, unclamped_lookup_ok; elsewhere the
// variant with the exact floor division clamps to the border cell); the raw 16-bit cell goes to the cost wave
// instead of a shifted byte.  Same operations on the same operands as before: the oracle's bits.
typedef float pipe_f2 __attribute__((ext_vector_type(2)));

template <bool POW2RES>
struct PipeWindow {
  pipe_f2 lo, origin;     // {xlo, ylo}, {win_c0, win_r0}
  float inv_res, res;
  int pitch_bytes, c0, r0, cols, rows;
  uint32_t base;          // LDS byte address of the window
  // the additive traction constants in VECTOR registers: v_fma_f64 takes one scalar operand, and with ratio and lo both
  // scalar the compiler copies lo into a register pair again in every step (two v_mov_b64 per step; profiles/r05_c3_isa.md)
  double lin_lo, ang_lo;
  __device__ __forceinline__ PipeWindow(const DevParams& P, const uint16_t* lds_map) {
    lin_lo = P.lin_lo; ang_lo = P.ang_lo;
    asm volatile("" : "+v"(lin_lo), "+v"(ang_lo));
    lo = pipe_f2{P.xlo, P.ylo};
    origin = pipe_f2{(float)P.win_c0, (float)P.win_r0};
    inv_res = P.inv_res; res = P.res;
    pitch_bytes = 2 * P.win_cols;
    c0 = P.win_c0; r0 = P.win_r0; cols = P.win_cols; rows = P.win_rows;
    base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint16_t*)lds_map;
  }
  // LDS byte address of the cell of (x, y)
  __device__ __forceinline__ uint32_t address(float x, float y) const {
    uint32_t xi, yi;
    if (POW2RES) {
      // d = fl(pos - lo) is the reference's float32 difference; d * inv_res and the subtraction of the integer window
      // origin are exact (device_math.h, cell_coord_pow2); the truncating conversion is the floor for q >= 0
      const pipe_f2 q = __builtin_elementwise_fma(pipe_f2{x, y} - lo, pipe_f2{inv_res, inv_res}, -origin);
      // (the instruction itself, not a C++ conversion -- which is undefined for a negative value: v_cvt_u32_f32 saturates
      //  at 0.  No upper clamp: the host selects this variant only where no rollout can leave the map, launch_plan.h:
      //  unclamped_lookup_ok; the steps past the horizon of the last chunk may index beyond the window -- an LDS read of
      //  bytes nobody uses, or zero past the allocation)
      asm("v_cvt_u32_f32 %0, %1" : "=v"(xi) : "v"(q.x));
      asm("v_cvt_u32_f32 %0, %1" : "=v"(yi) : "v"(q.y));
    } else {
      xi = (uint32_t)clamp_index(floordiv_to_int(x - lo.x, res, inv_res) - c0, cols);
      yi = (uint32_t)clamp_index(floordiv_to_int(y - lo.y, res, inv_res) - r0, rows);
    }
    uint32_t row, a;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(yi), "s"(pitch_bytes), "v"(base));  // (one SGPR per VOP3 on gfx9)
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(a) : "v"(xi), "v"(row));
    return a;
  }
  // The request only: the value stays a 16-bit load result until pipe_cell_arrived() -- a zero extension next to the
  // load would be the load's first user, and the wait for the lookup with it.
  __device__ __forceinline__ uint16_t lookup(float x, float y) const {
    return *reinterpret_cast<const __attribute__((address_space(3))) uint16_t*>(address(x, y));
  }
};
// The two 7-bit traction codes straight from the 16-bit load result (the compiler's s_waitcnt for the lookup lands
// here).  Through a uint32 the compiler puts a zero extension -- v_and_b32 0xffff -- in front of the masks: one more
// instruction on the position loop's dependent chain (cell -> code -> traction -> x -> coordinates -> address -> cell,
// ~10 cycles per dependent instruction: profiles/r04_chain_latency.txt).
__device__ __forceinline__ void pipe_cell_arrived(uint16_t raw, uint32_t& lin, uint32_t& ang) {
  // (handed over as a 16-bit FLOAT: an integer of 16 bits is zero-extended for the register operand -- the very
  //  instruction this is about --, a half is just its bits in the low half of the register)
  const _Float16 bits = __builtin_bit_cast(_Float16, raw);
  asm volatile("v_and_b32 %0, 0x7f, %2\n\tv_bfe_u32 %1, %2, 7, 7" : "=&v"(lin), "=v"(ang) : "v"(bits));
}

struct PipeState {
  float x, y;
  double x64, y64, th64, s, c;
  uint16_t cell;  // the 16-bit cell of (x, y): requested when x, y were formed
};

// (st.cell: the caller requests the start cell once the window is in LDS -- win.lookup(st.x, st.y))
__device__ __forceinline__ PipeState pipe_state_init(const DevParams& P) {
  PipeState st;
  st.x = P.x0; st.y = P.y0;
  st.x64 = (double)P.x0; st.y64 = (double)P.y0; st.th64 = (double)P.th0;
  sincos_f64<false>(st.th64, st.s, st.c);
  st.cell = 0;
  return st;
}

// One step: {dt*v, dt*w} in qd (exact products of float32 factors); leaves (x, y) after the step in *out_xy and the cell
// the step STARTED in (its obstacle / unknown bits are what the step pays: mppi.py:971-998) in *out_cell.
template <bool POW2RES, bool CHECK_ROTATION>
__device__ __forceinline__ void pipe_state_step(const DevParams& P, const PipeWindow<POW2RES>& win, PipeState& st,
                                                const double2 qd, float2* out_xy, uint16_t* out_cell) {
  const uint16_t c16 = st.cell;
  uint32_t lin, ang;
  pipe_cell_arrived(c16, lin, ang);
  const double vtr = fma(P.lin_ratio, (double)(int)lin, win.lin_lo);
  const double wtr = fma(P.ang_ratio, (double)(int)ang, win.ang_lo);
  const float x = (float)fma(vtr, qd.x * st.c, st.x64);
  const float y = (float)fma(vtr, qd.x * st.s, st.y64);
  const float th = (float)fma(wtr, qd.y, st.th64);
  st.cell = win.lookup(x, y);
  __builtin_amdgcn_sched_barrier(0);  // ---- everything below runs while the lookup is in flight
  st.x = x; st.y = y;
  st.x64 = (double)x;
  st.y64 = (double)y;
  const double th_new = (double)th;
  // exact increment of the ROUNDED heading (beyond the rotation's range -- never with the reference's parameters,
  // host-proved where CHECK_ROTATION is off -- the full evaluation)
  if (!CHECK_ROTATION || __all(fabs(th_new - st.th64) <= 0.36)) rotate_sincos_f64(th_new - st.th64, st.s, st.c);
  else sincos_f64<false>(th_new, st.s, st.c);
  st.th64 = th_new;
  *out_xy = make_float2(x, y);
  *out_cell = c16;
  __builtin_amdgcn_sched_barrier(0);
}

// C steps: reads {dt*v, dt*w} of step j from in_qd[j*64 + lane] (all C requests first), stores to out_xy / out_cell[j*64 + lane]
template <int C, bool POW2RES, bool CHECK_ROTATION>
__device__ __forceinline__ void pipe_state_chunk(const DevParams& P, const PipeWindow<POW2RES>& win, PipeState& st,
                                                 const double2* in_qd, float2* out_xy, uint16_t* out_cell, int lane) {
  double2 qd[C];
#pragma unroll
  for (int j = 0; j < C; ++j) qd[j] = in_qd[j * 64 + lane];
#pragma unroll
  for (int j = 0; j < C; ++j)
    pipe_state_step<POW2RES, CHECK_ROTATION>(P, win, st, qd[j], out_xy + j * 64 + lane, out_cell + j * 64 + lane);
}
// ... the horizon's last, shorter chunk: `count` < C steps (the steps past the horizon used to be integrated and ignored:
// 4 x 265 cycles at T = 100)
template <bool POW2RES, bool CHECK_ROTATION>
__device__ __forceinline__ void pipe_state_tail(const DevParams& P, const PipeWindow<POW2RES>& win, PipeState& st,
                                                const double2* in_qd, float2* out_xy, uint16_t* out_cell, int lane, int count) {
  for (int j = 0; j < count; ++j)
    pipe_state_step<POW2RES, CHECK_ROTATION>(P, win, st, in_qd[j * 64 + lane], out_xy + j * 64 + lane, out_cell + j * 64 + lane);
}

// Roles of the waves of one workgroup (W = blockDim / 192 triples, triple i = waves
// i, W+i, 2W+i):  producer -> state -> cost, each one chunk behind the previous.
//   producer  streams the noise: clipped controls of chunk k+1 into ring_vw,
//             control-cost products into cc_scratch (global, tile-major);
//   state     integrates chunk k from ring_vw into ring_xy;
//   cost      costs chunk k-1 from ring_xy; afterwards terminal + control costs.
// Tile-major arrays: element (t, n) of noise / cc_scratch lives at
// ((n / 64) * T + t) * 64 + n % 64, i.e. the T x 64 block of one wave is contiguous.
// The three cooperating roles of one workgroup of the pipelined rollout (threads [0, 192*W)): the
// whole kernel below after its noise-generating workgroups have branched off, and the exact
// exact re-execution path of a tile of the time-parallel kernel whose vote failed follows the same schedule
// (rollout_scan_exact_kernel.h: scan_exact_reexecute).
template <int C, bool POW2RES, bool CC_LDS>
__device__ __forceinline__ void pipe_tile_body(DevParams P, const uint16_t* __restrict__ cells16,
                                               const float2* __restrict__ noise, const float2* __restrict__ u,
                                               float* __restrict__ costs, float* __restrict__ w_rel,
                                               float* __restrict__ tile_beta, double* __restrict__ cc_scratch,
                                               int map_bytes, const int W) {
  extern __shared__ double2 uos[];
  [[maybe_unused]] const bool stamp_wg = blockIdx.x == 5;
  MPPI_STAMP(stamp_wg && threadIdx.x == 0, 0);
  // these few waves are the critical path of the iteration; the noise of the NEXT
  // iteration is generated concurrently by thousands of throughput-oriented waves on
  // the same SIMDs: win the issue arbitration against them
  __builtin_amdgcn_s_setprio(3);
  const int T = P.n_steps, N = P.n_local;
  // batched handle: all triples of a workgroup belong to one problem (host: W divides inst_tiles)
  u = select_instance(P, u, P.inst ? (int)(blockIdx.x * W) / P.inst_tiles : 0);
  float2* us = reinterpret_cast<float2*>(uos + T);
  uint16_t* lds_map = reinterpret_cast<uint16_t*>(uos + T + (T + 1) / 2);
  char* ring_base = reinterpret_cast<char*>(lds_map) + map_bytes;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = wave / W;  // 0 state, 1 cost, 2 producer
  const int triple = wave - role * W;
  using Ring = PipeRing<C>;
  float2* ring_xy = reinterpret_cast<float2*>(ring_base + (size_t)triple * Ring::kBytesPerPair);
  double2* ring_qd = reinterpret_cast<double2*>(ring_xy + 2 * Ring::kHalf);
  uint16_t* ring_cell = reinterpret_cast<uint16_t*>(ring_qd + 2 * Ring::kHalf);
  // control-cost products of this triple's 64 rollouts, [T][64] float64, when LDS has room
  double* cc_lds = reinterpret_cast<double*>(ring_base + (size_t)W * Ring::kBytesPerPair) + (size_t)triple * T * 64;

  // Only the producer waves read the staged controls (us) and control-cost ratios (uos): each of
  // them stages the arrays itself (the same values to the same addresses when there are several)
  // and goes on without a workgroup barrier -- a wave's LDS operations complete in order --
  // while the state and cost waves are already copying the map window.
  // (round 5: its first two chunks of noise are requested BEFORE the controls -- two round trips to memory one after
  //  the other put the producer's first chunk, and with it the first barrier, ~2k cycles behind the window copy)
  constexpr int kFirstNoise = 2 * C;
  float2 e_first[kFirstNoise];
  if (role == 2) {
    const int tile0 = blockIdx.x * W + triple;
    const float2* col0 = noise + (tile0 * 64 < N ? (size_t)tile0 * T * 64 + lane : (size_t)0);
#pragma unroll
    for (int j = 0; j < kFirstNoise; ++j) e_first[j] = col0[(size_t)min(j, T - 1) * 64];
    for (int t = lane; t < T; t += 64) {
      const float2 ut = u[t];
      us[t] = ut;
      uos[t] = make_double2((double)ut.x / P.s0sq, (double)ut.y / P.s1sq);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  [[maybe_unused]] const int stamp_base = 64 + 64 * role;
  MPPI_STAMP(stamp_wg && triple == 0, stamp_base + 0);
  copy_window_to_lds_direct(P, cells16, lds_map, 0, 128 * W);  // waves of roles 0 and 1; all vectors in flight
  MPPI_STAMP(stamp_wg && triple == 0, stamp_base + 1);

  const int tile = blockIdx.x * W + triple;  // 64 consecutive rollouts
  const int n = tile * 64 + lane;
  const bool live = n < N;
  const size_t tile_base = (size_t)tile * T * 64 + lane;  // + t*64: element (t, n)
  const int K = (T + C - 1) / C;
  // barrier schedule (identical for the three roles): one after the producer's
  // chunk 0, then one per k = 0..K

  if (role == 0) {
    const PipeWindow<POW2RES> win(P, lds_map);
    PipeState st = pipe_state_init(P);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the window has landed
    __syncthreads();  // controls of chunk 0 are in the ring; the window is in LDS
    st.cell = win.lookup(st.x, st.y);
    MPPI_STAMP(stamp_wg && triple == 0, stamp_base + 2);
    for (int k = 0; k <= K; ++k) {
      if (k < K) {
        const double2* in_qd = ring_qd + (size_t)(k & 1) * Ring::kHalf;
        float2* out_xy = ring_xy + (size_t)(k & 1) * Ring::kHalf;
        uint16_t* out_cell = ring_cell + (size_t)(k & 1) * Ring::kHalf;
        if (T - k * C >= C) pipe_state_chunk<C, POW2RES, false>(P, win, st, in_qd, out_xy, out_cell, lane);
        else pipe_state_tail<POW2RES, false>(P, win, st, in_qd, out_xy, out_cell, lane, T - k * C);
      }
      MPPI_STAMP(stamp_wg && triple == 0 && k < 32, stamp_base + 3 + k);
      __syncthreads();
    }
    MPPI_STAMP(stamp_wg && triple == 0, stamp_base + 40);
  } else if (role == 2) {
    // the tile past N (if any) reads the last valid tile's noise and writes nothing
    const bool tile_ok = tile * 64 < N;
    const float2* col = noise + (tile_ok ? tile_base : (size_t)0);
    double* my_cc = CC_LDS ? cc_lds + lane : cc_scratch + tile_base;
    float2 e_cur[C], e_nxt[C];
    // Round 5: the staged controls and ratios of the whole chunk are read first (the same addresses in every lane:
    // broadcasts) and everything is stored after the arithmetic -- one wait per chunk.  Before, every step read its
    // u[t], waited, stored, read its ratio inside a predicated block, waited again (each wait also drains the store
    // in front of it): ~350 cycles per step, the slowest of the three roles (stamps: profiles/r05_pipe_notes.md).
    // Steps past the horizon repeat step T - 1: the same products to the same address.
    auto produce = [&](int chunk, const float2 (&e)[C]) {
      double2* out_qd = ring_qd + (size_t)(chunk & 1) * Ring::kHalf;
      const double dt64 = (double)P.dt;
      float2 ut[C];
      double2 ur[C];
#pragma unroll
      for (int j = 0; j < C; ++j) {
        const int t = min(chunk * C + j, T - 1);
        ut[j] = us[t];
        ur[j] = uos[t];
      }
      double cc[C];
#pragma unroll
      for (int j = 0; j < C; ++j) {
        out_qd[j * 64 + lane] = make_double2(dt64 * (double)clip_f32(ut[j].x + e[j].x, P.v_lo, P.v_hi),
                                             dt64 * (double)clip_f32(ut[j].y + e[j].y, P.w_lo, P.w_hi));
        cc[j] = control_cost(P, ur[j], e[j]);
      }
      if (tile_ok) {
#pragma unroll
        for (int j = 0; j < C; ++j) my_cc[(size_t)min(chunk * C + j, T - 1) * 64] = cc[j];
      }
    };
#pragma unroll
    for (int j = 0; j < C; ++j) e_cur[j] = e_first[j];
#pragma unroll
    for (int j = 0; j < C; ++j) e_nxt[j] = e_first[C + j];
    produce(0, e_cur);
    MPPI_STAMP(stamp_wg && triple == 0, stamp_base + 2);
    __syncthreads();
    for (int k = 0; k <= K; ++k) {
#pragma unroll
      for (int j = 0; j < C; ++j) e_cur[j] = e_nxt[j];
#pragma unroll
      for (int j = 0; j < C; ++j) e_nxt[j] = col[(size_t)min((k + 2) * C + j, T - 1) * 64];
      if (k + 1 < K) produce(k + 1, e_cur);
      MPPI_STAMP(stamp_wg && triple == 0 && k < 32, stamp_base + 3 + k);
      __syncthreads();
    }
    MPPI_STAMP(stamp_wg && triple == 0, stamp_base + 40);
  } else {
    const double dt64 = (double)P.dt, gt2 = (double)P.gt2;
    float cost = 0.0f;
    double d2 = 1e9;
    bool done = false, reached = false;
    const double* my_cc = CC_LDS ? cc_lds + lane : cc_scratch + (live ? tile_base : (size_t)lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the window has landed
    __syncthreads();
    MPPI_STAMP(stamp_wg && triple == 0, stamp_base + 2);
    for (int k = 0; k <= K; ++k) {
      if (k >= 1) {
        const int t0 = (k - 1) * C;
        const float2* in_xy = ring_xy + (size_t)((k - 1) & 1) * Ring::kHalf;
        const uint16_t* in_cell = ring_cell + (size_t)((k - 1) & 1) * Ring::kHalf;
        const int count = min(C, T - t0);
        auto cost_step = [&](int j) {
          float2 xy = in_xy[j * 64 + lane];
          // obstacle / unknown bits (14, 15) of the cell the step STARTED in (mppi.py:971-998)
          uint32_t fl = (uint32_t)in_cell[j * 64 + lane] >> 14;
          double dx = (double)(P.xg - xy.x), dy = (double)(P.yg - xy.y);
          double nd2 = fma(dx, dx, dy * dy);
          float c1 = (float)((double)cost + fma(P.dist_weight, sqrt_newton_f64(nd2), dt64));
          c1 = c1 + ((fl & 1u) ? P.obs_cost : 0.0f);
          c1 = c1 + ((fl & 2u) ? P.unk_cost : 0.0f);
          bool hit = nd2 <= gt2;
          bool act = !done;
          cost = act ? c1 : cost;
          d2 = act ? nd2 : d2;
          reached = reached || (act && hit);
          done = done || hit;
        };
        if (count == C) {  // straight-line: the LDS reads of the whole chunk are hoisted
#pragma unroll
          for (int j = 0; j < C; ++j) cost_step(j);
        } else {
          for (int j = 0; j < count; ++j) cost_step(j);
        }
      }
      MPPI_STAMP(stamp_wg && triple == 0 && k < 32, stamp_base + 3 + k);
      __syncthreads();
    }
    MPPI_STAMP(stamp_wg && triple == 0, stamp_base + 40);
    // terminal cost, then the control cost of all T steps (mppi.py:1005-1009); the
    // products were written by the producer wave of this workgroup before its last barrier
    double term = (reached ? 0.0 : 1.0) * sqrt(d2) / P.v_post_den;
    cost = (float)((double)cost + term);
    // Two batches of loads in flight (from the global scratch they come back from L2 with a long latency), in two
    // register sets used in turn (round 5; before, one set was copied into the other after every batch -- two moves
    // per step on a chain of three instructions: 52 cycles per addition, now ~30).  Starting the loads inside the step
    // loop made the compiler keep the batch registers live across it and cost 7 us; wider single batches were no better.
    constexpr int kTailBatch = 24;
    double ca[kTailBatch], cb[kTailBatch];
    // (a whole batch inside the horizon: one base address and immediate offsets; the clamped form of the last batch
    //  costs ~6 address instructions per load -- as many as the three-instruction addition it feeds, twice over)
    auto tail_load = [&](double (&dst)[kTailBatch], int t0) {
      if (t0 + kTailBatch <= T) {
        const double* at = my_cc + (size_t)t0 * 64;
#pragma unroll
        for (int j = 0; j < kTailBatch; ++j) dst[j] = at[j * 64];
      } else {
#pragma unroll
        for (int j = 0; j < kTailBatch; ++j) dst[j] = my_cc[(size_t)min(t0 + j, T - 1) * 64];
      }
    };
    auto tail_add = [&](const double (&src)[kTailBatch], int t0) {
      if (t0 + kTailBatch <= T) {
#pragma unroll
        for (int j = 0; j < kTailBatch; ++j) cost = (float)((double)cost + src[j]);
      } else {
#pragma unroll
        for (int j = 0; j < kTailBatch; ++j)
          if (t0 + j < T) cost = (float)((double)cost + src[j]);
      }
    };
    // (three register sets, two batches ahead -- enough to cover the L2 latency of the global scratch at T = 200,
    //  where this walk runs at 57 cycles per addition -- made the LAUNCH 21 us slower in round 5: the kernel's register
    //  allocation is one for all roles and was capped at 128, and 144 batch registers moved the state role's; with round
    //  6's launch bound of two triples -- 256 registers -- it fits and changes nothing: 36.4 vs 36.5 us at the T = 200 shard)
    tail_load(ca, 0);
    for (int t0 = 0; t0 < T; t0 += 2 * kTailBatch) {
      tail_load(cb, t0 + kTailBatch);
      tail_add(ca, t0);
      if (t0 + kTailBatch >= T) break;
      tail_load(ca, t0 + 2 * kTailBatch);
      tail_add(cb, t0 + kTailBatch);
    }
    MPPI_STAMP(stamp_wg && triple == 0, stamp_base + 41);
    if (live) costs[n] = cost;
    // first half of the control update (update_kernels.h): weights relative to the tile's minimum
    if (tile * 64 < N) emit_tile_weights(cost, live, P.lambda, n, tile, w_rel, tile_beta);
    MPPI_STAMP(stamp_wg && triple == 0, stamp_base + 42);
  }
}

// (at most two wave triples per workgroup -- launch_plan.h: try_launch_pipe --: six waves, so that the three roles' one
//  register allocation may use 256 registers per lane instead of the 128 a workgroup of 1024 threads would leave)
constexpr int kPipeMaxTriples = 2;
template <int C, bool POW2RES, bool CC_LDS>
__global__ __launch_bounds__(192 * kPipeMaxTriples) void k_rollout_pipe(DevParams P, const uint16_t* __restrict__ cells16,
                               const float2* __restrict__ noise, const float2* __restrict__ u,
                               float* __restrict__ costs, float* __restrict__ w_rel,
                               float* __restrict__ tile_beta, double* __restrict__ cc_scratch, int map_bytes,
                               int n_rollout_blocks, NoiseJob next_noise) {
  if ((int)blockIdx.x >= n_rollout_blocks) {
    // spare workgroups: the noise of the NEXT iteration, into the other noise buffer.
    // (Also giving every rollout group a fourth, noise-generating wave was measured: the
    // extra waves on the rollout CUs cost the critical waves more than they saved.)
    MPPI_STAMP(threadIdx.x == 0 && ((int)blockIdx.x == n_rollout_blocks || blockIdx.x == gridDim.x - 1),
               (int)blockIdx.x == n_rollout_blocks ? 16 : 18);
    if (next_noise.out)
      noise_generate<true>(next_noise, (blockIdx.x - n_rollout_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6),
                     (gridDim.x - n_rollout_blocks) * (blockDim.x >> 6));
    MPPI_STAMP(threadIdx.x == 0 && ((int)blockIdx.x == n_rollout_blocks || blockIdx.x == gridDim.x - 1),
               (int)blockIdx.x == n_rollout_blocks ? 17 : 19);
    return;
  }
  pipe_tile_body<C, POW2RES, CC_LDS>(P, cells16, noise, u, costs, w_rel, tile_beta, cc_scratch, map_bytes,
                                     (int)blockDim.x / 192);
}

// Bitonic sort of the workgroup's LDS array sc[0..m_pow2), descending (mppi.py:716-741 sorts
// the M sample costs before averaging the worst ceil(M*alpha)).  When every thread owns
// exactly one element the compare-exchange steps whose partner sits in the same wave
// (distance < 64) run in registers through lane shuffles: of the 55 steps for 1024 elements
// only 10 need LDS and a workgroup barrier (57 -> 25 us at M = 1024).
__device__ __forceinline__ void bitonic_sort_desc(float* sc, int m_pow2) {
  if ((int)blockDim.x == m_pow2) {
    const int i = threadIdx.x;
    float v = sc[i];
    for (int k = 2; k <= m_pow2; k <<= 1) {
      const bool desc = ((i & k) == 0);
      for (int j = k >> 1; j > 0; j >>= 1) {
        float other;
        if (j >= 64) {  // partner in another wave: through LDS
          sc[i] = v;
          __syncthreads();
          other = sc[i ^ j];
          __syncthreads();
        } else {
          other = __shfl_xor(v, j, 64);
        }
        // the lower index keeps the larger value in a descending run, the smaller otherwise
        const bool keep_max = (((i & j) == 0) == desc);
        v = keep_max ? fmaxf(v, other) : fminf(v, other);
      }
    }
    sc[i] = v;
    __syncthreads();
    return;
  }
  for (int k = 2; k <= m_pow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < m_pow2; i += blockDim.x) {
        int p = i ^ j;
        if (p > i) {
          float a = sc[i], b = sc[p];
          bool desc = ((i & k) == 0);
          if (desc ? (a < b) : (a > b)) { sc[i] = b; sc[p] = a; }
        }
      }
      __syncthreads();
    }
}

// -------------------------------------------------------------------------
// Stochastic rollouts (CVaR over M traction samples).  One workgroup per
// control sample n; lane m (strided if M > blockDim) owns traction sample m.
// Cost order (mppi.py:690-713): per step stage, obstacle, unknown; control cost
// of all T steps; terminal cost.  Then per n: sort descending if alpha < 1 and
// average the first ceil(M*alpha) with the reference's strided tree
// (mppi.py:716-755).  rollout_oversized_numba (M > 1024) is this kernel with a
// strided lane loop; its swap-without-compare 'sort' (mppi.py:881-895) is a
// defect and is not reproduced.
// LDS: [T] double2 control ratios | [Mp] float sample costs (Mp = next pow2).
// -------------------------------------------------------------------------
template <bool EXACT>
__global__ void k_rollout_tdm(DevParams P, const uint32_t* __restrict__ cellsM,
                              const float2* __restrict__ noise, const float2* __restrict__ u,
                              float* __restrict__ costs, float* __restrict__ sample_costs, int m_pow2) {
  extern __shared__ double2 uos[];
  float* sc = reinterpret_cast<float*>(uos + P.n_steps);
  u = select_instance(P, u, P.inst ? (int)blockIdx.x / P.n_inst : 0);
  stage_control_ratios(P, u, uos);
  const int n = blockIdx.x;
  const int T = P.n_steps, M = P.n_grids;

  for (int m = threadIdx.x; m < m_pow2; m += blockDim.x) {
    if (m >= M) {
      sc[m] = -__builtin_inff();  // padding sorts to the tail
      continue;
    }
    float x = P.x0, y = P.y0, th = P.th0;
    float cost = 0.0f;
    double d2 = 1e9;
    bool reached = false;
    for (int t = 0; t < T; ++t) {
      float2 e = noise[tile_index(t, n, T)];
      float2 ut = u[t];
      int xi = clamp_index(floordiv_to_int(x - P.xlo, P.res, P.inv_res), P.cols);
      int yi = clamp_index(floordiv_to_int(y - P.ylo, P.res, P.inv_res), P.rows);
      uint32_t cell = cellsM[(size_t)(yi * P.cols + xi) * M + m];
      float v = clip_f32(ut.x + e.x, P.v_lo, P.v_hi);
      float w = clip_f32(ut.y + e.y, P.w_lo, P.w_hi);
      StepOut o = unicycle_step<EXACT>(P, x, y, th, v, w, (int)(int8_t)(cell & 0xff),
                                       (int)(int8_t)((cell >> 8) & 0xff));
      x = o.x; y = o.y; th = o.th; d2 = o.d2;
      cost = add_stage_cost<EXACT>(P, cost, o.d2, (double)P.dt);
      cost = cost + (float)(int8_t)((cell >> 16) & 0xff) * P.obs_cost;
      cost = cost + (float)(int8_t)(cell >> 24) * P.unk_cost;
      if (o.d2 <= (double)P.gt2) { reached = true; break; }
    }
    for (int t = 0; t < T; ++t) {
      double cc = control_cost(P, uos[t], noise[tile_index(t, n, T)]);
      cost = EXACT ? (float)((double)cost + cc) : cost + (float)cc;
    }
    double term = (reached ? 0.0 : 1.0) * sqrt(d2) / P.v_post_den;
    cost = EXACT ? (float)((double)cost + term) : cost + (float)term;
    sc[m] = cost;
    if (sample_costs) sample_costs[(size_t)n * M + m] = cost;
  }
  __syncthreads();

  if (P.cvar_alpha < 1.0f) bitonic_sort_desc(sc, m_pow2);  // over the padded power-of-two array
  // strided tree sum of the first numel entries, float32, as mppi.py:744-751
  const int numel = P.numel;
  for (int s = 1; s < numel; s <<= 1) {
    for (int i = threadIdx.x; i < M; i += blockDim.x)
      if ((i % (2 * s) == 0) && (i + s < numel)) sc[i] = sc[i] + sc[i + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // shared[0]/numel: float32 / int -> float64 -> float32 store (mppi.py:755)
    costs[n] = (float)((double)sc[0] / (double)numel);
  }
}

// -------------------------------------------------------------------------
// Stochastic rollouts, throughput-tuned variant (exact math, bounded heading
// increment).  With N*M/64 waves (8 per SIMD at N=4096, M=128) this mode is bound by
// VALU throughput, so the instruction count per (n, m, t) is what matters:
//   * everything that depends only on (n, t) -- clipped controls, dt*v and dt*w in
//     float64, the control-cost products -- is computed once per workgroup into LDS
//     instead of once per lane;
//   * (cos, sin) by rotation with the exact increment of the rounded heading;
//   * plain floor(x * 2^k) for power-of-two resolutions;
//   * no divergent break: a `done` predicate, and a wave-uniform early exit.
// Same rounding points as k_rollout_tdm (the generic kernel stays as the fallback).
// LDS: [T] double2 {dt*v, dt*w} | [T] double control-cost products | [Mp] float costs.
// -------------------------------------------------------------------------
// Workgroups past the N control samples (the end of the grid: they start as the first rollout
// workgroups retire) generate the noise of the NEXT iteration into the other buffer, which saves
// the stand-alone generator's launch (5.9 us of 161 at C3).
// COSTF32 (MPPI_MATH_FAST): as in k_rollout_fused<cost f32> the state keeps the reference's rounding
// points (every lane walks its OWN sampled map: an ulp of drift is another cell) and the cost side is
// float32; the control cost of a control sample is the same for its M traction samples: one float64
// sum per workgroup, added once (the reference adds its T terms to every sample's running cost).
template <bool POW2RES, bool COSTF32 = false>
__global__ void k_rollout_tdm_fast(DevParams P, const uint32_t* __restrict__ cellsM,
                                   const float2* __restrict__ noise, const float2* __restrict__ u,
                                   float* __restrict__ costs, float* __restrict__ sample_costs, int m_pow2,
                                   int n_rollout_blocks, NoiseJob next_noise) {
  extern __shared__ double2 qd_sh[];
  if ((int)blockIdx.x >= n_rollout_blocks) {
    if (next_noise.out)
      noise_generate(next_noise, (blockIdx.x - n_rollout_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6),
                     (gridDim.x - n_rollout_blocks) * (blockDim.x >> 6));
    return;
  }
  const int T = P.n_steps, M = P.n_grids;
  double* cc_sh = reinterpret_cast<double*>(qd_sh + T);
  float* sc = reinterpret_cast<float*>(cc_sh + T);
  const int n = blockIdx.x;
  u = select_instance(P, u, P.inst ? n / P.n_inst : 0);
  const double dt64 = (double)P.dt;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    float2 ut = u[t];
    float2 e = noise[tile_index(t, n, T)];
    float v = clip_f32(ut.x + e.x, P.v_lo, P.v_hi);
    float w = clip_f32(ut.y + e.y, P.w_lo, P.w_hi);
    qd_sh[t] = make_double2(dt64 * (double)v, dt64 * (double)w);
    cc_sh[t] = control_cost(P, make_double2((double)ut.x / P.s0sq, (double)ut.y / P.s1sq), e);
  }
  __syncthreads();
  [[maybe_unused]] float cc_total = 0.0f;
  if constexpr (COSTF32) {  // (every wave sums for itself: T/64 reads per lane, no second barrier)
    double part = 0.0;
    for (int t = threadIdx.x & 63; t < T; t += 64) part += cc_sh[t];
    cc_total = (float)wave_sum_f64(part);
  }
  [[maybe_unused]] const float dwf = (float)P.dist_weight, gt2f = P.gt2;
  const double gt2 = (double)P.gt2;
  const float last_col = (float)(P.cols - 1), last_row = (float)(P.rows - 1);
  // Round 5 (instruction budget per (n, m, t), profiles/r05_c3_isa.md): the additive traction constants pinned in
  // vector registers (a v_fma_f64 takes ONE scalar operand: with ratio and lo both scalar the compiler copied lo into
  // a register pair again every step, two v_mov_b64 per step), both cell coordinates in one packed subtract and one
  // packed multiply, the square root without the select of an exact zero (sqrt_newton_nz_f64: added to >= dt).
  double lin_lo_v = P.lin_lo, ang_lo_v = P.ang_lo;
  asm volatile("" : "+v"(lin_lo_v), "+v"(ang_lo_v));
  const pipe_f2 map_lo = {P.xlo, P.ylo}, map_inv = {P.inv_res, P.inv_res};
  for (int m = threadIdx.x; m < m_pow2; m += blockDim.x) {
    if (m >= M) {
      sc[m] = -__builtin_inff();  // padding sorts to the tail
      continue;
    }
    float x = P.x0, y = P.y0, th = P.th0, cost = 0.0f;
    double x64 = (double)x, y64 = (double)y, th64 = (double)th, d2 = 1e9;
    double s, c;
    sincos_f64<false>(th64, s, c);
    bool done = false, reached = false;
    [[maybe_unused]] float d2f = 1e9f;
    auto step = [&](int t) {
      double2 qd = qd_sh[t];
      int xi, yi;
      if (POW2RES) {  // see cell_coord_pow2 (window = the whole map; fl(pos - lo) * inv_res is exact)
        const pipe_f2 q = (pipe_f2{x, y} - map_lo) * map_inv;
        xi = (int)__builtin_amdgcn_fmed3f(q.x, 0.0f, last_col);
        yi = (int)__builtin_amdgcn_fmed3f(q.y, 0.0f, last_row);
      } else {
        xi = clamp_index(floordiv_to_int(x - P.xlo, P.res, P.inv_res), P.cols);
        yi = clamp_index(floordiv_to_int(y - P.ylo, P.res, P.inv_res), P.rows);
      }
      uint32_t cell = cellsM[(size_t)(yi * P.cols + xi) * M + m];
      double vtr = fma(P.lin_ratio, (double)(int)(int8_t)(cell & 0xff), lin_lo_v);
      double wtr = fma(P.ang_ratio, (double)(int)(int8_t)((cell >> 8) & 0xff), ang_lo_v);
      float nx = (float)fma(vtr, qd.x * c, x64);
      float ny = (float)fma(vtr, qd.x * s, y64);
      float nth = (float)fma(wtr, qd.y, th64);
      double nd2 = 0.0;
      float c1;
      bool hit;
      [[maybe_unused]] float nd2f = 0.0f;
      if constexpr (COSTF32) {
        const float dxf = P.xg - nx, dyf = P.yg - ny;
        nd2f = fmaf(dxf, dxf, dyf * dyf);
        c1 = cost + fmaf(dwf, __builtin_sqrtf(nd2f), P.dt);
        hit = nd2f <= gt2f;
      } else {
        double dx = (double)(P.xg - nx), dy = (double)(P.yg - ny);
        nd2 = fma(dx, dx, dy * dy);
        c1 = (float)((double)cost + fma(P.dist_weight, sqrt_newton_nz_f64(nd2), dt64));
        hit = nd2 <= gt2;
      }
      c1 = c1 + (float)(int8_t)((cell >> 16) & 0xff) * P.obs_cost;
      c1 = c1 + (float)(int8_t)(cell >> 24) * P.unk_cost;
      bool act = !done;
      // the state may run on after the goal: only cost, d2 and the flags are frozen
      x = nx; y = ny; th = nth;
      x64 = (double)nx; y64 = (double)ny;
      double th_new = (double)nth;
      rotate_sincos_f64<true>(th_new - th64, s, c);
      th64 = th_new;
      cost = act ? c1 : cost;
      if constexpr (COSTF32) d2f = act ? nd2f : d2f;
      else d2 = act ? nd2 : d2;
      reached = reached || (act && hit);
      done = done || hit;
    };
    // straight-line groups of four steps; the early exit (every lane of the wave has reached
    // the goal) is tested once per group: it only ever skips work whose results are frozen
    int t = 0;
    for (; t + 4 <= T; t += 4) {
      step(t);
      step(t + 1);
      step(t + 2);
      step(t + 3);
      if (__all(done)) break;
    }
    if (!__all(done))
      for (; t < T; ++t) step(t);
    // control cost of all T steps, then the terminal cost (mppi.py:706-713)
    if constexpr (COSTF32) {
      cost = cost + cc_total;
      cost = cost + (reached ? 0.0f : __builtin_sqrtf(d2f) * (float)P.inv_v_post_den);
    } else {
      for (int t = 0; t < T; ++t) cost = (float)((double)cost + cc_sh[t]);
      double term = (reached ? 0.0 : 1.0) * sqrt(d2) / P.v_post_den;
      cost = (float)((double)cost + term);
    }
    sc[m] = cost;
    if (sample_costs) sample_costs[(size_t)n * M + m] = cost;
  }
  __syncthreads();
  if (P.cvar_alpha < 1.0f) bitonic_sort_desc(sc, m_pow2);
  const int numel = P.numel;
  for (int st = 1; st < numel; st <<= 1) {
    for (int i = threadIdx.x; i < M; i += blockDim.x)
      if ((i % (2 * st) == 0) && (i + st < numel)) sc[i] = sc[i] + sc[i + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) costs[n] = (float)((double)sc[0] / (double)numel);
}

// -------------------------------------------------------------------------
// CVaR of control sample n over ALL M = count * m_local traction samples when the samples are
// sharded over GPUs: slabs[r][n][l] is the cost of (n, sample r*m_local + l)
// as rank r's k_rollout_tdm* wrote it, all-gathered.  Same sort, same strided float32 tree, same
// float64 division as the tail of k_rollout_tdm (mppi.py:716-755): the result has the bits of
// the unsharded launch.
// -------------------------------------------------------------------------
__global__ void k_cvar_reduce(const float* __restrict__ slabs, int count, int n_total, int m_local, int numel,
                              float cvar_alpha, float* __restrict__ costs, float* __restrict__ sample_costs,
                              int m_pow2) {
  extern __shared__ float sc_red[];
  const int n = blockIdx.x, M = count * m_local;
  for (int m = threadIdx.x; m < m_pow2; m += blockDim.x) {
    float v = -__builtin_inff();  // padding sorts to the tail
    if (m < M) {
      const int r = m / m_local, l = m - r * m_local;
      v = slabs[((size_t)r * n_total + n) * m_local + l];
      if (sample_costs) sample_costs[(size_t)n * M + m] = v;
    }
    sc_red[m] = v;
  }
  __syncthreads();
  if (cvar_alpha < 1.0f) bitonic_sort_desc(sc_red, m_pow2);
  for (int st = 1; st < numel; st <<= 1) {
    for (int i = threadIdx.x; i < M; i += blockDim.x)
      if ((i % (2 * st) == 0) && (i + st < numel)) sc_red[i] = sc_red[i] + sc_red[i + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) costs[n] = (float)((double)sc_red[0] / (double)numel);
}

// -------------------------------------------------------------------------
// barebone notebook rollout: nominal unicycle, quadratic distance cost, disc
// obstacles tested at the post-step position.
// -------------------------------------------------------------------------
// ROT (round 6; exact math, host-proved |dt * w| <= 0.36 rad, T <= 2000): (cos, sin) of the heading by rotation with the
// exact increment of the float32-rounded heading instead of a full float64 sincos per step (the reference's only
// published timing is this kernel's configuration: N = 1000, T = 50 -- 16 waves walking 50 dependent steps), the noise
// in batches of eight loads, and the state keeps integrating past the goal (only the cost is frozen), as in
// k_rollout_fused.  Same operations on the same operands for everything that reaches the cost.
// KD (round 6): the disc count as a compile-time bound -- >= 0: exactly the discs 0 .. KD-1 are tested, slots past
// n_obstacles hold a disc nobody can touch (the same additions of +0.0) -- so that a batch of eight steps is ONE basic
// block and the scheduler may run step t's cost beside step t+1's state; -1: the run-time loop.
template <bool EXACT, bool ROT = false, int KD = -1>
__global__ __launch_bounds__(64) void k_rollout_barebone(DevParams P, const float2* __restrict__ obs_pos,
                                                         const float* __restrict__ obs_r,
                                                         const float2* __restrict__ noise,
                                                         const float2* __restrict__ u,
                                                         float* __restrict__ costs) {
  extern __shared__ double2 uos[];
  ktime_begin(P);
  // LDS: [T] double2 control ratios | [K] {float x, y, r, -} discs (a step looked each of them up in memory before:
  // two dependent scalar loads per disc and step on a wave that has nothing else to run)
  float4* discs = reinterpret_cast<float4*>(uos + P.n_steps);
  for (int k = threadIdx.x; k < P.n_obstacles; k += 64) {
    const float2 op = obs_pos[k];
    discs[k] = make_float4(op.x, op.y, obs_r[k], 0.0f);
  }
  if (KD > 0)  // (slots past the last disc: far away, radius 0 -- `diff > 0`, no hit, +0.0 added)
    for (int k = P.n_obstacles + (int)threadIdx.x; k < KD; k += 64) discs[k] = make_float4(1e18f, 1e18f, 0.0f, 0.0f);
  stage_control_ratios(P, u, uos);  // (ends with a barrier)
  const int n = blockIdx.x * 64 + threadIdx.x;
  const bool live = n < P.n_local;
  const int nn = live ? n : P.n_local - 1;
  const int T = P.n_steps;
  float x = P.x0, y = P.y0, th = P.th0;
  float cost = 0.0f;
  double d2 = 1e9;
  bool done = false, reached = false;
  [[maybe_unused]] double rs = 0.0, rc = 1.0;
  if (ROT) sincos_f64<false>((double)th, rs, rc);
  const float2* col = noise + tile_index(0, nn, T);  // this lane's column; rows are 64 apart
  auto step = [&](float2 ut, float2 e) {
    float v = clip_f32(ut.x + e.x, P.v_lo, P.v_hi);
    float w = clip_f32(ut.y + e.y, P.w_lo, P.w_hi);
    float nx, ny, nth;
    double nd2;
    float dtv = P.dt * v;  // float32 * float32 first (cell 3: dt_d*v_noisy*math.cos(...))
    if (EXACT) {
      double sn, cs;
      if (ROT) { sn = rs; cs = rc; }
      else sincos_f64<false>((double)th, sn, cs);
      nx = (float)fma((double)dtv, cs, (double)x);
      ny = (float)fma((double)dtv, sn, (double)y);
    } else {
      float sn, cs;
      sincosf(th, &sn, &cs);
      nx = fmaf(dtv, cs, x);
      ny = fmaf(dtv, sn, y);
    }
    nth = th + P.dt * w;
    double dx = (double)(P.xg - nx), dy = (double)(P.yg - ny);
    nd2 = fma(dx, dx, dy * dy);
    float c1 = (float)((double)cost + P.dist_weight * nd2);
#pragma unroll
    for (int k = 0; k < (KD >= 0 ? KD : P.n_obstacles); ++k) {
      const float4 op = discs[k];
      double ex = (double)(nx - op.x), ey = (double)(ny - op.y);
      double rr = (double)op.z * (double)op.z;
      double diff = fma(ex, ex, ey * ey) - rr;
      double hit = (diff > 0.0) ? 0.0 : 1.0;
      c1 = (float)((double)c1 + hit * (double)P.obs_cost);
    }
    if (ROT) {
      // (past the goal the state goes on -- nobody looks at it: cost, distance and the goal flag are frozen)
      rotate_sincos_f64((double)nth - (double)th, rs, rc);  // exact increment of the ROUNDED heading
      x = nx; y = ny; th = nth;
      const bool hit_goal = nd2 <= (double)P.gt2, act = !done;
      cost = act ? c1 : cost;
      d2 = act ? nd2 : d2;
      reached = reached || (act && hit_goal);
      done = done || hit_goal;
    } else if (!done) {
      x = nx; y = ny; th = nth; d2 = nd2; cost = c1;
      if (nd2 <= (double)P.gt2) { reached = true; done = true; }
    }
  };
  if (ROT) {
    float2 e_cur[kNoiseBatch], e_nxt[kNoiseBatch];
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) e_cur[j] = col[(size_t)min(j, T - 1) * 64];
    int t0 = 0;
    for (; t0 + kNoiseBatch <= T; t0 += kNoiseBatch) {
#pragma unroll
      for (int j = 0; j < kNoiseBatch; ++j) e_nxt[j] = col[(size_t)min(t0 + kNoiseBatch + j, T - 1) * 64];
      if constexpr (EXACT && KD >= 0) {
        // Round 6: a lone wave issues a DEPENDENT instruction every ~9 cycles, an independent one every ~5
        // (tools/microbench/icache_cold.hip, issue_cost.hip), and step by step this kernel is one long chain.  Per batch
        // of eight steps: first everything that does not depend on the position -- clipped controls, the heading (one
        // float32 addition per step) and the polynomials of the rotation's exact increments: eight independent streams --
        // then the positions with the rotation (three short chains), then what each step adds to the cost (independent
        // again), then the cost itself: the float32-rounded additions, 3 + 3 per disc dependent instructions per step
        // and nothing else in the chain.  The same operations on the same operands as step() above.
        double q64[kNoiseBatch], sd[kNoiseBatch], cd[kNoiseBatch];
#pragma unroll
        for (int j = 0; j < kNoiseBatch; ++j) {
          const float2 ut = u[t0 + j];
          const float v = clip_f32(ut.x + e_cur[j].x, P.v_lo, P.v_hi);
          const float w = clip_f32(ut.y + e_cur[j].y, P.w_lo, P.w_hi);
          q64[j] = (double)(P.dt * v);  // float32 * float32 first (cell 3: dt_d*v_noisy*math.cos(...))
          const float nth = th + P.dt * w;
          sincos_increment_f64((double)nth - (double)th, sd[j], cd[j]);  // exact increment of the ROUNDED heading
          th = nth;
        }
        // the positions (and the rotation): three short chains
        float nxa[kNoiseBatch], nya[kNoiseBatch];
#pragma unroll
        for (int j = 0; j < kNoiseBatch; ++j) {
          nxa[j] = (float)fma(q64[j], rc, (double)x);
          nya[j] = (float)fma(q64[j], rs, (double)y);
          apply_rotation_f64(sd[j], cd[j], rs, rc);
          x = nxa[j];
          y = nya[j];
        }
        // what each step adds to the cost -- distance term, one term per disc --: independent of each other
        double nd2a[kNoiseBatch], add0[kNoiseBatch], addk[kNoiseBatch][KD > 0 ? KD : 1];
#pragma unroll
        for (int j = 0; j < kNoiseBatch; ++j) {
          const double dx = (double)(P.xg - nxa[j]), dy = (double)(P.yg - nya[j]);
          nd2a[j] = fma(dx, dx, dy * dy);
          add0[j] = P.dist_weight * nd2a[j];
#pragma unroll
          for (int k = 0; k < KD; ++k) {
            const float4 op = discs[k];
            const double ex = (double)(nxa[j] - op.x), ey = (double)(nya[j] - op.y);
            const double rr = (double)op.z * (double)op.z;
            const double diff = fma(ex, ex, ey * ey) - rr;
            const double hit = (diff > 0.0) ? 0.0 : 1.0;
            addk[j][k] = hit * (double)P.obs_cost;
          }
        }
        // the cost: the float32-rounded additions in the reference's order -- the one long chain, nothing else in it
#pragma unroll
        for (int j = 0; j < kNoiseBatch; ++j) {
          float c1 = (float)((double)cost + add0[j]);
#pragma unroll
          for (int k = 0; k < KD; ++k) c1 = (float)((double)c1 + addk[j][k]);
          const bool hit_goal = nd2a[j] <= (double)P.gt2, act = !done;
          cost = act ? c1 : cost;
          d2 = act ? nd2a[j] : d2;
          reached = reached || (act && hit_goal);
          done = done || hit_goal;
        }
      } else {
#pragma unroll
        for (int j = 0; j < kNoiseBatch; ++j) step(u[t0 + j], e_cur[j]);
      }
#pragma unroll
      for (int j = 0; j < kNoiseBatch; ++j) e_cur[j] = e_nxt[j];
      if (__all(done)) break;
    }
    if (!__all(done))
      for (int t = t0; t < T; ++t) {  // (batch registers shifted down, not indexed: see k_rollout_fused)
        step(u[t], e_cur[0]);
#pragma unroll
        for (int j = 0; j + 1 < kNoiseBatch; ++j) e_cur[j] = e_cur[j + 1];
      }
  } else {
    for (int t = 0; t < T; ++t) {
      step(u[t], col[(size_t)t * 64]);
      if (__all(done)) break;
    }
  }
  cost = (float)((double)cost + (reached ? 0.0 : 1.0) * d2);
  int t0 = 0;
  for (; t0 + kNoiseBatch <= T; t0 += kNoiseBatch) {  // loads and products batched, the float32-rounded additions in order
    double cc[kNoiseBatch];
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) cc[j] = control_cost(P, uos[t0 + j], col[(size_t)(t0 + j) * 64]);
#pragma unroll
    for (int j = 0; j < kNoiseBatch; ++j) cost = (float)((double)cost + cc[j]);
  }
  for (int t = t0; t < T; ++t) cost = (float)((double)cost + control_cost(P, uos[t], col[(size_t)t * 64]));
  if (live) costs[n] = cost;
  ktime_end(P);
}

// -------------------------------------------------------------------------
// Visualisation rollouts -> states [V][T+1][3]
// across control noise (mppi.py:1194-1295): row 0 = u_cur, no noise, no clip;
// row b>0 = clip(u_prev + noise[b]); all on traction sample 0.
// across environments (mppi.py:1298-1351): u_cur over samples 0..V-1.
// MAPLESS: the barebone notebook's variant.
// -------------------------------------------------------------------------
template <bool ACROSS_ENVS, bool MAPLESS>
__global__ void k_state_rollout(DevParams P, const uint32_t* __restrict__ cells,
                                const float2* __restrict__ noise, const float2* __restrict__ u_prev,
                                const float2* __restrict__ u_cur, int n_vis, float* __restrict__ out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_vis) return;
  const int T = P.n_steps;
  const int M = ACROSS_ENVS ? P.n_grids : 1;
  const int m = ACROSS_ENVS ? b : 0;
  float* o = out + (size_t)b * (T + 1) * 3;
  float x = P.x0, y = P.y0, th = P.th0;
  o[0] = x; o[1] = y; o[2] = th;
  for (int t = 0; t < T; ++t) {
    float v, w;
    if (ACROSS_ENVS || b == 0) {
      v = u_cur[t].x;
      w = u_cur[t].y;
    } else {
      float2 e = noise[tile_index(t, b, T)];
      v = clip_f32(u_prev[t].x + e.x, P.v_lo, P.v_hi);
      w = clip_f32(u_prev[t].y + e.y, P.w_lo, P.w_hi);
    }
    if (MAPLESS) {
      double s, c;
      sincos_f64<false>((double)th, s, c);
      float dtv = P.dt * v;
      float nx = (float)fma((double)dtv, c, (double)x);
      float ny = (float)fma((double)dtv, s, (double)y);
      th = th + P.dt * w;
      x = nx; y = ny;
    } else {
      int xi = clamp_index(floordiv_to_int(x - P.xlo, P.res, P.inv_res), P.cols);
      int yi = clamp_index(floordiv_to_int(y - P.ylo, P.res, P.inv_res), P.rows);
      uint32_t cell = cells[(size_t)(yi * P.cols + xi) * M + m];
      StepOut so = unicycle_step<true>(P, x, y, th, v, w, (int)(int8_t)(cell & 0xff),
                                       (int)(int8_t)((cell >> 8) & 0xff));
      x = so.x; y = so.y; th = so.th;
    }
    o[3 * (t + 1)] = x; o[3 * (t + 1) + 1] = y; o[3 * (t + 1) + 2] = th;
  }
}

// ---- small helpers of the flag-synchronised kernels (rollout_scan*_kernel.h) ----------------------------------------
// interval barrier: this wave's LDS operations complete, its global loads stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// keeps the compiler from moving memory operations (LDS phases, early loads) across this point
__device__ __forceinline__ void pin_memory_order() { asm volatile("" ::: "memory"); }
// a compile-time index as a function argument (register sets used in turn)
template <int I>
struct PhaseTag {
  static constexpr int value = I;
};

}  // namespace mppi
