// rollout_spec_kernel.h -- k_rollout_spec: the latency-regime rollout with the heading chain taken
// off the map-lookup chain by SPECULATION ON THE TRACTION VALUE (gfx950, wave64).
//
// Replaces rollout_det_dyn_numba (mppi.py:916-1009) like k_rollout_pipe does, with the same
// rounding points and therefore the same bits; what changes is who waits for whom.
//
// In k_rollout_pipe one "state wave" per 64 rollouts walks, per step, the chain
//   position -> cell index -> LDS lookup -> traction -> heading update -> (cos, sin) rotation
//            -> position update
// (53 instructions, ~360 cycles per step measured, profiles/r02_stamps.md): the heading depends
// on the ANGULAR traction of the visited cell and the position on the heading, so nothing can run
// ahead.  But traction is piecewise constant, and over maps of nominal dynamics (the reference's
// own use_det_dynamics recipe, README.md:136-151) or large terrain patches a rollout sees the SAME
// (linear, angular) traction bytes for long stretches.  Here every tile of 64 rollouts assumes
// that all of its lookups return the bytes of the START cell:
//   producer wave P   noise -> clipped controls {dt*v, dt*w} (float64) and control-cost products,
//                     as in k_rollout_pipe; also copies the map window into LDS in bands of rows,
//                     each band just before the rollouts can reach it (the 40 KiB copy used to
//                     take 6.2k cycles of prologue);
//   heading wave  H   theta, (cos, sin) by the exact-increment rotation, and the products
//                     dt*v*cos, dt*v*sin -- with the ASSUMED angular traction: needs no lookup;
//   position wave V   x, y with the ASSUMED linear traction (a 3-instruction chain), and -- off
//                     that chain -- the lookup of every visited cell: checks the assumption,
//                     hands obstacle / unknown bits and the squared goal distance to
//   cost wave     C   sqrt, stage cost, penalties, goal test, float32-rounded accumulation; then
//                     terminal and control costs and the tile's half of the weight computation.
// Four waves, one per SIMD, one workgroup barrier per chunk of CH steps, each stage one chunk
// behind the previous: ~31 instructions per step on the busiest wave instead of 53, and no LDS
// latency in any loop-carried chain.
//
// When a lookup contradicts the assumption (V votes once per chunk), the tile falls back to the
// exact schedule of k_rollout_pipe from the start of that chunk: H retires, V becomes the state
// wave (it restores x, y from its own registers and theta, cos, sin from the snapshot H leaves
// in LDS at the start of every chunk; the producer's ring keeps four chunks so that the controls
// of the failed chunk are still there), the cost wave skips one interval.  Nothing computed under
// a wrong assumption is ever consumed: the cost wave is one chunk behind V, and V's vote comes
// before the barrier that would release its chunk.  Costs are therefore bit-identical to
// k_rollout_pipe / k_rollout_fused / the oracle on every map; only the time differs (a map whose
// traction changes from cell to cell falls back in the first chunk and pays ~2 intervals).
//
// Scheduling notes (measured, profiles/r02_stamps.md): a global load issued inside this kernel
// comes back after ~2.5k cycles (128 neighbouring workgroups are writing the next iteration's
// noise), an interval lasts ~1.5k.  So (1) the interval barrier is `s_waitcnt lgkmcnt(0);
// s_barrier` -- LDS traffic complete, global loads stay in flight (a __syncthreads() would
// drain them); (2) the producer keeps four chunks of noise and four bands of the window copy in
// flight in registers (its loop is unrolled by four so that the register sets are static);
// (3) every wave touches LDS in phases -- all reads of a chunk, compute, all writes -- because
// LDS operations retire in order and a read queued behind writes waits for them.
//
// LDS: [T] double2 ratios | [T] float2 u | map window | per tile rings (SpecRing) | [W] int fail |
//      (CC_LDS) per tile [K*CH][64] double control-cost products.
#pragma once
#include "rollout_kernels.h"

namespace mppi {

template <int CH>
struct SpecRing {
  static constexpr int kHalf = CH * 64;              // entries per chunk
  static constexpr int kQd = 4 * kHalf * 16;         // qd[4][CH][64] double2 {dt*v, dt*w}
  static constexpr int kPp = 2 * kHalf * 16;         // pp[2][CH][64] double2 {dt*v*cos, dt*v*sin}
  static constexpr int kNd2 = 2 * kHalf * 8;         // nd2[2][CH][64] double squared goal distance
  static constexpr int kFl = 2 * 64 * 4;             // fl[2][64] dword: 2 bits per step (obstacle | unknown << 1)
  static constexpr int kHs = 2 * 64 * (8 + 8 + 4);   // snapshot[2]: sin[64] double, cos[64] double, theta[64] float
  static constexpr int kBytesPerTile = kQd + kPp + kNd2 + kFl + kHs;
};

// interval barrier: this wave's LDS operations complete, its global loads stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// keeps the compiler from moving memory operations (LDS phases, early loads) across this point
__device__ __forceinline__ void pin_memory_order() { asm volatile("" ::: "memory"); }

// rows [row_begin, row_end) of the window, full width, by `n_threads` threads (tid 0..n_threads-1):
// batches of eight 16-byte loads per thread, indices clamped (the last vector is then written twice)
__device__ __forceinline__ void copy_window_rows(const DevParams& P, const uint16_t* __restrict__ cells16,
                                                 uint16_t* lds_map, int row_begin, int row_end, int tid,
                                                 int n_threads) {
  const int vpr = P.win_cols >> 3;  // 16-byte vectors per row
  const int total = (row_end - row_begin) * vpr;
  if (total <= 0 || tid < 0 || tid >= n_threads) return;
  const int src_pitch = P.pitch16 >> 3;
  const uint4* src = reinterpret_cast<const uint4*>(cells16) + ((size_t)P.win_r0 * P.pitch16 + P.win_c0) / 8;
  uint4* dst = reinterpret_cast<uint4*>(lds_map);
  const float inv_vpr = 1.0f / (float)vpr;
  for (int i0 = tid; i0 < total; i0 += 8 * n_threads) {
    uint4 v[8];
    int at[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = min(i0 + k * n_threads, total - 1);
      int r = (int)((float)i * inv_vpr);  // i < 2^20: within one of the quotient
      r -= (r * vpr > i);
      r += ((r + 1) * vpr <= i);
      const int c = i - r * vpr;
      at[k] = (row_begin + r) * vpr + c;
      v[k] = src[(size_t)(row_begin + r) * src_pitch + c];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[at[k]] = v[k];
  }
}

// One band of the progressive window copy: the rows at distance (r_in, r_out] from row0 on both
// sides, clipped to the window.  band_load() issues the loads (BV 16-byte vectors per thread; a
// wider band's surplus is copied at once), band_store() writes them to LDS -- intervals later.
// Always loads and stores BV vectors: an empty band re-copies one vector of the start row
// (branch-free: the register sets stay in registers).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // (native vector: register arrays of it stay in registers)

// One band of the progressive window copy: the rows at distance (r_in, r_out] from row0 on both
// sides, clipped to the window.  band_load() issues the loads (BV 16-byte vectors per thread; a
// wider band's surplus is copied at once), band_store() writes them to LDS -- intervals later.
// Vector k of thread `tid` is vector i = tid + k*n_threads of the band in row-major order over
// (side 0 rows, then side 1 rows): its (row within the band, column vector) = (i / vpr, i % vpr)
// does not depend on the band and is computed once (BandLanes).  Always loads and stores BV
// vectors: past the end of the band a thread re-copies one vector of the start row (branch-free:
// the register sets stay in registers).
template <int BV>
struct BandLanes {
  int r[BV], c[BV];
  __device__ __forceinline__ void init(int vpr, int tid, int n_threads) {
    const float inv_vpr = 1.0f / (float)vpr;
#pragma unroll
    for (int k = 0; k < BV; ++k) {
      const int i = tid + k * n_threads;
      int q = (int)((float)i * inv_vpr);  // i < 2^20: within one of the quotient
      q -= (q * vpr > i);
      q += ((q + 1) * vpr <= i);
      r[k] = q;
      c[k] = i - q * vpr;
    }
  }
};

template <int BV>
__device__ __forceinline__ void band_load(const DevParams& P, const uint16_t* __restrict__ cells16, int row0, int r_in,
                                          int r_out, int tid, int n_threads, uint16_t* lds_map,
                                          const BandLanes<BV>& L, u32x4 (&v)[BV], int (&at)[BV]) {
  const int vpr = P.win_cols >> 3;
  // side 0: rows [row0 + r_in + 1, row0 + r_out]; side 1: rows [row0 - r_out, row0 - r_in - 1]
  const int a0 = min(row0 + r_in + 1, P.win_rows), a1 = min(row0 + r_out + 1, P.win_rows);
  const int b0 = max(row0 - r_out, 0), b1 = max(row0 - r_in, 0);
  const int rows_a = max(a1 - a0, 0), rows_b = max(b1 - b0, 0);
  const int src_pitch = P.pitch16 >> 3;
  const u32x4* src = reinterpret_cast<const u32x4*>(cells16) + ((size_t)P.win_r0 * P.pitch16 + P.win_c0) / 8;
  const int total = (rows_a + rows_b) * vpr;
  if (total > BV * n_threads) {  // (uniform; rare: 6 vectors per lane at C2)
    const float inv_vpr = 1.0f / (float)vpr;
    for (int i = tid + BV * n_threads; i < total; i += n_threads) {
      int q = (int)((float)i * inv_vpr);
      q -= (q * vpr > i);
      q += ((q + 1) * vpr <= i);
      const int row = q < rows_a ? a0 + q : b0 + (q - rows_a);
      const int c = i - q * vpr;
      reinterpret_cast<u32x4*>(lds_map)[row * vpr + c] = src[(size_t)row * src_pitch + c];
    }
  }
#pragma unroll
  for (int k = 0; k < BV; ++k) {
    const int q = L.r[k];
    const bool in_a = q < rows_a, in_band = q < rows_a + rows_b;
    const int row = in_a ? a0 + q : (in_band ? b0 + (q - rows_a) : row0);
    const int c = in_band ? L.c[k] : 0;
    at[k] = row * vpr + c;
    v[k] = src[row * src_pitch + c];
  }
}

template <int BV>
__device__ __forceinline__ void band_store(uint16_t* lds_map, const u32x4 (&v)[BV], const int (&at)[BV]) {
  u32x4* dst = reinterpret_cast<u32x4*>(lds_map);
#pragma unroll
  for (int k = 0; k < BV; ++k) dst[at[k]] = v[k];
}

template <int I>
struct PhaseTag {
  static constexpr int value = I;
};

// Roles of the waves of one workgroup (W tiles, tile i = waves i, W+i, 2W+i, 3W+i):
// 0 producer P, 1 heading H, 2 position / state V, 3 cost C.  W is a template parameter: the
// producer holds ~230 registers of loads in flight when it is alone on its SIMD (W = 1), ~190 when
// two tiles share the workgroup.
template <int CH, bool POW2RES, bool CC_LDS, int W>
__global__ __launch_bounds__(256 * W) void k_rollout_spec(DevParams P, const uint16_t* __restrict__ cells16,
                               const float2* __restrict__ noise, const float2* __restrict__ u,
                               float* __restrict__ costs, float* __restrict__ w_rel,
                               float* __restrict__ tile_beta, double* __restrict__ cc_scratch, int map_bytes,
                               int n_rollout_blocks, int speculate, NoiseJob next_noise) {
  extern __shared__ double2 uos[];
  if ((int)blockIdx.x >= n_rollout_blocks) {
    // spare workgroups: the noise of the NEXT iteration, into the other noise buffer
    MPPI_STAMP(threadIdx.x == 0 && ((int)blockIdx.x == n_rollout_blocks || blockIdx.x == gridDim.x - 1),
               (int)blockIdx.x == n_rollout_blocks ? 16 : 18);
    if (next_noise.out)
      noise_generate<true>(next_noise, (blockIdx.x - n_rollout_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6),
                     (gridDim.x - n_rollout_blocks) * (blockDim.x >> 6));
    MPPI_STAMP(threadIdx.x == 0 && ((int)blockIdx.x == n_rollout_blocks || blockIdx.x == gridDim.x - 1),
               (int)blockIdx.x == n_rollout_blocks ? 17 : 19);
    return;
  }
  [[maybe_unused]] const bool stamp_wg = blockIdx.x == 5;
  MPPI_STAMP(stamp_wg && threadIdx.x == 0, 0);
  __builtin_amdgcn_s_setprio(3);  // win the issue arbitration against the noise-generating waves
  const int T = P.n_steps, N = P.n_local;
  u = select_instance(P, u, P.inst ? (int)(blockIdx.x * W) / P.inst_tiles : 0);
  // staged controls, padded to whole chunks (entries past T are zero): every read is unclamped
  const int Tp = (T + 7) & ~7;
  float2* us = reinterpret_cast<float2*>(uos + Tp);
  uint16_t* lds_map = reinterpret_cast<uint16_t*>(uos + Tp + Tp / 2);
  char* ring_base = reinterpret_cast<char*>(lds_map) + map_bytes;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = wave / W;
  const int tw = wave - role * W;  // tile within the workgroup
  using Ring = SpecRing<CH>;
  char* my_ring = ring_base + (size_t)tw * Ring::kBytesPerTile;
  double2* ring_qd = reinterpret_cast<double2*>(my_ring);
  double2* ring_pp = reinterpret_cast<double2*>(my_ring + Ring::kQd);
  double* ring_nd2 = reinterpret_cast<double*>(my_ring + Ring::kQd + Ring::kPp);
  uint32_t* ring_fl = reinterpret_cast<uint32_t*>(my_ring + Ring::kQd + Ring::kPp + Ring::kNd2);
  double* snap_s = reinterpret_cast<double*>(my_ring + Ring::kQd + Ring::kPp + Ring::kNd2 + Ring::kFl);
  double* snap_c = snap_s + 2 * 64;
  float* snap_th = reinterpret_cast<float*>(snap_c + 2 * 64);
  int* fail_flags = reinterpret_cast<int*>(ring_base + (size_t)W * Ring::kBytesPerTile);  // [W] (16 bytes reserved)
  const int K = (T + CH - 1) / CH;
  // control-cost products of this tile, [K*CH][64] float64 (rows past T are never written nor used)
  double* cc_lds = reinterpret_cast<double*>(ring_base + (size_t)W * Ring::kBytesPerTile + 16) + (size_t)tw * K * CH * 64;
  [[maybe_unused]] const int stamp_base = 64 + 64 * role;
  MPPI_STAMP(stamp_wg && tw == 0, stamp_base + 0);

  const int tile = blockIdx.x * W + tw;  // 64 consecutive rollouts
  const int n = tile * 64 + lane;
  const bool live = n < N;
  const size_t tile_base = (size_t)tile * T * 64 + lane;  // + t*64: element (t, n)

  // Geometry of the progressive window copy: the rows within radius_at(i) of the start row must
  // be in LDS when interval i begins (V integrates chunk i-1 then: positions at most i*CH steps
  // from the start; on the exact schedule without speculation it integrates chunk i: lead 1).
  const int row0 = clamp_index(floordiv_to_int(P.y0 - P.ylo, P.res, P.inv_res) - P.win_r0, P.win_rows);
  const float step_cells = P.win_step_cells;
  const int lead = speculate ? 0 : 1;
  auto radius_at = [&](int i) {
    if (!P.win_progressive) return 1 << 20;  // no bound on the spread: everything up front
    const float r = ceilf((float)((i + lead) * CH) * step_cells) + 3.0f;
    return (int)fminf(r, 1.0e6f);
  };

  if (role == 0) {
    // ---------------------------------------------------------------- producer (+ window copier)
    const bool tile_ok = tile * 64 < N;  // the tile past N (if any) reads the last valid tile's noise, writes nothing
    const float2* col = noise + (tile_ok ? tile_base : (size_t)0);
    const int copy_tid = tw * 64 + lane, copy_threads = W * 64;
    float2 e[4][CH];   // e[c & 3]: noise of chunk c, requested three intervals before it is consumed
    constexpr int BV = W == 1 ? 8 : 4;  // vectors per lane and band (360 per band at C2: 64 or 128 lanes copy)
    u32x4 bv[4][BV];   // bv[i & 3]: band of window rows requested in interval i, stored in interval i+2
    int bat[4][BV];
    BandLanes<BV> lanes;
    lanes.init(P.win_cols >> 3, copy_tid, copy_threads);
    // the noise buffers are padded by 8 chunks: rows past the horizon are read (and ignored)
    // without clamping, at immediate offsets from one running pointer
    auto load_noise = [&](float2 (&dst)[CH], int chunk) {
      const float2* at = col + (size_t)chunk * CH * 64;
#pragma unroll
      for (int j = 0; j < CH; ++j) dst[j] = at[j * 64];
    };
    load_noise(e[0], 0);
    load_noise(e[1], 1);
    load_noise(e[2], 2);
    load_noise(e[3], 3);
    // (register sets 2 and 3 are stored in intervals 0 and 1, before any band exists: harmless content)
    band_load<BV>(P, cells16, row0, 0, 0, copy_tid, copy_threads, lds_map, lanes, bv[2], bat[2]);
    band_load<BV>(P, cells16, row0, 0, 0, copy_tid, copy_threads, lds_map, lanes, bv[3], bat[3]);
    // each producer stages the controls itself (same values to the same addresses when there are
    // several): a wave's LDS operations complete in order, no workgroup barrier needed before use
    for (int t = lane; t < K * CH; t += 64) {
      const float2 ut = t < T ? u[t] : make_float2(0.0f, 0.0f);
      us[t] = ut;
      uos[t] = make_double2((double)ut.x / P.s0sq, (double)ut.y / P.s1sq);
    }
    if (lane == 0) fail_flags[tw] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double* my_cc = CC_LDS ? cc_lds + lane : cc_scratch + tile_base;
    auto produce = [&](int chunk, const float2 (&en)[CH]) {
      double2* out_qd = ring_qd + (size_t)(chunk & 3) * Ring::kHalf;
      const double dt64 = (double)P.dt;
      const float2* us_c = us + chunk * CH;
      const double2* uos_c = uos + chunk * CH;
      double2 qd[CH];
      double cc[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float2 ut = us_c[j];  // steps past the horizon are produced (zero controls) and ignored
        qd[j] = make_double2(dt64 * (double)clip_f32(ut.x + en[j].x, P.v_lo, P.v_hi),
                             dt64 * (double)clip_f32(ut.y + en[j].y, P.w_lo, P.w_hi));
        cc[j] = control_cost(P, uos_c[j], en[j]);
      }
      pin_memory_order();
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        out_qd[j * 64 + lane] = qd[j];
        if (tile_ok && chunk * CH + j < T) my_cc[(size_t)(chunk * CH + j) * 64] = cc[j];
      }
    };
    produce(0, e[0]);
    MPPI_STAMP(stamp_wg && tw == 0, stamp_base + 2);
    __syncthreads();
    // one interval; PH == k & 3 selects the register sets statically
    auto interval = [&](auto ph, int k) -> bool {
      constexpr int PH = decltype(ph)::value;
      // rows the rollouts can reach in interval k+3, requested now, stored at the end of interval k+2
      band_load<BV>(P, cells16, row0, radius_at(k + 2), radius_at(k + 3), copy_tid, copy_threads, lds_map, lanes, bv[PH],
                    bat[PH]);
      MPPI_STAMP(stamp_wg && tw == 0 && k < 32, 1024 + 4 * k);
      load_noise(e[PH], k + 4);  // (this set held chunk k, consumed in interval k-1)
      pin_memory_order();
      MPPI_STAMP(stamp_wg && tw == 0 && k < 32, 1024 + 4 * k + 1);
      if (k + 1 < K) produce(k + 1, e[(PH + 1) & 3]);
      MPPI_STAMP(stamp_wg && tw == 0 && k < 32, 1024 + 4 * k + 2);
      // products in global scratch: all of them out before the cost wave's tail reads them (once,
      // right after the last chunk; the interval barrier itself does not wait for global memory)
      if (!CC_LDS && k + 1 == K - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      band_store<BV>(lds_map, bv[(PH + 2) & 3], bat[(PH + 2) & 3]);
      MPPI_STAMP(stamp_wg && tw == 0 && k < 32, stamp_base + 3 + k);
      lds_barrier();
      MPPI_STAMP(stamp_wg && tw == 0 && k < 32, 1024 + 4 * k + 3);
      const int failed_at = fail_flags[tw] - 1;  // chunk whose assumption failed, or -1
      return k >= (failed_at >= 0 ? K + 2 : K + 1);
    };
    for (int k = 0;; k += 4) {
      if (interval(PhaseTag<0>(), k)) break;
      if (interval(PhaseTag<1>(), k + 1)) break;
      if (interval(PhaseTag<2>(), k + 2)) break;
      if (interval(PhaseTag<3>(), k + 3)) break;
    }
    MPPI_STAMP(stamp_wg && tw == 0, stamp_base + 40);
    return;
  }

  // rows within radius_at(2) of the start row, by the waves of the other three roles (the bands the
  // producers request reach LDS from interval 2 on)
  {
    const int r = radius_at(2);
    copy_window_rows(P, cells16, lds_map, max(row0 - r, 0), min(row0 + r + 1, P.win_rows),
                     (int)threadIdx.x - 64 * W, 192 * W);
  }
  MPPI_STAMP(stamp_wg && tw == 0, stamp_base + 1);
  const float win_c0f = (float)P.win_c0, win_r0f = (float)P.win_r0;
  const float win_last_col = (float)(P.win_cols - 1), win_last_row = (float)(P.win_rows - 1);
  const int win_pitch_bytes = 2 * P.win_cols;
  const char* lds_bytes = reinterpret_cast<const char*>(lds_map);
  auto lookup = [&](float x, float y) -> uint32_t {
    // the window holds every cell reachable within the horizon (host-proved); the clamp is for
    // memory safety only
    int xi, yi;
    if (POW2RES) {  // res is a power of two: see cell_coord_pow2
      xi = cell_coord_pow2(x, P.xlo, P.inv_res, win_c0f, win_last_col);
      yi = cell_coord_pow2(y, P.ylo, P.inv_res, win_r0f, win_last_row);
    } else {
      xi = clamp_index(floordiv_to_int(x - P.xlo, P.res, P.inv_res) - P.win_c0, P.win_cols);
      yi = clamp_index(floordiv_to_int(y - P.ylo, P.res, P.inv_res) - P.win_r0, P.win_rows);
    }
    return *reinterpret_cast<const uint16_t*>(lds_bytes + (__mul24(yi, win_pitch_bytes) + (xi << 1)));
  };
  __syncthreads();  // controls of chunk 0 in the ring, rows around the start cell in LDS
  MPPI_STAMP(stamp_wg && tw == 0, stamp_base + 2);
  // the assumption: every visited cell carries the traction bytes of the start cell
  const uint32_t ref = lookup(P.x0, P.y0) & 0x3fffu;
  const double vtr0 = fma(P.lin_ratio, (double)(int)(ref & 127u), P.lin_lo);
  const double wtr0 = fma(P.ang_ratio, (double)(int)((ref >> 7) & 127u), P.ang_lo);

  if (role == 1) {
    // ---------------------------------------------------------------- heading wave (assumed traction)
    if (!speculate) return;  // (a retired wave leaves the barrier count)
    float th = P.th0;
    double th64 = (double)th, s, c;
    sincos_f64<false>(th64, s, c);
    for (int k = 0; k < K; ++k) {
      if (fail_flags[tw] != 0) return;  // the tile fell back: V carries the heading itself from here
      const double2* in_qd = ring_qd + (size_t)(k & 3) * Ring::kHalf;
      double2* out_pp = ring_pp + (size_t)(k & 1) * Ring::kHalf;
      double2 qd[CH], pp[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) qd[j] = in_qd[j * 64 + lane];
      pin_memory_order();
      const double s_in = s, c_in = c;
      const float th_in = th;
      // (a) the heading chain, (b) the increment polynomials of all steps side by side, (c) the
      //     rotation chain with the products
      double sd[CH], cd[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        th = (float)fma(wtr0, qd[j].y, th64);
        const double th_new = (double)th;
        sd[j] = th_new - th64;  // exact increment of the ROUNDED heading
        th64 = th_new;
      }
#pragma unroll
      for (int j = 0; j < CH; ++j) sincos_increment_f64(sd[j], sd[j], cd[j]);
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        pp[j] = make_double2(qd[j].x * c, qd[j].x * s);
        apply_rotation_f64(sd[j], cd[j], s, c);
      }
      pin_memory_order();
      // the state this chunk started from, for V should the chunk have to be redone
      snap_s[(k & 1) * 64 + lane] = s_in;
      snap_c[(k & 1) * 64 + lane] = c_in;
      snap_th[(k & 1) * 64 + lane] = th_in;
#pragma unroll
      for (int j = 0; j < CH; ++j) out_pp[j * 64 + lane] = pp[j];
      MPPI_STAMP(stamp_wg && tw == 0 && k < 32, stamp_base + 3 + k);
      lds_barrier();
    }
    MPPI_STAMP(stamp_wg && tw == 0, stamp_base + 40);
    return;
  }

  if (role == 2) {
    // ---------------------------------------------------------------- position wave / state wave
    float x = P.x0, y = P.y0;
    double x64 = (double)x, y64 = (double)y;
    float th = P.th0;  // exact schedule only
    double th64 = (double)th, s = 0.0, c = 0.0;
    bool fallback = !speculate;
    if (fallback) sincos_f64<false>(th64, s, c);
    bool stuck = false;  // this rollout sits in a cell of zero linear traction
    int next = 0;  // next chunk to integrate
    for (int k = 0;; ++k) {
      if (!fallback && k >= 1 && next < K) {
        // chunk `next` = k-1 under the assumption: x, y are a 3-instruction chain, the lookups trail it
        const float xs = x, ys = y;
        const double2* in_pp = ring_pp + (size_t)(next & 1) * Ring::kHalf;
        double* out_nd2 = ring_nd2 + (size_t)(next & 1) * Ring::kHalf;
        double2 pp[CH];
        double nd2[CH];
        uint32_t c16[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) pp[j] = in_pp[j * 64 + lane];
        pin_memory_order();
        // (a) the position chain, (b) the lookups of all steps side by side, (c) per lane: a cell of
        //     zero linear traction is the end of the road -- the rollout never moves again, whatever
        //     its heading does -- so from there on its position is the one it entered the cell with
        //     (exactly what the reference computes), and it no longer tests the assumption;
        //     (d) goal distances
        float xa[CH + 1], ya[CH + 1];
        xa[0] = x;
        ya[0] = y;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const float xn = (float)fma(vtr0, pp[j].x, x64), yn = (float)fma(vtr0, pp[j].y, y64);
          x = stuck ? x : xn;
          y = stuck ? y : yn;
          x64 = (double)x;
          y64 = (double)y;
          xa[j + 1] = x;
          ya[j + 1] = y;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) c16[j] = lookup(xa[j], ya[j]);  // the cell step j STARTS in
        uint32_t bad = 0, fl = 0;
        const int steps_left = T - next * CH;  // steps of this chunk inside the horizon
        bool now_stuck = stuck;
        float fx = xa[0], fy = ya[0];
        uint32_t fc = c16[0];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const bool zero = (int)(c16[j] & 127u) == P.lin_zero_byte;
          const bool first = zero && !now_stuck;
          fx = first ? xa[j] : fx;
          fy = first ? ya[j] : fy;
          fc = first ? c16[j] : fc;
          now_stuck = now_stuck || zero;
          xa[j + 1] = now_stuck ? fx : xa[j + 1];
          ya[j + 1] = now_stuck ? fy : ya[j + 1];
          const uint32_t cell = now_stuck ? fc : c16[j];
          bad |= (!now_stuck && j < steps_left) ? ((cell ^ ref) & 0x3fffu) : 0u;
          fl |= (cell >> 14) << (2 * j);  // obstacle | unknown << 1 of the cell step j left
        }
        stuck = now_stuck;
        x = xa[CH];
        y = ya[CH];
        x64 = (double)x;
        y64 = (double)y;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const double dx = (double)(P.xg - xa[j + 1]), dy = (double)(P.yg - ya[j + 1]);
          nd2[j] = fma(dx, dx, dy * dy);
        }
        pin_memory_order();
#pragma unroll
        for (int j = 0; j < CH; ++j) out_nd2[j * 64 + lane] = nd2[j];
        ring_fl[(next & 1) * 64 + lane] = fl;
        if (__any(bad != 0)) {
          // some lane met other traction bytes: nothing of this chunk may be used.  Back to its start.
          if (lane == 0) {
            fail_flags[tw] = next + 1;
            if (P.spec_failures) {  // (once per tile: exact from here on)
              atomicAdd_system(P.spec_failures, 1u);
              __threadfence_system();  // performed before this launch can be seen to have finished
            }
          }
          x = xs; y = ys; x64 = (double)x; y64 = (double)y;
          th = snap_th[(next & 1) * 64 + lane];
          s = snap_s[(next & 1) * 64 + lane];
          c = snap_c[(next & 1) * 64 + lane];
          th64 = (double)th;
          fallback = true;  // (this wave knows at once; the others read the flag after the barrier)
        } else {
          ++next;
        }
      } else if (fallback && next < K && (speculate ? k >= next + 2 : k >= next)) {
        // exact schedule: lookup, traction, heading and position in one chain (k_rollout_pipe's state wave)
        const double2* in_qd = ring_qd + (size_t)(next & 3) * Ring::kHalf;
        double* out_nd2 = ring_nd2 + (size_t)(next & 1) * Ring::kHalf;
        double2 qd[CH];
        double nd2[CH];
        uint32_t fl = 0;
#pragma unroll
        for (int j = 0; j < CH; ++j) qd[j] = in_qd[j * 64 + lane];
        pin_memory_order();
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const uint32_t c16 = lookup(x, y);
          const double vtr = fma(P.lin_ratio, (double)(int)(c16 & 127u), P.lin_lo);
          const double wtr = fma(P.ang_ratio, (double)(int)((c16 >> 7) & 127u), P.ang_lo);
          x = (float)fma(vtr, qd[j].x * c, x64);
          y = (float)fma(vtr, qd[j].x * s, y64);
          th = (float)fma(wtr, qd[j].y, th64);
          x64 = (double)x;
          y64 = (double)y;
          const double th_new = (double)th;
          rotate_sincos_f64(th_new - th64, s, c);
          th64 = th_new;
          const double dx = (double)(P.xg - x), dy = (double)(P.yg - y);
          nd2[j] = fma(dx, dx, dy * dy);
          fl |= (c16 >> 14) << (2 * j);
        }
        pin_memory_order();
#pragma unroll
        for (int j = 0; j < CH; ++j) out_nd2[j * 64 + lane] = nd2[j];
        ring_fl[(next & 1) * 64 + lane] = fl;
        ++next;
      }
      MPPI_STAMP(stamp_wg && tw == 0 && k < 32, stamp_base + 3 + k);
      lds_barrier();
      const int failed_at = fail_flags[tw] - 1;
      if (k >= (failed_at >= 0 ? K + 2 : K + 1)) break;
    }
    MPPI_STAMP(stamp_wg && tw == 0, stamp_base + 40);
    return;
  }

  // -------------------------------------------------------------------- cost wave
  {
    const double dt64 = (double)P.dt, gt2 = (double)P.gt2;
    float cost = 0.0f;
    double d2 = 1e9;
    bool done = false, reached = false;
    int next = 0;  // next chunk to cost
    for (int k = 0;; ++k) {
      const int failed_at = fail_flags[tw] - 1;
      // chunk c is complete after interval c+1 (assumption held) or c+2 (exact schedule after a
      // failure); launched without speculation, after interval c
      const bool exact = failed_at >= 0 && next >= failed_at;
      const int ready_after = speculate ? next + (exact ? 2 : 1) : next;
      if (next < K && k > ready_after) {
        const int t0 = next * CH;
        const double* in_nd2 = ring_nd2 + (size_t)(next & 1) * Ring::kHalf;
        const int count = min(CH, T - t0);
        double nd2[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) nd2[j] = in_nd2[j * 64 + lane];
        const uint32_t fl = ring_fl[(next & 1) * 64 + lane];
        pin_memory_order();
        // (a) square roots and stage costs of all steps side by side, (b) the accumulation chain
        //     (steps past the horizon carry finite garbage: computed, not kept)
        double stage[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) stage[j] = fma(P.dist_weight, sqrt_newton_f64(nd2[j]), dt64);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          float c1 = (float)((double)cost + stage[j]);
          c1 = c1 + (((fl >> (2 * j)) & 1u) ? P.obs_cost : 0.0f);  // bits of the cell the step STARTED in
          c1 = c1 + (((fl >> (2 * j)) & 2u) ? P.unk_cost : 0.0f);  // (mppi.py:971-998)
          const bool hit = nd2[j] <= gt2, act = !done && j < count;
          cost = act ? c1 : cost;
          d2 = act ? nd2[j] : d2;
          reached = reached || (act && hit);
          done = done || (hit && j < count);
        }
        ++next;
      }
      MPPI_STAMP(stamp_wg && tw == 0 && k < 32, stamp_base + 3 + k);
      lds_barrier();
      const int failed_now = fail_flags[tw] - 1;
      if (k >= (failed_now >= 0 ? K + 2 : K + 1)) break;
    }
    MPPI_STAMP(stamp_wg && tw == 0, stamp_base + 40);
    // terminal cost, then the control cost of all T steps (mppi.py:1005-1009); the products were
    // written by the producer wave of this workgroup before its last barrier
    const double term = (reached ? 0.0 : 1.0) * sqrt(d2) / P.v_post_den;
    cost = (float)((double)cost + term);
    if (CC_LDS) {
      // chunks of CH float32-rounded additions, the next chunk's LDS reads in flight meanwhile
      // (row-padded array: every read is unconditional, at an immediate offset from one base)
      const double* my_cc = cc_lds + lane;
      double va[CH], vb[CH];
#pragma unroll
      for (int j = 0; j < CH; ++j) va[j] = my_cc[(size_t)j * 64];
      for (int ck = 0; ck < K; ++ck) {
        const double* nxt = my_cc + (size_t)min(ck + 1, K - 1) * CH * 64;
#pragma unroll
        for (int j = 0; j < CH; ++j) vb[j] = nxt[(size_t)j * 64];
        if ((ck + 1) * CH <= T) {
#pragma unroll
          for (int j = 0; j < CH; ++j) cost = (float)((double)cost + va[j]);
        } else {
#pragma unroll
          for (int j = 0; j < CH; ++j)
            if (ck * CH + j < T) cost = (float)((double)cost + va[j]);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) va[j] = vb[j];
      }
    } else {
      __threadfence_block();
      const double* my_cc = cc_scratch + (live ? tile_base : (size_t)lane);
      constexpr int kTailBatch = 24;
      double ca[kTailBatch], cb[kTailBatch];
#pragma unroll
      for (int j = 0; j < kTailBatch; ++j) ca[j] = my_cc[(size_t)min(j, T - 1) * 64];
#pragma unroll
      for (int j = 0; j < kTailBatch; ++j) cb[j] = my_cc[(size_t)min(kTailBatch + j, T - 1) * 64];
      for (int t0 = 0; t0 < T; t0 += kTailBatch) {
        if (t0 + kTailBatch <= T) {
#pragma unroll
          for (int j = 0; j < kTailBatch; ++j) cost = (float)((double)cost + ca[j]);
        } else {
#pragma unroll
          for (int j = 0; j < kTailBatch; ++j)
            if (t0 + j < T) cost = (float)((double)cost + ca[j]);
        }
#pragma unroll
        for (int j = 0; j < kTailBatch; ++j) ca[j] = cb[j];
#pragma unroll
        for (int j = 0; j < kTailBatch; ++j) cb[j] = my_cc[(size_t)min(t0 + 2 * kTailBatch + j, T - 1) * 64];
      }
    }
    MPPI_STAMP(stamp_wg && tw == 0, stamp_base + 41);
    if (live) costs[n] = cost;
    // first half of the control update (update_kernels.h): weights relative to the tile's minimum
    if (tile * 64 < N) emit_tile_weights(cost, live, P.lambda, n, tile, w_rel, tile_beta);
    MPPI_STAMP(stamp_wg && tw == 0, stamp_base + 42);
  }
}

}  // namespace mppi
