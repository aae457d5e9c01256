// rollout_scan_kernel.h -- k_rollout_scan: the deterministic-dynamics rollout parallel over the
// HORIZON (gfx950, wave64).  MPPI_MATH_FAST only: tolerance mode, not bit-identical.
//
// Replaces rollout_det_dyn_numba (mppi.py:916-1009) and, with GEN, sample_noise_numba
// (mppi.py:1354-1370) and, with packets, the pass of update_useq_numba over the noise
// (mppi.py:1177-1181).  tests/scan_model.py states the algorithm in numpy, checked against the
// oracle on the CPU; this file follows it operation for operation.
//
// Why.  The reference's rollout is a chain of T dependent steps per control sample; with one tile
// of 64 rollouts per CU (N = 8192: the north-star shard) every design that walks that chain is
// bound by the issue rate of a handful of waves (k_rollout_deep: 226 cycles per step, 18 us per
// launch, 11 % of the HBM roofline).  The chain exists because heading and position feed on the
// traction of the visited cell.  Under the assumption the speculative kernels already make --
// every visited cell carries the traction (vtr0, wtr0) of the start cell -- it is not a chain:
//     theta_t = theta_0 + wtr0*dt * sum_{k<t} w_k                      (mppi.py:990)
//     x_t     = x_0     + vtr0*dt * sum_{k<t} v_k cos(theta_k)         (mppi.py:988; y likewise)
// two prefix sums over the horizon.  One workgroup takes a tile of 64 rollouts, one WAVE a chunk
// of CH consecutive steps of it (lane = rollout): T = 100 is 13 waves of 8 steps, 1664 independent
// waves where the chain had 128.
//
// Phases (workgroup barriers between them; S = per-(chunk, lane) sums in LDS):
//   A  noise of the chunk (GEN: Philox counter blocks computed here -- the noise never exists in
//      memory; else: read, tile-major), clipped controls, control-cost terms -> LDS, heading
//      increments, their sum -> S
//   B  heading at the chunk start = theta_0 + sum of S over earlier chunks (float64); heading per
//      step, hardware sin / cos (of the float64-reduced fraction of a turn), position increments,
//      float32 prefix inside the chunk, their sum -> S
//   C  position at the chunk start (float64 across chunks), positions, cell lookups (global
//      gathers: off every critical path here), squared goal distances, stage costs
//   D  walk over the chunk's steps (selects, no branch): the vote on the assumption, the goal
//      break (mppi.py:1000-1002), rollouts FROZEN in a cell of zero linear traction (the padding
//      ring: they never move again, and pay the stage cost of where they stand for the rest of the
//      horizon, exactly as the reference computes) -> what every step adds to the cost, the chunk's
//      event into a 2-bit field of one LDS word per lane (ds_or)
//   D' the first chunk with an event decides for the later ones (reached: they add nothing;
//      frozen: they add the frozen addends); addends of every step -> LDS
//   E  ONE wave walks the T x 2 additions in the reference's order with the reference's float32
//      rounding after every step (CHAIN64: in float64, rounded per step, as the reference's CPU
//      path does), then the terminal cost, then the T control-cost additions (mppi.py:1005-1009).
//      Sums formed side by side cannot reproduce T sequential roundings; measured against the
//      oracle (tests/test_scan_model.py): tree sums leave 1 % of the costs beyond 1e-6 relative,
//      the walk leaves 0.02 % and 89 % bit-identical.  Then cost, tile weights (update_kernels.h).
//   F  (packets) every wave reduces w_rel * noise of its steps over the 64 rollouts: the tile's
//      contribution to the update, [T][tiles] float2 -- k_combine_tiles needs no pass over the noise.
// A failed vote (a rollout still moving meets a cell whose traction differs): phase E is replaced
// by a sequential float32 rollout of the tile by one wave, noise handed over through LDS.  The
// host stops launching this kernel on maps where most tiles fail (review_speculation).
//
// LDS: | rec [W][64] {CH stage addends (real), CH penalty addends (float)} (phases A-C: sums S) |
//      | ccr [W][64][CH] float | frec [W][64] {real, float} | term [64] double | evw [64] u32 | flags |
#pragma once
#include <type_traits>
#include "rollout_spec_kernel.h"

namespace mppi {

// tile contributions to the control update, written by phase F / read by k_combine_tiles
struct ScanPackets {
  float2* tnum;  // [T][n_tiles]: sum over the tile's rollouts of w_rel * noise(t)   (nullptr: no phase F)
  float* tden;   // [n_tiles]:    sum of w_rel
  int n_tiles;
};

template <int CH, bool CHAIN64>
struct ScanLds {
  using real = std::conditional_t<CHAIN64, double, float>;
  static constexpr int kRecBytes = CH * ((int)sizeof(real) + 4);  // per (chunk, lane)
  static constexpr int kFrecBytes = 16;
  __host__ __device__ static constexpr size_t rec(int W) { return (size_t)W * 64 * kRecBytes; }
  __host__ __device__ static constexpr size_t ccr(int W) { return (size_t)W * 64 * CH * 4; }
  __host__ __device__ static constexpr size_t frec(int W) { return (size_t)W * 64 * kFrecBytes; }
  __host__ __device__ static constexpr size_t total(int W) { return rec(W) + ccr(W) + frec(W) + 64 * 8 + 64 * 4 + 64; }
  // the three sums of phases A-C live where the records of phase D' go: [3][W][64] double
  static_assert(3 * 8 <= kRecBytes, "sums alias the records");
};

// sum over the 64 lanes, result in lane 63 (4 steps inside each row of 16 lanes, then row 0 -> 1,
// 2 -> 3 and rows 0..1 -> 2..3 through row_bcast): 6 DPP additions, fixed order
__device__ __forceinline__ float wave_sum_to_lane63_f32(float v) {
  v += dpp_f32<kDppXor1>(v);
  v += dpp_f32<kDppXor2>(v);
  v += dpp_f32<kDppHalfMirror>(v);
  v += dpp_f32<kDppMirror>(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));  // row_bcast:15
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));  // row_bcast:31
  return v;
}

// one Philox block -> the noise of steps (2*tp, 2*tp + 1) of global rollout `n_global`, with the
// very expressions of noise_row (rng_kernels.h): the standalone generator reproduces it bit for bit
__device__ __forceinline__ void scan_noise_pair(const NoiseJob& g, uint64_t epoch, unsigned int n_global,
                                                unsigned int pairs, unsigned int tp, float2& a, float2& b) {
  const uint64_t sub = (uint64_t)n_global * (uint64_t)pairs + (uint64_t)tp;
  const uint4 r = philox4x32_10(make_uint4((unsigned int)epoch, (unsigned int)(epoch >> 32), (unsigned int)sub,
                                           (unsigned int)(sub >> 32)),
                                make_uint2((unsigned int)g.seed, (unsigned int)(g.seed >> 32)));
  const float2 za = box_muller_fast(r.x, r.y), zb = box_muller_fast(r.z, r.w);
  a = make_float2(g.std0 * za.x, g.std1 * za.y);
  b = make_float2(g.std0 * zb.x, g.std1 * zb.y);
}

template <bool POW2RES>
__device__ __forceinline__ uint32_t scan_lookup(const DevParams& Q, const uint16_t* __restrict__ cells16, float x,
                                                float y) {
  int xi, yi;
  if (POW2RES) {
    xi = cell_coord_pow2(x, Q.xlo, Q.inv_res, 0.0f, (float)(Q.cols - 1));
    yi = cell_coord_pow2(y, Q.ylo, Q.inv_res, 0.0f, (float)(Q.rows - 1));
  } else {
    xi = clamp_index(floordiv_to_int(x - Q.xlo, Q.res, Q.inv_res), Q.cols);
    yi = clamp_index(floordiv_to_int(y - Q.ylo, Q.res, Q.inv_res), Q.rows);
  }
  return cells16[__mul24(yi, Q.pitch16) + xi];
}

// GEN: `gen` describes the Philox counters of THIS iteration's noise (out is ignored);
// !GEN: `noise` holds it (tile-major) and the spare workgroups (blockIdx >= n_rollout_blocks)
//       write the next iteration's (`next_noise`), as in k_rollout_deep.
template <int CH, bool POW2RES, bool GEN, bool CHAIN64>
__global__ __launch_bounds__(1024) void k_rollout_scan(DevParams P, const uint16_t* __restrict__ cells16,
                                                       const float2* __restrict__ noise, NoiseJob gen,
                                                       const float2* __restrict__ u, float* __restrict__ costs,
                                                       float* __restrict__ w_rel, float* __restrict__ tile_beta,
                                                       ScanPackets pk, int n_rollout_blocks, NoiseJob next_noise) {
  extern __shared__ double2 scan_lds[];
  if ((int)blockIdx.x >= n_rollout_blocks) {
    if (next_noise.out)
      noise_generate<true>(next_noise, (blockIdx.x - n_rollout_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6),
                           (gridDim.x - n_rollout_blocks) * (blockDim.x >> 6));
    return;
  }
  using L = ScanLds<CH, CHAIN64>;
  using real = typename L::real;
  const int c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // chunk of this wave
  const int lane = threadIdx.x & 63;
  const int W = (int)(blockDim.x >> 6);
  [[maybe_unused]] const bool stamp_wg = blockIdx.x == 5;
  [[maybe_unused]] const int stamp_base = 64 + 16 * c;
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 0);
  DevParams Q = P;
  const int tile = blockIdx.x;
  const float2* uq = select_instance(Q, u, Q.inst ? tile / Q.inst_tiles : 0);
  const int T = Q.n_steps, N = Q.n_local;
  const int n = tile * 64 + lane;
  const bool live = n < N;
  const int t0 = c * CH;

  char* base = reinterpret_cast<char*>(scan_lds);
  char* rec = base;                                                   // [W][64] records (phase D')
  double* sumS = reinterpret_cast<double*>(base);                     // [3][W][64] (phases A-C)
  float* ccr = reinterpret_cast<float*>(base + L::rec(W));            // [W][64][CH]
  char* frec = base + L::rec(W) + L::ccr(W);                          // [W][64] {real sg, float pen}
  double* term_sh = reinterpret_cast<double*>(frec + L::frec(W));     // [64]
  uint32_t* evw = reinterpret_cast<uint32_t*>(term_sh + 64);          // [64] 2 bits per chunk
  uint32_t* flags = evw + 64;                                         // [0] failed vote
  float* wsh = reinterpret_cast<float*>(flags + 4);                   // [..]: aliases nothing live in phase F
  (void)wsh;

  if (c == 0) {
    evw[lane] = 0u;
    if (lane == 0) flags[0] = 0u;
  }

  // the assumption: every visited cell carries the traction bytes of the start cell
  const uint32_t ref = scan_lookup<POW2RES>(Q, cells16, Q.x0, Q.y0) & 0x3fffu;

  // ---------------------------------------------------------------- A: noise, controls, heading increments
  float2 e[CH];
  if constexpr (GEN) {
    const uint64_t epoch = gen.epoch + (gen.gen_counter ? *gen.gen_counter : 0ull);
    const unsigned int pairs = (unsigned int)(T + 1) / 2u;
    const unsigned int n_global = (unsigned int)(gen.n_offset + min(n, N - 1));
#pragma unroll
    for (int jp = 0; jp < CH / 2; ++jp) {
      const unsigned int tp = (unsigned int)(t0 / 2 + jp);
      scan_noise_pair(gen, epoch, n_global, pairs, min(tp, pairs - 1u), e[2 * jp], e[2 * jp + 1]);
    }
  } else {
    const float2* col = noise + (size_t)tile * T * 64 + lane;
#pragma unroll
    for (int j = 0; j < CH; ++j) e[j] = col[(size_t)min(t0 + j, T - 1) * 64];
  }
  float2 ut[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const float2 g = uq[min(t0 + j, T - 1)];  // (uniform address: scalar loads)
    const bool valid = t0 + j < T;
    ut[j] = valid ? g : make_float2(0.0f, 0.0f);
    e[j] = valid ? e[j] : make_float2(0.0f, 0.0f);
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 1);
  const double vtr0 = fma(Q.lin_ratio, (double)(int)(ref & 127u), Q.lin_lo);
  const double wtr0 = fma(Q.ang_ratio, (double)(int)((ref >> 7) & 127u), Q.ang_lo);
  const float vtr0f = (float)vtr0;
  const double kth = wtr0 * (double)Q.dt;
  // lambda * (u0/s0^2 * e0 + u1/s1^2 * e1)   (mppi.py:1007-1009)
  const float k0 = (float)((double)Q.lambda / Q.s0sq), k1 = (float)((double)Q.lambda / Q.s1sq);
  float v[CH];
  double thl[CH];  // heading increments summed before step j (exclusive prefix inside the chunk)
  double th_sum = 0.0;
  {
    float cc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      v[j] = clip_f32(ut[j].x + e[j].x, Q.v_lo, Q.v_hi);
      const float w = clip_f32(ut[j].y + e[j].y, Q.w_lo, Q.w_hi);
      cc[j] = fmaf(k0 * ut[j].x, e[j].x, (k1 * ut[j].y) * e[j].y);
      thl[j] = th_sum;
      th_sum = fma(kth, (double)w, th_sum);
    }
    float4* dst = reinterpret_cast<float4*>(ccr + ((size_t)c * 64 + lane) * CH);
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) dst[q] = make_float4(cc[4 * q], cc[4 * q + 1], cc[4 * q + 2], cc[4 * q + 3]);
  }
  sumS[(size_t)c * 64 + lane] = th_sum;
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 2);
  lds_barrier();

  // ---------------------------------------------------------------- B: headings, position increments
  float lx[CH], ly[CH];  // inclusive prefix of the position increments inside the chunk
  {
    double th_base = (double)Q.th0;
    for (int i = 0; i < c; ++i) th_base += sumS[(size_t)i * 64 + lane];
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const double turns = (th_base + thl[j]) * 0.15915494309189535;  // 1 / (2 pi)
      const float fr = (float)__builtin_amdgcn_fract(turns);          // v_sin_f32 / v_cos_f32 take turns
      const float q = Q.dt * v[j];
      sx += vtr0f * (q * __builtin_amdgcn_cosf(fr));
      sy += vtr0f * (q * __builtin_amdgcn_sinf(fr));
      lx[j] = sx;
      ly[j] = sy;
    }
    sumS[(size_t)(W + c) * 64 + lane] = (double)sx;
    sumS[(size_t)(2 * W + c) * 64 + lane] = (double)sy;
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 3);
  lds_barrier();

  // ---------------------------------------------------------------- C: positions, lookups, stage costs
  real sg[CH];      // stage cost of step j: dt + dist_weight * distance after the step
  real n2[CH];      // squared goal distance after step j
  float pen[CH];    // obstacle / unknown penalties of the cell step j starts in
  real sg_pre0, n2_pre0;  // the same where the chunk starts (what a rollout frozen at step 0 pays)
  uint32_t zero_bits = 0, mism_bits = 0;
  {
    double bx = (double)Q.x0, by = (double)Q.y0;
    for (int i = 0; i < c; ++i) {
      bx += sumS[(size_t)(W + i) * 64 + lane];
      by += sumS[(size_t)(2 * W + i) * 64 + lane];
    }
    const float bxf = (float)bx, byf = (float)by;
    float xa[CH + 1], ya[CH + 1];
    xa[0] = bxf;
    ya[0] = byf;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      xa[j + 1] = bxf + lx[j];
      ya[j + 1] = byf + ly[j];
    }
    uint32_t cell[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) cell[j] = scan_lookup<POW2RES>(Q, cells16, xa[j], ya[j]);  // the cell step j STARTS in
    auto dist2 = [&](float x, float y) -> real {
      const real dx = (real)(Q.xg - x), dy = (real)(Q.yg - y);
      if constexpr (CHAIN64) return fma(dx, dx, dy * dy);
      else return fmaf(dx, dx, dy * dy);
    };
    auto stage = [&](real d2) -> real {
      if constexpr (CHAIN64) return fma(Q.dist_weight, sqrt_newton_nz_f64(d2), (double)Q.dt);
      else return fmaf((float)Q.dist_weight, __builtin_amdgcn_sqrtf(d2), Q.dt);
    };
    n2_pre0 = dist2(xa[0], ya[0]);
    sg_pre0 = stage(n2_pre0);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      n2[j] = dist2(xa[j + 1], ya[j + 1]);
      sg[j] = stage(n2[j]);
    }
    pin_memory_order();
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const uint32_t cl = cell[j];
      zero_bits |= ((int)(cl & 127u) == Q.lin_zero_byte ? 1u : 0u) << j;
      mism_bits |= (((cl ^ ref) & 0x3fffu) != 0u ? 1u : 0u) << j;
      pen[j] = __int_as_float(__float_as_int(Q.obs_cost) & __builtin_amdgcn_sbfe((int)cl, 14, 1)) +
               __int_as_float(__float_as_int(Q.unk_cost) & __builtin_amdgcn_sbfe((int)cl, 15, 1));
    }
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 4);

  // ---------------------------------------------------------------- D: the walk over the chunk's steps
  const real gt2 = (real)Q.gt2;
  real add_sg[CH];
  float add_pen[CH];
  bool alive = true, frozen = false, bad = false;
  uint32_t ev = 0;  // 0 none, 1 goal reached, 2 frozen (and not at the goal)
  real f_sg = (real)0, n2_end = (real)1e9;
  float f_pen = 0.0f;
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const bool valid = t0 + j < T;
    const real sg_pre = j == 0 ? sg_pre0 : sg[j - 1], n2_pre = j == 0 ? n2_pre0 : n2[j - 1];
    const bool z = valid && alive && !frozen && ((zero_bits >> j) & 1u);
    f_sg = z ? sg_pre : f_sg;
    f_pen = z ? pen[j] : f_pen;
    n2_end = z ? n2_pre : n2_end;
    frozen = frozen || z;
    const bool act = valid && alive;
    bad = bad || (act && !frozen && ((mism_bits >> j) & 1u));
    add_sg[j] = act ? (frozen ? f_sg : sg[j]) : (real)0;
    add_pen[j] = act ? (frozen ? f_pen : pen[j]) : 0.0f;
    const real n2_now = frozen ? n2_end : n2[j];
    n2_end = act ? n2_now : n2_end;
    const bool hit = act && n2_now <= gt2;
    ev = hit ? 1u : ev;
    alive = alive && !hit;
  }
  ev = (ev == 0u && frozen) ? 2u : ev;
  // (every lane ORs its own word: no conflict, no return value)
  atomicOr(&evw[lane], ev << (2 * c));
  {
    char* fr = frec + ((size_t)c * 64 + lane) * L::kFrecBytes;
    *reinterpret_cast<real*>(fr) = f_sg;
    *reinterpret_cast<float*>(fr + 8) = f_pen;
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 5);
  lds_barrier();

  // ---------------------------------------------------------------- D': what earlier chunks decided
  {
    const uint32_t word = evw[lane];
    const uint32_t below = word & ((1u << (2 * c)) - 1u);
    const bool dead = below != 0u;
    const int first = dead ? (__builtin_ctz(below | 0x80000000u) >> 1) : c;
    const uint32_t kind = (word >> (2 * first)) & 3u;
    const char* fr = frec + ((size_t)first * 64 + lane) * L::kFrecBytes;
    const real o_sg = *reinterpret_cast<const real*>(fr);
    const float o_pen = *reinterpret_cast<const float*>(fr + 8);
    const bool carry = dead && kind == 2u;  // frozen in an earlier chunk: its addends, every step to the end
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const bool valid = t0 + j < T;
      add_sg[j] = dead ? ((carry && valid) ? o_sg : (real)0) : add_sg[j];
      add_pen[j] = dead ? ((carry && valid) ? o_pen : 0.0f) : add_pen[j];
    }
    // the chunk where the rollout ends (first event, or the end of the horizon) owns the terminal cost
    const bool last_chunk = t0 + CH >= T;
    if (!dead && (ev != 0u || last_chunk)) {
      double term = 0.0;
      if (ev != 1u) term = sqrt((double)n2_end) / Q.v_post_den;  // (1 - reached) * sqrt(d2) / (v_post + 1e-6)
      term_sh[lane] = term;
    }
    if (__any(!dead && bad) && lane == 0) {
      atomicOr(&flags[0], 1u);
      if (c == 0 || true) {  // (counted once per tile below, by wave 0)
      }
    }
    char* out = rec + ((size_t)c * 64 + lane) * L::kRecBytes;
    if constexpr (CHAIN64) {
      double2* o2 = reinterpret_cast<double2*>(out);
#pragma unroll
      for (int q = 0; q < CH / 2; ++q) o2[q] = make_double2(add_sg[2 * q], add_sg[2 * q + 1]);
    } else {
      float4* o4 = reinterpret_cast<float4*>(out);
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) o4[q] = make_float4(add_sg[4 * q], add_sg[4 * q + 1], add_sg[4 * q + 2], add_sg[4 * q + 3]);
    }
    float4* p4 = reinterpret_cast<float4*>(out + CH * sizeof(real));
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) p4[q] = make_float4(add_pen[4 * q], add_pen[4 * q + 1], add_pen[4 * q + 2], add_pen[4 * q + 3]);
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 6);
  lds_barrier();

  // ---------------------------------------------------------------- E: the accumulation, in order
  const bool failed = flags[0] != 0u;  // (uniform over the workgroup)
  float* nz = reinterpret_cast<float*>(rec);  // failed vote: the tile's noise [T][64] float2 where the records were
  if (failed) {
    lds_barrier();  // everybody has read the flag and is done with the records
    float2* nz2 = reinterpret_cast<float2*>(nz);
#pragma unroll
    for (int j = 0; j < CH; ++j)
      if (t0 + j < T) nz2[(size_t)(t0 + j) * 64 + lane] = e[j];
    lds_barrier();
  }
  if (c == 0) {
    float cost = 0.0f;
    if (!failed) {
      for (int i = 0; i < W; ++i) {
        const char* in = rec + ((size_t)i * 64 + lane) * L::kRecBytes;
        real a[CH];
        float pn[CH];
        if constexpr (CHAIN64) {
          const double2* i2 = reinterpret_cast<const double2*>(in);
#pragma unroll
          for (int q = 0; q < CH / 2; ++q) {
            const double2 t2 = i2[q];
            a[2 * q] = t2.x;
            a[2 * q + 1] = t2.y;
          }
        } else {
          const float4* i4 = reinterpret_cast<const float4*>(in);
#pragma unroll
          for (int q = 0; q < CH / 4; ++q) {
            const float4 t4 = i4[q];
            a[4 * q] = t4.x; a[4 * q + 1] = t4.y; a[4 * q + 2] = t4.z; a[4 * q + 3] = t4.w;
          }
        }
        const float4* p4 = reinterpret_cast<const float4*>(in + CH * sizeof(real));
#pragma unroll
        for (int q = 0; q < CH / 4; ++q) {
          const float4 t4 = p4[q];
          pn[4 * q] = t4.x; pn[4 * q + 1] = t4.y; pn[4 * q + 2] = t4.z; pn[4 * q + 3] = t4.w;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if constexpr (CHAIN64) cost = (float)((double)cost + a[j]);
          else cost = cost + a[j];
          cost = cost + pn[j];  // obstacle + unknown (one addition: exact unless both are set)
        }
      }
      cost = (float)((double)cost + term_sh[lane]);
    } else {
      // ---- sequential rollout of the tile with the tractions of the visited cells (mppi.py:966-1002),
      //      float32 state, the stage additions as above
      if (lane == 0 && Q.spec_failures) {
        atomicAdd_system(Q.spec_failures, 1u);
        __threadfence_system();
      }
      const float2* nz2 = reinterpret_cast<const float2*>(nz);
      float x = Q.x0, y = Q.y0, th = Q.th0;
      real d2 = (real)1e9;
      bool done = false, reached = false;
      for (int t = 0; t < T; ++t) {
        const uint32_t cl = scan_lookup<POW2RES>(Q, cells16, x, y);
        const float2 en = nz2[(size_t)t * 64 + lane], un = uq[t];
        const float vv = clip_f32(un.x + en.x, Q.v_lo, Q.v_hi), ww = clip_f32(un.y + en.y, Q.w_lo, Q.w_hi);
        const float vtr = (float)fma(Q.lin_ratio, (double)(int)(cl & 127u), Q.lin_lo);
        const float wtr = (float)fma(Q.ang_ratio, (double)(int)((cl >> 7) & 127u), Q.ang_lo);
        const float fr = (float)__builtin_amdgcn_fract((double)th * 0.15915494309189535);
        const float q = Q.dt * vv;
        const float xn = fmaf(vtr, q * __builtin_amdgcn_cosf(fr), x), yn = fmaf(vtr, q * __builtin_amdgcn_sinf(fr), y);
        const float thn = fmaf(wtr * Q.dt, ww, th);
        const real dx = (real)(Q.xg - xn), dy = (real)(Q.yg - yn);
        real n2s, sgs;
        if constexpr (CHAIN64) {
          n2s = fma(dx, dx, dy * dy);
          sgs = fma(Q.dist_weight, sqrt_newton_nz_f64(n2s), (double)Q.dt);
        } else {
          n2s = fmaf(dx, dx, dy * dy);
          sgs = fmaf((float)Q.dist_weight, __builtin_amdgcn_sqrtf(n2s), Q.dt);
        }
        float c1;
        if constexpr (CHAIN64) c1 = (float)((double)cost + sgs);
        else c1 = cost + sgs;
        c1 = c1 + __int_as_float(__float_as_int(Q.obs_cost) & __builtin_amdgcn_sbfe((int)cl, 14, 1));
        c1 = c1 + __int_as_float(__float_as_int(Q.unk_cost) & __builtin_amdgcn_sbfe((int)cl, 15, 1));
        const bool hit = n2s <= gt2;
        cost = done ? cost : c1;
        d2 = done ? d2 : n2s;
        x = done ? x : xn;
        y = done ? y : yn;
        th = done ? th : thn;
        reached = reached || (!done && hit);
        done = done || hit;
        if (__all(done)) break;
      }
      cost = (float)((double)cost + (reached ? 0.0 : 1.0) * sqrt((double)d2) / Q.v_post_den);
    }
    // the control cost of all T steps, also after an early goal break (mppi.py:1007-1009)
    for (int i = 0; i < W; ++i) {
      const float4* in = reinterpret_cast<const float4*>(ccr + ((size_t)i * 64 + lane) * CH);
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        const float4 t4 = in[q];
        cost = cost + t4.x;
        cost = cost + t4.y;
        cost = cost + t4.z;
        cost = cost + t4.w;
      }
    }
    MPPI_STAMP(stamp_wg, stamp_base + 7);
    if (live) costs[n] = cost;
    // first half of the control update (update_kernels.h): weights relative to the tile's minimum
    const float beta = wave_min_f32(live ? cost : __builtin_inff());
    const float wr = live ? (float)exp(-1.0 / (double)Q.lambda * (double)(cost - beta)) : 0.0f;
    if (live) w_rel[n] = wr;
    if (lane == 0) tile_beta[tile] = beta;
    if (pk.tnum) {
      term_sh[lane] = 0.0;  // (keeps the layout simple: the weights travel through evw's neighbour)
      reinterpret_cast<float*>(evw)[lane] = wr;
      const float den = wave_sum_to_lane63_f32(wr);
      if (lane == 63) pk.tden[tile] = den;
    }
  }
  if (!pk.tnum) return;
  lds_barrier();

  // ---------------------------------------------------------------- F: the tile's share of the update
  {
    const float wr = reinterpret_cast<const float*>(evw)[lane];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const float sx = wave_sum_to_lane63_f32(wr * e[j].x), sy = wave_sum_to_lane63_f32(wr * e[j].y);
      if (lane == 63 && t0 + j < T) pk.tnum[(size_t)(t0 + j) * pk.n_tiles + tile] = make_float2(sx, sy);
    }
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 8);
}

}  // namespace mppi
