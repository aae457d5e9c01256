// rollout_scan_kernel.h -- k_rollout_scan: the deterministic-dynamics rollout parallel over the
// HORIZON (gfx950, wave64).  MPPI_MATH_FAST only: tolerance mode, not bit-identical.
//
// Replaces rollout_det_dyn_numba (mppi.py:916-1009) and, with GEN, sample_noise_numba
// (mppi.py:1354-1370) and the pass of update_useq_numba over the noise (mppi.py:1177-1181).
// tests/scan_model.py states the algorithm in numpy, checked against the oracle on the CPU; this
// file follows it operation for operation.
//
// Why.  The reference's rollout is a chain of T dependent steps per control sample; with one tile
// of 64 rollouts per CU (N = 8192: the north-star shard) every design that walks that chain is
// bound by the issue rate of a handful of waves (the five-stage pipeline of rounds 2-5: 226 cycles per step, 18 us per
// launch, 11 % of the HBM roofline).  The chain exists because heading and position feed on the
// traction of the visited cell.  Under the assumption the speculative kernels already make --
// every visited cell carries the traction (vtr0, wtr0) of the start cell -- it is not a chain:
//     theta_t = theta_0 + wtr0*dt * sum_{k<t} w_k                      (mppi.py:990)
//     x_t     = x_0     + vtr0*dt * sum_{k<t} v_k cos(theta_k)         (mppi.py:988; y likewise)
// two prefix sums over the horizon.  A workgroup takes a tile of R rollouts (R = 32: 256 tiles for
// N = 8192, one per CU), a wave 8 consecutive steps of it: lane = (rollout r, half h), 8 / S steps
// per lane with S = 64 / R.  T = 100 is 13 waves per workgroup.
//
// Phases (workgroup barriers between them; S[] = per-(chunk, rollout) sums in LDS, float64):
//   A  noise of the lane's steps (GEN: Philox counter blocks computed here -- the noise never exists
//      in memory; else: read, tile-major) -> LDS (phase F and the fallback read it from there);
//      clipped controls, control-cost terms -> LDS, heading increments in turns, float32 prefix
//      inside the chunk, its total -> S
//   B  heading at the chunk start = theta_0 + sum of S over earlier chunks (float64, one v_fract);
//      hardware sin / cos per step, position increments, float32 prefix, totals -> S
//   C  position at the chunk start (float64 across chunks), positions, cell lookups (global
//      gathers: off every critical path here), squared goal distances, stage costs
//   D  per lane, with bit masks: the first step that meets a cell of ZERO linear traction (the
//      padding ring: the rollout is frozen there for the rest of the horizon, exactly as the
//      reference computes), the first goal hit before it (mppi.py:1000-1002), the vote on the
//      assumption over the steps that count; the chunk's event into a 2-bit field of one LDS word
//      per rollout (ds_or)
//   D' the first chunk with an event ends the rollout: later chunks add nothing; what every step
//      adds to the cost (stage cost, penalties) -> LDS; the frozen rollout's float64 stage cost,
//      penalty and number of remaining steps; the terminal cost
//   E  ONE wave walks the additions in the reference's order with the reference's float32 rounding
//      after every one (sums formed side by side cannot reproduce T sequential roundings: tree sums
//      leave 1 % of the costs beyond 1e-6 relative to the oracle, the walk 0.04 % and 84 %
//      bit-identical -- tests/test_scan_model.py); then the frozen steps in closed form per binade
//      (frozen_block), the terminal cost, the T control-cost additions (mppi.py:1005-1009); cost,
//      weights relative to the tile's minimum (update_kernels.h)
//   F  two waves, lane = step: sum over the tile's rollouts of w_rel * noise(t, n) from LDS -- the
//      tile's contribution to the update (tile packets: update_kernels.h): no pass over the noise
//      is left.
// A failed vote (a rollout still moving meets a cell whose traction differs): phase E is replaced
// by a sequential float32 rollout of the tile by one wave.  The host stops launching this kernel on
// maps where most tiles fail (review_speculation).
#pragma once
#include <type_traits>
#include "rollout_kernels.h"

namespace mppi {

// tile contributions to the control update, written by the weight epilogue and phase F; read by
// k_combine_tiles or by the class reduction of the next rollout launch (update_kernels.h)
struct ScanPackets {
  float* tiles;  // [n_tiles][tile_packet_floats(T)]: {minimum cost, sum of w_rel, sum of w_rel * noise(t) for every t}
  int n_tiles;
};

template <int R>
struct ScanLds {
  static constexpr int S = 64 / R;     // lanes per rollout in a wave
  static constexpr int CHL = 8 / S;    // steps per lane
  // per workgroup of W waves (K = W * S chunks of CHL steps, Tp = 8 W steps)
  __host__ __device__ static constexpr size_t rec(int W) { return (size_t)W * 64 * CHL * 8; }   // {sg, pen}[CHL] per (chunk, rollout)
  __host__ __device__ static constexpr size_t ccr(int W) { return (size_t)W * 64 * CHL * 4; }   // cc[CHL]
  __host__ __device__ static constexpr size_t e2(int W) { return (size_t)W * 8 * R * 8; }       // noise [Tp][R] float2
  // + (a sharded iteration's update applied in this launch: update_kernels.h, PendingApply) the ranks'
  //   scales and the updated controls [Tp] float2
  __host__ __device__ static constexpr size_t small(int W) {
    return (size_t)R * (16 + 8 + 8 + 4) + 64 + 8 * (kMaxFoldedRanks + 2) + (size_t)W * 64;
  }
  __host__ __device__ static constexpr size_t total(int W) { return rec(W) + ccr(W) + e2(W) + small(W); }
  // the three float64 sums of phases A-C, [3][K][R], live where the records of phase D' go
  static_assert(3 * 8 <= CHL * 8, "sums alias the records");
};

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

// (WIDE: the 32-bit cells of the speed-map mode -- the same 16 bits + the risk traction byte in bits 16..23 -- behind
//  the same pointer, same pitch in cells: k_pack_cells32_risk)
template <bool POW2RES, bool WIDE = false>
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
  if (WIDE) return reinterpret_cast<const uint32_t*>(cells16)[__mul24(yi, Q.pitch16) + xi];
  return cells16[__mul24(yi, Q.pitch16) + xi];
}

// `count` further steps of a rollout that stands still: each adds the same float64 stage cost k
// (and penalty pen) to the float32 cost, rounded after every addition (mppi.py:994-998).  While the
// cost stays inside one binade the additions are exact multiples of its ulp, acc + m * round(k to
// the ulp), so whole runs are taken at once; the steps that cross into the next binade are added
// one by one.  tests/test_scan_model.py::test_frozen_block_equals_the_additions_one_by_one.
__device__ __forceinline__ float frozen_block(float acc, double k, float pen, int count) {
  if (pen != 0.0f) {  // (a zero-traction cell that is also an obstacle: step by step)
    for (; count > 0; --count) acc = (float)((double)acc + k) + pen;
    return acc;
  }
  while (count > 0) {
    acc = (float)((double)acc + k);
    --count;
    if (count == 0 || !(acc > 0.0f)) continue;
    const int e = (int)((__float_as_uint(acc) >> 23) & 0xffu) - 127;  // acc in [2^e, 2^(e+1))
    // 2^(e-23), 2^(23-e), 2^(e+1) straight from their bit patterns
    const double ulp = __longlong_as_double((long long)(1023 + e - 23) << 52);
    const double inv_ulp = __longlong_as_double((long long)(1023 + 23 - e) << 52);
    const double top = __longlong_as_double((long long)(1023 + e + 1) << 52);
    const double q = rint(k * inv_ulp) * ulp;
    if (!(q > 0.0)) continue;
    // steps that surely stay below the top of the binade: a float32 estimate of room / q, one short
    // (an underestimate only costs another trip round this loop)
    int m = (int)((float)(top - (double)acc) * __builtin_amdgcn_rcpf((float)q) * 0.999999f) - 1;
    m = min(m, count);
    if (m > 0 && fma((double)m, q, (double)acc) < top) {
      acc = (float)fma((double)m, q, (double)acc);
      count -= m;
    }
  }
  return acc;
}

// GEN: `gen` describes the Philox counters of THIS iteration's noise (out is ignored);
// !GEN: `noise` holds it (tile-major) and the spare workgroups (blockIdx >= n_rollout_blocks)
//       write the next iteration's (`next_noise`), as in k_rollout_pipe.
template <int R, bool POW2RES, bool GEN>
__global__ __launch_bounds__(1024) void k_rollout_scan(DevParams P, const uint16_t* __restrict__ cells16,
                                                       const float2* __restrict__ noise, NoiseJob gen,
                                                       const float2* __restrict__ u, float* __restrict__ costs,
                                                       float* __restrict__ w_rel, ScanPackets pk,
                                                       int n_rollout_blocks, NoiseJob next_noise,
                                                       PendingApply pend) {
  extern __shared__ double2 scan_lds[];
  if ((int)blockIdx.x >= n_rollout_blocks) {
    if (next_noise.out)
      noise_generate<true>(next_noise, (blockIdx.x - n_rollout_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6),
                           (gridDim.x - n_rollout_blocks) * (blockDim.x >> 6));
    return;
  }
  using L = ScanLds<R>;
  constexpr int S = L::S, CHL = L::CHL;
  const int c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave = 8 steps
  const int lane = threadIdx.x & 63;
  const int r = lane & (R - 1), h = lane / R;  // rollout of the tile, which CHL steps of the wave's 8
  const int W = (int)(blockDim.x >> 6);
  const int K = W * S;
  const int k = c * S + h;  // chunk
  [[maybe_unused]] const bool stamp_wg = blockIdx.x == 5;
  [[maybe_unused]] const int stamp_base = 64 + 16 * c;
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 0);
  MPPI_STAMP(threadIdx.x == 0 && blockIdx.x < 512, 2048 + 2 * blockIdx.x);  // every workgroup: entry ...
  DevParams Q = P;
  const int tile = blockIdx.x;
  const float2* uq = select_instance(Q, u, Q.inst ? (tile * R) / Q.n_inst : 0);
  const int T = Q.n_steps, N = Q.n_local;
  const int n = tile * R + r;
  const bool live = n < N;
  const int t0 = k * CHL;
  const int nvalid = min(max(T - t0, 0), CHL);  // steps of this lane inside the horizon

  char* base = reinterpret_cast<char*>(scan_lds);
  float4* rec = reinterpret_cast<float4*>(base);                       // [K][R] {sg[CHL], pen[CHL]}
  double* sumS = reinterpret_cast<double*>(base);                      // [3][K][R] (phases A-C)
  float* ccr = reinterpret_cast<float*>(base + L::rec(W));             // [K][R][CHL]
  float2* e2 = reinterpret_cast<float2*>(base + L::rec(W) + L::ccr(W));  // [8 W][R], column n of row t at n ^ (t & (R-1))
  char* small = base + L::rec(W) + L::ccr(W) + L::e2(W);
  double* fz_k = reinterpret_cast<double*>(small);                     // [R] frozen: stage cost per step
  double* term_sh = fz_k + R;                                          // [R] terminal cost
  uint32_t* evw = reinterpret_cast<uint32_t*>(term_sh + R);            // [R][2] 2 bits per chunk
  float* fz_pen = reinterpret_cast<float*>(evw + 2 * R);               // [R]
  int* fz_count = reinterpret_cast<int*>(fz_pen + R);                  // [R]
  float* wsh = reinterpret_cast<float*>(fz_count + R);                 // [R] weights (phase F)
  uint32_t* flags = reinterpret_cast<uint32_t*>(wsh + R);              // [0] failed vote
  double* scale_sh = reinterpret_cast<double*>(small + (size_t)R * 36 + 64);  // [kMaxFoldedRanks + 2]
  float2* u_sh = reinterpret_cast<float2*>(scale_sh + kMaxFoldedRanks + 2);   // [Tp]
  if (c == 0 && lane < R) {
    evw[2 * lane] = 0u;
    evw[2 * lane + 1] = 0u;
    fz_count[lane] = 0;
    if (lane == 0) flags[0] = 0u;
  }
  // a sharded iteration's update applied here (update_kernels.h, PendingApply) ...
  const bool folded = pend.packets != nullptr;
  // ... or, on one GPU, the previous launch's tile packets combined here (PendingApply::reduce_tiles, as in
  // k_rollout_scan_exact): waves 0 and 1 of workgroup `tile` take steps tile, tile + gridDim.x (and on round), with
  // the very function block t of k_combine_tiles runs; wave 0 then collects the published sequence.  The packets are
  // requested before anything else and combined behind the Philox blocks: a trip of ~1.5 us either way.
  const bool reducing = folded && pend.reduce_tiles != nullptr;
  const int n_red = min(2, W);  // (a horizon of <= 8 steps is one wave)
  const int red_t = tile + c * n_rollout_blocks;
  const bool reduce_here = reducing && c < n_red && red_t < T;
  StepLoads red_loads;
  float2 red_u = make_float2(0.0f, 0.0f);
  if (reduce_here) {
    red_loads = combine_step_issue(pend.reduce_tiles, pend.reduce_n_tiles, tile_packet_floats(T), red_t, lane);
    red_u = uq[red_t];
  }
  if (folded && !reducing) {
    if (c == 0) pending_apply_prepare(pend, lane, scale_sh);
    lds_barrier();
  }

  // ---------------------------------------------------------------- A: noise, controls, heading increments
  // (requested before the Philox blocks: their first touch is a trip to memory)
  float2 ut[CHL];
  if (reducing) {
    // (from LDS behind the barrier below)
  } else if (folded) {  // (wave-uniform) the 8 controls of this wave's steps from the ranks' packets
    if (lane < 8) {
      const int t = 8 * c + lane;
      const float2 v = t < T ? pending_apply_control(pend, scale_sh, uq, t) : make_float2(0.0f, 0.0f);
      u_sh[t] = v;
      if (tile == 0 && t < T) {
        pend.u_out[t] = v;
        pend.u_prev[t] = v;
      }
    }
    if (tile == 0 && c == 0 && lane == 0) {
      pend.stats[0] = scale_sh[kMaxFoldedRanks + 1];
      pend.stats[1] = scale_sh[kMaxFoldedRanks];
    }
#pragma unroll
    for (int j = 0; j < CHL; ++j) ut[j] = u_sh[t0 + j];
  } else {
#pragma unroll
    for (int j = 0; j < CHL; ++j) ut[j] = uq[min(t0 + j, T - 1)];
  }
  // the assumption: every visited cell carries the traction bytes of the start cell
  const uint32_t ref = scan_lookup<POW2RES>(Q, cells16, Q.x0, Q.y0) & 0x3fffu;
  float2 e[CHL];
  if constexpr (GEN) {
    const uint64_t epoch = gen.epoch + (gen.gen_counter ? *gen.gen_counter : 0ull);
    const unsigned int pairs = (unsigned int)(T + 1) / 2u;
    const unsigned int n_global = (unsigned int)(gen.n_offset + min(n, N - 1));
#pragma unroll
    for (int jp = 0; jp < CHL / 2; ++jp) {
      const unsigned int tp = (unsigned int)(t0 / 2 + jp);
      scan_noise_pair(gen, epoch, n_global, pairs, min(tp, pairs - 1u), e[2 * jp], e[2 * jp + 1]);
    }
  } else {
    const float2* col = noise + (size_t)(n >> 6) * T * 64 + (n & 63);
#pragma unroll
    for (int j = 0; j < CHL; ++j) e[j] = col[(size_t)min(t0 + j, T - 1) * 64];
  }
#pragma unroll
  for (int j = 0; j < CHL; ++j) {
    e[j] = j < nvalid ? e[j] : make_float2(0.0f, 0.0f);
    const int t = t0 + j;
    e2[t * R + (r ^ (t & (R - 1)))] = e[j];
  }
  if (reducing) {  // (workgroup-uniform)
    if (reduce_here) {
      const int stride = tile_packet_floats(T);
      publish_step(pend, combine_step_finish(red_loads, pend.reduce_tiles, pend.reduce_n_tiles, stride, red_t, pend.lambda, lane),
                   red_u, red_t, T, lane);
      for (int tt = red_t + n_red * n_rollout_blocks; tt < T; tt += n_red * n_rollout_blocks)
        publish_step(pend, combine_step(pend.reduce_tiles, pend.reduce_n_tiles, stride, tt, pend.lambda, lane), uq[tt], tt, T, lane);
    }
    if (c == 0) collect_published(pend, T, 8 * W, lane, u_sh);
    lds_barrier();
#pragma unroll
    for (int j = 0; j < CHL; ++j) ut[j] = u_sh[t0 + j];
  }
#pragma unroll
  for (int j = 0; j < CHL; ++j) ut[j] = j < nvalid ? ut[j] : make_float2(0.0f, 0.0f);
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 1);
  const double vtr0 = fma(Q.lin_ratio, (double)(int)(ref & 127u), Q.lin_lo);
  const double wtr0 = fma(Q.ang_ratio, (double)(int)((ref >> 7) & 127u), Q.ang_lo);
  const float kv = (float)vtr0 * Q.dt;                                         // position increment per unit speed
  const float kturn = (float)(wtr0 * (double)Q.dt * 0.15915494309189535);      // heading increment in turns per unit w
  // lambda * (u0/s0^2 * e0 + u1/s1^2 * e1)   (mppi.py:1007-1009)
  const float k0 = Q.cc_k0, k1 = Q.cc_k1;
  float qv[CHL];   // kv * clipped speed
  float lt[CHL];   // heading, in turns, gained inside the chunk BEFORE step j
  {
    float cc[CHL];
    float ts = 0.0f;
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      const float v = __builtin_amdgcn_fmed3f(ut[j].x + e[j].x, Q.v_lo, Q.v_hi);
      const float w = __builtin_amdgcn_fmed3f(ut[j].y + e[j].y, Q.w_lo, Q.w_hi);
      cc[j] = fmaf(k0 * ut[j].x, e[j].x, (k1 * ut[j].y) * e[j].y);
      qv[j] = kv * v;
      lt[j] = ts;
      ts += kturn * w;
    }
    float4* dst = reinterpret_cast<float4*>(ccr + ((size_t)k * R + r) * CHL);
#pragma unroll
    for (int q = 0; q < CHL / 4; ++q) dst[q] = make_float4(cc[4 * q], cc[4 * q + 1], cc[4 * q + 2], cc[4 * q + 3]);
    sumS[(size_t)k * R + r] = (double)ts;
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 2);
  lds_barrier();

  // ---------------------------------------------------------------- B: headings, position increments
  float lx[CHL], ly[CHL];  // position gained inside the chunk up to and including step j
  {
    double tb = (double)Q.th0 * 0.15915494309189535;
    for (int i = 0; i < c * S; ++i) tb += sumS[(size_t)i * R + r];  // (uniform trip count)
    if (S > 1 && h > 0) tb += sumS[(size_t)(c * S) * R + r];
    const float fb = (float)__builtin_amdgcn_fract(tb);  // v_sin_f32 / v_cos_f32 take turns
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      const float fr = fb + lt[j];
      sx = fmaf(qv[j], __builtin_amdgcn_cosf(fr), sx);
      sy = fmaf(qv[j], __builtin_amdgcn_sinf(fr), sy);
      lx[j] = sx;
      ly[j] = sy;
    }
    sumS[(size_t)(K + k) * R + r] = (double)sx;
    sumS[(size_t)(2 * K + k) * R + r] = (double)sy;
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 3);
  lds_barrier();

  // ---------------------------------------------------------------- C: positions, lookups, stage costs
  float sg[CHL];   // stage cost of step j: dt + dist_weight * distance after the step
  float n2[CHL];   // squared goal distance after step j
  float pen[CHL];  // obstacle / unknown penalties of the cell step j starts in
  float xa[CHL + 1], ya[CHL + 1];
  uint32_t zero_bits = 0, mism_bits = 0, hit_bits = 0;
  {
    double bx = (double)Q.x0, by = (double)Q.y0;
    for (int i = 0; i < c * S; ++i) {
      bx += sumS[(size_t)(K + i) * R + r];
      by += sumS[(size_t)(2 * K + i) * R + r];
    }
    if (S > 1 && h > 0) {
      bx += sumS[(size_t)(K + c * S) * R + r];
      by += sumS[(size_t)(2 * K + c * S) * R + r];
    }
    xa[0] = (float)bx;
    ya[0] = (float)by;
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      xa[j + 1] = xa[0] + lx[j];
      ya[j + 1] = ya[0] + ly[j];
    }
    uint32_t cell[CHL];
#pragma unroll
    for (int j = 0; j < CHL; ++j) cell[j] = scan_lookup<POW2RES>(Q, cells16, xa[j], ya[j]);  // the cell step j STARTS in
    const float dwf = (float)Q.dist_weight;
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      const float dx = Q.xg - xa[j + 1], dy = Q.yg - ya[j + 1];
      n2[j] = fmaf(dx, dx, dy * dy);
      sg[j] = fmaf(dwf, __builtin_amdgcn_sqrtf(n2[j]), Q.dt);
      hit_bits |= (n2[j] <= Q.gt2 ? 1u : 0u) << j;
    }
    pin_memory_order();
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      const uint32_t cl = cell[j];
      zero_bits |= ((int)(cl & 127u) == Q.lin_zero_byte ? 1u : 0u) << j;
      mism_bits |= (((cl ^ ref) & 0x3fffu) != 0u ? 1u : 0u) << j;
      pen[j] = __int_as_float(__float_as_int(Q.obs_cost) & __builtin_amdgcn_sbfe((int)cl, 14, 1)) +
               __int_as_float(__float_as_int(Q.unk_cost) & __builtin_amdgcn_sbfe((int)cl, 15, 1));
    }
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 4);

  // ---------------------------------------------------------------- D: the chunk's event
  const uint32_t vmask = (1u << nvalid) - 1u;
  const int s = __builtin_ctz((zero_bits & vmask) | (1u << CHL));                      // first step that freezes
  const int hh = __builtin_ctz((hit_bits & vmask & ((1u << s) - 1u)) | (1u << CHL));   // first goal hit before it
  const bool is_hit = hh < CHL, froze = !is_hit && s < nvalid;
  int n_act = is_hit ? hh + 1 : min(s, nvalid);  // steps of this chunk that add their own cost
  const bool bad = (mism_bits & ((1u << n_act) - 1u)) != 0u;
  const uint32_t ev = is_hit ? 1u : (froze ? 2u : 0u);  // 0 none, 1 goal reached, 2 frozen
  double f_k = 0.0, f_d2 = 1e9;
  float f_pen = 0.0f;
  bool f_hit = false;
  if (__any(froze)) {
    // a rollout frozen at step s stands at the pre-step position of s for the rest of the horizon:
    // what it pays per step, in float64 as the reference's CPU path computes it
    float fx = xa[0], fy = ya[0];
    f_pen = pen[0];
#pragma unroll
    for (int j = 1; j < CHL; ++j) {
      fx = s == j ? xa[j] : fx;
      fy = s == j ? ya[j] : fy;
      f_pen = s == j ? pen[j] : f_pen;
    }
    const double dx = (double)(Q.xg - fx), dy = (double)(Q.yg - fy);
    f_d2 = fma(dx, dx, dy * dy);
    f_k = fma(Q.dist_weight, sqrt_newton_nz_f64(f_d2), (double)Q.dt);
    f_hit = f_d2 <= (double)Q.gt2;
  }
  // (every rollout has its own pair of words; lanes of one rollout set different bits: no return value needed)
  if (ev != 0u) atomicOr(&evw[2 * r + ((2 * k) >> 5)], ev << ((2 * k) & 31));
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 5);
  lds_barrier();

  // ---------------------------------------------------------------- D': what earlier chunks decided
  {
    const uint32_t w0 = evw[2 * r], w1 = evw[2 * r + 1];
    const uint64_t word = ((uint64_t)w1 << 32) | w0;
    const bool dead = (word & ((1ull << (2 * k)) - 1ull)) != 0ull;  // an earlier chunk ended the rollout
    n_act = dead ? 0 : n_act;
    const bool owner = !dead && ev != 0u;                       // this chunk ends it
    const bool last = t0 < T && t0 + CHL >= T;                  // ... or it is the chunk the horizon ends in
    if (owner || (!dead && last)) {
      double term = 0.0;  // (1 - reached) * sqrt(d2) / (v_post + 1e-6)   (mppi.py:26-28, 1005)
      if (ev == 2u) {
        term = f_hit ? 0.0 : sqrt_newton_f64(f_d2) * Q.inv_v_post_den;
      } else if (ev == 0u) {
        float n2l = n2[0];
#pragma unroll
        for (int j = 1; j < CHL; ++j) n2l = (nvalid - 1 == j) ? n2[j] : n2l;
        term = sqrt_newton_f64((double)n2l) * Q.inv_v_post_den;
      }
      term_sh[r] = term;
      if (ev == 2u) {
        fz_k[r] = f_k;
        fz_pen[r] = f_pen;
        fz_count[r] = f_hit ? 1 : T - (t0 + s);
      }
    }
    if (__any(!dead && bad) && lane == 0) atomicOr(&flags[0], 1u);
    float4* out = rec + ((size_t)k * R + r) * (CHL / 2);
#pragma unroll
    for (int q = 0; q < CHL / 4; ++q)
      out[q] = make_float4(4 * q < n_act ? sg[4 * q] : 0.0f, 4 * q + 1 < n_act ? sg[4 * q + 1] : 0.0f,
                           4 * q + 2 < n_act ? sg[4 * q + 2] : 0.0f, 4 * q + 3 < n_act ? sg[4 * q + 3] : 0.0f);
#pragma unroll
    for (int q = 0; q < CHL / 4; ++q)
      out[CHL / 4 + q] = make_float4(4 * q < n_act ? pen[4 * q] : 0.0f, 4 * q + 1 < n_act ? pen[4 * q + 1] : 0.0f,
                                     4 * q + 2 < n_act ? pen[4 * q + 2] : 0.0f, 4 * q + 3 < n_act ? pen[4 * q + 3] : 0.0f);
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 6);
  lds_barrier();

  // ---------------------------------------------------------------- E: the accumulation, in order
  if (c == 0) {
    __builtin_amdgcn_s_setprio(3);  // (the critical path of the launch from here on)
    // (R = 32: lanes 32..63 mirror lanes 0..31)
    const bool failed = flags[0] != 0u;
    float cost = 0.0f;
    if (!failed) {
      // records [k][r] of CHL stage costs + CHL penalties; the next group of them is requested while
      // this one is added.  A lone wave issues about one instruction per 5 cycles, dependent or not:
      // what counts is the instruction count -- one pointer bump per group, reads at immediate
      // offsets (past the last record they fall into the arrays that follow: read, never added),
      // two register sets instead of copies
      constexpr int V = CHL / 2;   // float4 per record
      constexpr int G = 16 / CHL;  // records per group: 16 steps
      float4 ga[G * V], gb[G * V];
      const float4* at = rec + (size_t)r * V;
      auto load = [&](float4 (&dst)[G * V]) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int q = 0; q < V; ++q) dst[g * V + q] = at[(size_t)g * R * V + q];
        at += (size_t)G * R * V;
      };
      auto add_record = [&](const float4 (&src)[G * V], int g) {
#pragma unroll
        for (int q = 0; q < CHL / 4; ++q) {
          const float4 a = src[g * V + q], p4 = src[g * V + CHL / 4 + q];
          cost = (cost + a.x) + p4.x;  // stage cost, then obstacle + unknown (one addition: exact unless both are set)
          cost = (cost + a.y) + p4.y;
          cost = (cost + a.z) + p4.z;
          cost = (cost + a.w) + p4.w;
        }
      };
      load(ga);
      int i = 0;
      for (; i + 2 * G <= K; i += 2 * G) {
        load(gb);
#pragma unroll
        for (int g = 0; g < G; ++g) add_record(ga, g);
        load(ga);
#pragma unroll
        for (int g = 0; g < G; ++g) add_record(gb, g);
      }
      if (i + G <= K) {
        load(gb);
#pragma unroll
        for (int g = 0; g < G; ++g) add_record(ga, g);
        i += G;
#pragma unroll
        for (int g = 0; g < G - 1; ++g)
          if (i + g < K) add_record(gb, g);
      } else {
#pragma unroll
        for (int g = 0; g < G - 1; ++g)
          if (i + g < K) add_record(ga, g);
      }
      MPPI_STAMP(stamp_wg, stamp_base + 9);
      const int cnt = fz_count[r];
      if (__any(cnt > 0)) cost = frozen_block(cost, fz_k[r], fz_pen[r], cnt);
      MPPI_STAMP(stamp_wg, stamp_base + 10);
      cost = (float)((double)cost + term_sh[r]);
    } else {
      // ---- sequential rollout of the tile with the tractions of the visited cells (mppi.py:966-1002)
      if (lane == 0 && Q.spec_failures) {
        atomicAdd_system(Q.spec_failures, 1u);
        __threadfence_system();
      }
      const float dwf = (float)Q.dist_weight;
      float x = Q.x0, y = Q.y0, th = Q.th0;
      float d2 = 1e9f;
      bool done = false, reached = false;
      for (int t = 0; t < T; ++t) {
        const uint32_t cl = scan_lookup<POW2RES>(Q, cells16, x, y);
        const float2 en = e2[t * R + (r ^ (t & (R - 1)))], un = folded ? u_sh[t] : uq[t];
        const float vv = __builtin_amdgcn_fmed3f(un.x + en.x, Q.v_lo, Q.v_hi);
        const float ww = __builtin_amdgcn_fmed3f(un.y + en.y, Q.w_lo, Q.w_hi);
        const float vtr = (float)fma(Q.lin_ratio, (double)(int)(cl & 127u), Q.lin_lo);
        const float wtr = (float)fma(Q.ang_ratio, (double)(int)((cl >> 7) & 127u), Q.ang_lo);
        const float fr = (float)__builtin_amdgcn_fract((double)th * 0.15915494309189535);
        const float q = Q.dt * vv;
        const float xn = fmaf(vtr, q * __builtin_amdgcn_cosf(fr), x), yn = fmaf(vtr, q * __builtin_amdgcn_sinf(fr), y);
        const float thn = fmaf(wtr * Q.dt, ww, th);
        const float dx = Q.xg - xn, dy = Q.yg - yn;
        const float n2s = fmaf(dx, dx, dy * dy);
        float c1 = cost + fmaf(dwf, __builtin_amdgcn_sqrtf(n2s), Q.dt);
        c1 = c1 + __int_as_float(__float_as_int(Q.obs_cost) & __builtin_amdgcn_sbfe((int)cl, 14, 1));
        c1 = c1 + __int_as_float(__float_as_int(Q.unk_cost) & __builtin_amdgcn_sbfe((int)cl, 15, 1));
        const bool hit = n2s <= Q.gt2;
        cost = done ? cost : c1;
        d2 = done ? d2 : n2s;
        x = done ? x : xn;
        y = done ? y : yn;
        th = done ? th : thn;
        reached = reached || (!done && hit);
        done = done || hit;
        if (__all(done)) break;
      }
      cost = (float)((double)cost + (reached ? 0.0 : 1.0) * sqrt_newton_f64((double)d2) * Q.inv_v_post_den);
    }
    // the control cost of all T steps, also after an early goal break (mppi.py:1007-1009)
    {
      constexpr int V = CHL / 4;   // float4 per record
      constexpr int G = 32 / CHL;  // records per group: 32 steps
      float4 ga[G * V], gb[G * V];
      const float4* at = reinterpret_cast<const float4*>(ccr) + (size_t)r * V;
      auto load = [&](float4 (&dst)[G * V]) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int q = 0; q < V; ++q) dst[g * V + q] = at[(size_t)g * R * V + q];
        at += (size_t)G * R * V;
      };
      auto add_record = [&](const float4 (&src)[G * V], int g) {
#pragma unroll
        for (int q = 0; q < V; ++q) {
          cost = cost + src[g * V + q].x;
          cost = cost + src[g * V + q].y;
          cost = cost + src[g * V + q].z;
          cost = cost + src[g * V + q].w;
        }
      };
      load(ga);
      int i = 0;
      for (; i + 2 * G <= K; i += 2 * G) {
        load(gb);
#pragma unroll
        for (int g = 0; g < G; ++g) add_record(ga, g);
        load(ga);
#pragma unroll
        for (int g = 0; g < G; ++g) add_record(gb, g);
      }
      if (i + G <= K) {
        load(gb);
#pragma unroll
        for (int g = 0; g < G; ++g) add_record(ga, g);
        i += G;
#pragma unroll
        for (int g = 0; g < G - 1; ++g)
          if (i + g < K) add_record(gb, g);
      } else {
#pragma unroll
        for (int g = 0; g < G - 1; ++g)
          if (i + g < K) add_record(ga, g);
      }
    }
    MPPI_STAMP(stamp_wg, stamp_base + 7);
    const bool mine = live && lane < R;
    if (mine) costs[n] = cost;
    // first half of the control update (update_kernels.h): weights relative to the tile's minimum
    const float beta = wave_min_f32(live ? cost : __builtin_inff());
    // exp(-(c - beta)/lambda) = 2^(n + f): the fraction through v_exp_f32 (relative error ~1e-7 whatever
    // the argument), the integer through the exponent
    float wr = 0.0f;
    if (mine) {
      const double a2 = (double)(cost - beta) * Q.neg_log2e_over_lambda;  // <= 0
      const double nf = floor(a2);
      wr = ldexpf(__builtin_amdgcn_exp2f((float)(a2 - nf)), (int)fmax(nf, -200.0));
    }
    if (mine) w_rel[n] = wr;
    if (lane < R) wsh[lane] = wr;
    const float den = wave_sum_to_lane63_f32(wr);
    if (lane == 63) {
      *reinterpret_cast<float2*>(pk.tiles + (size_t)tile * tile_packet_floats(T)) = make_float2(beta, den);
    }
    MPPI_STAMP(stamp_wg, stamp_base + 11);
  }
  lds_barrier();

  // ---------------------------------------------------------------- F: the tile's share of the update
  if (c < 2) {
    const int t = 64 * c + lane;
    if (t < T) {
      const float2* row = e2 + (size_t)t * R;
      const int sw = t & (R - 1);
      float ax = 0.0f, ay = 0.0f;
#pragma unroll 8
      for (int m = 0; m < R; ++m) {
        const float wm = wsh[m];
        const float2 en = row[m ^ sw];
        ax = fmaf(wm, en.x, ax);
        ay = fmaf(wm, en.y, ay);
      }
      *reinterpret_cast<float2*>(pk.tiles + (size_t)tile * tile_packet_floats(T) + 2 + 2 * t) = make_float2(ax, ay);
    }
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 8);
  MPPI_STAMP(threadIdx.x == 0 && blockIdx.x < 512, 2049 + 2 * blockIdx.x);  // ... and wave 0's exit
}

}  // namespace mppi
