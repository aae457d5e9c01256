// rollout_scan_exact_kernel.h -- k_rollout_scan_exact: the time-parallel rollout with the
// reference's rounding points (MPPI_MATH_EXACT): bits identical to the oracle and to the pipelined
// kernels (gfx950, wave64).
//
// Replaces rollout_det_dyn_numba (mppi.py:916-1009), sample_noise_numba (mppi.py:1354-1370, GEN)
// and the pass of update_useq_numba over the noise (mppi.py:1177-1181).
//
// k_rollout_scan (rollout_scan_kernel.h) turns the horizon into prefix sums, which reassociates the
// reference's float32 roundings of heading and position: tolerance mode.  What the reference
// actually prescribes is much less than a chain of whole steps, though.  Under the assumption that
// every visited cell carries the traction of the start cell (the vote of the speculative kernels),
// the ONLY quantities that depend on their own previous value are three running sums, each rounded
// to float32 after every addition (float64 fma, float32 store -- the CPU path of the reference):
//     theta <- float32(fma(wtr0, dt*w_t, theta))                                 (mppi.py:990)
//     x     <- float32(fma(vtr0, (dt*v_t)*cos(theta_t), x))        (y likewise)  (mppi.py:988-989)
//     cost  <- float32(cost + stage_t), + obstacle, + unknown                    (mppi.py:994-998)
// Everything else -- the Philox blocks, clipping, the products dt*v, dt*w, sin / cos of every
// (rounded) heading, the cell lookups, distances, square roots, stage costs, the vote, goal and
// freeze events -- depends only on those sums' VALUES and is computed for all steps side by side:
// lane = (rollout, 4 consecutive steps), 13 waves per tile of 32 rollouts at T = 100.  The three
// sums are WALKED, one after the other, by one wave each (x and y by two waves at once): three
// dependent instructions per step (v_fma_f64, v_cvt_f32_f64, v_cvt_f64_f32) instead of the ~53 of
// the pipelined kernels' state wave.  Same operations on the same operands in the same order as the
// oracle: the same bits, by construction.
//
// Phases (workgroup barriers between them):
//   A   noise (GEN: Philox blocks; else read) -> LDS; clipped controls; dt*w -> LDS; the float64
//       control ratios u/std^2 of the wave's 8 steps -> LDS
//   W1  wave 0 walks theta over the horizon -> LDS (float32 heading BEFORE every step);
//       the other waves meanwhile: control-cost terms (float64, mppi.py:1007-1009) -> LDS
//   B   sin / cos of every heading (sincos_f64, as the other exact kernels), (dt*v)*cos, (dt*v)*sin -> LDS
//   W2  wave 0 walks x, wave 1 walks y -> LDS (float32 positions, T + 1 of them)
//   C   lookups (exact floor division), squared goal distances, square roots, stage costs (float64);
//       per lane the first freeze / goal hit among its steps, the vote; events -> LDS word (ds_or)
//   D'  first event wins; per-step addends -> LDS; frozen addend, terminal cost
//   E   wave 0 walks the cost: stage, obstacle, unknown per step (mppi.py:994-998), the frozen
//       steps, the terminal cost, the T control-cost terms (mppi.py:1005-1009); cost, tile weights
//   F   per-tile update sums, lane = step (as k_rollout_scan)
// A failed vote: the tile is rolled out sequentially by one wave with the arithmetic of
// k_rollout_map<DET, exact> (unicycle_step / add_stage_cost) -- the same bits again, slowly; the
// host stops launching this kernel on a map where most tiles fail (review_speculation).
#pragma once
#include "rollout_scan_kernel.h"

namespace mppi {

// LDS of one workgroup of W waves over R = 32 rollouts, Tp = 8 W steps, per (step, rollout):
//   e2   float2        noise (swizzled columns: phase F reads it lane = step)
//   ccr  double        control-cost term
//   p0   double        dt*w (A -> W1), then (dt*v)*cos (B -> W2);      later the records' first half
//   p1   double        (dt*v)*sin (B -> W2);                           later the records' second half
//   p2   float / float2  heading before the step (W1 -> B), then position (W2 -> C), Tp + 1 rows
// records (D' -> E), 16 bytes per step over p0 | p1: {double stage addend, float obstacle, float unknown}
struct ScanExactLds {
  static constexpr int R = 32, CHL = 4;
  __host__ __device__ static constexpr size_t plane(int W) { return (size_t)W * 8 * R * 8; }
  __host__ __device__ static constexpr size_t e2(int W) { return plane(W); }
  __host__ __device__ static constexpr size_t ccr(int W) { return plane(W); }
  __host__ __device__ static constexpr size_t p2(int W) { return (size_t)(W * 8 + 1) * R * 8; }
  __host__ __device__ static constexpr size_t small(int W) { return (size_t)W * 8 * (16 + 8) + R * 64 + 64 + 8 * (kMaxFoldedRanks + 2); }
  __host__ __device__ static constexpr size_t total(int W) { return e2(W) + ccr(W) + 2 * plane(W) + p2(W) + small(W); }
};

// the frozen steps of an exact walk: the closed form of frozen_block when there is no penalty
// (a tie of the float64 addend exactly between two float32 neighbours would need the running sum's
// parity: measure zero), step by step with both additions otherwise
__device__ __forceinline__ float frozen_block_exact(float acc, double k, float pen_o, float pen_u, int count) {
  if (pen_o != 0.0f || pen_u != 0.0f) {
    for (; count > 0; --count) {
      acc = (float)((double)acc + k);
      acc = acc + pen_o;
      acc = acc + pen_u;
    }
    return acc;
  }
  return frozen_block(acc, k, 0.0f, count);
}

template <bool POW2RES, bool GEN>
__global__ __launch_bounds__(1024) void k_rollout_scan_exact(DevParams P, const uint16_t* __restrict__ cells16,
                                                             const uint32_t* __restrict__ cells,
                                                             const float2* __restrict__ noise, NoiseJob gen,
                                                             const float2* __restrict__ u, float* __restrict__ costs,
                                                             float* __restrict__ w_rel, ScanPackets pk,
                                                             PendingApply pend) {
  extern __shared__ double2 scan_lds[];
  using L = ScanExactLds;
  constexpr int R = L::R, CHL = L::CHL, S = 64 / R;
  const int c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave = 8 steps
  const int lane = threadIdx.x & 63;
  const int r = lane & (R - 1), h = lane / R;
  const int W = (int)(blockDim.x >> 6);
  const int K = W * S;
  const int k = c * S + h;
  [[maybe_unused]] const bool stamp_wg = blockIdx.x == 5;
  [[maybe_unused]] const int stamp_base = 64 + 16 * c;
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 0);
  DevParams Q = P;
  const int tile = blockIdx.x;
  const float2* uq = select_instance(Q, u, Q.inst ? (tile * R) / Q.n_inst : 0);
  const int T = Q.n_steps, N = Q.n_local;
  const int Tp = 8 * W;
  const int n = tile * R + r;
  const bool live = n < N;
  const int t0 = k * CHL;
  const int nvalid = min(max(T - t0, 0), CHL);

  char* base = reinterpret_cast<char*>(scan_lds);
  float2* e2 = reinterpret_cast<float2*>(base);                                  // [Tp][R] (swizzled)
  double* ccr = reinterpret_cast<double*>(base + L::e2(W));                      // [K][R][CHL]
  double* p0 = reinterpret_cast<double*>(base + L::e2(W) + L::ccr(W));           // [Tp][R]
  double* p1 = p0 + (size_t)Tp * R;                                              // [Tp][R]
  char* rec = reinterpret_cast<char*>(p0);                                       // [K][R] {double sg[CHL], float po[CHL], float pu[CHL]}
  float2* pos = reinterpret_cast<float2*>(p1 + (size_t)Tp * R);                  // [Tp + 1][R]
  // the headings live in the UPPER half of the position rows: the position walk, which starts while
  // other waves still read headings, overwrites heading t' = 2t - Tp <= t when it stores position t,
  // and it gets to step t only after every wave up to step t's has finished with its headings
  float* th_sh = reinterpret_cast<float*>(pos) + (size_t)Tp * R;                 // [Tp][R]
  char* small = reinterpret_cast<char*>(pos) + L::p2(W);
  double2* uos = reinterpret_cast<double2*>(small);                              // [Tp] u / std^2
  double* fz_k = reinterpret_cast<double*>(uos + Tp);                            // [R]
  double* term_sh = fz_k + R;                                                    // [R]
  uint32_t* evw = reinterpret_cast<uint32_t*>(term_sh + R);                      // [R][2]
  float* fz_po = reinterpret_cast<float*>(evw + 2 * R);                          // [R]
  float* fz_pu = fz_po + R;                                                      // [R]
  int* fz_count = reinterpret_cast<int*>(fz_pu + R);                             // [R]
  float* wsh = reinterpret_cast<float*>(fz_count + R);                           // [R]
  uint32_t* flags = reinterpret_cast<uint32_t*>(wsh + R);
  // The walks start before the stage that feeds them has ended everywhere: wave g raises done_a[g]
  // when the increments of ITS 8 steps are in LDS (done_b[g]: its position increments), and the
  // walking wave waits for the flag of the group it is about to read -- the waves finish their
  // Philox blocks a SIMD's worth at a time, and the walk of the first groups fits in between.
  int* done_a = reinterpret_cast<int*>(flags + 4);  // [16]
  int* done_b = done_a + 16;                        // [16]
  // a sharded iteration's update applied here (update_kernels.h, PendingApply): the updated sequence
  double* scale_sh = reinterpret_cast<double*>(small + (size_t)Tp * 16 + R * 64 + 64);  // [kMaxFoldedRanks + 2]
  float2* u_sh = reinterpret_cast<float2*>(scale_sh + kMaxFoldedRanks + 2);              // [Tp]
  const bool folded = pend.packets != nullptr;
  if (folded && c == 0) pending_apply_prepare(pend, lane, scale_sh);
  if (c == 0 && lane < R) {
    evw[2 * lane] = 0u;
    evw[2 * lane + 1] = 0u;
    fz_count[lane] = 0;
    if (lane < 2) flags[lane] = 0u;  // [0] a failed vote, [1] a penalty somewhere in the tile
    done_a[lane] = 0;  // (R = 32 >= the two arrays of 16)
  }
  lds_barrier();  // (every wave has only just started)
  auto raise = [&](int* flag) {
    if (lane == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto wait_for = [&](const int* flag) {
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
  };

  // ---------------------------------------------------------------- A
  float2 ut[CHL];
  if (folded) {  // (wave-uniform) the 8 controls of this wave's steps from the ranks' packets
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
  const uint32_t ref = scan_lookup<POW2RES>(Q, cells16, Q.x0, Q.y0) & 0x3fffu;
  if (lane < 8) {  // the control ratios of this wave's 8 steps (float64 quotients: mppi.py:709)
    const float2 ul = folded ? u_sh[8 * c + lane] : uq[min(8 * c + lane, T - 1)];
    uos[8 * c + lane] = make_double2((double)ul.x / Q.s0sq, (double)ul.y / Q.s1sq);
  }
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
  const double dt64 = (double)Q.dt;
  double qx[CHL];  // dt * clipped speed: exact products of float32 factors
#pragma unroll
  for (int j = 0; j < CHL; ++j) {
    const bool valid = j < nvalid;
    ut[j] = valid ? ut[j] : make_float2(0.0f, 0.0f);
    e[j] = valid ? e[j] : make_float2(0.0f, 0.0f);
    const int t = t0 + j;
    e2[t * R + (r ^ (t & (R - 1)))] = e[j];
    const float v = clip_f32(ut[j].x + e[j].x, Q.v_lo, Q.v_hi);
    const float w = clip_f32(ut[j].y + e[j].y, Q.w_lo, Q.w_hi);
    qx[j] = dt64 * (double)v;
    p0[(size_t)t * R + r] = dt64 * (double)w;
  }
  const double vtr0 = fma(Q.lin_ratio, (double)(int)(ref & 127u), Q.lin_lo);
  const double wtr0 = fma(Q.ang_ratio, (double)(int)((ref >> 7) & 127u), Q.ang_lo);
  raise(&done_a[c]);
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 1);

  // ---------------------------------------------------------------- W1: the heading walk
  // one running sum rounded to float32 after every fma (a lone wave issues an instruction per ~5
  // cycles: what counts is the instruction count -- one pointer bump and a group of reads per 8 steps)
  // OUT: float rows of `out_stride` floats per step (R for the headings, 2 R for the float2
  // positions); every lane stores (lanes 32..63 mirror 0..31: the same value to the same address),
  // at immediate offsets from one pointer that moves once per 8 steps
  auto walk = [&](const double* inc, const int* ready, double coeff, float start, float* out, auto out_stride_tag) {
    constexpr int OS = decltype(out_stride_tag)::value;
    const double* at = inc + r;
    float* to = out + r * (OS / R);
    double a[8], b[8];
    int g_next = 0;
    auto load = [&](double (&dst)[8]) {  // the increments of the next group of 8 steps, once its wave has stored them
      if (g_next < W) wait_for(&ready[g_next]);
      ++g_next;
#pragma unroll
      for (int q = 0; q < 8; ++q) dst[q] = at[(size_t)q * R];
      at += 8 * R;
    };
    float vf = start;
    double v64 = (double)start;
    auto run = [&](const double (&src)[8]) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        to[q * OS] = vf;  // the value BEFORE the step
        vf = (float)fma(coeff, src[q], v64);
        v64 = (double)vf;
      }
      to += 8 * OS;
    };
    load(a);
    for (int g = 0; g + 2 <= W; g += 2) {
      load(b);
      run(a);
      load(a);
      run(b);
    }
    if (W & 1) run(a);
    return vf;  // the value after the last of the 8 W steps
  };
  if (c == 0) {
    __builtin_amdgcn_s_setprio(3);
    (void)walk(p0, done_a, wtr0, Q.th0, th_sh, PhaseTag<R>());
    __builtin_amdgcn_s_setprio(0);
  } else {
    // meanwhile: lambda * (u0/s0^2 * e0 + u1/s1^2 * e1) in float64   (mppi.py:1007-1009)
#pragma unroll
    for (int j = 0; j < CHL; ++j) ccr[((size_t)k * R + r) * CHL + j] = control_cost(Q, uos[min(t0 + j, Tp - 1)], e[j]);
  }
  if (c == 0) {
#pragma unroll
    for (int j = 0; j < CHL; ++j) ccr[((size_t)k * R + r) * CHL + j] = control_cost(Q, uos[min(t0 + j, Tp - 1)], e[j]);
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 2);
  lds_barrier();

  // ---------------------------------------------------------------- B: sin / cos of every heading
  {
    // (one full evaluation per lane; its other three headings by the exact-increment rotation of the
    //  pipelined kernels when the increment is small enough for it -- |delta| <= 0.36 rad, else in full)
    float thv[CHL];
#pragma unroll
    for (int j = 0; j < CHL; ++j) thv[j] = th_sh[(size_t)(t0 + j) * R + r];
    double s, cs;
    sincos_f64<false>((double)thv[0], s, cs);
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      const int t = t0 + j;
      p0[(size_t)t * R + r] = qx[j] * cs;
      p1[(size_t)t * R + r] = qx[j] * s;
      if (j + 1 < CHL) {
        const double delta = (double)thv[j + 1] - (double)thv[j];  // exact: both are float32 values
        if (__all(fabs(delta) <= 0.36)) rotate_sincos_f64(delta, s, cs);
        else sincos_f64<false>((double)thv[j + 1], s, cs);
      }
    }
  }
  raise(&done_b[c]);
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 3);

  // ---------------------------------------------------------------- W2: the position walks
  if (c == 0) {
    __builtin_amdgcn_s_setprio(3);
    float* px = reinterpret_cast<float*>(pos);
    px[(size_t)Tp * 2 * R + 2 * r] = walk(p0, done_b, vtr0, Q.x0, px, PhaseTag<2 * R>());
    if (W == 1) px[(size_t)Tp * 2 * R + 2 * r + 1] = walk(p1, done_b, vtr0, Q.y0, px + 1, PhaseTag<2 * R>());
    __builtin_amdgcn_s_setprio(0);
  } else if (c == 1) {
    __builtin_amdgcn_s_setprio(3);
    float* py = reinterpret_cast<float*>(pos) + 1;
    py[(size_t)Tp * 2 * R + 2 * r] = walk(p1, done_b, vtr0, Q.y0, py, PhaseTag<2 * R>());
    __builtin_amdgcn_s_setprio(0);
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 4);
  lds_barrier();

  // ---------------------------------------------------------------- C: lookups, stage costs, events
  double sg[CHL], n2[CHL];
  float xa[CHL + 1], ya[CHL + 1];
  float po[CHL], pu[CHL];
  uint32_t zero_bits = 0, mism_bits = 0, hit_bits = 0;
  const double gt2 = (double)Q.gt2;
  {
#pragma unroll
    for (int j = 0; j <= CHL; ++j) {
      const float2 pj = pos[(size_t)min(t0 + j, Tp) * R + r];
      xa[j] = pj.x;
      ya[j] = pj.y;
    }
    uint32_t cell[CHL];
#pragma unroll
    for (int j = 0; j < CHL; ++j) cell[j] = scan_lookup<POW2RES>(Q, cells16, xa[j], ya[j]);  // the cell step j STARTS in
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      const double dx = (double)(Q.xg - xa[j + 1]), dy = (double)(Q.yg - ya[j + 1]);
      n2[j] = fma(dx, dx, dy * dy);
      sg[j] = fma(Q.dist_weight, sqrt_newton_nz_f64(n2[j]), dt64);
      hit_bits |= (n2[j] <= gt2 ? 1u : 0u) << j;
    }
    pin_memory_order();
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      const uint32_t cl = cell[j];
      zero_bits |= ((int)(cl & 127u) == Q.lin_zero_byte ? 1u : 0u) << j;
      mism_bits |= (((cl ^ ref) & 0x3fffu) != 0u ? 1u : 0u) << j;
      po[j] = (cl & 0x4000u) ? Q.obs_cost : 0.0f;
      pu[j] = (cl & 0x8000u) ? Q.unk_cost : 0.0f;
    }
  }
  const uint32_t vmask = (1u << nvalid) - 1u;
  const int s = __builtin_ctz((zero_bits & vmask) | (1u << CHL));
  const int hh = __builtin_ctz((hit_bits & vmask & ((1u << s) - 1u)) | (1u << CHL));
  const bool is_hit = hh < CHL, froze = !is_hit && s < nvalid;
  int n_act = is_hit ? hh + 1 : min(s, nvalid);
  const bool bad = (mism_bits & ((1u << n_act) - 1u)) != 0u;
  const uint32_t ev = is_hit ? 1u : (froze ? 2u : 0u);
  double f_k = 0.0, f_d2 = 1e9;
  float f_po = 0.0f, f_pu = 0.0f;
  bool f_hit = false;
  if (__any(froze)) {
    float fx = xa[0], fy = ya[0];
    f_po = po[0];
    f_pu = pu[0];
#pragma unroll
    for (int j = 1; j < CHL; ++j) {
      fx = s == j ? xa[j] : fx;
      fy = s == j ? ya[j] : fy;
      f_po = s == j ? po[j] : f_po;
      f_pu = s == j ? pu[j] : f_pu;
    }
    // a rollout in a cell of zero linear traction stays where it is: x = float32(fma(0, ., x))
    const double dx = (double)(Q.xg - fx), dy = (double)(Q.yg - fy);
    f_d2 = fma(dx, dx, dy * dy);
    f_k = fma(Q.dist_weight, sqrt_newton_nz_f64(f_d2), dt64);
    f_hit = f_d2 <= gt2;
  }
  if (ev != 0u) atomicOr(&evw[2 * r + ((2 * k) >> 5)], ev << ((2 * k) & 31));
  {
    bool pen = false;
#pragma unroll
    for (int j = 0; j < CHL; ++j) pen = pen || (j < nvalid && (po[j] != 0.0f || pu[j] != 0.0f));
    if (__any(pen) && lane == 0) atomicOr(&flags[1], 1u);
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 5);
  lds_barrier();

  // ---------------------------------------------------------------- D'
  {
    const uint32_t w0 = evw[2 * r], w1 = evw[2 * r + 1];
    const uint64_t word = ((uint64_t)w1 << 32) | w0;
    const bool dead = (word & ((1ull << (2 * k)) - 1ull)) != 0ull;
    n_act = dead ? 0 : n_act;
    const bool owner = !dead && ev != 0u;
    const bool last = t0 < T && t0 + CHL >= T;
    if (owner || (!dead && last)) {
      double term = 0.0;  // (1 - reached) * sqrt(d2) / (v_post + 1e-6)   (mppi.py:26-28, 1005)
      if (ev == 2u) {
        term = f_hit ? 0.0 : sqrt(f_d2) / Q.v_post_den;
      } else if (ev == 0u) {
        double n2l = n2[0];
#pragma unroll
        for (int j = 1; j < CHL; ++j) n2l = (nvalid - 1 == j) ? n2[j] : n2l;
        term = sqrt(n2l) / Q.v_post_den;
      }
      term_sh[r] = term;
      if (ev == 2u) {
        fz_k[r] = f_k;
        fz_po[r] = f_po;
        fz_pu[r] = f_pu;
        fz_count[r] = f_hit ? 1 : T - (t0 + s);
      }
    }
    if (__any(!dead && bad) && lane == 0) atomicOr(&flags[0], 1u);
    char* out = rec + ((size_t)k * R + r) * (CHL * 16);
    double2* o2 = reinterpret_cast<double2*>(out);
    o2[0] = make_double2(0 < n_act ? sg[0] : 0.0, 1 < n_act ? sg[1] : 0.0);
    o2[1] = make_double2(2 < n_act ? sg[2] : 0.0, 3 < n_act ? sg[3] : 0.0);
    float4* o4 = reinterpret_cast<float4*>(out + CHL * 8);
    o4[0] = make_float4(0 < n_act ? po[0] : 0.0f, 1 < n_act ? po[1] : 0.0f, 2 < n_act ? po[2] : 0.0f, 3 < n_act ? po[3] : 0.0f);
    o4[1] = make_float4(0 < n_act ? pu[0] : 0.0f, 1 < n_act ? pu[1] : 0.0f, 2 < n_act ? pu[2] : 0.0f, 3 < n_act ? pu[3] : 0.0f);
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 6);
  lds_barrier();

  // ---------------------------------------------------------------- E: the cost walk
  if (c == 0) {
    __builtin_amdgcn_s_setprio(3);
    const bool failed = flags[0] != 0u;
    float cost = 0.0f;
    if (!failed) {
      // records of 4 steps: {double sg[4]; float po[4]; float pu[4]} = 4 x 16 bytes, two records per group.
      // A tile that met no obstacle or unknown cell (flags[1] clear) adds +0.0f twice per step, which
      // leaves a cost >= +0 as it is: its walk is the three instructions of the float64 add alone.
      auto stage_walk = [&](auto pen_tag) {
        constexpr bool PEN = decltype(pen_tag)::value != 0;
        constexpr int G = 2;
        double2 ga[G * 2], gb[G * 2];
        float4 fa[G * 2], fb[G * 2];
        const char* at = rec + (size_t)r * 64;
        auto load = [&](double2 (&d)[G * 2], float4 (&f)[G * 2]) {
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const char* in = at + (size_t)g * R * 64;
            d[2 * g] = reinterpret_cast<const double2*>(in)[0];
            d[2 * g + 1] = reinterpret_cast<const double2*>(in)[1];
            if constexpr (PEN) {
              f[2 * g] = reinterpret_cast<const float4*>(in + 32)[0];
              f[2 * g + 1] = reinterpret_cast<const float4*>(in + 32)[1];
            }
          }
          at += (size_t)G * R * 64;
        };
        auto step = [&](double sgv, float ov, float qv) {  // mppi.py:994, 997, 998
          cost = (float)((double)cost + sgv);
          if constexpr (PEN) {
            cost = cost + ov;
            cost = cost + qv;
          }
        };
        auto add_record = [&](const double2 (&d)[G * 2], const float4 (&f)[G * 2], int g) {
          const double2 s01 = d[2 * g], s23 = d[2 * g + 1];
          const float4 o = f[2 * g], q = f[2 * g + 1];
          step(s01.x, o.x, q.x);
          step(s01.y, o.y, q.y);
          step(s23.x, o.z, q.z);
          step(s23.y, o.w, q.w);
        };
        load(ga, fa);
        int i = 0;
        for (; i + 2 * G <= K; i += 2 * G) {
          load(gb, fb);
#pragma unroll
          for (int g = 0; g < G; ++g) add_record(ga, fa, g);
          load(ga, fa);
#pragma unroll
          for (int g = 0; g < G; ++g) add_record(gb, fb, g);
        }
        if (i + G <= K) {
          load(gb, fb);
#pragma unroll
          for (int g = 0; g < G; ++g) add_record(ga, fa, g);
          i += G;
#pragma unroll
          for (int g = 0; g < G - 1; ++g)
            if (i + g < K) add_record(gb, fb, g);
        } else {
#pragma unroll
          for (int g = 0; g < G - 1; ++g)
            if (i + g < K) add_record(ga, fa, g);
        }
      };
      if (flags[1] != 0u) stage_walk(PhaseTag<1>());
      else stage_walk(PhaseTag<0>());
      MPPI_STAMP(stamp_wg, stamp_base + 9);
      const int cnt = fz_count[r];
      if (__any(cnt > 0)) cost = frozen_block_exact(cost, fz_k[r], fz_po[r], fz_pu[r], cnt);
      MPPI_STAMP(stamp_wg, stamp_base + 10);
      cost = (float)((double)cost + term_sh[r]);
    } else {
      // ---- the tile step by step, with the tractions of the visited cells: k_rollout_map's arithmetic
      if (lane == 0 && Q.spec_failures) {
        atomicAdd_system(Q.spec_failures, 1u);
        __threadfence_system();
      }
      RolloutState st = {Q.x0, Q.y0, Q.th0, 0.0f, 1e9, false, false};
      for (int t = 0; t < T; ++t) {
        map_step<MAP_DET, true, false, false>(Q, cells, nullptr, nullptr, folded ? u_sh[t] : uq[t], e2[t * R + (r ^ (t & (R - 1)))], st);
        if (__all(st.done)) break;
      }
      cost = (float)((double)st.cost + (st.reached ? 0.0 : 1.0) * sqrt(st.d2) / Q.v_post_den);
    }
    // the control cost of all T steps, also after an early goal break (mppi.py:1007-1009)
    {
      constexpr int G = 4;  // records of 4 doubles
      double2 ga[G * 2], gb[G * 2];
      const double* at = ccr + (size_t)r * CHL;
      auto load = [&](double2 (&d)[G * 2]) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          d[2 * g] = reinterpret_cast<const double2*>(at + (size_t)g * R * CHL)[0];
          d[2 * g + 1] = reinterpret_cast<const double2*>(at + (size_t)g * R * CHL)[1];
        }
        at += (size_t)G * R * CHL;
      };
      auto add_record = [&](const double2 (&d)[G * 2], int g) {
        cost = (float)((double)cost + d[2 * g].x);
        cost = (float)((double)cost + d[2 * g].y);
        cost = (float)((double)cost + d[2 * g + 1].x);
        cost = (float)((double)cost + d[2 * g + 1].y);
      };
      // (steps past the horizon hold zero noise: their terms are +0.0)
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
    // first half of the control update (update_kernels.h): weights relative to the tile's minimum,
    // with emit_tile_weights' expression
    const float beta = wave_min_f32(live ? cost : __builtin_inff());
    const float wr = mine ? (float)exp(-1.0 / (double)Q.lambda * (double)(cost - beta)) : 0.0f;
    if (mine) w_rel[n] = wr;
    if (lane < R) wsh[lane] = wr;
    const float den = wave_sum_to_lane63_f32(wr);
    if (lane == 63) {
      pk.tbeta[tile] = beta;
      pk.tden[tile] = den;
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
      pk.tnum[(size_t)t * pk.n_tiles + tile] = make_float2(ax, ay);
    }
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 8);
}

}  // namespace mppi
