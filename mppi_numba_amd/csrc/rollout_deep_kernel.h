// rollout_deep_kernel.h -- k_rollout_deep: the latency-regime rollout as a five-stage software
// pipeline over ONE tile of 64 rollouts per CU, built on a speculation about the traction map
// (gfx950, wave64).
//
// Replaces rollout_det_dyn_numba (mppi.py:916-1009) with the reference's rounding points (bits
// identical to k_rollout_pipe / k_rollout_spec / the oracle).
//
// What bounds a step.  A wave that is alone on its SIMD issues ONE instruction per ~5 cycles,
// float32 or float64, however independent its instructions are; conversions to and from float64
// take 8, a 16-byte LDS store 27-52 (tools/microbench/issue_cost.hip).  N = 8192 gives every CU one
// tile, so a step costs what its busiest wave issues.  In k_rollout_pipe that is the state wave:
// position -> cell -> LDS lookup -> traction -> heading -> (cos, sin) rotation -> position is ONE
// dependent chain of 53 instructions (~340-360 cycles per step measured): the heading needs the
// angular traction of the visited cell, the position needs the heading.  Here every tile ASSUMES
// that all of its lookups return the traction bytes of the start cell -- traction is piecewise
// constant; over maps of nominal dynamics (the reference's own use_det_dynamics recipe,
// README.md:136-151) or large terrain patches the assumption holds for the whole horizon -- and
// the chain falls apart into stages that no longer wait for each other:
//   stage 0  P  streams the noise three chunks ahead and hands each chunk on the moment it is there;
//               the control-cost products (needed only by the tail) follow one interval later
//   stage 1  H  clipped controls v, w = clip(u + noise) (moved here from P: that wave shares its
//               SIMD with V and was the longest pole, 2170 -> 1940 cycles per interval);
//               heading theta' = float32(theta + wtr0*dt*w), (cos, sin) by the exact-increment
//               rotation, the products dt*v*cos, dt*v*sin            (no lookup needed)
//   stage 2  V  position x' = float32(x + vtr0*dt*v*cos) (3 instructions per axis), and -- off that
//               chain -- the LDS map lookup of every position
//   stage 3  S  the vote on the assumption; a rollout that enters a cell of ZERO linear traction
//               never moves again, whatever its heading does: it is frozen there, exactly as the
//               reference computes, and stops voting (the zero-traction padding ring of every map
//               would otherwise fail some lane of every tile); obstacle / unknown bits, squared goal
//               distance, sqrt (float32 seed + one Newton step), stage cost
//   stage 4  C  penalties, goal test, float32-rounded accumulation; then terminal cost, control
//               costs, the tile's half of the weight computation.
// Chunks of CH steps go from stage to stage through LDS rings, one light workgroup barrier (LDS
// complete, global loads in flight) per chunk.  (An 11-wave, 10-stage cut of the same work was
// measured too: the ring traffic and per-interval overhead of so many hand-offs took the SIMDs to
// ~4 cycles per instruction and the step to 325 cycles -- more waves than instructions to share.)
// A lookup that contradicts the assumption for a rollout that is still moving (S votes per
// chunk) abandons the speculation: all waves leave the pipeline after the same barrier, waves 0-2
// re-execute the tile from its first step on the exact schedule of k_rollout_pipe
// (pipe_tile_body), the others retire.  Nothing computed under the assumption reaches global
// memory before the last stage has seen the last chunk.  A map whose traction changes from cell to
// cell pays the three intervals until the first vote; the host can also launch with speculate = 0
// (exact schedule from the start).
//
// LDS: [Tp] double2 ratios | [Tp] float2 u | map window | rings (DeepRing) | 2 fail flags |
//      (CC_LDS) [Tp][64] double control-cost products          (Tp = T rounded up to 8)
#pragma once
#include <type_traits>
#include "rollout_spec_kernel.h"

namespace mppi {

template <int CH>
struct DeepRing {
  static constexpr int kE = CH * 64;             // entries per chunk
  static constexpr int oVw = 0;                  // vw[2][kE] float2  control noise (H adds u and clips)  P -> H
  static constexpr int oPp = oVw + 2 * kE * 8;   // pp[2][kE] double2 dt*v*{cos, sin}         H -> V
  static constexpr int oXy = oPp + 2 * kE * 16;  // xy[2][kE] float2  position after step     V -> S
  static constexpr int oCl = oXy + 2 * kE * 8;   // cl[2][kE] uint32  16-bit map cell         V -> S
  static constexpr int oN2 = oCl + 2 * kE * 4;   // n2[2][kE] double  squared goal distance   S -> C
  static constexpr int oFl = oN2 + 2 * kE * 8;   // fl[2][64] uint32  2 flag bits per step    S -> C
  static constexpr int oFail = oFl + 2 * 64 * 4;  // fail[2] int (by interval parity) + padding
  static constexpr int kBytes = oFail + 16;
};

enum DeepRole { kP = 0, kH, kS, kC, kV, kDeepWaves };  // (wave 4 shares its SIMD with wave 0: the lightest role)

// F32 (MPPI_MATH_FAST): the same pipeline in float32 -- no float64 intermediates, hardware sin / cos
// of the heading instead of the exact-increment rotation, hardware sqrt.  Costs then follow the
// reference to ~1e-6 relative instead of bit for bit (a rollout that grazes a cell border may
// land on the other side); it exists to MEASURE what the reference's rounding points cost in
// this design and as an opt-in for users who want the latency.  A failed
// vote still re-executes the tile on the exact float64 schedule.
template <int CH, bool POW2RES, bool CC_LDS, bool F32 = false>
__global__ __launch_bounds__(64 * kDeepWaves) void k_rollout_deep(
    DevParams P, const uint16_t* __restrict__ cells16, const float2* __restrict__ noise,
    const float2* __restrict__ u, float* __restrict__ costs, float* __restrict__ w_rel,
    float* __restrict__ tile_beta, double* __restrict__ cc_scratch, int map_bytes, int n_rollout_blocks,
    int speculate, NoiseJob next_noise) {
  extern __shared__ double2 uos[];
  MPPI_STAMP(threadIdx.x == 0 && blockIdx.x < 512, 2048 + 2 * blockIdx.x);  // every workgroup: entry ...
  if ((int)blockIdx.x >= n_rollout_blocks) {
    // spare workgroups: the noise of the NEXT iteration, into the other noise buffer
    MPPI_STAMP(threadIdx.x == 0 && ((int)blockIdx.x == n_rollout_blocks || blockIdx.x == gridDim.x - 1),
               (int)blockIdx.x == n_rollout_blocks ? 16 : 18);
    if (next_noise.out)
      noise_generate<true>(next_noise, (blockIdx.x - n_rollout_blocks) * (blockDim.x >> 6) + (threadIdx.x >> 6),
                     (gridDim.x - n_rollout_blocks) * (blockDim.x >> 6));
    MPPI_STAMP(threadIdx.x == 0 && ((int)blockIdx.x == n_rollout_blocks || blockIdx.x == gridDim.x - 1),
               (int)blockIdx.x == n_rollout_blocks ? 17 : 19);
    MPPI_STAMP(threadIdx.x == 0 && blockIdx.x < 512, 2049 + 2 * blockIdx.x);  // ... and exit
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  bool exact = !speculate;
  if (!exact) {
    // ================================================================ speculative pipeline
    [[maybe_unused]] const bool stamp_wg = blockIdx.x == 5;
    MPPI_STAMP(stamp_wg && threadIdx.x == 0, 0);
    __builtin_amdgcn_s_setprio(3);  // win the issue arbitration against the noise-generating waves
    DevParams Q = P;
    const float2* uq = select_instance(Q, u, Q.inst ? (int)blockIdx.x / Q.inst_tiles : 0);
    const int T = Q.n_steps, N = Q.n_local;
    const int Tp = (T + 7) & ~7;
    float2* us = reinterpret_cast<float2*>(uos + Tp);
    uint16_t* lds_map = reinterpret_cast<uint16_t*>(uos + Tp + Tp / 2);
    char* rings = reinterpret_cast<char*>(lds_map) + map_bytes;
    using R = DeepRing<CH>;
    float2* ring_vw = reinterpret_cast<float2*>(rings + R::oVw);
    double2* ring_pp = reinterpret_cast<double2*>(rings + R::oPp);
    float2* ring_xy = reinterpret_cast<float2*>(rings + R::oXy);
    uint32_t* ring_cl = reinterpret_cast<uint32_t*>(rings + R::oCl);
    double* ring_n2 = reinterpret_cast<double*>(rings + R::oN2);
    uint32_t* ring_fl = reinterpret_cast<uint32_t*>(rings + R::oFl);
    int* fail = reinterpret_cast<int*>(rings + R::oFail);
    double* cc_lds = reinterpret_cast<double*>(rings + R::kBytes);  // [Tp][64]
    const int K = (T + CH - 1) / CH;
    constexpr int kLastStage = 4;
    const int tile = blockIdx.x;
    const int n = tile * 64 + lane;
    const bool live = n < N;
    const bool tile_ok = tile * 64 < N;
    const size_t tile_base = (size_t)tile * T * 64 + lane;  // + t*64: element (t, n)
    constexpr int E = R::kE;
    [[maybe_unused]] const int stamp_base = 64 + 32 * wave;

    // ---- prologue.  No workgroup barrier before the pipeline starts: P and H need no map (H takes
    //      the traction bytes of the start cell straight from global memory), V's first lookups
    //      are two intervals away -- so the three waves that idle through intervals 0 and 1 (S, C, V)
    //      request the whole map window now and store it to LDS in interval 1.
    const float2* col = noise + (tile_ok ? tile_base : (size_t)0);
    float2 e[4][CH];  // producer only: e[c & 3] = noise of chunk c, requested four intervals ahead
    auto load_noise = [&](float2 (&dst)[CH], int chunk) {
      // (the noise buffers are padded: rows past the horizon are read unclamped and ignored)
      // (chunks past the horizon re-read the last one: those loads only have to come back quickly --
      //  the unrolled loop's header waits for everything in flight)
      const float2* at = col + (size_t)min(chunk, K - 1) * CH * 64;
#pragma unroll
      for (int j = 0; j < CH; ++j) dst[j] = at[j * 64];
    };
    // (the first two runs of 64 controls are requested BEFORE the noise: loads come back in order,
    //  and behind 32 noise loads each of the two dependent trips of the staging loop below cost
    //  ~2k cycles of the prologue)
    float2 u_first[2];
    if (wave == kP) {
      u_first[0] = uq[min(lane, T - 1)];
      u_first[1] = uq[min(lane + 64, T - 1)];
      load_noise(e[0], 0);
      load_noise(e[1], 1);
      load_noise(e[2], 2);
      load_noise(e[3], 2);  // (a defined value for the set the first interval "finishes": its products are not stored)
    }
    const float win_c0f = (float)Q.win_c0, win_r0f = (float)Q.win_r0;
    const float win_last_col = (float)(Q.win_cols - 1), win_last_row = (float)(Q.win_rows - 1);
    const int win_pitch_bytes = 2 * Q.win_cols;
    const char* lds_bytes = reinterpret_cast<const char*>(lds_map);
    auto window_coords = [&](float x, float y, int& xi, int& yi) {
      // the window holds every cell reachable within the horizon (host-proved); the clamp is for
      // memory safety only (and for the positions of frozen rollouts, which V keeps integrating)
      if (POW2RES) {  // res is a power of two: see cell_coord_pow2
        xi = cell_coord_pow2(x, Q.xlo, Q.inv_res, win_c0f, win_last_col);
        yi = cell_coord_pow2(y, Q.ylo, Q.inv_res, win_r0f, win_last_row);
      } else {
        xi = clamp_index(floordiv_to_int(x - Q.xlo, Q.res, Q.inv_res) - Q.win_c0, Q.win_cols);
        yi = clamp_index(floordiv_to_int(y - Q.ylo, Q.res, Q.inv_res) - Q.win_r0, Q.win_rows);
      }
    };
    auto lookup = [&](float x, float y) -> uint32_t {
      int xi, yi;
      window_coords(x, y, xi, yi);
      return *reinterpret_cast<const uint16_t*>(lds_bytes + (__mul24(yi, win_pitch_bytes) + (xi << 1)));
    };
    // the assumption: every visited cell carries the traction bytes of the start cell
    uint32_t ref;
    {
      int xi, yi;
      window_coords(Q.x0, Q.y0, xi, yi);
      ref = cells16[(size_t)(yi + Q.win_r0) * Q.pitch16 + (xi + Q.win_c0)] & 0x3fffu;
    }
    // the map window: vector i of the window (row-major, 16 bytes) by thread i % 192 of waves 2..4
    constexpr int kCopyVecs = 14;  // per thread in registers (2688 vectors = 42 KiB); a larger window's
                                   // remainder is copied before the first barrier
    u32x4 win_v[kCopyVecs];
    const int copy_tid = (int)threadIdx.x - 128;
    const int win_vpr = Q.win_cols >> 3, win_total = Q.win_rows * win_vpr;
    if (wave >= 2) {
      const u32x4* src = reinterpret_cast<const u32x4*>(cells16) + ((size_t)Q.win_r0 * Q.pitch16 + Q.win_c0) / 8;
      const int src_pitch = Q.pitch16 >> 3;
      const float inv_vpr = 1.0f / (float)win_vpr;
      auto source_of = [&](int i) {
        int r = (int)((float)i * inv_vpr);  // i < 2^20: within one of the quotient
        r -= (r * win_vpr > i);
        r += ((r + 1) * win_vpr <= i);
        return r * src_pitch + (i - r * win_vpr);
      };
#pragma unroll
      for (int q = 0; q < kCopyVecs; ++q) win_v[q] = src[source_of(min(copy_tid + q * 192, win_total - 1))];
      for (int i = copy_tid + kCopyVecs * 192; i < win_total; i += 192)
        reinterpret_cast<u32x4*>(lds_map)[i] = src[source_of(i)];
    }
    if (wave == kP) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int t = lane + 64 * r;
        if (t < Tp) {
          const float2 ut = t < T ? u_first[r] : make_float2(0.0f, 0.0f);
          us[t] = ut;
          if (F32) reinterpret_cast<float2*>(uos)[t] = make_float2(ut.x / (float)Q.s0sq, ut.y / (float)Q.s1sq);
          else uos[t] = make_double2((double)ut.x / Q.s0sq, (double)ut.y / Q.s1sq);
        }
      }
      for (int t = lane + 128; t < Tp; t += 64) {
        const float2 ut = t < T ? uq[t] : make_float2(0.0f, 0.0f);
        us[t] = ut;
        if (F32) reinterpret_cast<float2*>(uos)[t] = make_float2(ut.x / (float)Q.s0sq, ut.y / (float)Q.s1sq);
        else uos[t] = make_double2((double)ut.x / Q.s0sq, (double)ut.y / Q.s1sq);
      }
      if (lane < 2) fail[lane] = 0;  // (visible to the others after the first barrier; nobody reads before)
    }
    MPPI_STAMP(stamp_wg, stamp_base + 0);
    const double vtr0 = fma(Q.lin_ratio, (double)(int)(ref & 127u), Q.lin_lo);
    const double wtr0 = fma(Q.ang_ratio, (double)(int)((ref >> 7) & 127u), Q.ang_lo);
    [[maybe_unused]] const float vtr0f = (float)vtr0, wtr0f = (float)wtr0;
    using real = std::conditional_t<F32, float, double>;
    using real2 = std::conditional_t<F32, float2, double2>;
    MPPI_STAMP(stamp_wg, stamp_base + 1);
    // stored by waves 2..4 at the end of interval 1 (called from their loops below)
    auto store_window = [&]() {
      u32x4* dst = reinterpret_cast<u32x4*>(lds_map);
#pragma unroll
      for (int q = 0; q < kCopyVecs; ++q) dst[min(copy_tid + q * 192, win_total - 1)] = win_v[q];
    };

    // One interval of a stage: its work on chunk c = k - stage when that chunk exists, the light
    // barrier, the vote of the PREVIOUS interval (requested at the top, complete at the barrier;
    // S writes the slot of the interval it ran in, everybody reads the other one: race-free).
    // Returns 1 to leave on a failed vote, 2 after the last chunk has passed the last stage.
    auto interval_end = [&](int k, int flag) -> int {
      MPPI_STAMP(stamp_wg && k < 24, stamp_base + 2 + k);
      lds_barrier();
      asm volatile("" : "+v"(flag));  // (the flag register is written by the LDS unit: complete only here)
      if (flag) return 1;
      return k >= K - 1 + kLastStage ? 2 : 0;
    };
    auto read_flag = [&](int k) -> int {  // issued at the top of interval k: the vote of interval k-1
      if (k == 0) return 0;  // (the flags are being initialised)
      int v;
      typedef __attribute__((address_space(3))) int lds_int;
      const unsigned at = (unsigned)(size_t)(lds_int*)(fail + ((k + 1) & 1));  // LDS byte offset
      asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(at) : "memory");
      return v;
    };
    int outcome = 0;

    if (wave == kP) {
      // -------------------------------------------------------------- stage 0: controls and control costs
      // (loop unrolled by 4: static register sets.  The noise loads are issued in EVERY interval,
      //  also past the horizon, and the wait for the set consumed now -- requested three intervals
      //  earlier, two younger groups of CH loads may stay in flight -- is written out by hand with
      //  the registers as operands: hipcc's own vmcnt bookkeeping across the inline-asm barriers of
      //  the unrolled loop waits either for the newest loads or not at all.)
      real* my_cc = CC_LDS ? reinterpret_cast<real*>(cc_lds) + lane : reinterpret_cast<real*>(cc_scratch) + tile_base;
      auto step = [&](auto ph, int k) -> int {
        constexpr int PH = decltype(ph)::value;  // == k & 3: the register set holding chunk k
        constexpr int PM = (PH + 3) & 3;         // the set holding chunk k - 1
        MPPI_STAMP(stamp_wg && k == 5, 1100);
        const int flag = read_flag(k);
        // (a) chunk k's noise goes to the heading wave (which adds and clips the controls: it has issue
        //     slots to spare, this wave shares its SIMD with the position wave) as soon as it is there;
        //     nothing else in this interval is on anybody's critical path -- the first interval is one
        //     memory round trip plus eight stores instead of also a chunk of control-cost arithmetic
#pragma unroll
        for (int j = 0; j < CH; ++j)
          asm volatile("s_waitcnt vmcnt(%2)" : "+v"(e[PH][j].x), "+v"(e[PH][j].y) : "n"(2 * CH));
        MPPI_STAMP(stamp_wg && k == 5, 1101);
        if (k < K) {
          float2* out = ring_vw + (k & 1) * E;
#pragma unroll
          for (int j = 0; j < CH; ++j) out[j * 64 + lane] = e[PH][j];
        }
        MPPI_STAMP(stamp_wg && k == 5, 1102);
        // (b) the control-cost products of chunk k - 1 (needed only by the cost wave's tail), then that
        //     register set is reloaded with chunk k + 3
        const int c1 = min(max(k - 1, 0), K - 1);
        const real2* uos_c = reinterpret_cast<const real2*>(uos) + c1 * CH;
        real cc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          if constexpr (F32) cc[j] = Q.lambda * fmaf(uos_c[j].x, e[PM][j].x, uos_c[j].y * e[PM][j].y);
          else cc[j] = control_cost(Q, uos_c[j], e[PM][j]);
        }
        // the last use of this register set comes before its reload (results as operands of the
        // fence): the set then keeps its physical registers around the loop, and hipcc has no
        // copies of in-flight registers -- each behind an s_waitcnt vmcnt(0) -- to make at the latch
#pragma unroll
        for (int j = 0; j < CH; ++j) asm volatile("" : : "v"(cc[j]) : "memory");
        load_noise(e[PM], k + 3);
        pin_memory_order();
        MPPI_STAMP(stamp_wg && k == 5, 1103);
        if (k >= 1 && k - 1 < K) {
#pragma unroll
          for (int j = 0; j < CH; ++j)
            if (CC_LDS || (tile_ok && (k - 1) * CH + j < T))  // (the LDS array is padded to whole chunks: no predicate)
              my_cc[(size_t)((k - 1) * CH + j) * 64] = cc[j];
        }
        MPPI_STAMP(stamp_wg && k == 5, 1104);
        return interval_end(k, flag);
      };
      // (six rounds of the four register sets per loop iteration: at the loop header hipcc waits for
      //  EVERYTHING in flight -- a full memory round trip, 3.7k cycles measured when the 17th
      //  interval of a 100-step horizon crossed it -- so horizons up to 160 steps never get there)
      for (int k = 0;; k += 4) {
        if ((outcome = step(PhaseTag<0>(), k))) break;
        if ((outcome = step(PhaseTag<1>(), k + 1))) break;
        if ((outcome = step(PhaseTag<2>(), k + 2))) break;
        if ((outcome = step(PhaseTag<3>(), k + 3))) break;
      }
      // products in global scratch: out before the cost wave's tail reads them
      if (!CC_LDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (wave == kH) {
      // -------------------------------------------------------------- stage 1: heading (assumed traction)
      const double dt64 = (double)Q.dt;
      float th = Q.th0;
      double th64 = (double)th, s, c0;
      sincos_f64<false>(th64, s, c0);
      for (int k = 0;; ++k) {
        const int flag = read_flag(k), c = k - 1;
        if (c >= 0 && c < K) {
          const float2* in = ring_vw + (c & 1) * E;
          float2 vw[CH];
          double2 qd[CH], pp[CH];
          double sd[CH], cd[CH];
          const float2* us_c = us + c * CH;  // (staged by the producer before the first barrier; padded to Tp)
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const float2 e = in[j * 64 + lane], ut = us_c[j];
            vw[j] = make_float2(clip_f32(ut.x + e.x, Q.v_lo, Q.v_hi), clip_f32(ut.y + e.y, Q.w_lo, Q.w_hi));
          }
          pin_memory_order();
          if constexpr (F32) {
            // heading in float32, hardware sin / cos of every heading (no rotation chain to drift)
            float2 ppf[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
              const float qx = Q.dt * vw[j].x, qy = Q.dt * vw[j].y;
              ppf[j] = make_float2(qx * __cosf(th), qx * __sinf(th));
              th = fmaf(wtr0f, qy, th);
            }
            pin_memory_order();
            float2* outf = reinterpret_cast<float2*>(ring_pp) + (c & 1) * E;
#pragma unroll
            for (int j = 0; j < CH; ++j) outf[j * 64 + lane] = ppf[j];
            if ((outcome = interval_end(k, flag))) break;
            continue;
          }
#pragma unroll
          for (int j = 0; j < CH; ++j)  // dt*v, dt*w: exact products of float32 factors
            qd[j] = make_double2(dt64 * (double)vw[j].x, dt64 * (double)vw[j].y);
          // (a) the heading chain, (b) the increment polynomials of all steps side by side, (c) the
          //     rotation chain with the products
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
            pp[j] = make_double2(qd[j].x * c0, qd[j].x * s);
            apply_rotation_f64(sd[j], cd[j], s, c0);
          }
          pin_memory_order();
          double2* out = ring_pp + (c & 1) * E;
#pragma unroll
          for (int j = 0; j < CH; ++j) out[j * 64 + lane] = pp[j];
        }
        if ((outcome = interval_end(k, flag))) break;
      }
    } else if (wave == kV) {
      // -------------------------------------------------------------- stage 2: position chain, map lookups
      float x = Q.x0, y = Q.y0;
      double x64 = (double)x, y64 = (double)y;
      for (int k = 0;; ++k) {
        const int flag = read_flag(k), c = k - 2;
        if (c >= 0 && c < K) {
          const real2* in = reinterpret_cast<const real2*>(ring_pp) + (c & 1) * E;
          real2 pp[CH];
          float xa[CH + 1], ya[CH + 1];
          uint32_t cl[CH];
#pragma unroll
          for (int j = 0; j < CH; ++j) pp[j] = in[j * 64 + lane];
          pin_memory_order();
          xa[0] = x;
          ya[0] = y;
#pragma unroll
          for (int j = 0; j < CH; ++j) {  // (a 3-instruction chain per axis; one fma in float32)
            if constexpr (F32) {
              x = fmaf(vtr0f, pp[j].x, x);
              y = fmaf(vtr0f, pp[j].y, y);
            } else {
              x = (float)fma(vtr0, pp[j].x, x64);
              y = (float)fma(vtr0, pp[j].y, y64);
              x64 = (double)x;
              y64 = (double)y;
            }
            xa[j + 1] = x;
            ya[j + 1] = y;
          }
#pragma unroll
          for (int j = 0; j < CH; ++j) cl[j] = lookup(xa[j], ya[j]);  // the cell step j STARTS in
          pin_memory_order();
          float2* out_xy = ring_xy + (c & 1) * E;
          uint32_t* out_cl = ring_cl + (c & 1) * E;
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            out_xy[j * 64 + lane] = make_float2(xa[j + 1], ya[j + 1]);  // position AFTER step j
            out_cl[j * 64 + lane] = cl[j];
          }
        }
        if (k == 1) store_window();
        if ((outcome = interval_end(k, flag))) break;
      }
    } else if (wave == kS) {
      // -------------------------------------------------------------- stage 3: the vote; frozen rollouts;
      //                                                                 squared goal distance
      bool stuck = false;           // this rollout sits in a cell of zero linear traction
      float ex = Q.x0, ey = Q.y0;   // effective position (frozen once stuck)
      uint32_t ecell = 0;           // cell of the previous step (the cell a frozen rollout stays in)
      for (int k = 0;; ++k) {
        const int flag = read_flag(k), c = k - 3;
        if (c >= 0 && c < K) {
          const float2* in_xy = ring_xy + (c & 1) * E;
          const uint32_t* in_cl = ring_cl + (c & 1) * E;
          float2 xy[CH];
          uint32_t cl[CH];
          real n2[CH];
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            xy[j] = in_xy[j * 64 + lane];
            cl[j] = in_cl[j * 64 + lane];
          }
          pin_memory_order();
          uint32_t bad = 0, fl = 0;
          const int steps_left = T - c * CH;  // steps of this chunk inside the horizon
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const uint32_t cell = stuck ? ecell : cl[j];  // the cell this step starts in
            stuck = stuck || (int)(cell & 127u) == Q.lin_zero_byte;
            bad |= (!stuck && j < steps_left) ? ((cell ^ ref) & 0x3fffu) : 0u;
            fl |= (cell >> 14) << (2 * j);  // obstacle | unknown << 1 of the cell step j left
            ecell = cell;
            ex = stuck ? ex : xy[j].x;
            ey = stuck ? ey : xy[j].y;
            if constexpr (F32) {
              const float dx = Q.xg - ex, dy = Q.yg - ey;
              n2[j] = fmaf(dx, dx, dy * dy);
            } else {
              const double dx = (double)(Q.xg - ex), dy = (double)(Q.yg - ey);
              n2[j] = fma(dx, dx, dy * dy);
            }
          }
          pin_memory_order();
          real* out = reinterpret_cast<real*>(ring_n2) + (c & 1) * E;
#pragma unroll
          for (int j = 0; j < CH; ++j) out[j * 64 + lane] = n2[j];
          ring_fl[(c & 1) * 64 + lane] = fl;
          if (__any(bad != 0) && lane == 0) {
            fail[k & 1] = 1;  // read by everybody in interval k+1
            if (Q.spec_failures) {  // (once per tile: it leaves the pipeline)
              atomicAdd_system(Q.spec_failures, 1u);
              __threadfence_system();  // performed before this launch can be seen to have finished
            }
          }
        }
        if (k == 1) store_window();
        if ((outcome = interval_end(k, flag))) break;
      }
    } else {
      // -------------------------------------------------------------- stage 4: stage costs, accumulation
      const real gt2 = (real)Q.gt2, dt64 = (real)Q.dt;
      [[maybe_unused]] const float dist_weight_f = (float)Q.dist_weight;
      float cost = 0.0f;
      real d2 = (real)1e9;
      bool done = false, reached = false;
      for (int k = 0;; ++k) {
        const int flag = read_flag(k), c = k - 4;
        if (c >= 0 && c < K) {
          const real* in = reinterpret_cast<const real*>(ring_n2) + (c & 1) * E;
          real n2[CH], sg[CH];
#pragma unroll
          for (int j = 0; j < CH; ++j) n2[j] = in[j * 64 + lane];
          const uint32_t fl = ring_fl[(c & 1) * 64 + lane];
          pin_memory_order();
          // (a) the square roots of all steps side by side, (b) the accumulation chain
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            if constexpr (F32) sg[j] = fmaf(dist_weight_f, __builtin_amdgcn_sqrtf(n2[j]), dt64);  // v_sqrt_f32, 1 ulp
            else sg[j] = fma(Q.dist_weight, sqrt_newton_nz_f64(n2[j]), dt64);
          }
          const int count = min(CH, T - c * CH);
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            float c1;
            if constexpr (F32) c1 = cost + sg[j];
            else c1 = (float)((double)cost + sg[j]);
            // bits of the cell the step STARTED in (mppi.py:971-998): penalty or +0.0, selected by the
            // sign-extended flag bit (v_bfe_i32 + v_and_b32)
            c1 = c1 + __int_as_float(__float_as_int(Q.obs_cost) & __builtin_amdgcn_sbfe((int)fl, 2 * j, 1));
            c1 = c1 + __int_as_float(__float_as_int(Q.unk_cost) & __builtin_amdgcn_sbfe((int)fl, 2 * j + 1, 1));
            const bool hit = n2[j] <= gt2, act = !done && j < count;
            cost = act ? c1 : cost;
            d2 = act ? n2[j] : d2;
            reached = reached || (act && hit);
            done = done || (hit && j < count);
          }
        }
        if (k == 1) store_window();
        if ((outcome = interval_end(k, flag))) break;
      }
      if (outcome == 2) {
        MPPI_STAMP(stamp_wg, stamp_base + 30);
        // terminal cost, then the control cost of all T steps (mppi.py:1005-1009)
        if constexpr (F32) {
          cost = cost + (reached ? 0.0f : 1.0f) * __builtin_amdgcn_sqrtf(d2) / (float)Q.v_post_den;
          const float* my_ccf = CC_LDS ? reinterpret_cast<const float*>(cc_lds) + lane
                                       : reinterpret_cast<const float*>(cc_scratch) + (live ? tile_base : (size_t)lane);
          if (!CC_LDS) __threadfence_block();
          for (int t0 = 0; t0 < T; t0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = my_ccf[(size_t)min(t0 + j, T - 1) * 64];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (t0 + j < T) cost = cost + v[j];
          }
        } else {
        const double term = (reached ? 0.0 : 1.0) * sqrt(d2) / Q.v_post_den;
        cost = (float)((double)cost + term);
        if (CC_LDS) {
          // batches of 8 float32-rounded additions, the next batch's LDS reads in flight meanwhile
          // (row-padded array: every read is unconditional, at an immediate offset from one base)
          const double* my_cc = cc_lds + lane;
          double va[8], vb[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) va[j] = my_cc[(size_t)j * 64];
          for (int b = 0; b * 8 < T; ++b) {
            const double* nxt = my_cc + (size_t)min((b + 1) * 8, Tp - 8) * 64;
#pragma unroll
            for (int j = 0; j < 8; ++j) vb[j] = nxt[(size_t)j * 64];
            if (b * 8 + 8 <= T) {
#pragma unroll
              for (int j = 0; j < 8; ++j) cost = (float)((double)cost + va[j]);
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (b * 8 + j < T) cost = (float)((double)cost + va[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) va[j] = vb[j];
          }
        } else {
          __threadfence_block();
          const double* my_cc = cc_scratch + (live ? tile_base : (size_t)lane);
          for (int t0 = 0; t0 < T; t0 += 8) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = my_cc[(size_t)min(t0 + j, T - 1) * 64];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              if (t0 + j < T) cost = (float)((double)cost + v[j]);
          }
        }
        }  // (exact float64 tail)
        MPPI_STAMP(stamp_wg, stamp_base + 31);
        if (live) costs[n] = cost;
        // first half of the control update (update_kernels.h): weights relative to the tile's minimum
        if (tile_ok) emit_tile_weights(cost, live, Q.lambda, n, tile, w_rel, tile_beta);
        MPPI_STAMP(blockIdx.x < 512, 2049 + 2 * blockIdx.x);
      }
    }
    if (outcome == 2) return;  // the assumption held to the end
    exact = true;               // abandoned: every wave arrives here after the same barrier
  }
  // ================================================================== exact schedule (k_rollout_pipe)
  if (wave >= 3) return;
  pipe_tile_body<8, POW2RES, CC_LDS>(P, cells16, noise, u, costs, w_rel, tile_beta, cc_scratch, map_bytes, 1);
}

}  // namespace mppi
