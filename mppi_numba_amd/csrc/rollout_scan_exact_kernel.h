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
// lane = (rollout, 4 consecutive steps), one CHUNK WAVE per 8 steps of a tile of 32 rollouts.  The
// three sums are WALKED by one wave each (x and y in the two halves of one wave): three dependent
// instructions per step (v_fma_f64 / v_add_f64, v_cvt_f32_f64, v_cvt_f64_f32) instead of the ~53
// of the pipelined kernels' state wave.  Same operations on the same operands in the same order as
// the oracle -- with one reassociation: x + dt*vtr*v*cos(theta) is formed as fma(vtr, (dt*v)*cos(theta), x), another
// float64 value whose float32 rounding differs with probability ~1e-9 per step: the same bits in practice (the
// tests allow n / 500 rollouts a few ulps off and find none).
//
// The walks follow each other down the horizon, one group of 8 steps apart, and the chunk waves work
// between them -- no workgroup barrier between entry and the update sums, only flags in LDS
// (a plain store by the producer after its data -- the LDS executes a wave's instructions in order --,
// an acquire poll by the consumer):
//
//   chunk wave g                      walker
//   A  noise (GEN: Philox; else read), clipped controls, dt*w          -> a_done[g]
//                                     theta walk over group g          -> th_done[g]
//   B  sin / cos of the 8 headings, (dt*v)*cos, (dt*v)*sin             -> b_done[g]
//                                     x | y walk over group g          -> xy_done[g]
//   C  lookups, goal distances, sqrt, stage costs; freeze / goal events, the vote; what a rollout that
//      stops here goes on paying -> its (chunk, rollout) slot
//      (after ev_done[g-1]: all earlier events are known)              -> ev_done[g]
//   D' first event wins, per-step addends                              -> c_done[g]
//      last of all: the control-cost products (mppi.py:1007-1009)      -> cc_done[g]
//                                     cost walk over group g (stage, obstacle, unknown per step)
//   (a rollout stopped in a zero-traction cell keeps paying where it stands: ordinary records)
//   ... then, by the cost wave: terminal cost, the T control-cost terms
//   (mppi.py:1005-1009), cost, tile weights; barrier; F: per-tile update sums, lane = step.
//
// Workgroup = 3 walkers + ceil(T / 8) chunk waves: T <= 104.  Waves go to the four SIMDs of a CU
// round robin (wave & 3).  A walk keeps a SIMD's float64 pipe busy for about half of its ~55 cycles
// per step: the theta and the x | y walks share SIMD 0 (waves 0, 4; all three on one SIMD were
// measured at 87 cycles per step, throughput-bound), the cost walk, which runs last, is wave 1.  The
// chunk waves fill the remaining slots -- 2, 3, 4, 4 per SIMD -- and carry falling priorities in the
// order their groups are needed, so that on every SIMD the Philox blocks of the early groups finish
// first (kGroupOfWave); the two on SIMD 0 own the last groups.
// A failed vote: the tile is rolled out sequentially by the cost wave with the arithmetic of
// k_rollout_map<DET, exact> (map_step) -- the same bits again, slowly; the host stops launching this
// kernel on a map where most tiles fail (review_speculation).
#pragma once
#include "rollout_scan_kernel.h"

namespace mppi {

// LDS of one workgroup with W chunk waves over R = 32 rollouts, Tp = 8 W steps:
//   e2   [Tp][R] float2   noise (swizzled columns: phase F reads it lane = step)
//   ccr  [2W][R][4] double control-cost terms
//   grp  [W] x 4 KiB      per group of 8 steps: [8][R] double dt*w, overwritten by (dt*v)*cos once the
//                         theta walk has passed | [8][R] double (dt*v)*sin; both overwritten, once the
//                         position walk has passed, by the group's two records
//                         [R] {double stage[4]; float obstacle[4]; float unknown[4]}
//   p2   [Tp + 1][R] float2 positions (row t = before step t); the float32 headings [Tp + 1][R]
//                         live in its upper half until the positions overwrite them (see th_sh)
//   small: control ratios, terminal data, event words, weights, flags; per (chunk, rollout) what a rollout
//          that stops in that chunk goes on paying: {double stage cost; uint32 where / how}
// s_sleep argument of the flag polls (units of 64 cycles)
#ifndef MPPI_SCAN_POLL_SLEEP
#define MPPI_SCAN_POLL_SLEEP 2
#endif
struct ScanExactLds {
  static constexpr int R = 32, CHL = 4, kMaxChunkWaves = 13;
  // the waves a workgroup needs for W groups (walkers: waves 0, 4, 1; chunk waves: kGroupOfWave in the kernel)
  __host__ __device__ static constexpr int waves(int W) {
    constexpr int need[kMaxChunkWaves] = {6, 6, 6, 10, 10, 10, 14, 14, 14, 15, 16, 16, 16};  // 1 + the highest wave among groups 0 .. W-1 and the walkers
    return need[W - 1];
  }
  __host__ __device__ static constexpr size_t plane(int W) { return (size_t)W * 8 * R * 8; }
  __host__ __device__ static constexpr size_t e2(int W) { return plane(W); }
  __host__ __device__ static constexpr size_t ccr(int W) { return plane(W); }
  __host__ __device__ static constexpr size_t grp(int W) { return 2 * plane(W); }
  __host__ __device__ static constexpr size_t p2(int W) { return (size_t)(W * 8 + 1) * R * 8; }
  __host__ __device__ static constexpr size_t small(int W) {
    return (size_t)W * 8 * (16 + 8) + R * 64 + 64 + 8 * (kMaxFoldedRanks + 2) + 8 * 16 * 4 + (size_t)2 * W * R * 16;
  }
  __host__ __device__ static constexpr size_t total(int W) { return e2(W) + ccr(W) + grp(W) + p2(W) + small(W); }
};

// ---- a tile whose vote failed, re-executed on the exact three-wave schedule of k_rollout_pipe ---------------
// (rollout_kernels.h, pipe_tile_body: the same operations on the same operands, i.e. the bits of k_rollout_map and
// of the oracle; mppi.py:916-1009).  Round 3 rolled such a tile out with ONE wave, global cell lookups and a full
// sincos per step: ~300 us per launch on a map where tiles fail, against ~29 us for k_rollout_pipe -- a cliff the
// planner only left when half of its tiles failed.  Now: the 16-bit map window goes into the LDS the walks no
// longer need (all waves copy), then producer (wave 2: clipped controls from the noise in LDS) -> state (wave 0)
// -> cost (wave 1) one chunk of 8 steps apart, one workgroup barrier per chunk; the waves without a role only
// keep the barrier count.  Lanes 32..63 mirror lanes 0..31 (a tile is 32 rollouts).  Returns the cost after the
// last step and the terminal cost in the lanes of wave 1; the control-cost terms follow in the caller as for any
// other tile.
struct ScanFallback {
  int offset;     // bytes from the start of the dynamic LDS: [Tp] float2 controls | window | ring; < 0: no room,
  int map_bytes;  // ... the tile is rolled out step by step by one wave (global cells)
  // Round 5: a map on which the planner has stopped speculating (review_speculation) -- traction changes along the
  // paths: the reference's own semantic maps, CVaR-bin maps -- used to leave this kernel for k_rollout_pipe +
  // k_update_rows (two launches, noise through memory).  direct != 0: the launch skips the time-parallel attempt and
  // runs every tile on the exact three-wave schedule at once: the noise still comes from the Philox counters into LDS,
  // the update of the previous launch is still folded in front (combine_step / publish / collect) and the tile's update
  // sums still leave as a packet -- the iteration stays ONE launch and the kernel family (and with it the peer
  // exchange of a multi-GPU run) does not change with the map.  The window copy is requested at kernel entry (LDS
  // nobody else uses in this mode), by the cost wave, straight into LDS.
  int direct;
  // the host has proved |dt * w * traction| <= 0.36 rad for every step (launch_rollout_det's rot_ok): the state role
  // rotates (cos, sin) by the exact heading increment without looking at it.  (With the test inside the loop the
  // compiler lays the full evaluation out in line and the rotation behind a taken branch: 416 instead of 265 cycles
  // per step -- profiles/r05_pipe_notes.md)
  int rot_ok;
  // direct mode: the walks' groups and positions do not exist; the region sits right behind the control-cost terms and
  // the small arrays (control ratios, weights, flags, u_sh ...) behind it, at this offset
  int small_offset;
  static constexpr int kRingBytes = PipeRing<8>::kBytesPerPair;
  __host__ __device__ static constexpr size_t bytes(int Tp, int map_bytes) { return (size_t)Tp * 8 + (size_t)map_bytes + kRingBytes; }
};

// us_ready (direct mode): the controls of the launch, [Tp] float2 in LDS already (u_sh), the window copied and a
// workgroup barrier behind both -- nothing is staged and the schedule starts at once.  background(): what the waves
// without a role still owe (direct mode: the control-cost terms of their groups), run behind the first barrier of the
// schedule, where they would only wait; the producer wave runs it after its last chunk.
template <bool POW2RES, typename Background>
__device__ __forceinline__ float scan_exact_reexecute(const DevParams& P, const uint16_t* __restrict__ cells16, char* fb,
                                                      int map_bytes, const float2* e2, const float2* u_src, int T,
                                                      int c, int lane, const float2* us_ready, bool rot_ok,
                                                      Background&& background, bool leave_when_idle = false) {
  constexpr int C = 8, R = ScanExactLds::R;
  using Ring = PipeRing<C>;
  const int Tp = (T + 7) & ~7;
  float2* us_stage = reinterpret_cast<float2*>(fb);
  uint16_t* lds_map = reinterpret_cast<uint16_t*>(fb + (size_t)Tp * 8);
  char* ring_base = reinterpret_cast<char*>(lds_map) + map_bytes;
  float2* ring_xy = reinterpret_cast<float2*>(ring_base);
  double2* ring_qd = reinterpret_cast<double2*>(ring_xy + 2 * Ring::kHalf);
  uint16_t* ring_cell = reinterpret_cast<uint16_t*>(ring_qd + 2 * Ring::kHalf);
  const int r = lane & (R - 1);
  const int K = (T + C - 1) / C;
  [[maybe_unused]] const bool stamp_wg = blockIdx.x == 5 && c < 3;  // (stamps build: slots 11 .. 15 of the three roles' rows)
  [[maybe_unused]] const int stamp_base = 64 + 16 * c;
  MPPI_STAMP(stamp_wg, stamp_base + 11);
  const float2* us = us_ready;
  if (us_ready == nullptr) {
    // the window and the controls: everybody; the walks' data under them is dead
    copy_window_to_lds(P, cells16, lds_map, 0, (int)blockDim.x);
    for (int t = threadIdx.x; t < Tp; t += blockDim.x) us_stage[t] = t < T ? u_src[t] : make_float2(0.0f, 0.0f);
    __syncthreads();
    us = us_stage;
  }
  if (c < 3) __builtin_amdgcn_s_setprio(3);  // (the waves without a role may share a SIMD with one that has)
  float cost = 0.0f;
  const double dt64 = (double)P.dt;
  // Direct mode with enough waves: the FIRST chunk's controls come from eight of the waves without a role, one step
  // each, side by side -- the producer alone needs ~1.3k cycles for a chunk (a cold loop body, eight 16-byte stores),
  // and the state wave cannot take its first step before that chunk is there.
  const bool first_chunk_by_helpers = us_ready != nullptr && (int)(blockDim.x >> 6) >= 3 + C;
  if (first_chunk_by_helpers && c >= 3 && c < 3 + C) {
    const int t = c - 3;  // (< Tp; past a very short horizon the noise rows hold zeros)
    const float2 ut = us[t], e = e2[t * R + (r ^ (t & (R - 1)))];
    ring_qd[t * 64 + lane] = make_double2(dt64 * (double)clip_f32(ut.x + e.x, P.v_lo, P.v_hi),
                                          dt64 * (double)clip_f32(ut.y + e.y, P.w_lo, P.w_hi));
  }
  if (c == 2) {  // ---- producer: {dt * v, dt * w} of chunk k + 1 while the state wave integrates chunk k
    auto produce = [&](int chunk) {
      double2* out_qd = ring_qd + (size_t)(chunk & 1) * Ring::kHalf;
#pragma unroll
      for (int j = 0; j < C; ++j) {
        const int t = chunk * C + j;  // (< Tp: the noise rows past the horizon hold zeros)
        const float2 ut = us[t], e = e2[t * R + (r ^ (t & (R - 1)))];
        out_qd[j * 64 + lane] = make_double2(dt64 * (double)clip_f32(ut.x + e.x, P.v_lo, P.v_hi),
                                             dt64 * (double)clip_f32(ut.y + e.y, P.w_lo, P.w_hi));
      }
    };
    MPPI_STAMP(stamp_wg, stamp_base + 12);
    if (!first_chunk_by_helpers) produce(0);
    __syncthreads();
    // (this wave reaches every barrier ~1k cycles before the state wave: what it still owes -- direct mode: the control-
    //  cost terms of its group, which the cost wave waits for in front of its last walk -- fits into one of those gaps)
    const int bg_at = min(2, K);
    for (int k = 0; k <= K; ++k) {
      MPPI_STAMP(stamp_wg && k == 5, stamp_base + 13);  // (stamps build: chunk 5 -- released, work done)
      if (k + 1 < K) produce(k + 1);
      if (k == bg_at) background();
      MPPI_STAMP(stamp_wg && k == 5, stamp_base + 14);
      __syncthreads();
    }
    MPPI_STAMP(stamp_wg, stamp_base + 15);
    __builtin_amdgcn_s_setprio(0);
  } else if (c == 0) {  // ---- state: pipe_tile_body's role 0 (rollout_kernels.h, pipe_state_chunk)
    const PipeWindow<POW2RES> win(P, lds_map);
    PipeState st = pipe_state_init(P);
    __syncthreads();
    st.cell = win.lookup(st.x, st.y);
    MPPI_STAMP(stamp_wg, stamp_base + 12);
    auto run = [&](auto check) {
      for (int k = 0; k <= K; ++k) {
        MPPI_STAMP(stamp_wg && k == 5, stamp_base + 13);
        if (k < K) {
          constexpr bool CHK = decltype(check)::value != 0;
          const double2* in_qd = ring_qd + (size_t)(k & 1) * Ring::kHalf;
          float2* out_xy = ring_xy + (size_t)(k & 1) * Ring::kHalf;
          uint16_t* out_cell = ring_cell + (size_t)(k & 1) * Ring::kHalf;
          if (T - k * C >= C) pipe_state_chunk<C, POW2RES, CHK>(P, win, st, in_qd, out_xy, out_cell, lane);
          else pipe_state_tail<POW2RES, CHK>(P, win, st, in_qd, out_xy, out_cell, lane, T - k * C);
        }
        MPPI_STAMP(stamp_wg && k == 5, stamp_base + 14);
        __syncthreads();
      }
    };
    if (rot_ok) run(PhaseTag<0>());
    else run(PhaseTag<1>());
    MPPI_STAMP(stamp_wg, stamp_base + 15);
    __builtin_amdgcn_s_setprio(0);
  } else if (c == 1) {  // ---- cost: pipe_tile_body's role 1
    const double gt2 = (double)P.gt2;
    double d2 = 1e9;
    bool done = false, reached = false;
    __syncthreads();
    MPPI_STAMP(stamp_wg, stamp_base + 12);
    for (int k = 0; k <= K; ++k) {
      MPPI_STAMP(stamp_wg && k == 5, stamp_base + 13);
      if (k >= 1) {
        const int t0 = (k - 1) * C;
        const float2* in_xy = ring_xy + (size_t)((k - 1) & 1) * Ring::kHalf;
        const uint16_t* in_cell = ring_cell + (size_t)((k - 1) & 1) * Ring::kHalf;
        const int count = min(C, T - t0);
        for (int j = 0; j < count; ++j) {
          const float2 xy = in_xy[j * 64 + lane];
          const uint32_t fl = (uint32_t)in_cell[j * 64 + lane] >> 14;
          const double dx = (double)(P.xg - xy.x), dy = (double)(P.yg - xy.y);
          const double nd2 = fma(dx, dx, dy * dy);
          float c1 = (float)((double)cost + fma(P.dist_weight, sqrt_newton_f64(nd2), dt64));
          c1 = c1 + ((fl & 1u) ? P.obs_cost : 0.0f);
          c1 = c1 + ((fl & 2u) ? P.unk_cost : 0.0f);
          const bool hit = nd2 <= gt2, act = !done;
          cost = act ? c1 : cost;
          d2 = act ? nd2 : d2;
          reached = reached || (act && hit);
          done = done || hit;
        }
      }
      MPPI_STAMP(stamp_wg && k == 5, stamp_base + 14);
      __syncthreads();
    }
    MPPI_STAMP(stamp_wg, stamp_base + 15);
    cost = (float)((double)cost + (reached ? 0.0 : 1.0) * sqrt(d2) / P.v_post_den);
  } else {
    __syncthreads();
    background();
    // (direct mode: a wave with nothing left to do in this launch -- waves 4.. : phase F belongs to waves 2 and 3 --
    //  leaves the kernel here instead of keeping the schedule's barrier count: a barrier waits for the waves that are
    //  still alive, and with 16 of them a release took ~280 cycles after the last arrival against ~80 with 4: 200 cycles
    //  in every chunk of the state wave's critical path)
    if (leave_when_idle && c >= 4) __builtin_amdgcn_endpgm();
    for (int k = 0; k <= K; ++k) __syncthreads();
  }
  return cost;
}

// SPEED (round 6): rollout_det_dyn_w_speed_map_numba (mppi.py:1013-1111) on the time-parallel schedule.  That mode's
// dynamics run on NOMINAL traction (terrain.py:455-463: all mass in the last bin, the padding ring in the first), i.e.
// the assumption the walks rest on -- every visited cell carries the start cell's traction -- holds by construction;
// what the map changes from cell to cell is the TIME a step is charged with, dt / (risk speed + 1e-6)
// (mppi.py:1095-1096), and that is a term of the stage cost the chunk waves form side by side anyway (phase C: the
// risk byte comes with the cell: 32-bit cells, scan_lookup<.., WIDE>; one float64 division per step and lane, off
// every chain).  A vote that fails all the same (grids injected through the C API) is re-run by the cost wave alone on
// the general arithmetic (map_step<MAP_SPEED>: global cells + the risk map), counted, and the planner then leaves for
// k_rollout_fused<SPEED>; there is no DIRECT form of this mode (a window of 32-bit cells does not fit beside the noise).
template <bool POW2RES, bool GEN, bool DIRECT = false, bool SPEED = false>
__global__ __launch_bounds__(1024) void k_rollout_scan_exact(DevParams P, const uint16_t* __restrict__ cells16,
                                                             const uint32_t* __restrict__ cells,
                                                             const float2* __restrict__ noise, NoiseJob gen,
                                                             const float2* __restrict__ u, float* __restrict__ costs,
                                                             float* __restrict__ w_rel, ScanPackets pk,
                                                             PendingApply pend, ScanFallback fallback,
                                                             const int8_t* __restrict__ risk) {
  static_assert(!(DIRECT && SPEED), "the speed-map mode has no direct (exact-schedule) form of this kernel");
  extern __shared__ double2 scan_lds[];
  using L = ScanExactLds;
  constexpr int R = L::R, CHL = L::CHL;
  const int c = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  const int r = lane & (R - 1), h = lane / R;
  const int W = (P.n_steps + 7) >> 3;  // groups of 8 steps = chunk waves
  // role of this wave: walker 0 (theta), 1 (x | y), 2 (cost); or the chunk wave of group g; or none.
  // groups in the order their increments are needed x the order in which a SIMD's chunk waves finish
  // their Philox blocks: on SIMDs 1..3 first finishers waves 5, 2, 3; second 9, 6, 7; third 13, 10, 11;
  // fourth 14, 15.  The two chunk waves on SIMD 0 (8, 12) take the LAST groups: their Philox blocks run
  // while the walks still wait for input, their lookups when the theta walk is through.
  //                                   wave:   0   1  2  3   4  5  6  7   8  9 10 11  12 13 14  15
  constexpr int kGroupOfWave[16] =          {-1, -1, 1, 2, -1, 0, 4, 5, 11, 3, 7, 8, 12, 6, 9, 10};
  constexpr int kPrioOfWave[16] =           { 3,  3, 3, 3,  3, 3, 2, 2,  3, 2, 1, 1,  3, 1, 0,  0};
  // (one walker per SIMD -- waves 0, 1, 2 -- measured the same: profiles/r03_scan_notes.md)
  const int walker = c == 0 ? 0 : (c == 4 ? 1 : (c == 1 ? 2 : -1));
  int g = -1, prio = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    g = c == i ? kGroupOfWave[i] : g;
    prio = c == i ? kPrioOfWave[i] : prio;
  }
  g = g < W ? g : -1;
  const int k = 2 * g + h;  // this lane's chunk of 4 steps
  [[maybe_unused]] const bool stamp_wg = blockIdx.x == 5 || blockIdx.x == 200;  // (a workgroup that combines a step of a folded update, one that does not)
  [[maybe_unused]] const int stamp_base = (blockIdx.x == 200 ? 1024 : 64) + 16 * c;
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
  // One GPU, an update left to this launch (update_kernels.h, PendingApply::reduce_tiles): step tt of it is combined
  // by workgroup tt -- by its theta walker; with fewer workgroups than steps the x | y walkers take the next
  // gridDim.x steps, and so on round.  The tile packets of this wave's first step are requested before anything
  // else: they come from the other XCDs' writes of the previous launch, a trip of ~1.5 us.
  const bool folded = pend.packets != nullptr;
  const bool reducing = folded && pend.reduce_tiles != nullptr;
  const int red_t = tile + max(walker, 0) * (int)gridDim.x;
  const bool reduce_here = reducing && (walker == 0 || walker == 1) && red_t < T;
  StepLoads red_loads;
  float2 red_u = make_float2(0.0f, 0.0f);
  if (reduce_here) {
    red_loads = combine_step_issue(pend.reduce_tiles, pend.reduce_n_tiles, tile_packet_floats(T), red_t, lane);
    red_u = uq[red_t];
  }

  // the start cell (the traction every visited cell is assumed to carry): requested at once -- the theta walk needs
  // it for its very first step
  const uint32_t ref = scan_lookup<POW2RES, SPEED>(Q, cells16, Q.x0, Q.y0) & 0x3fffu;

  char* base = reinterpret_cast<char*>(scan_lds);
  // no time-parallel attempt (see ScanFallback): an instantiation of its own -- the speculative launch must not carry
  // a single instruction of this (its schedule is a local optimum the compiler leaves at the slightest change:
  // profiles/r04_scan_notes.md section 6; a run-time flag cost the C2 launch 1.2 us)
  constexpr bool direct = DIRECT;
  // the 16-bit map window: waves 5.. their shares, all vectors in flight, waited for before the barrier in front of
  // the schedule (not the walkers of a folded launch, waves 0 and 4: a wait for "all vector memory operations" would
  // also wait for the stores of the step they publish, ~4k cycles)
  if (direct && c >= 5)
    copy_window_to_lds_direct(Q, cells16, reinterpret_cast<uint16_t*>(base + fallback.offset + (size_t)Tp * 8), 320, (int)blockDim.x - 320);
  float2* e2 = reinterpret_cast<float2*>(base);                                  // [Tp][R] (swizzled)
  double* ccr = reinterpret_cast<double*>(base + L::e2(W));                      // [2W][R][CHL]
  char* grp = base + L::e2(W) + L::ccr(W);                                       // [W] x 4 KiB
  float2* pos = reinterpret_cast<float2*>(grp + L::grp(W));                      // [Tp + 1][R]
  // The headings live in the UPPER half of the position rows.  Position row p covers heading rows
  // 2p - Tp and 2p - Tp + 1: when the position walk stores row t + 1 (step t) it overwrites headings
  // of steps <= t, which the chunk waves up to step t's have consumed (b_done); and a heading row is
  // written (the theta walk runs ahead) into a position row of a LATER step, not yet written.
  float* th_sh = reinterpret_cast<float*>(pos) + (size_t)Tp * R;                 // [Tp + 1][R]
  char* small = DIRECT ? base + fallback.small_offset : reinterpret_cast<char*>(pos) + L::p2(W);
  double2* uos = reinterpret_cast<double2*>(small);                              // [Tp] u / std^2
  double* term_sh = reinterpret_cast<double*>(uos + Tp);                         // [R]
  uint32_t* evw = reinterpret_cast<uint32_t*>(term_sh + R);                      // [R][2]
  float* wsh = reinterpret_cast<float*>(evw + 2 * R);                            // [R]
  uint32_t* flags = reinterpret_cast<uint32_t*>(wsh + R);                        // [0] a failed vote
  // a sharded iteration's update applied here (update_kernels.h, PendingApply): the updated sequence
  double* scale_sh = reinterpret_cast<double*>(small + (size_t)Tp * 16 + R * 64 + 64);  // [kMaxFoldedRanks + 2]
  float2* u_sh = reinterpret_cast<float2*>(scale_sh + kMaxFoldedRanks + 2);              // [Tp]
  // hand-over flags, one per group of 8 steps: raised by the wave that has stored the group's data
  int* a_done = reinterpret_cast<int*>(u_sh + Tp);  // [16] heading increments (chunk wave)
  int* th_done = a_done + 16;                       // [16] headings (theta walk)
  int* b_done = th_done + 16;                       // [16] position increments (chunk wave)
  int* xy_done = b_done + 16;                       // [16] positions (position walk)
  int* ev_done = xy_done + 16;                      // [16] freeze / goal events of the group and of all before it (chunk wave)
  int* c_done = ev_done + 16;                       // [16] records (chunk wave): 1, or 3 = a penalty in the group
  int* cc_done = c_done + 16;                       // [16] control-cost terms (chunk wave)
  int* u_ready = cc_done + 16;                      // [0] the updated control sequence is in u_sh (wave 0; launches that apply all-gathered packets: one that combines tile packets hands u_sh over at the workgroup's barrier)
  double2* stop_sh = reinterpret_cast<double2*>(u_ready + 16);  // [2W][R] {stage cost of a rollout stopped in this chunk; bits}
  if (c == 4) {
    if (lane < R) {
      evw[2 * lane] = 0u;
      evw[2 * lane + 1] = 0u;
      if (lane == 0) flags[0] = 0u;
    }
    for (int i = lane; i < 8 * 16; i += 64) a_done[i] = 0;
  }
  // The workgroup's one barrier before the flags are used.  In a launch that combines the previous launch's tile
  // packets (reducing) every wave takes it where it would otherwise wait for the updated controls -- the chunk waves
  // behind their Philox blocks, wave 0 when it has collected the sequence -- so that nobody stands at a barrier while
  // the packets are on their way, and the barrier's release is the hand-over of u_sh.
  // (direct mode, reducing: this barrier is also the one in front of the exact schedule -- a wave that requested a
  //  share of the map window waits for it here, where it waits anyway)
  if (!reducing || c == 1 || (walker < 0 && g < 0)) {  // (every wave has only just started; the cost walker and a wave without a group have nothing else to do)
    if (direct && reducing && c >= 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  }
  MPPI_STAMP(stamp_wg && c < 16, stamp_base + 9);
  // Everything a flag guards is in LDS, and the LDS executes one wave's instructions in the order they
  // were issued: a flag written after the data IS after the data for every other wave.  No s_waitcnt
  // before the flag store (a release would drain the wave's outstanding LDS stores first: ~100 cycles
  // between two groups of a walk); the compiler just must not move it.
  auto raise = [&](int* flag, int value) {
    pin_memory_order();
    if (lane == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    pin_memory_order();
  };
  auto peek = [&](const int* flag) {  // wave-uniform
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
  };
  // Every wave raises every flag of its role unconditionally and in order -- no flag depends on data --
  // and all waves of a workgroup are resident: the waits cannot deadlock.  A build with -DMPPI_POLL_BOUND
  // (make -C csrc bounded) makes a wave that has polled for ~50 ms trap instead of hanging the device:
  // for work on this file (a wrong group index in a wait cost 30 GPU-minutes once); measured at 0.35 us
  // per launch (13.7 vs 13.4), so not the default.
#ifdef MPPI_POLL_BOUND
  constexpr int kMaxPolls = 1 << 19;
#endif
  auto wait_for = [&](const int* flag) {
    int v;
    [[maybe_unused]] int polls = 0;
    while ((v = peek(flag)) == 0) {
      __builtin_amdgcn_s_sleep(MPPI_SCAN_POLL_SLEEP);
#ifdef MPPI_POLL_BOUND
      if (++polls > kMaxPolls) __builtin_trap();
#endif
    }
    return v;
  };
  const double dt64 = (double)Q.dt;

  if (folded && (walker == 0 || walker == 1)) {
    __builtin_amdgcn_s_setprio(3);
    // The update this launch owes (update_kernels.h, PendingApply), while the chunk waves draw their Philox
    // blocks; the chunk waves pick the new sequence up from LDS behind u_ready or the barrier.  (Here, next to the entry code, not
    // inside the walkers' branch: a cold instruction fetch is ~1k cycles for a lone wave.)
    if (reducing) {
      // One GPU: this wave's step(s) of the update, exactly as block t of k_combine_tiles forms them, published to
      // all workgroups; every workgroup's theta walker then collects the whole sequence.
      MPPI_STAMP(stamp_wg, stamp_base + 5);
      if (reduce_here) {
        const int stride = tile_packet_floats(T);
        publish_step(pend, combine_step_finish(red_loads, pend.reduce_tiles, pend.reduce_n_tiles, stride, red_t, pend.lambda, lane),
                     red_u, red_t, T, lane);
        for (int tt = red_t + 2 * (int)gridDim.x; tt < T; tt += 2 * (int)gridDim.x)
          publish_step(pend, combine_step(pend.reduce_tiles, pend.reduce_n_tiles, stride, tt, pend.lambda, lane), uq[tt], tt, T, lane);
      }
      MPPI_STAMP(stamp_wg, stamp_base + 6);
      if (walker == 0) collect_published(pend, T, Tp, lane, u_sh);
      MPPI_STAMP(stamp_wg, stamp_base + 7);
      lds_barrier();  // (the workgroup's barrier: u_sh is there for everybody behind it)
    } else if (walker == 0) {
      // Several GPUs: the packets were all-gathered before the launch.  k_apply's expressions in k_apply's
      // order: the same bits in every workgroup and on every rank.
      pending_apply_prepare(pend, lane, scale_sh);
      pin_memory_order();
      for (int tt = lane; tt < Tp; tt += 64) {
        const float2 v = tt < T ? pending_apply_control(pend, scale_sh, uq, tt) : make_float2(0.0f, 0.0f);
        u_sh[tt] = v;
        if (tile == 0 && tt < T) {
          pend.u_out[tt] = v;
          pend.u_prev[tt] = v;
        }
      }
      if (tile == 0 && lane == 0) {
        pend.stats[0] = scale_sh[kMaxFoldedRanks + 1];
        pend.stats[1] = scale_sh[kMaxFoldedRanks];
      }
    }
    if (walker == 0 && !reducing) raise(&u_ready[0], 1);
  }

  // Groups handed to a walker.  Two register sets: while one group is walked the next one's operands
  // are already on their way, and the flag of the one after is read WITHOUT waiting (the answer is
  // looked at after the walk): neither the flag's nor the operands' LDS latency sits between two groups
  // unless the producers are behind.
  //   fetch(set, group)               loads the group's operands into register set 0 / 1
  //   run(set, group, flag value)     walks its 8 steps and publishes them
  auto glance = [&](const int* flag) {  // issues the read; settle() looks at it
    return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  auto settle = [&](int raw) {
    const int v = __builtin_amdgcn_readfirstlane(raw);
    pin_memory_order();  // (reads issued after the flag's value is here are served after it: in-order LDS)
    return v;
  };
  auto walk_groups = [&](const int* in_flags, auto&& fetch, auto&& run) {
    int v0 = wait_for(&in_flags[0]);
    fetch(PhaseTag<0>(), 0);
    int v1 = W > 1 ? peek(&in_flags[1]) : 0;
    if (v1) fetch(PhaseTag<1>(), 1);
    for (int gi = 0; gi < W; gi += 2) {
      // set 0 holds group gi; set 1 holds group gi + 1 if v1
      const int raw2 = gi + 2 < W ? glance(&in_flags[gi + 2]) : 0;
      run(PhaseTag<0>(), gi, v0);
      if (gi + 1 >= W) break;
      if (!v1) {
        v1 = wait_for(&in_flags[gi + 1]);
        fetch(PhaseTag<1>(), gi + 1);
      }
      int v2 = settle(raw2);
      if (v2) fetch(PhaseTag<0>(), gi + 2);
      const int raw3 = gi + 3 < W ? glance(&in_flags[gi + 3]) : 0;
      run(PhaseTag<1>(), gi + 1, v1);
      if (gi + 2 >= W) break;
      if (!v2) {
        v2 = wait_for(&in_flags[gi + 2]);
        fetch(PhaseTag<0>(), gi + 2);
      }
      v0 = v2;
      v1 = settle(raw3);
      if (v1) fetch(PhaseTag<1>(), gi + 3);
    }
  };

  // what every tile ends with, by its cost wave (wave 1): the control cost of all T steps after the terminal cost, the
  // cost, the weights relative to the tile's minimum and the header of the tile's packet
  auto finish_tile = [&](float cost) {
    // the control cost of all T steps, also after an early goal break (mppi.py:1007-1009); steps past
    // the horizon hold zero noise: their terms are +0.0
    for ([[maybe_unused]] int polls = 0;
         !__all(lane >= W || __hip_atomic_load(&cc_done[lane & 15], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0);) {
      __builtin_amdgcn_s_sleep(MPPI_SCAN_POLL_SLEEP);
#ifdef MPPI_POLL_BOUND
      if (++polls > kMaxPolls) __builtin_trap();
#endif
    }
    {
      const double* at = ccr + (size_t)r * CHL;
      double2 ca[2][4];
      auto load = [&](auto set, int gi) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const double* in = at + (size_t)(2 * gi + hh) * R * CHL;
          ca[decltype(set)::value][2 * hh] = reinterpret_cast<const double2*>(in)[0];
          ca[decltype(set)::value][2 * hh + 1] = reinterpret_cast<const double2*>(in)[1];
        }
      };
      auto add = [&](auto set) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          cost = (float)((double)cost + ca[decltype(set)::value][i].x);
          cost = (float)((double)cost + ca[decltype(set)::value][i].y);
        }
      };
      load(PhaseTag<0>(), 0);
      for (int gi = 0; gi < W; gi += 2) {
        if (gi + 1 < W) load(PhaseTag<1>(), gi + 1);
        add(PhaseTag<0>());
        if (gi + 1 >= W) break;
        if (gi + 2 < W) load(PhaseTag<0>(), gi + 2);
        add(PhaseTag<1>());
      }
    }
    MPPI_STAMP(stamp_wg, stamp_base + 3);
    const bool mine = live && lane < R;
    if (mine) costs[n] = cost;
    // first half of the control update (update_kernels.h): weights relative to the tile's minimum.
    // exp(-(c - beta)/lambda) = 2^(n + f): the fraction through v_exp_f32 (relative error ~1e-7 whatever
    // the argument), the integer through the exponent (as k_rollout_scan; a float64 exp is ~1k cycles of
    // this wave's serial tail, and u is held to 1e-5 of the range, not to bits)
    const float beta = wave_min_f32(live ? cost : __builtin_inff());
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
    __builtin_amdgcn_s_setprio(0);
    MPPI_STAMP(stamp_wg, stamp_base + 4);
  };

  // a chunk wave's noise (registers, from its Philox blocks to its control-cost terms) and the control-cost terms of
  // its steps: lambda * (u0/s0^2 * e0 + u1/s1^2 * e1) in float64 (mppi.py:1007-1009), needed after the terminal cost only
  float2 e[CHL];
  auto chunk_ccr = [&]() {
    if (lane < 8) {  // the control ratios of the 8 steps (float64 quotients: mppi.py:709)
      const float2 ul = folded ? u_sh[8 * g + lane] : uq[min(8 * g + lane, T - 1)];
      uos[8 * g + lane] = make_double2((double)ul.x / Q.s0sq, (double)ul.y / Q.s1sq);
    }
#pragma unroll
    for (int j = 0; j < CHL; ++j) ccr[((size_t)k * R + r) * CHL + j] = control_cost(Q, uos[min(t0 + j, Tp - 1)], e[j]);
    raise(&cc_done[g], 1);
  };

  if (direct && walker >= 0) {
    // (direct mode: nothing to walk; the three roles of the exact schedule start behind the barrier below)
  } else if (walker == 0 || walker == 1) {
    // ================================================================ the theta walk (wave 0), the x | y walk (wave 4)
    // one running sum rounded to float32 after every fma; a lone wave issues an instruction per ~5
    // cycles: what counts is the instruction count -- one pointer per group, immediate offsets (the
    // row stride is a compile-time constant of each of the two instances below)
    __builtin_amdgcn_s_setprio(3);
    auto walk = [&](auto stride_tag, double coeff, float start, float* out, size_t half, const int* in_flags,
                    int* out_flags) {
      constexpr int OS = decltype(stride_tag)::value;  // floats per row
      float vf = start;
      double v64 = (double)vf;
      out[0] = vf;  // row 0: the value before step 0
      double inc[2][8];
      walk_groups(
          in_flags,
          [&](auto set, int gi) {
            const double* at = reinterpret_cast<const double*>(grp + (size_t)gi * 4096 + half) + r;
#pragma unroll
            for (int q = 0; q < 8; ++q) inc[decltype(set)::value][q] = at[(size_t)q * R];
          },
          [&](auto set, int gi, int) {
            float* to = out + (size_t)(8 * gi + 1) * OS;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              vf = (float)fma(coeff, inc[decltype(set)::value][q], v64);
              v64 = (double)vf;
              to[q * OS] = vf;  // row t + 1: the value after step t
            }
            raise(&out_flags[gi], 1);
          });
    };
    if (walker == 0) {
      // theta: rows of R floats (the upper lanes mirror the lower ones: the same value to the same address)
      walk(PhaseTag<R>(), fma(Q.ang_ratio, (double)(int)((ref >> 7) & 127u), Q.ang_lo), Q.th0, th_sh + r, 0, a_done, th_done);
    } else {
      // positions: rows of R float2, x in lanes 0..31, y in lanes 32..63 (the group's second half of increments)
      walk(PhaseTag<2 * R>(), fma(Q.lin_ratio, (double)(int)(ref & 127u), Q.lin_lo), h == 0 ? Q.x0 : Q.y0,
           reinterpret_cast<float*>(pos) + 2 * r + h, (size_t)h * (8 * R * 8), b_done, xy_done);
    }
    __builtin_amdgcn_s_setprio(0);
    MPPI_STAMP(stamp_wg, stamp_base + 1);
  } else if (walker == 2) {
    // ================================================================ the cost walk (wave 1)
    __builtin_amdgcn_s_setprio(3);
    float cost = 0.0f;
    double2 sg[2][4];  // [set][chunk of the group * 2 + half]
    float4 fo[2][2], fu[2][2];
    // terminal cost of a rollout that met no event: from the position after the last step.  Computed
    // while this wave waits for the last group's records (the position walk is several groups ahead).
    double term_plain = 0.0;
    bool term_ready = false;
    auto plain_terminal = [&]() {
      const float2 pf = pos[(size_t)T * R + r];
      const double dx = (double)(Q.xg - pf.x), dy = (double)(Q.yg - pf.y);
      term_plain = sqrt(fma(dx, dx, dy * dy)) / Q.v_post_den;  // mppi.py:26-28, 1005
      term_ready = true;
    };
    walk_groups(
        c_done,
        [&](auto set, int gi) {
          constexpr int S = decltype(set)::value;
          const char* in = grp + (size_t)gi * 4096 + (size_t)r * 64;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            sg[S][2 * hh] = reinterpret_cast<const double2*>(in + hh * 2048)[0];
            sg[S][2 * hh + 1] = reinterpret_cast<const double2*>(in + hh * 2048)[1];
            fo[S][hh] = reinterpret_cast<const float4*>(in + hh * 2048 + 32)[0];
            fu[S][hh] = reinterpret_cast<const float4*>(in + hh * 2048 + 32)[1];
          }
        },
        [&](auto set, int gi, int flag_value) {
          constexpr int S = decltype(set)::value;
          // (a group that met no obstacle or unknown cell adds +0.0f twice per step, which leaves a
          //  cost >= +0 as it is: its walk is the three instructions of the float64 add alone)
          if (flag_value & 2) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const double2 s01 = sg[S][2 * hh], s23 = sg[S][2 * hh + 1];
              const float4 o = fo[S][hh], q = fu[S][hh];
              cost = (float)((double)cost + s01.x); cost = cost + o.x; cost = cost + q.x;  // mppi.py:994, 997, 998
              cost = (float)((double)cost + s01.y); cost = cost + o.y; cost = cost + q.y;
              cost = (float)((double)cost + s23.x); cost = cost + o.z; cost = cost + q.z;
              cost = (float)((double)cost + s23.y); cost = cost + o.w; cost = cost + q.w;
            }
          } else {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const double2 s01 = sg[S][2 * hh], s23 = sg[S][2 * hh + 1];
              cost = (float)((double)cost + s01.x);
              cost = (float)((double)cost + s01.y);
              cost = (float)((double)cost + s23.x);
              cost = (float)((double)cost + s23.y);
            }
          }
          // (stamps build: when the walk left groups 0, 6, 9 .. 12)
          MPPI_STAMP(stamp_wg && (gi == 0 || gi == 6), stamp_base + (gi == 0 ? 5 : 6));
          MPPI_STAMP(stamp_wg && gi >= 9 && gi <= 12, stamp_base + (gi == 9 ? 7 : gi));
          // (between two groups, as soon as the position walk is through: this wave mostly waits for records)
          if (!term_ready && peek(&xy_done[W - 1])) plain_terminal();
        });
    if (!term_ready) {
      (void)wait_for(&xy_done[W - 1]);
      plain_terminal();
    }
    MPPI_STAMP(stamp_wg, stamp_base + 1);
    const bool failed = flags[0] != 0u;  // (every chunk wave has published its vote before its c_done)
    if (!failed) {
      MPPI_STAMP(stamp_wg, stamp_base + 2);
      const bool had_event = (evw[2 * r] | evw[2 * r + 1]) != 0u;
      cost = (float)((double)cost + (had_event ? term_sh[r] : term_plain));
      finish_tile(cost);
    } else if (fallback.offset < 0) {
      // ---- no room for the map window: the tile step by step, with the tractions of the visited cells, by this
      //      wave alone (k_rollout_map's arithmetic; slow, and counted: the host stops launching this kernel on a
      //      map where that happens -- review_speculation)
      if (lane == 0 && Q.spec_failures) {
        atomicAdd_system(Q.spec_failures, 1u);
        __threadfence_system();
      }
      RolloutState st = {Q.x0, Q.y0, Q.th0, 0.0f, 1e9, false, false};
      for (int t = 0; t < T; ++t) {
        map_step<SPEED ? MAP_SPEED : MAP_DET, true, false, false>(Q, cells, risk, nullptr, folded ? u_sh[t] : uq[t],
                                                                  e2[t * R + (r ^ (t & (R - 1)))], st);
        if (__all(st.done)) break;
      }
      cost = (float)((double)st.cost + (st.reached ? 0.0 : 1.0) * sqrt(st.d2) / Q.v_post_den);
      finish_tile(cost);
    }
    // (a failed vote with room for the window: all waves re-execute the tile behind the barrier below)
  } else if (g >= 0) {
    // ================================================================ chunk wave g: steps 8 g .. 8 g + 7
    switch (prio) {  // (earlier groups first on their SIMD)
      case 3: __builtin_amdgcn_s_setprio(3); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      default: break;
    }
    double* const inc_x = reinterpret_cast<double*>(grp + (size_t)g * 4096);  // [8][R]: dt*w, then (dt*v)*cos
    double* const inc_y = inc_x + 8 * R;                                      // [8][R]: (dt*v)*sin
    // ---------------------------------------------------------------- A
    float2 ut[CHL];
    if (!folded) {  // (requested before the Philox blocks: their first touch is a trip to memory)
#pragma unroll
      for (int j = 0; j < CHL; ++j) ut[j] = uq[min(t0 + j, T - 1)];
    }
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
    MPPI_STAMP(stamp_wg && c < 16, stamp_base + 10);
    // (the noise goes to LDS before the wait for the controls: less is left to do once they are there)
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      e[j] = j < nvalid ? e[j] : make_float2(0.0f, 0.0f);
      const int t = t0 + j;
      e2[t * R + (r ^ (t & (R - 1)))] = e[j];
    }
    if (folded) {  // (wave-uniform) the sequence this launch's update leaves: formed by wave 0 meanwhile
      if (reducing) {
        if (direct && c >= 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this wave's share of the map window)
        lds_barrier();
      } else {
        (void)wait_for(&u_ready[0]);
      }
#pragma unroll
      for (int j = 0; j < CHL; ++j) ut[j] = u_sh[t0 + j];
    }
    if (!direct) {  // (wave-uniform; direct mode: the noise is all this wave owes before the exact schedule starts; its
                    //  control-cost terms follow behind the schedule's first barrier -- chunk_ccr as background)
    double qx[CHL];  // dt * clipped speed: exact products of float32 factors
#pragma unroll
    for (int j = 0; j < CHL; ++j) {
      const bool valid = j < nvalid;
      ut[j] = valid ? ut[j] : make_float2(0.0f, 0.0f);
      const float v = clip_f32(ut[j].x + e[j].x, Q.v_lo, Q.v_hi);
      const float w = clip_f32(ut[j].y + e[j].y, Q.w_lo, Q.w_hi);
      qx[j] = dt64 * (double)v;
      inc_x[(size_t)(CHL * h + j) * R + r] = dt64 * (double)w;
    }
    raise(&a_done[g], 1);
    if ((c & 3) == 0) __builtin_amdgcn_s_setprio(0);  // (beside the theta and x | y walks from here on)
    MPPI_STAMP(stamp_wg && c < 16, stamp_base + 1);
    // (everything the walks do not wait for comes after their flags)

    // ---------------------------------------------------------------- the control-cost terms of this wave's steps.
    // Here, while the theta walk has not reached this group yet (every group waits >= 0.5k cycles for its headings):
    // at the end of the wave they competed with the late groups' lookups and records for the SIMD, and the noise
    // stayed in registers all the way
    chunk_ccr();
    MPPI_STAMP(stamp_wg && c < 16, stamp_base + 2);

    // ---------------------------------------------------------------- B: sin / cos of this wave's headings
    (void)wait_for(&th_done[g]);
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
        inc_x[(size_t)(CHL * h + j) * R + r] = qx[j] * cs;
        inc_y[(size_t)(CHL * h + j) * R + r] = qx[j] * s;
        if (j + 1 < CHL) {
          const double delta = (double)thv[j + 1] - (double)thv[j];  // exact: both are float32 values
          if (__all(fabs(delta) <= 0.36)) rotate_sincos_f64(delta, s, cs);
          else sincos_f64<false>((double)thv[j + 1], s, cs);
        }
      }
    }
    raise(&b_done[g], 1);
    MPPI_STAMP(stamp_wg && c < 16, stamp_base + 3);

    // ---------------------------------------------------------------- C: lookups, stage costs, events
    (void)wait_for(&xy_done[g]);
    MPPI_STAMP(stamp_wg && c < 16, stamp_base + 4);
    double sg[CHL], n2[CHL];
    [[maybe_unused]] double stt[CHL];  // SPEED: the time each step is charged with
    float xa[CHL + 1], ya[CHL + 1];
    float po[CHL], pu[CHL];
    uint32_t zero_bits = 0, mism_bits = 0, hit_bits = 0;
    const double gt2 = (double)Q.gt2;
    {
#pragma unroll
      for (int j = 0; j <= CHL; ++j) {
        const float2 pj = pos[(size_t)(t0 + j) * R + r];  // (row t0 + 4 <= 8 g + 8: stored with this group)
        xa[j] = pj.x;
        ya[j] = pj.y;
      }
      uint32_t cell[CHL];
#pragma unroll
      for (int j = 0; j < CHL; ++j) cell[j] = scan_lookup<POW2RES, SPEED>(Q, cells16, xa[j], ya[j]);  // the cell step j STARTS in
#pragma unroll
      for (int j = 0; j < CHL; ++j) {
        const double dx = (double)(Q.xg - xa[j + 1]), dy = (double)(Q.yg - ya[j + 1]);
        n2[j] = fma(dx, dx, dy * dy);
        if constexpr (SPEED) sg[j] = sqrt_newton_f64(n2[j]);  // (the step's time needs the cell: below)
        else sg[j] = fma(Q.dist_weight, sqrt_newton_nz_f64(n2[j]), dt64);
        hit_bits |= (n2[j] <= gt2 ? 1u : 0u) << j;
      }
      pin_memory_order();
      MPPI_STAMP(stamp_wg && c < 16, stamp_base + 7);
#pragma unroll
      for (int j = 0; j < CHL; ++j) {
        const uint32_t cl = cell[j];
        zero_bits |= ((int)(cl & 127u) == Q.lin_zero_byte ? 1u : 0u) << j;
        mism_bits |= (((cl ^ ref) & 0x3fffu) != 0u ? 1u : 0u) << j;
        po[j] = (cl & 0x4000u) ? Q.obs_cost : 0.0f;
        pu[j] = (cl & 0x8000u) ? Q.unk_cost : 0.0f;
        if constexpr (SPEED) {
          // dt over the risk-aware effective speed of the cell the step starts in (mppi.py:1095-1096)
          const double eff = fma(Q.lin_ratio, (double)(int)(int8_t)(cl >> 16), Q.lin_lo);
          stt[j] = dt64 / (eff + 1e-6);
          sg[j] = fma(Q.dist_weight, sg[j], stt[j]);
        }
      }
    }
    MPPI_STAMP(stamp_wg && c < 16, stamp_base + 11);
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
      [[maybe_unused]] double f_st = dt64;
      if constexpr (SPEED) f_st = stt[0];
#pragma unroll
      for (int j = 1; j < CHL; ++j) {
        fx = s == j ? xa[j] : fx;
        fy = s == j ? ya[j] : fy;
        f_po = s == j ? po[j] : f_po;
        f_pu = s == j ? pu[j] : f_pu;
        if constexpr (SPEED) f_st = s == j ? stt[j] : f_st;
      }
      // a rollout in a cell of zero linear traction stays where it is: x = float32(fma(0, ., x))
      const double dx = (double)(Q.xg - fx), dy = (double)(Q.yg - fy);
      f_d2 = fma(dx, dx, dy * dy);
      if constexpr (SPEED) f_k = fma(Q.dist_weight, sqrt_newton_f64(f_d2), f_st);  // (it goes on paying that cell's time)
      else f_k = fma(Q.dist_weight, sqrt_newton_nz_f64(f_d2), dt64);
      f_hit = f_d2 <= gt2;
    }
    // A rollout that stops in a zero-traction cell goes on paying the stage cost and the penalties of the
    // place where it stands, step after step (once, if it stands inside the goal circle): what and from
    // where, for whoever writes the records of the later steps (bits: 0-1 step of the chunk, 2 inside the
    // goal circle, 3 obstacle, 4 unknown).  Written by every lane that sees a stop, owner of the event or not.
    if (froze)
      stop_sh[(size_t)k * R + r] = make_double2(
          f_k, __longlong_as_double((long long)((uint32_t)s | (f_hit ? 4u : 0u) | (f_po != 0.0f ? 8u : 0u) | (f_pu != 0.0f ? 16u : 0u))));
    // the events of ALL earlier chunks must be in evw when this wave looks: then the first event wins
    // (a chain from wave to wave that must stay short -- it is the last stage's critical path: publish and
    //  pass the baton, only then look who owns what)
    MPPI_STAMP(stamp_wg && c < 16, stamp_base + 12);
    if (g > 0) (void)wait_for(&ev_done[g - 1]);
    if (ev != 0u) atomicOr(&evw[2 * r + ((2 * k) >> 5)], ev << ((2 * k) & 31));
    raise(&ev_done[g], 1);
    MPPI_STAMP(stamp_wg && c < 16, stamp_base + 5);

    // ---------------------------------------------------------------- D': the records of this wave's steps
    bool group_pen;
    {
      // (the lower half's event of this very wave is in evw: LDS operations of a wave complete in order)
      const uint32_t w0 = evw[2 * r], w1 = evw[2 * r + 1];
      const uint64_t word = ((uint64_t)w1 << 32) | w0;
      const bool dead = (word & ((1ull << (2 * k)) - 1ull)) != 0ull;
      n_act = dead ? 0 : n_act;
      const bool owner = !dead && ev != 0u;
      if (owner) {
        // (1 - reached) * sqrt(d2) / (v_post + 1e-6)   (mppi.py:26-28, 1005): zero after a goal hit, from the
        // frozen position otherwise; a rollout without any event: the cost wave, from the final position
        term_sh[r] = (ev == 2u && !f_hit) ? sqrt(f_d2) / Q.v_post_den : 0.0;
      }
      MPPI_STAMP(stamp_wg && c < 16, stamp_base + 13);
      if (__any(!dead && bad) && lane == 0) atomicOr(&flags[0], 1u);
      // the first event of the rollout, here or earlier: a stop?
      const int first = word != 0ull ? (__builtin_ctzll(word) >> 1) : 0;
      const bool stopped = word != 0ull && ((word >> (2 * first)) & 3ull) == 2ull;
      double fk = 0.0;
      float fo = 0.0f, fu = 0.0f;
      int f_begin = 0, f_end = 0;
      if (__any(stopped)) {
        const double2 sl = stop_sh[(size_t)(stopped ? first : k) * R + r];  // (written before that chunk's event was published)
        const uint32_t bits = (uint32_t)__double_as_longlong(sl.y);
        fk = sl.x;
        fo = (bits & 8u) ? Q.obs_cost : 0.0f;
        fu = (bits & 16u) ? Q.unk_cost : 0.0f;
        f_begin = stopped ? first * CHL + (int)(bits & 3u) : 0;
        f_end = stopped ? ((bits & 4u) ? f_begin + 1 : T) : 0;
      }
      MPPI_STAMP(stamp_wg && c < 16, stamp_base + 14);
      bool pen = false;
#pragma unroll
      for (int j = 0; j < CHL; ++j) {
        const bool own = j < n_act, standing = t0 + j >= f_begin && t0 + j < f_end;
        sg[j] = own ? sg[j] : (standing ? fk : 0.0);
        po[j] = own ? po[j] : (standing ? fo : 0.0f);
        pu[j] = own ? pu[j] : (standing ? fu : 0.0f);
        pen = pen || po[j] != 0.0f || pu[j] != 0.0f;
      }
      group_pen = __any(pen);
      MPPI_STAMP(stamp_wg && c < 16, stamp_base + 15);
      // the group's records over its (consumed) position increments
      char* out = grp + (size_t)g * 4096 + (size_t)h * 2048 + (size_t)r * 64;
      double2* o2 = reinterpret_cast<double2*>(out);
      o2[0] = make_double2(sg[0], sg[1]);
      o2[1] = make_double2(sg[2], sg[3]);
      float4* o4 = reinterpret_cast<float4*>(out + CHL * 8);
      o4[0] = make_float4(po[0], po[1], po[2], po[3]);
      o4[1] = make_float4(pu[0], pu[1], pu[2], pu[3]);
    }
    raise(&c_done[g], group_pen ? 3 : 1);
    MPPI_STAMP(stamp_wg && c < 16, stamp_base + 6);
    }  // !direct
  }
  if (direct) {
    // ---- no time-parallel attempt (ScanFallback::direct): the noise is in e2, the controls in u_sh (a folded launch) or
    //      in memory, the window's vectors were requested at entry
    if (!reducing) {  // (reducing: the workgroup's first barrier was the hand-over of all three)
      if (c >= 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the map window has landed in LDS
      lds_barrier();
    }
    const float cost = scan_exact_reexecute<POW2RES>(Q, cells16, base + fallback.offset, fallback.map_bytes, e2, uq, T, c, lane,
                                                     folded ? u_sh : (const float2*)nullptr, fallback.rot_ok != 0,
                                                     [&]() { if (g >= 0) chunk_ccr(); }, /*leave_when_idle=*/true);
    if (c == 1) {
      __builtin_amdgcn_s_setprio(3);
      finish_tile(cost);
    }
    lds_barrier();
  } else {
  lds_barrier();
  if (fallback.offset >= 0 && flags[0] != 0u) {  // (workgroup-uniform: every vote is in)
    // ---- a failed vote: the tile again, on the exact pipelined schedule (scan_exact_reexecute above)
    if (c == 1 && lane == 0 && Q.spec_failures) atomicAdd_system(Q.spec_failures, 1u);  // (the host decides whether this map pays)
    const float cost = scan_exact_reexecute<POW2RES>(Q, cells16, base + fallback.offset, fallback.map_bytes, e2,
                                                     folded ? u_sh : uq, T, c, lane, (const float2*)nullptr, fallback.rot_ok != 0, []() {});
    if (c == 1) {
      __builtin_amdgcn_s_setprio(3);
      finish_tile(cost);
    }
    lds_barrier();
  }
  }

  // ---------------------------------------------------------------- F: the tile's share of the update
  // (lane = step, two waves.  Every chunk wave its own 8 steps with a DPP tree over the rollouts was
  //  measured: 2.2k cycles against 1.3k -- 200 four-byte stores at the very end of the launch)
  if (c == 2 || c == 3) {
    const int t = 64 * (c - 2) + lane;
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
}

}  // namespace mppi
