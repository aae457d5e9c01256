// update_kernels.h -- min-subtract + exp-weighted control update (gfx950).
//
// Replaces update_useq_numba[1, 32] (mppi.py:1113-1191): one 32-thread block
// doing N*T*2 float32 atomics.  Here:
//   tile weights    w_rel[n] = exp(-(c_n - beta_tile)/lambda) per tile of 64 rollouts, emitted
//                   by the pipelined rollout kernel's epilogue (k_tile_weights otherwise)
//   k_update_rows   grid T, independent workgroups: rescale by exp(-(beta_tile-beta)/lambda),
//                   den and num[t] in float64; applies the update (one GPU) or writes the
//                   rank packet {beta, den, num[T][2]} for the all-gather (several)
//   k_apply         grid 1: combine the packets of all GPUs, u += num/den, clip
// Everything is a deterministic tree (no atomics): the same inputs give the same bits on every run.  What IS
// bit-identical to the reference are the COSTS; u is held to 1e-5 of the control range (north_star's bound; the
// reference's own update is unordered float32 atomics), and its last bits follow the summation tree: the tiles of
// the rollout kernel that ran (64 rollouts; 32 for the time-parallel kernels, whose weights go through v_exp_f32 and
// whose tile sums are float32 fma chains) and the shards of a multi-GPU run (float64 partial sums per rank).  For a
// given kernel family and GPU count every path to u -- update launch, next-launch fold, all-gather + k_apply, peer
// exchange -- runs the same expressions in the same order and gives the same bits (tests/test_gpu_reduce_fold.py,
// tests/test_gpu_p2p.py, tests/test_gpu_multi.py).
//
// Exactness of the split across GPUs: with beta = min_g beta_g,
//   sum_n exp(-(c_n-beta)/l) x_n = sum_g exp(-(beta_g-beta)/l) * sum_{n in g} exp(-(c_n-beta_g)/l) x_n
#pragma once
#include "device_math.h"

namespace mppi {

constexpr int kUpdateThreads = 256;

// packet layout (doubles): [0] beta_g  [1] den_g  [2 + 2t + c] num_g[t][c]
__host__ __device__ inline int packet_len(int n_steps) { return 2 + 2 * n_steps; }

// u[t] = clip(u[t] + num[t]/den); u_prev mirrors it (the reference aliases
// u_prev_d to u_cur_d before the update, mppi.py:362)
// u_mirror: the same sequence in host-mapped pinned memory (posted PCIe writes), so that solve()
// only has to wait for the stream instead of queueing a device-to-host copy behind it
__device__ __forceinline__ float2 updated_control(float2 ut, double nx, double ny, double den, float v_lo,
                                                  float v_hi, float w_lo, float w_hi) {
  ut.x = clip_f32(ut.x + (float)(nx / den), v_lo, v_hi);
  ut.y = clip_f32(ut.y + (float)(ny / den), w_lo, w_hi);
  return ut;
}

__device__ __forceinline__ void apply_update(float2* u, float2* u_prev, float2* u_mirror, int t, double nx,
                                             double ny, double den, float v_lo, float v_hi, float w_lo,
                                             float w_hi) {
  const float2 ut = updated_control(u[t], nx, ny, den, v_lo, v_hi, w_lo, w_hi);
  u[t] = ut;
  u_prev[t] = ut;
  if (u_mirror) u_mirror[t] = ut;
}

// ---- the update of a sharded iteration, applied by its CONSUMER ---------------------------------
// With the control samples sharded over G GPUs an iteration ends with every rank holding the G
// packets {beta_g, den_g, num_g[t]} (one all-gather).  Turning them into u is k_apply below: one
// tiny dependent launch, ~2-4 us on a 20 us iteration.  The time-parallel rollout kernels
// (rollout_scan*.h) take the packets instead and form u[t] themselves -- every wave the 8 steps it
// owns, with k_apply's expressions in k_apply's order (same bits on every rank); the first tile's
// workgroup also stores the sequence, into the OTHER control buffer (the rest of the grid is still
// reading the old one).  packets == nullptr: nothing pending, u is read as it is.
constexpr int kMaxFoldedRanks = 16;

// The peer exchange (no RCCL on the iteration's path).  Every rank owns an INBOX in fine-grained device memory:
// [2 sets][world][T steps][4 words]; rank g's numbers for step t -- the doubles beta_g, den_g, num_g[t].x, num_g[t].y of
// its packet (packet_len) -- arrive as four 8-byte words, the doubles themselves, over kNotArrived (all ones): each word
// is its own flag, so nothing orders them and nobody fences.  Set `set` is used by this exchange; the reader clears the
// words it has read (the peers write this set again two exchanges later, after they have received what this rank
// sends in between, from its next launch: exchange_step).  world <= kMaxFoldedRanks.
struct PeerExchange {
  unsigned long long* inbox[kMaxFoldedRanks];  // every rank's inbox as THIS device addresses it; [rank] is the own one
  int world, rank, set;                        // world == 0: no exchange
  unsigned int* fault;                         // host-mapped word: raised when a peer's numbers did not arrive in time
  int max_polls;                               // ... i.e. after this many polls (~0.3 us each)
};
// (+ one PING word per rank behind the two sets: mppi_planner_p2p_ping)
constexpr int kInboxWords = 4;                          // words per (rank, step): beta_g, den_g, num_g[t].x, num_g[t].y
constexpr unsigned long long kNotArrived = ~0ull;       // (a NaN no arithmetic produces; a number that IS that pattern is sent with its lowest bit flipped)
__host__ __device__ inline size_t inbox_ping_offset(int world, int n_steps) { return (size_t)2 * world * n_steps * kInboxWords; }
__host__ __device__ inline size_t inbox_words(int world, int n_steps) { return inbox_ping_offset(world, n_steps) + kMaxFoldedRanks; }

// Can this rank's peers be heard?  Every rank writes `token` into its slot of every inbox and waits -- for a bounded
// number of polls, WITHOUT trapping -- until all ranks' tokens have arrived in its own.  result[0] = ranks heard.
// Run by all ranks at the same time before the exchange is trusted (a set-up where peer stores never become visible
// to a running kernel would otherwise only show as trapped rollout launches).
__global__ void k_p2p_ping(PeerExchange X, size_t ping_offset, unsigned long long token, int max_polls, int* result) {
  const int lane = threadIdx.x;
  if (lane < X.world)
    __hip_atomic_store(X.inbox[lane] + ping_offset + X.rank, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long* own = X.inbox[X.rank] + ping_offset;
  int heard = 0;
  for (int polls = 0; polls < max_polls; ++polls) {
    const bool got = lane < X.world && __hip_atomic_load(const_cast<unsigned long long*>(own) + lane, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_SYSTEM) == token;
    heard = __popcll(__ballot(got));
    if (heard == X.world) break;
    __builtin_amdgcn_s_sleep(16);
  }
  if (lane == 0) *result = heard;
}

struct PendingApply {
  const double* packets;  // [world][stride] as all-gathered; nullptr: none (or: see reduce_tiles)
  float2* u_out;          // the other control buffer
  float2* u_prev;
  double* stats;          // {beta, den} of the update
  int world, stride;      // stride: doubles per rank (= packet_len(T) for the single-problem handle)
  float lambda, v_lo, v_hi, w_lo, w_hi;
  // One GPU, iteration loop of the time-parallel exact kernel (round 4): the previous launch left one packet per
  // TILE and no update kernel ran.  Workgroup t of this launch (t < T; its otherwise idle theta walker) combines the
  // tile packets for step t exactly as block t of k_combine_tiles would -- the same function, the same bits -- and
  // publishes u[t] to every other workgroup through `published`: one 8-byte word {float x; float y} per step (kNotPublished until then),
  // written and polled at agent scope (the word carries its own flag: no fence, no second round trip).  Set
  // `flag_set` is used by this launch and the other one is cleared for the launch after next; k_combine_tiles, which
  // closes every loop, clears both.
  const float* reduce_tiles;        // [reduce_n_tiles][tile_packet_floats(T)]; nullptr: not this mode
  int reduce_n_tiles;
  unsigned long long* published;    // [2][T][kPublishedStride] words, one used per step
  int flag_set;
  // the hand-over is bounded and fails soft (round 5): a workgroup that has polled `max_polls` times for a step --
  // its publisher is not running: fewer workgroups resident than the launch has, a second tenant or a CU mask --
  // raises this host-mapped word, takes what memory holds for the missing steps (garbage: the call is lost) and goes
  // on, so the launch ends, the host sees the word at its next synchronisation, returns MPPI_ERR_BUSY and stops
  // folding updates into rollout launches on this handle.  No trap on any default path.
  unsigned int* fault;
  int max_polls;
  // Several GPUs in that mode (round 4, the peer exchange): the workgroup that has combined the LOCAL tiles for
  // step t writes this rank's four numbers for that step -- beta_g, den_g, num_g[t] -- straight into every rank's
  // inbox (peer access / IPC-mapped fine-grained memory: xGMI on a multi-GPU node) and waits for the other ranks'
  // numbers for the same step in its own, before it forms and publishes u[t]: no collective, no extra launch.
  PeerExchange peers;
};

// ---- tile packets of the time-parallel kernels (rollout_scan*.h), tile-major ------------------------------
// per tile of 32 (64) rollouts: [0] beta_tile (minimum cost)  [1] sum of w_rel  [2 + 2t + c] sum of w_rel * noise(t)[c]
__host__ __device__ inline int tile_packet_floats(int n_steps) { return 2 + 2 * n_steps; }

// one wave of a workgroup: the scale of every rank's packet and the common denominator (k_apply's
// lines) -> scale_sh[0 .. world), scale_sh[kMaxFoldedRanks] = den, [kMaxFoldedRanks + 1] = beta
__device__ __forceinline__ void pending_apply_prepare(const PendingApply& A, int lane, double* scale_sh) {
  const double mine = A.packets[(size_t)min(lane, A.world - 1) * A.stride];
  double beta = A.packets[0];
  for (int g = 1; g < A.world; ++g) beta = fmin(beta, A.packets[(size_t)g * A.stride]);
  const double neg_inv_lambda = -1.0 / (double)A.lambda;
  if (lane < A.world) scale_sh[lane] = exp(neg_inv_lambda * (mine - beta));
  if (lane == 0) {  // (the wave's own LDS writes are visible to it in order)
    double den = 0.0;
    for (int g = 0; g < A.world; ++g) den += scale_sh[g] * A.packets[(size_t)g * A.stride + 1];  // (as k_apply: product, then sum)
    scale_sh[kMaxFoldedRanks] = den;
    scale_sh[kMaxFoldedRanks + 1] = beta;
  }
}

// one control of the updated sequence (any lane, after pending_apply_prepare)
__device__ __forceinline__ float2 pending_apply_control(const PendingApply& A, const double* scale_sh,
                                                        const float2* __restrict__ u_old, int t) {
  double nx = 0.0, ny = 0.0;
  for (int g = 0; g < A.world; ++g) {
    const double sg = scale_sh[g];
    const double2 num = *reinterpret_cast<const double2*>(A.packets + (size_t)g * A.stride + 2 + 2 * t);
    nx = fma(sg, num.x, nx);
    ny = fma(sg, num.y, ny);
  }
  return updated_control(u_old[t], nx, ny, scale_sh[kMaxFoldedRanks], A.v_lo, A.v_hi, A.w_lo, A.w_hi);
}

// ---- the combination of the tile packets for ONE step, by one wave (k_combine_tiles; PendingApply::reduce_tiles)
// beta = min over the tiles' minima, scale_tile = exp(-(beta_tile - beta)/lambda) through v_exp_f32 (the argument is
// an exact float32 difference), den and num[t] in float64: each lane its tiles lane, lane + 64, ... in order, then a
// fixed butterfly -- the same bits wherever it runs.  Four tiles per lane cover N = 8192 (256 tiles of 32 rollouts)
// in one batch of loads; everything is requested before anything is waited for.
struct StepSums {
  float beta;
  double den, nx, ny;
};
// (issued apart from their use: a rollout launch requests them before its first barrier)
struct StepLoads {
  static constexpr int U = 4;
  float2 bd[U], m[U];
};
__device__ __forceinline__ StepLoads combine_step_issue(const float* __restrict__ tile_packets, int n_tiles, int stride,
                                                        int t, int lane) {
  StepLoads L;
#pragma unroll
  for (int q = 0; q < StepLoads::U; ++q) {
    const int g = min(lane + 64 * q, n_tiles - 1);
    const float* at = tile_packets + (size_t)g * stride;
    L.bd[q] = *reinterpret_cast<const float2*>(at);
    L.m[q] = *reinterpret_cast<const float2*>(at + 2 + 2 * t);
  }
  return L;
}
__device__ __forceinline__ StepSums combine_step_finish(StepLoads L, const float* __restrict__ tile_packets, int n_tiles,
                                                        int stride, int t, float lambda, int lane) {
  constexpr int U = StepLoads::U;
#pragma unroll
  for (int q = 0; q < U; ++q)
    if (lane + 64 * q >= n_tiles) {
      L.bd[q] = make_float2(__builtin_inff(), 0.0f);
      L.m[q] = make_float2(0.0f, 0.0f);
    }
  float bm = fminf(fminf(L.bd[0].x, L.bd[1].x), fminf(L.bd[2].x, L.bd[3].x));
  for (int g = lane + 64 * U; g < n_tiles; g += 64) bm = fminf(bm, tile_packets[(size_t)g * stride]);
  StepSums S;
  S.beta = wave_min_f32(bm);
  const float beta = S.beta;
  const float scale = -1.4426950408889634f / lambda;
  double den = 0.0, nx = 0.0, ny = 0.0;
  auto take = [&](float tb, float td, float2 tn) {
    const double s = (double)__builtin_amdgcn_exp2f((tb - beta) * scale);
    den = fma(s, (double)td, den);
    nx = fma(s, (double)tn.x, nx);
    ny = fma(s, (double)tn.y, ny);
  };
#pragma unroll
  for (int q = 0; q < U; ++q) take(L.bd[q].x == __builtin_inff() ? beta : L.bd[q].x, L.bd[q].y, L.m[q]);
  for (int g = lane + 64 * U; g < n_tiles; g += 64) {
    const float* at = tile_packets + (size_t)g * stride;
    take(at[0], at[1], *reinterpret_cast<const float2*>(at + 2 + 2 * t));
  }
  S.den = wave_sum_f64(den);
  S.nx = wave_sum_f64(nx);
  S.ny = wave_sum_f64(ny);
  return S;
}
__device__ __forceinline__ StepSums combine_step(const float* __restrict__ tile_packets, int n_tiles, int stride, int t,
                                                 float lambda, int lane) {
  return combine_step_finish(combine_step_issue(tile_packets, n_tiles, stride, t, lane), tile_packets, n_tiles, stride, t,
                             lambda, lane);
}

// One wave, after it has combined the local tiles for step t: send, receive, and combine over the ranks with
// k_apply's expressions in k_apply's order (the bits of the all-gather + k_apply path).  Returns the sums of the whole
// job in S (beta as double in *beta_out).  Bounded polls: when a rank's numbers have not arrived after X.max_polls
// polls (a few seconds by default: a dead or badly stalled peer) the wave raises X.fault -- a host-mapped word -- and goes on with what it has; every later wait
// of this and the following launches sees the word and does not wait again, and the call that synchronises next
// returns MPPI_ERR_COMM: garbage in u, but a running device and a process that can report.
__device__ __forceinline__ StepSums exchange_step(const PeerExchange& X, const StepSums& mine, int t, int n_steps, float lambda,
                                                  int lane, double* beta_out) {
  constexpr int NW = kInboxWords;
  const size_t set_words = (size_t)X.world * n_steps * NW;
  unsigned long long* own = X.inbox[X.rank];
  // (the words of this set for this step were cleared by this very workgroup when it had read them two exchanges ago
  //  -- a launch boundary ago at least: see below)
  // this rank's four words: lane i < 4 holds number i
  const double vals[4] = {(double)mine.beta, mine.den, mine.nx, mine.ny};
  unsigned long long word = 0ull;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (lane == k) word = (unsigned long long)__double_as_longlong(vals[k]);
  word = word == kNotArrived ? word ^ 1ull : word;
  const size_t slot = ((size_t)X.set * X.world + X.rank) * n_steps + t;  // (the same place in every inbox)
  for (int q = 0; q < X.world; ++q)
    if (lane < NW) __hip_atomic_store(X.inbox[q] + slot * NW + lane, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // receive: lane i polls number i & 3 of rank i >> 2 (world <= 16: one word per lane)
  unsigned long long got = kNotArrived;
  const unsigned long long* in = own + (size_t)X.set * set_words;
  for (int polls = 0;; ++polls) {
    bool all_there = true;
    {
      const int q = lane >> 2;
      if (q < X.world && got == kNotArrived) {
        got = __hip_atomic_load(const_cast<unsigned long long*>(in) + ((size_t)q * n_steps + t) * NW + (lane & 3), __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_SYSTEM);
        all_there = got != kNotArrived;
      }
    }
    if (__all(all_there)) break;
    if ((polls & 255) == 255 && X.fault &&
        (polls > X.max_polls || __hip_atomic_load(X.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)) {
      if (lane == 0) __hip_atomic_store(X.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
    if (!X.fault && polls > (1 << 24)) __builtin_trap();
    __builtin_amdgcn_s_sleep(2);
  }
  // Everything for this step has been read: clear the words for the exchange after next.  Nobody writes them before
  // then -- a peer sends into this set again only after it has received what this rank sends in the NEXT exchange,
  // which this rank's next launch does, and these stores are performed when this launch ends.  (Clearing the other
  // set BEFORE sending, as the first version did, put a round trip to memory in front of every send.)
  if (lane < NW * X.world)
    __hip_atomic_store(const_cast<unsigned long long*>(in) + ((size_t)(lane >> 2) * n_steps + t) * NW + (lane & 3), kNotArrived,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // k_apply's lines over the ranks' numbers (wave-uniform: every lane computes the same)
  const unsigned int got_lo = (unsigned int)got, got_hi = (unsigned int)(got >> 32);
  auto number = [&](int q, int k) {  // double k of rank q
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)got_hi, NW * q + k);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)got_lo, NW * q + k);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  };
  double beta = number(0, 0);
  for (int q = 1; q < X.world; ++q) beta = fmin(beta, number(q, 0));
  const double neg_inv_lambda = -1.0 / (double)lambda;
  StepSums S;
  S.den = 0.0;
  S.nx = 0.0;
  S.ny = 0.0;
  for (int q = 0; q < X.world; ++q) S.den += exp(neg_inv_lambda * (number(q, 0) - beta)) * number(q, 1);
  for (int q = 0; q < X.world; ++q) {
    const double sg = exp(neg_inv_lambda * (number(q, 0) - beta));
    S.nx = fma(sg, number(q, 2), S.nx);
    S.ny = fma(sg, number(q, 3), S.ny);
  }
  S.beta = (float)beta;
  *beta_out = beta;
  return S;
}

// The published sequence: ONE 8-byte word per step, {float u.x; float u.y}; a word that has not been written holds
// kNotPublished (all ones: a pair of NaNs with a payload no arithmetic produces; a control that did come out as that
// very pattern is published with its lowest bit flipped -- still a NaN).  The words sit 4 KiB apart -- 256 workgroups
// poll them while ~100 publish: next to each other they are one hot spot behind one memory channel.
constexpr int kPublishedStride = 512;  // words per step
constexpr unsigned long long kNotPublished = ~0ull;
__host__ __device__ inline size_t published_words(int n_steps) { return (size_t)2 * n_steps * kPublishedStride; }
__device__ __forceinline__ unsigned long long published_word(float2 v) {
  const unsigned long long w = ((unsigned long long)__float_as_uint(v.y) << 32) | (unsigned long long)__float_as_uint(v.x);
  return w == kNotPublished ? w ^ 1ull : w;
}

// PendingApply::reduce_tiles, one wave of workgroup `tile`: step t of the update -> its published word (and the
// handle's other control buffer, u_prev, the update's {beta, den}); the words of the other set are cleared
__device__ __forceinline__ void publish_step(const PendingApply& A, StepSums S, float2 u_old, int t, int n_steps,
                                             int lane) {
  double beta = (double)S.beta;
  if (A.peers.world > 0) S = exchange_step(A.peers, S, t, n_steps, A.lambda, lane, &beta);
  if (lane == 0) {
    const float2 ut = updated_control(u_old, S.nx, S.ny, S.den, A.v_lo, A.v_hi, A.w_lo, A.w_hi);
    unsigned long long* mine = A.published + ((size_t)A.flag_set * n_steps + t) * kPublishedStride;
    unsigned long long* other = A.published + ((size_t)(A.flag_set ^ 1) * n_steps + t) * kPublishedStride;
    __hip_atomic_store(mine, published_word(ut), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    other[0] = kNotPublished;  // (polled again two launches from now at the earliest)
    A.u_out[t] = ut;
    A.u_prev[t] = ut;
    if (t == 0) {
      A.stats[0] = beta;
      A.stats[1] = S.den;
    }
  }
}

// every workgroup, one wave: the published sequence -> u_sh[0 .. Tp) (LDS); steps past the horizon are zero.
// A lane polls the steps lane, lane + 64, ... and stops asking for a step once it has it.
// The publishers are the first workgroups of the grid, dispatched before any workgroup that waits for them, so with
// the whole grid resident (one workgroup per CU: what the launch plan provides on a device of its own) this cannot
// deadlock.  It is bounded all the same: see PendingApply::fault.
__device__ __forceinline__ void collect_published(const PendingApply& A, int n_steps, int padded_steps, int lane,
                                                  float2* u_sh) {
  unsigned long long* words = A.published + (size_t)A.flag_set * n_steps * kPublishedStride;
  // (with a peer exchange the publishers may themselves be waiting for another rank, up to peers.max_polls of THEIR
  // polls, and then publish what they have and raise the peers' fault word: the collectors must outlast that)
  const int limit = A.peers.world > 0 ? 8 * min(max(A.peers.max_polls, 1 << 17), 1 << 27) : max(A.max_polls, 1);
  for (int base = 0; base < padded_steps; base += 128) {  // (one round for T <= 128)
    unsigned long long w[2] = {kNotPublished, kNotPublished};
    for (int polls = 0;; ++polls) {
      bool all_there = true;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = base + lane + 64 * q;
        if (t < n_steps && w[q] == kNotPublished) {
          w[q] = __hip_atomic_load(words + (size_t)t * kPublishedStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          all_there = all_there && w[q] != kNotPublished;
        }
      }
      if (__all(all_there)) break;
      // give up: after `limit` polls, or as soon as another workgroup of this or an earlier launch has
      if (polls > limit || ((polls & 255) == 255 && A.fault &&
                            __hip_atomic_load(A.fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)) {
        if (!A.fault) __builtin_trap();  // (no word to raise: a handle built without one -- not the library's)
        if (lane == 0) __hip_atomic_store(A.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int q = 0; q < 2; ++q) {  // (whatever the other control buffer holds: the call is lost, the launch ends)
          const int t = base + lane + 64 * q;
          if (t < n_steps && w[q] == kNotPublished) w[q] = published_word(A.u_out[t]);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int t = base + lane + 64 * q;
      if (t < padded_steps)
        u_sh[t] = t < n_steps ? make_float2(__uint_as_float((unsigned int)w[q]), __uint_as_float((unsigned int)(w[q] >> 32)))
                              : make_float2(0.0f, 0.0f);
    }
  }
}

// ---- stage 1: weights relative to the minimum of each tile of 64 rollouts ----------
// w_rel[n] = float32(exp(-(c_n - beta_tile)/lambda)), tile_beta[tile] = min of the tile's
// costs.  Needs no global minimum, so the pipelined rollout kernel emits it in its
// epilogue (one exp per lane); k_tile_weights does it for the other rollout kernels.
// The identity at the top of this file makes the later rescaling exact.
__device__ __forceinline__ void emit_tile_weights(float cost, bool live, float lambda, int n, int tile,
                                                  float* __restrict__ w_rel, float* __restrict__ tile_beta) {
  float beta = wave_min_f32(live ? cost : __builtin_inff());
  if (live) w_rel[n] = (float)exp(-1.0 / (double)lambda * (double)(cost - beta));
  if ((threadIdx.x & 63) == 0) tile_beta[tile] = beta;
}

__global__ __launch_bounds__(64) void k_tile_weights(const float* __restrict__ costs, int n, float lambda,
                                                     float* __restrict__ w_rel, float* __restrict__ tile_beta) {
  int i = blockIdx.x * 64 + threadIdx.x;
  bool live = i < n;
  emit_tile_weights(live ? costs[i] : 0.0f, live, lambda, i, blockIdx.x, w_rel, tile_beta);
}

// ---- stage 2: the update, one launch, T independent workgroups ------------------------
// Workgroup t: beta = min over tiles; scale_tile = exp(-(beta_tile - beta)/lambda) (LDS);
// den = sum_n scale*w_rel and num = sum_n scale*w_rel*eps(t, n) in float64 (each thread a
// strided subset in order, fixed butterfly, fixed wave order: deterministic, and den comes
// out bit-identical in every workgroup); then
//   APPLY  (single GPU): u[t] = clip(u[t] + num/den)
//   !APPLY (one of several GPUs): num -> this rank's packet for the all-gather.
// stats[0] = beta, stats[1] = den (for the normalised weights the host may ask for).
// Batched handle: blockIdx.y = problem b, which owns rollouts [b*n, (b+1)*n), tiles
// [b*n_tiles, (b+1)*n_tiles), u[b], stats[b], packet segment b.
constexpr int kRowThreads = 1024;

// TC rows per workgroup: 1 keeps the workgroup count up when there are few rows (one problem,
// latency matters); 4 amortises the minimum / scale prologue and the reductions when the
// grid would otherwise be thousands of short-lived workgroups (batched handles).  Each
// thread sums its strided subset in increasing i whatever TC and UN are, so the result does
// not depend on them.
// FROM_COSTS: `w_rel` is the cost vector; the tile minima and the tile-relative weights are
// formed here with the very expressions of emit_tile_weights (same bits), which saves the
// k_tile_weights launch after the rollout kernels that have no such epilogue (CVaR, speed-map
// fallback, barebone).
// THREADS (round 6): 1024, or 256 with four times the loads in flight per thread -- the same bytes on their way with a
// quarter of the waves, for the launches beside which the next iteration's noise generator runs (the batched handles:
// launch_plan.h, beside_update): the generator is bound by instruction issue and needs the wave slots this kernel, bound
// by memory, can do without.  Each thread still sums its strided subset in increasing i; the strides differ, so `u`
// differs in its last bits between the two (held to 1e-5 of the range against the oracle either way).
// progress: "this launch has started" for the gate kernel in front of that generator (rollout_kernels.h: k_wait_progress).
template <bool APPLY, int TC, bool FROM_COSTS = false, int THREADS = kRowThreads>
__global__ __launch_bounds__(THREADS) void k_update_rows(const float* __restrict__ w_rel,
                                                             const float* __restrict__ tile_beta, int n, int n_tiles,
                                                             const float2* __restrict__ noise, int n_steps,
                                                             float lambda, double* __restrict__ rank_packet,
                                                             float2* __restrict__ u, float2* __restrict__ u_prev,
                                                             float2* __restrict__ u_mirror, float v_lo, float v_hi,
                                                             float w_lo, float w_hi, double* __restrict__ stats,
                                                             unsigned long long* __restrict__ gen_counter,
                                                             unsigned long long* __restrict__ ktime, int ktime_waves,
                                                             unsigned long long* __restrict__ progress,
                                                             unsigned long long progress_value) {
  extern __shared__ float scale_sh[];  // [n_tiles]
  if (progress && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
    __hip_atomic_store(progress, progress_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // mppi_planner_time_kernels (else nullptr): when every wave of this launch entered / left (as DevParams::ktime)
  const int ktime_slot = ((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * (THREADS / 64) + (int)(threadIdx.x >> 6);
  if (ktime && (threadIdx.x & 63) == 0) ktime[ktime_slot] = (unsigned long long)wall_clock64();
  // graph replay: one more generation of noise has been consumed (rng_kernels.h NoiseJob); no
  // generator runs while an update does
  if (gen_counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *gen_counter += 1ull;
  constexpr int kCols = 1 + 2 * TC;    // den, then (x, y) per row
  __shared__ double red[THREADS / 64][kCols];
  __shared__ float redf[THREADS / 64];
  const int t0 = blockIdx.x * TC, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const int inst = blockIdx.y;
    w_rel += (size_t)inst * n;
    tile_beta += (size_t)inst * n_tiles;
    noise += (size_t)inst * n_tiles * n_steps * 64;  // tile-major: whole tiles per problem
    rank_packet += (size_t)inst * packet_len(n_steps);
    u += (size_t)inst * n_steps;
    u_prev += (size_t)inst * n_steps;
    if (u_mirror) u_mirror += (size_t)inst * n_steps;
    stats += 2 * inst;
  }
  [[maybe_unused]] const bool stamp_wg = blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x == 0;
  MPPI_STAMP(stamp_wg, 512);
  const double neg_inv_lambda = -1.0 / (double)lambda;
  // rows past the horizon (last workgroup) are clamped for the loads and never written
  size_t row_off[TC];
#pragma unroll
  for (int j = 0; j < TC; ++j) row_off[j] = (size_t)min(t0 + j, n_steps - 1) * 64;
  constexpr int UN = (TC == 1 ? 8 : 2) * (kRowThreads / THREADS);  // independent loads in flight per thread: UN * (TC + 1)
  // The first batch of this thread's weights and noise is requested before the minimum / scale
  // prologue below, which needs ~2k cycles of its own: the stream's first round trip hides behind it.
  // (loads return in order: the tile minima the prologue waits for go first)
  float tb0 = __builtin_inff();
  if (!FROM_COSTS && (int)threadIdx.x < n_tiles) tb0 = tile_beta[threadIdx.x];
  float wr0[UN];
  float2 e0[UN][TC];
  const bool have0 = (int)threadIdx.x + (UN - 1) * THREADS < n;
  if (have0) {
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int idx = threadIdx.x + k * THREADS;
      wr0[k] = w_rel[idx];
      const float2* tile = noise + (size_t)(idx >> 6) * n_steps * 64 + (idx & 63);
#pragma unroll
      for (int j = 0; j < TC; ++j) e0[k][j] = tile[row_off[j]];
    }
  }
  float* tb_sh = scale_sh + n_tiles;  // FROM_COSTS: [n_tiles] tile minima
  if (FROM_COSTS) {
    for (int g = wave; g < n_tiles; g += THREADS / 64) {
      const int i = g * 64 + lane;
      const float m = wave_min_f32(i < n ? w_rel[i] : __builtin_inff());
      if (lane == 0) tb_sh[g] = m;
    }
    __syncthreads();
    tile_beta = tb_sh;
  }
  float b = tb0;
  for (int g = threadIdx.x + (FROM_COSTS ? 0 : THREADS); g < n_tiles; g += THREADS) b = fminf(b, tile_beta[g]);
  b = wave_min_f32(b);
  if (lane == 0) redf[wave] = b;
  __syncthreads();
  MPPI_STAMP(stamp_wg, 513);
  float beta = redf[0];
  for (int k = 1; k < THREADS / 64; ++k) beta = fminf(beta, redf[k]);
  if (!FROM_COSTS && (int)threadIdx.x < n_tiles) scale_sh[threadIdx.x] = (float)exp(neg_inv_lambda * (double)(tb0 - beta));
  for (int g = threadIdx.x + (FROM_COSTS ? 0 : THREADS); g < n_tiles; g += THREADS)
    scale_sh[g] = (float)exp(neg_inv_lambda * (double)(tile_beta[g] - beta));
  __syncthreads();
  MPPI_STAMP(stamp_wg, 514);
  double den = 0.0, nx[TC], ny[TC];
#pragma unroll
  for (int j = 0; j < TC; ++j) nx[j] = ny[j] = 0.0;
  int i = threadIdx.x;                 // THREADS is a multiple of 64: i>>6 is the tile, i&63 the lane
  auto consume = [&](int at, float (&wr)[UN], float2 (&e)[UN][TC]) {
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int tile = (at + k * THREADS) >> 6;
      if (FROM_COSTS) wr[k] = (float)exp(neg_inv_lambda * (double)(wr[k] - tb_sh[tile]));  // emit_tile_weights
      double w = (double)scale_sh[tile] * (double)wr[k];
      den += w;
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        nx[j] = fma(w, (double)e[k][j].x, nx[j]);
        ny[j] = fma(w, (double)e[k][j].y, ny[j]);
      }
    }
  };
  if (have0) {
    consume(i, wr0, e0);
    i += UN * THREADS;
  }
  for (; i + (UN - 1) * THREADS < n; i += UN * THREADS) {
    float wr[UN];
    float2 e[UN][TC];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int idx = i + k * THREADS;
      wr[k] = w_rel[idx];
      const float2* tile = noise + (size_t)(idx >> 6) * n_steps * 64 + (idx & 63);
#pragma unroll
      for (int j = 0; j < TC; ++j) e[k][j] = tile[row_off[j]];
    }
    consume(i, wr, e);
  }
  for (; i < n; i += THREADS) {
    float wr1 = w_rel[i];
    if (FROM_COSTS) wr1 = (float)exp(neg_inv_lambda * (double)(wr1 - tb_sh[i >> 6]));
    double w = (double)scale_sh[i >> 6] * (double)wr1;
    const float2* tile = noise + (size_t)(i >> 6) * n_steps * 64 + (i & 63);
    den += w;
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      float2 e = tile[row_off[j]];
      nx[j] = fma(w, (double)e.x, nx[j]);
      ny[j] = fma(w, (double)e.y, ny[j]);
    }
  }
  MPPI_STAMP(stamp_wg, 515);
  den = wave_sum_f64(den);
#pragma unroll
  for (int j = 0; j < TC; ++j) {
    nx[j] = wave_sum_f64(nx[j]);
    ny[j] = wave_sum_f64(ny[j]);
  }
  if (lane == 0) {
    red[wave][0] = den;
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      red[wave][1 + 2 * j] = nx[j];
      red[wave][2 + 2 * j] = ny[j];
    }
  }
  __syncthreads();
  if (threadIdx.x < kCols) {  // fixed wave order: ((w0 + w1) + w2) + ...
    double acc = red[0][threadIdx.x];
    for (int k = 1; k < THREADS / 64; ++k) acc += red[k][threadIdx.x];
    red[0][threadIdx.x] = acc;
  }
  __syncthreads();
  MPPI_STAMP(stamp_wg, 516);
  if (threadIdx.x < TC && t0 + (int)threadIdx.x < n_steps) {
    const int j = threadIdx.x, t = t0 + j;
    const double d = red[0][0], sx = red[0][1 + 2 * j], sy = red[0][2 + 2 * j];
    if (APPLY) {
      apply_update(u, u_prev, u_mirror, t, sx, sy, d, v_lo, v_hi, w_lo, w_hi);
    } else {
      rank_packet[2 + 2 * t] = sx;
      rank_packet[3 + 2 * t] = sy;
    }
    if (t == 0) {
      rank_packet[0] = (double)beta;
      rank_packet[1] = d;
      stats[0] = (double)beta;
      stats[1] = d;
    }
  }
  MPPI_STAMP(stamp_wg, 517);
  if (ktime && (threadIdx.x & 63) == 0) ktime[ktime_waves + ktime_slot] = (unsigned long long)wall_clock64();
}

// ---- stage 2 after the time-parallel kernels (rollout_scan*.h): the rollout launch has already reduced
// w_rel * noise over each of its tiles (tile packets, tile-major: tile_packet_floats); what is left
// of update_useq_numba (mppi.py:1147-1191) is the combination of the tiles -- tiles x (2T + 2) floats
// instead of a pass over the noise.  Grid (T, problems), one wave per step: beta = min over
// tile_beta, scale_tile = exp(-(beta_tile - beta)/lambda), den / num in float64 in a fixed order
// (identical den in every workgroup), then as k_update_rows: apply, or this rank's packet.
// Inside an iteration loop on one GPU this launch does not run at all: the next rollout launch combines the
// tile packets itself, step t in workgroup t (PendingApply::reduce_tiles, the same combine_step: the same bits);
// it closes a loop (and serves batched handles, several GPUs, the tolerance kernel).  gen_bump: the iterations
// whose noise generation this launch accounts for (graph replay: itself plus those whose update ran inside a
// rollout launch).  published: the words of both flag sets, cleared for the next loop.
template <bool APPLY>
__global__ __launch_bounds__(64) void k_combine_tiles(const float* __restrict__ tile_packets, int n_tiles,
                                                      int n_steps, float lambda, double* __restrict__ rank_packet,
                                                      float2* __restrict__ u, float2* __restrict__ u_prev,
                                                      float2* __restrict__ u_mirror, float v_lo, float v_hi,
                                                      float w_lo, float w_hi, double* __restrict__ stats,
                                                      unsigned long long* __restrict__ gen_counter,
                                                      unsigned long long gen_bump,
                                                      unsigned long long* __restrict__ published, PeerExchange peers) {
  if (gen_counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *gen_counter += gen_bump;
  const int t = blockIdx.x, inst = blockIdx.y, lane = threadIdx.x;
  if (published && inst == 0 && lane < 2) published[((size_t)lane * n_steps + t) * kPublishedStride] = kNotPublished;
  [[maybe_unused]] const bool stamp_wg = (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1) && blockIdx.y == 0;
  [[maybe_unused]] const int stamp_base = blockIdx.x == 0 ? 520 : 528;
  MPPI_STAMP(stamp_wg, stamp_base + 0);
  const int stride = tile_packet_floats(n_steps);
  tile_packets += (size_t)inst * n_tiles * stride;  // problem `inst` owns tiles [inst * n_tiles, (inst + 1) * n_tiles)
  rank_packet += (size_t)inst * packet_len(n_steps);
  u += (size_t)inst * n_steps;
  u_prev += (size_t)inst * n_steps;
  if (u_mirror) u_mirror += (size_t)inst * n_steps;
  stats += 2 * inst;
  const float2 u_old = APPLY ? u[t] : make_float2(0.0f, 0.0f);
  StepSums S = combine_step(tile_packets, n_tiles, stride, t, lambda, lane);
  double beta64 = (double)S.beta;
  // (several GPUs, peer exchange: the last iteration of a loop exchanges here and applies: no collective)
  if (APPLY && peers.world > 0) S = exchange_step(peers, S, t, n_steps, lambda, lane, &beta64);
  MPPI_STAMP(stamp_wg, stamp_base + 3);
  if (lane == 0) {
    if (APPLY) {
      const float2 ut = updated_control(u_old, S.nx, S.ny, S.den, v_lo, v_hi, w_lo, w_hi);
      u[t] = ut;
      u_prev[t] = ut;
      if (u_mirror) u_mirror[t] = ut;
    } else {
      rank_packet[2 + 2 * t] = S.nx;
      rank_packet[3 + 2 * t] = S.ny;
    }
    if (t == 0) {
      rank_packet[0] = beta64;
      rank_packet[1] = S.den;
      stats[0] = beta64;
      stats[1] = S.den;
    }
  }
  MPPI_STAMP(stamp_wg, stamp_base + 4);
}

// combine the packets of all ranks (identical on every GPU, fixed g order).
// Batched handle: blockIdx.x = problem b; rank g's packet for it is packets[(g*B + b)*len].
// u_in / u: the sequence before and after (the same buffer, or the handle's two control buffers: a
// sharded handle alternates them so that a rollout launch can apply the update itself, PendingApply)
__global__ __launch_bounds__(kUpdateThreads) void k_apply(const double* __restrict__ packets, int world,
                                                          int rank, int n_steps, float lambda,
                                                          const float2* u_in, float2* u, float2* __restrict__ u_prev,
                                                          float2* __restrict__ u_mirror, float v_lo, float v_hi,
                                                          float w_lo, float w_hi, double* __restrict__ stats) {
  const int len = packet_len(n_steps);
  const size_t stride = (size_t)gridDim.x * len;  // doubles per rank
  packets += (size_t)blockIdx.x * len;
  u += (size_t)blockIdx.x * n_steps;
  u_in += (size_t)blockIdx.x * n_steps;
  u_prev += (size_t)blockIdx.x * n_steps;
  if (u_mirror) u_mirror += (size_t)blockIdx.x * n_steps;
  stats += 2 * blockIdx.x;
  double beta = packets[0];
  for (int g = 1; g < world; ++g) beta = fmin(beta, packets[g * stride]);
  const double neg_inv_lambda = -1.0 / (double)lambda;
  double den = 0.0;
  for (int g = 0; g < world; ++g) den += exp(neg_inv_lambda * (packets[g * stride] - beta)) * packets[g * stride + 1];
  if (threadIdx.x == 0) {
    stats[0] = beta;
    stats[1] = den;
  }
  for (int t = threadIdx.x; t < n_steps; t += kUpdateThreads) {
    double nx = 0.0, ny = 0.0;
    for (int g = 0; g < world; ++g) {
      double sg = exp(neg_inv_lambda * (packets[g * stride] - beta));
      nx = fma(sg, packets[g * stride + 2 + 2 * t], nx);
      ny = fma(sg, packets[g * stride + 3 + 2 * t], ny);
    }
    const float2 ut = updated_control(u_in[t], nx, ny, den, v_lo, v_hi, w_lo, w_hi);
    u[t] = ut;
    u_prev[t] = ut;
    if (u_mirror) u_mirror[t] = ut;
  }
}

// normalised weights for the host (weights_d of the reference), on demand:
// w_n = exp(-(c_n - beta)/lambda) / den with the global beta and den of the last update
__global__ void k_weights_out(const float* __restrict__ costs, const double* __restrict__ stats, float lambda,
                              int n, int n_inst, float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* st = stats + 2 * (i / n_inst);  // {beta, den} of the problem this rollout belongs to
  out[i] = (float)(exp(-1.0 / (double)lambda * ((double)costs[i] - st[0])) / st[1]);
}

// device-side shift of the control sequence: u[:-k] = u[k:], tail kept (mppi.py:539-541)
__global__ void k_shift_u(float2* __restrict__ u, int n_steps, int k) {
  // one block per problem: read everything, barrier, write
  extern __shared__ float2 tmp[];
  u += (size_t)blockIdx.x * n_steps;
  for (int t = threadIdx.x; t < n_steps; t += blockDim.x) tmp[t] = u[t];
  __syncthreads();
  for (int t = threadIdx.x; t + k < n_steps; t += blockDim.x) u[t] = tmp[t + k];
}

// host layout (N,T,2) <-> device tile-major layout
__global__ void k_noise_to_device_layout(const float2* __restrict__ host_layout, int n, int t_steps,
                                         float2* __restrict__ dev_layout) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * t_steps) return;
  int t = (int)(i / n), r = (int)(i % n);
  dev_layout[tile_index(t, r, t_steps)] = host_layout[(size_t)r * t_steps + t];
}
__global__ void k_noise_to_host_layout(const float2* __restrict__ dev_layout, int n, int t_steps,
                                       float2* __restrict__ host_layout) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * t_steps) return;
  int r = (int)(i / t_steps), t = (int)(i % t_steps);
  host_layout[i] = dev_layout[tile_index(t, r, t_steps)];
}

}  // namespace mppi
