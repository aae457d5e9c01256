// rng_kernels.h -- control-noise and traction-map sampling (gfx950).
//
// Replaces sample_noise_numba (mppi.py:1354-1370) and sample_grids_numba
// (terrain.py:633-695).
//
// Default generator: rocRAND Philox4x32-10 (device API).  It is counter based:
// the counter is the GLOBAL (rollout, step) index -- respectively the global
// (sample, row, column group) index -- and the call epoch, so a sharded run
// draws exactly the numbers a single-GPU run draws, and no generator state is
// stored in HBM (numba keeps 16 bytes of xoroshiro state per thread and
// round-trips it on every call).
//
// Compatibility generator: numba.cuda.random's xoroshiro128+ with its stream
// per thread and its thread -> cell mapping, bit for bit, so that seed -> u can
// be checked against the reference end to end.
#pragma once
#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <stdint.h>

#include "device_math.h"

namespace mppi {

// ---------------- xoroshiro128+ (numba/cuda/random.py:45-139) ----------------
__host__ __device__ inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

__host__ __device__ inline uint64_t xoroshiro_next(uint64_t& s0, uint64_t& s1) {
  uint64_t result = s0 + s1;
  uint64_t t = s1 ^ s0;
  s0 = rotl64(s0, 55) ^ t ^ (t << 14);
  s1 = rotl64(t, 36);
  return result;
}

__host__ __device__ inline float xoroshiro_uniform_f32(uint64_t& s0, uint64_t& s1) {
  // float32(uint64 >> 11) * 2^-53 in float64)
  return (float)((double)(xoroshiro_next(s0, s1) >> 11) * (1.0 / 9007199254740992.0));
}

// Box-Muller as the reference's CPU path evaluates it (random.py:175-196 on
// Python floats): float32 uniforms, float64 log / sqrt / cos, second normal of
// the pair discarded.
__device__ inline double xoroshiro_normal(uint64_t& s0, uint64_t& s1) {
  const float two_pi_f32 = 6.28318530717958647692f;
  float u1 = xoroshiro_uniform_f32(s0, s1);
  float u2 = xoroshiro_uniform_f32(s0, s1);
  return sqrt(-2.0 * log((double)u1)) * cos((double)(two_pi_f32 * u2));
}

// host: SplitMix64 seeding of stream 0, stream k = stream k-1 jumped by 2^64
// (random.py:46-68, 102-125, 225-240)
inline void xoroshiro_init_host(uint64_t* states, long n, uint64_t seed) {
  if (n < 1) return;
  static const uint64_t kJump[2] = {0xbeac0467eba5facbULL, 0xd86b048b86aa9922ULL};
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  uint64_t s0 = z, s1 = z;
  states[0] = s0;
  states[1] = s1;
  for (long i = 1; i < n; ++i) {
    uint64_t a = 0, b = 0;
    for (int w = 0; w < 2; ++w)
      for (int bit = 0; bit < 64; ++bit) {
        if (kJump[w] & (1ULL << bit)) {
          a ^= s0;
          b ^= s1;
        }
        xoroshiro_next(s0, s1);
      }
    s0 = a;
    s1 = b;
    states[2 * i] = s0;
    states[2 * i + 1] = s1;
  }
}

// ---------------- control noise ----------------
// Philox4x32-10 counter block (Salmon et al. 2011), the generator of rocRAND's
// rocrand_state_philox4x32_10: counter {lo, hi, subsequence lo, subsequence hi}, key = seed.
// rocrand_init() + rocrand4() compute one block to initialise and a second one eagerly;
// here exactly one block is computed per four normals.  k_philox_selftest checks this
// function against rocRAND's engine bit for bit (tests/test_gpu_scale.py).
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply (v_mad_u64_u32) per product instead of mul_hi + mul_lo
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x;
    unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c.z;
    c = make_uint4((unsigned int)(p1 >> 32) ^ c.y ^ k.x, (unsigned int)p1, (unsigned int)(p0 >> 32) ^ c.w ^ k.y,
                   (unsigned int)p0);
    k.x += 0x9E3779B9u;
    k.y += 0xBB67AE85u;
  }
  return c;
}

// Box-Muller on two 32-bit words, as rocRAND's box_muller() maps them (u in (0,1], angle
// in (0, 2pi]) but with the hardware log2 / sqrt / sin / cos (v_sin_f32 takes turns)
__device__ __forceinline__ float2 box_muller_fast(unsigned int x, unsigned int y) {
  float u = fmaf((float)x, 2.3283064365386963e-10f, 2.3283064365386963e-10f);
  float turn = fmaf((float)y, 2.3283064365386963e-10f, 2.3283064365386963e-10f);
  float s = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u));  // -2 ln u = -2 ln2 log2 u
  return make_float2(s * __builtin_amdgcn_sinf(turn), s * __builtin_amdgcn_cosf(turn));
}

// noise(t, n) = u_std * N(0,1), tile-major.
//   Philox: one counter block per (global rollout n, step pair t/2): counter.zw = n*ceil(T/2)+t/2,
//           counter.xy = call epoch -> independent of the number of GPUs, nothing stored
//   xoroshiro: stream n*T+t, two normals per call (4 draws), state kept (reference-compatible)
// One description serves the standalone kernel and the spare workgroups of the pipelined
// rollout kernel, which generate the noise of the NEXT iteration while the current one is
// integrated.
struct NoiseJob {
  float2* out;        // tile-major target buffer (nullptr: nothing to do)
  uint64_t* states;   // xoroshiro states or nullptr (Philox)
  uint64_t seed, epoch;
  // graph replay: kernel arguments are frozen in a captured graph, so the Philox call epoch is
  // epoch + *gen_counter, where the counter lives on the device and every update kernel
  // increments it (nullptr: epoch alone, passed by value per launch)
  const uint64_t* gen_counter;
  int n_local, n_offset, n_steps;
  float std0, std1;
};

// one ROW of 64 lanes per wave iteration: Philox -> (tile, step pair), xoroshiro -> (tile, step)
// (32-bit row arithmetic: the 64-bit divisions of a flat index cost more than the generator)
template <bool STREAMING = false>
__device__ __forceinline__ void noise_row(const NoiseJob& j, unsigned int row, int lane) {
  if (j.states == nullptr) {
    const unsigned int pairs = (unsigned int)(j.n_steps + 1) / 2u;
    const unsigned int tile = row / pairs, tp = row - tile * pairs;
    const int n = (int)tile * 64 + lane;
    if (n >= j.n_local) return;
    uint64_t sub = (uint64_t)(j.n_offset + n) * (uint64_t)pairs + (uint64_t)tp;
    const uint64_t epoch = j.epoch + (j.gen_counter ? *j.gen_counter : 0ull);  // uniform: a scalar load
    uint4 r = philox4x32_10(make_uint4((unsigned int)epoch, (unsigned int)(epoch >> 32), (unsigned int)sub,
                                       (unsigned int)(sub >> 32)),
                            make_uint2((unsigned int)j.seed, (unsigned int)(j.seed >> 32)));
    float2 a = box_muller_fast(r.x, r.y), b = box_muller_fast(r.z, r.w);
    float2* o = j.out + ((size_t)tile * j.n_steps + 2 * tp) * 64 + lane;
    if (STREAMING) {
      // written past the caches: these are the spare workgroups of a latency-regime rollout launch, and
      // 6.5 MB of write-allocated lines in the L2 the rollout tiles are reading through cost the
      // launch 2.8 us at C2 (20.8 -> 18.0 us).  The standalone generator keeps ordinary stores: the
      // throughput-regime rollout that follows it wants the noise in the L2 / infinity cache (C4: 83
      // -> 106 us with streaming stores).
      __builtin_nontemporal_store(j.std0 * a.x, &o[0].x);
      __builtin_nontemporal_store(j.std1 * a.y, &o[0].y);
      if (2 * (int)tp + 1 < j.n_steps) {
        __builtin_nontemporal_store(j.std0 * b.x, &o[64].x);
        __builtin_nontemporal_store(j.std1 * b.y, &o[64].y);
      }
    } else {
      o[0] = make_float2(j.std0 * a.x, j.std1 * a.y);
      if (2 * (int)tp + 1 < j.n_steps) o[64] = make_float2(j.std0 * b.x, j.std1 * b.y);
    }
  } else {
    const unsigned int steps = (unsigned int)j.n_steps;
    const unsigned int tile = row / steps, t = row - tile * steps;
    const int n = (int)tile * 64 + lane;
    if (n >= j.n_local) return;
    size_t k = (size_t)n * j.n_steps + t;
    uint64_t s0 = j.states[2 * k], s1 = j.states[2 * k + 1];
    double z0 = xoroshiro_normal(s0, s1);
    double z1 = xoroshiro_normal(s0, s1);
    j.states[2 * k] = s0;
    j.states[2 * k + 1] = s1;
    j.out[(size_t)row * 64 + lane] = make_float2((float)((double)j.std0 * z0), (float)((double)j.std1 * z1));
  }
}

__host__ __device__ inline size_t noise_items(int n_local, int n_steps, bool philox) {
  return (size_t)((n_local + 63) / 64) * 64 * (size_t)(philox ? (n_steps + 1) / 2 : n_steps);
}

// wave `wave_id` of `n_waves` cooperating waves: rows wave_id, wave_id + n_waves, ...
// STREAMING: nontemporal stores (see noise_row)
template <bool STREAMING = false>
__device__ __forceinline__ void noise_generate(const NoiseJob& j, unsigned int wave_id, unsigned int n_waves) {
  const unsigned int rows = (unsigned int)(noise_items(j.n_local, j.n_steps, j.states == nullptr) >> 6);
  const int lane = threadIdx.x & 63;
  for (unsigned int row = wave_id; row < rows; row += n_waves) noise_row<STREAMING>(j, row, lane);
}

__global__ __launch_bounds__(256) void k_noise(NoiseJob job) {
  noise_generate(job, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), gridDim.x * (blockDim.x >> 6));
}

// out[0] = number of (seed, subsequence, offset) triples for which philox4x32_10() differs
// from rocRAND's own engine
__global__ void k_philox_selftest(int* out) {
  unsigned int tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t seed = 0x1234ABCD5678EF01ULL * (tid % 7 + 1), sub = (uint64_t)tid * 0x9E3779B97F4A7C15ULL,
           block = (uint64_t)tid * 977u + (tid % 3 ? 0 : 0xFFFFFFFF0ULL);
  rocrand_state_philox4x32_10 st;
  rocrand_init(seed, sub, 4ULL * block, &st);
  uint4 want = rocrand4(&st);
  uint4 got = philox4x32_10(make_uint4((unsigned int)block, (unsigned int)(block >> 32), (unsigned int)sub,
                                       (unsigned int)(sub >> 32)),
                            make_uint2((unsigned int)seed, (unsigned int)(seed >> 32)));
  if (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w) atomicAdd(out, 1);
}

// ---------------- traction-map sampling ----------------
// inverse-CDF draw from the int8 PMF (bins sum to 100), terrain.py:682-689
__device__ __forceinline__ int8_t draw_bin(const int8_t* __restrict__ pmf, size_t cell, size_t plane, int bins,
                                           const int8_t* __restrict__ table, int8_t target) {
  int8_t cum = 0;
  for (int b = 0; b < bins; ++b) {
    cum = (int8_t)(cum + pmf[(size_t)b * plane + cell]);
    if (target <= cum) return table[b];
  }
  return table[bins - 1];  // malformed PMF (sum < target): the reference leaves the cell unwritten
}

// Philox: one thread per (sample g, row r, 4 consecutive columns): one counter
// block = 4 uniforms; 4-byte-contiguous stores.  out is [G][out_rows][out_stride].
__global__ __launch_bounds__(256) void k_sample_grids_philox(const int8_t* __restrict__ pmf, int bins, int rows,
                                                             int cols, const int8_t* __restrict__ table,
                                                             double alpha_dyn, uint64_t seed, uint64_t epoch,
                                                             int n_grids, int8_t* __restrict__ out, int out_rows,
                                                             int out_stride, uint64_t item_base) {
  // item_base: this handle's samples are [g0, g0 + n_grids) of a larger set (samples sharded over
  // GPUs): g0 * rows * groups, so that the draws are those of the unsharded handle
  const int groups = (cols + 3) / 4;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t total = (size_t)n_grids * rows * groups;
  if (i >= total) return;
  int cg = (int)(i % groups);
  int r = (int)((i / groups) % rows);
  int g = (int)(i / ((size_t)groups * rows));
  rocrand_state_philox4x32_10 st;
  rocrand_init(seed, (uint64_t)i + item_base, 4ULL * epoch, &st);
  float4 uu = rocrand_uniform4(&st);  // (0, 1]
  float uv[4] = {uu.x, uu.y, uu.z, uu.w};
  const size_t plane = (size_t)rows * cols;
  int8_t* o = out + ((size_t)g * out_rows + r) * out_stride;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int c = cg * 4 + k;
    if (c < cols) {
      int8_t target = (int8_t)(int)ceil((double)uv[k] * 100.0 * alpha_dyn);
      o[c] = draw_bin(pmf, (size_t)r * cols + c, plane, bins, table, target);
    }
  }
}

// Philox, column-resident variant (the default): a thread owns 4 consecutive cells of one row
// and draws them for `g_chunk` samples in a row.  The cumulative PMF of its cells is read ONCE
// (the kernel above re-reads it for each of the G samples) and kept packed, 4 cells per
// register; "first bin whose cumulative mass reaches the target" is then evaluated for the 4
// cells at once with byte-parallel arithmetic:
//     bit 7 of ((cum | 0x80) - target)  <=>  cum >= target        (cum, target in [0, 127])
// scanning the bins from the last to the first.  Same distribution and the same int8
// semantics as draw_bin: the cumulative sum wraps like the reference's int8, a wrapped
// (negative) value never matches, and an unmatched cell gets the last bin's value.
// 100 us -> 14 us for 128 samples of a 16-bin 260x260 map.
// cumulative PMF of 4 consecutive cells of row r (columns cg*4 ..), packed one byte per cell,
// and the int8 value each bin maps to, replicated into the 4 bytes
template <int MAXB>
__device__ __forceinline__ void load_packed_thresholds(const int8_t* __restrict__ pmf, int bins, int rows, int cols,
                                                       const int8_t* __restrict__ table, int r, int cg,
                                                       uint32_t (&cum)[MAXB], uint32_t (&val)[MAXB]) {
  const size_t plane = (size_t)rows * cols;
  // all loads first (independent addresses: they pipeline), then the running sums
  uint32_t mass[MAXB];
  if ((cols & 3) == 0) {  // rows are word-aligned: the 4 cells of a bin are one 32-bit load
    const uint32_t* src = reinterpret_cast<const uint32_t*>(pmf + (size_t)r * cols + (size_t)cg * 4);
#pragma unroll
    for (int b = 0; b < MAXB; ++b) mass[b] = (b < bins) ? src[(size_t)b * (plane / 4)] : 0u;
  } else {
#pragma unroll
    for (int b = 0; b < MAXB; ++b) {
      mass[b] = 0;
      if (b < bins) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          mass[b] |= (uint32_t)(uint8_t)pmf[(size_t)b * plane + (size_t)r * cols + min(cg * 4 + k, cols - 1)] << (8 * k);
      }
    }
  }
  uint32_t run = 0;
#pragma unroll
  for (int b = 0; b < MAXB; ++b) {
    // byte-wise run += mass, wrapping like the reference's int8 accumulator
    run = ((run & 0x7f7f7f7fu) + (mass[b] & 0x7f7f7f7fu)) ^ ((run ^ mass[b]) & 0x80808080u);
    // a wrapped (negative) sum never satisfies target <= cum: store 0 for it
    const uint32_t negative = ((run & 0x80808080u) >> 7) * 255u;
    cum[b] = (b < bins ? (run & ~negative) : 0u) | 0x80808080u;  // (bins past the end: 0, never reached)
    val[b] = (b < bins) ? (uint32_t)(uint8_t)table[b] * 0x01010101u : 0u;
  }
}

// The 4 cells of cell group `cgid` in the sample PAIR (2p, 2p+1): one Philox block, each 32-bit
// word split into two 16-bit uniforms (h + 1) / 65536 in (0, 1] -- the targets are integers in
// [1, 100], 16 bits resolve the PMF's percent steps to 1.5e-5 -- so that a block serves 8 draws.
template <int MAXB>
__device__ __forceinline__ void draw_packed_pair(const uint32_t (&cum)[MAXB], const uint32_t (&val)[MAXB], int bins,
                                                 uint32_t last, uint64_t seed, uint64_t epoch, uint64_t pair_index,
                                                 double scale, uint32_t& picked0, uint32_t& picked1) {
  const uint4 x = philox4x32_10(make_uint4((unsigned int)epoch, (unsigned int)(epoch >> 32),
                                           (unsigned int)pair_index, (unsigned int)(pair_index >> 32)),
                                make_uint2((unsigned int)seed, (unsigned int)(seed >> 32)));
  const unsigned int xs[4] = {x.x, x.y, x.z, x.w};
  uint32_t target0 = 0, target1 = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // terrain.py:676: int8(ceil(rnd * 100 * alpha)), rnd uniform in (0, 1]
    const int t0 = (int)(int8_t)(int)ceil((double)((xs[k] & 0xffffu) + 1u) * (1.0 / 65536.0) * scale);
    const int t1 = (int)(int8_t)(int)ceil((double)((xs[k] >> 16) + 1u) * (1.0 / 65536.0) * scale);
    target0 |= (uint32_t)min(max(t0, 0), 127) << (8 * k);
    target1 |= (uint32_t)min(max(t1, 0), 127) << (8 * k);
  }
  picked0 = picked1 = last;
#pragma unroll
  for (int b = MAXB - 1; b >= 0; --b) {
    if (b < bins) {
      const uint32_t r0 = (((cum[b] - target0) & 0x80808080u) >> 7) * 255u;  // 0xFF per cell that matched
      const uint32_t r1 = (((cum[b] - target1) & 0x80808080u) >> 7) * 255u;
      picked0 = (val[b] & r0) | (picked0 & ~r0);
      picked1 = (val[b] & r1) | (picked1 & ~r1);
    }
  }
}

template <int MAXB>
__global__ __launch_bounds__(256) void k_sample_grids_philox_cols(const int8_t* __restrict__ pmf, int bins, int rows,
                                                                  int cols, const int8_t* __restrict__ table,
                                                                  double alpha_dyn, uint64_t seed, uint64_t epoch,
                                                                  int n_grids, int g_chunk, int8_t* __restrict__ out,
                                                                  int out_rows, int out_stride, uint64_t pair_base) {
  // pair_base: (first sample of this handle / 2) * rows * groups (samples sharded over GPUs)
  const int groups = (cols + 3) / 4;
  const int cgid = blockIdx.x * 256 + threadIdx.x;
  if (cgid >= rows * groups) return;
  const int r = cgid / groups, cg = cgid - r * groups;
  uint32_t cum[MAXB], val[MAXB];
  load_packed_thresholds<MAXB>(pmf, bins, rows, cols, table, r, cg, cum, val);
  const uint32_t last = (uint32_t)(uint8_t)table[bins - 1] * 0x01010101u;
  const int g0 = blockIdx.y * g_chunk, g1 = min(g0 + g_chunk, n_grids);  // g_chunk is even
  const double scale = 100.0 * alpha_dyn;
  // hipMalloc'ed base, stride a multiple of 4, group inside the row: one aligned 32-bit store
  const bool word_store = (out_stride & 3) == 0 && cg * 4 + 3 < cols;
  auto store = [&](int g, uint32_t picked) {
    int8_t* o = out + ((size_t)g * out_rows + r) * out_stride + cg * 4;
    if (word_store) {  // uniform except for the last group of a row
      *reinterpret_cast<uint32_t*>(o) = picked;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (cg * 4 + k < cols) o[k] = (int8_t)(picked >> (8 * k));
    }
  };
  for (int g = g0; g < g1; g += 2) {
    const uint64_t pair_index = pair_base + (uint64_t)(g >> 1) * (uint64_t)(rows * groups) + (uint64_t)cgid;
    uint32_t p0, p1;
    draw_packed_pair<MAXB>(cum, val, bins, last, seed, epoch, pair_index, scale, p0, p1);
    store(g, p0);
    if (g + 1 < n_grids) store(g + 1, p1);
  }
}

// The planner's view of the same draws, without the detour through the (G, R, C) int8 grids:
// linear and angular traction of one solve() sampled straight into the cell words the CVaR
// rollout gathers, cellsM[(r*cols + c)*M + m] = lin | ang << 8 | obstacle << 16 | unknown << 24.
// A wave owns one group of 4 cells (its PMF columns are loaded once), a lane the sample pair
// (2*lane, 2*lane + 1) (strided if M > 128): every store is 64 consecutive 8-byte pairs.
// Philox counters are those of k_sample_grids_philox_cols, so the int8 grids can be produced
// later from the same (seed, epoch) when somebody asks for them (mppi_tdm_get_sampled_grids).
template <int MAXB>
__global__ __launch_bounds__(256) void k_sample_cellsM_philox(
    const int8_t* __restrict__ lin_pmf, int lin_bins, const int8_t* __restrict__ lin_table, uint64_t lin_seed,
    uint64_t lin_epoch, const int8_t* __restrict__ ang_pmf, int ang_bins, const int8_t* __restrict__ ang_table,
    uint64_t ang_seed, uint64_t ang_epoch, const int8_t* __restrict__ obs, const int8_t* __restrict__ unk,
    int rows, int cols, double alpha_dyn, int n_grids, uint32_t* __restrict__ cells, uint64_t pair_base) {
  const int groups = (cols + 3) / 4;
  const int cgid = blockIdx.x * 4 + (threadIdx.x >> 6);  // one cell group per wave
  if (cgid >= rows * groups) return;
  const int lane = threadIdx.x & 63;
  const int r = cgid / groups, cg = cgid - r * groups;
  uint32_t lcum[MAXB], lval[MAXB], acum[MAXB], aval[MAXB];
  load_packed_thresholds<MAXB>(lin_pmf, lin_bins, rows, cols, lin_table, r, cg, lcum, lval);
  load_packed_thresholds<MAXB>(ang_pmf, ang_bins, rows, cols, ang_table, r, cg, acum, aval);
  const uint32_t llast = (uint32_t)(uint8_t)lin_table[lin_bins - 1] * 0x01010101u;
  const uint32_t alast = (uint32_t)(uint8_t)ang_table[ang_bins - 1] * 0x01010101u;
  uint32_t flags[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t ci = (size_t)r * cols + min(cg * 4 + k, cols - 1);
    flags[k] = ((uint32_t)(uint8_t)obs[ci] << 16) | ((uint32_t)(uint8_t)unk[ci] << 24);
  }
  const double scale = 100.0 * alpha_dyn;
  const bool pair_store = (n_grids & 1) == 0;  // rows of cellsM are 8-byte aligned
  for (int pr = lane; 2 * pr < n_grids; pr += 64) {
    const uint64_t pair_index = pair_base + (uint64_t)pr * (uint64_t)(rows * groups) + (uint64_t)cgid;
    uint32_t l0, l1, a0, a1;
    draw_packed_pair<MAXB>(lcum, lval, lin_bins, llast, lin_seed, lin_epoch, pair_index, scale, l0, l1);
    draw_packed_pair<MAXB>(acum, aval, ang_bins, alast, ang_seed, ang_epoch, pair_index, scale, a0, a1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = cg * 4 + k;
      if (c < cols) {
        const uint32_t w0 = ((l0 >> (8 * k)) & 0xffu) | (((a0 >> (8 * k)) & 0xffu) << 8) | flags[k];
        const uint32_t w1 = ((l1 >> (8 * k)) & 0xffu) | (((a1 >> (8 * k)) & 0xffu) << 8) | flags[k];
        uint32_t* o = cells + ((size_t)r * cols + c) * n_grids + 2 * pr;
        if (pair_store) {
          *reinterpret_cast<uint2*>(o) = make_uint2(w0, w1);
        } else {
          o[0] = w0;
          if (2 * pr + 1 < n_grids) o[1] = w1;
        }
      }
    }
  }
}

// reference-compatible: launch geometry [(1,G),(tx,ty)] flattened; thread (i,j)
// of block g owns stream i*ty*G + g*ty + j and walks its tile row-major
// (terrain.py:645-668)
__global__ void k_sample_grids_xoroshiro(const int8_t* __restrict__ pmf, int bins, int rows, int cols,
                                         const int8_t* __restrict__ table, double alpha_dyn,
                                         uint64_t* __restrict__ states, int n_grids, int tx, int ty,
                                         int8_t* __restrict__ out, int out_rows, int out_stride) {
  int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= n_grids * tx * ty) return;
  int j = id % ty, i = (id / ty) % tx, g = id / (tx * ty);
  size_t stream = (size_t)i * ty * n_grids + (size_t)g * ty + j;
  uint64_t s0 = states[2 * stream], s1 = states[2 * stream + 1];
  int nr = (rows + tx - 1) / tx, nc = (cols + ty - 1) / ty;
  int r0 = min(i * nr, rows), r1 = min(r0 + nr, rows);
  int c0 = min(j * nc, cols), c1 = min(c0 + nc, cols);
  const size_t plane = (size_t)rows * cols;
  for (int r = r0; r < r1; ++r)
    for (int c = c0; c < c1; ++c) {
      float rnd = xoroshiro_uniform_f32(s0, s1);
      int8_t target = (int8_t)(long)ceil((double)rnd * 100.0 * alpha_dyn);
      int8_t cum = 0;
      for (int b = 0; b < bins; ++b) {
        cum = (int8_t)(cum + pmf[(size_t)b * plane + (size_t)r * cols + c]);
        if (target <= cum) {
          out[((size_t)g * out_rows + r) * out_stride + c] = table[b];
          break;
        }
      }
    }
  states[2 * stream] = s0;
  states[2 * stream + 1] = s1;
}

// ---------------- map packing ----------------
// cells[r*cols+c] = lin | ang<<8 | obs<<16 | unk<<24 from sample 0 of each TDM
__global__ void k_pack_cells_single(const int8_t* __restrict__ lin, const int8_t* __restrict__ ang, int grid_stride,
                                    const int8_t* __restrict__ obs, const int8_t* __restrict__ unk, int rows,
                                    int cols, uint32_t* __restrict__ cells) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  int r = i / cols, c = i % cols;
  uint32_t l = (uint8_t)lin[(size_t)r * grid_stride + c], a = (uint8_t)ang[(size_t)r * grid_stride + c];
  uint32_t o = (uint8_t)obs[i], k = (uint8_t)unk[i];
  cells[i] = l | (a << 8) | (o << 16) | (k << 24);
}

// 16-bit variant for the LDS-resident window: lin | ang<<7 | obs<<14 | unk<<15,
// row pitch `pitch` (multiple of 8 cells so that rows start 16-byte aligned)
__global__ void k_pack_cells16(const int8_t* __restrict__ lin, const int8_t* __restrict__ ang, int grid_stride,
                               const int8_t* __restrict__ obs, const int8_t* __restrict__ unk, int rows,
                               int cols, int pitch, uint16_t* __restrict__ cells16) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * pitch) return;
  int r = i / pitch, c = i % pitch;
  uint32_t v = 0;
  if (c < cols) {
    uint32_t l = (uint8_t)lin[(size_t)r * grid_stride + c] & 127u, a = (uint8_t)ang[(size_t)r * grid_stride + c] & 127u;
    uint32_t o = (uint8_t)obs[(size_t)r * cols + c] & 1u, k = (uint8_t)unk[(size_t)r * cols + c] & 1u;
    v = l | (a << 7) | (o << 14) | (k << 15);
  }
  cells16[i] = (uint16_t)v;
}

// speed-map mode: the same 16 bits plus the risk traction byte in bits 16..23 (32-bit cells,
// same pitch in cells), for the LDS window of k_rollout_fused<.., SPEED>
__global__ void k_pack_cells32_risk(const int8_t* __restrict__ lin, const int8_t* __restrict__ ang, int grid_stride,
                                    const int8_t* __restrict__ obs, const int8_t* __restrict__ unk,
                                    const int8_t* __restrict__ risk, int rows, int cols, int pitch,
                                    uint32_t* __restrict__ cells32) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * pitch) return;
  int r = i / pitch, c = i % pitch;
  uint32_t v = 0;
  if (c < cols) {
    uint32_t l = (uint8_t)lin[(size_t)r * grid_stride + c] & 127u, a = (uint8_t)ang[(size_t)r * grid_stride + c] & 127u;
    uint32_t o = (uint8_t)obs[(size_t)r * cols + c] & 1u, k = (uint8_t)unk[(size_t)r * cols + c] & 1u;
    v = l | (a << 7) | (o << 14) | (k << 15) | ((uint32_t)(uint8_t)risk[(size_t)r * cols + c] << 16);
  }
  cells32[i] = v;
}

// cellsM[(r*cols+c)*M + m]: transpose (M,R,C) -> (R,C,M) through an LDS tile
// of 64 samples x 64 cells so that both the byte reads (along c) and the word
// writes (along m) are contiguous.
__global__ __launch_bounds__(256) void k_pack_cells_multi(const int8_t* __restrict__ lin,
                                                          const int8_t* __restrict__ ang, int grid_rows,
                                                          int grid_stride, const int8_t* __restrict__ obs,
                                                          const int8_t* __restrict__ unk, int rows, int cols,
                                                          int n_grids, uint32_t* __restrict__ cells) {
  __shared__ uint16_t tile[64][65];
  const int col_tiles = (cols + 63) / 64;
  const int c0 = (blockIdx.x % col_tiles) * 64;
  const int r = blockIdx.x / col_tiles;
  const int m0 = blockIdx.y * 64;
  const size_t plane = (size_t)grid_rows * grid_stride;
  // read: thread (q, lane) with lane = cell within the tile, q steps over samples
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  for (int mm = q; mm < 64; mm += 4) {
    int m = m0 + mm, c = c0 + lane;
    uint16_t v = 0;
    if (m < n_grids && c < cols) {
      size_t off = plane * m + (size_t)r * grid_stride + c;
      v = (uint16_t)((uint8_t)lin[off] | ((uint16_t)(uint8_t)ang[off] << 8));
    }
    tile[mm][lane] = v;
  }
  __syncthreads();
  // write: lane = sample within the tile, q steps over cells
  for (int cc = q; cc < 64; cc += 4) {
    int c = c0 + cc, m = m0 + lane;
    if (m < n_grids && c < cols) {
      size_t ci = (size_t)r * cols + c;
      uint32_t o = (uint8_t)obs[ci], k = (uint8_t)unk[ci];
      cells[ci * n_grids + m] = (uint32_t)tile[lane][cc] | (o << 16) | (k << 24);
    }
  }
}

}  // namespace mppi
