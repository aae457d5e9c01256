// update_kernels.h -- min-subtract + exp-weighted control update (gfx950).
//
// Replaces update_useq_numba[1, 32] (mppi.py:1113-1191): one 32-thread block
// doing N*T*2 float32 atomics.  Here:
//   k_weights   grid N/256   beta = min cost, w_n = exp(-(c_n-beta)/lambda),
//                            per-block partial sums of w (float64)
//   k_wsum      grid (T,NCH) partial[ch][t] = sum_{n in chunk} w_n * eps[t][n]
//                            (float64 accumulation, wave-shuffle reductions)
//   k_finish    grid 1       fixed-order sums -> packet {beta, den, num[T][2]}
//   k_apply     grid 1       combine the packets of all GPUs, u += num/den, clip
// Everything is a deterministic tree (no atomics): the same inputs give the
// same bits on every run, and the result does not depend on the GPU count
// beyond float64 rounding of the partial sums.
//
// Exactness of the split across GPUs: with beta = min_g beta_g,
//   sum_n exp(-(c_n-beta)/l) x_n = sum_g exp(-(beta_g-beta)/l) * sum_{n in g} exp(-(c_n-beta_g)/l) x_n
#pragma once
#include "device_math.h"

namespace mppi {

constexpr int kUpdateThreads = 256;

// packet layout (doubles): [0] beta_g  [1] den_g  [2 + 2t + c] num_g[t][c]
__host__ __device__ inline int packet_len(int n_steps) { return 2 + 2 * n_steps; }

__global__ __launch_bounds__(kUpdateThreads) void k_block_min_from_costs(const float* __restrict__ costs, int n,
                                                                         float* __restrict__ block_min) {
  __shared__ float red[kUpdateThreads / 64];
  int i = blockIdx.x * kUpdateThreads + threadIdx.x;
  float v = (i < n) ? costs[i] : __builtin_inff();
  v = wave_min_f32(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = red[0];
    for (int k = 1; k < kUpdateThreads / 64; ++k) m = fminf(m, red[k]);
    block_min[blockIdx.x] = m;
  }
}

// weights[n] = float32(exp(-1/lambda * float32(c_n - beta)))      (mppi.py:1152-1154)
// den_part[b] = sum over the block's weights (float64)
__global__ __launch_bounds__(kUpdateThreads) void k_weights(const float* __restrict__ costs, int n,
                                                            const float* __restrict__ block_min, int n_min,
                                                            float lambda, float* __restrict__ weights,
                                                            double* __restrict__ den_part,
                                                            double* __restrict__ packet) {
  __shared__ float redf[kUpdateThreads / 64];
  __shared__ double redd[kUpdateThreads / 64];
  float m = __builtin_inff();
  for (int i = threadIdx.x; i < n_min; i += kUpdateThreads) m = fminf(m, block_min[i]);
  m = wave_min_f32(m);
  if ((threadIdx.x & 63) == 0) redf[threadIdx.x >> 6] = m;
  __syncthreads();
  float beta = redf[0];
  for (int k = 1; k < kUpdateThreads / 64; ++k) beta = fminf(beta, redf[k]);

  int i = blockIdx.x * kUpdateThreads + threadIdx.x;
  float w = 0.0f;
  if (i < n) {
    double neg_inv_lambda = -1.0 / (double)lambda;
    w = (float)exp(neg_inv_lambda * (double)(costs[i] - beta));
    weights[i] = w;
  }
  double s = wave_sum_f64((double)w);
  if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = redd[0];
    for (int k = 1; k < kUpdateThreads / 64; ++k) tot += redd[k];
    den_part[blockIdx.x] = tot;
    if (blockIdx.x == 0) packet[0] = (double)beta;
  }
}

// partial[ch][t] = sum_{n in chunk ch} w_n * eps[t][n]  (noise is [T][N] float2:
// the block streams 8-byte elements, coalesced)
__global__ __launch_bounds__(kUpdateThreads) void k_wsum(const float* __restrict__ weights,
                                                         const float2* __restrict__ noise, int n, int chunk,
                                                         double2* __restrict__ partial) {
  __shared__ double2 red[kUpdateThreads / 64];
  const int t = blockIdx.x, ch = blockIdx.y;
  const int lo = ch * chunk;
  const int hi = min(lo + chunk, n);
  const float2* row = noise + (size_t)t * n;
  double ax = 0.0, ay = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += kUpdateThreads) {
    double w = (double)weights[i];
    float2 e = row[i];
    ax = fma(w, (double)e.x, ax);
    ay = fma(w, (double)e.y, ay);
  }
  ax = wave_sum_f64(ax);
  ay = wave_sum_f64(ay);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = make_double2(ax, ay);
  __syncthreads();
  if (threadIdx.x == 0) {
    double2 tot = red[0];
    for (int k = 1; k < kUpdateThreads / 64; ++k) {
      tot.x += red[k].x;
      tot.y += red[k].y;
    }
    partial[(size_t)ch * gridDim.x + t] = tot;
  }
}

__device__ __forceinline__ double block_sum_fixed_order(const double* __restrict__ v, int count, double* red) {
  // each thread sums a strided subset sequentially, then a fixed 64-lane
  // butterfly and a fixed 4-way sum: deterministic
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += kUpdateThreads) s += v[i];
  s = wave_sum_f64(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  double tot = red[0];
  for (int k = 1; k < kUpdateThreads / 64; ++k) tot += red[k];
  __syncthreads();
  return tot;
}

// u[t] = clip(u[t] + num[t]/den); u_prev mirrors it (the reference aliases
// u_prev_d to u_cur_d before the update, mppi.py:362)
__device__ __forceinline__ void apply_update(float2* u, float2* u_prev, int t, double nx, double ny, double den,
                                             float v_lo, float v_hi, float w_lo, float w_hi) {
  float2 ut = u[t];
  ut.x = clip_f32(ut.x + (float)(nx / den), v_lo, v_hi);
  ut.y = clip_f32(ut.y + (float)(ny / den), w_lo, w_hi);
  u[t] = ut;
  u_prev[t] = ut;
}

// fixed-order reduction of the partials into this GPU's packet; with a single
// GPU (APPLY) the update is applied in the same launch
template <bool APPLY>
__global__ __launch_bounds__(kUpdateThreads) void k_finish(const double2* __restrict__ partial, int n_chunks,
                                                           const double* __restrict__ den_part, int n_den,
                                                           int n_steps, double* __restrict__ packet,
                                                           float2* __restrict__ u, float2* __restrict__ u_prev,
                                                           float v_lo, float v_hi, float w_lo, float w_hi,
                                                           double* __restrict__ weight_scale) {
  __shared__ double red[kUpdateThreads / 64];
  double den = block_sum_fixed_order(den_part, n_den, red);
  if (threadIdx.x == 0) {
    packet[1] = den;
    if (APPLY) *weight_scale = 1.0 / den;
  }
  for (int t = threadIdx.x; t < n_steps; t += kUpdateThreads) {
    double nx = 0.0, ny = 0.0;
    for (int ch = 0; ch < n_chunks; ++ch) {
      double2 p = partial[(size_t)ch * n_steps + t];
      nx += p.x;
      ny += p.y;
    }
    packet[2 + 2 * t] = nx;
    packet[3 + 2 * t] = ny;
    if (APPLY) apply_update(u, u_prev, t, nx, ny, den, v_lo, v_hi, w_lo, w_hi);
  }
}

// combine the packets of all ranks (identical on every GPU, fixed g order)
__global__ __launch_bounds__(kUpdateThreads) void k_apply(const double* __restrict__ packets, int world,
                                                          int rank, int n_steps, float lambda,
                                                          float2* __restrict__ u, float2* __restrict__ u_prev,
                                                          float v_lo, float v_hi, float w_lo, float w_hi,
                                                          double* __restrict__ weight_scale) {
  const int len = packet_len(n_steps);
  double beta = packets[0];
  for (int g = 1; g < world; ++g) beta = fmin(beta, packets[(size_t)g * len]);
  const double neg_inv_lambda = -1.0 / (double)lambda;
  double den = 0.0;
  for (int g = 0; g < world; ++g)
    den += exp(neg_inv_lambda * (packets[(size_t)g * len] - beta)) * packets[(size_t)g * len + 1];
  if (threadIdx.x == 0)
    *weight_scale = exp(neg_inv_lambda * (packets[(size_t)rank * len] - beta)) / den;
  for (int t = threadIdx.x; t < n_steps; t += kUpdateThreads) {
    double nx = 0.0, ny = 0.0;
    for (int g = 0; g < world; ++g) {
      double sg = exp(neg_inv_lambda * (packets[(size_t)g * len] - beta));
      nx = fma(sg, packets[(size_t)g * len + 2 + 2 * t], nx);
      ny = fma(sg, packets[(size_t)g * len + 3 + 2 * t], ny);
    }
    apply_update(u, u_prev, t, nx, ny, den, v_lo, v_hi, w_lo, w_hi);
  }
}

// normalised weights for the host (weights_d of the reference): w * scale
__global__ void k_scale_weights(const float* __restrict__ w, const double* __restrict__ scale, int n,
                                float* __restrict__ out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)((double)w[i] * (*scale));
}

// device-side shift of the control sequence: u[:-k] = u[k:], tail kept (mppi.py:539-541)
__global__ void k_shift_u(float2* __restrict__ u, int n_steps, int k) {
  // single block: read everything, barrier, write
  extern __shared__ float2 tmp[];
  for (int t = threadIdx.x; t < n_steps; t += blockDim.x) tmp[t] = u[t];
  __syncthreads();
  for (int t = threadIdx.x; t + k < n_steps; t += blockDim.x) u[t] = tmp[t + k];
}

// host layout (N,T,2) <-> device layout [T][N] float2
__global__ void k_noise_to_device_layout(const float2* __restrict__ host_layout, int n, int t_steps,
                                         float2* __restrict__ dev_layout) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * t_steps) return;
  int t = (int)(i / n), r = (int)(i % n);
  dev_layout[i] = host_layout[(size_t)r * t_steps + t];
}
__global__ void k_noise_to_host_layout(const float2* __restrict__ dev_layout, int n, int t_steps,
                                       float2* __restrict__ host_layout) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)n * t_steps) return;
  int r = (int)(i / t_steps), t = (int)(i % t_steps);
  host_layout[i] = dev_layout[(size_t)t * n + r];
}

}  // namespace mppi
