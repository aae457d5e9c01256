// Can EVERY workgroup of a launch combine the previous launch's tile packets by itself (one trip through memory)
// sooner than "workgroup t combines step t, publishes, everyone collects" (two trips: 13k cycles until u is in LDS)?
// 256 workgroups x 16 waves; RW "reduce" waves stream all 256 packets x 202 floats (lane = element, tiles in slices),
// the other waves run Philox-like integer work (the chunk waves' noise).  Stamps: cycles from the workgroup's entry.
//   hipcc --offload-arch=gfx950 -O3 -o stream_combine stream_combine.hip && ./stream_combine
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int NT = 256, E = 202, STRIDE = 202;

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

template <int RW, int DEPTH, bool BUSY>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ in, float* __restrict__ out, float lambda,
                                          unsigned long long* stamps, int tail_spin) {
  __shared__ double scales[NT];
  __shared__ double partial[16][4 * 64];
  __shared__ int flag_scales, flag_done;
  __shared__ float u_sh[2 * 104];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long t0 = now();
  if (threadIdx.x == 0) { flag_scales = 0; flag_done = 0; }
  // reduce waves: 0, 1, 2 (one per SIMD), then 3, 4, ...
  const bool reducer = wave < RW;
  float acc_out = 0.0f;
  if (reducer) {
    __builtin_amdgcn_s_setprio(3);
    constexpr int PER = NT / RW;  // tiles of this wave
    const int first = wave * PER;
    // the first batch of loads is requested before anything else
    float v[DEPTH][4];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[d][i] = in[(size_t)(first + d) * STRIDE + min(lane + 64 * i, E - 1)];
    __syncthreads();
    if (wave == 0) {
      float2 bd[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bd[q] = *reinterpret_cast<const float2*>(in + (size_t)(lane + 64 * q) * STRIDE);
      float bm = fminf(fminf(bd[0].x, bd[1].x), fminf(bd[2].x, bd[3].x));
      for (int o = 32; o > 0; o >>= 1) bm = fminf(bm, __shfl_xor(bm, o));
      if (lane == 0) stamps[blockIdx.x * 16 + 1] = now() - t0;
      const float sc = -1.4426950408889634f / lambda;
#pragma unroll
      for (int q = 0; q < 4; ++q) scales[lane + 64 * q] = (double)__builtin_amdgcn_exp2f((bd[q].x - bm) * sc);
      __hip_atomic_store(&flag_scales, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (lane == 0) stamps[blockIdx.x * 16 + 2] = now() - t0;
    }
    while (__hip_atomic_load(&flag_scales, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = 0; b < PER; b += DEPTH) {
      float nv[DEPTH][4];
      if (b + DEPTH < PER) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
          for (int i = 0; i < 4; ++i) nv[d][i] = in[(size_t)(first + b + DEPTH + d) * STRIDE + min(lane + 64 * i, E - 1)];
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const double s = scales[first + b + d];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = fma(s, (double)v[d][i], a[i]);
      }
#pragma unroll
      for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[d][i] = nv[d][i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) partial[wave][lane + 64 * i] = a[i];
    if (lane == 0) {
      stamps[blockIdx.x * 16 + 4 + wave] = now() - t0;
      __hip_atomic_fetch_add(&flag_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (wave == 0) {
      while (__hip_atomic_load(&flag_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < RW) __builtin_amdgcn_s_sleep(1);
      double tot[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tot[i] = partial[0][lane + 64 * i];
        for (int w = 1; w < RW; ++w) tot[i] += partial[w][lane + 64 * i];
        partial[0][lane + 64 * i] = tot[i];
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      // controls: lane = step (two rounds): num / den
      const double den = partial[0][1];
      for (int t = lane; t < 100; t += 64) {
        u_sh[2 * t] = (float)(partial[0][2 + 2 * t] / den);
        u_sh[2 * t + 1] = (float)(partial[0][3 + 2 * t] / den);
      }
      if (lane == 0) stamps[blockIdx.x * 16 + 3] = now() - t0;
      acc_out = u_sh[lane];
    }
  } else {
    __syncthreads();
    if (BUSY) {  // Philox-like: ~1.5k cycles of integer issue per wave
      unsigned a = lane + wave, b = lane * 3, c = lane * 7, d = lane * 11;
#pragma unroll 4
      for (int i = 0; i < 96; ++i) {
        a = a * 0x9E3779B9u + b;
        b = __umulhi(b, 0xBB67AE85u) ^ c;
        c = c * 0xD2511F53u + d;
        d = __umulhi(d, 0xCD9E8D57u) ^ a;
      }
      acc_out = (float)(a ^ b ^ c ^ d) * 1e-12f;
      if (lane == 0 && wave == 15) stamps[blockIdx.x * 16 + 12] = now() - t0;
    }
  }
  __syncthreads();
  // the "rollout": spin, then this workgroup's packet for the next launch
  for (int i = 0; i < tail_spin; ++i) acc_out = acc_out * 1.0000001f + 1e-9f;
  if (threadIdx.x < E) {
    float val = 0.001f * (float)((threadIdx.x * 7 + blockIdx.x) % 97) + acc_out * 1e-20f;
    if (threadIdx.x == 0) val = 100.0f + (float)(blockIdx.x % 13);
    out[(size_t)blockIdx.x * STRIDE + threadIdx.x] = val;
  }
  if (threadIdx.x == 0) stamps[blockIdx.x * 16 + 13] = now() - t0;
}

template <int RW, int DEPTH, bool BUSY>
static void run(const char* name, float* bufs[2], unsigned long long* stamps_d) {
  std::vector<unsigned long long> h(NT * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<RW, DEPTH, BUSY>), dim3(NT), dim3(1024), 0, 0, bufs[i & 1], bufs[(i & 1) ^ 1], 1.0f, stamps_d, 2000);
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<RW, DEPTH, BUSY>), dim3(NT), dim3(1024), 0, 0, bufs[i & 1], bufs[(i & 1) ^ 1], 1.0f, stamps_d, 2000);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h.data(), stamps_d, h.size() * 8, hipMemcpyDeviceToHost);
  auto col = [&](int c) { std::vector<unsigned long long> v; for (int b = 0; b < NT; ++b) v.push_back(h[b * 16 + c]); std::sort(v.begin(), v.end()); return v; };
  auto m = col(1), s = col(2), u = col(3), p0 = col(4), pl = col(4 + RW - 1), ph = col(12), en = col(13);
  printf("%-28s minima back %5llu/%5llu  scales %5llu  partial w0 %5llu  last %5llu  u ready min %5llu med %5llu max %5llu  philox %5llu  end %6llu   %.2f us/launch\n",
         name, m[NT / 2], m[NT - 1], s[NT / 2], p0[NT / 2], pl[NT / 2], u[0], u[NT / 2], u[NT - 1], ph[NT / 2], en[NT / 2], ms * 1e3 / 50);
}

int main() {
  float* bufs[2];
  unsigned long long* stamps;
  hipMalloc(&bufs[0], NT * STRIDE * 4); hipMalloc(&bufs[1], NT * STRIDE * 4);
  hipMalloc(&stamps, NT * 16 * 8);
  hipMemset(stamps, 0, NT * 16 * 8);
  std::vector<float> init(NT * STRIDE, 0.5f);
  for (int b = 0; b < NT; ++b) init[b * STRIDE] = 100.0f + b % 13;
  hipMemcpy(bufs[0], init.data(), init.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(bufs[1], init.data(), init.size() * 4, hipMemcpyHostToDevice);
  run<1, 8, true>("1 wave depth 8 busy", bufs, stamps);
  run<2, 8, true>("2 waves depth 8 busy", bufs, stamps);
  run<4, 8, true>("4 waves depth 8 busy", bufs, stamps);
  run<4, 8, false>("4 waves depth 8 idle", bufs, stamps);
  run<4, 4, true>("4 waves depth 4 busy", bufs, stamps);
  run<4, 16, true>("4 waves depth 16 busy", bufs, stamps);
  run<8, 8, true>("8 waves depth 8 busy", bufs, stamps);
  run<8, 4, true>("8 waves depth 4 busy", bufs, stamps);
  run<8, 8, false>("8 waves depth 8 idle", bufs, stamps);
  run<16, 4, false>("16 waves depth 4", bufs, stamps);
  run<16, 8, false>("16 waves depth 8", bufs, stamps);
  return 0;
}
