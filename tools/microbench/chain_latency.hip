// Latency of DEPENDENT instruction chains for one wave (gfx950) -- what a "walk" of the time-parallel
// exact kernel pays per step (rollout_scan_exact_kernel.h): every instruction waits for the one before.
//   hipcc --offload-arch=gfx950 -O3 -o chain_latency chain_latency.hip && ./chain_latency
// Columns: the wave alone on its SIMD | with one / three other waves on the SAME SIMD that issue
// independent integer work at lower priority (the chunk waves beside a walker).
#include <hip/hip_runtime.h>
#include <cstdio>

#define STEPS 512

// KIND: which chain.  `busy` waves (wave index >= 4, same SIMD as wave 0 when index % 4 == 0) spin on
// integer multiply-adds until wave 0 has finished.
template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, double c, double inc, double K, float pf, int busy_on_simd0, int busy_kind = 0) {
  __shared__ float sink[64 * 8];
  __shared__ int stop;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) stop = 0;
  __syncthreads();
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    double v = 1.0 + lane * 0.125, t, s;
    float vf = 1.0f + lane, o = pf, q = pf;
    const unsigned addr = lane * 4;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < STEPS; ++i) {
      if (KIND == 0) {  // the theta / x / y walk as it is: fma, cvt, cvt (+ LDS store of the float)
        asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v) : "v"(c), "v"(inc));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(vf) : "v"(v));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v) : "v"(vf));
        asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(vf) : "memory");
      }
      if (KIND == 1) {  // rounding to 24 bits by two fma: s = t*K + t, r = s - t*K   (K = 2^29)
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(t) : "v"(c), "v"(inc), "v"(v));
        asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(s) : "v"(t), "v"(K));
        asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(v) : "v"(t), "v"(K), "v"(s));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(vf) : "v"(v));  // off the chain
        asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(vf) : "memory");
      }
      if (KIND == 2) {  // clean cost walk: add, cvt, cvt
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(inc));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(vf) : "v"(v));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v) : "v"(vf));
      }
      if (KIND == 3) {  // ... with the fma rounding
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(t) : "v"(v), "v"(inc));
        asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(s) : "v"(t), "v"(K));
        asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(v) : "v"(t), "v"(K), "v"(s));
      }
      if (KIND == 4) {  // cost walk with penalties: cvt, add64, cvt, add32, add32
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v) : "v"(vf));
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(inc));
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(vf) : "v"(v));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(vf) : "v"(o));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(vf) : "v"(q));
      }
      if (KIND == 5) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v) : "v"(c), "v"(inc));
      if (KIND == 6) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(inc));
      if (KIND == 7) {
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(vf) : "v"(v));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v) : "v"(vf));
      }
      if (KIND == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(vf) : "v"(o));
      if (KIND == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v) : "v"(c));
      if (KIND == 10) {  // penalties in float64 with the fma rounding: 3 x (add, fma, fma)
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(t) : "v"(v), "v"(inc));
        asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(s) : "v"(t), "v"(K));
        asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(v) : "v"(t), "v"(K), "v"(s));
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(t) : "v"(v), "v"(c));
        asm volatile("v_fma_f64 %0, %1, %2, %1" : "=v"(s) : "v"(t), "v"(K));
        asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(v) : "v"(t), "v"(K), "v"(s));
      }
      if (KIND == 11) {  // v_ldexp-free variant: t + M - M with a constant M (binade known)
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(t) : "v"(c), "v"(inc), "v"(v));
        asm volatile("v_add_f64 %0, %1, %2" : "=v"(s) : "v"(t), "v"(K));
        asm volatile("v_add_f64 %0, %1, -%2" : "=v"(v) : "v"(s), "v"(K));
      }
      if (KIND == 12) {  // float32 add chain with an LDS store per step
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(vf) : "v"(o));
        asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(vf) : "memory");
      }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) {
      out[0] = t1 - t0;
      out[1] = (unsigned long long)(v + vf);
      __hip_atomic_store(&stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  } else if ((wave & 3) == 0 ? (wave / 4 <= busy_on_simd0) : false) {
    // a busy neighbour on SIMD 0, lower priority.  busy_kind 0: independent integer chains (Philox-like);
    // 1: independent float64 fma streams; 2: float64 sqrt / rcp (transcendental unit) ; 3: LDS reads and writes
    unsigned a = lane, b = lane * 3, cc = lane * 7, d = lane * 11;
    double f0 = 1.0 + lane, f1 = 2.0 + lane, f2 = 3.0 + lane, f3 = 4.0 + lane;
    while (__hip_atomic_load(&stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
      if (busy_kind == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          a = a * 0x9E3779B9u + b;
          b = b * 0xBB67AE85u + cc;
          cc = cc * 0xD2511F53u + d;
          d = d * 0xCD9E8D57u + a;
        }
      } else if (busy_kind == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f0) : "v"(c), "v"(inc));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f1) : "v"(c), "v"(inc));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f2) : "v"(c), "v"(inc));
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f3) : "v"(c), "v"(inc));
        }
      } else if (busy_kind == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          asm volatile("v_sqrt_f64 %0, %0" : "+v"(f0));
          asm volatile("v_rcp_f64 %0, %0" : "+v"(f1));
          asm volatile("v_sqrt_f64 %0, %0" : "+v"(f2));
          asm volatile("v_rcp_f64 %0, %0" : "+v"(f3));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          sink[(lane + 64 * (i & 3)) & 511] = (float)f0;
          f0 += (double)sink[(lane * 2 + i) & 511];
        }
      }
    }
    sink[lane] = (float)(a ^ b ^ cc ^ d) + (float)(f0 + f1 + f2 + f3);
  }
}

template <int KIND>
static double run(unsigned long long* dev, int busy, int kind = 0) {
  // 16 waves: wave w sits on SIMD w % 4; waves 4, 8, 12 share SIMD 0 with the measured wave
  hipLaunchKernelGGL((k<KIND>), dim3(1), dim3(1024), 0, 0, dev, 1.0000001, 1e-3, 536870912.0, 0.5f, busy, kind);
  unsigned long long h[2];
  hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
  return (double)h[0] / STEPS;
}

int main() {
  unsigned long long* dev;
  hipMalloc(&dev, 16);
  const char* names[] = {"walk now: fma64 cvt cvt +ds_write",
                         "walk new: fma64 fma64 fma64 (+cvt, ds_write off chain)",
                         "cost clean now: add64 cvt cvt",
                         "cost clean new: add64 fma64 fma64",
                         "cost penalties now: cvt add64 cvt add32 add32",
                         "fma64 alone",
                         "add64 alone",
                         "cvt_f32_f64 + cvt_f64_f32",
                         "add32 alone",
                         "mul64 alone",
                         "cost penalties new: 2 x (add64 fma64 fma64)",
                         "walk, constant M: fma64 add64 add64",
                         "add32 + ds_write"};
  printf("%-56s %9s %9s %9s   (cycles per step, s_memtime/readcyclecounter units)\n", "chain", "alone", "+1 busy", "+3 busy");
#define ROW(K) printf("%-56s %9.1f %9.1f %9.1f\n", names[K], run<K>(dev, 0), run<K>(dev, 1), run<K>(dev, 3));
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12)
  printf("\nthe same chains beside THREE busy waves of another kind (lower priority, same SIMD): float64 fma | float64 sqrt, rcp | LDS\n");
#define ROW2(K) printf("%-56s %9.1f %9.1f %9.1f\n", names[K], run<K>(dev, 3, 1), run<K>(dev, 3, 2), run<K>(dev, 3, 3));
  ROW2(0) ROW2(2) ROW2(4) ROW2(12)
  // clock: readcyclecounter ticks per microsecond
  {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipEventRecord(a);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((k<0>), dim3(1), dim3(1024), 0, 0, dev, 1.0000001, 1e-3, 536870912.0, 0.5f, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2];
    hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
    printf("one launch of kind 0: %.2f us wall per launch, %llu ticks inside\n", ms * 1e3 / 50, h[0]);
  }
  return 0;
}
