// Issue cost of single instructions for ONE wave alone on its SIMD (gfx950): N independent
// register chains of the same instruction, timed with s_memtime.  Developer tool:
//   hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64

template <int KIND, int CHAINS>
__global__ void k(unsigned long long* out, double seed, float fseed) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  double d[8]; float f[8]; int iv[8]; f32x4 q[4];
  for (int i = 0; i < 4; ++i) q[i] = (f32x4){fseed, fseed, fseed, fseed};
  for (int i = 0; i < 8; ++i) { d[i] = seed + i + threadIdx.x; f[i] = fseed + i + threadIdx.x; iv[i] = i + threadIdx.x; }
  __shared__ double lds[8 * 64 * 4];
  lds[threadIdx.x] = seed;
  __syncthreads();
  unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
#pragma unroll
    for (int ii = 0; ii < 16 * CHAINS; ++ii) {
      const int i = ii % CHAINS;
      if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(seed));
      if (KIND == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(seed));
      if (KIND == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(seed));
      if (KIND == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
      if (KIND == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
      if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(fseed));
      if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(fseed));
      if (KIND == 7) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(iv[i]));
      if (KIND == 8) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(fseed));
      if (KIND == 9) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
      if (KIND == 10) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(iv[i]) : "v"(iv[(i + 1) & 7]));
      if (KIND == 11) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(iv[i]) : "v"(f[i]));
      if (KIND == 12) { asm volatile("ds_read_b128 %0, %1" : "=v"(q[i & 3]) : "v"((threadIdx.x & 63) * 16 + (i & 1) * 1024)); }
      if (KIND == 13) { asm volatile("ds_write_b128 %1, %0" : : "v"(q[i & 3]), "v"((threadIdx.x & 63) * 16 + (i & 1) * 1024) : "memory"); }
      if (KIND == 14) { asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"((threadIdx.x & 63) * 8 + (i & 1) * 512)); }
      if (KIND == 15) { asm volatile("ds_write_b64 %1, %0" : : "v"(d[i]), "v"((threadIdx.x & 63) * 8 + (i & 1) * 512) : "memory"); }
      if (KIND == 16) { asm volatile("ds_read_u16 %0, %1" : "=v"(iv[i]) : "v"(((threadIdx.x * 37 + i * 11) & 1023) * 2)); }
    }
    if (KIND >= 12) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  double acc = 0; for (int i = 0; i < 8; ++i) acc += d[i] + f[i] + iv[i] + q[i & 3].x;
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)acc; }
}

template <int KIND, int CHAINS>
double run(unsigned long long* dev, int waves) {
  hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(1), dim3(64 * waves), 0, 0, dev, 1.0000001, 1.00001f);
  unsigned long long h[2];
  hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
  return (double)h[0] / (REP * CHAINS * 16);
}

int main() {
  unsigned long long* dev;
  hipMalloc(&dev, 16);
  const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_fma_f32", "v_cndmask_b32",
                         "v_cvt_f64_i32", "v_med3_f32", "v_sqrt_f32", "v_mad_u32_u24", "v_cvt_i32_f32", "ds_read_b128", "ds_write_b128",
                         "ds_read_b64", "ds_write_b64", "ds_read_u16(rand)"};
  printf("%-18s %10s %10s %10s | 4 waves (1/SIMD) 8 chains | 8 waves (2/SIMD) 8 chains\n", "instr", "1 chain", "4 chains", "8 chains");
#define ROW(K) printf("%-18s %10.2f %10.2f %10.2f | %10.2f | %10.2f\n", names[K], run<K, 1>(dev, 1), run<K, 4>(dev, 1), run<K, 8>(dev, 1), run<K, 8>(dev, 4), run<K, 8>(dev, 8));
  ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11) ROW(12) ROW(13) ROW(14) ROW(15) ROW(16)
  return 0;
}
