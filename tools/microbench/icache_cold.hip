// Does a kernel find its instructions in the instruction cache when the same kernel ran just before it?  (gfx950)
//   hipcc --offload-arch=gfx950 -O3 -o icache_cold icache_cold.hip && ./icache_cold
// One wave per CU runs a straight-line body of BODY dependent v_fma_f32 instructions (4 bytes each... 8 with literals:
// here plain register operands, 8-byte VOP3) TWICE inside one launch and stamps s_memtime around each pass: pass 1 fetches
// the code from wherever it is after the previous launch, pass 2 from the instruction cache.  200 launches back to back.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BODY>
__device__ __forceinline__ float body(float v, float a, float b) {
#pragma unroll
  for (int i = 0; i < BODY; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
  return v;
}

template <int BODY>
__global__ __launch_bounds__(64) void k(unsigned long long* stamps, float* sink, float a, float b, int passes) {
  float v = threadIdx.x;
  unsigned long long t[5];
  t[0] = __builtin_readcyclecounter();
  for (int p = 0; p < passes; ++p) {  // (a loop: ONE copy of the body in the code)
    v = body<BODY>(v, a, b);
    t[p + 1] = __builtin_readcyclecounter();
  }
  if (threadIdx.x == 0) {
    for (int p = 0; p <= passes; ++p) stamps[blockIdx.x * 8 + p] = t[p];
  }
  if (v == 12345.0f) sink[0] = v;
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

template <int BODY>
int run(const char* what, bool other_kernel_between) {
  const int launches = 200, wgs = 256;
  unsigned long long* dev;
  float* sink;
  CHECK(hipMalloc(&dev, sizeof(unsigned long long) * 8 * wgs * launches));
  CHECK(hipMalloc(&sink, 4));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int i = 0; i < launches; ++i) {
    hipLaunchKernelGGL(k<BODY>, dim3(wgs), dim3(64), 0, s, dev + (size_t)8 * wgs * i, sink, 1.0001f, 0.5f, 3);
    if (other_kernel_between) hipLaunchKernelGGL(k<BODY + 1>, dim3(wgs), dim3(64), 0, s, dev + (size_t)8 * wgs * i + 4, sink, 1.0001f, 0.5f, 1);
  }
  CHECK(hipStreamSynchronize(s));
  std::vector<unsigned long long> h((size_t)8 * wgs * launches);
  CHECK(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> p1, p2, p3, first1;
  for (int i = 0; i < launches; ++i)
    for (int g = 0; g < wgs; ++g) {
      const unsigned long long* t = &h[((size_t)i * wgs + g) * 8];
      if (i == 0) first1.push_back((double)(t[1] - t[0]));
      if (i >= 10) { p1.push_back((double)(t[1] - t[0])); p2.push_back((double)(t[2] - t[1])); p3.push_back((double)(t[3] - t[2])); }
    }
  printf("%-46s body %5d instructions (%3d KB): launch 0 pass 1 %7.0f cycles | later launches: pass 1 %7.0f  pass 2 %7.0f  pass 3 %7.0f\n",
         what, BODY, BODY * 8 / 1024, median(first1), median(p1), median(p2), median(p3));
  CHECK(hipFree(dev));
  CHECK(hipFree(sink));
  return 0;
}

int main() {
  if (run<512>("same kernel back to back", false)) return 1;
  if (run<2048>("same kernel back to back", false)) return 1;
  if (run<6000>("same kernel back to back", false)) return 1;
  if (run<2048>("another kernel of the same size in between", true)) return 1;
  return 0;
}
