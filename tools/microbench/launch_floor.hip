// What a kernel boundary costs between two back-to-back launches of the C2 shape (256 workgroups x 1024 threads x 140 KB of
// LDS: one workgroup per CU, gfx950) -- i.e. what "launch k+1 before launch k ends" or a persistent loop over the iterations
// could hide at most (profiles/r06_overlap_notes.md).
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
// Every wave stamps the device's constant 100 MHz clock when it enters and before it leaves; in between the workgroup
// keeps busy for `busy_us`.  Printed per configuration, medians over the launches of one stream:
//   wall       host time per launch of the back-to-back loop
//   first-in   last wave of launch k out  ->  first wave of launch k+1 in   (the boundary itself)
//   all-in     first wave in -> last wave in                                (the dispatcher fills 256 CUs)
//   all-out    first wave out -> last wave out
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_busy(unsigned long long* stamps, int waves_total, int busy_ticks, int touch) {
  extern __shared__ int lds[];
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const unsigned long long t0 = wall_clock64();
  if ((threadIdx.x & 63) == 0) stamps[wave] = t0;
  if (touch) lds[threadIdx.x] = (int)t0;
  while ((long long)(wall_clock64() - t0) < busy_ticks) __builtin_amdgcn_s_sleep(8);
  if (touch && lds[(threadIdx.x + 64) & 1023] == 12345) stamps[0] = 0;
  if ((threadIdx.x & 63) == 0) stamps[waves_total + wave] = wall_clock64();
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  const int launches = 200;
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  struct Cfg { int wgs, threads, lds; double busy_us; } cfgs[] = {
    {256, 1024, 140 * 1024, 12.0}, {256, 1024, 140 * 1024, 2.0}, {256, 1024, 0, 12.0}, {256, 256, 60 * 1024, 12.0}, {1024, 256, 0, 12.0},
  };
  for (const Cfg& c : cfgs) {
    const int waves = c.wgs * (c.threads / 64);
    unsigned long long* dev = nullptr;
    CHECK(hipMalloc(&dev, sizeof(unsigned long long) * 2 * waves * launches));
    CHECK(hipMemset(dev, 0, sizeof(unsigned long long) * 2 * waves * launches));
    if (c.lds > 64 * 1024) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_busy), hipFuncAttributeMaxDynamicSharedMemorySize, c.lds));
    const int ticks = (int)(c.busy_us * 100.0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_busy, dim3(c.wgs), dim3(c.threads), c.lds, s, dev, waves, ticks, 1);
    CHECK(hipStreamSynchronize(s));
    const auto h0 = std::chrono::steady_clock::now();
    for (int i = 0; i < launches; ++i)
      hipLaunchKernelGGL(k_busy, dim3(c.wgs), dim3(c.threads), c.lds, s, dev + (size_t)2 * waves * i, waves, ticks, 1);
    CHECK(hipStreamSynchronize(s));
    const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count() / launches;
    std::vector<unsigned long long> h((size_t)2 * waves * launches);
    CHECK(hipMemcpy(h.data(), dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> first_in, all_in, all_out, span;
    unsigned long long prev_last_out = 0;
    for (int i = 0; i < launches; ++i) {
      const unsigned long long* in = h.data() + (size_t)2 * waves * i;
      const unsigned long long* out = in + waves;
      const unsigned long long in_min = *std::min_element(in, in + waves), in_max = *std::max_element(in, in + waves);
      const unsigned long long out_min = *std::min_element(out, out + waves), out_max = *std::max_element(out, out + waves);
      if (i > 10) {
        first_in.push_back(((double)in_min - (double)prev_last_out) / 100.0);
        all_in.push_back((double)(in_max - in_min) / 100.0);
        all_out.push_back((double)(out_max - out_min) / 100.0);
        span.push_back((double)(out_max - in_min) / 100.0);
      }
      prev_last_out = out_max;
    }
    printf("wgs %4d x %4d threads, lds %6d B, busy %4.1f us: wall %6.2f us/launch | boundary (last out -> first in) %5.2f | all-in %5.2f | all-out %5.2f | first in -> last out %6.2f\n",
           c.wgs, c.threads, c.lds, c.busy_us, wall, median(first_in), median(all_in), median(all_out), median(span));
    CHECK(hipFree(dev));
  }
  return 0;
}
