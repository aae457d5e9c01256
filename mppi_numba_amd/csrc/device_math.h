// device_math.h -- arithmetic building blocks of the rollout kernels (gfx950).
//
// The reference's CPU path (numba CUDA simulator) evaluates every kernel body
// on numpy scalars: float32 (op) float32 stays float32, anything that meets a
// Python float / int literal or `**2` is float64, and every store into a
// float32 array rounds once.  MPPI_MATH_EXACT keeps exactly those rounding
// points (float64 where the reference has float64) so that trajectories and
// costs come out bit-identical; the float64 work sits off the critical
// dependency chain of the integrator (it overlaps the map gather), see
// DESIGN.md.  This file is compiled with -ffp-contract=off: an fma is only
// used where written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mppi {

// Developer instrumentation (tools/stamp_timeline.py): -DMPPI_STAMPS builds a library whose
// kernels record the shader clock (s_memtime) of lane 0 of chosen waves at chosen points into
// g_stamps; mppi_debug_read_stamps() copies them out.  Compiled out of the product build.
#ifdef MPPI_STAMPS
__device__ unsigned long long g_stamps[4096];
#define MPPI_STAMP(cond, slot)                                                                  \
  do {                                                                                          \
    if ((cond) && (threadIdx.x & 63) == 0) g_stamps[(slot)] = __builtin_readcyclecounter();     \
  } while (0)
#else
#define MPPI_STAMP(cond, slot) \
  do {                         \
  } while (0)
#endif

// Tile-major layout of per-(step, rollout) arrays (noise, control-cost products):
// the T x 64 block of every 64 consecutive rollouts is contiguous, so that the wave
// integrating those rollouts walks one ~T*512-byte run (one page, one DRAM row
// neighbourhood) instead of T rows 8*N bytes apart, while a fixed step over
// consecutive rollouts is still 64-element contiguous for the reductions.
__host__ __device__ __forceinline__ size_t tile_index(int t, int n, int n_steps) {
  return ((size_t)(n >> 6) * n_steps + t) * 64 + (n & 63);
}

// xi = int32((x - xlo) // res)  (mppi.py:971-972).  numpy's float32 floor
// division returns the floor of the EXACT quotient (it goes through fmod), so
// floorf(a / b) is not enough when a / b rounds up to an integer.  q0 is within
// one of the answer; the fused remainder has the sign of the exact remainder.
__device__ __forceinline__ int floordiv_to_int(float a, float b, float inv_b) {
  float q = floorf(a * inv_b);
  float r = fmaf(-q, b, a);
  q = (r < 0.0f) ? q - 1.0f : q;
  q = (r >= b) ? q + 1.0f : q;
  return (int)q;
}

// float64 sin/cos of an angle that is exactly a float32 value.
// Cody-Waite reduction by pi/2 (33-bit head so that k*head is exact for
// |k| < 2^20) followed by the classic degree-13 / degree-14 minimax kernels on
// [-pi/4, pi/4] (coefficients as published in fdlibm's k_sin.c / k_cos.c).
// Absolute error < 2e-16: after the float32 rounding of x + dt*v*tr*cos(th)
// this is indistinguishable from libm's cos() except with probability ~1e-9
// per step.  Large arguments take the library path.
// `BOUNDED` promises |x| <= 1e5 (the host proves it from |theta0| + T*dt*|w|max,
// see launch_rollout); then the hot loop carries no branch at all.
template <bool BOUNDED>
__device__ __forceinline__ void sincos_f64(double x, double& s_out, double& c_out) {
  if (!BOUNDED) {
    if (__builtin_expect(fabs(x) > 100000.0, 0)) {
      double s, c;
      sincos(x, &s, &c);
      s_out = s;
      c_out = c;
      return;
    }
  }
  const double two_over_pi = 6.36619772367581382433e-01;
  const double pio2_hi = 1.57079632673412561417e+00;
  const double pio2_lo = 6.07710050650619224932e-11;
  double fn = rint(x * two_over_pi);
  double y = fma(-fn, pio2_lo, fma(-fn, pio2_hi, x));
  int n = (int)fn;
  double z = y * y;
  // sin kernel
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  double sy = fma(y * z, ps, y);
  // cos kernel
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  double cy = fma(z * z, pc, fma(z, -0.5, 1.0));
  double s = (n & 1) ? cy : sy;
  double c = (n & 1) ? sy : cy;
  s = (n & 2) ? -s : s;
  c = ((n + 1) & 2) ? -c : c;
  s_out = s;
  c_out = c;
}

// Cell coordinate along one axis when the resolution is a power of two, relative to a window
// that starts `origin` cells into the map and clamped to [0, last]:
//   clamp(floor((pos - lo) / res) - origin, 0, last)
// in four instructions (sub, fma, med3, cvt) instead of seven.  Exactness: d = fl(pos - lo) is
// the reference's float32 difference; d * inv_res is exact (power of two); subtracting the
// integer `origin` from it is exact as well (a multiple of ulp(q) no larger than q), so the fma
// returns q - origin exactly and floor(q) - origin == floor(q - origin).  After the clamp the
// value is >= 0, where the truncating conversion is the floor.
__device__ __forceinline__ int cell_coord_pow2(float pos, float lo, float inv_res, float origin, float last) {
  float q = fmaf(pos - lo, inv_res, -origin);
  return (int)__builtin_amdgcn_fmed3f(q, 0.0f, last);
}

// (cos, sin) of theta+delta from (cos, sin) of theta, |delta| <= 0.36 rad (the host
// proves the bound from dt*max|w|*max traction).  Taylor kernels: the first
// omitted terms are delta^15/15! < 2e-19 and delta^14/14! < 8e-18; with the
// rounding of ~18 float64 operations the error per step is a few 1e-16 and grows
// linearly with the number of steps (no renormalisation): ~1e-13 after 1000
// steps, against the 3e-12 that would be needed to move a float32 rounding of
// x + dt*v*tr*cos(theta) with probability 1e-6.
// d = a*b + c as the three-operand v_fma_f64.  For a Horner step acc = z*acc + K the compiler
// prefers the two-operand v_fmac_f64 (accumulator tied to the addend) and then has to copy the
// constant K into the accumulator first: one v_mov_b64 per step, 11 per rotation on the
// critical wave.  Same IEEE operation, no copy.
__device__ __forceinline__ double fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// FMA3: use fma3 for the Horner steps.  Measured per kernel (profiles/r01_ablation.md): the CVaR
// kernel gains 10 %; the pipelined and the throughput kernel lose 2-8 % (the opaque asm costs
// the scheduler more than the copies cost those loops), so they keep the compiler's choice.
// the two halves of rotate_sincos_f64 below, for callers that evaluate the polynomials of several
// steps side by side (independent instruction streams) before walking the rotation chain
__device__ __forceinline__ void sincos_increment_f64(double delta, double& sd, double& cd) {
  const double z = delta * delta;
  double ps = fma(z, 1.6059043836821613e-10, -2.5052108385441720e-08);  // 1/13!, -1/11!
  ps = fma(z, ps, 2.7557319223985893e-06);                               // 1/9!
  ps = fma(z, ps, -1.9841269841269841e-04);                              // -1/7!
  ps = fma(z, ps, 8.3333333333333332e-03);                               // 1/5!
  ps = fma(z, ps, -1.6666666666666666e-01);                              // -1/3!
  sd = fma(delta * z, ps, delta);
  double pc = fma(z, 2.0876756987868100e-09, -2.7557319223985888e-07);   // 1/12!, -1/10!
  pc = fma(z, pc, 2.4801587301587302e-05);                               // 1/8!
  pc = fma(z, pc, -1.3888888888888889e-03);                              // -1/6!
  pc = fma(z, pc, 4.1666666666666664e-02);                               // 1/4!
  pc = fma(z, pc, -0.5);
  cd = fma(z, pc, 1.0);
}
__device__ __forceinline__ void apply_rotation_f64(double sd, double cd, double& s, double& c) {
  const double c2 = fma(c, cd, -(s * sd));
  const double s2 = fma(s, cd, c * sd);
  c = c2;
  s = s2;
}

template <bool FMA3 = false>
__device__ __forceinline__ void rotate_sincos_f64(double delta, double& s, double& c) {
  auto step = [](double a, double b, double k) { return FMA3 ? fma3(a, b, k) : fma(a, b, k); };
  double z = delta * delta;
  double ps = step(z, 1.6059043836821613e-10, -2.5052108385441720e-08);  // 1/13!, -1/11!
  ps = step(z, ps, 2.7557319223985893e-06);                               // 1/9!
  ps = step(z, ps, -1.9841269841269841e-04);                              // -1/7!
  ps = step(z, ps, 8.3333333333333332e-03);                               // 1/5!
  ps = step(z, ps, -1.6666666666666666e-01);                              // -1/3!
  double sd = fma(delta * z, ps, delta);
  double pc = step(z, 2.0876756987868100e-09, -2.7557319223985888e-07);   // 1/12!, -1/10!
  pc = step(z, pc, 2.4801587301587302e-05);                               // 1/8!
  pc = step(z, pc, -1.3888888888888889e-03);                              // -1/6!
  pc = step(z, pc, 4.1666666666666664e-02);                               // 1/4!
  pc = step(z, pc, -0.5);
  double cd = step(z, pc, 1.0);
  double c2 = fma(c, cd, -(s * sd));
  double s2 = fma(s, cd, c * sd);
  c = c2;
  s = s2;
}

// sqrt of a non-negative float64 to ~1e-14 relative: hardware float32 sqrt /
// reciprocal as the seed, one Newton step in float64 (the library's correctly
// rounded sqrt costs ~100 dependent cycles per call on gfx950 -- measured).
// Used where the result is added to a float32 cost: an error of 1e-14*sqrt(d2)
// against half an ulp of the cost changes a rounding with probability ~1e-9.
__device__ __forceinline__ double sqrt_newton_f64(double a) {
  // branch-free; a is a squared distance (0 <= a << 1e30).  a == 0 (or a float32
  // underflow) yields exactly 0.
  float yf = __builtin_amdgcn_sqrtf((float)a);
  double y0 = (double)yf;
  double r = fma(-y0, y0, a);
  double h = 0.5 * (double)__builtin_amdgcn_rcpf(yf);
  double y1 = fma(r, h, y0);
  return (yf > 0.0f) ? y1 : 0.0;
}

// The same square root for callers that add it, scaled, to a float64 of order >= 1e-3 before
// rounding to float32 (the stage cost): the seed is kept away from zero instead of selecting an
// exact 0 at the end -- for a < 1e-60 the result is ~1e-30 instead of 0, invisible in that sum --
// which saves the compare and the two selects of the float64 result.
__device__ __forceinline__ double sqrt_newton_nz_f64(double a) {
  const float yf = fmaxf(__builtin_amdgcn_sqrtf((float)a), 1.0e-30f);
  const double y0 = (double)yf;
  const double r = fma(-y0, y0, a);
  const double h = 0.5 * (double)__builtin_amdgcn_rcpf(yf);
  return fma(r, h, y0);
}

__device__ __forceinline__ float clip_f32(float v, float lo, float hi) {
  // max(lo, min(hi, v)) as the reference writes it
  return fmaxf(lo, fminf(hi, v));
}

// order-preserving float <-> uint mapping (for min reductions on integers)
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(uint32_t k) {
  uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(b);
}

// 64-lane reductions (wave64; all lanes end with the result) without LDS traffic: four DPP steps
// inside every row of 16 lanes (xor 1, xor 2, half-row mirror, row mirror), then the four row
// results through v_readlane.  (__shfl_xor is ds_bpermute: ~100 dependent cycles per step, six
// steps -- measured 4.2k cycles for the three float64 sums of the update kernel's epilogue.)
// Fixed order: deterministic, the same on every run and in every kernel.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;

__device__ __forceinline__ float wave_min_f32(float v) {
  v = fminf(v, dpp_f32<kDppXor1>(v));
  v = fminf(v, dpp_f32<kDppXor2>(v));
  v = fminf(v, dpp_f32<kDppHalfMirror>(v));
  v = fminf(v, dpp_f32<kDppMirror>(v));
  const int b = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
  return fminf(fminf(r0, r1), fminf(r2, r3));
}
// sum over the 64 lanes, result in lane 63 only (4 steps inside each row of 16 lanes, then row 0 -> 1,
// 2 -> 3 and rows 0..1 -> 2..3 through row_bcast): 6 DPP additions, fixed order
__device__ __forceinline__ float wave_sum_to_lane63_f32(float v) {
  v += dpp_f32<kDppXor1>(v);
  v += dpp_f32<kDppXor2>(v);
  v += dpp_f32<kDppHalfMirror>(v);
  v += dpp_f32<kDppMirror>(v);
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));  // row_bcast:15
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));  // row_bcast:31
  return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {  // ... handed to every lane
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63_f32(v)), 63));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<kDppXor1>(v);
  v += dpp_f64<kDppXor2>(v);
  v += dpp_f64<kDppHalfMirror>(v);
  v += dpp_f64<kDppMirror>(v);
  const long long b = __double_as_longlong(v);
  auto row = [&](int lane) {
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
  };
  return ((row(0) + row(16)) + row(32)) + row(48);
}

}  // namespace mppi
