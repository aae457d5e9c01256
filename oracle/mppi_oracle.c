/*
 * mppi_oracle.c -- CPU restatement of the reference's MPPI hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product
 * (mppi_numba_amd/) never does and has no CPU fallback.
 *
 * What it restates: the reference has no native code -- its kernels are
 * numba @cuda.jit functions, and its own CPU path is the numba CUDA simulator
 * (NUMBA_ENABLE_CUDASIM=1), where every kernel body runs as plain Python on
 * numpy scalars.  This file follows THAT arithmetic: numpy-1.x scalar
 * promotion (float32 (op) float32 -> float32, anything touching a Python
 * float/int or `**2` -> float64) with a rounding to float32 at every store
 * into a float32 array.  Each function cites the reference lines it follows
 * (paths relative to /root/reference/mppi_numba unless stated).
 *
 * Pinning: tests/test_oracle_golden.py checks every function below against
 * tests/golden/ *.npz, which oracle/gen_golden.py produced by running the
 * unmodified reference under the simulator in this container.
 *
 * Third-party arithmetic restated here: numba.cuda.random (xoroshiro128+,
 * numba 0.54.1 as installed under /opt/conda; the reference leaves numba
 * unpinned, README.md:66).
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp -ffp-contract=off).
 * -ffp-contract=off and no -ffast-math are REQUIRED: the roundings are the
 * specification.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------
 * Parameters.  Mirrors move_mppi_task_vars_to_device (mppi.py:214-234): every
 * field that the reference casts to np.float32 is a float here; dist_weight is
 * passed to the kernels as a Python number (mppi.py:252,326,399) -> double.
 * lin_lo/ang_lo are bin_values_bounds_d[0]; lin_ratio/ang_ratio are
 * 0.01*(bounds[1]-bounds[0]) evaluated by the caller in the bounds array's own
 * dtype (float32 via set_TDM_from_PMF_grid terrain.py:401, the caller's dtype
 * via set_TDM_from_semantic_grid terrain.py:333), as mppi.py:674-675 does.
 * ---------------------------------------------------------------------- */
typedef struct {
  float res;            /* lin_tdm.res                                  */
  float xlo, ylo;       /* padded_xlimits[0], padded_ylimits[0]         */
  float vrange[2];
  float wrange[2];
  float xgoal[2];
  float x0[3];
  float u_std[2];
  float dt;
  float goal_tolerance;
  float v_post_rollout;
  float lambda_weight;
  float cvar_alpha;
  float obs_cost;
  float unknown_cost;
  float _pad;
  double dist_weight;
  double lin_lo, lin_ratio;
  double ang_lo, ang_ratio;
} oracle_params;

int oracle_params_size(void) { return (int)sizeof(oracle_params); }

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* Python's min(a, b) / max(a, b) on two numbers: the first argument wins ties. */
static inline float py_min(float a, float b) { return (b < a) ? b : a; }
static inline float py_max(float a, float b) { return (b > a) ? b : a; }

/* numpy float32 floor division (np.float32 // np.float32): exact remainder via
 * fmod, quotient reconstructed from it, then snapped to the integer below.
 * Used for the cell index xi = int32((x - xlo) // res), mppi.py:971-972. */
static float floordiv_f32(float a, float b) {
  float mod = fmodf(a, b);
  float div = (a - mod) / b;
  if (mod != 0.0f) {
    if ((b < 0.0f) != (mod < 0.0f)) div -= 1.0f;
  }
  if (div != 0.0f) {
    float fl = floorf(div);
    if (div - fl > 0.5f) fl += 1.0f;
    return fl;
  }
  return copysignf(0.0f, a / b);
}

/* Python negative indexing, as numpy arrays do under the simulator.  The
 * reference relies on the zero-traction padding to keep indices in range. */
static inline int wrap_index(int i, int n) {
  if (i < 0) i += n;
  if (i < 0) i = 0;
  if (i >= n) i = n - 1;
  return i;
}

/* one Euler step of the unicycle with traction, shared by every rollout
 * flavour: mppi.py:679-687 == 977-990 == 1074-1090.  Products are float64
 * (vtraction is float64 because of the 0.01 literal), stores are float32. */
static inline void unicycle_step(float x[3], float dt, double vtr, double wtr, float v, float w) {
  double th = (double)x[2];
  float nx = (float)((double)x[0] + (double)dt * vtr * (double)v * cos(th));
  float ny = (float)((double)x[1] + (double)dt * vtr * (double)v * sin(th));
  float nt = (float)((double)x[2] + (double)dt * wtr * (double)w);
  x[0] = nx;
  x[1] = ny;
  x[2] = nt;
}

static inline double dist2_to_goal(const oracle_params* p, const float x[3]) {
  /* (xgoal[0]-x[0])**2 + (xgoal[1]-x[1])**2 : float32 differences, then
   * `**2` promotes to float64 (np.float32 ** int -> float64). */
  double dx = (double)(float)(p->xgoal[0] - x[0]);
  double dy = (double)(float)(p->xgoal[1] - x[1]);
  return dx * dx + dy * dy;
}

static inline double control_cost_term(const oracle_params* p, const float* u_t, const float* eps_t) {
  /* lambda*((u[t,0]/std0**2)*eps0 + (u[t,1]/std1**2)*eps1), mppi.py:708-710 */
  double s0 = (double)p->u_std[0] * (double)p->u_std[0];
  double s1 = (double)p->u_std[1] * (double)p->u_std[1];
  double a = ((double)u_t[0] / s0) * (double)eps_t[0];
  double b = ((double)u_t[1] / s1) * (double)eps_t[1];
  return (double)p->lambda_weight * (a + b);
}

static inline double terminal_cost(const oracle_params* p, double d2, int reached) {
  /* term_cost, mppi.py:26-28 */
  return (1.0 - (double)(float)reached) * sqrt(d2) / ((double)p->v_post_rollout + 1e-6);
}

/* ------------------------------------------------------------------------
 * rollout_det_dyn_numba (mppi.py:916-1009) and, when risk != NULL,
 * rollout_det_dyn_w_speed_map_numba (mppi.py:1013-1111).
 *   lin, ang : sample grid 0, int8, row stride `grid_stride` (= max_map_dim[1])
 *   obs, unk : padded masks (Rp, Cp) int8, contiguous
 *   risk     : padded risk traction map (Rp, Cp) int8 or NULL
 *   noise    : (N, T, 2) float32, u : (T, 2) float32
 * Order of accumulation: stage, obstacle, unknown per step; then terminal;
 * then the control cost of ALL T steps (also after an early goal break).
 * ---------------------------------------------------------------------- */
void oracle_rollout_det(const oracle_params* p, const int8_t* lin, const int8_t* ang,
                        int grid_rows, int grid_stride, const int8_t* obs, const int8_t* unk,
                        const int8_t* risk, int rp, int cp, const float* noise, const float* u,
                        int n_rollouts, int n_steps, float* costs) {
  const float gt2 = p->goal_tolerance * p->goal_tolerance;
#pragma omp parallel for schedule(static)
  for (int n = 0; n < n_rollouts; ++n) {
    const float* eps = noise + (size_t)n * n_steps * 2;
    float cost = 0.0f;
    float x[3] = {p->x0[0], p->x0[1], p->x0[2]};
    double d2 = 1e9;
    int reached = 0;
    for (int t = 0; t < n_steps; ++t) {
      int xi = (int)floordiv_f32(x[0] - p->xlo, p->res);
      int yi = (int)floordiv_f32(x[1] - p->ylo, p->res);
      int gx = wrap_index(xi, grid_stride), gy = wrap_index(yi, grid_rows);
      int mx = wrap_index(xi, cp), my = wrap_index(yi, rp);
      double vtr = p->lin_lo + p->lin_ratio * (double)lin[(size_t)gy * grid_stride + gx];
      double wtr = p->ang_lo + p->ang_ratio * (double)ang[(size_t)gy * grid_stride + gx];
      float v = py_max(p->vrange[0], py_min(p->vrange[1], u[2 * t] + eps[2 * t]));
      float w = py_max(p->wrange[0], py_min(p->wrange[1], u[2 * t + 1] + eps[2 * t + 1]));
      unicycle_step(x, p->dt, vtr, wtr, v, w);
      d2 = dist2_to_goal(p, x);
      double step_time = (double)p->dt;
      if (risk) {
        double eff = p->lin_lo + p->lin_ratio * (double)risk[(size_t)my * cp + mx];
        step_time = (double)p->dt / (eff + 1e-6);
      }
      /* stage_cost, mppi.py:20-22 */
      cost = (float)((double)cost + (step_time + p->dist_weight * sqrt(d2)));
      cost = cost + (float)obs[(size_t)my * cp + mx] * p->obs_cost;
      cost = cost + (float)unk[(size_t)my * cp + mx] * p->unknown_cost;
      if (d2 <= (double)gt2) {
        reached = 1;
        break;
      }
    }
    cost = (float)((double)cost + terminal_cost(p, d2, reached));
    for (int t = 0; t < n_steps; ++t)
      cost = (float)((double)cost + control_cost_term(p, u + 2 * t, eps + 2 * t));
    costs[n] = cost;
  }
}

static int cmp_desc(const void* a, const void* b) {
  float fa = *(const float*)a, fb = *(const float*)b;
  return (fa < fb) - (fa > fb);
}

/* ------------------------------------------------------------------------
 * rollout_numba (mppi.py:613-755): control sample n over traction sample m,
 * then per n the mean of the ceil(M*alpha) largest costs (CVaR) or of all.
 *   lin, ang : (M, grid_rows, grid_stride) int8
 *   per_sample (optional, may be NULL): (N, M) costs before the reduction
 * Order: stage/obs/unk per step; control cost of all T steps; terminal.
 * The odd-even transposition sort of mppi.py:719-740 runs ceil(M/2) double
 * rounds, i.e. >= M phases: the array ends fully sorted (descending), so any
 * correct sort restates it.  The strided tree sum of mppi.py:744-751 is kept
 * as written (float32 additions in that exact association).
 * rollout_oversized_numba (mppi.py:760-913) computes the same thing for
 * cvar_alpha == 1; for alpha < 1 its 'sort' swaps without comparing
 * (mppi.py:881-895) and is not restated.
 * ---------------------------------------------------------------------- */
void oracle_rollout_tdm(const oracle_params* p, const int8_t* lin, const int8_t* ang,
                        int n_grids, int grid_rows, int grid_stride, const int8_t* obs,
                        const int8_t* unk, int rp, int cp, const float* noise, const float* u,
                        int n_rollouts, int n_steps, float* costs, float* per_sample) {
  const float gt2 = p->goal_tolerance * p->goal_tolerance;
  const size_t plane = (size_t)grid_rows * grid_stride;
  /* numel = math.ceil(block_width*cvar_alpha_d): int * np.float32 -> float64 */
  int numel = (int)ceil((double)n_grids * (double)p->cvar_alpha);
#pragma omp parallel
  {
    float* c = (float*)malloc(sizeof(float) * (size_t)n_grids);
#pragma omp for schedule(static)
    for (int n = 0; n < n_rollouts; ++n) {
      const float* eps = noise + (size_t)n * n_steps * 2;
      for (int m = 0; m < n_grids; ++m) {
        const int8_t* lg = lin + plane * m;
        const int8_t* ag = ang + plane * m;
        float cost = 0.0f;
        float x[3] = {p->x0[0], p->x0[1], p->x0[2]};
        double d2 = 1e9;
        int reached = 0;
        for (int t = 0; t < n_steps; ++t) {
          int xi = (int)floordiv_f32(x[0] - p->xlo, p->res);
          int yi = (int)floordiv_f32(x[1] - p->ylo, p->res);
          int gx = wrap_index(xi, grid_stride), gy = wrap_index(yi, grid_rows);
          int mx = wrap_index(xi, cp), my = wrap_index(yi, rp);
          double vtr = p->lin_lo + p->lin_ratio * (double)lg[(size_t)gy * grid_stride + gx];
          double wtr = p->ang_lo + p->ang_ratio * (double)ag[(size_t)gy * grid_stride + gx];
          float v = py_max(p->vrange[0], py_min(p->vrange[1], u[2 * t] + eps[2 * t]));
          float w = py_max(p->wrange[0], py_min(p->wrange[1], u[2 * t + 1] + eps[2 * t + 1]));
          unicycle_step(x, p->dt, vtr, wtr, v, w);
          d2 = dist2_to_goal(p, x);
          cost = (float)((double)cost + ((double)p->dt + p->dist_weight * sqrt(d2)));
          cost = cost + (float)obs[(size_t)my * cp + mx] * p->obs_cost;
          cost = cost + (float)unk[(size_t)my * cp + mx] * p->unknown_cost;
          if (d2 <= (double)gt2) {
            reached = 1;
            break;
          }
        }
        for (int t = 0; t < n_steps; ++t)
          cost = (float)((double)cost + control_cost_term(p, u + 2 * t, eps + 2 * t));
        cost = (float)((double)cost + terminal_cost(p, d2, reached));
        c[m] = cost;
      }
      if (per_sample) memcpy(per_sample + (size_t)n * n_grids, c, sizeof(float) * (size_t)n_grids);
      if (p->cvar_alpha < 1.0f) qsort(c, (size_t)n_grids, sizeof(float), cmp_desc);
      for (int s = 1; s < numel; s *= 2)
        for (int tid = 0; tid < n_grids; tid += 2 * s)
          if (tid + s < numel) c[tid] = c[tid] + c[tid + s];
      /* costs_d[bid] = shared[0]/numel : np.float32 / int -> float64 -> float32 */
      costs[n] = (float)((double)c[0] / (double)numel);
    }
    free(c);
  }
}

/* ------------------------------------------------------------------------
 * barebone rollout_numba (barebone_mppi_numba.ipynb cell 3, raw line 336):
 * nominal unicycle, stage = dist_weight*d2, circular obstacles tested at the
 * POST-step position, terminal = (1-reached)*d2.
 * ---------------------------------------------------------------------- */
void oracle_rollout_barebone(const oracle_params* p, const float* obs_pos, const float* obs_r,
                             int n_obs, const float* noise, const float* u, int n_rollouts,
                             int n_steps, float* costs) {
  const float gt2 = p->goal_tolerance * p->goal_tolerance;
#pragma omp parallel for schedule(static)
  for (int n = 0; n < n_rollouts; ++n) {
    const float* eps = noise + (size_t)n * n_steps * 2;
    float cost = 0.0f;
    float x[3] = {p->x0[0], p->x0[1], p->x0[2]};
    double d2 = 1e9;
    int reached = 0;
    for (int t = 0; t < n_steps; ++t) {
      float v = py_max(p->vrange[0], py_min(p->vrange[1], u[2 * t] + eps[2 * t]));
      float w = py_max(p->wrange[0], py_min(p->wrange[1], u[2 * t + 1] + eps[2 * t + 1]));
      /* dt_d*v_noisy*math.cos(theta): float32*float32 first, then * Python float */
      double th = (double)x[2];
      float dtv = p->dt * v;
      float nx = (float)((double)x[0] + (double)dtv * cos(th));
      float ny = (float)((double)x[1] + (double)dtv * sin(th));
      float nt = x[2] + p->dt * w;
      x[0] = nx; x[1] = ny; x[2] = nt;
      d2 = dist2_to_goal(p, x);
      cost = (float)((double)cost + p->dist_weight * d2);
      for (int k = 0; k < n_obs; ++k) {
        double ex = (double)(float)(x[0] - obs_pos[2 * k]);
        double ey = (double)(float)(x[1] - obs_pos[2 * k + 1]);
        double diff = ex * ex + ey * ey - (double)obs_r[k] * (double)obs_r[k];
        /* (1-numba.float32(dist_diff>0))*obs_cost_d : float64 * float32 */
        double hit = 1.0 - (double)(diff > 0.0);
        cost = (float)((double)cost + hit * (double)p->obs_cost);
      }
      if (d2 <= (double)gt2) {
        reached = 1;
        break;
      }
    }
    cost = (float)((double)cost + (1.0 - (double)(float)reached) * d2);
    for (int t = 0; t < n_steps; ++t)
      cost = (float)((double)cost + control_cost_term(p, u + 2 * t, eps + 2 * t));
    costs[n] = cost;
  }
}

/* ------------------------------------------------------------------------
 * update_useq_numba[1, P] (mppi.py:1113-1191; P = 32 at every call site).
 * costs is clobbered exactly as the reference does (reused for the weight
 * sum).  The float32 atomic adds of mppi.py:1179-1182 have no defined order
 * across threads; this restatement runs thread 0, then 1, ... (each thread:
 * t outer, i inner), which is one of the orders the reference can produce.
 * ---------------------------------------------------------------------- */
void oracle_update_useq(float lambda_weight, float* costs, const float* noise, float* weights,
                        const float* vrange, const float* wrange, float* u, int numel, int n_steps,
                        int num_threads) {
  const int P = num_threads;
  const int gap = (int)ceil((double)numel / (double)P);
  int* s0 = (int*)malloc(sizeof(int) * (size_t)P);
  int* e0 = (int*)malloc(sizeof(int) * (size_t)P);
  for (int p = 0; p < P; ++p) {
    long st = (long)p * gap;
    s0[p] = (int)(st < numel ? st : numel);
    e0[p] = (s0[p] + gap < numel) ? s0[p] + gap : numel;
  }
  for (int p = 0; p < P; ++p) {
    if (s0[p] < numel) weights[s0[p]] = costs[s0[p]];
    for (int i = s0[p]; i < e0[p]; ++i) weights[s0[p]] = py_min(weights[s0[p]], costs[i]);
  }
  for (long s = gap; s < numel; s *= 2)
    for (int p = 0; p < P; ++p)
      if ((s0[p] % (2 * s) == 0) && (s0[p] + s < numel))
        weights[s0[p]] = py_min(weights[s0[p]], weights[s0[p] + s]);
  const float beta = weights[0];
  const double neg_inv_lambda = -1.0 / (double)lambda_weight;
  for (int i = 0; i < numel; ++i)
    weights[i] = (float)exp(neg_inv_lambda * (double)(float)(costs[i] - beta));
  for (int i = 0; i < numel; ++i) costs[i] = weights[i];
  for (int p = 0; p < P; ++p)
    for (int i = s0[p] + 1; i < e0[p]; ++i) costs[s0[p]] = costs[s0[p]] + costs[i];
  for (long s = gap; s < numel; s *= 2)
    for (int p = 0; p < P; ++p)
      if ((s0[p] % (2 * s) == 0) && (s0[p] + s < numel))
        costs[s0[p]] = costs[s0[p]] + costs[s0[p] + s];
  const float total = costs[0];
  for (int i = 0; i < numel; ++i) weights[i] = weights[i] / total;
  for (int p = 0; p < P; ++p)
    for (int t = 0; t < n_steps; ++t)
      for (int i = s0[p]; i < e0[p]; ++i) {
        const float* e = noise + ((size_t)i * n_steps + t) * 2;
        u[2 * t] = u[2 * t] + weights[i] * e[0];
        u[2 * t + 1] = u[2 * t + 1] + weights[i] * e[1];
      }
  for (int t = 0; t < n_steps; ++t) {
    u[2 * t] = py_max(vrange[0], py_min(vrange[1], u[2 * t]));
    u[2 * t + 1] = py_max(wrange[0], py_min(wrange[1], u[2 * t + 1]));
  }
  free(s0);
  free(e0);
}

/* ------------------------------------------------------------------------
 * numba.cuda.random, xoroshiro128+ (numba/cuda/random.py:45-98 next/rotl/
 * SplitMix64 seeding, 102-125 jump, 129-139 uint64->unit float, 175-196
 * Box-Muller normal, 225-240 state array initialisation).
 * ---------------------------------------------------------------------- */
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static inline uint64_t xoro_next(uint64_t* s) {
  uint64_t s0 = s[0], s1 = s[1];
  uint64_t result = s0 + s1;
  s1 ^= s0;
  s[0] = rotl64(s0, 55) ^ s1 ^ (s1 << 14);
  s[1] = rotl64(s1, 36);
  return result;
}

static void xoro_jump(uint64_t* s) {
  static const uint64_t JUMP[2] = {0xbeac0467eba5facbULL, 0xd86b048b86aa9922ULL};
  uint64_t a = 0, b = 0;
  for (int i = 0; i < 2; ++i)
    for (int bit = 0; bit < 64; ++bit) {
      if (JUMP[i] & (1ULL << bit)) {
        a ^= s[0];
        b ^= s[1];
      }
      xoro_next(s);
    }
  s[0] = a;
  s[1] = b;
}

/* states: n x {s0, s1}; stream k is stream k-1 jumped by 2**64 */
void oracle_xoroshiro_init(uint64_t* states, long n, uint64_t seed) {
  if (n < 1) return;
  uint64_t z = seed + 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  states[0] = z;
  states[1] = z;
  for (long i = 1; i < n; ++i) {
    states[2 * i] = states[2 * (i - 1)];
    states[2 * i + 1] = states[2 * (i - 1) + 1];
    xoro_jump(states + 2 * i);
  }
}

static inline float xoro_uniform_f32(uint64_t* s) {
  return (float)((double)(xoro_next(s) >> 11) * (1.0 / 9007199254740992.0));
}

/* xoroshiro128p_normal_float32 as the simulator evaluates it: math.log /
 * math.cos return Python floats, so everything after the two float32 uniforms
 * is float64; the result is NOT rounded to float32 before the caller uses it. */
static inline double xoro_normal_sim(uint64_t* s) {
  const float two_pi_f32 = (float)(2.0 * M_PI);
  float u1 = xoro_uniform_f32(s);
  float u2 = xoro_uniform_f32(s);
  return sqrt(-2.0 * log((double)u1)) * cos((double)(float)(two_pi_f32 * u2));
}

double oracle_xoroshiro_normal(uint64_t* states, long index) { return xoro_normal_sim(states + 2 * index); }
float oracle_xoroshiro_uniform(uint64_t* states, long index) { return xoro_uniform_f32(states + 2 * index); }

/* sample_noise_numba[N, T] (mppi.py:1354-1370): stream n*T+t gives both
 * channels of noise[n, t, :]. */
void oracle_sample_noise(uint64_t* states, const float* u_std, int n_rollouts, int n_steps,
                         float* noise) {
  const long total = (long)n_rollouts * n_steps;
#pragma omp parallel for schedule(static)
  for (long k = 0; k < total; ++k) {
    double z0 = xoro_normal_sim(states + 2 * k);
    double z1 = xoro_normal_sim(states + 2 * k);
    noise[2 * k] = (float)((double)u_std[0] * z0);
    noise[2 * k + 1] = (float)((double)u_std[1] * z1);
  }
}

/* ------------------------------------------------------------------------
 * sample_grids_numba (terrain.py:633-695), launched [(1,G),(tx,ty)].
 * Thread (i,j) of block g uses stream i*ty*G + g*ty + j (terrain.py:657-658)
 * and walks its ceil(Rp/tx) x ceil(Cp/ty) tile row-major, one uniform per
 * cell.  bin_to_int8[b] = np.int8(100.*(bin_values[b]-lo)/(hi-lo)) is
 * evaluated by the caller in the dtypes the reference holds on the device
 * (terrain.py:689).  Only out[:, :Rp, :Cp] is written; out has row stride
 * out_stride and plane out_rows*out_stride.
 * ---------------------------------------------------------------------- */
void oracle_sample_grids(const int8_t* pmf, int n_bins, int rp, int cp, uint64_t* states,
                         int n_grids, int tx, int ty, const int8_t* bin_to_int8, double alpha_dyn,
                         int8_t* out, int out_rows, int out_stride) {
  const int nr = (int)ceil((double)rp / (double)tx);
  const int nc = (int)ceil((double)cp / (double)ty);
  const size_t cells = (size_t)rp * cp;
  for (int g = 0; g < n_grids; ++g)
    for (int i = 0; i < tx; ++i)
      for (int j = 0; j < ty; ++j) {
        uint64_t* st = states + 2 * ((long)i * ty * n_grids + (long)g * ty + j);
        int r0 = i * nr < rp ? i * nr : rp, r1 = r0 + nr < rp ? r0 + nr : rp;
        int c0 = j * nc < cp ? j * nc : cp, c1 = c0 + nc < cp ? c0 + nc : cp;
        for (int r = r0; r < r1; ++r)
          for (int c = c0; c < c1; ++c) {
            float rnd = xoro_uniform_f32(st);
            /* np.int8(math.ceil(rand*100.0*alpha_dyn)) */
            int8_t target = (int8_t)(long)ceil((double)rnd * 100.0 * alpha_dyn);
            int8_t cum = 0;
            for (int b = 0; b < n_bins; ++b) {
              cum = (int8_t)(cum + pmf[(size_t)b * cells + (size_t)r * cp + c]);
              if (target <= cum) {
                out[((size_t)g * out_rows + r) * out_stride + c] = bin_to_int8[b];
                break;
              }
            }
          }
      }
}

/* ------------------------------------------------------------------------
 * Visualisation rollouts.
 * get_state_rollout_across_control_noise[V,1] (mppi.py:1194-1295): row 0 is
 * u_cur without noise and WITHOUT clipping; row b>0 is clip(u_prev+noise[b]).
 * All rows use traction sample 0.  out: (V, T+1, 3).
 * ---------------------------------------------------------------------- */
void oracle_state_rollout_noise(const oracle_params* p, const int8_t* lin, const int8_t* ang,
                                int grid_rows, int grid_stride, const float* noise,
                                const float* u_prev, const float* u_cur, int n_vis, int n_steps,
                                float* out) {
  for (int b = 0; b < n_vis; ++b) {
    float* o = out + (size_t)b * (n_steps + 1) * 3;
    float x[3] = {p->x0[0], p->x0[1], p->x0[2]};
    memcpy(o, x, sizeof(x));
    const float* eps = noise + (size_t)b * n_steps * 2;
    for (int t = 0; t < n_steps; ++t) {
      int xi = (int)floordiv_f32(x[0] - p->xlo, p->res);
      int yi = (int)floordiv_f32(x[1] - p->ylo, p->res);
      int gx = wrap_index(xi, grid_stride), gy = wrap_index(yi, grid_rows);
      double vtr = p->lin_lo + p->lin_ratio * (double)lin[(size_t)gy * grid_stride + gx];
      double wtr = p->ang_lo + p->ang_ratio * (double)ang[(size_t)gy * grid_stride + gx];
      float v, w;
      if (b == 0) {
        v = u_cur[2 * t];
        w = u_cur[2 * t + 1];
      } else {
        v = py_max(p->vrange[0], py_min(p->vrange[1], u_prev[2 * t] + eps[2 * t]));
        w = py_max(p->wrange[0], py_min(p->wrange[1], u_prev[2 * t + 1] + eps[2 * t + 1]));
      }
      unicycle_step(x, p->dt, vtr, wtr, v, w);
      memcpy(o + 3 * (t + 1), x, sizeof(x));
    }
  }
}

/* get_state_rollout_across_envs_numba[1,V] (mppi.py:1298-1351): u_cur rolled
 * out over traction samples 0..V-1, no clipping. */
void oracle_state_rollout_envs(const oracle_params* p, const int8_t* lin, const int8_t* ang,
                               int grid_rows, int grid_stride, const float* u_cur, int n_vis,
                               int n_steps, float* out) {
  const size_t plane = (size_t)grid_rows * grid_stride;
  for (int m = 0; m < n_vis; ++m) {
    float* o = out + (size_t)m * (n_steps + 1) * 3;
    float x[3] = {p->x0[0], p->x0[1], p->x0[2]};
    memcpy(o, x, sizeof(x));
    for (int t = 0; t < n_steps; ++t) {
      int xi = (int)floordiv_f32(x[0] - p->xlo, p->res);
      int yi = (int)floordiv_f32(x[1] - p->ylo, p->res);
      int gx = wrap_index(xi, grid_stride), gy = wrap_index(yi, grid_rows);
      double vtr = p->lin_lo + p->lin_ratio * (double)lin[plane * m + (size_t)gy * grid_stride + gx];
      double wtr = p->ang_lo + p->ang_ratio * (double)ang[plane * m + (size_t)gy * grid_stride + gx];
      unicycle_step(x, p->dt, vtr, wtr, u_cur[2 * t], u_cur[2 * t + 1]);
      memcpy(o + 3 * (t + 1), x, sizeof(x));
    }
  }
}

/* barebone get_state_rollout_across_control_noise (notebook cell 3, raw line
 * 412): same without maps. */
void oracle_state_rollout_barebone(const oracle_params* p, const float* noise, const float* u_prev,
                                   const float* u_cur, int n_vis, int n_steps, float* out) {
  for (int b = 0; b < n_vis; ++b) {
    float* o = out + (size_t)b * (n_steps + 1) * 3;
    float x[3] = {p->x0[0], p->x0[1], p->x0[2]};
    memcpy(o, x, sizeof(x));
    const float* eps = noise + (size_t)b * n_steps * 2;
    for (int t = 0; t < n_steps; ++t) {
      float v, w;
      if (b == 0) {
        v = u_cur[2 * t];
        w = u_cur[2 * t + 1];
      } else {
        v = py_max(p->vrange[0], py_min(p->vrange[1], u_prev[2 * t] + eps[2 * t]));
        w = py_max(p->wrange[0], py_min(p->wrange[1], u_prev[2 * t + 1] + eps[2 * t + 1]));
      }
      double th = (double)x[2];
      float dtv = p->dt * v;
      float nx = (float)((double)x[0] + (double)dtv * cos(th));
      float ny = (float)((double)x[1] + (double)dtv * sin(th));
      float nt = x[2] + p->dt * w;
      x[0] = nx; x[1] = ny; x[2] = nt;
      memcpy(o + 3 * (t + 1), x, sizeof(x));
    }
  }
}
