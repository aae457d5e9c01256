// world_kernels.h -- the simulated world of the closed-loop demo, on the device (gfx950).
//
// SURVEY.md section 8f rank 4.  The reference keeps the "true" world on the host: one traction
// realisation per cell drawn from the terrains' densities (TDM_Numba.sample_grids_true_dist,
// terrain.py:586-608), held in a TractionGrid (terrain.py:750-785) whose get(x, y) is called once
// per control step by the notebooks' loop (test.ipynb cell 4: solve -> TractionGrid.get -> Euler
// step in float64 -> shift_and_update -> goal check).  Every control step therefore crosses the
// PCIe bus twice (8*T bytes of controls out, the new start state in).  Here the world is a device
// object: k_world_sample draws it, k_world_get answers batches of queries, and k_world_step does
// what the notebook's loop body does between two solve() calls -- for every problem of a batched
// handle -- without leaving the GPU: the new start state goes straight into the planner's
// per-problem record (BatchInst, with the LDS window origin the host would have planned), the
// control sequence is shifted in place, the trajectory is logged in device memory.
#pragma once
#include "rng_kernels.h"
#include "rollout_kernels.h"

namespace mppi {

struct WorldGrid {
  const double* lin;  // [rows][cols] float64, as TractionGrid.lin_traction (terrain.py:757-762)
  const double* ang;
  int rows, cols;
  double res, xlo, ylo;
};

// Python's / numpy's float floor division a // b (floatobject.c float_divmod, npy_divmod):
// fmod-based, so that the quotient is exact where floor(a / b) would round across an integer.
__device__ __forceinline__ double py_floordiv(double a, double b) {
  double mod = fmod(a, b);
  double div = (a - mod) / b;
  if (mod != 0.0 && ((b < 0.0) != (mod < 0.0))) div -= 1.0;
  if (div == 0.0) return copysign(0.0, a / b);
  double fl = floor(div);
  if (div - fl > 0.5) fl += 1.0;
  return fl;
}

// TractionGrid.get (terrain.py:776-782): the cell of (x, y); (0, 0) outside the grid
__device__ __forceinline__ void world_lookup(const WorldGrid& G, double x, double y, double& lin, double& ang) {
  const double fx = py_floordiv(x - G.xlo, G.res), fy = py_floordiv(y - G.ylo, G.res);
  lin = 0.0;
  ang = 0.0;
  if (!(fx >= 0.0) || !(fx < (double)G.cols) || !(fy >= 0.0) || !(fy < (double)G.rows)) return;
  const size_t at = (size_t)(int)fy * G.cols + (size_t)(int)fx;
  lin = G.lin[at];
  ang = G.ang[at];
}

__global__ void k_world_get(WorldGrid G, const double* __restrict__ xy, int count, double* __restrict__ lin_out,
                            double* __restrict__ ang_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  double l, a;
  world_lookup(G, xy[2 * i], xy[2 * i + 1], l, a);
  lin_out[i] = l;
  ang_out[i] = a;
}

// sample_grids_true_dist (terrain.py:586-608): every cell gets its own draw from the densities of
// its terrain type, linear and angular traction independently.  The densities are host objects
// (scipy / numpy samplers); what travels to the device is the pool of samples every Terrain
// already keeps (lin_saved_samples / ang_saved_samples, terrain.py:44-46): a cell draws one pool
// entry uniformly -- Philox4x32-10, counter = cell index, key = seed -- i.e. from the empirical
// distribution of the pool.
__global__ void k_world_sample(const int32_t* __restrict__ terrain_of_cell, int n_cells, int n_terrains,
                               const double* __restrict__ lin_pool, const double* __restrict__ ang_pool, int pool_len,
                               uint64_t seed, uint64_t epoch, double* __restrict__ lin, double* __restrict__ ang) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cells) return;
  const uint4 r = philox4x32_10(make_uint4((unsigned)i, 0u, (unsigned)epoch, (unsigned)(epoch >> 32)),
                                make_uint2((unsigned)seed, (unsigned)(seed >> 32)));
  int tt = terrain_of_cell[i];
  tt = tt < 0 ? 0 : (tt >= n_terrains ? n_terrains - 1 : tt);
  const size_t base = (size_t)tt * pool_len;
  lin[i] = lin_pool[base + __umulhi(r.x, (unsigned)pool_len)];
  ang[i] = ang_pool[base + __umulhi(r.y, (unsigned)pool_len)];
}

// What one control step of the closed loop needs besides the planner's own buffers.
struct WorldLoop {
  double* state;      // [B][3] float64 (x, y, theta): the notebook's xhist row
  double* xhist;      // [B][max_steps + 1][3]
  float2* uhist;      // [B][max_steps]
  int* done;          // [B] 0 = running, else the number of steps taken when the goal was reached
  int* done_count;    // host-mapped: problems that have reached their goal
  float2* u_final;    // [B][T] the controls a problem had when it reached its goal (shifted once)
  int max_steps;
  double dt, goal_tolerance;
  // the LDS window plan of the planner (plan_lds_window): origin per problem around its start cell
  int win_active, reach, win_rows, win_cols, map_rows, map_pitch;
  double xlo, ylo, res;
};

// One block per problem.  The body of the notebook's loop after solve() (test.ipynb cell 4):
//   u_curr = useq[0]; (lt, at) = traction_grid.get(x, y)
//   x += dt*lt*cos(th)*u_curr[0]; y += dt*lt*sin(th)*u_curr[0]; th += dt*at*u_curr[1]   (float64)
//   shift_and_update(x_new, useq, 1)   (mppi.py:534-542: x0 <- x_new, u[:-1] = u[1:])
//   goal check: ||x_new[:2] - goal|| <= goal_tolerance
// A problem that has reached its goal is left alone (state, controls, log).
__global__ void k_world_step(WorldGrid G, WorldLoop L, BatchInst* __restrict__ inst, float2* __restrict__ u,
                             int n_steps, int step) {
  extern __shared__ float2 shifted[];
  __shared__ int reached_now;
  const int b = blockIdx.x;
  if (L.done[b]) return;  // (uniform over the block)
  if (threadIdx.x == 0) reached_now = 0;
  float2* ub = u + (size_t)b * n_steps;
  for (int t = threadIdx.x; t < n_steps; t += blockDim.x) shifted[t] = ub[t];
  __syncthreads();
  if (threadIdx.x == 0) {
    const double x = L.state[3 * b], y = L.state[3 * b + 1], th = L.state[3 * b + 2];
    const float2 u0 = shifted[0];
    L.uhist[(size_t)b * L.max_steps + step] = u0;
    double lt, at;
    world_lookup(G, x, y, lt, at);
    const double x1 = x + L.dt * lt * cos(th) * (double)u0.x;
    const double y1 = y + L.dt * lt * sin(th) * (double)u0.x;
    const double th1 = th + L.dt * at * (double)u0.y;
    L.state[3 * b] = x1;
    L.state[3 * b + 1] = y1;
    L.state[3 * b + 2] = th1;
    double* row = L.xhist + ((size_t)b * (L.max_steps + 1) + step + 1) * 3;
    row[0] = x1; row[1] = y1; row[2] = th1;
    BatchInst I = inst[b];
    I.x0 = (float)x1; I.y0 = (float)y1; I.th0 = (float)th1;  // params['x0'] -> float32 (mppi.py:214-234)
    if (L.win_active) {
      const long xi0 = (long)floor(((double)I.x0 - L.xlo) / L.res), yi0 = (long)floor(((double)I.y0 - L.ylo) / L.res);
      long r0 = yi0 - L.reach, c0 = xi0 - L.reach;
      r0 = r0 < 0 ? 0 : r0;
      c0 = (c0 < 0 ? 0 : c0) / 8 * 8;
      const long rmax = (long)L.map_rows - L.win_rows, cmax = (long)L.map_pitch - L.win_cols;
      I.win_r0 = (int)(r0 < rmax ? r0 : rmax);
      I.win_c0 = (int)(c0 < cmax ? c0 : cmax);
    }
    inst[b] = I;
    const double dx = x1 - (double)I.xg, dy = y1 - (double)I.yg;
    if (sqrt(dx * dx + dy * dy) <= L.goal_tolerance) {
      L.done[b] = step + 1;
      reached_now = 1;
      atomicAdd_system(L.done_count, 1);
      __threadfence_system();
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t + 1 < n_steps; t += blockDim.x) ub[t] = shifted[t + 1];
  // The host looks at done_count only every few steps, and the planner keeps optimising the
  // controls of a finished problem until then: what the notebook's loop leaves behind -- the last
  // solution, shifted once -- is put aside here and restored when the loop ends (k_world_restore).
  if (reached_now) {
    float2* uf = L.u_final + (size_t)b * n_steps;
    for (int t = threadIdx.x; t < n_steps; t += blockDim.x) uf[t] = shifted[min(t + 1, n_steps - 1)];
  }
}

__global__ void k_world_restore(WorldLoop L, float2* __restrict__ u, float2* __restrict__ u_prev, int n_steps) {
  const int b = blockIdx.x;
  if (!L.done[b]) return;
  for (int t = threadIdx.x; t < n_steps; t += blockDim.x) {
    const float2 v = L.u_final[(size_t)b * n_steps + t];
    u[(size_t)b * n_steps + t] = v;
    u_prev[(size_t)b * n_steps + t] = v;
  }
}

}  // namespace mppi
