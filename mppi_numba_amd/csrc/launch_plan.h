// launch_plan.h -- which kernels an iteration launches and with what geometry: packing of the map cells,
// LDS window planning, the regime selection of the rollout kernels, the update / exchange launches, the
// iteration loop with its hipGraph replay.  Host code only; the C ABI shims that call into it are in
// mppi_api.hip (which is the only file that includes this one: everything here has internal linkage).
#pragma once
#include "handles.h"

static int tdm_max_byte(const mppi_tdm* t) { return t->injected ? (int)t->injected_max : t->table_max; }

// ---- helpers -------------------------------------------------------------------
static int check_tdms(const mppi_planner* p, const mppi_tdm* lin, const mppi_tdm* ang) {
  if (p->cfg.mode == MPPI_MODE_BAREBONE) return MPPI_OK;
  REQUIRE(lin && ang, MPPI_ERR_INVALID, "lin/ang TDM required in this mode");
  REQUIRE(lin->maps_set && ang->maps_set, MPPI_ERR_STATE, "TDM maps not set");
  REQUIRE(lin->cfg.device == p->cfg.device && ang->cfg.device == p->cfg.device, MPPI_ERR_INVALID,
          "planner and TDMs live on different devices");
  REQUIRE(lin->rows == ang->rows && lin->cols == ang->cols, MPPI_ERR_INVALID,
          "lin and ang TDMs differ in padded size (%dx%d vs %dx%d)", lin->rows, lin->cols, ang->rows,
          ang->cols);
  REQUIRE(lin->cfg.num_grids == p->cfg.num_grid_samples && ang->cfg.num_grids == p->cfg.num_grid_samples,
          MPPI_ERR_INVALID, "TDM num_grids (%d, %d) != planner num_grid_samples (%d)", lin->cfg.num_grids,
          ang->cfg.num_grids, p->cfg.num_grid_samples);
  REQUIRE(lin->cfg.max_rows == ang->cfg.max_rows && lin->cfg.max_cols == ang->cfg.max_cols, MPPI_ERR_INVALID,
          "lin and ang TDMs differ in max_map_dim");
  REQUIRE(p->cfg.mode != MPPI_MODE_SPEED_MAP || lin->has_risk, MPPI_ERR_STATE,
          "speed-map mode needs lin TDM's risk traction map");
  return MPPI_OK;
}

static DevParams make_dev_params(const mppi_planner* p, const mppi_tdm* lin, const mppi_tdm* ang) {
  const mppi_params& a = p->params;
  DevParams d;
  memset(&d, 0, sizeof(d));
  d.x0 = a.x0[0]; d.y0 = a.x0[1]; d.th0 = a.x0[2];
  d.xg = a.xgoal[0]; d.yg = a.xgoal[1];
  d.v_lo = a.vrange[0]; d.v_hi = a.vrange[1];
  d.w_lo = a.wrange[0]; d.w_hi = a.wrange[1];
  d.dt = a.dt;
  d.gt2 = a.goal_tolerance * a.goal_tolerance;  // float32 product (mppi.py:960)
  d.lambda = a.lambda_weight;
  d.obs_cost = a.obs_cost;
  d.unk_cost = a.unknown_cost;
  d.res = a.res > 0.f ? a.res : 1.f;
  d.inv_res = 1.0f / d.res;
  d.xlo = a.xlo; d.ylo = a.ylo;
  d.cvar_alpha = a.cvar_alpha;
  d.numel = (int)std::ceil((double)p->cfg.num_grid_samples * (double)a.cvar_alpha);
  if (d.numel < 1) d.numel = 1;
  if (d.numel > p->cfg.num_grid_samples) d.numel = p->cfg.num_grid_samples;
  // (samples sharded over GPUs: the local kernel's own reduction is not used; cvar_numel())
  d.dist_weight = a.dist_weight;
  d.v_post_den = (double)a.v_post_rollout + 1e-6;
  if (lin) { d.lin_lo = lin->lo; d.lin_ratio = lin->ratio; d.rows = lin->rows; d.cols = lin->cols; }
  if (ang) { d.ang_lo = ang->lo; d.ang_ratio = ang->ratio; }
  d.lin_zero_byte = -1;
  for (int b = 0; b < 128 && lin; ++b)
    if (std::fma(d.lin_ratio, (double)b, d.lin_lo) == 0.0) { d.lin_zero_byte = b; break; }
  d.lin_max_byte = lin ? tdm_max_byte(lin) : 0;
  d.ang_max_byte = ang ? tdm_max_byte(ang) : 0;
  d.s0sq = (double)a.u_std[0] * (double)a.u_std[0];
  d.s1sq = (double)a.u_std[1] * (double)a.u_std[1];
  d.n_local = p->n_local;
  d.n_steps = p->cfg.num_steps;
  d.n_grids = p->cfg.num_grid_samples;
  d.n_obstacles = p->n_obstacles;
  d.inst = p->inst_set ? p->inst_dev : nullptr;
  d.inst_tiles = p->inst_tiles;
  d.n_inst = p->n_inst;
  d.spec_failures = p->spec_fail_dev;
  d.cc_k0 = (float)((double)a.lambda_weight / d.s0sq);
  d.cc_k1 = (float)((double)a.lambda_weight / d.s1sq);
  d.inv_v_post_den = 1.0 / d.v_post_den;
  d.neg_log2e_over_lambda = -1.4426950408889634 / (double)a.lambda_weight;
  return d;
}

static int reserve_cells(mppi_planner* p, const mppi_tdm* lin, int M) {
  size_t need = (size_t)lin->rows * lin->cols * M;
  if (need > p->cells_capacity) {
    HIP_TRY(hipStreamSynchronize(p->stream));
    dev_free(p->cells);
    p->cells_capacity = 0;  // (stays 0 if the allocation below fails)
    TRY(dev_alloc(&p->cells, need));
    p->cells_capacity = need;
  }
  return MPPI_OK;
}

// solve() of a CVaR planner, Philox generators: both TDMs sampled straight into the cell words
// the rollout gathers (k_sample_cellsM_philox), no (G, R, C) int8 grids and no transpose; the
// int8 grids follow on demand from the same counters (tdm_materialize).  Returns false when
// the ordinary sample + pack path has to run.
static bool sample_into_cells(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, double alpha_dyn, int* rc) {
  *rc = MPPI_OK;
  static const bool disabled = getenv("MPPI_NO_FUSED_SAMPLING") != nullptr;  // developer switch (ablation)
  if (disabled || p->cfg.mode != MPPI_MODE_TDM || lin == ang) return false;
  if (lin->cfg.rng != MPPI_RNG_PHILOX || ang->cfg.rng != MPPI_RNG_PHILOX) return false;
  if (lin->bins > 64 || ang->bins > 64 || !(alpha_dyn > 0.0)) return false;
  if (lin->first_sample != ang->first_sample) return false;
  const int M = p->cfg.num_grid_samples;
  // a wave of this kernel sets up the thresholds of its 4 cells for M/64 rounds of draws: measured
  // against sample + sample + transpose, 65 vs 61 us at M = 128 and 192 vs 363 us at M = 1024
  if (M < 192) return false;
  if ((*rc = reserve_cells(p, lin, M)) != MPPI_OK) return true;
  const long cell_groups = (long)lin->rows * ((lin->cols + 3) / 4);
  const int bins = std::max(lin->bins, ang->bins);
  dim3 grid((unsigned)ceil_div(cell_groups, 4));
#define MPPI_SAMPLE_CELLS(MAXB)                                                                                    \
  hipLaunchKernelGGL(k_sample_cellsM_philox<MAXB>, grid, dim3(256), 0, p->stream, lin->pmf, lin->bins, lin->table, \
                     lin->cfg.seed, lin->epoch, ang->pmf, ang->bins, ang->table, ang->cfg.seed, ang->epoch,        \
                     lin->obs, lin->unk, lin->rows, lin->cols, alpha_dyn, M, p->cells,                            \
                     (uint64_t)(lin->first_sample >> 1) * (uint64_t)cell_groups)
  if (bins <= 8) MPPI_SAMPLE_CELLS(8);
  else if (bins <= 16) MPPI_SAMPLE_CELLS(16);
  else if (bins <= 32) MPPI_SAMPLE_CELLS(32);
  else MPPI_SAMPLE_CELLS(64);
#undef MPPI_SAMPLE_CELLS
  if (hipGetLastError() != hipSuccess) {
    *rc = fail(MPPI_ERR_HIP, "k_sample_cellsM_philox launch failed");
    return true;
  }
  for (mppi_tdm* t : {lin, ang}) {
    t->sampled_epoch = t->epoch++;
    t->sampled_alpha = alpha_dyn;
    t->sampled_maps_version = t->maps_version;
    t->grid_stale = true;  // the int8 grids of these draws do not exist yet
    t->injected = false;
    ++t->grid_version;
  }
  p->cells16_valid = false;
  p->risk_ref = lin->risk;
  p->packed_lin = lin;
  p->packed_ang = ang;
  p->packed_lin_grid = lin->grid_version;
  p->packed_ang_grid = ang->grid_version;
  if (p->packed_lin_maps != lin->maps_version) p->speculation_off = false;  // a new map: speculate again
  p->packed_lin_maps = lin->maps_version;
  return true;
}

// (re)build the packed cell words when the sampled grids or the masks changed
static int ensure_packed(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang) {
  if (p->cfg.mode == MPPI_MODE_BAREBONE) return MPPI_OK;
  // solve() samples the traction grids itself; the stage-level entry points use what is there
  REQUIRE(lin->grid_version > 0 && ang->grid_version > 0, MPPI_ERR_STATE,
          "traction grids have never been sampled: call mppi_tdm_sample_grids (or mppi_planner_solve) first");
  if (p->packed_lin == lin && p->packed_ang == ang && p->packed_lin_grid == lin->grid_version &&
      p->packed_ang_grid == ang->grid_version && p->packed_lin_maps == lin->maps_version)
    return MPPI_OK;
  const int M = p->cfg.num_grid_samples;
  TRY(reserve_cells(p, lin, M));
  // (another planner may have sampled these TDMs straight into ITS cell words)
  TRY(tdm_materialize(lin, p->stream));
  TRY(tdm_materialize(ang, p->stream));
  p->cells16_valid = false;
  if (M == 1) {
    hipLaunchKernelGGL(k_pack_cells_single, dim3(ceil_div((long)lin->rows * lin->cols, 256)), dim3(256), 0,
                       p->stream, lin->grid, ang->grid, lin->cfg.max_cols, lin->obs, lin->unk, lin->rows,
                       lin->cols, p->cells);
    auto grid_7bit = [](const mppi_tdm* t) {
      return t->injected ? (t->injected_min >= 0) : t->compact_ok;
    };
    if (lin->compact_ok && grid_7bit(lin) && grid_7bit(ang)) {
      p->pitch16 = ceil_div(lin->cols, 8) * 8;
      // speed-map mode: 32-bit cells (16 bits + risk byte) in the same buffer, twice the 16-bit units
      const bool with_risk = p->cfg.mode == MPPI_MODE_SPEED_MAP && lin->has_risk;
      size_t need16 = (size_t)lin->rows * p->pitch16 * (with_risk ? 2 : 1);
      if (need16 > p->cells16_capacity) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        dev_free(p->cells16);
        p->cells16_capacity = 0;  // (stays 0 if the allocation below fails)
        TRY(dev_alloc(&p->cells16, need16));
        p->cells16_capacity = need16;
      }
      if (with_risk)
        hipLaunchKernelGGL(k_pack_cells32_risk, dim3(ceil_div((long)lin->rows * p->pitch16, 256)), dim3(256), 0,
                           p->stream, lin->grid, ang->grid, lin->cfg.max_cols, lin->obs, lin->unk, lin->risk,
                           lin->rows, lin->cols, p->pitch16, reinterpret_cast<uint32_t*>(p->cells16));
      else
        hipLaunchKernelGGL(k_pack_cells16, dim3(ceil_div((long)need16, 256)), dim3(256), 0, p->stream, lin->grid,
                           ang->grid, lin->cfg.max_cols, lin->obs, lin->unk, lin->rows, lin->cols, p->pitch16,
                           p->cells16);
      p->cells16_valid = true;
      p->cells16_with_risk = with_risk;
    }
  } else {
    dim3 grid((unsigned)(lin->rows * ceil_div(lin->cols, 64)), (unsigned)ceil_div(M, 64));
    hipLaunchKernelGGL(k_pack_cells_multi, grid, dim3(256), 0, p->stream, lin->grid, ang->grid,
                       lin->cfg.max_rows, lin->cfg.max_cols, lin->obs, lin->unk, lin->rows, lin->cols, M,
                       p->cells);
  }
  HIP_TRY(hipGetLastError());
  p->risk_ref = lin->risk;
  p->packed_lin = lin;
  p->packed_ang = ang;
  p->packed_lin_grid = lin->grid_version;
  p->packed_ang_grid = ang->grid_version;
  if (p->packed_lin_maps != lin->maps_version) p->speculation_off = false;  // a new map: speculate again
  p->packed_lin_maps = lin->maps_version;
  return MPPI_OK;
}

// describe one noise generation (advances the Philox epoch)
static NoiseJob make_noise_job(mppi_planner* p, float2* target) {
  NoiseJob j;
  j.out = target;
  j.states = (p->cfg.rng == MPPI_RNG_XOROSHIRO) ? p->states : nullptr;
  j.seed = p->cfg.seed;
  // graph mode: the epoch is split into a by-value part that stays the same from one replay to
  // the next and the device-side count of executed updates (bumps_launched mirrors it: every
  // earlier update is ahead of this generator in stream order)
  j.gen_counter = p->graph_on ? (const uint64_t*)p->gen_dev : nullptr;
  j.epoch = p->graph_on ? p->noise_epoch - p->bumps_launched : p->noise_epoch;
  j.n_local = p->n_local;
  j.n_offset = p->n_offset;
  j.n_steps = p->cfg.num_steps;
  j.std0 = p->params.u_std[0];
  j.std1 = p->params.u_std[1];
  if (p->cfg.rng == MPPI_RNG_PHILOX) ++p->noise_epoch;
  return j;
}

static int launch_noise(mppi_planner* p, float2* target, hipStream_t stream = nullptr) {
  long total = (long)noise_items(p->n_local, p->cfg.num_steps, p->cfg.rng == MPPI_RNG_PHILOX);  // one thread per item
  NoiseJob job = make_noise_job(p, target);
  hipLaunchKernelGGL(k_noise, dim3(ceil_div(total, 256)), dim3(256), 0, stream ? stream : p->stream, job);
  HIP_TRY(hipGetLastError());
  return MPPI_OK;
}

// Decide whether the deterministic rollout can keep its map in LDS: the 16-bit cell
// window must cover every cell reachable from x0 within the horizon and fit next to
// the staged controls.  Fills the window fields of `d`.
static bool plan_lds_window(mppi_planner* p, DevParams& d, size_t* lds_bytes) {
  const int T = p->cfg.num_steps;
  const size_t head = sizeof(double2) * ((size_t)T + (size_t)(T + 1) / 2);
  const size_t budget = (size_t)p->lds_per_cu - 1024;  // leave room for the runtime's own use
  if (!p->cells16_valid) return false;
  if (p->cfg.mode == MPPI_MODE_SPEED_MAP ? !p->cells16_with_risk : p->cfg.mode != MPPI_MODE_DET) return false;
  const size_t cell_bytes = p->cells16_with_risk ? sizeof(uint32_t) : sizeof(uint16_t);
  const mppi_params& a = p->params;
  d.pitch16 = p->pitch16;
  const size_t whole = (size_t)d.rows * p->pitch16 * cell_bytes;
  // cells reachable from x0 within the horizon (plus a margin), columns in multiples of 8
  double vmax = std::fmax(std::fabs((double)a.vrange[0]), std::fabs((double)a.vrange[1]));
  double trmax = std::fmax(std::fabs(d.lin_lo), std::fabs(d.lin_lo + (double)d.lin_max_byte * d.lin_ratio));
  double reach_m = (double)T * (double)a.dt * vmax * trmax;
  size_t bytes = whole + 1;
  long r0 = 0, r1 = d.rows, c0 = 0, c1 = p->pitch16;
  if (p->inst_set) {
    // batched handle: one window SIZE for all problems (the full reach square, clipped to the
    // map size), one ORIGIN per problem, shifted inwards at the map border
    for (BatchInst& I : p->inst_host) I.win_r0 = I.win_c0 = 0;
    d.win_step_cells = std::isfinite(reach_m) ? (float)((double)a.dt * vmax * trmax / (double)a.res) : 0.0f;
    d.win_progressive = std::isfinite(reach_m) ? 1 : 0;
    if (std::isfinite(reach_m)) {
      long reach = (long)std::ceil(reach_m / (double)a.res) + 2;
      long wr = std::min((long)d.rows, 2 * reach + 1);
      long wc = std::min((long)p->pitch16, (2 * reach + 1 + 7) / 8 * 8 + 8);
      size_t wbytes = (size_t)wr * (size_t)wc * cell_bytes;
      if (wbytes < whole) {
        if (head + wbytes > budget) return false;
        for (BatchInst& I : p->inst_host) {
          long xi0 = (long)std::floor(((double)I.x0 - (double)a.xlo) / (double)a.res);
          long yi0 = (long)std::floor(((double)I.y0 - (double)a.ylo) / (double)a.res);
          I.win_r0 = (int)std::min(std::max(0L, yi0 - reach), (long)d.rows - wr);
          I.win_c0 = (int)std::min(std::max(0L, xi0 - reach) / 8 * 8, (long)p->pitch16 - wc);
        }
        d.win_r0 = 0; d.win_c0 = 0; d.win_rows = (int)wr; d.win_cols = (int)wc;
        *lds_bytes = head + wbytes;
        return true;
      }
    }
    if (head + whole > budget) return false;
    d.win_r0 = 0; d.win_c0 = 0; d.win_rows = d.rows; d.win_cols = p->pitch16;
    *lds_bytes = head + whole;
    return true;
  }
  if (std::isfinite(reach_m)) {
    long reach = (long)std::ceil(reach_m / (double)a.res) + 2;
    long xi0 = (long)std::floor(((double)a.x0[0] - (double)a.xlo) / (double)a.res);
    long yi0 = (long)std::floor(((double)a.x0[1] - (double)a.ylo) / (double)a.res);
    r0 = std::max(0L, yi0 - reach);
    r1 = std::min((long)d.rows, yi0 + reach + 1);
    c0 = std::max(0L, xi0 - reach) / 8 * 8;
    c1 = std::min((long)p->pitch16, (std::min((long)d.cols, xi0 + reach + 1) + 7) / 8 * 8);
    if (r1 > r0 && c1 > c0) bytes = (size_t)(r1 - r0) * (size_t)(c1 - c0) * cell_bytes;
  }
  // (either way the rollouts spread at most step_cells per step)
  d.win_step_cells = std::isfinite(reach_m) ? (float)((double)a.dt * vmax * trmax / (double)a.res) : 0.0f;
  d.win_progressive = std::isfinite(reach_m) ? 1 : 0;
  if (bytes < whole) {  // the reach window is smaller: less to copy, more LDS left
    if (head + bytes > budget) return false;
    d.win_r0 = (int)r0; d.win_c0 = (int)c0; d.win_rows = (int)(r1 - r0); d.win_cols = (int)(c1 - c0);
    *lds_bytes = head + bytes;
    return true;
  }
  if (head + whole > budget) return false;
  d.win_r0 = 0; d.win_c0 = 0; d.win_rows = d.rows; d.win_cols = p->pitch16;
  *lds_bytes = head + whole;
  return true;
}

// The exact schedule's state role (rollout_kernels.h, PipeWindow<POW2RES = true>) forms its LDS address without the
// [0, last] clamp -- one instruction per coordinate on the horizon's only dependent chain.  That is sound on a map no
// rollout can leave: the reference's padded maps (a ring of zero-traction cells at least as wide as one step is long:
// terrain.py:511-583 -- whoever enters it stays), or a reach square that lies strictly inside the map.  Anything else
// (an mppi_tdm filled through the C API without such a ring, a window clipped by the border) takes the variant with the
// exact floor division and the clamp, i.e. the border cell, like k_rollout_map and k_rollout_fused.
static bool unclamped_lookup_ok(const mppi_planner* p, const DevParams& d) {
  const mppi_params& a = p->params;
  const mppi_tdm* lin = p->packed_lin;
  const double vmax = std::fmax(std::fabs((double)a.vrange[0]), std::fabs((double)a.vrange[1]));
  const double trmax = std::fmax(std::fabs(d.lin_lo), std::fabs(d.lin_lo + (double)d.lin_max_byte * d.lin_ratio));
  const double step_cells = (double)a.dt * vmax * trmax / (double)a.res;
  if (!std::isfinite(step_cells)) return false;
  if (lin) {
    const int ring = lin->injected ? lin->injected_sink_ring : lin->maps_sink_ring;
    if ((double)ring >= std::ceil(step_cells + 1e-3)) return true;
  }
  if (p->inst_set) return false;  // (per-problem windows are shifted inwards at the border)
  const double reach = std::ceil((double)p->cfg.num_steps * step_cells) + 2.0;
  const double xi0 = std::floor(((double)a.x0[0] - (double)a.xlo) / (double)a.res);
  const double yi0 = std::floor(((double)a.x0[1] - (double)a.ylo) / (double)a.res);
  return yi0 - reach > 0.0 && yi0 + reach + 1.0 < (double)d.rows && xi0 - reach > 0.0 && xi0 + reach + 1.0 < (double)d.cols;
}

// Waves (tiles of 64 rollouts) per workgroup of the one-wave-per-tile kernels that keep the map
// window in LDS.  The window makes it one workgroup per CU, so the workgroup is sized to cover
// the problem in one round: at least 4 waves (one per SIMD, and enough lanes to copy the
// window), at most 16.  Batched handle: a workgroup stays inside one problem, i.e. the count
// divides the tiles per problem -- among the divisors the one with the fewest rounds, then the
// smallest (1536 tiles: 8 waves in 1 round, not 4 waves in 2; measured 130 -> 105 us).
static int fused_waves_per_workgroup(const mppi_planner* p, int n_rollouts) {
  const int tiles = ceil_div(n_rollouts, 64);
  int waves = ceil_div(tiles, p->num_cus);
  waves = waves < 4 ? 4 : (waves > 16 ? 16 : waves);
  if (!p->inst_set) return waves;
  auto pick = [&](int lowest) {
    int best = 0, best_rounds = 1 << 30;
    for (int d = lowest; d <= 16; ++d) {
      if (p->inst_tiles % d != 0) continue;
      const int rounds = ceil_div(ceil_div(tiles, d), p->num_cus);
      if (rounds < best_rounds) { best = d; best_rounds = rounds; }  // ascending: ties keep the smallest
    }
    return best;
  };
  const int at_least_four = pick(4);
  return at_least_four ? at_least_four : pick(1);
}

// per-problem start / goal / window origin -> device, when they changed
static int upload_instances(mppi_planner* p) {
  if (!p->inst_set || !p->inst_dirty) return MPPI_OK;
  HIP_TRY(hipMemcpyAsync(p->inst_dev, p->inst_host.data(), sizeof(BatchInst) * (size_t)p->B,
                         hipMemcpyHostToDevice, p->stream));
  p->inst_dirty = false;
  return MPPI_OK;
}

// Rollout and update launches go through the extended launch call: with p->kev_start / kev_stop
// set (mppi_planner_time_kernels) the runtime stamps the dispatch's own begin / end -- what
// rocprofv3 reads -- into those events; with both null it is an ordinary launch.
// (A launch whose noise came from the second stream and that does not look at the generator's flag itself is ordered
//  behind it by the event, here, at the last moment: which kernel runs is decided deep inside the launch paths.)
static void settle_noise_wait(mppi_planner* p) {
  if (!p->noise_wait_pending) return;
  p->noise_wait_pending = false;
  (void)hipStreamWaitEvent(p->stream, p->ev_noise_ready, 0);
}
#define MPPI_KLAUNCH(kernel, grid, block, lds, stream, ...) \
  (settle_noise_wait(p), hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, p->kev_start, p->kev_stop, 0, __VA_ARGS__))
// ... the kernels that do (DevParams::noise_flag; their DevParams through noise_flag_params)
#define MPPI_KLAUNCH_WAITS_ITSELF(kernel, grid, block, lds, stream, ...) \
  hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, p->kev_start, p->kev_stop, 0, __VA_ARGS__)
static DevParams noise_flag_params(mppi_planner* p, DevParams d) {
  static const bool no_flag = getenv("MPPI_NO_NOISE_FLAG") != nullptr;  // developer switch (ablation): the event wait of rounds 1-5
  d.flag_fault = p->flag_fault_dev;
  if (p->noise_wait_pending && p->noise_flag_dev && !no_flag && !p->stream_flags_off) {
    d.noise_flag = p->noise_flag_dev;
    d.noise_flag_expect = p->noise_flag_expect;
    p->noise_wait_pending = false;
  } else {
    settle_noise_wait(p);
  }
  if (p->progress_dev && !no_flag && !p->graph_on && !p->stream_flags_off) {  // (a captured launch would signal a stale number)
    d.progress = p->progress_dev;
    d.progress_value = ++p->progress_seq;
    p->progress_signalled = true;
  }
  return d;
}


// ---- k_rollout_scan (rollout_scan_kernel.h): the time-parallel rollout of MPPI_MATH_FAST --------
// Eligible: deterministic-dynamics mode, 16-bit cells (the reference's own maps always are), a horizon
// of at most 16 waves of 8 steps, LDS for the per-step records, and a map the speculation pays on.
// (struct ScanPlan: handles.h -- the planner caches one)

// Where k_rollout_scan_exact re-executes a tile on the exact schedule: in the LDS of the walks' groups and positions
// (dead by then; unused in direct mode), or behind everything.  Returns the total LDS of the launch, 0: no room.
static size_t scan_fallback_place(const mppi_planner* p, const DevParams& d, int W, size_t lds_now, int* offset, int* map_bytes_out) {
  const size_t map_bytes = (size_t)d.win_rows * (size_t)d.win_cols * sizeof(uint16_t);
  const size_t need = ScanFallback::bytes(8 * W, (int)map_bytes);
  const size_t budget = (size_t)p->lds_per_cu - 1024;
  const size_t total = (ScanExactLds::total(W) + 15) & ~(size_t)15;
  if (need <= ScanExactLds::grp(W) + ScanExactLds::p2(W)) {
    *offset = (int)(ScanExactLds::e2(W) + ScanExactLds::ccr(W));
    *map_bytes_out = (int)map_bytes;
    return lds_now;
  }
  if (total + need <= budget) {
    *offset = (int)total;
    *map_bytes_out = (int)map_bytes;
    return std::max(lds_now, total + need);
  }
  return 0;
}

static bool scan_plan_compute(const mppi_planner* p, ScanPlan* out);

// scan_plan() is asked several times per iteration (which kernel runs, whether it generates its noise, whether the next
// launch can take this one's update, whether the peer exchange is usable): the answer is cached on everything it is
// derived from.  (Batched handles are planned afresh every time: their plan also rewrites the per-problem window origins.)
static bool scan_plan(const mppi_planner* p, ScanPlan* out) {
  struct Key {
    mppi_params params;
    const void *lin, *ang;
    uint64_t lin_grid, ang_grid, lin_maps;
    int debug_flags, speculation_off, cells16_valid, cells16_with_risk, params_set, pad;
  } key;
  if (p->inst_set) return scan_plan_compute(p, out);
  memset(&key, 0, sizeof(key));
  key.params = p->params;
  key.lin = p->packed_lin; key.ang = p->packed_ang;
  key.lin_grid = p->packed_lin_grid; key.ang_grid = p->packed_ang_grid; key.lin_maps = p->packed_lin_maps;
  key.debug_flags = p->debug_flags; key.speculation_off = p->speculation_off; key.cells16_valid = p->cells16_valid;
  key.cells16_with_risk = p->cells16_with_risk; key.params_set = p->params_set;
  mppi_planner* q = const_cast<mppi_planner*>(p);  // (the cache only)
  if (q->scan_key.size() != sizeof(key) || memcmp(q->scan_key.data(), &key, sizeof(key)) != 0) {
    q->scan_cached_ok = scan_plan_compute(p, &q->scan_cached);
    q->scan_key.assign(reinterpret_cast<const unsigned char*>(&key), reinterpret_cast<const unsigned char*>(&key) + sizeof(key));
  }
  if (q->scan_cached_ok && out) *out = q->scan_cached;
  return q->scan_cached_ok;
}

static bool scan_plan_compute(const mppi_planner* p, ScanPlan* out) {
  static const bool disabled = getenv("MPPI_NO_SCAN") != nullptr;  // developer switch (ablation)
  if (disabled || (p->debug_flags & MPPI_DEBUG_NO_SCAN_KERNEL)) return false;
  // (round 6) the speed-map mode too: its dynamics run on nominal traction -- the walks' assumption by construction --
  // and the risk byte comes with the (32-bit) cell; exact arithmetic only, no direct form (rollout_scan_exact_kernel.h)
  const bool speed = p->cfg.mode == MPPI_MODE_SPEED_MAP;
  if (p->cfg.mode != MPPI_MODE_DET && !speed) return false;
  if (!p->cells16_valid || p->cells16_with_risk != speed) return false;
  if (speed && p->cfg.math != MPPI_MATH_EXACT) return false;
  // (MPPI_DEBUG_NO_SPECULATION: "the speculative kernels on their exact schedule from the first step" -- for this kernel
  //  that is the direct launch, whatever the map has shown so far: how the tests reach it on any map)
  const bool stopped = p->speculation_off && !(p->debug_flags & MPPI_DEBUG_KEEP_SPECULATING);
  if (stopped && p->cfg.math != MPPI_MATH_EXACT) return false;  // (the tolerance kernel has no exact schedule inside)
  const bool direct = (stopped || (p->debug_flags & MPPI_DEBUG_NO_SPECULATION)) && p->cfg.math == MPPI_MATH_EXACT;
  static const bool no_direct = getenv("MPPI_NO_SCAN_DIRECT") != nullptr;  // developer switch (ablation): k_rollout_pipe as in round 4
  if (direct && (no_direct || (p->debug_flags & MPPI_DEBUG_NO_SCAN_DIRECT) || p->cfg.math != MPPI_MATH_EXACT ||
                 !p->packed_lin || !p->packed_ang || speed))
    return false;
  const int T = p->cfg.num_steps;
  ScanPlan plan;
  plan.direct = direct;
  plan.exact = p->cfg.math == MPPI_MATH_EXACT;
  plan.chunk_waves = ceil_div(T, 8);
  if (plan.exact && plan.chunk_waves > ScanExactLds::kMaxChunkWaves) return false;  // (T <= 104; beyond, k_rollout_pipe)
  plan.waves = plan.exact ? ScanExactLds::waves(plan.chunk_waves) : plan.chunk_waves;
  if (plan.waves > 16) return false;
  plan.tile = (!plan.exact && (p->debug_flags & MPPI_DEBUG_SCAN_FULL_TILES)) ? 64 : 32;
  // one round of workgroups over the CUs: beyond that the kernels with one wave per tile win (a
  // 32-rollout workgroup lasts ~10 us whatever N is: N = 16384 would be two rounds against 18 us
  // of k_rollout_pipe, N = 65536 eight against 78 us of k_rollout_fused)
  if (ceil_div(p->n_local, plan.tile) > p->num_cus) return false;
  plan.lds = plan.exact ? ScanExactLds::total(plan.chunk_waves)
                        : (plan.tile == 64 ? ScanLds<64>::total(plan.waves) : ScanLds<32>::total(plan.waves));
  // (the accumulating wave reads up to two groups of records past the last one: keep that inside the allocation)
  plan.lds = std::max(plan.lds, (size_t)40 * 1024);
  if (plan.lds > (size_t)p->lds_per_cu - 1024) return false;
  int res_exp = 0;
  plan.pow2res = std::frexp((double)p->params.res, &res_exp) == 0.5;  // res == 2^k exactly
  // (the template flag also selects the unclamped address of the exact schedule inside the kernel -- direct launches
  //  and re-executed tiles: only where no rollout can leave the map)
  const DevParams d0 = make_dev_params(p, p->packed_lin, p->packed_ang);
  // (the speed-map form has no exact schedule inside: its lookups all clamp)
  plan.pow2res = plan.pow2res && (speed || (p->packed_lin && unclamped_lookup_ok(p, d0)));
  if (plan.direct) {
    // the exact schedule needs the 16-bit window of the cells reachable within the horizon in LDS, next to the noise,
    // and the exact-increment rotation (|dt * w * traction| <= 0.36 rad): else k_rollout_pipe / the general kernels
    DevParams d = d0;
    size_t lds_win = 0;
    // (a batched handle's per-problem window origins are recomputed into its host mirror: the values launch_rollout_det
    //  computes from the same start states -- idempotent)
    if (!plan_lds_window(const_cast<mppi_planner*>(p), d, &lds_win)) return false;
    // LDS of a direct launch: noise | control-cost terms | {controls, window, ring} | small arrays (the walks' groups
    // and positions do not exist)
    const int W = plan.chunk_waves;
    const size_t map_bytes = (size_t)d.win_rows * (size_t)d.win_cols * sizeof(uint16_t);
    const size_t need = (ScanFallback::bytes(8 * W, (int)map_bytes) + 15) & ~(size_t)15;
    plan.fallback_offset = (int)(ScanExactLds::e2(W) + ScanExactLds::ccr(W));
    plan.fallback_map_bytes = (int)map_bytes;
    plan.small_offset = plan.fallback_offset + (int)need;
    plan.lds = std::max((size_t)plan.small_offset + ScanExactLds::small(W), (size_t)40 * 1024);
    if (plan.lds > (size_t)p->lds_per_cu - 1024) return false;
  }
  if (out) *out = plan;
  return true;
}

// the iteration loop may let the rollout launch generate its own noise: Philox counters only
static bool scan_generates_noise(const mppi_planner* p) {
  static const bool disabled = getenv("MPPI_SCAN_READ_NOISE") != nullptr;  // developer switch (ablation)
  return !disabled && !(p->debug_flags & MPPI_DEBUG_SCAN_READ_NOISE) && p->cfg.rng == MPPI_RNG_PHILOX &&
         scan_plan(p, nullptr);
}

// the noise of the last iteration into noise_buf when it exists as counters only
static int materialize_noise(mppi_planner* p) {
  if (!p->noise_virtual) return MPPI_OK;
  NoiseJob j;
  memset(&j, 0, sizeof(j));
  j.out = p->noise;
  j.seed = p->cfg.seed;
  j.epoch = p->noise_epoch - (uint64_t)p->noise_virtual_back;  // the block the last rollout launch consumed
  j.n_local = p->n_local;
  j.n_offset = p->n_offset;
  j.n_steps = p->cfg.num_steps;
  j.std0 = p->params.u_std[0];
  j.std1 = p->params.u_std[1];
  const long total = (long)noise_items(p->n_local, p->cfg.num_steps, true);
  hipLaunchKernelGGL(k_noise, dim3(ceil_div(total, 256)), dim3(256), 0, p->stream, j);
  HIP_TRY(hipGetLastError());
  p->noise_virtual = false;
  return MPPI_OK;
}

// this exchange's view of the inboxes; every exchange uses the other set
static PeerExchange make_peer_exchange(mppi_planner* p) {
  PeerExchange X;
  memset(&X, 0, sizeof(X));
  for (int g = 0; g < p->cfg.world_size; ++g) X.inbox[g] = p->peer_inbox[g];
  X.world = p->cfg.world_size;
  X.rank = p->cfg.rank;
  X.fault = p->p2p_fault_dev;
  // how long a rank waits for its peers before it gives the call up (~0.3 us per poll: ~5 s by default; the ranks of a
  // control loop call solve() together, but a host may be late by a map update or a garbage collection)
  static const int max_polls = getenv("MPPI_P2P_MAX_POLLS") ? atoi(getenv("MPPI_P2P_MAX_POLLS")) : (1 << 24);
  X.max_polls = max_polls;
  X.set = p->p2p_index & 1;
  ++p->p2p_index;
  ++p->p2p_exchanges;
  return X;
}

// One GPU: whether a rollout launch over `tiles` workgroups can combine its predecessor's tile packets itself
// (update_kernels.h, PendingApply::reduce_tiles): step t in workgroup t, at most two steps per idle walker wave.
static bool tiles_can_reduce(int tiles, int n_steps) { return 4 * tiles >= n_steps; }

static int launch_scan(mppi_planner* p, const DevParams& d, const ScanPlan& plan_in, bool have_window) {
  ScanPlan plan = plan_in;
  const int N = p->n_local, T = p->cfg.num_steps;
  const bool speed = p->cfg.mode == MPPI_MODE_SPEED_MAP;
  REQUIRE(!speed || (plan.exact && !plan.direct), MPPI_ERR_STATE, "internal: speed-map mode on a kernel that does not serve it");
  const int tiles = ceil_div(N, plan.tile);
  REQUIRE(p->tile_packets[0] && p->tile_packets[1], MPPI_ERR_STATE, "internal: no tile packet buffers on this handle");
  const bool gen = p->scan_gen_now;
  ScanPackets pk;
  pk.tiles = p->tile_packets[p->tpk_cur ^ 1];  // (the other one may be read by this very launch: reduce_pending)
  pk.n_tiles = tiles;
  NoiseJob gen_job, next_job;
  memset(&gen_job, 0, sizeof(gen_job));
  memset(&next_job, 0, sizeof(next_job));
  int extra = 0;
  if (gen) {
    gen_job = make_noise_job(p, nullptr);  // (advances the Philox epoch: this iteration's block)
  } else {
    // a loop that stores its noise (debug switch; the stage-level calls): CUs without a workgroup
    // produce the next iteration's, as in k_rollout_pipe.  (Producing it in the launch's own tail, by
    // the waves that idle while one wave accumulates the costs, was measured: the stage gained is lost
    // again to the slower accumulation and the noise reads -- profiles/r03_scan_notes.md.)
    static const bool no_fused_noise = getenv("MPPI_NO_FUSED_NOISE") != nullptr;  // developer switch
    if (p->next_noise_wanted && tiles < p->num_cus && !no_fused_noise && p->cfg.rng == MPPI_RNG_PHILOX && !plan.exact) {
      extra = p->num_cus - tiles;
      next_job = make_noise_job(p, p->noise_buf[p->noise_cur ^ 1]);
      p->next_noise_done = true;
    }
  }
  // a sharded iteration whose update has not been applied yet: this launch does it (PendingApply)
  PendingApply pend;
  memset(&pend, 0, sizeof(pend));
  if (p->apply_pending || p->reduce_pending) {
    const mppi_params& a = p->params;
    pend.packets = p->packets;
    pend.u_out = p->u_alt;
    pend.u_prev = p->u_prev;
    pend.stats = p->stats;
    pend.world = p->cfg.world_size;
    pend.stride = p->B * packet_len(T);
    pend.lambda = a.lambda_weight;
    pend.v_lo = a.vrange[0]; pend.v_hi = a.vrange[1];
    pend.w_lo = a.wrange[0]; pend.w_hi = a.wrange[1];
  }
  if (p->reduce_pending) {
    // the previous launch's tile packets, combined and applied by this launch (no update kernel ran); several GPUs:
    // with the peer exchange in between
    REQUIRE(!p->apply_pending, MPPI_ERR_STATE, "internal: tile packets left to a launch that cannot reduce them");
    REQUIRE(tiles_can_reduce(tiles, T) && plan.tile == p->scan_tile, MPPI_ERR_STATE, "internal: %d workgroups cannot combine the tile packets of %d steps", tiles, T);
    pend.packets = p->packets;  // (not read in this mode; non-null: "an update is pending")
    pend.world = 1;
    pend.reduce_tiles = p->tile_packets[p->tpk_cur];
    pend.reduce_n_tiles = ceil_div(N, p->scan_tile);
    pend.published = p->published;
    pend.flag_set = p->reduce_index & 1;
    pend.fault = p->fold_fault_dev;
    pend.max_polls = p->fold_max_polls;
    if (p->p2p_on) pend.peers = make_peer_exchange(p);  // (advances the inbox set)
  }
  if (!plan.direct) p->spec_launches += 1;
  // Exact kernel: where a tile whose vote fails is re-executed (rollout_scan_exact_kernel.h, scan_exact_reexecute):
  // controls, 16-bit map window and one chunk ring in the LDS of the walks' groups and positions, dead by then --
  // or behind everything when the horizon is too short for that region to hold them.
  ScanFallback fallback = {-1, 0, 0, 0, 0};
  {  // the exact-increment rotation of the state role applies (as launch_rollout_det's rot_ok)
    const mppi_params& a = p->params;
    const double wmax = std::fmax(std::fabs((double)a.wrange[0]), std::fabs((double)a.wrange[1]));
    const double trmax = std::fmax(std::fabs(d.ang_lo), std::fabs(d.ang_lo + (double)d.ang_max_byte * d.ang_ratio));
    const double dmax = (double)a.dt * wmax * trmax;
    fallback.rot_ok = (std::isfinite(dmax) && dmax <= 0.36) ? 1 : 0;
  }
  static const bool no_fast_fallback = getenv("MPPI_SCAN_SLOW_FALLBACK") != nullptr;  // developer switch (ablation)
  if (plan.direct) {
    REQUIRE(have_window && plan.fallback_offset >= 0, MPPI_ERR_STATE, "internal: direct exact schedule without its map window");
    fallback.offset = plan.fallback_offset;
    fallback.map_bytes = plan.fallback_map_bytes;
    fallback.small_offset = plan.small_offset;
    fallback.direct = 1;
  } else if (plan.exact && have_window && !no_fast_fallback && !speed) {  // (speed-map: 32-bit cells, no exact schedule inside)
    const size_t lds = scan_fallback_place(p, d, plan.chunk_waves, plan.lds, &fallback.offset, &fallback.map_bytes);
    if (lds) plan.lds = lds;
  }
  // Developer experiment (profiles/r06_overlap_notes.md; TIMING ONLY -- the launches race on the tile packets): every
  // other launch goes to the second stream with nothing ordering it behind its predecessor, i.e. the upper bound of what
  // "launch k+1 before launch k ends"  could hide.
  static const bool alt_streams = getenv("MPPI_EXPERIMENT_ALT_STREAMS") != nullptr;
  static unsigned alt_toggle = 0;
  hipStream_t scan_stream = (alt_streams && (alt_toggle++ & 1)) ? p->noise_stream : p->stream;
#define MPPI_LAUNCH_SCAN_EXACT(P2, GEN)                                                                    \
  do {                                                                                                    \
    auto kern = speed ? k_rollout_scan_exact<P2, GEN, false, true>                                        \
                      : (plan.direct ? k_rollout_scan_exact<P2, GEN, true> : k_rollout_scan_exact<P2, GEN, false>); \
    if (plan.lds > 64 * 1024)                                                                             \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds));            \
    MPPI_KLAUNCH(kern, dim3(tiles), dim3(64 * plan.waves), plan.lds, scan_stream, d, p->cells16, p->cells, \
                 p->noise, gen_job, p->u, p->costs, p->w_rel, pk, pend, fallback,                          \
                 (const int8_t*)(speed ? p->risk_ref : nullptr));                                          \
  } while (0)
#define MPPI_LAUNCH_SCAN(RR, P2, GEN)                                                                      \
  do {                                                                                                    \
    auto kern = k_rollout_scan<RR, P2, GEN>;                                                              \
    if (plan.lds > 64 * 1024)                                                                             \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds));            \
    MPPI_KLAUNCH(kern, dim3(tiles + extra), dim3(64 * plan.waves), plan.lds, p->stream, d, p->cells16,    \
                 p->noise, gen_job, p->u, p->costs, p->w_rel, pk, tiles, next_job, pend);                  \
  } while (0)
#define MPPI_LAUNCH_SCAN_G(RR, P2)               \
  do {                                           \
    if (gen) MPPI_LAUNCH_SCAN(RR, P2, true);     \
    else MPPI_LAUNCH_SCAN(RR, P2, false);        \
  } while (0)
  if (plan.exact) {
    if (plan.pow2res && gen) MPPI_LAUNCH_SCAN_EXACT(true, true);
    else if (plan.pow2res) MPPI_LAUNCH_SCAN_EXACT(true, false);
    else if (gen) MPPI_LAUNCH_SCAN_EXACT(false, true);
    else MPPI_LAUNCH_SCAN_EXACT(false, false);
  } else if (plan.tile == 64 && plan.pow2res) MPPI_LAUNCH_SCAN_G(64, true);
  else if (plan.tile == 64) MPPI_LAUNCH_SCAN_G(64, false);
  else if (plan.pow2res) MPPI_LAUNCH_SCAN_G(32, true);
  else MPPI_LAUNCH_SCAN_G(32, false);
#undef MPPI_LAUNCH_SCAN_G
#undef MPPI_LAUNCH_SCAN
#undef MPPI_LAUNCH_SCAN_EXACT
  HIP_TRY(hipGetLastError());
  const bool applied_here = p->apply_pending || p->reduce_pending;
  if (applied_here) {  // the first tile's workgroup has written the updated sequence into the other buffer
    std::swap(p->u, p->u_alt);
    p->u_parity ^= 1;
    if (p->reduce_pending) {
      ++p->reduce_index;
      ++p->reduced_applies;
    } else {
      ++p->folded_applies;
    }
    p->apply_pending = p->reduce_pending = false;
  }
  p->tpk_cur ^= 1;
  char buf[384];
  snprintf(buf, sizeof(buf),
           "k_rollout_scan%s tile=%d waves=%d pow2res=%d noise=%s lds=%zu noise_blocks=%d problems=%d%s%s",
           plan.exact ? (speed ? "_exact speed_map" : "_exact") : "", plan.tile, plan.waves, (int)plan.pow2res, gen ? "in-kernel" : "read", plan.lds, extra,
           p->inst_set ? p->B : 0, !plan.exact ? "" : (plan.direct ? " direct=1 (exact three-wave schedule, no speculation)" : (fallback.offset >= 0 ? " failed_tiles=pipelined" : " failed_tiles=one_wave")), applied_here ? (pend.reduce_tiles ? " applies_update=1 reduces_tiles=1" : " applies_update=1") : "");
  p->last_rollout = buf;
  p->tile_packets_fresh = false;  // (w_rel is relative to this kernel's own tiles: tbeta, not tile_beta)
  p->scan_packets_fresh = true;
  p->scan_tile = plan.tile;
  p->noise_virtual = gen;
  p->noise_virtual_back = 1 + (next_job.out ? 1 : 0);  // (the launch may have produced its successor's block too)
  return MPPI_OK;
}

// ---- deterministic-dynamics mode: which rollout kernel runs ------------------
// Decided per launch from measurements (profiles/r06_families.md; history: r01_ablation.md, r02_stamps.md, r03_scan_notes.md):
//   every tile of 32 rollouts a CU, T <= 104    k_rollout_scan_exact / k_rollout_scan (time-parallel; launch_scan above)
//   one or two tiles of 64 per CU               k_rollout_pipe   (exact three-wave schedule)
//   beyond (throughput regime)                  k_rollout_fused  (one wave per tile, 4..16 per CU)
//   no LDS window / no incremental trig         k_rollout_map    (general)
// (Rounds 2-5 had two speculative pipelines between the first two, k_rollout_deep and k_rollout_spec; re-measured at the
//  end of round 6 against the exact pipeline and the throughput kernel on one box neither won by 5 % anywhere it was still
//  selected -- k_rollout_spec lost by a factor of two at T = 200 -- and both were removed: profiles/r06_families.md.)
struct DetRegime {
  bool have_window;       // the 16-bit cell window reachable within the horizon fits in LDS
  size_t lds_win;         // ... bytes of {staged controls, window}
  bool rot_ok;            // incremental trig applies (|dt*w*traction| <= 0.36 rad, T <= 2000, exact math)
  bool pow2res;           // resolution is a power of two: four-instruction cell coordinates
  bool pow2res_unclamped; // ... and no rollout can leave the map: the exact schedule's address without a clamp (unclamped_lookup_ok)
  bool rot_ok_fast;       // ... the same bound under MPPI_MATH_FAST (k_rollout_fused<one pass>)
};

template <bool EXACT>
static int try_launch_pipe(mppi_planner* p, DevParams& d, const DetRegime& r, bool* launched) {
  *launched = false;
  const int N = p->n_local, T = p->cfg.num_steps;
  [[maybe_unused]] const bool have_window = r.have_window, rot_ok = r.rot_ok, pow2res = r.pow2res_unclamped;
  [[maybe_unused]] const size_t lds_win = r.lds_win;
  static const bool no_pipe = getenv("MPPI_NO_PIPE") != nullptr;  // developer switch (ablation)
  if (have_window && rot_ok && !no_pipe) {
    // pipelined kernel: the map window in LDS + the incremental trig
    const size_t map_bytes = lds_win - sizeof(double2) * ((size_t)T + (size_t)(T + 1) / 2);
    int pairs = ceil_div(ceil_div(N, 64), p->num_cus);  // wave triples per workgroup
    if (pairs < 1) pairs = 1;
    if (pairs > kPipeMaxTriples) pairs = kPipeMaxTriples + 1;  // (beyond the kernel's launch bounds: the throughput kernel's)
    // batched handle: the triples of a workgroup share one problem's window and controls
    if (p->inst_set) while (p->inst_tiles % pairs != 0) --pairs;
    const size_t budget = (size_t)p->lds_per_cu - 1024;
    auto ring_bytes = [&](int chunk) {
      return (size_t)pairs * (2 * (size_t)chunk * 64 * (sizeof(float2) + sizeof(double2)) + 2 * (size_t)chunk * 64 * 2);
    };
    int chunk = 0;
    for (;;) {  // fewer triples per workgroup (more workgroups than CUs) before giving the kernel up
      // (6: the whole-map window of the long horizons -- 135 KiB at T = 200 -- leaves room for chunks of six steps but
      //  not of eight; every chunk costs ~340 cycles beside its steps, so six instead of four is 28 cycles per step)
      for (int cnd : {8, 6, 4, 2})
        if (lds_win + ring_bytes(cnd) <= budget) { chunk = cnd; break; }
      if (chunk > 0 || pairs == 1) break;
      --pairs;
      if (p->inst_set) while (p->inst_tiles % pairs != 0) --pairs;
    }
    // The pipelined kernel is the low-latency choice: it wins while one workgroup per CU covers the problem with one
    // wave triple, and with two when chunks of four steps at least fit beside the window (measured at the end of round 6,
    // N x T, us per iteration pipe | fused: 8192 x 200 44 | 79; 16384 x 100 39 | 53; 16384 x 200 63 | 85;
    // 32768 x 100 55 | 60; 32768 x 200 (chunks of two) 102 | 77; 49152 x 100 (three triples) 74 | 48:
    // profiles/r06_families.md).  Beyond that the fused kernel, 4..16 waves per CU, has the better throughput.
    const bool latency_regime = pairs <= kPipeMaxTriples && (pairs == 1 || chunk >= 4) && ceil_div(ceil_div(N, 64), pairs) <= p->num_cus;
    if (chunk > 0 && latency_regime) {
      // control-cost products in LDS when there is room, else in a global scratch array
      const size_t cc_bytes = (size_t)pairs * T * 64 * sizeof(double);
      const bool cc_lds = lds_win + ring_bytes(chunk) + cc_bytes <= budget && !(p->debug_flags & MPPI_DEBUG_CC_GLOBAL);
      const size_t lds_total = lds_win + ring_bytes(chunk) + (cc_lds ? cc_bytes : 0);
      const int block = 192 * pairs;
      const int grid = ceil_div(N, 64 * pairs);
      // spare CUs generate the next iteration's noise inside this launch
      NoiseJob next_job;
      memset(&next_job, 0, sizeof(next_job));
      int extra = 0;
      static const bool no_fused_noise = getenv("MPPI_NO_FUSED_NOISE") != nullptr;  // developer switch
      if (p->next_noise_wanted && grid < p->num_cus && !no_fused_noise) {  // (no spare CU otherwise: in line)
        extra = p->num_cus - grid;
        next_job = make_noise_job(p, p->noise_buf[p->noise_cur ^ 1]);
        p->next_noise_done = true;
      }
      if (!cc_lds && !p->cc_scratch) TRY(dev_alloc(&p->cc_scratch, (size_t)ceil_div(N, 64) * 64 * T));
#define MPPI_LAUNCH_PIPE(CH, P2, CL)                                                                  \
  do {                                                                                                \
auto kern = k_rollout_pipe<CH, P2, CL>;                                                           \
if (lds_total > 64 * 1024)                                                                        \
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total));       \
MPPI_KLAUNCH(kern, dim3(grid + extra), dim3(block), lds_total, p->stream, d, p->cells16,    \
                   p->noise, p->u, p->costs, p->w_rel, p->tile_beta, p->cc_scratch,                \
                   (int)map_bytes, grid, next_job);                                                \
  } while (0)
#define MPPI_LAUNCH_PIPE_C(P2, CL)            \
  do {                                        \
if (chunk == 8) MPPI_LAUNCH_PIPE(8, P2, CL);      \
else if (chunk == 6) MPPI_LAUNCH_PIPE(6, P2, CL); \
else if (chunk == 4) MPPI_LAUNCH_PIPE(4, P2, CL); \
else MPPI_LAUNCH_PIPE(2, P2, CL);                 \
  } while (0)
      if (pow2res && cc_lds) MPPI_LAUNCH_PIPE_C(true, true);
      else if (pow2res) MPPI_LAUNCH_PIPE_C(true, false);
      else if (cc_lds) MPPI_LAUNCH_PIPE_C(false, true);
      else MPPI_LAUNCH_PIPE_C(false, false);
#undef MPPI_LAUNCH_PIPE_C
      {
        char buf[256];
        snprintf(buf, sizeof(buf),
                 "k_rollout_pipe chunk=%d pow2res=%d cc_lds=%d triples_per_wg=%d window=%dx%d@(%d,%d) lds=%zu "
                 "noise_blocks=%d problems=%d",
                 chunk, (int)pow2res, (int)cc_lds, pairs, d.win_rows, d.win_cols, d.win_r0, d.win_c0, lds_total,
                 extra, p->inst_set ? p->B : 0);
        p->last_rollout = buf;
      }
#undef MPPI_LAUNCH_PIPE
      p->tile_packets_fresh = true;
      *launched = true;
      return MPPI_OK;
    }
  }
  return MPPI_OK;
}

template <bool EXACT, bool BOUNDED>
static int launch_windowed_or_general(mppi_planner* p, DevParams& d, const DetRegime& r) {
  const int N = p->n_local, T = p->cfg.num_steps;
  [[maybe_unused]] const bool have_window = r.have_window, rot_ok = r.rot_ok, pow2res = r.pow2res;
  [[maybe_unused]] const size_t lds_win = r.lds_win;
  const size_t lds_map = sizeof(double2) * ((size_t)T + (size_t)(T + 1) / 2);  // + staged u[t]
  static const bool no_window = getenv("MPPI_NO_WINDOW") != nullptr;  // developer switch (ablation)
  if (have_window && !no_window) {
    // the window makes it one workgroup per CU: size the workgroup so that the grid
    // is at most one wave of workgroups over the CUs
    // (at least 4 waves: one per SIMD, and four times the lanes to copy the window)
    const int waves = fused_waves_per_workgroup(p, N);
    int block = 64 * waves;
    static const bool no_fused = getenv("MPPI_NO_FUSED") != nullptr;  // developer switch (ablation)
    if ((rot_ok || r.rot_ok_fast) && !no_fused) {
      // (MPPI_MATH_FAST: one pass over the noise, the control cost added once)
      // (round 6: workgroups of at most four waves -- nobody hides a wave's trips to memory: k_rollout_fused, LONE)
      static const bool no_lone = getenv("MPPI_NO_FUSED_LONE") != nullptr;  // developer switch (ablation)
      const bool lone = EXACT && waves <= 4 && !no_lone;
      auto fused = EXACT ? (lone ? (pow2res ? k_rollout_fused<true, false, false, true> : k_rollout_fused<false, false, false, true>)
                                 : (pow2res ? k_rollout_fused<true> : k_rollout_fused<false>))
                         : (pow2res ? k_rollout_fused<true, false, true> : k_rollout_fused<false, false, true>);
      // Round 6 (exact mode reads the noise twice): what the window leaves of the CU's LDS keeps the noise of the first
      // steps for the second pass -- one workgroup per CU either way (DevParams::stash_steps; N = 65536, T = 100: 56 of
      // the 100 steps of each of the four waves)
      static const bool no_stash = getenv("MPPI_NO_NOISE_STASH") != nullptr;  // developer switch (ablation)
      size_t lds_fused = lds_win;
      d.stash_steps = d.stash_offset = 0;
      if (EXACT && !no_stash) {
        const size_t off = (lds_win + 15) & ~(size_t)15, limit = (size_t)p->lds_per_cu - 1024;
        if (limit > off) {
          const int fit = (int)((limit - off) / ((size_t)waves * 64 * sizeof(float2)));
          const int steps = std::min(fit, T) & ~7;
          if (steps >= 8) {
            d.stash_steps = steps;
            d.stash_offset = (int)off;
            lds_fused = off + (size_t)waves * (size_t)steps * 64 * sizeof(float2);
          }
        }
      }
      if (lds_fused > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fused),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fused));
      MPPI_KLAUNCH_WAITS_ITSELF(fused, dim3(ceil_div(N, block)), dim3(block), lds_fused, p->stream, noise_flag_params(p, d), p->cells16,
                                p->noise, p->u, p->costs, p->w_rel, p->tile_beta);
      char buf[200];
      snprintf(buf, sizeof(buf), "k_rollout_fused%s pow2res=%d waves_per_wg=%d window=%dx%d problems=%d noise_kept=%d",
               EXACT ? "" : "<one pass>", (int)pow2res, waves, d.win_rows, d.win_cols, p->inst_set ? p->B : 0, d.stash_steps);
      p->last_rollout = buf;
      p->tile_packets_fresh = true;
            return MPPI_OK;
    }
    auto kern = k_rollout_map<MAP_DET, EXACT, BOUNDED, true>;
    if (lds_win > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
    MPPI_KLAUNCH(kern, dim3(ceil_div(N, block)), dim3(block), lds_win, p->stream, d, p->cells,
                       p->cells16, (const int8_t*)nullptr, p->noise, p->u, p->costs);
    {
      char buf[200];
      snprintf(buf, sizeof(buf), "k_rollout_map det lds_window exact=%d waves_per_wg=%d window=%dx%d problems=%d",
               (int)EXACT, waves, d.win_rows, d.win_cols, p->inst_set ? p->B : 0);
      p->last_rollout = buf;
    }
  } else {
    MPPI_KLAUNCH((k_rollout_map<MAP_DET, EXACT, BOUNDED, false>), dim3(ceil_div(N, 64)), dim3(64),
                       lds_map, p->stream, d, p->cells, (const uint16_t*)nullptr, (const int8_t*)nullptr,
                       p->noise, p->u, p->costs);
    p->last_rollout = "k_rollout_map det global_cells exact=" + std::to_string((int)EXACT);
  }
  return MPPI_OK;
}

template <bool EXACT, bool BOUNDED>
static int launch_rollout_det(mppi_planner* p, DevParams d) {
  const int T = p->cfg.num_steps;
  p->tile_packets_fresh = false;
  size_t lds_win = 0;
  bool have_window = plan_lds_window(p, d, &lds_win);
  TRY(upload_instances(p));
  {
    ScanPlan plan;
    if (scan_plan(p, &plan)) return launch_scan(p, d, plan, have_window);
  }
  // incremental trig: needs a heading increment |dt*w*traction| <= 0.36 rad and T <= 2000
  bool rot_ok = false, rot_ok_fast = false, pow2res = false;
  {
    const mppi_params& a = p->params;
    double wmax = std::fmax(std::fabs((double)a.wrange[0]), std::fabs((double)a.wrange[1]));
    double trmax = std::fmax(std::fabs(d.ang_lo), std::fabs(d.ang_lo + (double)d.ang_max_byte * d.ang_ratio));
    double dmax = (double)a.dt * wmax * trmax;
    rot_ok = EXACT && BOUNDED && std::isfinite(dmax) && dmax <= 0.36 && T <= 2000;
    rot_ok_fast = !EXACT && p->theta_bounded && std::isfinite(dmax) && dmax <= 0.36 && T <= 2000;
    int res_exp = 0;
    pow2res = std::frexp((double)a.res, &res_exp) == 0.5;  // res == 2^k exactly
  }
  DetRegime r;
  r.have_window = have_window; r.lds_win = lds_win; r.rot_ok = rot_ok; r.rot_ok_fast = rot_ok_fast; r.pow2res = pow2res;
  r.pow2res_unclamped = pow2res && unclamped_lookup_ok(p, d);
  bool launched = false;
  TRY(try_launch_pipe<EXACT>(p, d, r, &launched));
  if (!launched) TRY((launch_windowed_or_general<EXACT, BOUNDED>(p, d, r)));
  HIP_TRY(hipGetLastError());
  return MPPI_OK;
}

template <bool EXACT, bool BOUNDED>
static int launch_rollout_speed_map(mppi_planner* p, DevParams d) {
  const int N = p->n_local, T = p->cfg.num_steps;
  [[maybe_unused]] const int M = p->cfg.num_grid_samples;
  [[maybe_unused]] size_t lds = sizeof(double2) * (size_t)T;
  [[maybe_unused]] const size_t lds_map = sizeof(double2) * ((size_t)T + (size_t)(T + 1) / 2);  // + staged u[t]
  p->tile_packets_fresh = false;
  size_t lds_win = 0;
  const bool have_window = plan_lds_window(p, d, &lds_win);  // 32-bit cells: 16 bits + risk byte
  TRY(upload_instances(p));
  {
    // latency regime (every tile of 32 rollouts a CU, T <= 104): the time-parallel kernel, one launch per iteration
    ScanPlan plan;
    if (scan_plan(p, &plan)) {
      d.pitch16 = p->pitch16;
      return launch_scan(p, d, plan, have_window);
    }
  }
  const mppi_params& a = p->params;
  double wmax = std::fmax(std::fabs((double)a.wrange[0]), std::fabs((double)a.wrange[1]));
  double trmax = std::fmax(std::fabs(d.ang_lo), std::fabs(d.ang_lo + (double)d.ang_max_byte * d.ang_ratio));
  double dmax = (double)a.dt * wmax * trmax;
  static const bool no_fused = getenv("MPPI_NO_FUSED") != nullptr;  // developer switch (ablation)
  if (have_window && (BOUNDED || (!EXACT && p->theta_bounded)) && std::isfinite(dmax) && dmax <= 0.36 && T <= 2000 && !no_fused) {
    const int waves = fused_waves_per_workgroup(p, N);
    int res_exp = 0;
    const bool pow2res = std::frexp((double)a.res, &res_exp) == 0.5;
    auto fused = EXACT ? (pow2res ? k_rollout_fused<true, true> : k_rollout_fused<false, true>)
                       : (pow2res ? k_rollout_fused<true, true, true> : k_rollout_fused<false, true, true>);
    if (lds_win > 64 * 1024)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fused),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
    MPPI_KLAUNCH_WAITS_ITSELF(fused, dim3(ceil_div(N, 64 * waves)), dim3(64 * waves), lds_win, p->stream, noise_flag_params(p, d), p->cells16,
                              p->noise, p->u, p->costs, p->w_rel, p->tile_beta);
    char buf[200];
    snprintf(buf, sizeof(buf), "k_rollout_fused%s speed_map pow2res=%d waves_per_wg=%d window=%dx%d problems=%d",
             EXACT ? "" : "<one pass>", (int)pow2res, waves, d.win_rows, d.win_cols, p->inst_set ? p->B : 0);
    p->last_rollout = buf;
    p->tile_packets_fresh = true;
    HIP_TRY(hipGetLastError());
    return MPPI_OK;
  }
  MPPI_KLAUNCH((k_rollout_map<MAP_SPEED, EXACT, BOUNDED, false>), dim3(ceil_div(N, 64)), dim3(64),
                     lds_map, p->stream, d, p->cells, (const uint16_t*)nullptr, (const int8_t*)p->risk_ref,
                     p->noise, p->u, p->costs);
  p->last_rollout = "k_rollout_map speed_map global_cells exact=" + std::to_string((int)EXACT);
  HIP_TRY(hipGetLastError());
  return MPPI_OK;
}

template <bool EXACT, bool BOUNDED>
static int launch_rollout_tdm(mppi_planner* p, DevParams d) {
  const int N = p->n_local, T = p->cfg.num_steps;
  [[maybe_unused]] const int M = p->cfg.num_grid_samples;
  [[maybe_unused]] size_t lds = sizeof(double2) * (size_t)T;
  [[maybe_unused]] const size_t lds_map = sizeof(double2) * ((size_t)T + (size_t)(T + 1) / 2);  // + staged u[t]
  int mp2 = next_pow2(M);
  int threads = ceil_div(M, 64) * 64;
  if (threads > 1024) threads = 1024;
  lds += sizeof(float) * (size_t)mp2;
  REQUIRE(lds <= 160 * 1024, MPPI_ERR_INVALID, "T=%d, M=%d need %zu bytes of LDS (> 160 KiB)", T, M, lds);
  if (lds > 64 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rollout_tdm<EXACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  if (p->want_sample_costs && !p->sample_costs && p->m_count == 1) TRY(dev_alloc(&p->sample_costs, (size_t)N * M));
  if (p->m_count > 1 && !p->slabs) TRY(dev_alloc(&p->slabs, (size_t)p->m_count * N * M));
  // sharded samples: the per-sample costs go into this rank's slab of the gather buffer
  float* const sc_dst = p->m_count > 1 ? p->slabs + (size_t)p->m_rank * N * M
                                       : (p->want_sample_costs ? p->sample_costs : nullptr);
  p->tile_packets_fresh = false;
  TRY(upload_instances(p));
  {
    const mppi_params& a = p->params;
    double wmax = std::fmax(std::fabs((double)a.wrange[0]), std::fabs((double)a.wrange[1]));
    double trmax = std::fmax(std::fabs(d.ang_lo), std::fabs(d.ang_lo + (double)d.ang_max_byte * d.ang_ratio));
    double dmax = (double)a.dt * wmax * trmax;
    const size_t lds_fast = (sizeof(double2) + sizeof(double)) * (size_t)T + sizeof(float) * (size_t)mp2;
    if ((BOUNDED || (!EXACT && p->theta_bounded)) && std::isfinite(dmax) && dmax <= 0.36 && T <= 2000 && lds_fast <= 64 * 1024) {
      int res_exp = 0;
      const bool pow2res = std::frexp((double)a.res, &res_exp) == 0.5;
      float* sc_out = sc_dst;
      // the next iteration's noise by workgroups appended to the grid (they run in the launch's tail)
      NoiseJob next_job;
      memset(&next_job, 0, sizeof(next_job));
      int extra = 0;
      static const bool no_fused_noise = getenv("MPPI_NO_FUSED_NOISE") != nullptr;  // developer switch
      if (p->next_noise_wanted && !no_fused_noise && p->cfg.rng == MPPI_RNG_PHILOX) {
        const long rows = (long)(noise_items(p->n_local, T, true) >> 6);
        extra = (int)std::min<long>(2L * p->num_cus, ceil_div(rows, (long)(threads / 64) * 4));
        if (extra > 0) {
          next_job = make_noise_job(p, p->noise_buf[p->noise_cur ^ 1]);
          p->next_noise_done = true;
        }
      }
      if (pow2res)
        MPPI_KLAUNCH((k_rollout_tdm_fast<true, !EXACT>), dim3(N + extra), dim3(threads), lds_fast, p->stream, d, p->cells,
                           p->noise, p->u, p->costs, sc_out, mp2, N, next_job);
      else
        MPPI_KLAUNCH((k_rollout_tdm_fast<false, !EXACT>), dim3(N + extra), dim3(threads), lds_fast, p->stream, d, p->cells,
                           p->noise, p->u, p->costs, sc_out, mp2, N, next_job);
      p->last_rollout = std::string(EXACT ? "k_rollout_tdm_fast" : "k_rollout_tdm_fast<cost f32>") + " pow2res=" +
                        (pow2res ? "1" : "0") + " noise_blocks=" + std::to_string(extra);
      HIP_TRY(hipGetLastError());
      return MPPI_OK;
    }
  }
  MPPI_KLAUNCH((k_rollout_tdm<EXACT>), dim3(N), dim3(threads), lds, p->stream, d, p->cells, p->noise,
                     p->u, p->costs, sc_dst, mp2);
  p->last_rollout = "k_rollout_tdm exact=" + std::to_string((int)EXACT);
  HIP_TRY(hipGetLastError());
  return MPPI_OK;
}

template <bool EXACT, bool BOUNDED>
static int launch_rollout_t(mppi_planner* p, DevParams d) {
  p->scan_packets_fresh = false;
  if (p->noise_virtual && !p->scan_gen_now) TRY(materialize_noise(p));  // the coming kernel reads its noise
  switch (p->cfg.mode) {
    case MPPI_MODE_DET: return launch_rollout_det<EXACT, BOUNDED>(p, d);
    case MPPI_MODE_SPEED_MAP: return launch_rollout_speed_map<EXACT, BOUNDED>(p, d);
    case MPPI_MODE_TDM: return launch_rollout_tdm<EXACT, BOUNDED>(p, d);
    case MPPI_MODE_BAREBONE: {
      const int N = p->n_local;
      p->tile_packets_fresh = false;
      // (cos, sin) by rotation where the host can bound the heading increment: |dt * w| <= 0.36 rad, T <= 2000
      const mppi_params& a = p->params;
      const double dmax = (double)a.dt * std::fmax(std::fabs((double)a.wrange[0]), std::fabs((double)a.wrange[1]));
      const bool rot = EXACT && std::isfinite(dmax) && dmax <= 0.36 && p->cfg.num_steps <= 2000;
      const size_t lds_bb = sizeof(double2) * (size_t)p->cfg.num_steps + sizeof(float4) * (size_t)std::max(1, p->n_obstacles);
      REQUIRE(lds_bb <= 64 * 1024, MPPI_ERR_INVALID, "%d disc obstacles and %d steps: more than 64 KiB of LDS", p->n_obstacles, p->cfg.num_steps);
      static const bool no_kd = getenv("MPPI_BAREBONE_NO_KD") != nullptr;  // developer switch (ablation)
      if (rot && !no_kd && p->n_obstacles <= 2)
        MPPI_KLAUNCH((k_rollout_barebone<EXACT, true, 2>), dim3(ceil_div(N, 64)), dim3(64), lds_bb + 2 * sizeof(float4),
                     p->stream, d, p->obs_pos, p->obs_r, p->noise, p->u, p->costs);
      else if (rot && !no_kd && p->n_obstacles <= 4)
        MPPI_KLAUNCH((k_rollout_barebone<EXACT, true, 4>), dim3(ceil_div(N, 64)), dim3(64), lds_bb + 4 * sizeof(float4),
                     p->stream, d, p->obs_pos, p->obs_r, p->noise, p->u, p->costs);
      else if (rot)
        MPPI_KLAUNCH((k_rollout_barebone<EXACT, true>), dim3(ceil_div(N, 64)), dim3(64), lds_bb,
                     p->stream, d, p->obs_pos, p->obs_r, p->noise, p->u, p->costs);
      else
        MPPI_KLAUNCH((k_rollout_barebone<EXACT, false>), dim3(ceil_div(N, 64)), dim3(64), lds_bb,
                     p->stream, d, p->obs_pos, p->obs_r, p->noise, p->u, p->costs);
      p->last_rollout = "k_rollout_barebone exact=" + std::to_string((int)EXACT) + " rotation=" + std::to_string((int)rot);
      break;
    }
    default:
      return fail(MPPI_ERR_INVALID, "bad mode");
  }
  HIP_TRY(hipGetLastError());
  return MPPI_OK;
}


static int launch_apply(mppi_planner* p);
static bool next_rollout_reduces_tiles(const mppi_planner* p);
static int settle_reduce_pending(mppi_planner* p);

static int launch_rollout(mppi_planner* p, const DevParams& d) {
  REQUIRE(p->B == 1 || p->inst_set, MPPI_ERR_STATE,
          "num_instances = %d: call mppi_planner_set_instances before solving", p->B);
  REQUIRE((size_t)p->cfg.num_steps * sizeof(double2) <= 64 * 1024, MPPI_ERR_INVALID, "num_steps %d too large",
          p->cfg.num_steps);
  // an update left to its consumer (launch_update) and a rollout kernel that will not take it
  if (p->apply_pending && !(p->cfg.mode == MPPI_MODE_DET && scan_plan(p, nullptr))) TRY(launch_apply(p));  // (sharded updates fold in deterministic-dynamics mode only)
  if (p->reduce_pending && !next_rollout_reduces_tiles(p)) TRY(settle_reduce_pending(p));
  // |theta| can never exceed |theta0| + T*dt*max|w|*max(traction): when that is far
  // inside the range of the two-term pi/2 reduction, the kernels drop the libm branch
  const mppi_params& a = p->params;
  double wmax = std::fmax(std::fabs((double)a.wrange[0]), std::fabs((double)a.wrange[1]));
  double trmax = std::fmax(std::fabs(d.ang_lo), std::fabs(d.ang_lo + (double)d.ang_max_byte * d.ang_ratio));
  if (p->cfg.mode == MPPI_MODE_BAREBONE) trmax = 1.0;
  double th0_max = std::fabs((double)a.x0[2]);
  if (p->inst_set) {
    th0_max = 0.0;
    for (const BatchInst& I : p->inst_host) th0_max = std::fmax(th0_max, std::fabs((double)I.th0));
  }
  double theta_bound = th0_max + (double)p->cfg.num_steps * (double)a.dt * wmax * trmax;
  bool bounded = std::isfinite(theta_bound) && theta_bound < 5.0e4;
  p->theta_bounded = bounded;
  // mppi_planner_time_kernels on a two-stream loop: the kernels of the throughput regime stamp their own begin / end
  DevParams dk = d;
  dk.ktime = p->ktime_rollout_slot;
  dk.ktime_waves = p->ktime_waves;
  if (p->cfg.math != MPPI_MATH_EXACT) return launch_rollout_t<false, false>(p, dk);
  return bounded ? launch_rollout_t<true, true>(p, dk) : launch_rollout_t<true, false>(p, dk);
}

// tile-relative weights (unless the rollout kernel just emitted them) + the row kernel:
// applies the update on a single GPU; with several GPUs leaves this rank's packet in
// packets[rank] for the exchange
// ---- CVaR mode, samples sharded over GPUs: all-gather of the (N, M/G) cost slabs, then every
//      rank reduces all N control samples over all M costs ---------------
static int cvar_numel(const mppi_planner* p) {
  const int M = p->cfg.num_grid_samples * p->m_count;
  int numel = (int)std::ceil((double)M * (double)p->params.cvar_alpha);  // mppi.py:716-717 over all M samples
  return numel < 1 ? 1 : (numel > M ? M : numel);
}

static int launch_cvar_reduce(mppi_planner* p) {
  const int N = p->n_local, Ml = p->cfg.num_grid_samples, M = Ml * p->m_count;
  const int mp2 = next_pow2(M);
  const int threads = mp2 > 1024 ? 1024 : (mp2 < 64 ? 64 : mp2);  // one element per thread when it fits
  const size_t lds = sizeof(float) * (size_t)mp2;
  REQUIRE(lds <= 64 * 1024, MPPI_ERR_INVALID, "M = %d samples over all shards: too many for the CVaR reduction", M);
  if (p->want_sample_costs && p->sample_costs == nullptr) TRY(dev_alloc(&p->sample_costs, (size_t)N * M));
  hipLaunchKernelGGL(k_cvar_reduce, dim3(N), dim3(threads), lds, p->stream, p->slabs, p->m_count, N, Ml, cvar_numel(p),
                     p->params.cvar_alpha, p->costs, p->want_sample_costs ? p->sample_costs : (float*)nullptr, mp2);
  HIP_TRY(hipGetLastError());
  p->tile_packets_fresh = false;
  p->sample_costs_local_only = false;
  return MPPI_OK;
}

// inside the iteration loop: RCCL all-gather of the slabs on the planner's stream, then the reduction
static int exchange_sample_costs(mppi_planner* p) {
  if (p->m_count <= 1) return MPPI_OK;
  REQUIRE(p->comm, MPPI_ERR_STATE,
          "samples sharded over %d ranks but no communicator: call mppi_planner_comm_init "
          "(or drive rollout / sample_costs_local / sample_costs_apply / update yourself)", p->m_count);
  const size_t len = (size_t)p->n_local * p->cfg.num_grid_samples;
  RCCL_TRY(g_rccl.AllGather(p->slabs + (size_t)p->m_rank * len, p->slabs, len, ncclFloat, p->comm, p->stream));
  return launch_cvar_reduce(p);
}

static int launch_update_local(mppi_planner* p, bool apply_here) {
  const int N = p->n_local, T = p->cfg.num_steps;
  const mppi_params& a = p->params;
  double* my_packet = p->packets + (size_t)p->cfg.rank * p->B * packet_len(T);
  if (p->scan_packets_fresh) {
    // the rollout launch (k_rollout_scan) has reduced w_rel * noise over every tile: combine the tiles
    p->scan_packets_fresh = false;
    p->tile_packets_fresh = false;
    const dim3 grid(T, p->B);
    unsigned long long* gen = p->graph_on ? p->gen_dev : (unsigned long long*)nullptr;
    const int per_problem = ceil_div(p->n_inst, p->scan_tile);
    // (graph replay: this launch also accounts for the iterations whose update ran inside a rollout launch)
    const unsigned long long bump = 1ull + (unsigned long long)p->bumps_owed;
    const float* tiles = p->tile_packets[p->tpk_cur];
    PeerExchange peers;
    memset(&peers, 0, sizeof(peers));
    if (apply_here && p->p2p_on && p->cfg.world_size > 1) peers = make_peer_exchange(p);  // (exchanged inside the launch)
    if (apply_here)
      MPPI_KLAUNCH((k_combine_tiles<true>), grid, dim3(64), 0, p->stream, tiles, per_problem,
                   T, a.lambda_weight, my_packet, p->u, p->u_prev, (p->mirror_now ? p->u_host_dev : (float2*)nullptr), a.vrange[0], a.vrange[1],
                   a.wrange[0], a.wrange[1], p->stats, gen, bump, p->published, peers);
    else
      MPPI_KLAUNCH((k_combine_tiles<false>), grid, dim3(64), 0, p->stream, tiles, per_problem,
                   T, a.lambda_weight, my_packet, p->u, p->u_prev, (p->mirror_now ? p->u_host_dev : (float2*)nullptr), a.vrange[0], a.vrange[1],
                   a.wrange[0], a.wrange[1], p->stats, gen, bump, p->published, peers);
    if (p->graph_on) p->bumps_launched += bump;
    p->bumps_owed = 0;
    p->reduce_index = 0;  // (the flags are all clear again)
    HIP_TRY(hipGetLastError());
    return MPPI_OK;
  }
  if (p->noise_virtual) TRY(materialize_noise(p));  // the row kernel streams the noise
  // rollout kernels without the weight epilogue: the row kernel forms the tile weights itself
  // from the costs (same bits) unless there are too many tiles for its LDS arrays
  const bool from_costs = !p->tile_packets_fresh && 2 * sizeof(float) * (size_t)p->inst_tiles <= 60 * 1024;
  if (!p->tile_packets_fresh && !from_costs)
    MPPI_KLAUNCH(k_tile_weights, dim3(p->n_tiles), dim3(64), 0, p->stream, p->costs, N, a.lambda_weight,
                       p->w_rel, p->tile_beta);
  p->tile_packets_fresh = false;
  const size_t lds = sizeof(float) * (size_t)p->inst_tiles * (from_costs ? 2 : 1);
  REQUIRE(lds <= 60 * 1024, MPPI_ERR_INVALID, "too many rollouts per GPU for the update kernel (%d)", N);
  // rows per workgroup: see k_update_rows
  const bool many_rows = (long)T * p->B >= 2048;
  const dim3 grid(many_rows ? ceil_div(T, 4) : T, p->B);
// (slim: 256 threads with four times the loads in flight each and at most two workgroups per CU -- 60 KB of LDS each --,
//  for the launches beside which the next iteration's noise is generated: k_update_rows, THREADS)
#define MPPI_LAUNCH_ROWS(APPLY, TC, FC)                                                                         \
  do {                                                                                                          \
    if (slim)                                                                                                   \
      MPPI_KLAUNCH((k_update_rows<APPLY, TC, FC, 256>), grid, dim3(256), std::max(lds, (size_t)60 * 1024), p->stream, \
                   FC ? p->costs : p->w_rel, p->tile_beta, p->n_inst, p->inst_tiles, p->noise, T,               \
                   a.lambda_weight, my_packet, p->u, p->u_prev, (p->mirror_now ? p->u_host_dev : (float2*)nullptr), a.vrange[0], a.vrange[1], \
                   a.wrange[0], a.wrange[1], p->stats, p->graph_on ? p->gen_dev : (unsigned long long*)nullptr, \
                   p->ktime_update_slot, p->ktime_waves, p->progress_dev, p->progress_seq);                     \
    else                                                                                                        \
      MPPI_KLAUNCH((k_update_rows<APPLY, TC, FC>), grid, dim3(kRowThreads), lds, p->stream,                     \
                   FC ? p->costs : p->w_rel, p->tile_beta, p->n_inst, p->inst_tiles, p->noise, T,               \
                   a.lambda_weight, my_packet, p->u, p->u_prev, (p->mirror_now ? p->u_host_dev : (float2*)nullptr), a.vrange[0], a.vrange[1], \
                   a.wrange[0], a.wrange[1], p->stats, p->graph_on ? p->gen_dev : (unsigned long long*)nullptr, \
                   p->ktime_update_slot, p->ktime_waves, (unsigned long long*)nullptr, 0ull);                   \
  } while (0)
#define MPPI_LAUNCH_ROWS_TC(APPLY, FC)        \
  do {                                        \
    if (many_rows) MPPI_LAUNCH_ROWS(APPLY, 4, FC); \
    else MPPI_LAUNCH_ROWS(APPLY, 1, FC);           \
  } while (0)
  const bool slim = p->update_signals;
  if (apply_here && from_costs) MPPI_LAUNCH_ROWS_TC(true, true);
  else if (apply_here) MPPI_LAUNCH_ROWS_TC(true, false);
  else if (from_costs) MPPI_LAUNCH_ROWS_TC(false, true);
  else MPPI_LAUNCH_ROWS_TC(false, false);
#undef MPPI_LAUNCH_ROWS_TC
#undef MPPI_LAUNCH_ROWS
  if (p->graph_on) ++p->bumps_launched;
  HIP_TRY(hipGetLastError());
  p->update_signalled = p->update_signals;
  return MPPI_OK;
}

// (out of place into the handle's other control buffer, like an update applied by its consumer: every
//  sharded iteration flips the two, and an even number of iterations -- a graph -- restores them)
static int launch_apply(mppi_planner* p) {
  const mppi_params& a = p->params;
  hipLaunchKernelGGL(k_apply, dim3(p->B), dim3(kUpdateThreads), 0, p->stream, p->packets, p->cfg.world_size,
                     p->cfg.rank, p->cfg.num_steps, a.lambda_weight, p->u, p->u_alt, p->u_prev,
                     (p->mirror_now ? p->u_host_dev : (float2*)nullptr), a.vrange[0],
                     a.vrange[1], a.wrange[0], a.wrange[1], p->stats);
  HIP_TRY(hipGetLastError());
  std::swap(p->u, p->u_alt);
  p->u_parity ^= 1;
  p->apply_pending = false;
  return MPPI_OK;
}

// Whether the NEXT rollout launch of this handle can apply a sharded update itself (PendingApply).
static bool next_rollout_applies_updates(const mppi_planner* p) {
  static const bool disabled = getenv("MPPI_NO_FOLDED_APPLY") != nullptr;  // developer switch (ablation)
  return !disabled && !(p->debug_flags & MPPI_DEBUG_NO_FOLDED_APPLY) && p->B == 1 && !p->inst_set && p->m_count == 1 &&
         p->cfg.world_size <= kMaxFoldedRanks && p->cfg.mode == MPPI_MODE_DET && !p->mirror_now && scan_plan(p, nullptr);
}

// Several GPUs without a collective: the peer exchange is connected and the kernels that carry it will run (the
// time-parallel kernels leave tile packets; one problem per handle).  Decided from things every rank has alike.
static bool p2p_usable(const mppi_planner* p) {
  ScanPlan plan;
  return p->p2p_on && p->cfg.world_size > 1 && p->cfg.world_size <= kMaxFoldedRanks && p->B == 1 && !p->inst_set &&
         p->m_count == 1 && p->cfg.mode == MPPI_MODE_DET && scan_plan(p, &plan);
}

// Whether the NEXT rollout launch can combine and apply the tile packets of this one (no update kernel).
static bool next_rollout_reduces_tiles(const mppi_planner* p) {
  static const bool disabled = getenv("MPPI_NO_REDUCE_FOLD") != nullptr;  // developer switch (ablation)
  ScanPlan plan;
  return !disabled && !p->fold_off && !(p->debug_flags & (MPPI_DEBUG_NO_FOLDED_APPLY | MPPI_DEBUG_NO_REDUCE_FOLD)) && p->B == 1 && !p->inst_set &&
         p->m_count == 1 && ((p->cfg.world_size == 1 && !p->comm) || p2p_usable(p)) &&
         (p->cfg.mode == MPPI_MODE_DET || p->cfg.mode == MPPI_MODE_SPEED_MAP) && !p->mirror_now &&
         scan_plan(p, &plan) && plan.tile == p->scan_tile &&
         tiles_can_reduce(ceil_div(p->n_local, plan.tile), p->cfg.num_steps);
}

// tile packets left to a rollout launch that will not come (another kernel family, a stage-level call): the
// ordinary combination
static int settle_reduce_pending(mppi_planner* p) {
  p->reduce_pending = false;
  p->scan_packets_fresh = true;
  return launch_update_local(p, true);
}

// `defer_exchange` (mppi_group_iterate_async): stop after this rank's packet; the caller issues the
// all-gathers of all its devices inside one RCCL group and then launches k_apply on each
// `may_leave_apply`: another iteration follows on this stream: the update may be left to its rollout launch
static int launch_update(mppi_planner* p, bool prof, bool defer_exchange = false, bool may_leave_apply = false) {
  p->mirror_done = p->mirror_now;
  if (defer_exchange) return launch_update_local(p, false);
  // (a communicator on a single rank is honoured too: it exercises the same path as N ranks)
  // (samples sharded: every rank holds all N costs and all the noise -- the update is local)
  if ((p->cfg.world_size == 1 && !p->comm) || p->m_count > 1 || (p2p_usable(p) && p->scan_packets_fresh)) {
    if (may_leave_apply && !prof && p->scan_packets_fresh && next_rollout_reduces_tiles(p)) {
      // the next rollout launch reduces this one's tile packets and applies the update itself: no launch here
      p->scan_packets_fresh = false;
      p->reduce_pending = true;
      if (p->graph_on) ++p->bumps_owed;
      return MPPI_OK;
    }
    TRY(launch_update_local(p, true));
    if (prof) {
      HIP_TRY(hipEventRecord(p->ev_stage[3], p->stream));
      HIP_TRY(hipEventRecord(p->ev_stage[4], p->stream));
    }
    return MPPI_OK;
  }
  REQUIRE(p->comm, MPPI_ERR_STATE,
          "world_size %d but no communicator: call mppi_planner_comm_init (or use update_local/update_apply)",
          p->cfg.world_size);
  TRY(launch_update_local(p, false));
  const int len = p->B * packet_len(p->cfg.num_steps);
  if (prof) HIP_TRY(hipEventRecord(p->ev_stage[3], p->stream));
  // one all-gather of (2T+2) doubles per problem and iteration, in place
  TraceRange tr("mppi:all_gather_packets");
  RCCL_TRY(g_rccl.AllGather(p->packets + (size_t)p->cfg.rank * len, p->packets, (size_t)len, ncclDouble, p->comm,
                            p->stream));
  if (prof) HIP_TRY(hipEventRecord(p->ev_stage[4], p->stream));
  if (may_leave_apply && !prof && next_rollout_applies_updates(p)) {
    p->apply_pending = true;
    return MPPI_OK;
  }
  return launch_apply(p);
}

// One iteration: {noise unless it was produced ahead, rollout (+ the next iteration's noise when
// `want_next`), update}.  `have_noise`: noise_buf[noise_cur ^ 1] already holds this iteration's
// noise; on return it says the same for the following iteration.
static int launch_iteration(mppi_planner* p, const DevParams& d, bool& have_noise, bool want_next, bool prof,
                            bool defer_exchange = false, bool may_leave_apply = false) {
  // (below ~4M rollout-steps the generator takes less than the ~12 us a cross-stream dependency costs)
  static const bool no_side_stream = getenv("MPPI_NO_SIDE_STREAM") != nullptr;  // developer switch
  // and above 8 rollout waves per CU the register file has room for one generator wave per SIMD only: the generator
  // crawls beside the rollout, slows it, and still collides with the update (measured in round 1, profiles/r01_ablation.md,
  // and again in round 6 with MPPI_SIDE_STREAM_MAX_WAVES=16: C5 187 us against 178, profiles/r06_ns_notes.md section 3)
  static const int side_max_waves = getenv("MPPI_SIDE_STREAM_MAX_WAVES") ? atoi(getenv("MPPI_SIDE_STREAM_MAX_WAVES")) : 8;  // developer switch
  const bool side_stream_pays = (long)p->n_local * p->cfg.num_steps >= 4L * 1000 * 1000 &&
                                ceil_div(ceil_div(p->n_local, 64), p->num_cus) <= side_max_waves && !no_side_stream;
  const bool ktime_stamps = p->ktime_index >= 0 && p->ktime_dev && p->ktime_use_stamps;  // (mppi_planner_time_kernels)
  if (prof) HIP_TRY(hipEventRecord(p->ev_stage[0], p->stream));
  // MPPI_MATH_FAST over a map the time-parallel kernel takes: the rollout launch computes its noise
  // from the Philox counters itself; nothing is generated ahead, nothing is stored
  const bool gen_in_rollout = scan_generates_noise(p);
  p->scan_gen_now = gen_in_rollout;
  if (gen_in_rollout) {
    if (have_noise) discard_noise_ahead(p);  // (produced ahead by an earlier, different kind of launch)
    have_noise = false;
    want_next = false;
  } else if (have_noise) {
    p->noise_cur ^= 1;
    // (ordered behind the generator by the rollout launch itself: its flag, or the event -- settle_noise_wait)
    if (p->noise_on_side_stream) p->noise_wait_pending = true;
    p->noise_on_side_stream = false;
    p->noise_virtual = false;
  } else {
    TraceRange tr("mppi:noise");
    p->noise_virtual = false;
    TRY(launch_noise(p, p->noise_buf[p->noise_cur]));
  }
  p->noise = p->noise_buf[p->noise_cur];
  p->next_noise_wanted = want_next;
  p->next_noise_done = false;
  if (prof) HIP_TRY(hipEventRecord(p->ev_stage[1], p->stream));
  // the other noise buffer was last read by the previous update, which is behind us on this stream
  // The generator of the next iteration's noise (second stream, below) must not start before the previous update -- the
  // last reader of the buffer it overwrites -- is complete.  A rollout launch that signals its own start (DevParams::
  // progress) provides that without anything in front of it on this stream; the event recorded here held the launch back
  // by ~6 us.  Whether the coming launch signals is known once it has been made (the kernel is chosen deep inside the
  // launch paths): the event is left out when the previous iteration's launch did, and made up for behind the launch
  // (one iteration without overlap) should this one not.
  p->progress_signalled = false;
  bool buf_free_recorded = false;
  if (want_next && side_stream_pays && (!p->progress_capable_last || p->graph_on)) {
    HIP_TRY(hipEventRecord(p->ev_buf_free, p->stream));
    buf_free_recorded = true;
  }
  {
    TraceRange tr("mppi:rollout");
    // mppi_planner_time_kernels.  A loop with the second stream in it is timed by the kernels themselves (device
    // stamps, no event anywhere near the launches -- see there); every other loop by start / stop events on the launches.
    if (ktime_stamps) {
      p->ktime_markers = true;
      p->ktime_rollout_slot = p->ktime_dev + 4 * (size_t)p->ktime_waves * (size_t)p->ktime_index;
    } else if (p->ktime_index >= 0) {
      p->kev_start = p->ktime_events[4 * (size_t)p->ktime_index];
      p->kev_stop = p->ktime_events[4 * (size_t)p->ktime_index + 1];
    }
    const int rc = launch_rollout(p, d);
    p->kev_start = p->kev_stop = nullptr;
    p->ktime_rollout_slot = nullptr;
    settle_noise_wait(p);  // (a path that launched nothing: whatever follows on the stream still needs the noise)
    TRY(rc);
  }
  if (p->m_count > 1) {
    TraceRange tr("mppi:exchange_sample_costs");
    TRY(exchange_sample_costs(p));
  }
  have_noise = p->next_noise_done;
  if (want_next && !have_noise && side_stream_pays) {
    if (buf_free_recorded) {
      HIP_TRY(hipStreamWaitEvent(p->noise_stream, p->ev_buf_free, 0));
    } else if (p->progress_signalled) {
      hipLaunchKernelGGL(k_wait_progress, dim3(1), dim3(64), 0, p->noise_stream, p->progress_dev, p->progress_seq, p->flag_fault_dev);
      HIP_TRY(hipGetLastError());
    } else {  // (neither: ordered behind the rollout launch itself)
      HIP_TRY(hipEventRecord(p->ev_buf_free, p->stream));
      HIP_TRY(hipStreamWaitEvent(p->noise_stream, p->ev_buf_free, 0));
    }
    TraceRange tr("mppi:noise_ahead");
    TRY(launch_noise(p, p->noise_buf[p->noise_cur ^ 1], p->noise_stream));
    p->used_side_stream = true;
    if (p->noise_flag_dev && !p->graph_on) {
      p->noise_flag_expect = ++p->noise_flag_seq;
      if (!(p->debug_flags & MPPI_DEBUG_DROP_NOISE_FLAG)) {  // (test hook: a generator the consumer never hears of)
        hipLaunchKernelGGL(k_set_noise_flag, dim3(1), dim3(1), 0, p->noise_stream, p->noise_flag_dev, p->noise_flag_expect);
        HIP_TRY(hipGetLastError());
      }
    }
    HIP_TRY(hipEventRecord(p->ev_noise_ready, p->noise_stream));
    have_noise = p->noise_on_side_stream = true;
    if (p->graph_on) {
      // graph mode: join before the update, which advances the epoch counter the generator reads
      // (and a captured iteration must not leave a fork open)
      HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_noise_ready, 0));
      p->noise_on_side_stream = false;
    }
  }
  p->progress_capable_last = p->progress_signalled;
  // Where the rollout fills the SIMDs by itself (more than 8 of its waves per CU: the batched handles, C5) a generator
  // beside it only takes its issue slots; but the update behind it is bound by memory: there the next iteration's noise is
  // generated beside the UPDATE launch -- which runs slim (k_update_rows<.., 256>: a quarter of the waves, the same bytes
  // in flight) and signals its start to the generator's gate kernel.
  static const bool no_beside_update = getenv("MPPI_NO_NOISE_BESIDE_UPDATE") != nullptr ||  // developer switch (ablation)
                                       getenv("MPPI_NO_NOISE_FLAG") != nullptr;             // (no flags: no second stream here)
  const bool beside_update = want_next && !have_noise && !side_stream_pays && !no_side_stream && !no_beside_update && !p->graph_on &&
                             !prof && !defer_exchange && p->noise_flag_dev && p->progress_dev && !p->stream_flags_off &&
                             p->cfg.rng == MPPI_RNG_PHILOX && (long)p->n_local * p->cfg.num_steps >= 4L * 1000 * 1000 &&
                             !p->scan_packets_fresh && p->cfg.world_size == 1 && !p->comm && p->m_count == 1 &&
                             2 * sizeof(float) * (size_t)p->inst_tiles <= 60 * 1024;
  p->update_signals = beside_update;
  p->update_signalled = false;
  if (beside_update) ++p->progress_seq;
  p->next_noise_wanted = false;
  p->scan_gen_now = false;
  if (prof) HIP_TRY(hipEventRecord(p->ev_stage[2], p->stream));
  TraceRange tr_update("mppi:update");
  const int ktime_slot = p->ktime_index;
  if (p->ktime_index >= 0) {
    if (ktime_stamps) {
      p->ktime_update_slot = p->ktime_dev + 4 * (size_t)p->ktime_waves * (size_t)p->ktime_index + 2 * (size_t)p->ktime_waves;
    } else {
      p->kev_start = p->ktime_events[4 * (size_t)p->ktime_index + 2];
      p->kev_stop = p->ktime_events[4 * (size_t)p->ktime_index + 3];
    }
    ++p->ktime_index;
  }
  {
    const int rc = launch_update(p, prof, defer_exchange, may_leave_apply);
    p->kev_start = p->kev_stop = nullptr;
    p->ktime_update_slot = nullptr;
    p->update_signals = false;
    TRY(rc);
    if (p->update_signalled) {  // the generator, gated on this update launch's start
      p->update_signalled = false;
      hipLaunchKernelGGL(k_wait_progress, dim3(1), dim3(64), 0, p->noise_stream, p->progress_dev, p->progress_seq, p->flag_fault_dev);
      HIP_TRY(hipGetLastError());
      TraceRange tr("mppi:noise_beside_update");
      TRY(launch_noise(p, p->noise_buf[p->noise_cur ^ 1], p->noise_stream));
      p->used_side_stream = true;
      p->noise_flag_expect = ++p->noise_flag_seq;
      hipLaunchKernelGGL(k_set_noise_flag, dim3(1), dim3(1), 0, p->noise_stream, p->noise_flag_dev, p->noise_flag_expect);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipEventRecord(p->ev_noise_ready, p->noise_stream));
      have_noise = p->noise_on_side_stream = true;
    }
    // (an update left to the next rollout launch has no launch of its own to time)
    if (ktime_slot >= 0 && (size_t)ktime_slot < p->ktime_update_ran.size()) p->ktime_update_ran[(size_t)ktime_slot] = !p->reduce_pending;
  }
  if (prof) HIP_TRY(hipEventRecord(p->ev_stage[5], p->stream));
  return MPPI_OK;
}

// Everything the launches of an iteration take by value or derive on the host: a captured graph
// may be replayed only while none of it has changed.
static void graph_signature(const mppi_planner* p, const DevParams& d, const mppi_tdm* lin, const mppi_tdm* ang,
                            std::vector<unsigned char>& out) {
  struct Sig {
    DevParams d;
    mppi_params params;
    const void *lin, *ang, *cells, *cells16, *cc, *sample_costs, *u;
    uint64_t lin_grid, ang_grid, lin_maps, epoch_bias;
    int noise_cur, inst_set, want_sample_costs, speculation_off, debug_flags, pad;
  } sig;
  memset(&sig, 0, sizeof(sig));
  sig.d = d;
  sig.params = p->params;
  if (p->inst_set) {  // batched handle: start and goal are read from device memory, not from arguments
    sig.d.x0 = sig.d.y0 = sig.d.th0 = sig.d.xg = sig.d.yg = 0.0f;
    memset(sig.params.x0, 0, sizeof(sig.params.x0));
    memset(sig.params.xgoal, 0, sizeof(sig.params.xgoal));
  }
  sig.lin = lin; sig.ang = ang; sig.cells = p->cells; sig.cells16 = p->cells16; sig.cc = p->cc_scratch;
  sig.sample_costs = p->sample_costs; sig.u = p->u;
  sig.lin_grid = p->packed_lin_grid; sig.ang_grid = p->packed_ang_grid; sig.lin_maps = p->packed_lin_maps;
  sig.epoch_bias = p->noise_epoch - p->bumps_launched;
  sig.noise_cur = p->noise_cur; sig.inst_set = p->inst_set; sig.want_sample_costs = p->want_sample_costs;
  sig.speculation_off = p->speculation_off ? 1 : 0; sig.debug_flags = p->debug_flags;
  sig.pad = (p->p2p_on ? 2 : 0) | (p->p2p_index & 1);  // (the inbox set of the peer exchange is a by-value argument)
  out.assign(reinterpret_cast<unsigned char*>(&sig), reinterpret_cast<unsigned char*>(&sig) + sizeof(sig));
}

// `timed`: bracket the iterations with events for mppi_planner_last_elapsed_ms / stage_times
// (iterate_async, profiling); solve() on the control path skips them
// called where the host has just waited for the stream: did speculation pay on this map?
static void review_speculation(mppi_planner* p) {
  if (!p->spec_fail_host) return;
  if (p->spec_launches == 0) {  // (nothing speculative ran: whatever the word holds is stale)
    *p->spec_fail_host = 0u;
    return;
  }
  const uint64_t failed = *p->spec_fail_host;
  // A launch lasts as long as its slowest tile, and a tile whose vote fails is rolled out twice: ONE failing tile
  // makes its launch slower than the exact schedule would have been.  Every speculative kernel runs ONE round of
  // workgroups (k_rollout_scan*: a tile per CU; the speculative pipelines of rounds 2-5 likewise: up to three tiles in
  // one workgroup per CU), so the criterion is per launch for all of them.  C2 shape, round 5: 15.2 us when every
  // vote holds, 21.6 us on the exact schedule (direct), ~40 us with a failing tile -- speculation pays while
  // P(fail) * (40 - 21.6) < (1 - P) * (21.6 - 15.2), i.e. fewer than one launch in four fails (the longer horizons'
  // speculative against exact pipeline then: 36 / 48 / 85 us, the same quarter).  The kernels count failed tiles: at
  // least one per four launches -> stop.
  if (4 * failed >= p->spec_launches) {
    p->speculation_off = true;
    // Several GPUs on the peer exchange: which tiles fail their vote differs from rank to rank (every rank rolls out
    // its own noise), so ranks reach this point in different iterations -- and the exchange only works while all of
    // them launch kernels that carry it.  Stopping is therefore allowed where the kernel FAMILY survives it
    // (k_rollout_scan_exact on its exact schedule: ScanPlan::direct -- same packets, same exchange, whatever the
    // other ranks run); where it would mean another family (window too large, tolerance kernel) this rank keeps
    // speculating: slower on such a map, never a rank that has left the exchange its peers still wait in.
    if (p->p2p_on && p->cfg.world_size > 1 && !scan_plan(p, nullptr)) p->speculation_off = false;
  }
  *p->spec_fail_host = 0u;
  p->spec_launches = 0;
}

// `part_of_group_loop` (mppi_group_iterate_async's round robin over devices): this call is one turn of a longer loop
// on this handle -- `more_follow`: further turns come, so the last iteration of this one may leave its update to the
// next turn's first rollout launch like any other iteration -- and the loop's events belong to the caller.
static int run_iterations(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, int iterations, bool timed = true,
                          bool mirror_last = false, bool part_of_group_loop = false, bool more_follow = false) {
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  TRY(check_tdms(p, lin, ang));
  TRY(ensure_packed(p, lin, ang));
  DevParams d = make_dev_params(p, lin, ang);
  p->iterations_since_wait += iterations;
  timed = (timed || p->profile_stages) && !part_of_group_loop;
  if (timed) HIP_TRY(hipEventRecord(p->ev_begin, p->stream));
  // The noise of iteration k+1 does not depend on iteration k.  When the pipelined rollout
  // kernel runs, its spare workgroups generate it into the other half of the double buffer
  // (same launch, no extra dependency); otherwise it is generated in line.
  // (a sharded handle replays too: RCCL's all-gather is captured into the graph with the kernels;
  //  without a communicator the exchange is host-staged and cannot be captured)
  const bool use_graph = p->graph_on && !p->profile_stages && (p->cfg.world_size == 1 || p->comm);
  if (!use_graph) {
    // Every iteration, the last one of a call included, asks for its successor's noise: when the
    // rollout kernel can produce it on the side (spare workgroups, second stream) the next call --
    // the next control step -- starts with its noise already there (`primed`).
    bool have_noise = p->primed;
    for (int k = 0; k < iterations; ++k) {
      // profiled iteration: a steady-state one when there is one, else the last
      bool prof = p->profile_stages && !part_of_group_loop && k == (iterations >= 3 ? iterations - 2 : iterations - 1);
      p->mirror_now = mirror_last && k == iterations - 1;
      const int rc = launch_iteration(p, d, have_noise, true, prof, false, k + 1 < iterations || more_follow);
      p->mirror_now = false;
      TRY(rc);
    }
    p->primed = have_noise;
  } else {
    // Graph mode.  Every iteration also asks for the noise of its successor (`primed`; kernels that
    // cannot produce it ahead generate in line instead), so that all iterations look alike; two of
    // them bring the noise double buffer back to where it was and are what gets captured.
    // Host-side effects of a launch (which kernel, window plan, instance upload, lazy allocations)
    // happen in the direct iteration that precedes any capture.
    bool have_noise = p->primed;
    int k = 0;
    if (p->inst_set && p->inst_dirty) {  // batched handle: new start states -> window origins, upload
      size_t unused = 0;
      DevParams plan = d;
      (void)plan_lds_window(p, plan, &unused);
      TRY(upload_instances(p));
    }
    // (a rollout kernel that computes its own noise has nothing to prime: its iterations are alike from the start)
    if (((!have_noise && !scan_generates_noise(p)) || !p->graph_warm) && k < iterations) {
      TRY(launch_iteration(p, d, have_noise, true, false));
      p->graph_warm = true;
      ++k;
    }
    const int chunk = p->graph_chunk;  // iterations per graph: even (noise double buffer)
    while (iterations - k >= chunk) {
      std::vector<unsigned char> sig;
      graph_signature(p, d, lin, ang, sig);
      sig.push_back(have_noise ? 1 : 0);
      const int slot = (p->noise_cur & 1) | ((p->u_parity & 1) << 1) | ((p->tpk_cur & 1) << 2);
      if (!p->graph_exec[slot] || sig != p->graph_sig[slot]) {
        if (p->graph_exec[slot]) { (void)hipGraphExecDestroy(p->graph_exec[slot]); p->graph_exec[slot] = nullptr; }
        if (p->graph[slot]) { (void)hipGraphDestroy(p->graph[slot]); p->graph[slot] = nullptr; }
        p->graph_sig[slot].clear();
        const bool primed_before = have_noise;
        const uint64_t spec_before = p->spec_launches;
        const int u_parity_before = p->u_parity, tpk_before = p->tpk_cur;
        HIP_TRY(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
        int rc = MPPI_OK;
        // (the last iteration of a graph applies its update itself: a graph starts and ends with nothing pending)
        for (int j = 0; j < chunk && rc == MPPI_OK; ++j)
          rc = launch_iteration(p, d, have_noise, true, false, false, j + 1 < chunk);
        hipError_t end = hipStreamEndCapture(p->stream, &p->graph[slot]);
        if (rc != MPPI_OK) return rc;
        HIP_TRY(end);
        REQUIRE(have_noise == primed_before, MPPI_ERR_STATE, "graph capture: the iterations are not alike");
        HIP_TRY(hipGraphInstantiate(&p->graph_exec[slot], p->graph[slot], nullptr, nullptr, 0));
        p->graph_sig[slot] = sig;
        p->graph_spec_launches[slot] = p->spec_launches - spec_before;
        // (one GPU, updates applied inside rollout launches: all but the last iteration of the graph change control
        //  buffers -- an odd number when the graph holds an even number of iterations)
        p->graph_u_flip[slot] = (p->u_parity ^ u_parity_before) & 1;
        p->graph_tpk_flip[slot] = (p->tpk_cur ^ tpk_before) & 1;
        ++p->graph_captures;
        // (capturing ran the host side of two iterations; the launch below runs their device side)
      } else {
        // the host-side counters a direct launch of the two iterations would have advanced
        if (p->cfg.rng == MPPI_RNG_PHILOX) p->noise_epoch += (uint64_t)chunk;
        p->bumps_launched += (uint64_t)chunk;
        // (the replayed kernels count their failed tiles like the captured ones did)
        p->spec_launches += p->graph_spec_launches[slot];
        if (p->graph_u_flip[slot]) {
          std::swap(p->u, p->u_alt);
          p->u_parity ^= 1;
        }
        p->tpk_cur ^= p->graph_tpk_flip[slot];
      }
      HIP_TRY(hipGraphLaunch(p->graph_exec[slot], p->stream));
      ++p->graph_replays;
      k += chunk;
    }
    for (; k < iterations; ++k) TRY(launch_iteration(p, d, have_noise, true, false, false, k + 1 < iterations));
    p->primed = have_noise;
  }
  static const bool alt_streams = getenv("MPPI_EXPERIMENT_ALT_STREAMS") != nullptr;
  if (alt_streams) {  // (developer experiment, launch_scan: the second stream joins here)
    HIP_TRY(hipEventRecord(p->ev_noise_ready, p->noise_stream));
    HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_noise_ready, 0));
  }
  if (timed) {
    HIP_TRY(hipEventRecord(p->ev_end, p->stream));
    p->elapsed_pending = true;
    p->last_iterations = iterations;
  }
  return MPPI_OK;
}

static int finish_timing(mppi_planner* p) {
  if (!p->elapsed_pending) return MPPI_OK;
  HIP_TRY(hipEventSynchronize(p->ev_end));
  HIP_TRY(hipEventElapsedTime(&p->last_elapsed_ms, p->ev_begin, p->ev_end));
  if (p->profile_stages && p->last_iterations > 0) {
    // ev_stage: 0 noise | 1 rollout | 2 update-local | 3 collective | 4 apply .. ev_end
    float noise, roll, upd, coll, tail;
    HIP_TRY(hipEventElapsedTime(&noise, p->ev_stage[0], p->ev_stage[1]));
    HIP_TRY(hipEventElapsedTime(&roll, p->ev_stage[1], p->ev_stage[2]));
    HIP_TRY(hipEventElapsedTime(&upd, p->ev_stage[2], p->ev_stage[3]));
    HIP_TRY(hipEventElapsedTime(&coll, p->ev_stage[3], p->ev_stage[4]));
    HIP_TRY(hipEventElapsedTime(&tail, p->ev_stage[4], p->ev_stage[5]));
    p->stage_ms[0] = noise;
    p->stage_ms[1] = roll;
    p->stage_ms[2] = upd + tail;
    p->stage_ms[3] = coll;
  }
  p->elapsed_pending = false;
  return MPPI_OK;
}

// grids are sampled once per solve(), not per optimisation iteration
// (mppi.py:247-248, 321-322, 391-394); the deterministic modes pass alpha_dyn = 1
static int sample_for_solve(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang) {
  if (p->cfg.mode == MPPI_MODE_BAREBONE) return MPPI_OK;
  TraceRange tr("mppi:sample_grids");
  double alpha = (p->cfg.mode == MPPI_MODE_TDM) ? p->params.alpha_dyn : 1.0;
  int rc = MPPI_OK;
  if (sample_into_cells(p, lin, ang, alpha, &rc)) return rc;
  TRY(tdm_sample_on(lin, alpha, p->stream));
  TRY(tdm_sample_on(ang, alpha, p->stream));
  return MPPI_OK;
}
