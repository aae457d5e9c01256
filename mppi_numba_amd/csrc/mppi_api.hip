// mppi_api.hip -- host side of libmppi_hip.so: handles, memory, launches, RCCL.
// C ABI declared in include/mppi_hip.h.  Built for gfx950 only:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
#include "../../include/mppi_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>

#include "rng_kernels.h"
#include "rollout_kernels.h"
#include "rollout_scan_kernel.h"
#include "rollout_scan_exact_kernel.h"
#include "map_kernels.h"
#include "update_kernels.h"
#include "world_kernels.h"

using namespace mppi;

#include <chrono>
#include "host_common.h"
#include "handles.h"

extern "C" const char* mppi_last_error(void) { return g_last_error.c_str(); }
extern "C" int mppi_abi_version(void) { return MPPI_HIP_ABI_VERSION; }

// ---------------------------------------------------------------------------
// device query (config.py:9-12)
// ---------------------------------------------------------------------------
extern "C" int mppi_device_count(int* count) {
  REQUIRE(count, MPPI_ERR_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(MPPI_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return MPPI_OK;
}

extern "C" int mppi_device_props_get(int device, mppi_device_props* out) {
  REQUIRE(out, MPPI_ERR_INVALID, "out is NULL");
  int n = 0;
  TRY(mppi_device_count(&n));
  REQUIRE(device >= 0 && device < n, MPPI_ERR_NO_DEVICE, "device %d not present (%d devices)", device, n);
  hipDeviceProp_t pr;
  HIP_TRY(hipGetDeviceProperties(&pr, device));
  memset(out, 0, sizeof(*out));
  out->max_threads_per_block = pr.maxThreadsPerBlock;
  out->max_block_dim_x = pr.maxThreadsDim[0];
  out->max_grid_dim_x = pr.maxGridSize[0];
  out->wavefront_size = pr.warpSize;
  out->compute_units = pr.multiProcessorCount;
  out->lds_bytes_per_cu = (int)pr.maxSharedMemoryPerMultiProcessor;
  snprintf(out->gcn_arch, sizeof(out->gcn_arch), "%s", pr.gcnArchName);
  snprintf(out->name, sizeof(out->name), "%s", pr.name);
  return MPPI_OK;
}


// live planners of this process: a planner remembers which TDMs its cell words were packed from (packed_lin / packed_ang,
// dereferenced again by scan_plan); a TDM that goes away takes those notes with it
static std::vector<mppi_planner*> g_planners;
static std::mutex g_planners_mutex;  // (handles are single-threaded each; the registry is shared)

extern "C" int mppi_tdm_destroy(mppi_tdm* t) {
  if (!t) return MPPI_OK;
  {
    std::lock_guard<std::mutex> lock(g_planners_mutex);
    for (mppi_planner* p : g_planners) {
      if (p->packed_lin == t || p->packed_ang == t) {
        p->packed_lin = p->packed_ang = nullptr;
        p->packed_lin_grid = p->packed_ang_grid = p->packed_lin_maps = ~0ULL;
      }
    }
  }
  (void)hipSetDevice(t->cfg.device);
  dev_free(t->grid);
  dev_free(t->pmf);
  dev_free(t->table);
  dev_free(t->obs);
  dev_free(t->unk);
  dev_free(t->risk);
  dev_free(t->raw);
  dev_free(t->bin_values);
  dev_free(t->prep_flags);
  dev_free(t->states);
  if (t->stream) (void)hipStreamDestroy(t->stream);
  delete t;
  return MPPI_OK;
}

extern "C" int mppi_tdm_create(const mppi_tdm_cfg* cfg, mppi_tdm** out) {
  REQUIRE(cfg && out, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(cfg->num_grids >= 1 && cfg->max_rows >= 1 && cfg->max_cols >= 1, MPPI_ERR_INVALID,
          "bad TDM dimensions (grids=%d rows=%d cols=%d)", cfg->num_grids, cfg->max_rows, cfg->max_cols);
  REQUIRE(cfg->thread_dim_x >= 1 && cfg->thread_dim_y >= 1, MPPI_ERR_INVALID, "bad thread_dim");
  REQUIRE(cfg->rng == MPPI_RNG_PHILOX || cfg->rng == MPPI_RNG_XOROSHIRO, MPPI_ERR_INVALID, "bad rng kind");
  mppi_device_props pr;
  TRY(mppi_device_props_get(cfg->device, &pr));
  HIP_TRY(hipSetDevice(cfg->device));
  mppi_tdm* t = new mppi_tdm();
  t->cfg = *cfg;
  int rc = MPPI_OK;
  do {
    if (hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) {
      rc = fail(MPPI_ERR_HIP, "hipStreamCreate failed");
      break;
    }
    size_t cells = (size_t)cfg->num_grids * cfg->max_rows * cfg->max_cols;
    if ((rc = dev_alloc(&t->grid, cells)) != MPPI_OK) break;
    // the reference leaves the batch uninitialised (terrain.py:168); zero is friendlier
    if (hipMemsetAsync(t->grid, 0, cells, t->stream) != hipSuccess) {
      rc = fail(MPPI_ERR_HIP, "memset failed");
      break;
    }
    if (cfg->rng == MPPI_RNG_XOROSHIRO) {
      t->n_states = (long)cfg->num_grids * cfg->thread_dim_x * cfg->thread_dim_y;
      std::vector<uint64_t> host(2 * (size_t)t->n_states);
      xoroshiro_init_host(host.data(), t->n_states, cfg->seed);
      if ((rc = dev_alloc(&t->states, host.size())) != MPPI_OK) break;
      if (hipMemcpy(t->states, host.data(), host.size() * sizeof(uint64_t), hipMemcpyHostToDevice) !=
          hipSuccess) {
        rc = fail(MPPI_ERR_HIP, "state upload failed");
        break;
      }
    }
    if (hipStreamSynchronize(t->stream) != hipSuccess) {
      rc = fail(MPPI_ERR_HIP, "sync failed");
      break;
    }
  } while (0);
  if (rc != MPPI_OK) {
    std::string keep = g_last_error;
    mppi_tdm_destroy(t);
    g_last_error = keep;
    return rc;
  }
  *out = t;
  return MPPI_OK;
}

static int tdm_reserve(mppi_tdm* t, int bins, size_t plane);

extern "C" int mppi_tdm_set_maps(mppi_tdm* t, const int8_t* pmf, int bins, int rows, int cols,
                                 const int8_t* bin_to_int8, double traction_lo, double traction_ratio,
                                 const int8_t* obstacle, const int8_t* unknown, const int8_t* risk) {
  REQUIRE(t && pmf && bin_to_int8 && obstacle && unknown, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(bins >= 1 && rows >= 1 && cols >= 1, MPPI_ERR_INVALID, "bad map dims");
  REQUIRE(rows <= t->cfg.max_rows && cols <= t->cfg.max_cols, MPPI_ERR_INVALID,
          "padded map %dx%d exceeds max_map_dim %dx%d", rows, cols, t->cfg.max_rows, t->cfg.max_cols);
  HIP_TRY(hipSetDevice(t->cfg.device));
  size_t plane = (size_t)rows * cols, vol = plane * bins;
  TRY(tdm_reserve(t, bins, plane));
  HIP_TRY(hipMemcpyAsync(t->pmf, pmf, vol, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipMemcpyAsync(t->table, bin_to_int8, (size_t)bins, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipMemcpyAsync(t->obs, obstacle, plane, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipMemcpyAsync(t->unk, unknown, plane, hipMemcpyHostToDevice, t->stream));
  if (risk) HIP_TRY(hipMemcpyAsync(t->risk, risk, plane, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  // a PMF with all mass in one bin per cell samples to the same grid every time
  bool one_hot = true;
  for (size_t c = 0; c < plane && one_hot; ++c) {
    int hundred = 0, other = 0;
    for (int b = 0; b < bins; ++b) {
      int8_t v = pmf[(size_t)b * plane + c];
      if (v == 100) ++hundred;
      else if (v != 0) ++other;
    }
    one_hot = (hundred == 1 && other == 0);
  }
  bool compact = true;
  for (int b = 0; b < bins && compact; ++b) compact = bin_to_int8[b] >= 0;
  for (size_t c = 0; c < plane && compact; ++c)
    compact = (obstacle[c] == 0 || obstacle[c] == 1) && (unknown[c] == 0 || unknown[c] == 1);
  t->compact_ok = compact;
  t->table_max = -128;
  for (int b = 0; b < bins; ++b) t->table_max = std::max(t->table_max, (int)bin_to_int8[b]);
  // border cells whose whole mass sits in a bin of zero traction: every sample of them is a sink
  t->maps_sink_ring = count_sink_rings(rows, cols, [&](int r, int c) {
    const size_t cell = (size_t)r * cols + c;
    for (int b = 0; b < bins; ++b)
      if (pmf[(size_t)b * plane + cell] == 100) return std::fma(traction_ratio, (double)bin_to_int8[b], traction_lo) == 0.0;
    return false;
  });
  t->one_hot = one_hot;
  t->bins = bins;
  t->rows = rows;
  t->cols = cols;
  t->has_risk = risk != nullptr;
  t->lo = traction_lo;
  t->ratio = traction_ratio;
  t->maps_set = true;
  t->injected_sink_ring = 0;  // (judged with the old traction bounds)
  ++t->maps_version;
  return MPPI_OK;
}

// (re)size the per-map device buffers
static int tdm_reserve(mppi_tdm* t, int bins, size_t plane) {
  size_t vol = plane * (size_t)bins;
  if (vol > t->pmf_capacity) {
    dev_free(t->pmf);
    t->pmf_capacity = 0;  // (stays 0 if the allocation below fails)
    TRY(dev_alloc(&t->pmf, vol));
    t->pmf_capacity = vol;
  }
  if (bins > t->table_capacity) {
    dev_free(t->table);
    t->table_capacity = 0;  // (stays 0 if the allocation below fails)
    TRY(dev_alloc(&t->table, (size_t)bins));
    t->table_capacity = bins;
  }
  if (plane > t->map_capacity) {
    dev_free(t->obs);
    dev_free(t->unk);
    dev_free(t->risk);
    t->map_capacity = 0;  // (stays 0 if an allocation below fails)
    TRY(dev_alloc(&t->obs, plane));
    TRY(dev_alloc(&t->unk, plane));
    TRY(dev_alloc(&t->risk, plane));
    t->map_capacity = plane;
  }
  return MPPI_OK;
}

// terrain.py:408-495 + 511-583 on the device: raw PMF (and masks) in, the padded maps the
// planner mode needs out.  See map_kernels.h.
extern "C" int mppi_tdm_set_maps_from_pmf(mppi_tdm* t, int kind, const int8_t* pmf, int bins, int src_rows,
                                          int src_cols, int valid_rows, int valid_cols, int pad_cells,
                                          const float* bin_values, const float bounds[2], double alpha,
                                          const int8_t* bin_to_int8, double traction_lo, double traction_ratio,
                                          const int8_t* obstacle, const int8_t* unknown, int* bad_columns) {
  REQUIRE(t && pmf && bin_values && bounds && bin_to_int8, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(kind >= PREP_TDM && kind <= PREP_SPEED, MPPI_ERR_INVALID, "bad preprocessing kind %d", kind);
  REQUIRE(bins >= 1 && src_rows >= 1 && src_cols >= 1, MPPI_ERR_INVALID, "bad map dims");
  REQUIRE(valid_rows >= 1 && valid_rows <= src_rows && valid_cols >= 1 && valid_cols <= src_cols && pad_cells >= 0,
          MPPI_ERR_INVALID, "bad crop %dx%d of %dx%d (pad %d)", valid_rows, valid_cols, src_rows, src_cols,
          pad_cells);
  REQUIRE(alpha > 0.0 && alpha <= 1.0, MPPI_ERR_INVALID, "alpha must be in (0, 1]");
  const int rows = valid_rows + 2 * pad_cells, cols = valid_cols + 2 * pad_cells;
  REQUIRE(rows <= t->cfg.max_rows && cols <= t->cfg.max_cols, MPPI_ERR_INVALID,
          "padded map %dx%d exceeds max_map_dim %dx%d", rows, cols, t->cfg.max_rows, t->cfg.max_cols);
  HIP_TRY(hipSetDevice(t->cfg.device));
  const size_t plane = (size_t)rows * cols, src_plane = (size_t)src_rows * src_cols;
  TRY(tdm_reserve(t, bins, plane));
  const size_t raw_need = src_plane * ((size_t)bins + 2);
  if (raw_need > t->raw_capacity) {
    dev_free(t->raw);
    t->raw_capacity = 0;  // (stays 0 if the allocation below fails)
    TRY(dev_alloc(&t->raw, raw_need));
    t->raw_capacity = raw_need;
  }
  if (bins > t->bin_values_capacity) {
    dev_free(t->bin_values);
    t->bin_values_capacity = 0;  // (stays 0 if the allocation below fails)
    TRY(dev_alloc(&t->bin_values, (size_t)bins));
    t->bin_values_capacity = bins;
  }
  if (!t->prep_flags) TRY(dev_alloc(&t->prep_flags, (size_t)4));
  int8_t* raw_obs = t->raw + src_plane * bins;
  int8_t* raw_unk = raw_obs + src_plane;
  HIP_TRY(hipMemcpyAsync(t->raw, pmf, src_plane * bins, hipMemcpyHostToDevice, t->stream));
  if (obstacle) HIP_TRY(hipMemcpyAsync(raw_obs, obstacle, src_plane, hipMemcpyHostToDevice, t->stream));
  if (unknown) HIP_TRY(hipMemcpyAsync(raw_unk, unknown, src_plane, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipMemcpyAsync(t->bin_values, bin_values, sizeof(float) * (size_t)bins, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipMemcpyAsync(t->table, bin_to_int8, (size_t)bins, hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipMemsetAsync(t->prep_flags, 0, 4 * sizeof(int), t->stream));
  PrepJob j;
  j.raw_pmf = t->raw;
  j.raw_obs = obstacle ? raw_obs : nullptr;
  j.raw_unk = unknown ? raw_unk : nullptr;
  j.bin_values = t->bin_values;
  j.bins = bins; j.src_rows = src_rows; j.src_cols = src_cols;
  j.valid_rows = valid_rows; j.valid_cols = valid_cols; j.pad = pad_cells;
  j.lo = bounds[0];
  j.span = bounds[1] - bounds[0];  // float32 subtraction, as numpy does on the float32 bounds
  j.alpha = alpha;
  j.kind = kind;
  j.pmf = t->pmf; j.obs = t->obs; j.unk = t->unk; j.risk = t->risk;
  j.flags = t->prep_flags;
  hipLaunchKernelGGL(k_prepare_maps, dim3(ceil_div((long)plane, 256)), dim3(256), 0, t->stream, j);
  HIP_TRY(hipGetLastError());
  int flags[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(flags, t->prep_flags, sizeof(flags), hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  if (bad_columns) *bad_columns = flags[0];
  bool compact = flags[2] == 0;
  t->table_max = -128;
  for (int b = 0; b < bins; ++b) {
    compact = compact && bin_to_int8[b] >= 0;
    t->table_max = std::max(t->table_max, (int)bin_to_int8[b]);
  }
  t->compact_ok = compact;
  // (the ring k_prepare_maps writes: all mass in bin 0)
  t->maps_sink_ring = std::fma(traction_ratio, (double)bin_to_int8[0], traction_lo) == 0.0
                          ? std::min(pad_cells, (int)mppi_tdm::kSinkRingCap) : 0;
  t->one_hot = (kind != PREP_TDM) || flags[1] == 0;
  t->bins = bins;
  t->rows = rows;
  t->cols = cols;
  t->has_risk = kind == PREP_SPEED;
  t->lo = traction_lo;
  t->ratio = traction_ratio;
  t->maps_set = true;
  t->injected_sink_ring = 0;  // (judged with the old traction bounds)
  ++t->maps_version;
  return MPPI_OK;
}

// the maps as they are on the device (any pointer may be NULL): pmf (bins, rows, cols),
// obstacle / unknown / risk (rows, cols)
extern "C" int mppi_tdm_get_maps(mppi_tdm* t, int8_t* pmf, int8_t* obstacle, int8_t* unknown, int8_t* risk) {
  REQUIRE(t, MPPI_ERR_INVALID, "NULL tdm");
  REQUIRE(t->maps_set, MPPI_ERR_STATE, "TDM maps not set");
  HIP_TRY(hipSetDevice(t->cfg.device));
  const size_t plane = (size_t)t->rows * t->cols;
  if (pmf) HIP_TRY(hipMemcpyAsync(pmf, t->pmf, plane * t->bins, hipMemcpyDeviceToHost, t->stream));
  if (obstacle) HIP_TRY(hipMemcpyAsync(obstacle, t->obs, plane, hipMemcpyDeviceToHost, t->stream));
  if (unknown) HIP_TRY(hipMemcpyAsync(unknown, t->unk, plane, hipMemcpyDeviceToHost, t->stream));
  if (risk) {
    REQUIRE(t->has_risk, MPPI_ERR_STATE, "this TDM holds no risk traction map");
    HIP_TRY(hipMemcpyAsync(risk, t->risk, plane, hipMemcpyDeviceToHost, t->stream));
  }
  HIP_TRY(hipStreamSynchronize(t->stream));
  return MPPI_OK;
}

// the Philox draws of (epoch, alpha_dyn) into the (G, R, C) int8 grids
static int tdm_launch_philox(mppi_tdm* t, double alpha_dyn, uint64_t epoch, hipStream_t stream) {
  const int G = t->cfg.num_grids;
  {
    const long cell_groups = (long)t->rows * ((t->cols + 3) / 4);
    if (t->bins <= 64) {
      // enough workgroups to fill the chip, as many samples per thread as that allows
      int chunks = ceil_div(256L * 1024, cell_groups);
      chunks = chunks < 1 ? 1 : (chunks > G ? G : chunks);
      const int g_chunk = (ceil_div(G, chunks) + 1) & ~1;  // even: a Philox block serves a pair of samples
      dim3 grid((unsigned)ceil_div(cell_groups, 256), (unsigned)ceil_div(G, g_chunk));
#define MPPI_SAMPLE(MAXB)                                                                                       \
  hipLaunchKernelGGL(k_sample_grids_philox_cols<MAXB>, grid, dim3(256), 0, stream, t->pmf, t->bins, t->rows, t->cols, \
                     t->table, alpha_dyn, t->cfg.seed, epoch, G, g_chunk, t->grid, t->cfg.max_rows, t->cfg.max_cols, \
                     (uint64_t)(t->first_sample >> 1) * (uint64_t)cell_groups)
      if (t->bins <= 8) MPPI_SAMPLE(8);
      else if (t->bins <= 16) MPPI_SAMPLE(16);
      else if (t->bins <= 32) MPPI_SAMPLE(32);
      else MPPI_SAMPLE(64);
#undef MPPI_SAMPLE
    } else {
      long total = (long)G * cell_groups;
      hipLaunchKernelGGL(k_sample_grids_philox, dim3(ceil_div(total, 256)), dim3(256), 0, stream, t->pmf, t->bins,
                         t->rows, t->cols, t->table, alpha_dyn, t->cfg.seed, epoch, G, t->grid,
                         t->cfg.max_rows, t->cfg.max_cols, (uint64_t)t->first_sample * (uint64_t)cell_groups);
    }
  }
  HIP_TRY(hipGetLastError());
  return MPPI_OK;
}

// the int8 grids, if the current draws only exist as a planner's cell words so far
static int tdm_materialize(mppi_tdm* t, hipStream_t stream) {
  if (!t->grid_stale) return MPPI_OK;
  TRY(tdm_launch_philox(t, t->sampled_alpha, t->sampled_epoch, stream));
  t->grid_stale = false;
  return MPPI_OK;
}

// enqueue the sampling kernel on `stream` (no synchronisation)
static int tdm_sample_on(mppi_tdm* t, double alpha_dyn, hipStream_t stream) {
  REQUIRE(t->maps_set, MPPI_ERR_STATE, "TDM maps not set");
  if (t->one_hot && t->sampled_maps_version == t->maps_version && alpha_dyn > 0.0) return MPPI_OK;
  const int G = t->cfg.num_grids;
  if (t->cfg.rng == MPPI_RNG_PHILOX) {
    TRY(tdm_launch_philox(t, alpha_dyn, t->epoch, stream));
    t->sampled_epoch = t->epoch;
    t->grid_stale = false;
    ++t->epoch;
  } else {
    int threads = G * t->cfg.thread_dim_x * t->cfg.thread_dim_y;
    hipLaunchKernelGGL(k_sample_grids_xoroshiro, dim3(ceil_div(threads, 64)), dim3(64), 0, stream, t->pmf,
                       t->bins, t->rows, t->cols, t->table, alpha_dyn, t->states, G, t->cfg.thread_dim_x,
                       t->cfg.thread_dim_y, t->grid, t->cfg.max_rows, t->cfg.max_cols);
  }
  HIP_TRY(hipGetLastError());
  t->sampled_maps_version = t->maps_version;
  t->sampled_alpha = alpha_dyn;
  ++t->grid_version;
  t->injected = false;
  return MPPI_OK;
}

extern "C" int mppi_tdm_set_sample_shard(mppi_tdm* t, int first_sample) {
  REQUIRE(t, MPPI_ERR_INVALID, "NULL tdm");
  REQUIRE(first_sample >= 0 && (first_sample & 1) == 0, MPPI_ERR_INVALID,
          "first_sample %d: must be even and >= 0 (a Philox block serves a pair of samples)", first_sample);
  REQUIRE(first_sample == 0 || t->cfg.rng == MPPI_RNG_PHILOX, MPPI_ERR_INVALID,
          "sample shards need the counter-based generator (MPPI_RNG_PHILOX)");
  if (first_sample != t->first_sample) {
    t->first_sample = first_sample;
    t->sampled_maps_version = ~0ULL;  // (a one-hot PMF is re-sampled too: cheap, and keeps the rule simple)
  }
  return MPPI_OK;
}

extern "C" int mppi_tdm_sample_grids(mppi_tdm* t, double alpha_dyn) {
  REQUIRE(t, MPPI_ERR_INVALID, "NULL tdm");
  HIP_TRY(hipSetDevice(t->cfg.device));
  TRY(tdm_sample_on(t, alpha_dyn, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  return MPPI_OK;
}

extern "C" int mppi_tdm_set_sampled_grids(mppi_tdm* t, const int8_t* grids, int rows, int cols) {
  REQUIRE(t && grids, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(rows >= 1 && cols >= 1 && rows <= t->cfg.max_rows && cols <= t->cfg.max_cols, MPPI_ERR_INVALID,
          "window %dx%d does not fit max_map_dim", rows, cols);
  HIP_TRY(hipSetDevice(t->cfg.device));
  for (int g = 0; g < t->cfg.num_grids; ++g)
    HIP_TRY(hipMemcpy2DAsync(t->grid + (size_t)g * t->cfg.max_rows * t->cfg.max_cols, (size_t)t->cfg.max_cols,
                             grids + (size_t)g * rows * cols, (size_t)cols, (size_t)cols, (size_t)rows,
                             hipMemcpyHostToDevice, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  t->injected_min = 127;
  t->injected_max = -128;
  for (size_t i = 0; i < (size_t)t->cfg.num_grids * rows * cols; ++i) {
    if (grids[i] < t->injected_min) t->injected_min = grids[i];
    if (grids[i] > t->injected_max) t->injected_max = grids[i];
  }
  t->injected_sink_ring = (!t->maps_set || rows != t->rows || cols != t->cols) ? 0 : count_sink_rings(rows, cols, [&](int r, int c) {
    for (int g = 0; g < t->cfg.num_grids; ++g)
      if (std::fma(t->ratio, (double)grids[((size_t)g * rows + r) * cols + c], t->lo) != 0.0) return false;
    return true;
  });
  ++t->grid_version;
  t->sampled_maps_version = ~0ULL;  // injected grids are not a cached sample
  t->grid_stale = false;             // (a lazy sample, if any, is superseded)
  t->injected = true;                // arbitrary bytes: the 16-bit cell format is not guaranteed
  return MPPI_OK;
}

extern "C" int mppi_tdm_get_sampled_grids(mppi_tdm* t, int8_t* out) {
  REQUIRE(t && out, MPPI_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(t->cfg.device));
  size_t bytes = (size_t)t->cfg.num_grids * t->cfg.max_rows * t->cfg.max_cols;
  TRY(tdm_materialize(t, t->stream));
  HIP_TRY(hipMemcpyAsync(out, t->grid, bytes, hipMemcpyDeviceToHost, t->stream));
  HIP_TRY(hipStreamSynchronize(t->stream));
  return MPPI_OK;
}

extern "C" int mppi_tdm_rng_states(mppi_tdm* t, uint64_t* out, long capacity, long* count) {
  REQUIRE(t && count, MPPI_ERR_INVALID, "NULL argument");
  *count = t->n_states;
  if (!out || t->n_states == 0) return MPPI_OK;
  REQUIRE(capacity >= t->n_states, MPPI_ERR_INVALID, "capacity %ld < %ld states", capacity, t->n_states);
  HIP_TRY(hipSetDevice(t->cfg.device));
  HIP_TRY(hipMemcpy(out, t->states, 2 * sizeof(uint64_t) * (size_t)t->n_states, hipMemcpyDeviceToHost));
  return MPPI_OK;
}

extern "C" int mppi_trace_ranges_enabled(void) {
  if (!g_roctx.tried) roctx_load();
  return g_roctx.push ? 1 : 0;
}

extern "C" int mppi_comm_unique_id(char id[MPPI_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= MPPI_COMM_ID_BYTES, "ncclUniqueId larger than expected");
  REQUIRE(id, MPPI_ERR_INVALID, "NULL id");
  TRY(rccl_load());
  ncclUniqueId uid;
  RCCL_TRY(g_rccl.GetUniqueId(&uid));
  memset(id, 0, MPPI_COMM_ID_BYTES);
  memcpy(id, &uid, sizeof(uid));
  return MPPI_OK;
}

extern "C" int mppi_planner_destroy(mppi_planner* p) {
  if (!p) return MPPI_OK;
  {
    std::lock_guard<std::mutex> lock(g_planners_mutex);
    g_planners.erase(std::remove(g_planners.begin(), g_planners.end(), p), g_planners.end());
  }
  (void)hipSetDevice(p->cfg.device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(p->comm);
  for (int g = 0; g < kMaxFoldedRanks; ++g)
    if (p->peer_mapped[g] && p->peer_inbox[g]) (void)hipIpcCloseMemHandle(p->peer_inbox[g]);
  if (p->inbox) (void)hipFree(p->inbox);
  if (p->p2p_fault_host) (void)hipHostFree(p->p2p_fault_host);
  if (p->fold_fault_host) (void)hipHostFree(p->fold_fault_host);
  dev_free(p->inst_dev);
  if (p->u_host) (void)hipHostFree(p->u_host);
  if (p->u_stage) (void)hipHostFree(p->u_stage);
  if (p->ev_u_staged) (void)hipEventDestroy(p->ev_u_staged);
  dev_free(p->noise_buf[0]);
  dev_free(p->noise_buf[1]);
  dev_free(p->staging);
  dev_free(p->u);
  dev_free(p->u_prev);
  dev_free(p->u_alt);
  dev_free(p->costs);
  dev_free(p->weights_out);
  dev_free(p->w_rel);
  dev_free(p->tile_beta);
  dev_free(p->tile_packets[0]);
  dev_free(p->tile_packets[1]);
  dev_free(p->published);
  dev_free(p->packets);
  dev_free(p->stats);
  dev_free(p->cells);
  dev_free(p->cells16);
  dev_free(p->cc_scratch);
  dev_free(p->sample_costs);
  dev_free(p->states);
  dev_free(p->obs_pos);
  dev_free(p->obs_r);
  dev_free(p->state_rollout);
  dev_free(p->slabs);
  for (hipEvent_t e : p->ktime_events)
    if (e) (void)hipEventDestroy(e);
  dev_free(p->ktime_dev);
  dev_free(p->noise_flag_dev);
  dev_free(p->progress_dev);
  if (p->flag_fault_host) (void)hipHostFree(p->flag_fault_host);
  if (p->spec_fail_host) (void)hipHostFree(p->spec_fail_host);
  dev_free(p->loop_state);
  dev_free(p->loop_xhist);
  dev_free(p->loop_uhist);
  dev_free(p->loop_done);
  dev_free(p->loop_u_final);
  if (p->loop_done_count) (void)hipHostFree(p->loop_done_count);
  if (p->ev_begin) (void)hipEventDestroy(p->ev_begin);
  if (p->ev_end) (void)hipEventDestroy(p->ev_end);
  for (auto& e : p->ev_stage)
    if (e) (void)hipEventDestroy(e);
  drop_graphs(p);
  dev_free(p->gen_dev);
  if (p->noise_stream) {
    (void)hipStreamSynchronize(p->noise_stream);
    (void)hipStreamDestroy(p->noise_stream);
  }
  if (p->ev_buf_free) (void)hipEventDestroy(p->ev_buf_free);
  if (p->ev_noise_ready) (void)hipEventDestroy(p->ev_noise_ready);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
  return MPPI_OK;
}

static int planner_alloc(mppi_planner* p) {
  if (!p->spec_fail_host) {
    HIP_TRY(hipHostMalloc((void**)&p->spec_fail_host, sizeof(unsigned int), hipHostMallocMapped));
    *p->spec_fail_host = 0u;
    HIP_TRY(hipHostGetDevicePointer((void**)&p->spec_fail_dev, p->spec_fail_host, 0));
  }
  const mppi_planner_cfg& c = p->cfg;
  const size_t N = (size_t)p->n_local, T = (size_t)c.num_steps;
  HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&p->noise_stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&p->ev_buf_free, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&p->ev_noise_ready, hipEventDisableTiming));
  TRY(dev_alloc(&p->noise_flag_dev, (size_t)1));
  HIP_TRY(hipMemset(p->noise_flag_dev, 0, sizeof(unsigned long long)));
  TRY(dev_alloc(&p->progress_dev, (size_t)1));
  HIP_TRY(hipMemset(p->progress_dev, 0, sizeof(unsigned long long)));
  HIP_TRY(hipHostMalloc((void**)&p->flag_fault_host, sizeof(unsigned int), hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer((void**)&p->flag_fault_dev, p->flag_fault_host, 0));
  *p->flag_fault_host = 0u;
  HIP_TRY(hipEventCreate(&p->ev_begin));
  HIP_TRY(hipEventCreate(&p->ev_end));
  for (auto& e : p->ev_stage) HIP_TRY(hipEventCreate(&e));
  const size_t n_tiled = (size_t)ceil_div((long)N, 64) * 64;  // tile-major arrays cover whole tiles
  // (+ 8 chunks of 8 rows of slack behind the last tile: batch loads clamp rows, not tiles)
  const size_t noise_pad = 8 * 16 * 64;
  for (int b = 0; b < 2; ++b) {
    TRY(dev_alloc(&p->noise_buf[b], n_tiled * T + noise_pad));
    HIP_TRY(hipMemsetAsync(p->noise_buf[b], 0, (n_tiled * T + noise_pad) * sizeof(float2), p->stream));
  }
  p->noise = p->noise_buf[0];
  TRY(dev_alloc(&p->staging, N * T));
  const size_t B = (size_t)p->B;
  TRY(dev_alloc(&p->u, B * T));
  TRY(dev_alloc(&p->u_prev, B * T));
  TRY(dev_alloc(&p->u_alt, B * T));
  TRY(dev_alloc(&p->inst_dev, B));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->u_host), B * T * sizeof(float2), hipHostMallocMapped));
  HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&p->u_host_dev), p->u_host, 0));
  memset(p->u_host, 0, B * T * sizeof(float2));
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->u_stage), B * T * sizeof(float2), hipHostMallocDefault));
  HIP_TRY(hipEventCreateWithFlags(&p->ev_u_staged, hipEventDisableTiming));
  p->inst_host.assign(B, BatchInst{});
  TRY(dev_alloc(&p->costs, N));
  TRY(dev_alloc(&p->weights_out, N));
  p->n_tiles = ceil_div((long)N, 64);
  TRY(dev_alloc(&p->w_rel, N));
  TRY(dev_alloc(&p->tile_beta, (size_t)p->n_tiles));
  TRY(dev_alloc(&p->packets, (size_t)c.world_size * B * packet_len((int)T)));
  if (c.mode == MPPI_MODE_DET || c.mode == MPPI_MODE_SPEED_MAP) {  // the time-parallel kernels' tile packets (handles.h)
    const size_t tiles = (size_t)ceil_div((long)N, 32);
    for (int b = 0; b < 2; ++b) {
      TRY(dev_alloc(&p->tile_packets[b], tiles * (size_t)tile_packet_floats((int)T)));
      HIP_TRY(hipMemsetAsync(p->tile_packets[b], 0, tiles * (size_t)tile_packet_floats((int)T) * sizeof(float), p->stream));
    }
    TRY(dev_alloc(&p->published, published_words((int)T)));
    HIP_TRY(hipMemsetAsync(p->published, 0xff, sizeof(unsigned long long) * published_words((int)T), p->stream));  // (kNotPublished)
    HIP_TRY(hipHostMalloc((void**)&p->fold_fault_host, sizeof(unsigned int), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&p->fold_fault_dev, p->fold_fault_host, 0));
    *p->fold_fault_host = 0u;
  }
  TRY(dev_alloc(&p->stats, 2 * B));
  TRY(dev_alloc(&p->state_rollout, (size_t)c.num_vis_state_rollouts * (T + 1) * 3));
  HIP_TRY(hipMemsetAsync(p->u, 0, B * T * sizeof(float2), p->stream));  // u_seq0 = zeros (mppi.py:93)
  HIP_TRY(hipMemsetAsync(p->u_prev, 0, B * T * sizeof(float2), p->stream));
  HIP_TRY(hipMemsetAsync(p->u_alt, 0, B * T * sizeof(float2), p->stream));
  HIP_TRY(hipMemsetAsync(p->costs, 0, N * sizeof(float), p->stream));
  {
    std::vector<double> initial_stats(2 * B);
    for (size_t b = 0; b < B; ++b) { initial_stats[2 * b] = 0.0; initial_stats[2 * b + 1] = 1.0; }
    HIP_TRY(hipMemcpy(p->stats, initial_stats.data(), sizeof(double) * 2 * B, hipMemcpyHostToDevice));
  }
  if (c.rng == MPPI_RNG_XOROSHIRO) {
    // numba creates N*T states on the host, 2^64-jump apart (mppi.py:118); a
    // shard keeps the slice of the global stream array that it owns
    long total = (long)c.num_control_rollouts * (long)B * c.num_steps;  // rank-major over (rank, problem, rollout)
    std::vector<uint64_t> host(2 * (size_t)total);
    xoroshiro_init_host(host.data(), total, c.seed);
    p->n_states = (long)N * (long)T;
    TRY(dev_alloc(&p->states, 2 * (size_t)p->n_states));
    HIP_TRY(hipMemcpy(p->states, host.data() + 2 * (size_t)p->n_offset * T,
                      2 * sizeof(uint64_t) * (size_t)p->n_states, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  return MPPI_OK;
}

extern "C" int mppi_planner_create(const mppi_planner_cfg* cfg, mppi_planner** out) {
  REQUIRE(cfg && out, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(cfg->mode >= MPPI_MODE_DET && cfg->mode <= MPPI_MODE_BAREBONE, MPPI_ERR_INVALID, "bad mode %d",
          cfg->mode);
  REQUIRE(cfg->num_control_rollouts >= 1 && cfg->num_steps >= 1, MPPI_ERR_INVALID, "bad N=%d or T=%d",
          cfg->num_control_rollouts, cfg->num_steps);
  REQUIRE(cfg->num_grid_samples >= 1 && cfg->num_vis_state_rollouts >= 1, MPPI_ERR_INVALID, "bad M or V");
  REQUIRE(cfg->mode == MPPI_MODE_TDM || cfg->num_grid_samples == 1, MPPI_ERR_INVALID,
          "num_grid_samples must be 1 unless MPPI_MODE_TDM");
  REQUIRE(cfg->world_size >= 1 && cfg->rank >= 0 && cfg->rank < cfg->world_size, MPPI_ERR_INVALID,
          "bad rank %d / world %d", cfg->rank, cfg->world_size);
  REQUIRE(cfg->num_control_rollouts % cfg->world_size == 0, MPPI_ERR_INVALID,
          "num_control_rollouts (%d) must be a multiple of world_size (%d)", cfg->num_control_rollouts,
          cfg->world_size);
  REQUIRE(cfg->rng == MPPI_RNG_PHILOX || cfg->rng == MPPI_RNG_XOROSHIRO, MPPI_ERR_INVALID, "bad rng kind");
  REQUIRE(cfg->math == MPPI_MATH_EXACT || cfg->math == MPPI_MATH_FAST, MPPI_ERR_INVALID, "bad math kind");
  mppi_device_props pr;
  TRY(mppi_device_props_get(cfg->device, &pr));
  REQUIRE(strncmp(pr.gcn_arch, "gfx950", 6) == 0, MPPI_ERR_NO_DEVICE,
          "device %d is %s; this library is built for gfx950 (MI355X) only", cfg->device, pr.gcn_arch);
  HIP_TRY(hipSetDevice(cfg->device));
  const int n_problems = cfg->num_instances > 1 ? cfg->num_instances : 1;
  const int n_inst = cfg->num_control_rollouts / cfg->world_size;
  REQUIRE(cfg->num_instances >= 0 && n_problems <= 65535, MPPI_ERR_INVALID, "bad num_instances %d",
          cfg->num_instances);
  REQUIRE(n_problems == 1 || (n_inst % 64 == 0 && cfg->mode != MPPI_MODE_BAREBONE), MPPI_ERR_INVALID,
          "num_instances > 1 needs num_control_rollouts/world_size (%d) to be a multiple of 64 and a map mode",
          n_inst);
  REQUIRE((long)n_problems * n_inst <= (1L << 30), MPPI_ERR_INVALID, "too many rollouts per GPU");
  const int n_local = n_problems * n_inst;
  const int device_cus = pr.compute_units, device_lds = pr.lds_bytes_per_cu;
  REQUIRE(cfg->num_vis_state_rollouts <= n_inst || cfg->mode == MPPI_MODE_TDM, MPPI_ERR_INVALID,
          "num_vis_state_rollouts exceeds local rollouts");
  REQUIRE(cfg->mode != MPPI_MODE_TDM || cfg->num_vis_state_rollouts <= cfg->num_grid_samples, MPPI_ERR_INVALID,
          "num_vis_state_rollouts exceeds num_grid_samples");
  mppi_planner* p = new mppi_planner();
  p->cfg = *cfg;
  p->n_local = n_local;
  p->B = n_problems;
  p->n_inst = n_inst;
  p->inst_tiles = ceil_div(n_inst, 64);
  p->n_offset = cfg->rank * p->n_local;
  p->num_cus = device_cus > 0 ? device_cus : 256;
  p->lds_per_cu = device_lds >= 64 * 1024 ? device_lds : 64 * 1024;
  memset(&p->params, 0, sizeof(p->params));
  int rc = planner_alloc(p);
  if (rc != MPPI_OK) {
    std::string keep = g_last_error;
    mppi_planner_destroy(p);
    g_last_error = keep;
    return rc;
  }
  {
    std::lock_guard<std::mutex> lock(g_planners_mutex);
    g_planners.push_back(p);
  }
  *out = p;
  return MPPI_OK;
}

// Graph mode keeps the noise of the next iteration ready.  Dropping it gives its Philox epoch
// back, so that the sequence of noise blocks the iterations consume stays the one of the direct
// loop (the xoroshiro-compatible generator cannot rewind: it simply moves on).
static void discard_noise_ahead(mppi_planner* p) {
  if (p->primed && p->cfg.rng == MPPI_RNG_PHILOX) --p->noise_epoch;
  p->primed = false;
  // a generator still in flight on the second stream must not overlap the next one on the main
  // stream (they share the xoroshiro states)
  if (p->noise_on_side_stream) (void)hipStreamWaitEvent(p->stream, p->ev_noise_ready, 0);
  p->noise_on_side_stream = false;
}

extern "C" int mppi_planner_set_params(mppi_planner* p, const mppi_params* params) {
  REQUIRE(p && params, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(params->lambda_weight > 0.0f, MPPI_ERR_INVALID, "lambda_weight must be > 0");
  REQUIRE(params->num_opt >= 0, MPPI_ERR_INVALID, "num_opt must be >= 0");
  REQUIRE(p->cfg.mode == MPPI_MODE_BAREBONE || params->res > 0.0f, MPPI_ERR_INVALID, "res must be > 0");
  REQUIRE(params->u_std[0] > 0.0f && params->u_std[1] > 0.0f, MPPI_ERR_INVALID, "u_std must be > 0");
  if (p->params_set && (p->params.u_std[0] != params->u_std[0] || p->params.u_std[1] != params->u_std[1]))
    discard_noise_ahead(p);  // it was scaled with the old standard deviations
  // tile-relative weights emitted by the last rollout's epilogue are exp(-(c - beta_tile)/lambda_old):
  // a stage-level update() after a temperature change must form them again from the costs
  if (p->params_set && p->params.lambda_weight != params->lambda_weight) p->tile_packets_fresh = false;
  p->params = *params;
  p->params_set = true;
  p->inst_dirty = true;  // the per-problem window origins depend on the reach
  return MPPI_OK;
}

extern "C" int mppi_planner_set_disc_obstacles(mppi_planner* p, const float* positions, const float* radii,
                                               int count) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  REQUIRE(count >= 0 && (count == 0 || (positions && radii)), MPPI_ERR_INVALID, "bad obstacle arrays");
  // the barebone mirror hands the obstacles over with every solve() (the notebook uploads them per call,
  // barebone_mppi_numba.ipynb cell 3): unchanged discs cost a comparison, not a synchronisation and two allocations
  if (count == p->n_obstacles && (size_t)count * 2 == p->obs_pos_host.size() &&
      (count == 0 || (memcmp(positions, p->obs_pos_host.data(), sizeof(float) * 2 * (size_t)count) == 0 &&
                      memcmp(radii, p->obs_r_host.data(), sizeof(float) * (size_t)count) == 0)))
    return MPPI_OK;
  HIP_TRY(hipSetDevice(p->cfg.device));
  HIP_TRY(hipStreamSynchronize(p->stream));
  dev_free(p->obs_pos);
  dev_free(p->obs_r);
  p->n_obstacles = 0;
  p->obs_pos_host.clear();
  p->obs_r_host.clear();
  if (count > 0) {
    TRY(dev_alloc(&p->obs_pos, (size_t)count));
    TRY(dev_alloc(&p->obs_r, (size_t)count));
    HIP_TRY(hipMemcpy(p->obs_pos, positions, sizeof(float2) * (size_t)count, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->obs_r, radii, sizeof(float) * (size_t)count, hipMemcpyHostToDevice));
    p->obs_pos_host.assign(positions, positions + 2 * (size_t)count);
    p->obs_r_host.assign(radii, radii + (size_t)count);
  }
  p->n_obstacles = count;
  drop_graphs(p);  // (the count is a by-value argument of the captured launches)
  return MPPI_OK;
}

// Every entry point that waits for the planner's stream does it through this: the wait itself, then the words the
// kernels of the drained launches may have raised -- a peer whose numbers did not arrive (MPPI_ERR_COMM), a rollout
// launch that gave the hand-over of the controls up (MPPI_ERR_BUSY; include/mppi_hip.h: "the next synchronising call").
static int drain_stream(mppi_planner* p);

static int copy_out(mppi_planner* p, void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipSetDevice(p->cfg.device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, p->stream));
  return drain_stream(p);
}

extern "C" int mppi_planner_set_u(mppi_planner* p, const float* u) {
  REQUIRE(p && u, MPPI_ERR_INVALID, "NULL argument");
  // on the control path (shift_and_update): no pageable copy, no stream synchronisation -- the
  // caller's array is consumed before returning, the device copy is ordered on the planner's stream
  HIP_TRY(hipSetDevice(p->cfg.device));
  const size_t bytes = sizeof(float2) * (size_t)p->B * (size_t)p->cfg.num_steps;
  if (p->u_stage_busy) HIP_TRY(hipEventSynchronize(p->ev_u_staged));
  memcpy(p->u_stage, u, bytes);
  HIP_TRY(hipMemcpyAsync(p->u, p->u_stage, bytes, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipEventRecord(p->ev_u_staged, p->stream));
  p->u_stage_busy = true;
  return MPPI_OK;
}
extern "C" int mppi_planner_get_u(mppi_planner* p, float* u) {
  REQUIRE(p && u, MPPI_ERR_INVALID, "NULL argument");
  return copy_out(p, u, p->u, sizeof(float2) * (size_t)p->B * (size_t)p->cfg.num_steps);
}
extern "C" int mppi_planner_get_u_prev(mppi_planner* p, float* u) {
  REQUIRE(p && u, MPPI_ERR_INVALID, "NULL argument");
  return copy_out(p, u, p->u_prev, sizeof(float2) * (size_t)p->B * (size_t)p->cfg.num_steps);
}

extern "C" int mppi_planner_shift_u(mppi_planner* p, int k) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  if (k <= 0 || k >= p->cfg.num_steps) return MPPI_OK;  // u[:-k] = u[k:] is empty then
  HIP_TRY(hipSetDevice(p->cfg.device));
  hipLaunchKernelGGL(k_shift_u, dim3(p->B), dim3(256), sizeof(float2) * (size_t)p->cfg.num_steps, p->stream, p->u,
                     p->cfg.num_steps, k);
  HIP_TRY(hipGetLastError());
  // (control path: no synchronisation -- every consumer of u is ordered behind this on the stream)
  return MPPI_OK;
}

// Batched multi-query: the start state and the goal of every problem (everything else is
// shared through mppi_params).  With count == 1 a single-problem handle takes the same
// kernel path as a batch (the parity tests compare the two).
extern "C" int mppi_planner_set_instances(mppi_planner* p, int count, const float* x0, const float* xgoal) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL argument");
  if (count == 0 && p->B == 1) {  // a single-problem handle goes back to the start / goal of mppi_params
    if (p->inst_set) drop_graphs(p);
    p->inst_set = false;
    p->inst_dirty = false;
    return MPPI_OK;
  }
  REQUIRE(x0 && xgoal, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(count == p->B, MPPI_ERR_INVALID, "count %d != num_instances %d of this handle", count, p->B);
  REQUIRE(p->cfg.mode != MPPI_MODE_BAREBONE, MPPI_ERR_INVALID, "no instances in the barebone mode");
  for (int b = 0; b < count; ++b) {
    BatchInst& I = p->inst_host[(size_t)b];
    I.x0 = x0[3 * b]; I.y0 = x0[3 * b + 1]; I.th0 = x0[3 * b + 2];
    I.xg = xgoal[2 * b]; I.yg = xgoal[2 * b + 1];
    REQUIRE(std::isfinite(I.x0) && std::isfinite(I.y0) && std::isfinite(I.th0) && std::isfinite(I.xg) &&
                std::isfinite(I.yg),
            MPPI_ERR_INVALID, "instance %d: non-finite start or goal", b);
  }
  p->inst_set = true;
  p->inst_dirty = true;
  return MPPI_OK;
}

// largest traction byte that can be in the TDM's grid right now
#include "launch_plan.h"

extern "C" int mppi_planner_iterate_async(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, int iterations) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  REQUIRE(iterations >= 0, MPPI_ERR_INVALID, "iterations < 0");
  HIP_TRY(hipSetDevice(p->cfg.device));
  return run_iterations(p, lin, ang, iterations);
}

// Waiting for the planner's stream where a control loop waits for it (solve, synchronize).  hipStreamSynchronize
// waits actively for a short while and then parks the thread; being woken costs ~4 us (tools/call_overhead.py,
// profiles/r05_call_overhead.txt: a call of 20 iterations 324 -> 315 us when the stream is polled instead).  Polling is
// not free either: after a hipStreamQuery loop the next enqueue of this thread is ~2 us slower, and a wait as short as
// one iteration never reaches the parking phase (tools/control_step_latency.py, control step blocking / polling:
// num_opt = 1 40.6 / 43.7 us, 2: 57.3 / 61.6, 4: 97.1 / 97.3, 8: 162.7 / 161.5).  So: poll when four or more iterations
// have been enqueued since the last wait (MPPI_SYNC_POLL_FROM), block otherwise.  MPPI_SYNC_SPIN_US=0: never poll.
static int wait_for_stream(mppi_planner* p) {
  static const long spin_us = getenv("MPPI_SYNC_SPIN_US") ? atol(getenv("MPPI_SYNC_SPIN_US")) : 400;
  static const long kPollFromIterations = getenv("MPPI_SYNC_POLL_FROM") ? atol(getenv("MPPI_SYNC_POLL_FROM")) : 4;
  const bool long_wait = p->iterations_since_wait >= kPollFromIterations;
  p->iterations_since_wait = 0;
  if (spin_us > 0 && long_wait) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(p->stream);
      if (e == hipSuccess) return MPPI_OK;
      if (e != hipErrorNotReady) HIP_TRY(e);
      (void)hipGetLastError();  // (hipErrorNotReady is sticky in hipGetLastError)
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
    }
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  return MPPI_OK;
}

// after the stream has drained: did a peer fail to deliver (update_kernels.h, exchange_step)?
static int check_peer_fault(mppi_planner* p) {
  if (p->p2p_fault_host && *p->p2p_fault_host != 0u) {
    p->p2p_on = false;  // (the inboxes are in an unknown state: reconnect before using the exchange again)
    return fail(MPPI_ERR_COMM, "peer exchange: another rank's numbers did not arrive within the time limit "
                               "(a rank died or stalled); the control sequence of this call is not valid");
  }
  return MPPI_OK;
}

// after the stream has drained: did a workgroup give up waiting for the controls its siblings publish inside a
// rollout launch (update_kernels.h, collect_published)?  Then not all workgroups of that launch were resident side by
// side -- the device is shared or masked -- and this handle stops folding its updates into rollout launches.
static int check_fold_fault(mppi_planner* p) {
  if (p->fold_fault_host && *p->fold_fault_host != 0u) {
    *p->fold_fault_host = 0u;
    p->fold_off = true;
    ++p->fold_faults;
    drop_graphs(p);  // (captured loops fold)
    p->graph_warm = false;
    // (the words of the loop that broke off: a later k_combine_tiles clears them as well; the handle does not fold again)
    (void)hipMemsetAsync(p->published, 0xff, sizeof(unsigned long long) * published_words(p->cfg.num_steps), p->stream);
    return fail(MPPI_ERR_BUSY, "a rollout launch could not hand the updated controls over between its workgroups in time: "
                               "not all of them were running side by side (device shared or masked); the control sequence "
                               "of this call is not valid -- set it again; from now on this handle updates through a "
                               "launch of its own per iteration");
  }
  return MPPI_OK;
}

// after the stream has drained: did a launch give up waiting for the noise generator on the second stream, or the
// generator's gate for the launch it follows (rollout_kernels.h: DevParams::noise_flag / progress)?  Then the two
// could not run side by side -- a tool that executes one kernel at a time (rocprofv3 --pmc), a device shared in an odd
// way -- and this handle orders its streams with events from now on.
static int check_flag_fault(mppi_planner* p) {
  if (p->flag_fault_host && *p->flag_fault_host != 0u) {
    *p->flag_fault_host = 0u;
    p->stream_flags_off = true;
    p->progress_capable_last = false;
    drop_graphs(p);
    return fail(MPPI_ERR_BUSY, "a launch waited in vain for the noise generator on the planner's second stream (or the generator "
                               "for the launch it follows): the two cannot run side by side here (a tool that executes one "
                               "kernel at a time, such as rocprofv3 --pmc?).  The control sequence of this call is not valid -- "
                               "set it again; from now on this handle orders its streams with events (MPPI_NO_NOISE_FLAG=1 does "
                               "so from the start)");
  }
  return MPPI_OK;
}

// what follows every wait for the stream (the wait itself: wait_for_stream on the control path, else drain_stream)
static int after_drain(mppi_planner* p, bool review = true) {
  if (review) review_speculation(p);  // (the control loop's waits: is speculating on this map paying?)
  TRY(check_flag_fault(p));
  TRY(check_peer_fault(p));
  TRY(check_fold_fault(p));
  return MPPI_OK;
}

static int drain_stream(mppi_planner* p) {
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->iterations_since_wait = 0;
  return after_drain(p, /*review=*/false);
}

extern "C" int mppi_planner_synchronize(mppi_planner* p) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  HIP_TRY(hipSetDevice(p->cfg.device));
  TRY(wait_for_stream(p));
  TRY(after_drain(p));
  return finish_timing(p);
}

extern "C" int mppi_planner_set_fold_poll_limit(mppi_planner* p, int polls) {
  REQUIRE(p && polls >= 1, MPPI_ERR_INVALID, "bad argument");
  p->fold_max_polls = polls;
  drop_graphs(p);  // (a by-value argument of the captured launches)
  // ... and the handle folds its updates into rollout launches again: one co-tenant that held CUs for a moment need not
  // cost a handle the one-launch iteration for the rest of its life (a device that is still shared raises the fault again)
  p->fold_off = false;
  return MPPI_OK;
}

extern "C" int mppi_planner_fold_state(mppi_planner* p, int* folding, long* faults) {
  REQUIRE(p && folding && faults, MPPI_ERR_INVALID, "NULL argument");
  *folding = p->fold_off ? 0 : 1;
  *faults = (long)p->fold_faults;
  return MPPI_OK;
}

// test hook: workgroups that do nothing but hold a compute unit (100 KiB of LDS each) for a while
__global__ __launch_bounds__(64) void k_debug_occupy(unsigned long long ticks, int* sink) {
  extern __shared__ int occupy_lds[];
  occupy_lds[threadIdx.x] = (int)threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (sink && occupy_lds[threadIdx.x ^ 1] == -1) *sink = 1;
}

extern "C" int mppi_debug_occupy_cus(int device, int workgroups, int milliseconds) {
  REQUIRE(workgroups >= 1 && workgroups <= 4096 && milliseconds >= 1 && milliseconds <= 2000, MPPI_ERR_INVALID, "bad argument");
  HIP_TRY(hipSetDevice(device));
  // a stream of another priority: the runtime maps streams of one priority onto a small pool of hardware queues,
  // and two streams that share a queue run one after the other (measured: the planner's loop simply waited)
  // (a test hook: one stream per device, never destroyed -- hipStreamDestroy waits for the stream's work)
  static hipStream_t sides[64] = {};
  REQUIRE(device >= 0 && device < 64, MPPI_ERR_INVALID, "device %d", device);
  hipStream_t& side = sides[device];
  if (!side) {
    int least = 0, greatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIP_TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, greatest));
  }
  int rate_khz = 0;
  HIP_TRY(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, device));
  if (rate_khz <= 0) rate_khz = 100000;
  const size_t lds = 100 * 1024;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_debug_occupy), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_debug_occupy, dim3(workgroups), dim3(64), lds, side, (unsigned long long)rate_khz * (unsigned long long)milliseconds,
                     (int*)nullptr);
  HIP_TRY(hipGetLastError());
  return MPPI_OK;
}

extern "C" int mppi_planner_solve(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, float* u_out) {
  REQUIRE(p && u_out, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  HIP_TRY(hipSetDevice(p->cfg.device));
  TraceRange tr("mppi:solve");
  TRY(check_tdms(p, lin, ang));
  TRY(sample_for_solve(p, lin, ang));
  p->mirror_done = false;
  TRY(run_iterations(p, lin, ang, p->params.num_opt, /*timed=*/false, /*mirror_last=*/true));
  const size_t u_bytes = sizeof(float2) * (size_t)p->B * (size_t)p->cfg.num_steps;
  // with at least one iteration the last update kernel has written the host-mapped mirror (a
  // replayed graph has not: its launches are the loop's ordinary ones)
  if (p->params.num_opt < 1 || !p->mirror_done)
    HIP_TRY(hipMemcpyAsync(p->u_host, p->u, u_bytes, hipMemcpyDeviceToHost, p->stream));
  TRY(wait_for_stream(p));
  TRY(after_drain(p));
  memcpy(u_out, p->u_host, u_bytes);
  return finish_timing(p);
}

// ---- the simulated world on the device (terrain.py:586-608, 750-785) -------
struct mppi_world {
  int device = 0;
  int rows = 0, cols = 0;
  double res = 1.0, xlo = 0.0, ylo = 0.0;
  double* lin = nullptr;
  double* ang = nullptr;
  uint64_t draws = 0;  // sample_true_dist calls so far (Philox counter word)
};

static WorldGrid world_grid(const mppi_world* w) {
  WorldGrid g;
  g.lin = w->lin; g.ang = w->ang; g.rows = w->rows; g.cols = w->cols;
  g.res = w->res; g.xlo = w->xlo; g.ylo = w->ylo;
  return g;
}

extern "C" int mppi_world_destroy(mppi_world* w) {
  if (!w) return MPPI_OK;
  (void)hipSetDevice(w->device);
  dev_free(w->lin);
  dev_free(w->ang);
  delete w;
  return MPPI_OK;
}

extern "C" int mppi_world_create(int device, int rows, int cols, double res, double xlo, double ylo,
                                 const double* lin, const double* ang, mppi_world** out) {
  REQUIRE(out, MPPI_ERR_INVALID, "NULL out");
  REQUIRE(rows > 0 && cols > 0 && (long)rows * cols < (1L << 30), MPPI_ERR_INVALID, "bad grid shape %d x %d", rows, cols);
  REQUIRE(res > 0.0 && std::isfinite(res) && std::isfinite(xlo) && std::isfinite(ylo), MPPI_ERR_INVALID,
          "bad resolution / limits");
  int count = 0;
  TRY(mppi_device_count(&count));
  REQUIRE(device >= 0 && device < count, MPPI_ERR_NO_DEVICE, "device %d of %d", device, count);
  HIP_TRY(hipSetDevice(device));
  mppi_world* w = new mppi_world();
  w->device = device; w->rows = rows; w->cols = cols; w->res = res; w->xlo = xlo; w->ylo = ylo;
  const size_t cells = (size_t)rows * cols;
  int rc = dev_alloc(&w->lin, cells);
  if (rc == MPPI_OK) rc = dev_alloc(&w->ang, cells);
  if (rc != MPPI_OK) { mppi_world_destroy(w); return rc; }
  hipError_t e = hipSuccess;
  if (lin) e = hipMemcpy(w->lin, lin, cells * sizeof(double), hipMemcpyHostToDevice);
  else e = hipMemset(w->lin, 0, cells * sizeof(double));
  if (e == hipSuccess) {
    if (ang) e = hipMemcpy(w->ang, ang, cells * sizeof(double), hipMemcpyHostToDevice);
    else e = hipMemset(w->ang, 0, cells * sizeof(double));
  }
  if (e != hipSuccess) { mppi_world_destroy(w); HIP_TRY(e); }
  *out = w;
  return MPPI_OK;
}

extern "C" int mppi_world_get_grids(mppi_world* w, double* lin, double* ang) {
  REQUIRE(w && lin && ang, MPPI_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(w->device));
  const size_t bytes = (size_t)w->rows * w->cols * sizeof(double);
  HIP_TRY(hipMemcpy(lin, w->lin, bytes, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(ang, w->ang, bytes, hipMemcpyDeviceToHost));
  return MPPI_OK;
}

extern "C" int mppi_world_get(mppi_world* w, const double* xy, int count, double* lin_out, double* ang_out) {
  REQUIRE(w && xy && lin_out && ang_out, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(count >= 0, MPPI_ERR_INVALID, "count < 0");
  if (count == 0) return MPPI_OK;
  HIP_TRY(hipSetDevice(w->device));
  double *xy_d = nullptr, *out_d = nullptr;
  TRY(dev_alloc(&xy_d, (size_t)2 * count));
  int rc = dev_alloc(&out_d, (size_t)2 * count);
  if (rc != MPPI_OK) { dev_free(xy_d); return rc; }
  hipError_t e = hipMemcpy(xy_d, xy, sizeof(double) * 2 * (size_t)count, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_world_get, dim3(ceil_div(count, 256)), dim3(256), 0, 0, world_grid(w), xy_d, count, out_d,
                       out_d + count);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(lin_out, out_d, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(ang_out, out_d + count, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost);
  dev_free(xy_d);
  dev_free(out_d);
  HIP_TRY(e);
  return MPPI_OK;
}

extern "C" int mppi_world_sample_true_dist(mppi_world* w, const int32_t* terrain_of_cell, int n_terrains,
                                           const double* lin_pool, const double* ang_pool, int pool_len,
                                           uint64_t seed) {
  REQUIRE(w && terrain_of_cell && lin_pool && ang_pool, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(n_terrains > 0 && pool_len > 0, MPPI_ERR_INVALID, "empty terrain table");
  HIP_TRY(hipSetDevice(w->device));
  const int cells = w->rows * w->cols;
  int32_t* ids = nullptr;
  double* pools = nullptr;
  const size_t pool_count = (size_t)n_terrains * pool_len;
  TRY(dev_alloc(&ids, (size_t)cells));
  int rc = dev_alloc(&pools, 2 * pool_count);
  if (rc != MPPI_OK) { dev_free(ids); return rc; }
  hipError_t e = hipMemcpy(ids, terrain_of_cell, sizeof(int32_t) * (size_t)cells, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(pools, lin_pool, sizeof(double) * pool_count, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(pools + pool_count, ang_pool, sizeof(double) * pool_count, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_world_sample, dim3(ceil_div(cells, 256)), dim3(256), 0, 0, ids, cells, n_terrains, pools,
                       pools + pool_count, pool_len, seed, w->draws, w->lin, w->ang);
    e = hipGetLastError();
    ++w->draws;
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  dev_free(ids);
  dev_free(pools);
  HIP_TRY(e);
  return MPPI_OK;
}

// The notebooks' closed loop (test.ipynb cell 4) for every problem of a batched handle, without a
// host round trip per control step: {sample grids, num_opt iterations, k_world_step} x max_steps,
// all on the planner's stream.  The host looks at a device-mapped counter every `check_every`
// steps and stops once every problem has reached its goal.
extern "C" int mppi_planner_closed_loop(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, mppi_world* w, int max_steps,
                                        double dt, double goal_tolerance, const double* x_init, double* xhist, float* uhist,
                                        int* steps_taken) {
  REQUIRE(p && w && xhist && uhist && steps_taken, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  REQUIRE(p->inst_set, MPPI_ERR_STATE, "closed_loop needs per-problem start states (mppi_planner_set_instances)");
  REQUIRE(p->cfg.world_size == 1, MPPI_ERR_STATE, "closed_loop drives an unsharded handle");
  REQUIRE(w->device == p->cfg.device, MPPI_ERR_INVALID, "world on device %d, planner on %d", w->device, p->cfg.device);
  REQUIRE(max_steps >= 1 && max_steps <= (1 << 20), MPPI_ERR_INVALID, "max_steps %d", max_steps);
  HIP_TRY(hipSetDevice(p->cfg.device));
  TraceRange tr("mppi:closed_loop");
  TRY(check_tdms(p, lin, ang));
  const int B = p->B, T = p->cfg.num_steps;
  if (!p->loop_done_count) {
    HIP_TRY(hipHostMalloc((void**)&p->loop_done_count, sizeof(int), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&p->loop_done_count_dev, p->loop_done_count, 0));
    TRY(dev_alloc(&p->loop_state, (size_t)3 * B));
    TRY(dev_alloc(&p->loop_done, (size_t)B));
    TRY(dev_alloc(&p->loop_u_final, (size_t)B * T));
  }
  if (max_steps > p->loop_capacity) {
    dev_free(p->loop_xhist);
    dev_free(p->loop_uhist);
    p->loop_capacity = 0;
    TRY(dev_alloc(&p->loop_xhist, (size_t)3 * B * ((size_t)max_steps + 1)));
    TRY(dev_alloc(&p->loop_uhist, (size_t)B * (size_t)max_steps));
    p->loop_capacity = max_steps;
  }
  // initial state and log (rows never reached stay NaN, as in the notebook's np.zeros(...)*np.nan)
  const size_t rows = (size_t)max_steps + 1;
  std::vector<double> x0((size_t)3 * B);
  for (int b = 0; b < B; ++b)
    for (int k = 0; k < 3; ++k)
      x0[(size_t)3 * b + k] = x_init ? x_init[3 * b + k]
                                     : (double)(k == 0 ? p->inst_host[b].x0 : (k == 1 ? p->inst_host[b].y0 : p->inst_host[b].th0));
  if (x_init) {  // the planner sees the float32 of the float64 state (mppi.py:214-234)
    for (int b = 0; b < B; ++b) {
      p->inst_host[b].x0 = (float)x_init[3 * b]; p->inst_host[b].y0 = (float)x_init[3 * b + 1];
      p->inst_host[b].th0 = (float)x_init[3 * b + 2];
    }
    p->inst_dirty = true;
  }
  const double nan = std::nan("");
  for (size_t i = 0; i < (size_t)B * rows * 3; ++i) xhist[i] = nan;
  for (size_t i = 0; i < (size_t)B * max_steps * 2; ++i) uhist[i] = std::nanf("");
  for (int b = 0; b < B; ++b) memcpy(xhist + (size_t)b * rows * 3, &x0[(size_t)3 * b], 3 * sizeof(double));
  HIP_TRY(hipMemcpyAsync(p->loop_xhist, xhist, sizeof(double) * 3 * B * rows, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->loop_uhist, uhist, sizeof(float2) * (size_t)B * max_steps, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemcpyAsync(p->loop_state, x0.data(), sizeof(double) * 3 * B, hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipMemsetAsync(p->loop_done, 0, sizeof(int) * (size_t)B, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  *p->loop_done_count = 0;
  WorldLoop L;
  memset(&L, 0, sizeof(L));
  auto plan_loop = [&]() -> int {
    // the window plan the host would make for every new start state, handed to the step kernel
    TRY(ensure_packed(p, lin, ang));
    DevParams plan = make_dev_params(p, lin, ang);
    size_t lds_unused = 0;
    const bool windowed = plan_lds_window(p, plan, &lds_unused);
    TRY(upload_instances(p));
    L.state = p->loop_state; L.xhist = p->loop_xhist; L.uhist = p->loop_uhist; L.done = p->loop_done;
    L.done_count = p->loop_done_count_dev;
    L.u_final = p->loop_u_final;
    L.max_steps = max_steps;
    L.dt = dt > 0.0 ? dt : (double)p->params.dt;
    L.goal_tolerance = goal_tolerance;
    L.xlo = (double)p->params.xlo; L.ylo = (double)p->params.ylo; L.res = (double)plan.res;
    L.map_rows = plan.rows; L.map_pitch = p->pitch16;
    L.win_rows = plan.win_rows; L.win_cols = plan.win_cols;
    // (a window smaller than the map is a reach square: its half width is what plan_lds_window used)
    L.win_active = windowed && (plan.win_rows < plan.rows || plan.win_cols < p->pitch16) ? 1 : 0;
    if (L.win_active) {
      const mppi_params& a = p->params;
      double vmax = std::fmax(std::fabs((double)a.vrange[0]), std::fabs((double)a.vrange[1]));
      double trmax = std::fmax(std::fabs(plan.lin_lo), std::fabs(plan.lin_lo + (double)plan.lin_max_byte * plan.lin_ratio));
      L.reach = (int)((long)std::ceil((double)T * (double)a.dt * vmax * trmax / (double)a.res) + 2);
    }
    return MPPI_OK;
  };
  const WorldGrid G = world_grid(w);
  const int check_every = 16;
  HIP_TRY(hipEventRecord(p->ev_begin, p->stream));
  int launched = 0;
  for (int step = 0; step < max_steps; ++step) {
    TRY(sample_for_solve(p, lin, ang));
    if (step == 0) TRY(plan_loop());  // (needs the sampled grids packed: after the first draw)
    TRY(run_iterations(p, lin, ang, p->params.num_opt, /*timed=*/false));
    hipLaunchKernelGGL(k_world_step, dim3(B), dim3(256), sizeof(float2) * (size_t)T, p->stream, G, L, p->inst_dev, p->u,
                       T, step);
    HIP_TRY(hipGetLastError());
    ++launched;
    if (launched % check_every == 0) {
      TraceRange tr_wait("mppi:closed_loop_check");
      HIP_TRY(hipStreamSynchronize(p->stream));
      review_speculation(p);  // (the host has waited anyway: is speculating on this map paying?)
      TRY(check_fold_fault(p));
      if (*p->loop_done_count >= B) break;
      // the host-side guards of the next launches (|theta| bounds of the incremental trig, window
      // plan) look at the start states: bring the mirror of the per-problem records up to date
      HIP_TRY(hipMemcpy(p->inst_host.data(), p->inst_dev, sizeof(BatchInst) * (size_t)B, hipMemcpyDeviceToHost));
    }
  }
  // finished problems get back the controls they had at the goal (the planner went on until the host looked)
  if (launched > 0) {
    hipLaunchKernelGGL(k_world_restore, dim3(B), dim3(256), 0, p->stream, L, p->u, p->u_prev, T);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(p->ev_end, p->stream));
  p->elapsed_pending = true;
  p->last_iterations = launched * (p->params.num_opt > 0 ? p->params.num_opt : 1);
  HIP_TRY(hipMemcpyAsync(xhist, p->loop_xhist, sizeof(double) * 3 * B * rows, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipMemcpyAsync(uhist, p->loop_uhist, sizeof(float2) * (size_t)B * max_steps, hipMemcpyDeviceToHost, p->stream));
  std::vector<int> done((size_t)B);
  HIP_TRY(hipMemcpyAsync(done.data(), p->loop_done, sizeof(int) * (size_t)B, hipMemcpyDeviceToHost, p->stream));
  // the per-problem records the step kernel has been writing: bring the host mirror up to date
  HIP_TRY(hipMemcpyAsync(p->inst_host.data(), p->inst_dev, sizeof(BatchInst) * (size_t)B, hipMemcpyDeviceToHost, p->stream));
  TRY(drain_stream(p));
  p->inst_dirty = false;
  for (int b = 0; b < B; ++b) steps_taken[b] = done[b] ? done[b] : launched;
  return finish_timing(p);
}

// ---- stage-level entry points -------------------------------------------------
extern "C" int mppi_planner_sample_noise(mppi_planner* p) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  HIP_TRY(hipSetDevice(p->cfg.device));
  if (p->primed) {
    // the next block of the noise sequence has already been generated (by the last iteration of
    // the previous call): hand it out instead of skipping it
    if (p->noise_on_side_stream) HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_noise_ready, 0));
    p->noise_on_side_stream = false;
    p->noise_cur ^= 1;
    p->noise = p->noise_buf[p->noise_cur];
    p->primed = false;
    p->noise_virtual = false;
    TRY(drain_stream(p));
    return MPPI_OK;
  }
  p->noise_virtual = false;
  TRY(launch_noise(p, p->noise));
  TRY(drain_stream(p));
  return MPPI_OK;
}

extern "C" int mppi_planner_set_noise(mppi_planner* p, const float* noise) {
  REQUIRE(p && noise, MPPI_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(p->cfg.device));
  size_t count = (size_t)p->n_local * p->cfg.num_steps;
  p->noise_virtual = false;
  HIP_TRY(hipMemcpyAsync(p->staging, noise, count * sizeof(float2), hipMemcpyHostToDevice, p->stream));
  hipLaunchKernelGGL(k_noise_to_device_layout, dim3(ceil_div((long)count, 256)), dim3(256), 0, p->stream,
                     p->staging, p->n_local, p->cfg.num_steps, p->noise);
  HIP_TRY(hipGetLastError());
  TRY(drain_stream(p));
  return MPPI_OK;
}

extern "C" int mppi_planner_get_noise(mppi_planner* p, float* noise) {
  REQUIRE(p && noise, MPPI_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(p->cfg.device));
  size_t count = (size_t)p->n_local * p->cfg.num_steps;
  TRY(materialize_noise(p));
  hipLaunchKernelGGL(k_noise_to_host_layout, dim3(ceil_div((long)count, 256)), dim3(256), 0, p->stream, p->noise,
                     p->n_local, p->cfg.num_steps, p->staging);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(noise, p->staging, count * sizeof(float2), hipMemcpyDeviceToHost, p->stream));
  TRY(drain_stream(p));
  return MPPI_OK;
}

extern "C" int mppi_planner_rollout(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  HIP_TRY(hipSetDevice(p->cfg.device));
  TRY(check_tdms(p, lin, ang));
  TRY(ensure_packed(p, lin, ang));
  DevParams d = make_dev_params(p, lin, ang);
  TRY(launch_rollout(p, d));
  if (p->m_count > 1) {
    // samples sharded over ranks: with a communicator the slabs are exchanged and reduced here, as
    // inside the iteration loop; without one the caller owes sample_costs_local / sample_costs_apply
    // before any update (the costs hold the CVaR over this rank's samples only)
    if (p->comm) TRY(exchange_sample_costs(p));
    else p->sample_costs_local_only = true;
  }
  TRY(drain_stream(p));
  return MPPI_OK;
}

extern "C" int mppi_planner_set_costs(mppi_planner* p, const float* costs) {
  REQUIRE(p && costs, MPPI_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(p->cfg.device));
  HIP_TRY(hipMemcpyAsync(p->costs, costs, sizeof(float) * (size_t)p->n_local, hipMemcpyHostToDevice, p->stream));
  p->tile_packets_fresh = false;
  p->scan_packets_fresh = false;
  p->sample_costs_local_only = false;
  TRY(drain_stream(p));
  return MPPI_OK;
}

extern "C" int mppi_planner_get_costs(mppi_planner* p, float* costs) {
  REQUIRE(p && costs, MPPI_ERR_INVALID, "NULL argument");
  return copy_out(p, costs, p->costs, sizeof(float) * (size_t)p->n_local);
}

extern "C" int mppi_planner_get_sample_costs(mppi_planner* p, float* costs) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  REQUIRE(p->cfg.mode == MPPI_MODE_TDM, MPPI_ERR_STATE, "per-sample costs exist in MPPI_MODE_TDM only");
  if (!costs) {  // arm: the next rollout records them
    p->want_sample_costs = true;
    return MPPI_OK;
  }
  REQUIRE(p->sample_costs, MPPI_ERR_STATE, "call once with NULL before the rollout to arm recording");
  // (samples sharded over GPUs: all M = count * num_grid_samples costs, gathered)
  return copy_out(p, costs, p->sample_costs,
                  sizeof(float) * (size_t)p->n_local * p->cfg.num_grid_samples * (size_t)p->m_count);
}

// ---- CVaR mode with the traction samples sharded over GPUs -------------
extern "C" int mppi_planner_set_sample_sharding(mppi_planner* p, int rank, int count) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  REQUIRE(count >= 1 && rank >= 0 && rank < count, MPPI_ERR_INVALID, "bad sample shard %d of %d", rank, count);
  REQUIRE(count == 1 || p->cfg.mode == MPPI_MODE_TDM, MPPI_ERR_INVALID, "only MPPI_MODE_TDM has samples to shard");
  REQUIRE(count == 1 || p->cfg.world_size == 1, MPPI_ERR_INVALID,
          "a handle shards either its control samples (world_size %d) or its traction samples, not both",
          p->cfg.world_size);
  REQUIRE(count == 1 || (p->cfg.num_grid_samples & 1) == 0, MPPI_ERR_INVALID,
          "num_grid_samples per shard (%d) must be even", p->cfg.num_grid_samples);
  REQUIRE(!p->comm || (rank == p->m_rank && count == p->m_count), MPPI_ERR_STATE,
          "the communicator was created for another shard layout");
  HIP_TRY(hipSetDevice(p->cfg.device));
  if (count != p->m_count) {
    TRY(drain_stream(p));
    dev_free(p->slabs);
    dev_free(p->sample_costs);
    drop_graphs(p);
  }
  p->m_rank = rank;
  p->m_count = count;
  return MPPI_OK;
}

extern "C" int mppi_planner_sample_costs_local(mppi_planner* p, float* slab) {
  REQUIRE(p && slab, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(p->m_count > 1, MPPI_ERR_STATE, "samples are not sharded (mppi_planner_set_sample_sharding)");
  REQUIRE(p->slabs, MPPI_ERR_STATE, "no rollout yet");
  const size_t len = (size_t)p->n_local * p->cfg.num_grid_samples;
  return copy_out(p, slab, p->slabs + (size_t)p->m_rank * len, sizeof(float) * len);
}

extern "C" int mppi_planner_sample_costs_apply(mppi_planner* p, const float* slabs, int count) {
  REQUIRE(p && slabs, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(p->m_count > 1 && count == p->m_count, MPPI_ERR_INVALID, "expected the slabs of %d shards, got %d",
          p->m_count, count);
  REQUIRE(p->slabs && p->params_set, MPPI_ERR_STATE, "no rollout yet");
  HIP_TRY(hipSetDevice(p->cfg.device));
  const size_t len = (size_t)p->n_local * p->cfg.num_grid_samples;
  HIP_TRY(hipMemcpyAsync(p->slabs, slabs, sizeof(float) * len * (size_t)count, hipMemcpyHostToDevice, p->stream));
  TRY(launch_cvar_reduce(p));
  TRY(drain_stream(p));
  return MPPI_OK;
}

extern "C" int mppi_planner_update(mppi_planner* p) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  REQUIRE(!p->sample_costs_local_only, MPPI_ERR_STATE,
          "samples sharded over %d ranks: the costs of the last rollout cover this rank's samples only -- exchange "
          "them first (mppi_planner_sample_costs_local / _apply, or a communicator)", p->m_count);
  HIP_TRY(hipSetDevice(p->cfg.device));
  TRY(launch_update(p, false));
  TRY(drain_stream(p));
  return MPPI_OK;
}

extern "C" int mppi_planner_packet_len(mppi_planner* p, int* doubles) {
  REQUIRE(p && doubles, MPPI_ERR_INVALID, "NULL argument");
  *doubles = p->B * packet_len(p->cfg.num_steps);
  return MPPI_OK;
}

extern "C" int mppi_planner_update_local(mppi_planner* p, double* packet) {
  REQUIRE(p && packet, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  REQUIRE(!p->sample_costs_local_only, MPPI_ERR_STATE, "sample shards: exchange the per-sample costs before the update");
  HIP_TRY(hipSetDevice(p->cfg.device));
  TRY(launch_update_local(p, false));
  const int len = p->B * packet_len(p->cfg.num_steps);
  HIP_TRY(hipMemcpyAsync(packet, p->packets + (size_t)p->cfg.rank * len, sizeof(double) * (size_t)len,
                         hipMemcpyDeviceToHost, p->stream));
  TRY(drain_stream(p));
  return MPPI_OK;
}

extern "C" int mppi_planner_update_apply(mppi_planner* p, const double* packets, int count) {
  REQUIRE(p && packets, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(count == p->cfg.world_size, MPPI_ERR_INVALID, "expected %d packets, got %d", p->cfg.world_size, count);
  HIP_TRY(hipSetDevice(p->cfg.device));
  const int len = p->B * packet_len(p->cfg.num_steps);
  HIP_TRY(hipMemcpyAsync(p->packets, packets, sizeof(double) * (size_t)len * (size_t)count,
                         hipMemcpyHostToDevice, p->stream));
  TRY(launch_apply(p));
  TRY(drain_stream(p));
  return MPPI_OK;
}

extern "C" int mppi_planner_update_apply_and_rollout(mppi_planner* p, const double* packets, int count,
                                                     mppi_tdm* lin, mppi_tdm* ang) {
  REQUIRE(p && packets, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  REQUIRE(count == p->cfg.world_size, MPPI_ERR_INVALID, "expected %d packets, got %d", p->cfg.world_size, count);
  // (control samples sharded; a handle whose traction SAMPLES are sharded exchanges cost slabs, not packets:
  //  mppi_planner_rollout / sample_costs_local / sample_costs_apply / update)
  REQUIRE(p->m_count == 1, MPPI_ERR_STATE, "update_apply_and_rollout serves handles that shard their control samples");
  HIP_TRY(hipSetDevice(p->cfg.device));
  TRY(check_tdms(p, lin, ang));
  TRY(ensure_packed(p, lin, ang));
  const int len = p->B * packet_len(p->cfg.num_steps);
  HIP_TRY(hipMemcpyAsync(p->packets, packets, sizeof(double) * (size_t)len * (size_t)count,
                         hipMemcpyHostToDevice, p->stream));
  DevParams d = make_dev_params(p, lin, ang);
  // the rollout launch applies the update when it is one that can (launch_rollout settles it otherwise)
  if (next_rollout_applies_updates(p)) p->apply_pending = true;
  else TRY(launch_apply(p));
  const int rc = launch_rollout(p, d);
  if (p->apply_pending) {  // (the rollout launch failed before it could take the packets: the update must not be lost)
    const int ra = launch_apply(p);
    if (rc == MPPI_OK) TRY(ra);
  }
  TRY(rc);
  TRY(drain_stream(p));
  return MPPI_OK;
}

extern "C" int mppi_planner_get_weights(mppi_planner* p, float* weights) {
  REQUIRE(p && weights, MPPI_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(p->cfg.device));
  hipLaunchKernelGGL(k_weights_out, dim3(ceil_div(p->n_local, 256)), dim3(256), 0, p->stream, p->costs, p->stats,
                     p->params.lambda_weight, p->n_local, p->n_inst, p->weights_out);
  HIP_TRY(hipGetLastError());
  return copy_out(p, weights, p->weights_out, sizeof(float) * (size_t)p->n_local);
}

extern "C" int mppi_planner_get_state_rollout(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, float* out) {
  return mppi_planner_get_instance_state_rollout(p, lin, ang, 0, out);
}

extern "C" int mppi_planner_get_instance_state_rollout(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang,
                                                       int instance, float* out) {
  REQUIRE(p && out, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  REQUIRE(instance >= 0 && instance < p->B, MPPI_ERR_INVALID, "instance %d of %d", instance, p->B);
  REQUIRE(p->B == 1 || p->inst_set, MPPI_ERR_STATE, "instances not set");
  HIP_TRY(hipSetDevice(p->cfg.device));
  TRY(check_tdms(p, lin, ang));
  TRY(ensure_packed(p, lin, ang));  // uses the already sampled grids (mppi.py:572-573)
  TRY(materialize_noise(p));
  DevParams d = make_dev_params(p, lin, ang);
  // one problem of a batched handle: its start state, its controls, its slice of the noise
  struct Rebased {
    float2 *noise, *u, *u_prev;
  } view = {p->noise, p->u, p->u_prev};
  if (p->inst_set) {
    const BatchInst& I = p->inst_host[(size_t)instance];
    d.x0 = I.x0; d.y0 = I.y0; d.th0 = I.th0; d.xg = I.xg; d.yg = I.yg;
    d.inst = nullptr;
    d.n_local = p->n_inst;
    view.noise += (size_t)instance * p->inst_tiles * p->cfg.num_steps * 64;
    view.u += (size_t)instance * p->cfg.num_steps;
    view.u_prev += (size_t)instance * p->cfg.num_steps;
  }
  const int V = p->cfg.num_vis_state_rollouts;
  dim3 grid(ceil_div(V, 64)), block(64);
  switch (p->cfg.mode) {
    case MPPI_MODE_TDM:
      REQUIRE(V <= p->cfg.num_grid_samples, MPPI_ERR_INVALID, "V > M");
      hipLaunchKernelGGL((k_state_rollout<true, false>), grid, block, 0, p->stream, d, p->cells, view.noise,
                         view.u_prev, view.u, V, p->state_rollout);
      break;
    case MPPI_MODE_BAREBONE:
      hipLaunchKernelGGL((k_state_rollout<false, true>), grid, block, 0, p->stream, d, p->cells, view.noise,
                         view.u_prev, view.u, V, p->state_rollout);
      break;
    default:
      hipLaunchKernelGGL((k_state_rollout<false, false>), grid, block, 0, p->stream, d, p->cells, view.noise,
                         view.u_prev, view.u, V, p->state_rollout);
  }
  HIP_TRY(hipGetLastError());
  return copy_out(p, out, p->state_rollout, sizeof(float) * (size_t)V * (p->cfg.num_steps + 1) * 3);
}

extern "C" int mppi_planner_rng_states(mppi_planner* p, uint64_t* out, long capacity, long* count) {
  REQUIRE(p && count, MPPI_ERR_INVALID, "NULL argument");
  *count = p->n_states;
  if (!out || p->n_states == 0) return MPPI_OK;
  REQUIRE(capacity >= p->n_states, MPPI_ERR_INVALID, "capacity %ld < %ld states", capacity, p->n_states);
  return copy_out(p, out, p->states, 2 * sizeof(uint64_t) * (size_t)p->n_states);
}

extern "C" int mppi_planner_set_profiling(mppi_planner* p, int enabled) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  p->profile_stages = enabled != 0;
  return MPPI_OK;
}

extern "C" int mppi_planner_stage_times(mppi_planner* p, float ms[4]) {
  REQUIRE(p && ms, MPPI_ERR_INVALID, "NULL argument");
  TRY(finish_timing(p));
  memcpy(ms, p->stage_ms, sizeof(p->stage_ms));
  return MPPI_OK;
}

// Average duration of the rollout launch and of the update launch over `reps` ordinary iterations
// of the loop: every launch carries its own start / stop events (hipExtLaunchKernelGGL), which the
// runtime fills with the dispatch's begin / end timestamps -- the figures rocprofv3 --kernel-trace
// reports -- so nothing is inserted between the kernels and nothing is synchronised until the end.
// (A mode whose rollout or update is more than one launch reports the LAST launch of each.)
extern "C" int mppi_planner_time_kernels(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, int reps, float* us_rollout,
                                         float* us_update) {
  REQUIRE(p && us_rollout && us_update, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(reps >= 1 && reps <= 4096, MPPI_ERR_INVALID, "reps %d", reps);
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  REQUIRE(!p->graph_on, MPPI_ERR_STATE, "switch graph replay off for kernel timing");
  HIP_TRY(hipSetDevice(p->cfg.device));
  TRY(check_tdms(p, lin, ang));
  while (p->ktime_events.size() < 4 * (size_t)reps) {
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreate(&e));
    p->ktime_events.push_back(e);
  }
  p->used_side_stream = false;
  TRY(run_iterations(p, lin, ang, 2, /*timed=*/false));  // steady state first (and: does this loop use the second stream?)
  p->ktime_use_stamps = p->used_side_stream;
  p->ktime_update_ran.assign((size_t)reps, 1);
  // (the one-wave-per-tile rollout kernels: at most a whole workgroup of 16 waves beyond the tiles; k_update_rows:
  //  16 waves per row and problem)
  p->ktime_waves = std::max(ceil_div(p->n_local, 64) + 16, p->cfg.num_steps * p->B * (kRowThreads / 64) + 16);
  if (p->ktime_use_stamps)  // (32 bytes per wave and launch: at most ~64 MB of stamps)
    reps = std::max(8, std::min(reps, (int)std::min<size_t>(64, ((size_t)64 << 20) / (32 * (size_t)p->ktime_waves))));
  const size_t stamp_words = p->ktime_use_stamps ? 4 * (size_t)p->ktime_waves * (size_t)reps : 0;  // (a loop timed by events needs none)
  if (stamp_words > p->ktime_dev_capacity) {
    dev_free(p->ktime_dev);
    p->ktime_dev_capacity = 0;
    TRY(dev_alloc(&p->ktime_dev, stamp_words));
    p->ktime_dev_capacity = stamp_words;
  }
  if (stamp_words) HIP_TRY(hipMemsetAsync(p->ktime_dev, 0, sizeof(unsigned long long) * stamp_words, p->stream));  // (0: this wave did not stamp)
  p->ktime_index = 0;
  p->ktime_markers = false;
  const int rc = run_iterations(p, lin, ang, reps, /*timed=*/false);
  p->ktime_index = -1;
  p->ktime_use_stamps = false;
  p->kev_start = p->kev_stop = nullptr;
  TRY(rc);
  TRY(drain_stream(p));
  auto between = [&](size_t a, size_t b, double* ms) -> int {
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, p->ktime_events[a], p->ktime_events[b]));
    *ms = (double)f;
    return MPPI_OK;
  };
  double sum[2] = {0.0, 0.0};
  int counted[2] = {0, 0};
  // Every launch's own start / stop event pair -- unless the loop runs the next iteration's noise on a second stream
  // (the throughput regime).  There the runtime stamps the START event of the rollout launch when the queue reaches the
  // cross-stream wait in front of it, i.e. while the previous kernel still runs (round 5: 70.98 us "in loop" for a
  // kernel traced at 48.92 us), and launches that carry events no longer overlap the second stream's kernel the way
  // plain ones do (round 6: the kernel itself -- first wave in to last wave out -- lasted 73 us in such a loop, 48 us in
  // the plain one; an empty dispatch behind the wait and differences of END timestamps were tried: 70.8 and 84.1 us).
  // Such a loop is therefore run PLAIN and its kernels stamp their launches themselves on the device's constant-rate
  // clock: when each wave entered and left (DevParams::ktime, k_update_rows; two 8-byte stores per wave) -- first in to
  // last out is what rocprofv3 reports as the dispatch's begin -> end.
  if (p->ktime_markers) {
    const size_t W = (size_t)p->ktime_waves;
    std::vector<unsigned long long> stamps(4 * W * (size_t)reps);
    HIP_TRY(hipMemcpy(stamps.data(), p->ktime_dev, sizeof(unsigned long long) * stamps.size(), hipMemcpyDeviceToHost));
    int rate_khz = 0;
    HIP_TRY(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, p->cfg.device));
    if (rate_khz <= 0) rate_khz = 100000;
    for (int r = 0; r < reps; ++r)
      for (int k = 0; k < 2; ++k) {
        const unsigned long long* in = stamps.data() + 4 * W * (size_t)r + 2 * W * (size_t)k;
        unsigned long long first_in = ~0ull, last_out = 0ull;
        for (size_t w = 0; w < W; ++w) {
          if (in[w] != 0ull && in[w] < first_in) first_in = in[w];
          if (in[W + w] > last_out) last_out = in[W + w];
        }
        if (first_in == ~0ull || last_out <= first_in) continue;  // (a kernel family that does not stamp; an update folded elsewhere)
        sum[k] += (double)(last_out - first_in) / (double)rate_khz;  // ms
        ++counted[k];
      }
    // (nothing stamped: reported as 0 -- bench.py then says the launch was not timed in the loop)
    *us_rollout = (float)(1e3 * sum[0] / std::max(1, counted[0]));
    *us_update = (float)(1e3 * sum[1] / reps);
    return MPPI_OK;
  }
  for (int r = 0; r < reps; ++r) {
    double ms = 0.0;
    TRY(between(4 * (size_t)r, 4 * (size_t)r + 1, &ms));
    sum[0] += ms;
    ++counted[0];
    if (!p->ktime_update_ran[(size_t)r]) continue;  // (applied inside the next rollout launch: counted there)
    TRY(between(4 * (size_t)r + 2, 4 * (size_t)r + 3, &ms));
    sum[1] += ms;
    ++counted[1];
  }
  *us_rollout = (float)(1e3 * sum[0] / std::max(1, counted[0]));
  *us_update = (float)(1e3 * sum[1] / reps);  // (per ITERATION: an update folded into the next rollout launch has no launch of its own)
  return MPPI_OK;
}

extern "C" int mppi_planner_last_elapsed_ms(mppi_planner* p, float* ms) {
  REQUIRE(p && ms, MPPI_ERR_INVALID, "NULL argument");
  TRY(finish_timing(p));
  *ms = p->last_elapsed_ms;
  return MPPI_OK;
}

// Developer measurement (profiles/r01_ablation.md): `iterations` x {noise,
// rollout, update} launched directly versus replayed from a hipGraph captured off the planner's
// stream, `replays` times each, wall clock per iteration in microseconds.  A measurement only: a
// replay reuses the captured by-value arguments (Philox epoch, start state, noise buffer parity),
// so it is not a way to run the planner.
extern "C" int mppi_planner_graph_probe(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, int iterations, int replays,
                                        float* us_direct, float* us_graph) {
  REQUIRE(p && us_direct && us_graph, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(iterations >= 2 && iterations % 2 == 0 && replays >= 1, MPPI_ERR_INVALID,
          "iterations must be even (noise double buffer) and replays >= 1");
  REQUIRE(p->cfg.world_size == 1 && !p->comm, MPPI_ERR_INVALID, "single-GPU measurement");
  HIP_TRY(hipSetDevice(p->cfg.device));
  const bool profile = p->profile_stages;
  p->profile_stages = false;
  auto now_us = [] {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
  };
  // warm: buffers packed, instances uploaded, kernel attributes set
  TRY(run_iterations(p, lin, ang, iterations));
  HIP_TRY(hipStreamSynchronize(p->stream));
  TRY(finish_timing(p));
  double t0 = now_us();
  for (int r = 0; r < replays; ++r) TRY(run_iterations(p, lin, ang, iterations));
  HIP_TRY(hipStreamSynchronize(p->stream));
  *us_direct = (float)((now_us() - t0) / ((double)replays * iterations));
  TRY(finish_timing(p));

  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  HIP_TRY(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
  int rc = run_iterations(p, lin, ang, iterations);
  hipError_t end = hipStreamEndCapture(p->stream, &graph);
  p->elapsed_pending = false;  // the events were captured, not recorded
  if (rc != MPPI_OK) {
    if (graph) (void)hipGraphDestroy(graph);
    return rc;
  }
  HIP_TRY(end);
  HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  HIP_TRY(hipGraphLaunch(exec, p->stream));  // first launch uploads the executable graph
  HIP_TRY(hipStreamSynchronize(p->stream));
  t0 = now_us();
  for (int r = 0; r < replays; ++r) HIP_TRY(hipGraphLaunch(exec, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  *us_graph = (float)((now_us() - t0) / ((double)replays * iterations));
  (void)hipGraphExecDestroy(exec);
  (void)hipGraphDestroy(graph);
  p->profile_stages = profile;
  return MPPI_OK;
}

// hipGraph replay of the iteration loop (off by default).  See run_iterations.
extern "C" int mppi_planner_set_graph_replay(mppi_planner* p, int iterations_per_graph) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  const int enabled = iterations_per_graph != 0;
  REQUIRE(!enabled || (iterations_per_graph >= 2 && iterations_per_graph % 2 == 0 && iterations_per_graph <= 256),
          MPPI_ERR_INVALID, "iterations_per_graph must be 0 (off) or even in [2, 256], got %d",
          iterations_per_graph);
  REQUIRE(!enabled || p->cfg.world_size == 1 || p->comm, MPPI_ERR_INVALID,
          "graph replay of a sharded handle needs its RCCL communicator first (mppi_planner_comm_init): "
          "a host-staged exchange cannot be captured");
  HIP_TRY(hipSetDevice(p->cfg.device));
  HIP_TRY(hipStreamSynchronize(p->stream));
  drop_graphs(p);
  discard_noise_ahead(p);
  if (enabled && !p->gen_dev) TRY(dev_alloc(&p->gen_dev, (size_t)1));
  // lazy allocations of the launch paths must not happen inside a capture
  if (enabled && p->cfg.mode == MPPI_MODE_DET && !p->cc_scratch)
    TRY(dev_alloc(&p->cc_scratch, (size_t)ceil_div(p->n_local, 64) * 64 * p->cfg.num_steps));
  if (enabled) HIP_TRY(hipMemset(p->gen_dev, 0, sizeof(unsigned long long)));
  p->bumps_launched = 0;
  p->graph_warm = false;
  p->graph_on = enabled != 0;
  if (enabled) p->graph_chunk = iterations_per_graph;
  return MPPI_OK;
}

extern "C" int mppi_planner_graph_stats(mppi_planner* p, long* captures, long* replays) {
  REQUIRE(p && captures && replays, MPPI_ERR_INVALID, "NULL argument");
  *captures = p->graph_captures;
  *replays = p->graph_replays;
  return MPPI_OK;
}

extern "C" int mppi_planner_set_debug_flags(mppi_planner* p, int flags) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  p->debug_flags = flags;
  drop_graphs(p);  // (a captured graph holds the kernels chosen under the old flags)
  p->graph_warm = false;  // (another kernel family may run next: its host-side effects happen in a direct iteration)
  return MPPI_OK;
}

extern "C" int mppi_planner_describe_last_rollout(mppi_planner* p, char* buf, int capacity) {
  REQUIRE(p && buf && capacity > 0, MPPI_ERR_INVALID, "bad argument");
  snprintf(buf, (size_t)capacity, "%s", p->last_rollout.c_str());
  return MPPI_OK;
}

// developer instrumentation: see MPPI_STAMP in device_math.h
extern "C" int mppi_debug_read_stamps(unsigned long long* out, int count, int clear) {
#ifdef MPPI_STAMPS
  REQUIRE(out && count >= 0 && count <= 4096, MPPI_ERR_INVALID, "bad stamp request");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)count, 0,
                              hipMemcpyDeviceToHost));
  if (clear) {
    static unsigned long long zeros[4096];
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice));
  }
  return MPPI_OK;
#else
  (void)out; (void)count; (void)clear;
  return fail(MPPI_ERR_STATE, "this library was built without -DMPPI_STAMPS (make stamps)");
#endif
}

extern "C" int mppi_selftest_philox(int device, int* mismatches) {
  REQUIRE(mismatches, MPPI_ERR_INVALID, "NULL argument");
  mppi_device_props pr;
  TRY(mppi_device_props_get(device, &pr));
  HIP_TRY(hipSetDevice(device));
  int* d = nullptr;
  TRY(dev_alloc(&d, (size_t)1));
  HIP_TRY(hipMemset(d, 0, sizeof(int)));
  hipLaunchKernelGGL(k_philox_selftest, dim3(256), dim3(256), 0, 0, d);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpy(mismatches, d, sizeof(int), hipMemcpyDeviceToHost));
  dev_free(d);
  return MPPI_OK;
}

extern "C" int mppi_planner_comm_count(mppi_planner* p, int* ranks) {
  REQUIRE(p && ranks, MPPI_ERR_INVALID, "NULL argument");
  *ranks = 0;
  if (!p->comm) return MPPI_OK;
  RCCL_TRY(g_rccl.CommCount(p->comm, ranks));
  return MPPI_OK;
}

// ---- one process, several devices --------------------------------------------------------
// A single control thread driving G shard handles (one per device): the communicators are created
// inside one RCCL group, and every iteration issues the G all-gathers inside one group as well
// (a lone blocking ncclCommInitRank / collective per device from one thread would deadlock).
extern "C" int mppi_group_comm_init(mppi_planner** ps, int count) {
  REQUIRE(ps && count >= 1, MPPI_ERR_INVALID, "bad handle array");
  for (int g = 0; g < count; ++g) {
    REQUIRE(ps[g], MPPI_ERR_INVALID, "NULL planner %d", g);
    REQUIRE(!ps[g]->comm, MPPI_ERR_STATE, "planner %d already has a communicator", g);
    REQUIRE(ps[g]->cfg.world_size == count && ps[g]->cfg.rank == g, MPPI_ERR_INVALID,
            "planner %d is rank %d of %d; the group wants rank %d of %d", g, ps[g]->cfg.rank, ps[g]->cfg.world_size,
            g, count);
    for (int h = 0; h < g; ++h)
      REQUIRE(ps[h]->cfg.device != ps[g]->cfg.device, MPPI_ERR_INVALID,
              "planners %d and %d share device %d (RCCL wants one rank per device)", h, g, ps[g]->cfg.device);
  }
  TRY(rccl_load());
  ncclUniqueId uid;
  RCCL_TRY(g_rccl.GetUniqueId(&uid));
  RCCL_TRY(g_rccl.GroupStart());
  int rc = MPPI_OK;
  for (int g = 0; g < count && rc == MPPI_OK; ++g) {
    if (hipSetDevice(ps[g]->cfg.device) != hipSuccess) { rc = fail(MPPI_ERR_HIP, "hipSetDevice(%d) failed", ps[g]->cfg.device); break; }
    ncclResult_t r = g_rccl.CommInitRank(&ps[g]->comm, count, uid, g);
    if (r != ncclSuccess) rc = fail(MPPI_ERR_COMM, "ncclCommInitRank(rank %d) failed: %s", g, g_rccl.GetErrorString(r));
  }
  ncclResult_t end = g_rccl.GroupEnd();
  if (rc == MPPI_OK && end != ncclSuccess) rc = fail(MPPI_ERR_COMM, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(end));
  if (rc != MPPI_OK)
    for (int g = 0; g < count; ++g) ps[g]->comm = nullptr;  // (whatever was created is leaked rather than half-used)
  return rc;
}

// `iterations` x {per device: noise, rollout, shard packet; one group of all-gathers; per device:
// apply}.  Asynchronous like mppi_planner_iterate_async: synchronise each handle afterwards.
extern "C" int mppi_group_iterate_async(mppi_planner** ps, mppi_tdm** lins, mppi_tdm** angs, int count,
                                        int iterations) {
  REQUIRE(ps && lins && angs && count >= 1 && iterations >= 0, MPPI_ERR_INVALID, "bad arguments");
  {
    // The peer exchange connected on every handle (mppi_group_p2p_connect) and usable by the kernels that will run:
    // nothing to coordinate on the host -- each device's loop is enqueued as if it were alone, the launches wait for
    // each other's numbers on the devices.  (Decided alike for all: they share sizes, mode and maps.)
    bool all_p2p = count > 1;
    for (int g = 0; g < count && all_p2p; ++g) all_p2p = ps[g] && ps[g]->params_set && ps[g]->p2p_on && lins[g] && angs[g];
    // (which kernels will run is decided from the packed maps: pack them first -- a group's very first call would
    //  otherwise take the RCCL path, or fail for want of a communicator it does not need)
    for (int g = 0; g < count && all_p2p; ++g) {
      HIP_TRY(hipSetDevice(ps[g]->cfg.device));
      TRY(check_tdms(ps[g], lins[g], angs[g]));
      TRY(ensure_packed(ps[g], lins[g], angs[g]));
      all_p2p = p2p_usable(ps[g]);
    }
    if (all_p2p) {
      // Every launch of device g spins inside the kernel until the other devices' numbers for the same iteration have
      // arrived, so no device may be handed more launches than its queue takes before the others have theirs: a few
      // iterations per device in turn, round robin (all of device 0's first could fill its launch queue and block this
      // thread with nothing enqueued anywhere else -- until the poll limit raises the fault word).
      constexpr int kTurn = 4;
      for (int g = 0; g < count; ++g) {
        HIP_TRY(hipSetDevice(ps[g]->cfg.device));
        HIP_TRY(hipEventRecord(ps[g]->ev_begin, ps[g]->stream));
      }
      for (int k = 0; k < iterations; k += kTurn)
        for (int g = 0; g < count; ++g) {
          HIP_TRY(hipSetDevice(ps[g]->cfg.device));
          TRY(run_iterations(ps[g], lins[g], angs[g], std::min(kTurn, iterations - k), /*timed=*/false, /*mirror_last=*/false,
                             /*part_of_group_loop=*/true, /*more_follow=*/k + kTurn < iterations));
        }
      for (int g = 0; g < count; ++g) {
        mppi_planner* p = ps[g];
        HIP_TRY(hipSetDevice(p->cfg.device));
        HIP_TRY(hipEventRecord(p->ev_end, p->stream));
        p->elapsed_pending = true;
        p->last_iterations = iterations;
      }
      return MPPI_OK;
    }
  }
  std::vector<DevParams> d((size_t)count);
  std::vector<char> have((size_t)count);
  for (int g = 0; g < count; ++g) {
    mppi_planner* p = ps[g];
    REQUIRE(p && p->comm, MPPI_ERR_STATE, "planner %d has no communicator (mppi_group_comm_init)", g);
    REQUIRE(p->params_set, MPPI_ERR_STATE, "planner %d: params not set", g);
    REQUIRE(!p->graph_on, MPPI_ERR_STATE, "graph replay and group iteration do not combine");
    HIP_TRY(hipSetDevice(p->cfg.device));
    TRY(check_tdms(p, lins[g], angs[g]));
    TRY(ensure_packed(p, lins[g], angs[g]));
    d[(size_t)g] = make_dev_params(p, lins[g], angs[g]);
    have[(size_t)g] = p->primed;
    HIP_TRY(hipEventRecord(p->ev_begin, p->stream));
  }
  for (int k = 0; k < iterations; ++k) {
    for (int g = 0; g < count; ++g) {
      HIP_TRY(hipSetDevice(ps[g]->cfg.device));
      bool h = have[(size_t)g] != 0;
      TRY(launch_iteration(ps[g], d[(size_t)g], h, true, false, /*defer_exchange=*/true));
      have[(size_t)g] = h;
    }
    RCCL_TRY(g_rccl.GroupStart());
    int rc = MPPI_OK;
    for (int g = 0; g < count && rc == MPPI_OK; ++g) {
      mppi_planner* p = ps[g];
      const int len = p->B * packet_len(p->cfg.num_steps);
      ncclResult_t r = g_rccl.AllGather(p->packets + (size_t)p->cfg.rank * len, p->packets, (size_t)len, ncclDouble,
                                        p->comm, p->stream);
      if (r != ncclSuccess) rc = fail(MPPI_ERR_COMM, "ncclAllGather(rank %d) failed: %s", g, g_rccl.GetErrorString(r));
    }
    ncclResult_t end = g_rccl.GroupEnd();
    if (rc == MPPI_OK && end != ncclSuccess) rc = fail(MPPI_ERR_COMM, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(end));
    TRY(rc);
    for (int g = 0; g < count; ++g) {
      HIP_TRY(hipSetDevice(ps[g]->cfg.device));
      TRY(launch_apply(ps[g]));
    }
  }
  for (int g = 0; g < count; ++g) {
    mppi_planner* p = ps[g];
    HIP_TRY(hipSetDevice(p->cfg.device));
    p->primed = have[(size_t)g] != 0;
    HIP_TRY(hipEventRecord(p->ev_end, p->stream));
    p->elapsed_pending = true;
    p->last_iterations = iterations;
  }
  return MPPI_OK;
}

// ---- the peer exchange (include/mppi_hip.h; update_kernels.h: PeerExchange) -----------------------------------
static int p2p_alloc_inbox(mppi_planner* p, hipIpcMemHandle_t* handle) {
  if (p->inbox) {
    // (exported again, i.e. about to be connected again -- after a fault the words of a broken exchange may still be here)
    HIP_TRY(hipStreamSynchronize(p->stream));
    HIP_TRY(hipMemset(p->inbox, 0xff, sizeof(unsigned long long) * inbox_words(p->cfg.world_size, p->cfg.num_steps)));
    if (handle) HIP_TRY(hipIpcGetMemHandle(handle, p->inbox));
    return MPPI_OK;
  }
  REQUIRE(p->cfg.world_size <= kMaxFoldedRanks, MPPI_ERR_INVALID, "the peer exchange serves up to %d ranks", kMaxFoldedRanks);
  const size_t bytes = sizeof(unsigned long long) * inbox_words(p->cfg.world_size, p->cfg.num_steps);
  // fine-grained memory: the peers' stores must be seen by kernels that are already running here.  (Uncached, then
  // ordinary memory as fall-backs where the runtime will not export the former between processes: on ONE device --
  // several ranks sharing a GPU, the test set-up -- device-scope coherence is enough.)
  struct Kind { unsigned flags; const char* name; bool ext; };
  const Kind kinds[] = {{hipDeviceMallocFinegrained, "fine-grained", true}, {hipDeviceMallocUncached, "uncached", true}, {0u, "coarse-grained", false}};
  for (const Kind& k : kinds) {
    void* ptr = nullptr;
    hipError_t e = k.ext ? hipExtMallocWithFlags(&ptr, bytes, k.flags) : hipMalloc(&ptr, bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); continue; }
    hipIpcMemHandle_t h;
    if (handle && hipIpcGetMemHandle(&h, ptr) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(ptr);
      continue;
    }
    if (handle) *handle = h;
    p->inbox = static_cast<unsigned long long*>(ptr);
    p->inbox_kind = k.name;
    HIP_TRY(hipMemset(p->inbox, 0xff, bytes));  // (kNotArrived everywhere; the ping words: any token but all ones)
    return MPPI_OK;
  }
  return fail(MPPI_ERR_HIP, "could not allocate an exportable inbox for the peer exchange");
}

static int p2p_alloc_fault(mppi_planner* p);

extern "C" int mppi_planner_p2p_export(mppi_planner* p, char handle[MPPI_P2P_HANDLE_BYTES]) {
  static_assert(sizeof(hipIpcMemHandle_t) <= MPPI_P2P_HANDLE_BYTES, "hipIpcMemHandle_t larger than expected");
  REQUIRE(p && handle, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(p->cfg.mode == MPPI_MODE_DET && p->B == 1 && p->m_count == 1, MPPI_ERR_INVALID,
          "the peer exchange serves single-problem deterministic-dynamics handles with all their traction samples");
  HIP_TRY(hipSetDevice(p->cfg.device));
  hipIpcMemHandle_t h;
  TRY(p2p_alloc_inbox(p, &h));
  TRY(p2p_alloc_fault(p));
  memset(handle, 0, MPPI_P2P_HANDLE_BYTES);
  memcpy(handle, &h, sizeof(h));
  return MPPI_OK;
}

static int p2p_alloc_fault(mppi_planner* p) {
  if (!p->p2p_fault_host) {
    HIP_TRY(hipHostMalloc((void**)&p->p2p_fault_host, sizeof(unsigned int), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void**)&p->p2p_fault_dev, p->p2p_fault_host, 0));
  }
  *p->p2p_fault_host = 0u;
  return MPPI_OK;
}

static void p2p_disconnect(mppi_planner* p) {
  for (int g = 0; g < kMaxFoldedRanks; ++g) {
    if (p->peer_mapped[g] && p->peer_inbox[g]) (void)hipIpcCloseMemHandle(p->peer_inbox[g]);
    p->peer_inbox[g] = nullptr;
    p->peer_mapped[g] = false;
  }
  p->p2p_on = false;
}

extern "C" int mppi_planner_p2p_connect(mppi_planner* p, const char* handles, int count) {
  REQUIRE(p && handles, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(count == p->cfg.world_size, MPPI_ERR_INVALID, "expected the inbox handles of %d ranks, got %d", p->cfg.world_size, count);
  REQUIRE(p->inbox, MPPI_ERR_STATE, "call mppi_planner_p2p_export first");
  HIP_TRY(hipSetDevice(p->cfg.device));
  HIP_TRY(hipStreamSynchronize(p->stream));
  p2p_disconnect(p);
  drop_graphs(p);
  for (int g = 0; g < count; ++g) {
    if (g == p->cfg.rank) {
      p->peer_inbox[g] = p->inbox;
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, handles + (size_t)g * MPPI_P2P_HANDLE_BYTES, sizeof(h));
    void* ptr = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      p2p_disconnect(p);
      return fail(MPPI_ERR_HIP, "hipIpcOpenMemHandle(rank %d's inbox) failed: %s", g, hipGetErrorString(e));
    }
    p->peer_inbox[g] = static_cast<unsigned long long*>(ptr);
    p->peer_mapped[g] = true;
  }
  TRY(p2p_alloc_fault(p));  // (and cleared)
  p->p2p_on = true;
  p->p2p_index = 0;
  return MPPI_OK;
}

// one process, several devices: the handles' inboxes addressed directly (peer access)
extern "C" int mppi_group_p2p_connect(mppi_planner** ps, int count) {
  REQUIRE(ps && count >= 1 && count <= kMaxFoldedRanks, MPPI_ERR_INVALID, "bad group of %d", count);
  bool several_devices = false;
  for (int g = 0; g < count; ++g) {
    REQUIRE(ps[g] && ps[g]->cfg.world_size == count && ps[g]->cfg.rank == g, MPPI_ERR_INVALID, "planner %d is not rank %d of %d", g, g, count);
    REQUIRE(ps[g]->m_count == 1 && ps[g]->cfg.mode == MPPI_MODE_DET && ps[g]->B == 1, MPPI_ERR_INVALID,
            "the peer exchange serves single-problem deterministic-dynamics handles with all their traction samples");
    several_devices = several_devices || ps[g]->cfg.device != ps[0]->cfg.device;
  }
  // peer access first: memory allocated afterwards is mapped for the peers that have it enabled
  for (int g = 0; g < count; ++g) {
    HIP_TRY(hipSetDevice(ps[g]->cfg.device));
    for (int q = 0; q < count; ++q) {
      if (ps[q]->cfg.device == ps[g]->cfg.device) continue;
      int can = 0;
      HIP_TRY(hipDeviceCanAccessPeer(&can, ps[g]->cfg.device, ps[q]->cfg.device));
      REQUIRE(can, MPPI_ERR_HIP, "device %d cannot access device %d", ps[g]->cfg.device, ps[q]->cfg.device);
      hipError_t e = hipDeviceEnablePeerAccess(ps[q]->cfg.device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e);
      (void)hipGetLastError();
    }
  }
  for (int g = 0; g < count; ++g) {
    HIP_TRY(hipSetDevice(ps[g]->cfg.device));
    TRY(p2p_alloc_inbox(ps[g], nullptr));
    TRY(p2p_alloc_fault(ps[g]));
    // (coarse-grained memory is coherent at kernel boundaries only: good enough for ranks that share ONE device, the
    //  test set-up; across devices a running kernel would never see its peers' stores)
    REQUIRE(!several_devices || strcmp(ps[g]->inbox_kind, "coarse-grained") != 0, MPPI_ERR_HIP,
            "device %d: no fine-grained memory for the peer exchange's inbox", ps[g]->cfg.device);
  }
  for (int g = 0; g < count; ++g) {
    mppi_planner* p = ps[g];
    HIP_TRY(hipSetDevice(p->cfg.device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p2p_disconnect(p);
    drop_graphs(p);
    for (int q = 0; q < count; ++q) p->peer_inbox[q] = ps[q]->inbox;
    p->p2p_on = true;
    p->p2p_index = 0;
  }
  // Do the peers' stores reach kernels that are already running?  All ranks ping at once (their kernels wait for each
  // other, so every device gets its launch before any is waited for); a group that cannot hear itself is not connected.
  // Ranks that SHARE a device (the one-GPU test set-up) are not pinged: their planners' streams can map onto one hardware
  // queue and run one after the other (mppi_debug_occupy_cus records exactly that), the first ping would then spin until
  // its poll limit and report a connection that works as broken; through one device's L2 there is nothing to find out.
  bool shared_device = false;
  for (int g = 0; g < count; ++g)
    for (int q = 0; q < g; ++q) shared_device = shared_device || ps[q]->cfg.device == ps[g]->cfg.device;
  if (!shared_device) {
    std::vector<int*> results((size_t)count, nullptr);
    int rc = MPPI_OK;
    int launched = 0;
    for (int g = 0; g < count && rc == MPPI_OK; ++g) {
      mppi_planner* p = ps[g];
      if (hipSetDevice(p->cfg.device) != hipSuccess) { rc = fail(MPPI_ERR_HIP, "hipSetDevice failed"); break; }
      rc = dev_alloc(&results[(size_t)g], 1);
      if (rc != MPPI_OK) break;
      PeerExchange X;
      memset(&X, 0, sizeof(X));
      for (int q = 0; q < count; ++q) X.inbox[q] = p->peer_inbox[q];
      X.world = count;
      X.rank = g;
      hipLaunchKernelGGL(k_p2p_ping, dim3(1), dim3(64), 0, p->stream, X, inbox_ping_offset(count, p->cfg.num_steps),
                         0x70696e67ull /* "ping" */, 2000 * 1000, results[(size_t)g]);
      launched = g + 1;
    }
    for (int g = 0; g < count; ++g) {
      mppi_planner* p = ps[g];
      int heard = 0;
      // (also on the error path: a ping that was launched still polls -- bounded -- and writes its result buffer)
      if (g < launched) {
        (void)hipSetDevice(p->cfg.device);
        const bool drained = hipStreamSynchronize(p->stream) == hipSuccess;
        if (rc == MPPI_OK) {
          if (!drained || hipMemcpy(&heard, results[(size_t)g], sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(MPPI_ERR_HIP, "peer exchange ping on device %d failed", p->cfg.device);
          else if (heard != count)
            rc = fail(MPPI_ERR_COMM, "peer exchange: device %d heard %d of %d ranks (stores of peers do not reach running kernels)",
                      p->cfg.device, heard, count);
        }
      }
      dev_free(results[(size_t)g]);
    }
    if (rc != MPPI_OK) {
      for (int g = 0; g < count; ++g) p2p_disconnect(ps[g]);
      return rc;
    }
    // (the ping words hold the token now; a later mppi_planner_p2p_ping uses another one)
  }
  return MPPI_OK;
}

// All ranks together, right after connecting (and a host barrier): do the peers' stores reach this rank's running
// kernels?  *heard = ranks whose token arrived within ~timeout_ms (world_size: the exchange can be trusted).
extern "C" int mppi_planner_p2p_ping(mppi_planner* p, unsigned long long token, int timeout_ms, int* heard) {
  REQUIRE(p && heard, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(p->inbox && p->peer_inbox[p->cfg.rank], MPPI_ERR_STATE, "the peer exchange is not connected");
  REQUIRE(token != 0ull && token != ~0ull, MPPI_ERR_INVALID, "tokens 0 and all-ones are what empty slots hold");
  HIP_TRY(hipSetDevice(p->cfg.device));
  int* result = nullptr;
  TRY(dev_alloc(&result, 1));
  PeerExchange X;
  memset(&X, 0, sizeof(X));
  for (int g = 0; g < p->cfg.world_size; ++g) X.inbox[g] = p->peer_inbox[g];
  X.world = p->cfg.world_size;
  X.rank = p->cfg.rank;
  const int max_polls = std::max(1, timeout_ms) * 1000;  // (~1 us per poll: a sleep of 16 x 64 cycles + the load)
  hipLaunchKernelGGL(k_p2p_ping, dim3(1), dim3(64), 0, p->stream, X, inbox_ping_offset(p->cfg.world_size, p->cfg.num_steps),
                     token, max_polls, result);
  int host = 0;
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(&host, result, sizeof(int), hipMemcpyDeviceToHost, p->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
  (void)hipFree(result);
  HIP_TRY(e);
  *heard = host;
  return MPPI_OK;
}

// (measurements: the same handle with the exchange switched off falls back to its communicator; all ranks alike)
extern "C" int mppi_planner_p2p_set_enabled(mppi_planner* p, int enabled) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  REQUIRE(!enabled || (p->inbox && p->peer_inbox[p->cfg.rank]), MPPI_ERR_STATE, "the peer exchange is not connected");
  HIP_TRY(hipSetDevice(p->cfg.device));
  HIP_TRY(hipStreamSynchronize(p->stream));
  drop_graphs(p);
  // (switching the exchange off after MPPI_ERR_COMM leaves a handle that works over RCCL or the host again -- with
  // whatever u the broken call left, so set u before iterating; switching it ON again needs a new connect)
  if (!enabled && p->p2p_fault_host) *p->p2p_fault_host = 0u;
  REQUIRE(!enabled || !p->p2p_fault_host || *p->p2p_fault_host == 0u, MPPI_ERR_STATE, "the peer exchange failed: connect again first");
  p->p2p_on = enabled != 0;
  return MPPI_OK;
}

extern "C" int mppi_planner_p2p_stats(mppi_planner* p, int* connected, long* exchanges, char* kind, int capacity) {
  REQUIRE(p && connected && exchanges, MPPI_ERR_INVALID, "NULL argument");
  *connected = p->p2p_on ? 1 : 0;
  *exchanges = (long)p->p2p_exchanges;
  if (kind && capacity > 0) snprintf(kind, (size_t)capacity, "%s", p->inbox_kind);
  return MPPI_OK;
}

extern "C" int mppi_planner_comm_init(mppi_planner* p, const char id[MPPI_COMM_ID_BYTES]) {
  REQUIRE(p && id, MPPI_ERR_INVALID, "NULL argument");
  REQUIRE(!p->comm, MPPI_ERR_STATE, "communicator already initialised");
  TRY(rccl_load());
  HIP_TRY(hipSetDevice(p->cfg.device));
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  // (one communicator per handle: over the ranks that share its control samples, or its traction samples)
  if (p->m_count > 1) RCCL_TRY(g_rccl.CommInitRank(&p->comm, p->m_count, uid, p->m_rank));
  else RCCL_TRY(g_rccl.CommInitRank(&p->comm, p->cfg.world_size, uid, p->cfg.rank));
  return MPPI_OK;
}
