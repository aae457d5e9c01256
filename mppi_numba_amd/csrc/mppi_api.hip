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
#include <string>
#include <vector>

#include "rng_kernels.h"
#include "rollout_kernels.h"
#include "rollout_spec_kernel.h"
#include "rollout_deep_kernel.h"
#include "rollout_scan_kernel.h"
#include "map_kernels.h"
#include "update_kernels.h"
#include "world_kernels.h"

using namespace mppi;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(MPPI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                  __LINE__);                                                                  \
  } while (0)

#define REQUIRE(cond, code, ...)             \
  do {                                       \
    if (!(cond)) return fail(code, __VA_ARGS__); \
  } while (0)

#define TRY(expr)           \
  do {                      \
    int _rc = (expr);       \
    if (_rc != MPPI_OK) return _rc; \
  } while (0)

extern "C" const char* mppi_last_error(void) { return g_last_error.c_str(); }
extern "C" int mppi_abi_version(void) { return MPPI_HIP_ABI_VERSION; }

template <typename T>
static int dev_alloc(T** p, size_t count) {
  *p = nullptr;
  if (count == 0) count = 1;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
  return MPPI_OK;
}
template <typename T>
static void dev_free(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

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

// ---------------------------------------------------------------------------
// TDM
// ---------------------------------------------------------------------------
struct mppi_tdm {
  mppi_tdm_cfg cfg;
  hipStream_t stream = nullptr;
  int8_t* grid = nullptr;  // [G][max_rows][max_cols] int8 (reference layout)
  int8_t* pmf = nullptr;   // [B][rows][cols]
  size_t pmf_capacity = 0;
  int8_t* table = nullptr;  // [B] bin -> int8 traction
  int table_capacity = 0;
  int8_t* obs = nullptr;  // [rows][cols]
  int8_t* unk = nullptr;
  int8_t* risk = nullptr;
  size_t map_capacity = 0;
  // staging of the raw inputs of mppi_tdm_set_maps_from_pmf (device-side preprocessing)
  int8_t* raw = nullptr;  // raw PMF | raw obstacle | raw unknown
  size_t raw_capacity = 0;
  float* bin_values = nullptr;
  int bin_values_capacity = 0;
  int* prep_flags = nullptr;
  uint64_t* states = nullptr;  // xoroshiro-compatible generator only
  long n_states = 0;
  int bins = 0, rows = 0, cols = 0;
  bool has_risk = false, maps_set = false, one_hot = false;
  bool compact_ok = false;  // masks are 0/1 and every traction byte is in [0,127]: 16-bit cells usable
  int table_max = 127;      // largest traction byte the sampler can write
  double lo = 0.0, ratio = 0.0;
  uint64_t epoch = 0;         // Philox call counter
  uint64_t maps_version = 0;  // bumped by set_maps
  uint64_t grid_version = 0;  // bumped whenever `grid` changes
  uint64_t sampled_maps_version = ~0ULL;
  double sampled_alpha = -1.0;
  // solve() of a CVaR planner samples straight into the planner's cell words (Philox only):
  // the int8 grids are then produced on demand from the same counters
  bool grid_stale = false;      // `grid` does not hold the draws of (sampled_epoch, sampled_alpha) yet
  uint64_t sampled_epoch = 0;   // Philox epoch of the current draws
  bool injected = false;  // grids came from mppi_tdm_set_sampled_grids
  int8_t injected_max = 0, injected_min = 0;
  // samples sharded over GPUs (mppi_tdm_set_sample_shard): this handle's G grids are samples
  // [first_sample, first_sample + G) of the unsharded set; even, a Philox block serves a pair
  int first_sample = 0;
};

extern "C" int mppi_tdm_destroy(mppi_tdm* t) {
  if (!t) return MPPI_OK;
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
  t->one_hot = one_hot;
  t->bins = bins;
  t->rows = rows;
  t->cols = cols;
  t->has_risk = risk != nullptr;
  t->lo = traction_lo;
  t->ratio = traction_ratio;
  t->maps_set = true;
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
  t->one_hot = (kind != PREP_TDM) || flags[1] == 0;
  t->bins = bins;
  t->rows = rows;
  t->cols = cols;
  t->has_risk = kind == PREP_SPEED;
  t->lo = traction_lo;
  t->ratio = traction_ratio;
  t->maps_set = true;
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

// ---------------------------------------------------------------------------
// roctx ranges (tracing hook): MPPI_ROCTX=1 marks, on the host thread that
// enqueues them, solve / sample_grids / noise / rollout / exchange / update / closed_loop, for
// `rocprofv3 --marker-trace --kernel-trace`.  The library is looked up at run time; without the
// variable (or the library) a range costs one predictable branch.
// ---------------------------------------------------------------------------
struct RoctxApi {
  bool tried = false;
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
static RoctxApi g_roctx;

static void roctx_load() {
  g_roctx.tried = true;
  const char* on = getenv("MPPI_ROCTX");
  if (!on || !*on || *on == '0') return;
  const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so",
                         "libroctx64.so.4", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "/opt/rocm/lib/libroctx64.so"};
  for (const char* n : names) {
    void* h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!h) continue;
    g_roctx.push = (decltype(g_roctx.push))dlsym(h, "roctxRangePushA");
    g_roctx.pop = (decltype(g_roctx.pop))dlsym(h, "roctxRangePop");
    if (g_roctx.push && g_roctx.pop) return;
    g_roctx.push = nullptr;
    g_roctx.pop = nullptr;
  }
}

struct TraceRange {
  bool open = false;
  explicit TraceRange(const char* name) {
    if (!g_roctx.tried) roctx_load();
    if (g_roctx.push) { g_roctx.push(name); open = true; }
  }
  ~TraceRange() {
    if (open) g_roctx.pop();
  }
  TraceRange(const TraceRange&) = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};

extern "C" int mppi_trace_ranges_enabled(void) {
  if (!g_roctx.tried) roctx_load();
  return g_roctx.push ? 1 : 0;
}

// ---------------------------------------------------------------------------
// RCCL, loaded on first use so that single-GPU users never touch it
// ---------------------------------------------------------------------------
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static int rccl_load() {
  if (g_rccl.handle) return MPPI_OK;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  REQUIRE(h, MPPI_ERR_COMM, "cannot load librccl.so: %s", dlerror());
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
  g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(h, "ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(h, "ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  REQUIRE(g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather &&
              g_rccl.GetErrorString && g_rccl.CommCount && g_rccl.GroupStart && g_rccl.GroupEnd,
          MPPI_ERR_COMM, "librccl.so lacks expected symbols");
  g_rccl.handle = h;
  return MPPI_OK;
}

#define RCCL_TRY(expr)                                                                  \
  do {                                                                                  \
    ncclResult_t _r = (expr);                                                           \
    if (_r != ncclSuccess)                                                              \
      return fail(MPPI_ERR_COMM, "%s failed: %s", #expr, g_rccl.GetErrorString(_r));   \
  } while (0)

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

// ---------------------------------------------------------------------------
// planner
// ---------------------------------------------------------------------------
struct mppi_planner {
  mppi_planner_cfg cfg;
  hipStream_t stream = nullptr;
  int n_local = 0, n_offset = 0;
  // batched multi-query: B problems, each n_inst rollouts (inst_tiles tiles of 64) on this GPU;
  // n_local = B * n_inst.  Per-problem start / goal / window origin live in inst_dev.
  int B = 1, n_inst = 0, inst_tiles = 0;
  std::vector<BatchInst> inst_host;
  BatchInst* inst_dev = nullptr;
  bool inst_set = false, inst_dirty = false;
  // pinned, device-mapped (B,T) mirror of u: the update kernels write it (u_host_dev is the device
  // view of the same memory), solve() reads it after the stream has drained -- no copy on the hot path
  float2* u_host = nullptr;
  float2* u_host_dev = nullptr;
  // Only solve() reads the mirror, after its LAST iteration: the update launches of every other
  // iteration are spared the posted write across PCIe (their completion waits for it).
  bool mirror_now = false;   // the coming update launch writes the mirror
  bool mirror_done = false;  // ... the last one did
  // set_u(): pinned staging + asynchronous copy; the next set_u waits for the previous copy only
  float2* u_stage = nullptr;
  hipEvent_t ev_u_staged = nullptr;
  bool u_stage_busy = false;
  // device buffers
  float2* noise = nullptr;    // tile-major (n_local, T): the buffer the NEXT rollout/update reads
  float2* noise_buf[2] = {nullptr, nullptr};  // double buffer: noise of iteration k+1 is generated
  int noise_cur = 0;                           // while iteration k runs (in-launch or on noise_stream)
  // throughput regime (no idle CU for in-launch generation): the noise of iteration k+1 runs beside
  // rollout k on a second stream while the rollout leaves wave slots free (run_iterations)
  hipStream_t noise_stream = nullptr;
  hipEvent_t ev_buf_free = nullptr, ev_noise_ready = nullptr;
  // hipGraph replay of the iteration loop (mppi_planner_set_graph_replay): two iterations
  // (one round of the noise double buffer) captured once, replayed while nothing a kernel
  // argument carries has changed.  See run_iterations.
  bool graph_on = false;
  int graph_chunk = 2;                    // iterations per captured graph
  unsigned long long* gen_dev = nullptr;  // device: update kernels executed since graph mode was enabled
  uint64_t bumps_launched = 0;            // host mirror of *gen_dev once the stream has drained
  bool primed = false;                    // noise_buf[noise_cur ^ 1] already holds the NEXT iteration's noise
  bool graph_warm = false;                // one direct iteration has run since graph mode was enabled
  // one cached graph per parity of the noise double buffer (a call with an odd number of
  // iterations leaves the other parity behind)
  hipGraph_t graph[2] = {nullptr, nullptr};
  hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
  std::vector<unsigned char> graph_sig[2];  // everything the captured launches took by value
  uint64_t graph_spec_tiles[2] = {0, 0};    // speculative tiles one replay of the graph launches
  long graph_replays = 0, graph_captures = 0;
  std::string last_rollout;        // which rollout kernel variant the last launch used (diagnostic)
  int debug_flags = 0;             // mppi_planner_set_debug_flags (tests pin every kernel variant through it)
  bool next_noise_wanted = false;  // the coming rollout launch should also generate noise_buf[cur^1]
  bool next_noise_done = false;    // ... and it did
  bool noise_on_side_stream = false;  // the noise produced ahead is still in flight on noise_stream
  float2* staging = nullptr;  // (n_local,T) host-layout staging for set/get_noise
  float2* u = nullptr;        // [T]
  float2* u_prev = nullptr;   // [T]
  float* costs = nullptr;     // [n_local]
  float* weights_out = nullptr;  // [n_local] normalised weights, filled on request
  float* w_rel = nullptr;      // [n_local] exp(-(c - beta_tile)/lambda)
  float* tile_beta = nullptr;  // [n_tiles] minimum cost of each tile of 64 rollouts
  int n_tiles = 0;
  bool tile_packets_fresh = false;  // w_rel / tile_beta written by the rollout kernel for the current costs
  // k_rollout_scan (MPPI_MATH_FAST): per-tile sums of w_rel * noise, consumed by k_combine_tiles
  float2* tnum = nullptr;  // [T][tiles]
  float* tden = nullptr;   // [tiles]
  float* tbeta = nullptr;  // [tiles] minimum cost of each of the kernel's tiles (32 or 64 rollouts)
  int scan_tile = 32;      // rollouts per tile of the last such launch
  bool scan_packets_fresh = false;  // ... written by the last rollout launch for the current costs
  // the iteration loop of such a handle generates the noise INSIDE the rollout launch (Philox counter
  // blocks, never stored): noise_buf is then stale, and whoever wants the noise of the last iteration
  // (get_noise, get_state_rollout, a stage-level update) has it regenerated from the same counters
  bool scan_gen_now = false;   // the coming rollout launch is to generate its own noise
  bool noise_virtual = false;  // the noise of the last iteration exists as counters only ...
  int noise_virtual_back = 1;  // ... of Philox epoch noise_epoch - noise_virtual_back
  double* packets = nullptr;  // [world][2+2T]; own packet at [rank]
  double* stats = nullptr;    // {beta, den} of the last update
  uint32_t* cells = nullptr;
  size_t cells_capacity = 0;
  double* cc_scratch = nullptr;  // [T][n_local] control-cost products of the pipelined rollout
  uint16_t* cells16 = nullptr;  // 16-bit cells, row pitch multiple of 8 (LDS window source)
  size_t cells16_capacity = 0;
  int pitch16 = 0;
  bool cells16_valid = false;
  bool cells16_with_risk = false;  // cells16 holds 32-bit cells with the risk byte (speed-map mode)
  int num_cus = 256;
  int lds_per_cu = 160 * 1024;
  int8_t* risk_ref = nullptr;
  float* sample_costs = nullptr;  // [n_local][M], allocated on first request
  bool want_sample_costs = false;
  uint64_t* states = nullptr;  // xoroshiro-compatible generator only
  long n_states = 0;
  float2* obs_pos = nullptr;
  float* obs_r = nullptr;
  int n_obstacles = 0;
  float* state_rollout = nullptr;  // [V][T+1][3]
  // host state
  mppi_params params;
  bool params_set = false;
  uint64_t noise_epoch = 0;
  const mppi_tdm* packed_lin = nullptr;
  const mppi_tdm* packed_ang = nullptr;
  uint64_t packed_lin_grid = ~0ULL, packed_ang_grid = ~0ULL, packed_lin_maps = ~0ULL;
  // timing
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_stage[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool profile_stages = false;
  float stage_ms[4] = {0, 0, 0, 0};
  float last_elapsed_ms = 0.f;
  bool elapsed_pending = false;
  int last_iterations = 0;
  // Speculative rollout kernels (k_rollout_deep / k_rollout_spec) on a map where the traction
  // changes from cell to cell: every tile fails its vote and re-runs on the exact schedule, slower
  // than launching k_rollout_pipe in the first place (N = 8192, T = 200 over a CVaR-bin map: 85 vs
  // 49 us).  The kernels count failed tiles in a host-mapped word; whenever the host has
  // synchronised anyway it compares that with the tiles launched and, past one half, stops
  // speculating until the packed map changes.
  unsigned int* spec_fail_host = nullptr;  // pinned, device-mapped
  unsigned int* spec_fail_dev = nullptr;   // device view of the same word
  uint64_t spec_tiles_launched = 0;
  bool speculation_off = false;
  // mppi_planner_time_kernels: dispatch begin / end of the rollout and update launches of the
  // iterations it runs (4 events per iteration), picked up by MPPI_KLAUNCH
  hipEvent_t kev_start = nullptr, kev_stop = nullptr;
  std::vector<hipEvent_t> ktime_events;
  int ktime_index = -1;
  // comm
  ncclComm_t comm = nullptr;
  // CVaR mode with the M traction samples sharded over GPUs (mppi_planner_set_sample_sharding):
  // this handle rolls ALL N control samples over its cfg.num_grid_samples grids; the per-(n, m)
  // costs of all shards are all-gathered and every rank forms the CVaR of every control sample
  int m_rank = 0, m_count = 1;
  // a stage-level rollout of such a handle leaves the CVaR over the LOCAL samples in costs: the update
  // must not run before the slabs of all shards have been reduced (launch_cvar_reduce)
  bool sample_costs_local_only = false;
  float* slabs = nullptr;  // [m_count][n_local][M_local]
  // closed loop on the device (mppi_planner_closed_loop): world state, trajectory log
  double* loop_state = nullptr;   // [B][3]
  double* loop_xhist = nullptr;   // [B][loop_capacity + 1][3]
  float2* loop_uhist = nullptr;   // [B][loop_capacity]
  int* loop_done = nullptr;       // [B]
  int* loop_done_count = nullptr;      // pinned, device-mapped
  int* loop_done_count_dev = nullptr;  // device view of the same int
  int loop_capacity = 0;
};

static void drop_graphs(mppi_planner* p) {
  for (int i = 0; i < 2; ++i) {
    if (p->graph_exec[i]) (void)hipGraphExecDestroy(p->graph_exec[i]);
    if (p->graph[i]) (void)hipGraphDestroy(p->graph[i]);
    p->graph_exec[i] = nullptr;
    p->graph[i] = nullptr;
    p->graph_sig[i].clear();
  }
}

extern "C" int mppi_planner_destroy(mppi_planner* p) {
  if (!p) return MPPI_OK;
  (void)hipSetDevice(p->cfg.device);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(p->comm);
  dev_free(p->inst_dev);
  if (p->u_host) (void)hipHostFree(p->u_host);
  if (p->u_stage) (void)hipHostFree(p->u_stage);
  if (p->ev_u_staged) (void)hipEventDestroy(p->ev_u_staged);
  dev_free(p->noise_buf[0]);
  dev_free(p->noise_buf[1]);
  dev_free(p->staging);
  dev_free(p->u);
  dev_free(p->u_prev);
  dev_free(p->costs);
  dev_free(p->weights_out);
  dev_free(p->w_rel);
  dev_free(p->tile_beta);
  dev_free(p->tnum);
  dev_free(p->tden);
  dev_free(p->tbeta);
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
  if (p->spec_fail_host) (void)hipHostFree(p->spec_fail_host);
  dev_free(p->loop_state);
  dev_free(p->loop_xhist);
  dev_free(p->loop_uhist);
  dev_free(p->loop_done);
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
  HIP_TRY(hipEventCreate(&p->ev_begin));
  HIP_TRY(hipEventCreate(&p->ev_end));
  for (auto& e : p->ev_stage) HIP_TRY(hipEventCreate(&e));
  const size_t n_tiled = (size_t)ceil_div((long)N, 64) * 64;  // tile-major arrays cover whole tiles
  // (+ 8 chunks of 8 rows: k_rollout_spec prefetches rows past the horizon of the last tile unclamped)
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
  TRY(dev_alloc(&p->stats, 2 * B));
  TRY(dev_alloc(&p->state_rollout, (size_t)c.num_vis_state_rollouts * (T + 1) * 3));
  HIP_TRY(hipMemsetAsync(p->u, 0, B * T * sizeof(float2), p->stream));  // u_seq0 = zeros (mppi.py:93)
  HIP_TRY(hipMemsetAsync(p->u_prev, 0, B * T * sizeof(float2), p->stream));
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
  HIP_TRY(hipSetDevice(p->cfg.device));
  HIP_TRY(hipStreamSynchronize(p->stream));
  dev_free(p->obs_pos);
  dev_free(p->obs_r);
  p->n_obstacles = count;
  if (count > 0) {
    TRY(dev_alloc(&p->obs_pos, (size_t)count));
    TRY(dev_alloc(&p->obs_r, (size_t)count));
    HIP_TRY(hipMemcpy(p->obs_pos, positions, sizeof(float2) * (size_t)count, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(p->obs_r, radii, sizeof(float) * (size_t)count, hipMemcpyHostToDevice));
  }
  return MPPI_OK;
}

static int copy_out(mppi_planner* p, void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipSetDevice(p->cfg.device));
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return MPPI_OK;
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
  // (either way the rollouts spread at most step_cells per step: k_rollout_spec copies the window in
  //  bands of rows as they go)
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
#define MPPI_KLAUNCH(kernel, grid, block, lds, stream, ...) \
  hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, p->kev_start, p->kev_stop, 0, __VA_ARGS__)


// ---- k_rollout_scan (rollout_scan_kernel.h): the time-parallel rollout of MPPI_MATH_FAST --------
// Eligible: deterministic-dynamics mode, 16-bit cells (the reference's own maps always are), a horizon
// of at most 16 waves of 8 steps, LDS for the per-step records, and a map the speculation pays on.
struct ScanPlan {
  int waves = 0;       // 8 steps each = waves per workgroup
  int tile = 32;       // rollouts per workgroup: 32 (two lanes per rollout) or 64
  size_t lds = 0;
  bool pow2res = false;
};

static bool scan_plan(const mppi_planner* p, ScanPlan* out) {
  static const bool disabled = getenv("MPPI_NO_SCAN") != nullptr;  // developer switch (ablation)
  if (disabled || (p->debug_flags & MPPI_DEBUG_NO_SCAN_KERNEL)) return false;
  if (p->cfg.math != MPPI_MATH_FAST || p->cfg.mode != MPPI_MODE_DET) return false;
  if (!p->cells16_valid || p->cells16_with_risk) return false;
  if (p->speculation_off && !(p->debug_flags & MPPI_DEBUG_KEEP_SPECULATING)) return false;
  const int T = p->cfg.num_steps;
  ScanPlan plan;
  plan.waves = ceil_div(T, 8);
  if (plan.waves > 16) return false;
  plan.tile = (p->debug_flags & MPPI_DEBUG_SCAN_FULL_TILES) ? 64 : 32;
  plan.lds = plan.tile == 64 ? ScanLds<64>::total(plan.waves) : ScanLds<32>::total(plan.waves);
  // (the accumulating wave reads up to two groups of records past the last one: keep that inside the allocation)
  plan.lds = std::max(plan.lds, (size_t)40 * 1024);
  if (plan.lds > (size_t)p->lds_per_cu - 1024) return false;
  int res_exp = 0;
  plan.pow2res = std::frexp((double)p->params.res, &res_exp) == 0.5;  // res == 2^k exactly
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

static int launch_scan(mppi_planner* p, const DevParams& d, const ScanPlan& plan) {
  const int N = p->n_local, T = p->cfg.num_steps;
  const int tiles = ceil_div(N, plan.tile);
  if (!p->tnum) {  // (sized for the smaller tile)
    const size_t cap = (size_t)ceil_div(N, 32);
    TRY(dev_alloc(&p->tnum, (size_t)T * cap));
    TRY(dev_alloc(&p->tden, cap));
    TRY(dev_alloc(&p->tbeta, cap));
  }
  const bool gen = p->scan_gen_now;
  ScanPackets pk;
  pk.tnum = p->tnum;
  pk.tden = p->tden;
  pk.tbeta = p->tbeta;
  pk.n_tiles = tiles;
  NoiseJob gen_job, next_job;
  memset(&gen_job, 0, sizeof(gen_job));
  memset(&next_job, 0, sizeof(next_job));
  int extra = 0;
  if (gen) {
    gen_job = make_noise_job(p, nullptr);  // (advances the Philox epoch: this iteration's block)
  } else {
    // a loop that stores its noise (debug switch; the stage-level calls): CUs without a workgroup
    // produce the next iteration's, as in k_rollout_deep.  (Producing it in the launch's own tail, by
    // the waves that idle while one wave accumulates the costs, was measured: the stage gained is lost
    // again to the slower accumulation and the noise reads -- profiles/r03_scan_notes.md.)
    static const bool no_fused_noise = getenv("MPPI_NO_FUSED_NOISE") != nullptr;  // developer switch
    if (p->next_noise_wanted && tiles < p->num_cus && !no_fused_noise && p->cfg.rng == MPPI_RNG_PHILOX) {
      extra = p->num_cus - tiles;
      next_job = make_noise_job(p, p->noise_buf[p->noise_cur ^ 1]);
      p->next_noise_done = true;
    }
  }
  p->spec_tiles_launched += (uint64_t)tiles;
#define MPPI_LAUNCH_SCAN(RR, P2, GEN)                                                                      \
  do {                                                                                                    \
    auto kern = k_rollout_scan<RR, P2, GEN>;                                                              \
    if (plan.lds > 64 * 1024)                                                                             \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan.lds));            \
    MPPI_KLAUNCH(kern, dim3(tiles + extra), dim3(64 * plan.waves), plan.lds, p->stream, d, p->cells16,    \
                 p->noise, gen_job, p->u, p->costs, p->w_rel, pk, tiles, next_job);                        \
  } while (0)
#define MPPI_LAUNCH_SCAN_G(RR, P2)               \
  do {                                           \
    if (gen) MPPI_LAUNCH_SCAN(RR, P2, true);     \
    else MPPI_LAUNCH_SCAN(RR, P2, false);        \
  } while (0)
  if (plan.tile == 64 && plan.pow2res) MPPI_LAUNCH_SCAN_G(64, true);
  else if (plan.tile == 64) MPPI_LAUNCH_SCAN_G(64, false);
  else if (plan.pow2res) MPPI_LAUNCH_SCAN_G(32, true);
  else MPPI_LAUNCH_SCAN_G(32, false);
#undef MPPI_LAUNCH_SCAN_G
#undef MPPI_LAUNCH_SCAN
  HIP_TRY(hipGetLastError());
  char buf[256];
  snprintf(buf, sizeof(buf),
           "k_rollout_scan tile=%d waves=%d pow2res=%d noise=%s lds=%zu noise_blocks=%d problems=%d",
           plan.tile, plan.waves, (int)plan.pow2res, gen ? "in-kernel" : "read", plan.lds, extra, p->inst_set ? p->B : 0);
  p->last_rollout = buf;
  p->tile_packets_fresh = false;  // (w_rel is relative to this kernel's own tiles: tbeta, not tile_beta)
  p->scan_packets_fresh = true;
  p->scan_tile = plan.tile;
  p->noise_virtual = gen;
  p->noise_virtual_back = 1 + (next_job.out ? 1 : 0);  // (the launch may have produced its successor's block too)
  return MPPI_OK;
}

template <bool EXACT, bool BOUNDED>
static int launch_rollout_t(mppi_planner* p, DevParams d) {
  const int N = p->n_local, T = p->cfg.num_steps, M = p->cfg.num_grid_samples;
  size_t lds = sizeof(double2) * (size_t)T;
  const size_t lds_map = sizeof(double2) * ((size_t)T + (size_t)(T + 1) / 2);  // + staged u[t]
  p->scan_packets_fresh = false;
  if (p->noise_virtual && !p->scan_gen_now) TRY(materialize_noise(p));  // the coming kernel reads its noise
  switch (p->cfg.mode) {
    case MPPI_MODE_DET: {
      p->tile_packets_fresh = false;
      size_t lds_win = 0;
      bool have_window = plan_lds_window(p, d, &lds_win);
      TRY(upload_instances(p));
      if (!EXACT) {
        ScanPlan plan;
        if (scan_plan(p, &plan)) return launch_scan(p, d, plan);
      }
      static const bool no_pipe = getenv("MPPI_NO_PIPE") != nullptr;  // developer switch (ablation)
      // incremental trig: needs a heading increment |dt*w*traction| <= 0.36 rad and T <= 2000
      bool rot_ok = false, pow2res = false;
      {
        const mppi_params& a = p->params;
        double wmax = std::fmax(std::fabs((double)a.wrange[0]), std::fabs((double)a.wrange[1]));
        double trmax = std::fmax(std::fabs(d.ang_lo), std::fabs(d.ang_lo + (double)d.ang_max_byte * d.ang_ratio));
        double dmax = (double)a.dt * wmax * trmax;
        rot_ok = EXACT && BOUNDED && std::isfinite(dmax) && dmax <= 0.36 && T <= 2000;
        int res_exp = 0;
        pow2res = std::frexp((double)a.res, &res_exp) == 0.5;  // res == 2^k exactly
      }
      static const bool no_deep = getenv("MPPI_NO_DEEP") != nullptr;  // developer switch (ablation)
      // MPPI_MATH_FAST: the same five-stage pipeline in float32 (hardware sin / cos: |theta| must stay
      // inside v_sin_f32's +-256 revolutions)
      bool fast_deep_ok = false;
      if (!EXACT) {
        const mppi_params& a = p->params;
        double wmax = std::fmax(std::fabs((double)a.wrange[0]), std::fabs((double)a.wrange[1]));
        double trmax = std::fmax(std::fabs(d.ang_lo), std::fabs(d.ang_lo + (double)d.ang_max_byte * d.ang_ratio));
        double th0_max = std::fabs((double)a.x0[2]);
        if (p->inst_set) for (const BatchInst& I : p->inst_host) th0_max = std::fmax(th0_max, std::fabs((double)I.th0));
        const double bound = th0_max + (double)T * (double)a.dt * wmax * trmax;
        fast_deep_ok = std::isfinite(bound) && bound < 1500.0 && T <= 2000;
      }
      const bool keep_speculating = !p->speculation_off || (p->debug_flags & MPPI_DEBUG_KEEP_SPECULATING);
      if (have_window && (EXACT ? rot_ok : fast_deep_ok) && !no_pipe && !no_deep && keep_speculating &&
          !(p->debug_flags & (MPPI_DEBUG_NO_SPEC_KERNEL | MPPI_DEBUG_NO_DEEP_KERNEL)) &&
          ceil_div(N, 64) <= p->num_cus) {
        // five-stage speculative pipeline, one tile per CU (rollout_deep_kernel.h)
        const size_t map_bytes = lds_win - sizeof(double2) * ((size_t)T + (size_t)(T + 1) / 2);
        const size_t Tp = ((size_t)T + 7) & ~(size_t)7;
        const size_t head = sizeof(double2) * (Tp + Tp / 2);
        const size_t budget = (size_t)p->lds_per_cu - 1024;
        const size_t cc_bytes = Tp * 64 * sizeof(double);
        // the exact re-execution path (pipe_tile_body<8>) lives in the same allocation
        const size_t exact_need = lds_win + (size_t)PipeRing<8>::kBytesPerPair;
        int chunk = 0;
        auto ring_size = [](int c) {
          return c == 8 ? (size_t)DeepRing<8>::kBytes : c == 4 ? (size_t)DeepRing<4>::kBytes : (size_t)DeepRing<2>::kBytes;
        };
        static const int forced_chunk = getenv("MPPI_DEEP_CHUNK") ? atoi(getenv("MPPI_DEEP_CHUNK")) : 0;  // developer switch
        for (int cnd : {8, 4, 2})
          if (head + map_bytes + ring_size(cnd) <= budget && (!forced_chunk || cnd <= forced_chunk)) { chunk = cnd; break; }
        // (chunks of 8 with the control-cost products in LDS, else of 4 with them in LDS, else as found)
        if (chunk == 8 && head + map_bytes + ring_size(8) + cc_bytes > budget &&
            head + map_bytes + ring_size(4) + cc_bytes <= budget)
          chunk = 4;
        if (chunk > 0 && exact_need <= budget) {
          const size_t rings = ring_size(chunk);
          const size_t spec_need = head + map_bytes + rings;
          const bool cc_lds = spec_need + cc_bytes <= budget && exact_need + (size_t)T * 64 * sizeof(double) <= budget &&
                              !(p->debug_flags & MPPI_DEBUG_CC_GLOBAL);
          const size_t lds_total = std::max(spec_need + (cc_lds ? cc_bytes : 0),
                                            exact_need + (cc_lds ? (size_t)T * 64 * sizeof(double) : 0));
          const int grid = ceil_div(N, 64);
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
          const int speculate = (p->debug_flags & MPPI_DEBUG_NO_SPECULATION) ? 0 : 1;
          if (speculate) p->spec_tiles_launched += (uint64_t)ceil_div(N, 64);
#define MPPI_LAUNCH_DEEP(CH, P2, CL)                                                                   \
  do {                                                                                                \
    auto kern = k_rollout_deep<CH, P2, CL, !EXACT>;                                                   \
    if (lds_total > 64 * 1024)                                                                        \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total));       \
    MPPI_KLAUNCH(kern, dim3(grid + extra), dim3(64 * kDeepWaves), lds_total, p->stream, d,      \
                       p->cells16, p->noise, p->u, p->costs, p->w_rel, p->tile_beta, p->cc_scratch,   \
                       (int)map_bytes, grid, speculate, next_job);                                     \
  } while (0)
#define MPPI_LAUNCH_DEEP_C(P2, CL)            \
  do {                                        \
    if (chunk == 8) MPPI_LAUNCH_DEEP(8, P2, CL);      \
    else if (chunk == 4) MPPI_LAUNCH_DEEP(4, P2, CL); \
    else MPPI_LAUNCH_DEEP(2, P2, CL);                 \
  } while (0)
          if (pow2res && cc_lds) MPPI_LAUNCH_DEEP_C(true, true);
          else if (pow2res) MPPI_LAUNCH_DEEP_C(true, false);
          else if (cc_lds) MPPI_LAUNCH_DEEP_C(false, true);
          else MPPI_LAUNCH_DEEP_C(false, false);
#undef MPPI_LAUNCH_DEEP_C
#undef MPPI_LAUNCH_DEEP
          char buf[320];
          snprintf(buf, sizeof(buf),
                   "k_rollout_deep%s chunk=%d pow2res=%d cc_lds=%d speculate=%d window=%dx%d@(%d,%d) lds=%zu "
                   "noise_blocks=%d problems=%d",
                   EXACT ? "" : "<f32>", chunk, (int)pow2res, (int)cc_lds, speculate, d.win_rows, d.win_cols, d.win_r0, d.win_c0, lds_total,
                   extra, p->inst_set ? p->B : 0);
          p->last_rollout = buf;
          p->tile_packets_fresh = true;
          break;
        }
      }
      static const bool no_spec = getenv("MPPI_NO_SPEC") != nullptr;  // developer switch (ablation)
      if (have_window && rot_ok && !no_pipe && !no_spec && keep_speculating &&
          !(p->debug_flags & MPPI_DEBUG_NO_SPEC_KERNEL)) {
        // speculative 4-wave pipeline (rollout_spec_kernel.h): same regime as the pipelined kernel below
        const size_t map_bytes = lds_win - sizeof(double2) * ((size_t)T + (size_t)(T + 1) / 2);
        // (this kernel pads the staged controls to a multiple of 8 steps)
        const size_t Tp = ((size_t)T + 7) & ~(size_t)7;
        const size_t lds_win = map_bytes + sizeof(double2) * (Tp + Tp / 2);
        int tiles_wg = ceil_div(ceil_div(N, 64), p->num_cus);
        if (tiles_wg < 1) tiles_wg = 1;
        if (tiles_wg > 3) tiles_wg = 3;  // (three tiles per CU: the pipelined kernel below)
        if (p->inst_set) while (p->inst_tiles % tiles_wg != 0) --tiles_wg;
        const size_t budget = (size_t)p->lds_per_cu - 1024;
        auto ring_bytes = [&](int chunk) {
          const size_t per_tile = chunk == 8 ? SpecRing<8>::kBytesPerTile : chunk == 4 ? SpecRing<4>::kBytesPerTile
                                                                                      : SpecRing<2>::kBytesPerTile;
          return (size_t)tiles_wg * per_tile + 16;
        };
        int chunk = 0;
        for (;;) {
          for (int cnd : {8, 4, 2})
            if (lds_win + ring_bytes(cnd) <= budget) { chunk = cnd; break; }
          if (chunk > 0 || tiles_wg == 1) break;
          --tiles_wg;
          if (p->inst_set) while (p->inst_tiles % tiles_wg != 0) --tiles_wg;
        }
        const bool latency_regime = tiles_wg <= 2 && ceil_div(ceil_div(N, 64), tiles_wg) <= p->num_cus;
        if (chunk > 0 && latency_regime) {
          // (rows padded to whole chunks: the cost wave reads them at immediate offsets)
          const size_t cc_bytes = (size_t)tiles_wg * ceil_div(T, chunk) * chunk * 64 * sizeof(double);
          const bool cc_lds = lds_win + ring_bytes(chunk) + cc_bytes <= budget && !(p->debug_flags & MPPI_DEBUG_CC_GLOBAL);
          const size_t lds_total = lds_win + ring_bytes(chunk) + (cc_lds ? cc_bytes : 0);
          const int block = 256 * tiles_wg;
          const int grid = ceil_div(N, 64 * tiles_wg);
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
          const int speculate = (p->debug_flags & MPPI_DEBUG_NO_SPECULATION) ? 0 : 1;
          if (speculate) p->spec_tiles_launched += (uint64_t)ceil_div(N, 64);
#define MPPI_LAUNCH_SPEC(CH, P2, CL)                                                                   \
  do {                                                                                                \
    auto kern = tiles_wg == 1 ? k_rollout_spec<CH, P2, CL, 1> : k_rollout_spec<CH, P2, CL, 2>;        \
    if (lds_total > 64 * 1024)                                                                        \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total));       \
    MPPI_KLAUNCH(kern, dim3(grid + extra), dim3(block), lds_total, p->stream, d, p->cells16,    \
                       p->noise, p->u, p->costs, p->w_rel, p->tile_beta, p->cc_scratch,                \
                       (int)map_bytes, grid, speculate, next_job);                                     \
  } while (0)
#define MPPI_LAUNCH_SPEC_C(P2, CL)            \
  do {                                        \
    if (chunk == 8) MPPI_LAUNCH_SPEC(8, P2, CL);      \
    else if (chunk == 4) MPPI_LAUNCH_SPEC(4, P2, CL); \
    else MPPI_LAUNCH_SPEC(2, P2, CL);                 \
  } while (0)
          if (pow2res && cc_lds) MPPI_LAUNCH_SPEC_C(true, true);
          else if (pow2res) MPPI_LAUNCH_SPEC_C(true, false);
          else if (cc_lds) MPPI_LAUNCH_SPEC_C(false, true);
          else MPPI_LAUNCH_SPEC_C(false, false);
#undef MPPI_LAUNCH_SPEC_C
#undef MPPI_LAUNCH_SPEC
          char buf[320];
          snprintf(buf, sizeof(buf),
                   "k_rollout_spec chunk=%d pow2res=%d cc_lds=%d tiles_per_wg=%d speculate=%d window=%dx%d@(%d,%d) "
                   "progressive=%d lds=%zu noise_blocks=%d problems=%d",
                   chunk, (int)pow2res, (int)cc_lds, tiles_wg, speculate, d.win_rows, d.win_cols, d.win_r0, d.win_c0,
                   d.win_progressive, lds_total, extra, p->inst_set ? p->B : 0);
          p->last_rollout = buf;
          p->tile_packets_fresh = true;
          break;
        }
      }
      if (have_window && rot_ok && !no_pipe) {
        // pipelined kernel: the map window in LDS + the incremental trig
        const size_t map_bytes = lds_win - sizeof(double2) * ((size_t)T + (size_t)(T + 1) / 2);
        int pairs = ceil_div(ceil_div(N, 64), p->num_cus);  // wave triples per workgroup
        if (pairs < 1) pairs = 1;
        if (pairs > 5) pairs = 5;                            // 15 waves = 960 threads
        // batched handle: the triples of a workgroup share one problem's window and controls
        if (p->inst_set) while (p->inst_tiles % pairs != 0) --pairs;
        const size_t budget = (size_t)p->lds_per_cu - 1024;
        auto ring_bytes = [&](int chunk) {
          return (size_t)pairs * (2 * (size_t)chunk * 64 * (sizeof(float2) + sizeof(double2)) + 2 * (size_t)chunk * 64);
        };
        int chunk = 0;
        for (;;) {  // fewer triples per workgroup (more workgroups than CUs) before giving the kernel up
          for (int cnd : {8, 4, 2})
            if (lds_win + ring_bytes(cnd) <= budget) { chunk = cnd; break; }
          if (chunk > 0 || pairs == 1) break;
          --pairs;
          if (p->inst_set) while (p->inst_tiles % pairs != 0) --pairs;
        }
        // The pipelined kernel is the low-latency choice: it wins while one workgroup per CU covers
        // the problem with at most three triples (measured, profiles/r01_ablation.md: 1 triple
        // 53 vs 79 us, 2 triples 76 vs 83 us, 4 triples a tie, two rounds 162 vs 88 us at T=200).
        // Beyond that the fused kernel below, 4..16 waves per CU, has the better throughput.
        const bool latency_regime = pairs <= 3 && ceil_div(ceil_div(N, 64), pairs) <= p->num_cus;
        if (chunk > 0 && latency_regime) {
          // control-cost products in LDS when there is room, else in a global scratch array
          const size_t cc_bytes = (size_t)pairs * T * 64 * sizeof(double);
          const bool cc_lds = lds_win + ring_bytes(chunk) + cc_bytes <= budget;
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
          break;
        }
      }
      static const bool no_window = getenv("MPPI_NO_WINDOW") != nullptr;  // developer switch (ablation)
      if (have_window && !no_window) {
        // the window makes it one workgroup per CU: size the workgroup so that the grid
        // is at most one wave of workgroups over the CUs
        // (at least 4 waves: one per SIMD, and four times the lanes to copy the window)
        const int waves = fused_waves_per_workgroup(p, N);
        int block = 64 * waves;
        static const bool no_fused = getenv("MPPI_NO_FUSED") != nullptr;  // developer switch (ablation)
        if (rot_ok && !no_fused) {
          auto fused = pow2res ? k_rollout_fused<true> : k_rollout_fused<false>;
          if (lds_win > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fused),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
          MPPI_KLAUNCH(fused, dim3(ceil_div(N, block)), dim3(block), lds_win, p->stream, d, p->cells16,
                             p->noise, p->u, p->costs, p->w_rel, p->tile_beta);
          char buf[200];
          snprintf(buf, sizeof(buf), "k_rollout_fused pow2res=%d waves_per_wg=%d window=%dx%d problems=%d",
                   (int)pow2res, waves, d.win_rows, d.win_cols, p->inst_set ? p->B : 0);
          p->last_rollout = buf;
          p->tile_packets_fresh = true;
          break;
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
      break;
    }
    case MPPI_MODE_SPEED_MAP: {
      p->tile_packets_fresh = false;
      size_t lds_win = 0;
      const bool have_window = plan_lds_window(p, d, &lds_win);  // 32-bit cells: 16 bits + risk byte
      TRY(upload_instances(p));
      const mppi_params& a = p->params;
      double wmax = std::fmax(std::fabs((double)a.wrange[0]), std::fabs((double)a.wrange[1]));
      double trmax = std::fmax(std::fabs(d.ang_lo), std::fabs(d.ang_lo + (double)d.ang_max_byte * d.ang_ratio));
      double dmax = (double)a.dt * wmax * trmax;
      static const bool no_fused = getenv("MPPI_NO_FUSED") != nullptr;  // developer switch (ablation)
      if (have_window && EXACT && BOUNDED && std::isfinite(dmax) && dmax <= 0.36 && T <= 2000 && !no_fused) {
        const int waves = fused_waves_per_workgroup(p, N);
        int res_exp = 0;
        const bool pow2res = std::frexp((double)a.res, &res_exp) == 0.5;
        auto fused = pow2res ? k_rollout_fused<true, true> : k_rollout_fused<false, true>;
        if (lds_win > 64 * 1024)
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fused),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
        MPPI_KLAUNCH(fused, dim3(ceil_div(N, 64 * waves)), dim3(64 * waves), lds_win, p->stream, d, p->cells16,
                           p->noise, p->u, p->costs, p->w_rel, p->tile_beta);
        char buf[200];
        snprintf(buf, sizeof(buf), "k_rollout_fused speed_map pow2res=%d waves_per_wg=%d window=%dx%d problems=%d",
                 (int)pow2res, waves, d.win_rows, d.win_cols, p->inst_set ? p->B : 0);
        p->last_rollout = buf;
        p->tile_packets_fresh = true;
        break;
      }
      MPPI_KLAUNCH((k_rollout_map<MAP_SPEED, EXACT, BOUNDED, false>), dim3(ceil_div(N, 64)), dim3(64),
                         lds_map, p->stream, d, p->cells, (const uint16_t*)nullptr, (const int8_t*)p->risk_ref,
                         p->noise, p->u, p->costs);
      p->last_rollout = "k_rollout_map speed_map global_cells exact=" + std::to_string((int)EXACT);
      break;
    }
    case MPPI_MODE_TDM: {
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
        if (EXACT && BOUNDED && std::isfinite(dmax) && dmax <= 0.36 && T <= 2000 && lds_fast <= 64 * 1024) {
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
            MPPI_KLAUNCH((k_rollout_tdm_fast<true>), dim3(N + extra), dim3(threads), lds_fast, p->stream, d, p->cells,
                               p->noise, p->u, p->costs, sc_out, mp2, N, next_job);
          else
            MPPI_KLAUNCH((k_rollout_tdm_fast<false>), dim3(N + extra), dim3(threads), lds_fast, p->stream, d, p->cells,
                               p->noise, p->u, p->costs, sc_out, mp2, N, next_job);
          p->last_rollout = std::string("k_rollout_tdm_fast pow2res=") + (pow2res ? "1" : "0") +
                            " noise_blocks=" + std::to_string(extra);
          break;
        }
      }
      MPPI_KLAUNCH((k_rollout_tdm<EXACT>), dim3(N), dim3(threads), lds, p->stream, d, p->cells, p->noise,
                         p->u, p->costs, sc_dst, mp2);
      p->last_rollout = "k_rollout_tdm exact=" + std::to_string((int)EXACT);
      break;
    }
    case MPPI_MODE_BAREBONE:
      p->tile_packets_fresh = false;
      MPPI_KLAUNCH((k_rollout_barebone<EXACT>), dim3(ceil_div(N, 64)), dim3(64), lds, p->stream, d,
                         p->obs_pos, p->obs_r, p->noise, p->u, p->costs);
      p->last_rollout = "k_rollout_barebone exact=" + std::to_string((int)EXACT);
      break;
    default:
      return fail(MPPI_ERR_INVALID, "bad mode");
  }
  HIP_TRY(hipGetLastError());
  return MPPI_OK;
}

static int launch_rollout(mppi_planner* p, const DevParams& d) {
  REQUIRE(p->B == 1 || p->inst_set, MPPI_ERR_STATE,
          "num_instances = %d: call mppi_planner_set_instances before solving", p->B);
  REQUIRE((size_t)p->cfg.num_steps * sizeof(double2) <= 64 * 1024, MPPI_ERR_INVALID, "num_steps %d too large",
          p->cfg.num_steps);
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
  if (p->cfg.math != MPPI_MATH_EXACT) return launch_rollout_t<false, false>(p, d);
  return bounded ? launch_rollout_t<true, true>(p, d) : launch_rollout_t<true, false>(p, d);
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
    const int per_problem = ceil_div(p->n_inst, p->scan_tile), total = ceil_div(p->n_local, p->scan_tile);
    if (apply_here)
      MPPI_KLAUNCH((k_combine_tiles<true>), grid, dim3(64), 0, p->stream, p->tbeta, p->tden, p->tnum, per_problem, total,
                   T, a.lambda_weight, my_packet, p->u, p->u_prev, (p->mirror_now ? p->u_host_dev : (float2*)nullptr), a.vrange[0], a.vrange[1],
                   a.wrange[0], a.wrange[1], p->stats, gen);
    else
      MPPI_KLAUNCH((k_combine_tiles<false>), grid, dim3(64), 0, p->stream, p->tbeta, p->tden, p->tnum, per_problem, total,
                   T, a.lambda_weight, my_packet, p->u, p->u_prev, (p->mirror_now ? p->u_host_dev : (float2*)nullptr), a.vrange[0], a.vrange[1],
                   a.wrange[0], a.wrange[1], p->stats, gen);
    if (p->graph_on) ++p->bumps_launched;
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
#define MPPI_LAUNCH_ROWS(APPLY, TC, FC)                                                                         \
  MPPI_KLAUNCH((k_update_rows<APPLY, TC, FC>), grid, dim3(kRowThreads), lds, p->stream,                   \
                     FC ? p->costs : p->w_rel, p->tile_beta, p->n_inst, p->inst_tiles, p->noise, T,             \
                     a.lambda_weight, my_packet, p->u, p->u_prev, (p->mirror_now ? p->u_host_dev : (float2*)nullptr), a.vrange[0], a.vrange[1],      \
                     a.wrange[0], a.wrange[1], p->stats, p->graph_on ? p->gen_dev : (unsigned long long*)nullptr)
#define MPPI_LAUNCH_ROWS_TC(APPLY, FC)        \
  do {                                        \
    if (many_rows) MPPI_LAUNCH_ROWS(APPLY, 4, FC); \
    else MPPI_LAUNCH_ROWS(APPLY, 1, FC);           \
  } while (0)
  if (apply_here && from_costs) MPPI_LAUNCH_ROWS_TC(true, true);
  else if (apply_here) MPPI_LAUNCH_ROWS_TC(true, false);
  else if (from_costs) MPPI_LAUNCH_ROWS_TC(false, true);
  else MPPI_LAUNCH_ROWS_TC(false, false);
#undef MPPI_LAUNCH_ROWS_TC
#undef MPPI_LAUNCH_ROWS
  if (p->graph_on) ++p->bumps_launched;
  HIP_TRY(hipGetLastError());
  return MPPI_OK;
}

static int launch_apply(mppi_planner* p) {
  const mppi_params& a = p->params;
  hipLaunchKernelGGL(k_apply, dim3(p->B), dim3(kUpdateThreads), 0, p->stream, p->packets, p->cfg.world_size,
                     p->cfg.rank, p->cfg.num_steps, a.lambda_weight, p->u, p->u_prev, (p->mirror_now ? p->u_host_dev : (float2*)nullptr), a.vrange[0],
                     a.vrange[1], a.wrange[0], a.wrange[1], p->stats);
  HIP_TRY(hipGetLastError());
  return MPPI_OK;
}

// `defer_exchange` (mppi_group_iterate_async): stop after this rank's packet; the caller issues the
// all-gathers of all its devices inside one RCCL group and then launches k_apply on each
static int launch_update(mppi_planner* p, bool prof, bool defer_exchange = false) {
  p->mirror_done = p->mirror_now;
  if (defer_exchange) return launch_update_local(p, false);
  // (a communicator on a single rank is honoured too: it exercises the same path as N ranks)
  // (samples sharded: every rank holds all N costs and all the noise -- the update is local)
  if ((p->cfg.world_size == 1 && !p->comm) || p->m_count > 1) {
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
  return launch_apply(p);
}

// One iteration: {noise unless it was produced ahead, rollout (+ the next iteration's noise when
// `want_next`), update}.  `have_noise`: noise_buf[noise_cur ^ 1] already holds this iteration's
// noise; on return it says the same for the following iteration.
static int launch_iteration(mppi_planner* p, const DevParams& d, bool& have_noise, bool want_next, bool prof,
                            bool defer_exchange = false) {
  // (below ~4M rollout-steps the generator takes less than the ~12 us a cross-stream dependency costs)
  static const bool no_side_stream = getenv("MPPI_NO_SIDE_STREAM") != nullptr;  // developer switch
  // and above 8 rollout waves per CU the register file has no room for the generator's waves: it
  // then runs in the rollout's tail and collides with the update (measured, profiles/r01_ablation.md)
  const bool side_stream_pays = (long)p->n_local * p->cfg.num_steps >= 4L * 1000 * 1000 &&
                                ceil_div(ceil_div(p->n_local, 64), p->num_cus) <= 8 && !no_side_stream;
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
    if (p->noise_on_side_stream) HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_noise_ready, 0));
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
  if (want_next && side_stream_pays) HIP_TRY(hipEventRecord(p->ev_buf_free, p->stream));
  {
    TraceRange tr("mppi:rollout");
    if (p->ktime_index >= 0) {
      p->kev_start = p->ktime_events[4 * (size_t)p->ktime_index];
      p->kev_stop = p->ktime_events[4 * (size_t)p->ktime_index + 1];
    }
    const int rc = launch_rollout(p, d);
    p->kev_start = p->kev_stop = nullptr;
    TRY(rc);
  }
  if (p->m_count > 1) {
    TraceRange tr("mppi:exchange_sample_costs");
    TRY(exchange_sample_costs(p));
  }
  have_noise = p->next_noise_done;
  if (want_next && !have_noise && side_stream_pays) {
    HIP_TRY(hipStreamWaitEvent(p->noise_stream, p->ev_buf_free, 0));
    TraceRange tr("mppi:noise_ahead");
    TRY(launch_noise(p, p->noise_buf[p->noise_cur ^ 1], p->noise_stream));
    HIP_TRY(hipEventRecord(p->ev_noise_ready, p->noise_stream));
    have_noise = p->noise_on_side_stream = true;
    if (p->graph_on) {
      // graph mode: join before the update, which advances the epoch counter the generator reads
      // (and a captured iteration must not leave a fork open)
      HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_noise_ready, 0));
      p->noise_on_side_stream = false;
    }
  }
  p->next_noise_wanted = false;
  p->scan_gen_now = false;
  if (prof) HIP_TRY(hipEventRecord(p->ev_stage[2], p->stream));
  TraceRange tr_update("mppi:update");
  if (p->ktime_index >= 0) {
    p->kev_start = p->ktime_events[4 * (size_t)p->ktime_index + 2];
    p->kev_stop = p->ktime_events[4 * (size_t)p->ktime_index + 3];
    ++p->ktime_index;
  }
  {
    const int rc = launch_update(p, prof, defer_exchange);
    p->kev_start = p->kev_stop = nullptr;
    TRY(rc);
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
    const void *lin, *ang, *cells, *cells16, *cc, *sample_costs;
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
  sig.sample_costs = p->sample_costs;
  sig.lin_grid = p->packed_lin_grid; sig.ang_grid = p->packed_ang_grid; sig.lin_maps = p->packed_lin_maps;
  sig.epoch_bias = p->noise_epoch - p->bumps_launched;
  sig.noise_cur = p->noise_cur; sig.inst_set = p->inst_set; sig.want_sample_costs = p->want_sample_costs;
  sig.speculation_off = p->speculation_off ? 1 : 0; sig.debug_flags = p->debug_flags;
  out.assign(reinterpret_cast<unsigned char*>(&sig), reinterpret_cast<unsigned char*>(&sig) + sizeof(sig));
}

// `timed`: bracket the iterations with events for mppi_planner_last_elapsed_ms / stage_times
// (iterate_async, profiling); solve() on the control path skips them
// called where the host has just waited for the stream: did speculation pay on this map?
static void review_speculation(mppi_planner* p) {
  if (!p->spec_fail_host) return;
  if (p->spec_tiles_launched == 0) {  // (nothing speculative ran: whatever the word holds is stale)
    *p->spec_fail_host = 0u;
    return;
  }
  const uint64_t failed = *p->spec_fail_host;
  if (2 * failed >= p->spec_tiles_launched) p->speculation_off = true;
  *p->spec_fail_host = 0u;
  p->spec_tiles_launched = 0;
}

static int run_iterations(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, int iterations, bool timed = true,
                          bool mirror_last = false) {
  REQUIRE(p->params_set, MPPI_ERR_STATE, "params not set");
  TRY(check_tdms(p, lin, ang));
  TRY(ensure_packed(p, lin, ang));
  DevParams d = make_dev_params(p, lin, ang);
  timed = timed || p->profile_stages;
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
      bool prof = p->profile_stages && k == (iterations >= 3 ? iterations - 2 : iterations - 1);
      p->mirror_now = mirror_last && k == iterations - 1;
      const int rc = launch_iteration(p, d, have_noise, true, prof);
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
    if ((!have_noise || !p->graph_warm) && k < iterations) {
      TRY(launch_iteration(p, d, have_noise, true, false));
      p->graph_warm = true;
      ++k;
    }
    const int chunk = p->graph_chunk;  // iterations per graph: even (noise double buffer)
    while (iterations - k >= chunk) {
      std::vector<unsigned char> sig;
      graph_signature(p, d, lin, ang, sig);
      sig.push_back(have_noise ? 1 : 0);
      const int slot = p->noise_cur & 1;
      if (!p->graph_exec[slot] || sig != p->graph_sig[slot]) {
        if (p->graph_exec[slot]) { (void)hipGraphExecDestroy(p->graph_exec[slot]); p->graph_exec[slot] = nullptr; }
        if (p->graph[slot]) { (void)hipGraphDestroy(p->graph[slot]); p->graph[slot] = nullptr; }
        p->graph_sig[slot].clear();
        const bool primed_before = have_noise;
        const uint64_t spec_before = p->spec_tiles_launched;
        HIP_TRY(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
        int rc = MPPI_OK;
        for (int j = 0; j < chunk && rc == MPPI_OK; ++j) rc = launch_iteration(p, d, have_noise, true, false);
        hipError_t end = hipStreamEndCapture(p->stream, &p->graph[slot]);
        if (rc != MPPI_OK) return rc;
        HIP_TRY(end);
        REQUIRE(have_noise == primed_before, MPPI_ERR_STATE, "graph capture: the iterations are not alike");
        HIP_TRY(hipGraphInstantiate(&p->graph_exec[slot], p->graph[slot], nullptr, nullptr, 0));
        p->graph_sig[slot] = sig;
        p->graph_spec_tiles[slot] = p->spec_tiles_launched - spec_before;
        ++p->graph_captures;
        // (capturing ran the host side of two iterations; the launch below runs their device side)
      } else {
        // the host-side counters a direct launch of the two iterations would have advanced
        if (p->cfg.rng == MPPI_RNG_PHILOX) p->noise_epoch += (uint64_t)chunk;
        p->bumps_launched += (uint64_t)chunk;
        // (the replayed kernels count their failed tiles like the captured ones did)
        p->spec_tiles_launched += p->graph_spec_tiles[slot];
      }
      HIP_TRY(hipGraphLaunch(p->graph_exec[slot], p->stream));
      ++p->graph_replays;
      k += chunk;
    }
    for (; k < iterations; ++k) TRY(launch_iteration(p, d, have_noise, true, false));
    p->primed = have_noise;
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

extern "C" int mppi_planner_iterate_async(mppi_planner* p, mppi_tdm* lin, mppi_tdm* ang, int iterations) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  REQUIRE(iterations >= 0, MPPI_ERR_INVALID, "iterations < 0");
  HIP_TRY(hipSetDevice(p->cfg.device));
  return run_iterations(p, lin, ang, iterations);
}

extern "C" int mppi_planner_synchronize(mppi_planner* p) {
  REQUIRE(p, MPPI_ERR_INVALID, "NULL planner");
  HIP_TRY(hipSetDevice(p->cfg.device));
  HIP_TRY(hipStreamSynchronize(p->stream));
  review_speculation(p);
  return finish_timing(p);
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
  HIP_TRY(hipStreamSynchronize(p->stream));
  review_speculation(p);
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
      if (*p->loop_done_count >= B) break;
    }
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
  HIP_TRY(hipStreamSynchronize(p->stream));
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
    HIP_TRY(hipStreamSynchronize(p->stream));
    return MPPI_OK;
  }
  p->noise_virtual = false;
  TRY(launch_noise(p, p->noise));
  HIP_TRY(hipStreamSynchronize(p->stream));
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
  HIP_TRY(hipStreamSynchronize(p->stream));
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
  HIP_TRY(hipStreamSynchronize(p->stream));
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
  HIP_TRY(hipStreamSynchronize(p->stream));
  return MPPI_OK;
}

extern "C" int mppi_planner_set_costs(mppi_planner* p, const float* costs) {
  REQUIRE(p && costs, MPPI_ERR_INVALID, "NULL argument");
  HIP_TRY(hipSetDevice(p->cfg.device));
  HIP_TRY(hipMemcpyAsync(p->costs, costs, sizeof(float) * (size_t)p->n_local, hipMemcpyHostToDevice, p->stream));
  p->tile_packets_fresh = false;
  p->scan_packets_fresh = false;
  p->sample_costs_local_only = false;
  HIP_TRY(hipStreamSynchronize(p->stream));
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
    HIP_TRY(hipStreamSynchronize(p->stream));
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
  HIP_TRY(hipStreamSynchronize(p->stream));
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
  HIP_TRY(hipStreamSynchronize(p->stream));
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
  HIP_TRY(hipStreamSynchronize(p->stream));
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
  HIP_TRY(hipStreamSynchronize(p->stream));
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
  TRY(run_iterations(p, lin, ang, 2, /*timed=*/false));  // steady state first
  p->ktime_index = 0;
  const int rc = run_iterations(p, lin, ang, reps, /*timed=*/false);
  p->ktime_index = -1;
  p->kev_start = p->kev_stop = nullptr;
  TRY(rc);
  HIP_TRY(hipStreamSynchronize(p->stream));
  double sum[2] = {0.0, 0.0};
  for (int r = 0; r < reps; ++r)
    for (int k = 0; k < 2; ++k) {
      float ms = 0.f;
      HIP_TRY(hipEventElapsedTime(&ms, p->ktime_events[4 * (size_t)r + 2 * k], p->ktime_events[4 * (size_t)r + 2 * k + 1]));
      sum[k] += (double)ms;
    }
  *us_rollout = (float)(1e3 * sum[0] / reps);
  *us_update = (float)(1e3 * sum[1] / reps);
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
