// host_common.h -- error reporting, allocation helpers and the run-time lookups of roctx / RCCL shared by
// the host side of libmppi_hip.so (included by mppi_api.hip only: everything here has internal linkage).
#pragma once

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
