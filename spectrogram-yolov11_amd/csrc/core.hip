// core.hip — version / thread-local error string of libsy11.
#include "common.h"
#include "tune.h"
#include "det.h"
#include <string.h>

static thread_local char g_err[512] = "";

void sy11_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int sy11_version(void) { return SY11_VERSION; }
extern "C" const char* sy11_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------ run-time options
static int g_opt[OPT_COUNT];
static std::once_flag g_opt_once;
static const char* const kOptName[OPT_COUNT] = {"tune", "tune_log", "igemm_cfg", "wgrad_cfg", "igemm_korder", "igemm_deep", "igemm_bpol", "dgrad_s2_halo", "row_map", "deterministic"};
static void opt_init() {
  auto env = [](const char* n, int dflt) { const char* e = getenv(n); return e ? atoi(e) : dflt; };
  g_opt[OPT_TUNE] = env("SY11_TUNE", 1) != 0;
  g_opt[OPT_TUNE_LOG] = env("SY11_TUNE_LOG", 0) != 0;
  g_opt[OPT_IGEMM_CFG] = env("SY11_IGEMM_CFG", -1);
  g_opt[OPT_WGRAD_CFG] = env("SY11_WGRAD_CFG", -1);
  g_opt[OPT_IGEMM_KORDER] = env("SY11_IGEMM_KORDER", 1);
  g_opt[OPT_IGEMM_DEEP] = env("SY11_IGEMM_DEEP", 0);
  g_opt[OPT_IGEMM_BPOL] = env("SY11_IGEMM_BPOL", 0);
  g_opt[OPT_DGRAD_S2_HALO] = env("SY11_DGRAD_S2_HALO", 1);
  g_opt[OPT_ROW_MAP] = env("SY11_ROW_MAP", 1);
  g_opt[OPT_DETERMINISTIC] = env("SY11_DETERMINISTIC", 0) != 0;
}
int sy11_opt(int which) {
  std::call_once(g_opt_once, opt_init);
  return g_opt[which];
}
static int opt_index(const char* name) {
  if (name)
    for (int i = 0; i < OPT_COUNT; ++i)
      if (!strcmp(name, kOptName[i])) return i;
  return -1;
}
extern "C" int sy11_set_option(const char* name, int32_t value) {
  const int i = opt_index(name);
  SY11_REQUIRE(i >= 0, "set_option: unknown option '%s' (tune, tune_log, igemm_cfg, wgrad_cfg, igemm_korder, igemm_deep, igemm_bpol, dgrad_s2_halo, row_map, deterministic)", name ? name : "(null)");
  std::call_once(g_opt_once, opt_init);
  g_opt[i] = value;
  return SY11_OK;
}
extern "C" int sy11_get_option(const char* name, int32_t* value) {
  const int i = opt_index(name);
  SY11_REQUIRE(i >= 0 && value, "get_option: unknown option '%s' or null result pointer", name ? name : "(null)");
  *value = sy11_opt(i);
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ ordered-reduction workspace
// "deterministic" 1 (cfg/default.yaml:29 `deterministic: True`, utils/torch_utils.py:474-492 in the reference): every sum over
// workgroups is taken in a FIXED order — the reducing kernels write one partial row per workgroup into this workspace and a
// fold kernel adds the rows in index order (det.h).  One buffer per stream, grown geometrically, old buffers are kept alive (a
// captured hipGraph may still point into them); sized during the eager warm-up steps, so capture never allocates.
namespace {
struct DetWs { void* p = nullptr; size_t bytes = 0; };
std::mutex g_det_mu;
std::unordered_map<hipStream_t, DetWs> g_det_ws;
}  // namespace
float* sy11_det_workspace(hipStream_t st, size_t bytes) {
  std::lock_guard<std::mutex> g(g_det_mu);
  DetWs& w = g_det_ws[st];
  if (bytes <= w.bytes) return (float*)w.p;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
  if (cs != hipStreamCaptureStatusNone) {
    // cannot allocate inside a capture.  The engine captures on a fresh stream after eager warm-up steps of the same shapes on the
    // launch stream, and replays the graph ON that launch stream, in order with its eager kernels: the warm-up's buffer is free
    // whenever a node of this graph runs.  A stream that is itself part of a capture right now (the engine's filter-gradient
    // stream: the same stream object in the warm-up steps and in the capture, so it found its own buffer above) keeps its buffer
    // to itself — two branches of one graph run concurrently and must not fold through the same rows.
    for (auto& kv : g_det_ws) {
      if (kv.second.bytes < bytes || kv.first == st) continue;
      hipStreamCaptureStatus other = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(kv.first, &other) != hipSuccess) { (void)hipGetLastError(); continue; }
      if (other == hipStreamCaptureStatusNone) return (float*)kv.second.p;
    }
    return nullptr;                                                    // nothing large enough: the caller reports an error
  }
  size_t want = w.bytes ? w.bytes * 2 : (size_t)64 << 20;
  while (want < bytes) want *= 2;
  void* q = nullptr;
  if (hipMalloc(&q, want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  // the clean head (det.h): zero now, kept zero by the folds that read it.  (Not inside a capture — checked above — so a plain memset.)
  if (hipMemset(q, 0, want < SY11_DET_CLEAN_FLOATS * sizeof(float) ? want : SY11_DET_CLEAN_FLOATS * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError(); (void)hipFree(q); return nullptr;
  }
  w.p = q; w.bytes = want;                                             // the previous buffer is deliberately not freed (see above)
  return (float*)q;
}

__global__ __launch_bounds__(256) void zero_floats_kernel(float4* __restrict__ p, size_t n4) {
  const float4 z = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = z;
}
int sy11_zero_floats(float* p, size_t n, hipStream_t st) {               // p: 16-byte aligned (workspace base + multiples of 4 floats)
  if (n == 0) return SY11_OK;
  const size_t n4 = (n + 3) / 4;                                          // the workspace is allocated in 64 MB steps: the tail is ours
  size_t g = (n4 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(zero_floats_kernel, dim3((unsigned)g), dim3(256), 0, st, (float4*)p, n4);
  SY11_LAUNCH_CHECK("zero_floats");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ tuner pick tables
namespace sy11tune {
Cache& cache(int kind) {
  static Cache c[2];
  return c[kind & 1];
}
}  // namespace sy11tune

// records of 16 bytes: {u64 problem hash, i32 kind, i32 pick}
extern "C" int64_t sy11_tune_export(void* buf, int64_t capacity_bytes) {
  int64_t n = 0;
  for (int kind = 0; kind < 2; ++kind) {
    sy11tune::Cache& c = sy11tune::cache(kind);
    std::lock_guard<std::mutex> g(c.mu);
    for (const auto& kv : c.map) {
      if (buf && (n + 1) * 16 <= capacity_bytes) {
        unsigned char* p = (unsigned char*)buf + n * 16;
        const uint64_t h = kv.first;
        const int32_t k = kind, v = kv.second;
        memcpy(p, &h, 8); memcpy(p + 8, &k, 4); memcpy(p + 12, &v, 4);
      }
      ++n;
    }
  }
  return n * 16;                      // bytes needed (call with buf = NULL to size the buffer)
}
extern "C" int sy11_tune_import(const void* buf, int64_t bytes) {
  SY11_REQUIRE(bytes >= 0 && bytes % 16 == 0 && (buf || bytes == 0), "tune_import: buffer must hold whole 16-byte records");
  for (int64_t i = 0; i < bytes / 16; ++i) {
    const unsigned char* p = (const unsigned char*)buf + i * 16;
    uint64_t h; int32_t k, v;
    memcpy(&h, p, 8); memcpy(&k, p + 8, 4); memcpy(&v, p + 12, 4);
    SY11_REQUIRE(k == 0 || k == 1, "tune_import: record %ld has kind %d", (long)i, k);
    SY11_REQUIRE(v >= 0 && v < (k == 0 ? SY11_IGEMM_NCFG : SY11_WGRAD_NCFG), "tune_import: record %ld (kind %d) picks configuration %d, outside this build's table", (long)i, k, v);
  }
  for (int64_t i = 0; i < bytes / 16; ++i) {           // all records valid: apply
    const unsigned char* p = (const unsigned char*)buf + i * 16;
    uint64_t h; int32_t k, v;
    memcpy(&h, p, 8); memcpy(&k, p + 8, 4); memcpy(&v, p + 12, 4);
    sy11tune::cache(k).put(h, v);
  }
  return SY11_OK;
}
extern "C" int sy11_tune_clear(void) {
  for (int kind = 0; kind < 2; ++kind) {
    sy11tune::Cache& c = sy11tune::cache(kind);
    std::lock_guard<std::mutex> g(c.mu);
    c.map.clear();
  }
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ measured peak (bench.py)
// Pure-MFMA loop: every wave issues `iters` x 8 back-to-back v_mfma_f32_32x32x16_f16 on register operands (4 independent
// accumulators, no memory traffic) — the matrix-core throughput the chip actually sustains at the clock it holds under this
// load, to set beside the datasheet-class 2.5 PFLOP/s.  out[wave] receives a checksum so the loop cannot be optimised away.
__global__ __launch_bounds__(256) void peak_mfma_f16_kernel(int iters, float* __restrict__ out) {
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (i + 1)); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c3, 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = s;
}
// FLOPs of one launch = workgroups * 4 waves * iters * 8 * (2 * 32 * 32 * 16); out: workgroups * 4 floats
extern "C" int sy11_peak_mfma_f16(int32_t workgroups, int32_t iters, float* out, void* stream) {
  SY11_REQUIRE(workgroups > 0 && iters > 0 && out, "peak_mfma_f16: bad argument");
  hipLaunchKernelGGL(peak_mfma_f16_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, iters, out);
  SY11_LAUNCH_CHECK("peak_mfma_f16");
  return SY11_OK;
}
