// core.hip — version / thread-local error string of libsy11.
#include "common.h"
#include "tune.h"
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
static const char* const kOptName[OPT_COUNT] = {"tune", "tune_log", "igemm_cfg", "wgrad_cfg", "igemm_korder", "igemm_deep", "igemm_bpol", "dgrad_s2_halo", "row_map"};
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
  SY11_REQUIRE(i >= 0, "set_option: unknown option '%s' (tune, tune_log, igemm_cfg, wgrad_cfg, igemm_korder, igemm_deep, igemm_bpol, dgrad_s2_halo, row_map)", name ? name : "(null)");
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
    SY11_REQUIRE(v >= 0 && v < (k == 0 ? 19 : 12), "tune_import: record %ld (kind %d) picks configuration %d, outside this build's table", (long)i, k, v);
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
