// tune.h — first-call tile-shape selection for the GEMM-shaped kernels (igemm fwd/dgrad, wgrad).
//
// The best block tile depends on the layer (pixels vs channels vs reduction length vs how many workgroups the grid
// gets on 256 CUs) in ways the static heuristics only approximate (r01 sweeps: +-30 % per layer).  The first EAGER call
// of a given problem times every legal candidate on the caller's stream (2 launches each, HIP events) and caches the
// winner for the process; calls made while the stream is being captured into a hipGraph never measure — they use the
// cached pick or fall back to the heuristic, so graph capture stays legal.  SY11_TUNE=0 disables measuring,
// SY11_TUNE_LOG=1 prints every decision.
#pragma once
// sizes of the two tile-configuration tables (igemm.hip / wgrad.hip own the meaning of an index; core.hip validates imported picks
// against these — ONE definition, so that adding a configuration cannot leave the importer behind)
constexpr int SY11_IGEMM_NCFG = 26;
constexpr int SY11_WGRAD_NCFG = 20;

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>
#include <unordered_map>

// process-wide options (core.hip): defaults come from the environment (SY11_TUNE, SY11_TUNE_LOG, SY11_IGEMM_CFG,
// SY11_WGRAD_CFG, SY11_IGEMM_KORDER, SY11_IGEMM_DEEP, SY11_IGEMM_BPOL), sy11_set_option() changes them at run time
enum Sy11Opt { OPT_TUNE = 0, OPT_TUNE_LOG, OPT_IGEMM_CFG, OPT_WGRAD_CFG, OPT_IGEMM_KORDER, OPT_IGEMM_DEEP, OPT_IGEMM_BPOL, OPT_DGRAD_S2_HALO, OPT_ROW_MAP, OPT_DETERMINISTIC, OPT_COUNT };
int sy11_opt(int which);

namespace sy11tune {

inline bool enabled() { return sy11_opt(OPT_TUNE) == 1; }
inline bool logging() { return sy11_opt(OPT_TUNE_LOG) == 1; }
inline bool capturing(hipStream_t st) {
  hipStreamCaptureStatus s = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &s) != hipSuccess) { (void)hipGetLastError(); return true; }   // unknown -> do not measure
  return s != hipStreamCaptureStatusNone;
}
inline uint64_t hash(const int* v, int n) {
  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < n; ++i) { h ^= (uint32_t)v[i]; h *= 1099511628211ull; }
  return h;
}

struct Cache {
  std::unordered_map<uint64_t, int> map;
  std::mutex mu;
  bool get(uint64_t k, int* out) {
    std::lock_guard<std::mutex> g(mu);
    auto it = map.find(k);
    if (it == map.end()) return false;
    *out = it->second;
    return true;
  }
  void put(uint64_t k, int v) {
    std::lock_guard<std::mutex> g(mu);
    map[k] = v;
  }
};
// the process's pick tables (core.hip): kind 0 = igemm (fwd / dgrad), 1 = wgrad.  One table per kind so that the picks
// can be exported / imported as a whole (sy11_tune_export / sy11_tune_import: every rank of a data-parallel job runs the
// kernels rank 0 measured).
Cache& cache(int kind);

// run(cand) launches candidate `cand` on `st` and returns 0 on success.  Returns the fastest candidate, or -1 if
// nothing could be measured.
template <class F>
int pick(const int* cands, int ncand, F&& run, hipStream_t st, const char* what, const int* key, int nkey) {
  // one measurement at a time per process: the event pair is shared, and two threads timing candidates concurrently would also
  // time each other's kernels (the library's only other process-wide state is the option table and the pick tables, core.hip)
  static std::mutex mu;
  std::lock_guard<std::mutex> guard(mu);
  static hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!e0 && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) { (void)hipGetLastError(); return -1; }
  // candidates are timed with the rest of the GPU idle: the caller may be one of several streams (the engine launches filter
  // gradients on a second one), and kernels still draining elsewhere would be timed along with the first candidates.  Never
  // reached while `st` is being captured (the callers check), and measuring is a first-call event per problem key.
  (void)hipDeviceSynchronize();
  int best = -1;
  float best_ms = 0.f;
  for (int i = 0; i < ncand; ++i) {
    if (run(cands[i]) != 0) continue;                                   // warm-up (code object load, caches)
    if (hipEventRecord(e0, st) != hipSuccess) return -1;
    if (run(cands[i]) != 0 || run(cands[i]) != 0) continue;
    if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); return -1; }
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { (void)hipGetLastError(); continue; }
    if (logging()) fprintf(stderr, "[sy11 tune] %s cand %d: %.1f us\n", what, cands[i], ms * 500.f);
    if (best < 0 || ms < best_ms) { best = cands[i]; best_ms = ms; }
  }
  if (logging() && best >= 0) {
    fprintf(stderr, "[sy11 tune] %s key", what);
    for (int i = 0; i < nkey; ++i) fprintf(stderr, " %d", key[i]);
    fprintf(stderr, " -> %d (%.1f us)\n", best, best_ms * 500.f);
  }
  return best;
}

}  // namespace sy11tune
