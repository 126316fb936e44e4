// det.h — ordered ("deterministic") reductions across workgroups.
//
// Default mode: partial sums of different workgroups meet in HBM through f32 atomics (dense, slotted): the order of the adds — and
// with it the last bits of every BatchNorm statistic, BatchNorm-backward sum, filter gradient and loss partial — varies run to
// run.  With the "deterministic" option (SY11_DETERMINISTIC=1 / sy11_set_option("deterministic", 1); the reference's
// `deterministic: True`, cfg/default.yaml:29, utils/torch_utils.py:474-492) every such sum runs as
//     kernel: workgroup i -> partial row i of a zeroed workspace (ONE add per element: exact)      [rows x N]
//     fold  : rows added in index order by a fixed-shape tree (64 rows per stage)                   -> out[N] += total
// and the in-workgroup part of every reduction is an ordered fold as well (always, not only in this mode).  Two runs of the same
// program on the same inputs are then bit-identical, PROVIDED the tile choices are the same: run with the tuner off
// ("tune" 0, the heuristic tiles) or with an imported pick table.
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "tune.h"

inline bool sy11_det(int bit = 0) {
  static int mask = -2;
  if (mask == -2) { const char* e = getenv("SY11_DET_MASK"); mask = e ? atoi(e) : -1; }      // debugging aid: bit set = that family ordered
  return sy11_opt(OPT_DETERMINISTIC) != 0 && (bit == 0 || (mask & bit));
}
float* sy11_det_workspace(hipStream_t st, size_t bytes);                        // core.hip; nullptr = cannot allocate (inside a capture)
int sy11_zero_floats(float* p, size_t n, hipStream_t st);                       // core.hip: a plain kernel (a memset NODE of a captured graph
                                                                                // is not reliably ordered with its neighbours on this stack)
// out[c] += sum_r partials[r * stride + c], c < N; `scratch` = rows/64 * N floats (only read when rows > 256)     (elementwise.hip)
int sy11_fold_rows_ordered(long rows, int N, const float* partials, long stride, float* out, float* scratch, hipStream_t st);
// two buffers ([2][rows][N], buf_stride floats apart) in one launch per stage; scratch = 2 x sy11_fold_scratch_floats; same sums as two folds
int sy11_fold_rows_ordered2(long rows, int N, const float* partials, long stride, long buf_stride, float* out0, float* out1, float* scratch,
                            hipStream_t st);
inline size_t sy11_fold_scratch_floats(long rows, int N) { return rows > 256 ? (size_t)(rows / 32 + 128) * N : 0; }

// A zeroed [nbuf][rows][N] partial block plus fold scratch from the stream's workspace; nullptr on failure.
struct DetPartials {
  float* base = nullptr; float* scratch = nullptr; long rows = 0; int N = 0; int nbuf = 0; hipStream_t st = nullptr;
  bool acquire(hipStream_t stream, int buffers, long nrows, int n, bool zero = true) {
    st = stream; rows = nrows; N = n; nbuf = buffers;
    const size_t body = (size_t)buffers * nrows * n, scr = (buffers >= 2 ? 2 : 1) * sy11_fold_scratch_floats(nrows, n);
    base = sy11_det_workspace(stream, (body + scr + 4) * sizeof(float));
    if (!base) return false;
    scratch = base + body;
    return !zero || sy11_zero_floats(base, body, stream) == 0;     // zero = false: the caller's kernels STORE every element of every row
  }
  float* buf(int i) const { return base + (size_t)i * rows * N; }
  int fold(int i, float* out) const { return sy11_fold_rows_ordered(rows, N, buf(i), N, out, scratch, st); }
  // buffers 0 and 1 together (the two BatchNorm statistic / backward sums of a launch): half the fold launches, the same sums
  int fold01(float* out0, float* out1) const { return sy11_fold_rows_ordered2(rows, N, base, N, (long)rows * N, out0, out1, scratch, st); }
};
