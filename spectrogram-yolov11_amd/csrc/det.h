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
// clean: write zeros back over the partial rows once read (blocks from the clean head of the workspace, below)
int sy11_fold_rows_ordered(long rows, int N, float* partials, long stride, float* out, float* scratch, hipStream_t st, bool clean = false);
// two buffers ([2][rows][N], buf_stride floats apart) in one launch per stage; scratch = 2 x sy11_fold_scratch_floats; same sums as two folds
int sy11_fold_rows_ordered2(long rows, int N, float* partials, long stride, long buf_stride, float* out0, float* out1, float* scratch,
                            hipStream_t st, bool clean = false);
// out{0,1}[s * slot_stride + c] += sum of the partial rows r = s (mod slots), r ascending; two buffers per launch          (elementwise.hip)
int sy11_fold_rows_to_slots2(long rows, int N, float* partials, long stride, long buf_stride, int slots, long slot_stride, float* out0, float* out1,
                             hipStream_t st, bool clean = false);
inline size_t sy11_fold_scratch_floats(long rows, int N) { return rows > 256 ? (size_t)(rows / 32 + 128) * N : 0; }

// A [nbuf][rows][N] partial block plus fold scratch from the stream's workspace; nullptr on failure.
// Blocks whose kernels add into a ZEROED block (zero = true: the conv kernels' statistic rows — a workgroup writes only its own
// columns of its row) come from the CLEAN head of the workspace: it is zeroed once, when the workspace is allocated (core.hip), and
// every fold of such a block writes zeros back over what it has read — so the next user finds zeros again without a zero-fill launch
// (r04: 91 of them per step).  Blocks whose kernels store every element (zero = false) and all fold scratch live behind the clean head.
constexpr size_t SY11_DET_CLEAN_FLOATS = (size_t)8 << 20;      // 32 MB: the largest statistic block of yolo11s at batch 64 is 3.3 MB
struct DetPartials {
  float* base = nullptr; float* scratch = nullptr; long rows = 0; int N = 0; int nbuf = 0; hipStream_t st = nullptr; bool clean = false;
  bool acquire(hipStream_t stream, int buffers, long nrows, int n, bool zero = true) {
    st = stream; rows = nrows; N = n; nbuf = buffers;
    const size_t body = (size_t)buffers * nrows * n, scr = (buffers >= 2 ? 2 : 1) * sy11_fold_scratch_floats(nrows, n);
    clean = zero && body <= SY11_DET_CLEAN_FLOATS;
    float* ws = sy11_det_workspace(stream, (SY11_DET_CLEAN_FLOATS + (clean ? 0 : body) + scr + 4) * sizeof(float));
    if (!ws) return false;
    base = clean ? ws : ws + SY11_DET_CLEAN_FLOATS;
    scratch = ws + SY11_DET_CLEAN_FLOATS + (clean ? 0 : body);
    return clean || !zero || sy11_zero_floats(base, body, stream) == 0;     // zero = false: the caller's kernels STORE every element of every row
  }
  float* buf(int i) const { return base + (size_t)i * rows * N; }
  int fold(int i, float* out) const { return sy11_fold_rows_ordered(rows, N, buf(i), N, out, scratch, st, clean); }
  // buffers 0 and 1 together (the two BatchNorm statistic / backward sums of a launch): half the fold launches, the same sums
  // out_slots > 1: the caller's buffers are [out_slots][slot_stride] slot rows that its CONSUMER folds in a fixed order anyway (bn_finalize,
  // the BatchNorm backward apply's prologue): more than 256 partial rows then go into those slot rows in ONE plain launch (slot s = rows
  // s, s + slots, ... in index order) instead of down to a single row through two or three 64 -> 1 stages
  int fold01(float* out0, float* out1, int out_slots = 1, long slot_stride = 0) const {
    if (out_slots > 1 && rows > 256) return sy11_fold_rows_to_slots2(rows, N, base, N, (long)rows * N, out_slots, slot_stride, out0, out1, st, clean);
    return sy11_fold_rows_ordered2(rows, N, base, N, (long)rows * N, out0, out1, scratch, st, clean);
  }
};
