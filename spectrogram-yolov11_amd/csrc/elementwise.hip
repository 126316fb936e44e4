// elementwise.hip — HBM-bound NHWC kernels around the convolutions (gfx950): BatchNorm finalize / apply / backward,
// SiLU, residual add, strided channel-slice copies (concat by pointer), nearest 2x upsample, 5x5 max-pool.
//
// Common shape: a tensor view is M pixels x C channels with a pixel stride `ld`.  A thread owns ONE fixed
// channel vector (VEC = 16 bytes of channels when alignment allows, else 1 element) and walks rows, so the
// per-channel parameters are loaded once and every global access of a wave is a run of full 16-byte lanes.
#include "common.h"
#include "tune.h"
#include "det.h"
#include <type_traits>

template <int I, int N, typename F>
__device__ __forceinline__ void sy11_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sy11_static_for<I + 1, N>(f);
  }
}


template <typename T, int VEC> struct Vec {
  T v[VEC];
};
template <typename T, int VEC>
__device__ __forceinline__ void vload(const T* p, float* f) {
  if constexpr (VEC == 1) {
    f[0] = ElemTraits<T>::to_f(*p);
  } else {
    typedef T vt __attribute__((ext_vector_type(VEC)));
    const vt v = *(const vt*)p;
#pragma unroll
    for (int i = 0; i < VEC; ++i) f[i] = ElemTraits<T>::to_f(v[i]);
  }
}
template <typename T, int VEC>
__device__ __forceinline__ void vstore(T* p, const float* f) {
  if constexpr (VEC == 1) {
    *p = ElemTraits<T>::from_f(f[0]);
  } else {
    typedef T vt __attribute__((ext_vector_type(VEC)));
    vt v;
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = ElemTraits<T>::from_f(f[i]);
    *(vt*)p = v;
  }
}

// rows-per-block geometry shared by the row-walking kernels
struct RowGeom {
  int cpv;        // channel vectors per row
  int rows_pb;    // rows handled concurrently by one block (256 / min(cpv,256))
  int cblocks;    // blocks along channels (cpv > 256)
};
static RowGeom row_geom(int C, int vec) {
  RowGeom g;
  g.cpv = C / vec;
  const int cw = g.cpv < 256 ? g.cpv : 256;
  g.rows_pb = 256 / cw;
  g.cblocks = cdiv(g.cpv, 256);
  return g;
}
static bool vec_ok(int esz, int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
  const int vec = 16 / esz;
  if (C % vec) return false;
  for (int l : lds) if (l % vec) return false;
  for (const void* p : ptrs) if (p && ((uintptr_t)p & 15)) return false;
  return true;
}

// Which rows a (block, trip, unrolled load) touches.  row = blockIdx.x*bs + trip*ts + u*us + rsub, for rows below `mend`:
//   chunked (row_map 1, default): every block owns ONE contiguous run of `chunk` rows = 1-2 trips of U row groups and retires, so
//     the chip sweeps the tensor front to back like a flat 1-D elementwise kernel.  Measured r02 on 210 MB f16 tensors
//     (tools/bn_sweep.py): BN+SiLU apply 88.7 -> 75 us (5.6 TB/s), backward apply 130 -> 115 (5.4), reduce 101 -> 92;
//     one trip pays the per-channel parameter loads too often for C >= 128 (122 us), four or more serialise on memory latency
//     (84 / 92 us), two is the optimum;
//   strided (row_map 0, the r01 mapping): a 2048-block grid walks the tensor with a grid-sized stride, U loads a whole grid apart.
struct RowWalk {
  long bs, ts, us, chunk;     // chunk = 0: strided
  int grid;
};
static RowWalk row_walk(long M, int rows_pb, int U, int trips, long max_blocks) {
  RowWalk w;
  if (sy11_opt(OPT_ROW_MAP) == 0) {
    long g = (M + rows_pb * (long)U - 1) / (rows_pb * (long)U);
    const long cap = max_blocks < 2048 ? max_blocks : 2048;
    g = g < 1 ? 1 : (g > cap ? cap : g);
    w.grid = (int)g; w.bs = rows_pb; w.us = g * rows_pb; w.ts = w.us * U; w.chunk = 0;
    return w;
  }
  const long unit = (long)rows_pb * U;
  const long units = (M + unit - 1) / unit;
  long t = trips < 1 ? 1 : trips;
  long blocks = (units + t - 1) / t;
  if (blocks > max_blocks) t = (units + max_blocks - 1) / max_blocks;
  else if (blocks > 2048 && blocks < 6144) t = (units + 2047) / 2048;     // a partial second residency round costs more than longer blocks
  const long chunk = unit * t;
  w.chunk = chunk; w.bs = chunk; w.ts = unit; w.us = rows_pb;
  w.grid = (int)((M + chunk - 1) / chunk);
  return w;
}
__device__ __forceinline__ long walk_end(const RowWalk& w, long M) {
  if (w.chunk == 0) return M;
  const long e = ((long)blockIdx.x + 1) * w.chunk;
  return e < M ? e : M;
}

// ------------------------------------------------------------------------------------------------ BN finalize
__global__ __launch_bounds__(256) void bn_finalize_kernel(int C, int slots, double count, const float* __restrict__ ssum,
                                                          const float* __restrict__ ssq, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                                          float* running_var, float* mean, float* rstd, float* scale, float* shift) {
  // 32 channels per block; 8 lanes-groups split the slots of a channel, folded through LDS (all loads independent)
  __shared__ double red[2][8][32];
  const int cl = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s1 = 0, s2 = 0;
  if (c < C)
    for (int k = sg; k < slots; k += 8) { s1 += (double)ssum[(long)k * C + c]; s2 += (double)ssq[(long)k * C + c]; }
  red[0][sg][cl] = s1;
  red[1][sg][cl] = s2;
  __syncthreads();
  if (sg != 0 || c >= C) return;
#pragma unroll
  for (int k = 1; k < 8; ++k) { s1 += red[0][k][cl]; s2 += red[1][k][cl]; }
  const double mu = s1 / count;
  double var = s2 / count - mu * mu;
  if (var < 0) var = 0;
  const float r = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)mu;
  rstd[c] = r;
  const float sc = gamma[c] * r;
  scale[c] = sc;
  shift[c] = beta[c] - (float)mu * sc;
  if (running_mean) {
    const double unbiased = count > 1 ? var * count / (count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

extern "C" int sy11_bn_finalize(int32_t C, int32_t stat_slots, double count, const float* stat_sum, const float* stat_sq, const float* gamma,
                                const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                float* mean, float* rstd, float* scale, float* shift, void* stream) {
  SY11_REQUIRE(C > 0 && count > 0, "bn_finalize: bad C/count");
  SY11_REQUIRE(stat_sum && stat_sq && gamma && beta && mean && rstd && scale && shift, "bn_finalize: null pointer");
  SY11_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must both be given or both NULL");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, C, stat_slots > 1 ? stat_slots : 1, count, stat_sum, stat_sq, gamma,
                     beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift);
  SY11_LAUNCH_CHECK("bn_finalize");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ BN apply (+SiLU, +res)
// Per-channel coefficients of the three BN passes go through LDS: thread e of a block prepares channel (cbase + e) ONCE, then every
// thread reads its VEC channels back as 16-byte LDS vectors.  Loading them per thread from global memory is K x VEC scalars per thread
// — for C >= 128 as many L1 requests as the block's data when a block makes one or two trips (r02: 80x80x256 apply 78 -> 122 us).
// The first trip's data loads are issued BEFORE the barrier that publishes the coefficients, so their latencies overlap.
extern __shared__ __attribute__((aligned(16))) float s_coef[];

template <int VEC>
__device__ __forceinline__ void coef_get(const float* arr, int cl0, float (&out)[VEC]) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) out[i] = arr[cl0 + i];
}
struct RowLane {        // a thread's place in the row walk
  int cl, cv, rsub, c, nch, cbase, cwv;
  bool active;
};
template <int VEC>
__device__ __forceinline__ RowLane row_lane(int cpv, int rows_pb) {
  RowLane l;
  const int cw = cpv < 256 ? cpv : 256;
  l.cl = threadIdx.x % cw;
  l.cv = blockIdx.y * 256 + l.cl;
  l.rsub = threadIdx.x / cw;
  l.c = l.cv * VEC;
  l.nch = min(cw, cpv - (int)blockIdx.y * 256) * VEC;
  l.cbase = blockIdx.y * 256 * VEC;
  l.cwv = cw * VEC;
  l.active = l.cv < cpv && l.rsub < rows_pb;
  return l;
}
static inline size_t coef_bytes(const RowGeom& g, int vec, int k) { return (size_t)k * (g.cpv < 256 ? g.cpv : 256) * vec * sizeof(float); }

template <typename T, int VEC, int U, bool RES>
__device__ __forceinline__ void fwd_load(const T* y, int y_ld, const T* res, int res_ld, int c, long m, long us, long mend, float (&v)[U][VEC], float (&r)[U][VEC]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long mm = m + u * us < mend ? m + u * us : mend - 1;     // clamped: loads stay in bounds, stores are predicated
    vload<T, VEC>(y + mm * y_ld + c, v[u]);
    if (RES) vload<T, VEC>(res + mm * res_ld + c, r[u]);
  }
}

template <typename T, int VEC, bool SILU, bool RES>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(long M, int C, const T* __restrict__ y, int y_ld, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const T* __restrict__ res, int res_ld,
                                                         T* __restrict__ z, int z_ld, int cpv, int rows_pb, const RowWalk w) {
  constexpr int U = 4;                                   // rows in flight per thread (memory-level parallelism)
  const RowLane l = row_lane<VEC>(cpv, rows_pb);
  float* s_sc = s_coef;
  float* s_sh = s_coef + l.cwv;
  for (int e = threadIdx.x; e < l.nch; e += 256) {
    s_sc[e] = scale ? scale[l.cbase + e] : 1.f;
    s_sh[e] = shift ? shift[l.cbase + e] : 0.f;
  }
  const long mend = walk_end(w, M);
  long m = (long)blockIdx.x * w.bs + l.rsub;
  float v[U][VEC], r[U][VEC];
  if (l.active && m < mend) fwd_load<T, VEC, U, RES>(y, y_ld, res, res_ld, l.c, m, w.us, mend, v, r);
  __syncthreads();
  if (!l.active) return;
  float sc[VEC], sh[VEC];
  coef_get<VEC>(s_sc, l.cl * VEC, sc);
  coef_get<VEC>(s_sh, l.cl * VEC, sh);
  while (m < mend) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float t = v[u][i] * sc[i] + sh[i];
        if (SILU) t = silu_f(t);
        if (RES) t += r[u][i];
        v[u][i] = t;
      }
      if (m + u * w.us < mend) vstore<T, VEC>(z + (m + u * w.us) * z_ld + l.c, v[u]);
    }
    m += w.ts;
    if (m < mend) fwd_load<T, VEC, U, RES>(y, y_ld, res, res_ld, l.c, m, w.us, mend, v, r);
  }
}

#define SY11_BNF(VV, SS, RR) hipLaunchKernelGGL((bn_act_fwd_kernel<T, VV, SS, RR>), grid, block, coef_bytes(g, VV, 2), st, (long)M, C, (const T*)y, y_ld, scale, shift, (const T*)res, res_ld, (T*)z, z_ld, g.cpv, g.rows_pb, w)
extern "C" int sy11_bn_act_fwd(int32_t dtype, int64_t M, int32_t C, const void* y, int32_t y_ld, const float* scale,
                               const float* shift, int32_t silu, const void* res, int32_t res_ld, void* z, int32_t z_ld,
                               void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && M > 0 && C > 0 && y && z, "bn_act_fwd: bad argument");
  SY11_REQUIRE(y_ld >= C && z_ld >= C && (!res || res_ld >= C), "bn_act_fwd: pixel stride < C");
  const int esz = dtype_size(dtype);
  const bool v = vec_ok(esz, C, {y_ld, z_ld, res ? res_ld : 16}, {y, z, res});
  const RowGeom g = row_geom(C, v ? 16 / esz : 1);
  const RowWalk w = row_walk(M, g.rows_pb, 4, 1, 1L << 20);
  dim3 grid(w.grid, g.cblocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  SY11_DISPATCH_DTYPE(dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (v) { if (silu) { if (res) SY11_BNF(VE, true, true); else SY11_BNF(VE, true, false); } else { if (res) SY11_BNF(VE, false, true); else SY11_BNF(VE, false, false); } }
    else { if (silu) { if (res) SY11_BNF(1, true, true); else SY11_BNF(1, true, false); } else { if (res) SY11_BNF(1, false, true); else SY11_BNF(1, false, false); } }
  });
  SY11_LAUNCH_CHECK("bn_act_fwd");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ BN backward
template <typename T, int VEC, int U>
__device__ __forceinline__ void bwd_load(const T* y, int y_ld, const T* dz, int dz_ld, int c, long m, long us, long mend, float (&vy)[U][VEC], float (&vg)[U][VEC]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const long mm = m + u * us < mend ? m + u * us : mend - 1;
    vload<T, VEC>(y + mm * y_ld + c, vy[u]);
    vload<T, VEC>(dz + mm * dz_ld + c, vg[u]);
  }
}

// pass 1: per-channel sums of g = dz*act'(u), u = y*scale+shift, and of g*xhat, xhat = (y-mean)*rstd (rstd applied once per block)
template <typename T, int VEC, bool SILU>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(long M, int C, const T* __restrict__ y, int y_ld, const T* __restrict__ dz, int dz_ld,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            float* sum_g, float* sum_gx, int cpv, int rows_pb, int slots, const RowWalk w) {
  constexpr int U = 4;
  __shared__ float red[2][256][VEC > 1 ? VEC : 1];
  const RowLane l = row_lane<VEC>(cpv, rows_pb);
  float* s_sc = s_coef;
  float* s_sh = s_coef + l.cwv;
  float* s_mu = s_coef + 2 * l.cwv;
  for (int e = threadIdx.x; e < l.nch; e += 256) {
    s_sc[e] = scale[l.cbase + e];
    s_sh[e] = shift[l.cbase + e];
    s_mu[e] = mean[l.cbase + e];
  }
  const long mend = walk_end(w, M);
  long m = (long)blockIdx.x * w.bs + l.rsub;
  float vy[U][VEC], vg[U][VEC];
  if (l.active && m < mend) bwd_load<T, VEC, U>(y, y_ld, dz, dz_ld, l.c, m, w.us, mend, vy, vg);
  __syncthreads();
  float sg[VEC], sgx[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) sg[i] = sgx[i] = 0.f;
  if (l.active) {
    float mu[VEC], sc[VEC], sh[VEC];
    coef_get<VEC>(s_mu, l.cl * VEC, mu);
    coef_get<VEC>(s_sc, l.cl * VEC, sc);
    coef_get<VEC>(s_sh, l.cl * VEC, sh);
    while (m < mend) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float live = (m + u * w.us < mend) ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          float g = vg[u][i] * live;
          if (SILU) g *= dsilu_f(vy[u][i] * sc[i] + sh[i]);
          sg[i] += g;
          sgx[i] += g * (vy[u][i] - mu[i]);
        }
      }
      m += w.ts;
      if (m < mend) bwd_load<T, VEC, U>(y, y_ld, dz, dz_ld, l.c, m, w.us, mend, vy, vg);
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) { red[0][threadIdx.x][i] = sg[i]; red[1][threadIdx.x][i] = sgx[i]; }
  __syncthreads();
  // tree reduction over the row groups (all threads take part; rows_pb need not be a power of two)
  const int cw = l.cwv / VEC;
  int half = 1;
  while (half < rows_pb) half <<= 1;
  for (half >>= 1; half >= 1; half >>= 1) {
    if (l.rsub < half && l.rsub + half < rows_pb) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        red[0][threadIdx.x][i] += red[0][threadIdx.x + half * cw][i];
        red[1][threadIdx.x][i] += red[1][threadIdx.x + half * cw][i];
      }
    }
    __syncthreads();
  }
  // dense atomics: lane t adds channel (cbase + t) -> one 256-byte request group per 64 lanes instead of VEC-strided scalars
  // (same-line atomic REQUESTS, not bytes, are what the memory side serialises)
  // ordered mode: one row per workgroup x (slots == gridDim.x), every element of it written by exactly one workgroup (x, y): plain
  // stores, and the host skips the zero fill
  const bool exclusive = slots == (int)gridDim.x && slots > 1;
  for (int e = threadIdx.x; e < l.nch; e += 256) {
    const int cl2 = e / VEC, i2 = e - cl2 * VEC;
    const long ch = (long)(blockIdx.x % slots) * C + l.cbase + e;
    const float g0 = red[0][cl2][i2], g1 = red[1][cl2][i2] * rstd[l.cbase + e];
    if (exclusive) { sum_g[ch] = g0; sum_gx[ch] = g1; }
    else { atomicAdd(sum_g + ch, g0); atomicAdd(sum_gx + ch, g1); }
  }
}

#define SY11_BNR(VV, SS) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, VV, SS>), grid, block, coef_bytes(g, VV, 3), st, (long)M, C, (const T*)y, y_ld, (const T*)dz, dz_ld, mean, rstd, scale, shift, sum_g, sum_gx, g.cpv, g.rows_pb, slots, w)
extern "C" int sy11_bn_act_bwd_reduce(int32_t dtype, int64_t M, int32_t C, const void* y, int32_t y_ld, const void* dz,
                                      int32_t dz_ld, const float* mean, const float* rstd, const float* scale,
                                      const float* shift, int32_t silu, float* sum_g, float* sum_gx, int32_t sum_slots,
                                      void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && M > 0 && C > 0 && y && dz && mean && rstd && scale && shift && sum_g && sum_gx, "bn_act_bwd_reduce: bad argument");
  SY11_REQUIRE(y_ld >= C && dz_ld >= C, "bn_act_bwd_reduce: pixel stride < C");
  const int esz = dtype_size(dtype);
  const bool v = vec_ok(esz, C, {y_ld, dz_ld}, {y, dz});
  const RowGeom g = row_geom(C, v ? 16 / esz : 1);
  // small maps: one trip of 4 rows per thread (a second, dependent trip costs a memory round trip: 7.4 -> 5.7 us); larger ones >= 8 rows;
  // at most 4096 blocks (each ends in a tree reduction and 2 x C atomics)
  const long nb4 = (M + g.rows_pb * 4L - 1) / (g.rows_pb * 4L);
  const RowWalk w = row_walk(M, g.rows_pb, 4, nb4 > 640 ? 2 : 1, 4096);
  int slots = sum_slots > 1 ? sum_slots : 1;
  dim3 grid(w.grid, g.cblocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  DetPartials dp;                                   // ordered mode (det.h): one partial row per workgroup x, folded into row 0 of the caller's slots
  float* const sum_g_out = sum_g;
  float* const sum_gx_out = sum_gx;
  const bool det = sy11_det(4) && w.grid > 1;
  if (det) {
    if (!dp.acquire(st, 2, w.grid, C, false)) SY11_FAIL(SY11_ELAUNCH, "bn_act_bwd_reduce: ordered-reduction workspace unavailable");   // rows are stored whole: no zero fill
    sum_g = dp.buf(0); sum_gx = dp.buf(1); slots = w.grid;
  }
  SY11_DISPATCH_DTYPE(dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (v) { if (silu) SY11_BNR(VE, true); else SY11_BNR(VE, false); }
    else { if (silu) SY11_BNR(1, true); else SY11_BNR(1, false); }
  });
  SY11_LAUNCH_CHECK("bn_act_bwd_reduce");
  if (det) {
    return dp.fold01(sum_g_out, sum_gx_out, sum_slots > 1 ? sum_slots : 1, C);
  }
  return SY11_OK;
}

// pass 2: dy = gamma*rstd*(g - sum_g/M - xhat*sum_gx/M) = k0*g - k1 - (y - mean)*k2; block (0,*) also accumulates dgamma/dbeta
// RES: the layer's output was also a residual sum (Bottleneck shortcut, block.py:725): the gradient of the residual operand is dz
// itself — written (or added, f32 add, one rounding: what sy11_copy2d's accumulate does) from the dz values this pass holds anyway,
// instead of a separate three-pass copy launch
template <typename T, int VEC, bool SILU, bool RES>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(long M, int C, const T* __restrict__ y, int y_ld, const T* __restrict__ dz, int dz_ld,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ gamma, const float* __restrict__ sum_g,
                                                           const float* __restrict__ sum_gx, T* __restrict__ dy, int dy_ld, float* dgamma,
                                                           float* dbeta, int cpv, int rows_pb, int slots, const RowWalk w, T* resg, int resg_ld,
                                                           int resg_acc) {
  constexpr int U = 4;
  const RowLane l = row_lane<VEC>(cpv, rows_pb);
  float* s_sc = s_coef;
  float* s_sh = s_coef + l.cwv;
  float* s_mu = s_coef + 2 * l.cwv;
  float* s_k0 = s_coef + 3 * l.cwv;
  float* s_k1 = s_coef + 4 * l.cwv;
  float* s_k2 = s_coef + 5 * l.cwv;
  const long mend = walk_end(w, M);
  long m = (long)blockIdx.x * w.bs + l.rsub;
  float vy[U][VEC], vg[U][VEC];
  if (l.active && m < mend) bwd_load<T, VEC, U>(y, y_ld, dz, dz_ld, l.c, m, w.us, mend, vy, vg);
  {
    // channel e of this block's range per thread: fold the partial-sum slots, derive the three coefficients
    const float invM = 1.0f / (float)M;
    for (int e = threadIdx.x; e < l.nch; e += 256) {
      const int ch = l.cbase + e;
      float tg = 0.f, tgx = 0.f;
      // eight slots' loads in flight at a time (a `for k < slots` loop waits for every pair before issuing the next: on the
      // 20x20 / 40x40 maps those 8 dependent round trips were most of a workgroup's life); summed in slot order
      for (int k0 = 0; k0 < slots; k0 += 8) {
        float a8[8], b8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int k = k0 + j < slots ? k0 + j : slots - 1;
          a8[j] = sum_g[(long)k * C + ch];
          b8[j] = sum_gx[(long)k * C + ch];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (k0 + j < slots) { tg += a8[j]; tgx += b8[j]; }
      }
      if (blockIdx.x == 0 && dgamma) { atomicAdd(dgamma + ch, tgx); atomicAdd(dbeta + ch, tg); }
      const float rs = rstd[ch], gr = gamma[ch] * rs;
      s_sc[e] = scale[ch];
      s_sh[e] = shift[ch];
      s_mu[e] = mean[ch];
      s_k0[e] = gr;
      s_k1[e] = gr * tg * invM;
      s_k2[e] = gr * tgx * invM * rs;
    }
  }
  __syncthreads();
  if (!l.active) return;
  float mu[VEC], sc[VEC], sh[VEC], k0[VEC], k1[VEC], k2[VEC];
  coef_get<VEC>(s_mu, l.cl * VEC, mu);
  coef_get<VEC>(s_sc, l.cl * VEC, sc);
  coef_get<VEC>(s_sh, l.cl * VEC, sh);
  coef_get<VEC>(s_k0, l.cl * VEC, k0);
  coef_get<VEC>(s_k1, l.cl * VEC, k1);
  coef_get<VEC>(s_k2, l.cl * VEC, k2);
  while (m < mend) {
    if (RES) {
      float vr[U][VEC];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long mm = m + u * w.us < mend ? m + u * w.us : mend - 1;
#pragma unroll
        for (int i = 0; i < VEC; ++i) vr[u][i] = 0.f;
        if (resg_acc) vload<T, VEC>(resg + mm * resg_ld + l.c, vr[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) vr[u][i] += vg[u][i];
        if (m + u * w.us < mend) vstore<T, VEC>(resg + (m + u * w.us) * resg_ld + l.c, vr[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float g = vg[u][i];
        if (SILU) g *= dsilu_f(vy[u][i] * sc[i] + sh[i]);
        vg[u][i] = k0[i] * g - k1[i] - (vy[u][i] - mu[i]) * k2[i];
      }
      if (m + u * w.us < mend) vstore<T, VEC>(dy + (m + u * w.us) * dy_ld + l.c, vg[u]);
    }
    m += w.ts;
    if (m < mend) bwd_load<T, VEC, U>(y, y_ld, dz, dz_ld, l.c, m, w.us, mend, vy, vg);
  }
}

#define SY11_BNA(VV, SS)                                                                                                                         \
  do {                                                                                                                                          \
    if (res_grad) hipLaunchKernelGGL((bn_bwd_apply_kernel<T, VV, SS, true>), grid, block, coef_bytes(g, VV, 6), st, (long)M, C, (const T*)y, y_ld, (const T*)dz, dz_ld, mean, rstd, scale, shift, gamma, sum_g, sum_gx, (T*)dy, dy_ld, dgamma, dbeta, g.cpv, g.rows_pb, sum_slots > 1 ? sum_slots : 1, w, (T*)res_grad, res_ld, res_accumulate); \
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<T, VV, SS, false>), grid, block, coef_bytes(g, VV, 6), st, (long)M, C, (const T*)y, y_ld, (const T*)dz, dz_ld, mean, rstd, scale, shift, gamma, sum_g, sum_gx, (T*)dy, dy_ld, dgamma, dbeta, g.cpv, g.rows_pb, sum_slots > 1 ? sum_slots : 1, w, (T*)nullptr, 0, 0); \
  } while (0)
extern "C" int sy11_bn_act_bwd_apply_res(int32_t dtype, int64_t M, int32_t C, const void* y, int32_t y_ld, const void* dz,
                                         int32_t dz_ld, const float* mean, const float* rstd, const float* scale,
                                         const float* shift, const float* gamma, int32_t silu, const float* sum_g,
                                         const float* sum_gx, int32_t sum_slots, void* dy, int32_t dy_ld, float* dgamma, float* dbeta,
                                         void* res_grad, int32_t res_ld, int32_t res_accumulate, void* stream);
extern "C" int sy11_bn_act_bwd_apply(int32_t dtype, int64_t M, int32_t C, const void* y, int32_t y_ld, const void* dz,
                                     int32_t dz_ld, const float* mean, const float* rstd, const float* scale,
                                     const float* shift, const float* gamma, int32_t silu, const float* sum_g,
                                     const float* sum_gx, int32_t sum_slots, void* dy, int32_t dy_ld, float* dgamma, float* dbeta,
                                     void* stream) {
  return sy11_bn_act_bwd_apply_res(dtype, M, C, y, y_ld, dz, dz_ld, mean, rstd, scale, shift, gamma, silu, sum_g, sum_gx, sum_slots, dy, dy_ld,
                                   dgamma, dbeta, nullptr, 0, 0, stream);
}
extern "C" int sy11_bn_act_bwd_apply_res(int32_t dtype, int64_t M, int32_t C, const void* y, int32_t y_ld, const void* dz,
                                         int32_t dz_ld, const float* mean, const float* rstd, const float* scale,
                                         const float* shift, const float* gamma, int32_t silu, const float* sum_g,
                                         const float* sum_gx, int32_t sum_slots, void* dy, int32_t dy_ld, float* dgamma, float* dbeta,
                                         void* res_grad, int32_t res_ld, int32_t res_accumulate, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && M > 0 && C > 0 && y && dz && dy && mean && rstd && scale && shift && gamma && sum_g && sum_gx, "bn_act_bwd_apply: bad argument");
  SY11_REQUIRE(!res_grad || (res_ld >= C && res_grad != dz && res_grad != dy), "bn_act_bwd_apply: residual gradient needs its own buffer and a pixel stride >= C");
  SY11_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), "bn_act_bwd_apply: dgamma/dbeta both or neither");
  SY11_REQUIRE(y_ld >= C && dz_ld >= C && dy_ld >= C, "bn_act_bwd_apply: pixel stride < C");
  const int esz = dtype_size(dtype);
  const bool v = vec_ok(esz, C, {y_ld, dz_ld, dy_ld, res_grad ? res_ld : y_ld}, {y, dz, dy, res_grad ? res_grad : y});
  const RowGeom g = row_geom(C, v ? 16 / esz : 1);
  // the slot fold of the prologue is paid per block (C x slots x 2 loads): two trips per block on the big maps; on small maps with
  // many channels (20x20x512: the fold reads as many bytes as a two-trip block streams) four or eight
  static int trips_env = -1;
  if (trips_env < 0) { const char* e = getenv("SY11_BN_APPLY_TRIPS"); trips_env = e ? atoi(e) : 0; }
  int trips = (C >= 512 && M <= 65536) ? 4 : 2;      // r03 sweep (graph replay, us): 20x20x512 26.1 / 18.9 / 17.4 / 19.0 at 2 / 3 / 4 / 8 trips;
  if (trips_env > 0) trips = trips_env;              // 20x20x256 10.1 / 11.4 / 10.7 / 15.5, 40x40x256 and larger: flat or worse
  const RowWalk w = row_walk(M, g.rows_pb, 4, trips, 1L << 20);
  dim3 grid(w.grid, g.cblocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  SY11_DISPATCH_DTYPE(dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (v) { if (silu) SY11_BNA(VE, true); else SY11_BNA(VE, false); }
    else { if (silu) SY11_BNA(1, true); else SY11_BNA(1, false); }
  });
  SY11_LAUNCH_CHECK("bn_act_bwd_apply");
  return SY11_OK;
}

static inline int row_grid(long M, int rows_pb) {
  long g = (M + rows_pb * 4L - 1) / (rows_pb * 4L);        // 4 rows per thread per trip
  const long cap = 256L * 8;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// ------------------------------------------------------------------------------------------------ strided copy / accumulate
template <typename T, int VEC>
__global__ __launch_bounds__(256) void copy2d_kernel(long M, int C, const T* __restrict__ src, int src_ld, T* __restrict__ dst, int dst_ld,
                                                     int accumulate, int cpv, int rows_pb, const RowWalk w) {
  constexpr int U = 4;
  const int cw = cpv < 256 ? cpv : 256;
  const int cv = blockIdx.y * 256 + (threadIdx.x % cw);
  const int rsub = threadIdx.x / cw;
  if (cv >= cpv || rsub >= rows_pb) return;
  const int c = cv * VEC;
  const long mend = walk_end(w, M);
  for (long m = (long)blockIdx.x * w.bs + rsub; m < mend; m += w.ts) {
    float v[U][VEC], o[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long mm = m + u * w.us < mend ? m + u * w.us : mend - 1;
      vload<T, VEC>(src + mm * src_ld + c, v[u]);
      if (accumulate) vload<T, VEC>(dst + mm * dst_ld + c, o[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (accumulate) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[u][i] += o[u][i];
      }
      if (m + u * w.us < mend) vstore<T, VEC>(dst + (m + u * w.us) * dst_ld + c, v[u]);
    }
  }
}

extern "C" int sy11_copy2d(int32_t dtype, int64_t M, int32_t C, const void* src, int32_t src_ld, void* dst, int32_t dst_ld,
                           int32_t accumulate, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && M > 0 && C > 0 && src && dst, "copy2d: bad argument");
  SY11_REQUIRE(src_ld >= C && dst_ld >= C, "copy2d: pixel stride < C");
  const int esz = dtype_size(dtype);
  const bool v = vec_ok(esz, C, {src_ld, dst_ld}, {src, dst});
  const RowGeom g = row_geom(C, v ? 16 / esz : 1);
  const RowWalk w = row_walk(M, g.rows_pb, 4, 1, 1L << 20);
  dim3 grid(w.grid, g.cblocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  SY11_DISPATCH_DTYPE(dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (v) hipLaunchKernelGGL((copy2d_kernel<T, VE>), grid, block, 0, st, (long)M, C, (const T*)src, src_ld, (T*)dst, dst_ld, accumulate, g.cpv, g.rows_pb, w);
    else hipLaunchKernelGGL((copy2d_kernel<T, 1>), grid, block, 0, st, (long)M, C, (const T*)src, src_ld, (T*)dst, dst_ld, accumulate, g.cpv, g.rows_pb, w);
  });
  SY11_LAUNCH_CHECK("copy2d");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ nearest 2x upsample
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(int B, int H, int W, int C, const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld,
                                                             int cpv, int rows_pb) {
  const int cw = cpv < 256 ? cpv : 256;
  const int cv = blockIdx.y * 256 + (threadIdx.x % cw);
  const int rsub = threadIdx.x / cw;
  if (cv >= cpv || rsub >= rows_pb) return;
  const int c = cv * VEC;
  const long M = (long)B * H * W;
#pragma unroll 4
  for (long m = (long)blockIdx.x * rows_pb + rsub; m < M; m += (long)gridDim.x * rows_pb) {
    const unsigned mi = (unsigned)m;                     // B*H*W < 2^31 (checked by the host): 32-bit division, not the emulated 64-bit one
    const unsigned t = mi / (unsigned)W;
    const int w = (int)(mi - t * (unsigned)W);
    const int b = (int)(t / (unsigned)H), h = (int)(t - (unsigned)b * (unsigned)H);
    float v[VEC];
    vload<T, VEC>(x + m * x_ld + c, v);
    const long o = ((long)(b * 2 * H + 2 * h) * (2 * W) + 2 * w);
    vstore<T, VEC>(y + o * y_ld + c, v);
    vstore<T, VEC>(y + (o + 1) * y_ld + c, v);
    vstore<T, VEC>(y + (o + 2 * W) * y_ld + c, v);
    vstore<T, VEC>(y + (o + 2 * W + 1) * y_ld + c, v);
  }
}
template <typename T, int VEC>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(int B, int H, int W, int C, const T* __restrict__ dy, int dy_ld, T* __restrict__ dx, int dx_ld,
                                                             int accumulate, int cpv, int rows_pb) {
  const int cw = cpv < 256 ? cpv : 256;
  const int cv = blockIdx.y * 256 + (threadIdx.x % cw);
  const int rsub = threadIdx.x / cw;
  if (cv >= cpv || rsub >= rows_pb) return;
  const int c = cv * VEC;
  const long M = (long)B * H * W;
#pragma unroll 4
  for (long m = (long)blockIdx.x * rows_pb + rsub; m < M; m += (long)gridDim.x * rows_pb) {
    const unsigned mi = (unsigned)m;                     // B*H*W < 2^31 (checked by the host): 32-bit division, not the emulated 64-bit one
    const unsigned t = mi / (unsigned)W;
    const int w = (int)(mi - t * (unsigned)W);
    const int b = (int)(t / (unsigned)H), h = (int)(t - (unsigned)b * (unsigned)H);
    const long o = ((long)(b * 2 * H + 2 * h) * (2 * W) + 2 * w);
    float a[VEC], q[VEC];
    vload<T, VEC>(dy + o * dy_ld + c, a);
    vload<T, VEC>(dy + (o + 1) * dy_ld + c, q);
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[i] += q[i];
    vload<T, VEC>(dy + (o + 2 * W) * dy_ld + c, q);
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[i] += q[i];
    vload<T, VEC>(dy + (o + 2 * W + 1) * dy_ld + c, q);
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[i] += q[i];
    if (accumulate) {
      vload<T, VEC>(dx + m * dx_ld + c, q);
#pragma unroll
      for (int i = 0; i < VEC; ++i) a[i] += q[i];
    }
    vstore<T, VEC>(dx + m * dx_ld + c, a);
  }
}

extern "C" int sy11_upsample2x_fwd(int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, const void* x, int32_t x_ld,
                                   void* y, int32_t y_ld, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0 && x && y && x_ld >= C && y_ld >= C, "upsample2x_fwd: bad argument");
  SY11_REQUIRE((long)B * H * W < (1L << 31), "upsample2x_fwd: too many pixels");
  const int esz = dtype_size(dtype);
  const bool v = vec_ok(esz, C, {x_ld, y_ld}, {x, y});
  const RowGeom g = row_geom(C, v ? 16 / esz : 1);
  dim3 grid(row_grid((long)B * H * W, g.rows_pb), g.cblocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  SY11_DISPATCH_DTYPE(dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (v) hipLaunchKernelGGL((upsample2x_fwd_kernel<T, VE>), grid, block, 0, st, B, H, W, C, (const T*)x, x_ld, (T*)y, y_ld, g.cpv, g.rows_pb);
    else hipLaunchKernelGGL((upsample2x_fwd_kernel<T, 1>), grid, block, 0, st, B, H, W, C, (const T*)x, x_ld, (T*)y, y_ld, g.cpv, g.rows_pb);
  });
  SY11_LAUNCH_CHECK("upsample2x_fwd");
  return SY11_OK;
}
extern "C" int sy11_upsample2x_bwd(int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, const void* dy, int32_t dy_ld,
                                   void* dx, int32_t dx_ld, int32_t accumulate, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0 && dy && dx && dy_ld >= C && dx_ld >= C, "upsample2x_bwd: bad argument");
  SY11_REQUIRE((long)B * H * W < (1L << 31), "upsample2x_bwd: too many pixels");
  const int esz = dtype_size(dtype);
  const bool v = vec_ok(esz, C, {dy_ld, dx_ld}, {dy, dx});
  const RowGeom g = row_geom(C, v ? 16 / esz : 1);
  dim3 grid(row_grid((long)B * H * W, g.rows_pb), g.cblocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  SY11_DISPATCH_DTYPE(dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (v) hipLaunchKernelGGL((upsample2x_bwd_kernel<T, VE>), grid, block, 0, st, B, H, W, C, (const T*)dy, dy_ld, (T*)dx, dx_ld, accumulate, g.cpv, g.rows_pb);
    else hipLaunchKernelGGL((upsample2x_bwd_kernel<T, 1>), grid, block, 0, st, B, H, W, C, (const T*)dy, dy_ld, (T*)dx, dx_ld, accumulate, g.cpv, g.rows_pb);
  });
  SY11_LAUNCH_CHECK("upsample2x_bwd");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ 5x5 stride-1 max pool
// Both kernels are branch-free over the 25 window positions: a position outside the map loads a clamped (always legal) pixel and is
// neutralised arithmetically, so the 5 loads of a window row go out back to back instead of one branch-and-wait per position.
// ATen's rule: the first maximum in row-major window order wins, NaN propagates.
template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool5_fwd_kernel(int B, int H, int W, int C, const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld,
                                                           uint8_t* __restrict__ idx, int cpv, int rows_pb) {
  const int cw = cpv < 256 ? cpv : 256;
  const int cv = blockIdx.y * 256 + (threadIdx.x % cw);
  const int rsub = threadIdx.x / cw;
  if (cv >= cpv || rsub >= rows_pb) return;
  const int c = cv * VEC;
  const long M = (long)B * H * W;
  for (long m = (long)blockIdx.x * rows_pb + rsub; m < M; m += (long)gridDim.x * rows_pb) {
    const unsigned mi = (unsigned)m;                     // B*H*W < 2^31 (checked by the host): 32-bit division, not the emulated 64-bit one
    const unsigned t = mi / (unsigned)W;
    const int w = (int)(mi - t * (unsigned)W);
    const int h = (int)(t % (unsigned)H);
    // legal window rows / columns as bit masks, tested where a position is USED: 25 long-lived lane masks spill the scalar registers
    unsigned rmask = 0, cmask = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      rmask |= ((unsigned)(h + k - 2) < (unsigned)H ? 1u : 0u) << k;
      cmask |= ((unsigned)(w + k - 2) < (unsigned)W ? 1u : 0u) << k;
    }
    const T* xc = x + (m - (long)h * W - w) * x_ld + c;        // pixel (0, 0) of this image
    float best[VEC];
    int bi[VEC];
    if constexpr (sizeof(T) == 2) {
      // 16-bit inputs: value and window position share ONE sortable integer.  A 16-bit float widened to f32 has >= 13 zero low bits;
      // ord = bits ^ ((bits >> 31) & 0x7fffffe0) orders like the float (sign-magnitude -> two's complement) and keeps the low 5 bits
      // free for 25 - pos (earlier position = larger key = wins a tie, ATen's rule).  An illegal position loads a clamped duplicate of
      // a legal pixel with low bits 0, so it loses the tie against its twin: no per-element masks at all (the compare-and-select form
      // keeps 200 lane masks alive and spills the scalar registers).  +0 outranks -0 (ATen: whichever comes first).
      int key[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) key[i] = (int)0x80000000;
#pragma unroll
      for (int wy = 0; wy < 5; ++wy) {
        const int iy = min(max(h + wy - 2, 0), H - 1);         // clamped: always a legal address
        float v[5][VEC];
#pragma unroll
        for (int wx = 0; wx < 5; ++wx) {
          const int ix = min(max(w + wx - 2, 0), W - 1);
          vload<T, VEC>(xc + ((long)iy * W + ix) * x_ld, v[wx]);
        }
#pragma unroll
        for (int wx = 0; wx < 5; ++wx) {
          const int low = (int)((rmask >> wy) & (cmask >> wx) & 1u) * (25 - (wy * 5 + wx));
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            const int bits = __builtin_bit_cast(int, v[wx][i]);
            const int k = (bits ^ ((bits >> 31) & 0x7fffffe0)) + low;
            key[i] = k > key[i] ? k : key[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        bi[i] = 25 - (key[i] & 31);
        const int o = key[i] & ~31;
        best[i] = __builtin_bit_cast(float, o ^ ((o >> 31) & 0x7fffffe0));
      }
    } else {
      const int y0 = max(0, 2 - h), x0 = max(0, 2 - w);        // first legal position in row-major order: taken unconditionally
      vload<T, VEC>(xc + ((long)(h + y0 - 2) * W + (w + x0 - 2)) * x_ld, best);
#pragma unroll
      for (int i = 0; i < VEC; ++i) bi[i] = y0 * 5 + x0;
#pragma unroll
      for (int wy = 0; wy < 5; ++wy) {
        const int iy = min(max(h + wy - 2, 0), H - 1);
        float v[5][VEC];
#pragma unroll
        for (int wx = 0; wx < 5; ++wx) {
          const int ix = min(max(w + wx - 2, 0), W - 1);
          vload<T, VEC>(xc + ((long)iy * W + ix) * x_ld, v[wx]);
        }
#pragma unroll
        for (int wx = 0; wx < 5; ++wx) {
          const bool ok = ((rmask >> wy) & (cmask >> wx) & 1u) != 0;
#pragma unroll
          for (int i = 0; i < VEC; ++i)
            if (ok && v[wx][i] > best[i]) { best[i] = v[wx][i]; bi[i] = wy * 5 + wx; }   // strict >: first maximum wins
        }
      }
    }
    vstore<T, VEC>(y + m * y_ld + c, best);
    if (idx) {
      if constexpr (VEC == 8) {
        uint2 pk;
        pk.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
        pk.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
        *(uint2*)(idx + m * C + c) = pk;                // C % 8 == 0 and c % 8 == 0 on the vector path
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) idx[m * C + c + i] = (uint8_t)bi[i];
      }
    }
  }
}
// Separable form for 16-bit inputs with 16-byte channel vectors (r03): the (value, window position) key of maxpool5_fwd_kernel is a
// maximum of integers, so the 5 x 5 window splits into a row pass (5 keys per pixel, low bits 5 - wx) and a column pass over five
// row results (+ 5 (4 - wy)): 10 key operations and — with a thread walking R consecutive rows of one column — (R + 4) x 5 / R loads
// per output instead of 25 and 25.  Out-of-image positions load a clamped duplicate whose key has smaller low bits than its legal
// twin (0 for an illegal row, 5 (4 - wy) for an illegal column in a legal row), so they never win: the result key, hence value and
// argmax, is bit-identical to the 25-way form.
template <typename T, int R>
__global__ __launch_bounds__(256) void maxpool5_fwd_sep_kernel(int B, int H, int W, int C, const T* __restrict__ x, int x_ld, T* __restrict__ y, int y_ld,
                                                               uint8_t* __restrict__ idx, int cpv, int bands, long total) {
  constexpr int VEC = 8;
  static_assert(sizeof(T) == 2, "16-bit inputs");
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int cv = (int)(gid % cpv);
  long q = gid / cpv;
  const int w = (int)(q % W); q /= W;
  const int band = (int)(q % bands);
  const int b = (int)(q / bands);
  const int h0 = band * R, c = cv * VEC;
  const T* xb = x + ((long)b * H * W) * x_ld + c;
  unsigned cmask = 0;
  int ixs[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    cmask |= ((unsigned)(w + k - 2) < (unsigned)W ? 1u : 0u) << k;
    ixs[k] = min(max(w + k - 2, 0), W - 1);
  }
  auto row_pass = [&](int r, int (&kr)[VEC]) {            // r: window row index relative to h0 - 2
    const int iy = min(max(h0 - 2 + r, 0), H - 1);
    float v[5][VEC];
#pragma unroll
    for (int wx = 0; wx < 5; ++wx) vload<T, VEC>(xb + ((long)iy * W + ixs[wx]) * x_ld, v[wx]);
#pragma unroll
    for (int i = 0; i < VEC; ++i) kr[i] = (int)0x80000000;
#pragma unroll
    for (int wx = 0; wx < 5; ++wx) {
      const int low = (int)((cmask >> wx) & 1u) * (5 - wx);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const int bits = __builtin_bit_cast(int, v[wx][i]);
        const int k = (bits ^ ((bits >> 31) & 0x7fffffe0)) + low;
        kr[i] = k > kr[i] ? k : kr[i];
      }
    }
  };
  int kr[R + 4][VEC];
#pragma unroll
  for (int r = 0; r < R + 4; ++r) row_pass(r, kr[r]);
#pragma unroll
  for (int o = 0; o < R; ++o) {
    const int h = h0 + o;
    if (h >= H) break;
    int key[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) key[i] = (int)0x80000000;
#pragma unroll
    for (int wy = 0; wy < 5; ++wy) {
      const bool rok = (unsigned)(h + wy - 2) < (unsigned)H;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const int k = rok ? kr[o + wy][i] + 5 * (4 - wy) : (kr[o + wy][i] & ~31);
        key[i] = k > key[i] ? k : key[i];
      }
    }
    float best[VEC];
    unsigned bi[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      bi[i] = (unsigned)(25 - (key[i] & 31));
      const int ob = key[i] & ~31;
      best[i] = __builtin_bit_cast(float, ob ^ ((ob >> 31) & 0x7fffffe0));
    }
    const long m = ((long)b * H + h) * W + w;
    vstore<T, VEC>(y + m * y_ld + c, best);
    if (idx) {
      uint2 pk;
      pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
      pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
      *(uint2*)(idx + m * C + c) = pk;
    }
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(256) void maxpool5_bwd_kernel(int B, int H, int W, int C, const T* __restrict__ dy, int dy_ld,
                                                           const uint8_t* __restrict__ idx, T* __restrict__ dx, int dx_ld, int accumulate,
                                                           int cpv, int rows_pb) {
  const int cw = cpv < 256 ? cpv : 256;
  const int cv = blockIdx.y * 256 + (threadIdx.x % cw);
  const int rsub = threadIdx.x / cw;
  if (cv >= cpv || rsub >= rows_pb) return;
  const int c = cv * VEC;
  const long M = (long)B * H * W;
  for (long m = (long)blockIdx.x * rows_pb + rsub; m < M; m += (long)gridDim.x * rows_pb) {
    const unsigned mi = (unsigned)m;                     // B*H*W < 2^31 (checked by the host): 32-bit division, not the emulated 64-bit one
    const unsigned t = mi / (unsigned)W;
    const int w = (int)(mi - t * (unsigned)W);
    const int h = (int)(t % (unsigned)H);
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    if (accumulate) vload<T, VEC>(dx + m * dx_ld + c, acc);
    // every window q = p + (oy, ox) that contains p; p sits at window position (2-oy, 2-ox) of q
#pragma unroll
    for (int oy = -2; oy <= 2; ++oy) {
      const bool yok = (unsigned)(h + oy) < (unsigned)H;
      float g[5][VEC];
      uint8_t pi[5][VEC];
#pragma unroll
      for (int ox = -2; ox <= 2; ++ox) {
        const bool ok = yok && (unsigned)(w + ox) < (unsigned)W;
        const long q = m + (ok ? (long)oy * W + ox : 0);
        vload<T, VEC>(dy + q * dy_ld + c, g[ox + 2]);
        if constexpr (VEC == 8) {
          const uint2 pk = *(const uint2*)(idx + q * C + c);
#pragma unroll
          for (int i = 0; i < 4; ++i) { pi[ox + 2][i] = (uint8_t)(pk.x >> (8 * i)); pi[ox + 2][4 + i] = (uint8_t)(pk.y >> (8 * i)); }
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) pi[ox + 2][i] = idx[q * C + c + i];
        }
      }
#pragma unroll
      for (int ox = -2; ox <= 2; ++ox) {
        const bool ok = yok && (unsigned)(w + ox) < (unsigned)W;
        const int pos = (2 - oy) * 5 + (2 - ox);
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          if (ok && pi[ox + 2][i] == pos) acc[i] += g[ox + 2][i];
      }
    }
    vstore<T, VEC>(dx + m * dx_ld + c, acc);
  }
}

// Backward with the same walk (16-bit, 16-byte channel vectors, r03): a thread owns R consecutive rows of one column; the (dy, argmax)
// pairs of the five window rows around the current output row live in a ring of 5 x 5 register slots, so every row of pairs is loaded
// ONCE per thread ((R + 4) x 5 / R pairs per output instead of 25).  Same sum, same order (window rows top to bottom, columns left to
// right) as maxpool5_bwd_kernel: bit-identical.
template <typename T, int R>
__global__ __launch_bounds__(256) void maxpool5_bwd_walk_kernel(int B, int H, int W, int C, const T* __restrict__ dy, int dy_ld,
                                                                const uint8_t* __restrict__ idx, T* __restrict__ dx, int dx_ld, int accumulate,
                                                                int cpv, int bands, long total) {
  constexpr int VEC = 8;
  static_assert(sizeof(T) == 2, "16-bit inputs");
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int cv = (int)(gid % cpv);
  long q = gid / cpv;
  const int w = (int)(q % W); q /= W;
  const int band = (int)(q % bands);
  const int b = (int)(q / bands);
  const int h0 = band * R, c = cv * VEC;
  const long img = (long)b * H * W;
  bool cok[5];
  int ixs[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    cok[k] = (unsigned)(w + k - 2) < (unsigned)W;
    ixs[k] = cok[k] ? w + k - 2 : w;                     // an illegal window column loads the pixel's own column (never selected)
  }
  uint4 g[5][5];                                        // ring slot (window row mod 5) x window column: 8 gradient values
  uint2 pi[5][5];                                       //                                                  8 argmax bytes
  sy11_static_for<0, R + 4>([&](auto s_c) {             // a compile-time step index: the ring slots stay registers
    constexpr int s = decltype(s_c)::value;
    {
      const int qy = h0 - 2 + s, cy = min(max(qy, 0), H - 1);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const long qm = img + (long)cy * W + ixs[k];
        g[s % 5][k] = *(const uint4*)(dy + qm * dy_ld + c);
        pi[s % 5][k] = *(const uint2*)(idx + qm * C + c);
      }
    }
    if (s >= 4) {
      const int o = s - 4, h = h0 + o;
      if (h < H) {
        const long m = img + (long)h * W + w;
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        if (accumulate) vload<T, VEC>(dx + m * dx_ld + c, acc);
        // every window q = p + (oy, ox) that contains p; p sits at window position (2-oy, 2-ox) of q
#pragma unroll
        for (int oy = -2; oy <= 2; ++oy) {
          const bool yok = (unsigned)(h + oy) < (unsigned)H;
          const int slot = (o + oy + 2) % 5;
#pragma unroll
          for (int ox = -2; ox <= 2; ++ox) {
            const bool ok = yok && cok[ox + 2];
            const unsigned pos = (unsigned)((2 - oy) * 5 + (2 - ox));
            typedef T vt __attribute__((ext_vector_type(VEC)));
            const vt gv = __builtin_bit_cast(vt, g[slot][ox + 2]);
            const uint2 pk = pi[slot][ox + 2];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
              const unsigned pb = ((i < 4 ? pk.x : pk.y) >> (8 * (i & 3))) & 255u;
              if (ok && pb == pos) acc[i] += ElemTraits<T>::to_f(gv[i]);
            }
          }
        }
        vstore<T, VEC>(dx + m * dx_ld + c, acc);
      }
    }
  });
}

extern "C" int sy11_maxpool5_fwd(int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, const void* x, int32_t x_ld,
                                 void* y, int32_t y_ld, uint8_t* idx, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0 && x && y && x_ld >= C && y_ld >= C, "maxpool5_fwd: bad argument");
  SY11_REQUIRE((long)B * H * W < (1L << 31), "maxpool5_fwd: too many pixels");
  const int esz = dtype_size(dtype);
  const bool v = vec_ok(esz, C, {x_ld, y_ld}, {x, y});
  const RowGeom g = row_geom(C, v ? 16 / esz : 1);
  dim3 grid(row_grid((long)B * H * W, g.rows_pb), g.cblocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  static int sep_env = -1;
  if (sep_env < 0) { const char* e = getenv("SY11_MAXPOOL_SEP"); sep_env = e ? atoi(e) : 1; }
  if (sep_env && v && esz == 2) {                      // separable row / column maxima, 5 rows of one column per thread
    constexpr int RR = 5;
    const int bands = (H + RR - 1) / RR, cpv8 = C / 8;
    const long total = (long)B * bands * W * cpv8;
    dim3 gs((unsigned)((total + 255) / 256));
    if (dtype == SY11_F16) hipLaunchKernelGGL((maxpool5_fwd_sep_kernel<_Float16, RR>), gs, block, 0, st, B, H, W, C, (const _Float16*)x, x_ld, (_Float16*)y, y_ld, idx, cpv8, bands, total);
    else hipLaunchKernelGGL((maxpool5_fwd_sep_kernel<__bf16, RR>), gs, block, 0, st, B, H, W, C, (const __bf16*)x, x_ld, (__bf16*)y, y_ld, idx, cpv8, bands, total);
    SY11_LAUNCH_CHECK("maxpool5_fwd");
    return SY11_OK;
  }
  SY11_DISPATCH_DTYPE(dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (v) hipLaunchKernelGGL((maxpool5_fwd_kernel<T, VE>), grid, block, 0, st, B, H, W, C, (const T*)x, x_ld, (T*)y, y_ld, idx, g.cpv, g.rows_pb);
    else hipLaunchKernelGGL((maxpool5_fwd_kernel<T, 1>), grid, block, 0, st, B, H, W, C, (const T*)x, x_ld, (T*)y, y_ld, idx, g.cpv, g.rows_pb);
  });
  SY11_LAUNCH_CHECK("maxpool5_fwd");
  return SY11_OK;
}
extern "C" int sy11_maxpool5_bwd(int32_t dtype, int32_t B, int32_t H, int32_t W, int32_t C, const void* dy, int32_t dy_ld,
                                 const uint8_t* idx, void* dx, int32_t dx_ld, int32_t accumulate, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && B > 0 && H > 0 && W > 0 && C > 0 && dy && dx && idx && dy_ld >= C && dx_ld >= C, "maxpool5_bwd: bad argument");
  SY11_REQUIRE((long)B * H * W < (1L << 31), "maxpool5_bwd: too many pixels");
  const int esz = dtype_size(dtype);
  const bool v = vec_ok(esz, C, {dy_ld, dx_ld}, {dy, dx});
  const RowGeom g = row_geom(C, v ? 16 / esz : 1);
  dim3 grid(row_grid((long)B * H * W, g.rows_pb), g.cblocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  static int walk_env = -1;
  if (walk_env < 0) { const char* e = getenv("SY11_MAXPOOL_SEP"); walk_env = e ? atoi(e) : 1; }
  if (walk_env && v && esz == 2) {                     // five rows of one column per thread, each row of (dy, argmax) pairs loaded once
    constexpr int RR = 5;
    const int bands = (H + RR - 1) / RR, cpv8 = C / 8;
    const long total = (long)B * bands * W * cpv8;
    dim3 gs((unsigned)((total + 255) / 256));
    if (dtype == SY11_F16) hipLaunchKernelGGL((maxpool5_bwd_walk_kernel<_Float16, RR>), gs, block, 0, st, B, H, W, C, (const _Float16*)dy, dy_ld, idx, (_Float16*)dx, dx_ld, accumulate, cpv8, bands, total);
    else hipLaunchKernelGGL((maxpool5_bwd_walk_kernel<__bf16, RR>), gs, block, 0, st, B, H, W, C, (const __bf16*)dy, dy_ld, idx, (__bf16*)dx, dx_ld, accumulate, cpv8, bands, total);
    SY11_LAUNCH_CHECK("maxpool5_bwd");
    return SY11_OK;
  }
  SY11_DISPATCH_DTYPE(dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (v) hipLaunchKernelGGL((maxpool5_bwd_kernel<T, VE>), grid, block, 0, st, B, H, W, C, (const T*)dy, dy_ld, idx, (T*)dx, dx_ld, accumulate, g.cpv, g.rows_pb);
    else hipLaunchKernelGGL((maxpool5_bwd_kernel<T, 1>), grid, block, 0, st, B, H, W, C, (const T*)dy, dy_ld, idx, (T*)dx, dx_ld, accumulate, g.cpv, g.rows_pb);
  });
  SY11_LAUNCH_CHECK("maxpool5_bwd");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ dtype cast
template <typename S, typename D>
__global__ void cast_kernel(long n, const S* __restrict__ s, D* __restrict__ d) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    d[i] = ElemTraits<D>::from_f(ElemTraits<S>::to_f(s[i]));
}
extern "C" int sy11_cast(int32_t src_dtype, int32_t dst_dtype, int64_t n, const void* src, void* dst, void* stream) {
  SY11_REQUIRE(dtype_ok(src_dtype) && dtype_ok(dst_dtype) && n > 0 && src && dst, "cast: bad argument");
  long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  hipStream_t st = (hipStream_t)stream;
  SY11_DISPATCH_DTYPE(src_dtype, S, SY11_DISPATCH_DTYPE(dst_dtype, D, hipLaunchKernelGGL((cast_kernel<S, D>), dim3((unsigned)g), dim3(256), 0, st, (long)n, (const S*)src, (D*)dst)));
  SY11_LAUNCH_CHECK("cast");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ bias gradient + cast
// Backward of a bare nn.Conv2d with bias whose output is f32 (Detect's last 1x1 convs, head.py:44-55): the f32 gradient of the
// logits dz (M rows x N channels) becomes (a) the bias gradient  dbias[c] += sum_m dz[m][c]  and (b) the 16-bit operand dy of the
// filter- / input-gradient kernels, zero-padded to `npad` channels (a whole number of 16-byte vectors: nc = 2 -> 8).  One pass over
// dz instead of three fills, a BatchNorm-reduce launch, a zero fill and a strided copy.  A workgroup owns a contiguous run of rows;
// thread t keeps channel t % npad (the thread count is the largest multiple of npad <= 256, so a wave's loads are contiguous); the
// per-thread sums meet in LDS and are folded in row-group order: ONE ordered partial per workgroup and channel, added atomically
// into dbias or — `partials` != NULL (ordered-reduction mode) — stored to partials[workgroup][N] for a fixed-order fold kernel.
template <typename D, int VEC>
__global__ __launch_bounds__(256) void bias_grad_cast_kernel(long M, int N, int npad, const float* __restrict__ dz, int dz_ld, D* __restrict__ dy,
                                                             float* dbias, float* partials, long rows_per_block) {
  // VEC channels per thread (4 when N, npad, dz_ld are multiples of 4: 16-byte loads, 8-byte stores), else 1
  __shared__ float red[256 * VEC];
  const int cpr = npad / VEC;                                // channel vectors per row
  const int groups = blockDim.x / cpr;                       // row groups of this block
  const int cv = threadIdx.x % cpr, rg = threadIdx.x / cpr, c = cv * VEC;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float s[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s[i] = 0.f;
  constexpr int U = 4;                                       // rows in flight per thread
  for (long r = r0 + rg; r < r1; r += (long)groups * U) {
    float v[U][VEC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long rr = r + (long)u * groups;
      const bool live = rr < r1 && c < N;
      if (VEC == 4) {
        const float4 q = live ? *(const float4*)(dz + rr * dz_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        v[u][0] = q.x; v[u][1 % VEC] = q.y; v[u][2 % VEC] = q.z; v[u][3 % VEC] = q.w;
      } else {
        v[u][0] = live ? dz[rr * dz_ld + c] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long rr = r + (long)u * groups;
      if (rr < r1) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s[i] += v[u][i];
        vstore<D, VEC>(dy + rr * npad + c, v[u]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) red[threadIdx.x * VEC + i] = s[i];
  __syncthreads();
  if (rg == 0) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (c + i >= N) continue;
      float t = 0.f;
      for (int g = 0; g < groups; ++g) t += red[(g * cpr + cv) * VEC + i];        // row groups folded in index order
      if (partials) partials[(long)blockIdx.x * N + c + i] = t;
      else if (dbias) atomicAdd(dbias + c + i, t);
    }
  }
}
// out[c] += sum over `rows` partial rows, in a fixed order: one workgroup per channel, thread t takes rows t, t + 256, ..., then a
// binary tree over the 256 threads
__global__ __launch_bounds__(256) void fold_rows_kernel(int rows, int N, const float* __restrict__ partials, float* __restrict__ out) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float t = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) t += partials[(long)r * N + c];
  red[threadIdx.x] = t;
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    if (threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] += red[0];
}

extern "C" int sy11_bias_grad_cast(int32_t dtype, int64_t M, int32_t N, int32_t npad, const float* dz, int32_t dz_ld, void* dy, float* dbias,
                                   float* partials, int32_t partial_rows, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && M > 0 && N > 0 && npad >= N && npad <= 256 && dz && dy && dz_ld >= N, "bias_grad_cast: bad argument (N <= npad <= 256)");
  const bool v4 = N % 4 == 0 && npad % 4 == 0 && dz_ld % 4 == 0 && ((uintptr_t)dz & 15) == 0 && ((uintptr_t)dy & 15) == 0;
  const int cpr = v4 ? npad / 4 : npad;
  const int threads = 256 / cpr * cpr;
  // few, long workgroups: every workgroup ends in N same-line adds (atomic mode) or one partial row (ordered mode)
  long blocks = (M + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);
  if (partials) {
    SY11_REQUIRE(dbias && partial_rows > 0, "bias_grad_cast: ordered mode needs dbias and a partial-row count");
    if (blocks > partial_rows) blocks = partial_rows;
  }
  const long rpb = (M + blocks - 1) / blocks;
  blocks = (M + rpb - 1) / rpb;
  hipStream_t st = (hipStream_t)stream;
  SY11_DISPATCH_DTYPE(dtype, D, {
    if (v4) hipLaunchKernelGGL((bias_grad_cast_kernel<D, 4>), dim3((unsigned)blocks), dim3(threads), 0, st, (long)M, N, npad, dz, dz_ld, (D*)dy, dbias, partials, rpb);
    else hipLaunchKernelGGL((bias_grad_cast_kernel<D, 1>), dim3((unsigned)blocks), dim3(threads), 0, st, (long)M, N, npad, dz, dz_ld, (D*)dy, dbias, partials, rpb);
  });
  if (partials) hipLaunchKernelGGL(fold_rows_kernel, dim3(N), dim3(256), 0, st, (int)blocks, N, (const float*)partials, dbias);
  SY11_LAUNCH_CHECK("bias_grad_cast");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ ordered row fold (det.h)
// One stage: rows [64 * y, 64 * y + 64) of `src` -> row y of `dst` (ACCUM: added onto dst row 0 instead, used by the last stage
// with gridDim.y == 1).  Block = 64 columns x 4 row lanes; lane q adds rows q, q + 4, ... of its 64-row group in order, the four
// lanes are folded in index order: a fixed tree, whatever the launch timing.
// t += src[r * stride + c] for r = r_begin, r_begin + step, ... < r_end, IN THAT ORDER — eight loads in flight per lane (the plain loop waits
// for each load before the next: ~9 us per fold launch, r04 trace), the additions in the same order as the plain loop: same sums, bit for bit
__device__ __forceinline__ float fold_rows_in_order(float* __restrict__ src, long stride, int c, long r_begin, long r_end, long step, int clean) {
  float t = 0.f;
  long r = r_begin;
  for (; r + 7 * step < r_end; r += 8 * step) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[(r + i * step) * stride + c];
#pragma unroll
    for (int i = 0; i < 8; ++i) t += v[i];
    if (clean) {
#pragma unroll
      for (int i = 0; i < 8; ++i) src[(r + i * step) * stride + c] = 0.f;          // a block from the clean head of the workspace: leave zeros behind (det.h)
    }
  }
  for (; r < r_end; r += step) {
    t += src[r * stride + c];
    if (clean) src[r * stride + c] = 0.f;
  }
  return t;
}

template <bool ACCUM>
__global__ __launch_bounds__(256) void fold_stage_kernel(long rows, int N, float* __restrict__ src, long stride, float* __restrict__ dst, long group,
                                                         long src_bstride, long dst_bstride, float* __restrict__ dst1, int clean) {
  // blockIdx.z = buffer (two statistic buffers of one launch fold side by side: the same tree per buffer as two separate folds)
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long r0 = (long)blockIdx.y * group, r1 = r0 + group < rows ? r0 + group : rows;
  src += (long)blockIdx.z * src_bstride;
  float* out = ACCUM ? (blockIdx.z ? dst1 : dst) : dst + (long)blockIdx.z * dst_bstride;
  const float t = c < N ? fold_rows_in_order(src, stride, c, r0 + q, r1, 4, clean) : 0.f;
  red[q][cl] = t;
  __syncthreads();
  if (q == 0 && c < N) {
    const float tot = ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
    if (ACCUM) out[c] += tot; else out[(long)blockIdx.y * N + c] = tot;
  }
}
// Partial rows into the caller's SLOT rows (the consumer folds those): workgroup (column block, slot s, buffer) adds rows s, s + S, ... —
// lane q takes every fourth of them in ascending order, the four lanes are folded in index order: a fixed tree, one plain launch.
__global__ __launch_bounds__(256) void fold_slots_kernel(long rows, int N, float* __restrict__ src, long stride, long src_bstride, int S, long slot_stride,
                                                         float* __restrict__ out0, float* __restrict__ out1, int clean) {
  __shared__ float red[4][64];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const long s0 = blockIdx.y;
  src += (long)blockIdx.z * src_bstride;
  const float t = c < N ? fold_rows_in_order(src, stride, c, s0 + (long)S * q, rows, 4L * S, clean) : 0.f;
  red[q][cl] = t;
  __syncthreads();
  if (q == 0 && c < N) {
    float* out = (blockIdx.z ? out1 : out0) + s0 * slot_stride;
    out[c] += ((red[0][cl] + red[1][cl]) + red[2][cl]) + red[3][cl];
  }
}
int sy11_fold_rows_to_slots2(long rows, int N, float* partials, long stride, long buf_stride, int slots, long slot_stride, float* out0, float* out1,
                             hipStream_t st, bool clean) {
  if (rows <= 0 || N <= 0 || slots <= 0) return SY11_OK;
  hipLaunchKernelGGL(fold_slots_kernel, dim3(cdiv(N, 64), (unsigned)slots, 2), dim3(256), 0, st, rows, N, partials, stride, buf_stride, slots, slot_stride, out0, out1, clean ? 1 : 0);
  SY11_LAUNCH_CHECK("fold_rows_to_slots");
  return SY11_OK;
}

// the two buffers of a DetPartials block ([2][rows][N], `scratch` twice sy11_fold_scratch_floats) in one launch per stage
int sy11_fold_rows_ordered2(long rows, int N, float* partials, long stride, long buf_stride, float* out0, float* out1, float* scratch,
                            hipStream_t st, bool clean) {
  if (rows <= 0 || N <= 0) return SY11_OK;
  float* src = partials;
  int cl = clean ? 1 : 0;                                 // only the first stage reads the partial block itself
  long sstride = stride, sb = buf_stride;
  float* ping = scratch;
  while (rows > 256) {
    const long nr = (rows + 63) / 64;
    float* dst = ping;
    hipLaunchKernelGGL((fold_stage_kernel<false>), dim3(cdiv(N, 64), (unsigned)nr, 2), dim3(256), 0, st, rows, N, src, sstride, dst, 64L, sb, nr * N, (float*)nullptr, cl);
    src = dst; sstride = N; sb = nr * N; rows = nr; cl = 0;
    ping = dst + 2 * nr * N;
  }
  hipLaunchKernelGGL((fold_stage_kernel<true>), dim3(cdiv(N, 64), 1, 2), dim3(256), 0, st, rows, N, src, sstride, out0, rows, sb, 0L, out1, cl);
  SY11_LAUNCH_CHECK("fold_rows2");
  return SY11_OK;
}
int sy11_fold_rows_ordered(long rows, int N, float* partials, long stride, float* out, float* scratch, hipStream_t st, bool clean) {
  if (rows <= 0 || N <= 0) return SY11_OK;
  float* src = partials;
  int cl = clean ? 1 : 0;
  long sstride = stride;
  float* ping = scratch;
  while (rows > 256) {                                   // 64 rows -> 1 per stage until one workgroup column can finish
    const long nr = (rows + 63) / 64;
    float* dst = ping;
    hipLaunchKernelGGL((fold_stage_kernel<false>), dim3(cdiv(N, 64), (unsigned)nr), dim3(256), 0, st, rows, N, src, sstride, dst, 64L, 0L, 0L, (float*)nullptr, cl);
    src = dst; sstride = N; rows = nr; cl = 0;
    ping = dst + nr * N;                                 // next stage writes behind this one (scratch holds rows/64 + 64 rows)
  }
  hipLaunchKernelGGL((fold_stage_kernel<true>), dim3(cdiv(N, 64), 1), dim3(256), 0, st, rows, N, src, sstride, out, rows, 0L, 0L, (float*)nullptr, cl);
  SY11_LAUNCH_CHECK("fold_rows");
  return SY11_OK;
}
