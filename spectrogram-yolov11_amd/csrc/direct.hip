// direct.hip — direct (non-GEMM) convolutions for gfx950: depthwise KxK (Detect cls branch, attention `pe`) and the
// 3-channel stem.  Both are HBM/L2-bound (depthwise: 4.5 FLOP/B; stem: K = 27), so no MFMA: a thread owns a fixed
// 16-byte channel vector, keeps its filter taps in registers and walks pixels; neighbouring taps are L1/L2 hits.
#include "common.h"
#include "det.h"
#include <type_traits>
#include "bn_tail.h"

template <typename T, int VEC>
__device__ __forceinline__ void dvload(const T* p, float* f) {
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
__device__ __forceinline__ void dvstore(T* p, const float* f) {
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

// the VEC channels x 9 taps a thread owns are VEC*9 CONTIGUOUS filter elements ([C][3][3] layout): 9 x 16-byte loads
template <typename T, int VEC, int MAXT>
__device__ __forceinline__ void load_taps(const T* __restrict__ w, int c, int Tn, float (&wr)[MAXT][VEC]) {
  if constexpr (VEC > 1 && MAXT == 9) {
    if (Tn == 9) {
      T buf[VEC * 9];
      typedef T vt __attribute__((ext_vector_type(VEC)));
#pragma unroll
      for (int j = 0; j < 9; ++j) *(vt*)&buf[j * VEC] = *(const vt*)(w + (long)c * 9 + j * VEC);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < VEC; ++i) wr[t][i] = ElemTraits<T>::to_f(buf[i * 9 + t]);
      return;
    }
  }
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) wr[t][i] = t < Tn ? ElemTraits<T>::to_f(w[(long)(c + i) * Tn + t]) : 0.f;
}

struct DwArgs {
  const void* x; const void* w; void* y; const float* bias; float* stat_sum; float* stat_sq;
  int B, IH, IW, OH, OW, C, x_ld, y_ld;
  int KH, KW, SH, SW, PH, PW, DH, DW;
  unsigned flags;
  int cpv, rows_pb;
  long rows_per_block;
  int stat_slots;
  long part_stride;      // filter gradient, ordered mode (det.h): workgroup x adds into dw + x * part_stride instead of the shared dw
};

// MAXT = taps kept in registers (9 for 3x3)
template <typename T, int VEC, int MAXT>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const DwArgs a) {
  __shared__ float red[2][256][VEC];
  const int cw = a.cpv < 256 ? a.cpv : 256;
  const int cl = threadIdx.x % cw, cv = blockIdx.y * 256 + cl, rsub = threadIdx.x / cw;
  const bool active = cv < a.cpv && rsub < a.rows_pb;
  const int c = cv * VEC;
  const int Tn = a.KH * a.KW;
  float s1[VEC], s2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
  if (active) {
    float wr[MAXT][VEC], bv[VEC];
    load_taps<T, VEC, MAXT>((const T*)a.w, c, Tn, wr);
#pragma unroll
    for (int i = 0; i < VEC; ++i) bv[i] = a.bias ? a.bias[c + i] : 0.f;
    const int M = a.B * a.OH * a.OW;
    const int m0 = blockIdx.x * (int)a.rows_per_block;
    const int m1 = m0 + (int)a.rows_per_block < M ? m0 + (int)a.rows_per_block : M;
    const bool silu = a.flags & SY11_EPI_SILU;
    for (int m = m0 + rsub; m < m1; m += a.rows_pb) {
      const int q = m / a.OW, ox = m - q * a.OW;
      const int b = q / a.OH, oy = q - b * a.OH;
      float acc[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
      float tv[MAXT][VEC];
      float tm[MAXT];
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int r = t / a.KW, s = t - r * a.KW;
        const int iy = oy * a.SH - a.PH + r * a.DH, ix = ox * a.SW - a.PW + s * a.DW;
        const bool in = t < Tn && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
        const int cy = min(max(iy, 0), a.IH - 1), cx = min(max(ix, 0), a.IW - 1);     // clamped: always a legal address
        tm[t] = in ? 1.f : 0.f;
        dvload<T, VEC>((const T*)a.x + ((long)(b * a.IH + cy) * a.IW + cx) * a.x_ld + c, tv[t]);
      }
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] += tv[t][i] * (wr[t][i] * tm[t]);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        s1[i] += acc[i];
        s2[i] += acc[i] * acc[i];
        float v = acc[i] + bv[i];
        acc[i] = silu ? silu_f(v) : v;
      }
      dvstore<T, VEC>((T*)a.y + (long)m * a.y_ld + c, acc);
    }
  }
  if (a.stat_sum) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) { red[0][threadIdx.x][i] = s1[i]; red[1][threadIdx.x][i] = s2[i]; }
    __syncthreads();
    int half = 1;
    while (half < a.rows_pb) half <<= 1;
    for (half >>= 1; half >= 1; half >>= 1) {
      if (rsub < half && rsub + half < a.rows_pb) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          red[0][threadIdx.x][i] += red[0][threadIdx.x + half * cw][i];
          red[1][threadIdx.x][i] += red[1][threadIdx.x + half * cw][i];
        }
      }
      __syncthreads();
    }
    const int nch = min(cw, a.cpv - (int)blockIdx.y * 256) * VEC;       // dense, consecutive-lane atomics
    for (int e = threadIdx.x; e < nch; e += 256) {
      const int cl2 = e / VEC, i2 = e - cl2 * VEC;
      const long so = (long)(blockIdx.x % a.stat_slots) * a.C + blockIdx.y * 256 * VEC + e;
      atomicAdd(a.stat_sum + so, red[0][cl2][i2]);
      atomicAdd(a.stat_sq + so, red[1][cl2][i2]);
    }
  }
}

// dx[p][c] (+)= sum_t dy[(p + P - t*D)/S][c] * w[c][t]
template <typename T, int VEC, int MAXT>
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const DwArgs a, int dy_ld) {
  const int cw = a.cpv < 256 ? a.cpv : 256;
  const int cv = blockIdx.y * 256 + (threadIdx.x % cw), rsub = threadIdx.x / cw;
  if (cv >= a.cpv || rsub >= a.rows_pb) return;
  const int c = cv * VEC;
  const int Tn = a.KH * a.KW;
  float wr[MAXT][VEC];
  load_taps<T, VEC, MAXT>((const T*)a.w, c, Tn, wr);
  const int M = a.B * a.IH * a.IW;
  const bool accum = a.flags & SY11_EPI_ACCUM;
  for (int m = blockIdx.x * a.rows_pb + rsub; m < M; m += gridDim.x * a.rows_pb) {
    const int q = m / a.IW, ix = m - q * a.IW;
    const int b = q / a.IH, iy = q - b * a.IH;
    float acc[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
    float tv[MAXT][VEC];
    float tm[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int r = t / a.KW, s = t - r * a.KW;
      const int ny = iy + a.PH - r * a.DH, nx = ix + a.PW - s * a.DW;
      const int oy = ny / a.SH, ox = nx / a.SW;
      const bool ok = t < Tn && ny >= 0 && nx >= 0 && (ny % a.SH) == 0 && (nx % a.SW) == 0 && oy < a.OH && ox < a.OW;
      const int cy = min(max(oy, 0), a.OH - 1), cx = min(max(ox, 0), a.OW - 1);
      tm[t] = ok ? 1.f : 0.f;
      dvload<T, VEC>((const T*)a.x + ((long)(b * a.OH + cy) * a.OW + cx) * dy_ld + c, tv[t]);   // a.x = dy here
    }
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += tv[t][i] * (wr[t][i] * tm[t]);
    T* dp = (T*)a.y + (long)m * a.x_ld + c;                                                        // a.y = dx, stride x_ld
    if (accum) {
      float o[VEC];
      dvload<T, VEC>(dp, o);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[i] += o[i];
    }
    dvstore<T, VEC>(dp, acc);
  }
}

// dw[c][t] += sum_m dy[m][c] * x[pix(m,t)][c]
template <typename T, int VEC, int MAXT>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const DwArgs a, int dy_ld, float* dw) {
  __shared__ float red[256][MAXT * VEC + 1];
  const int cw = a.cpv < 256 ? a.cpv : 256;
  const int cl = threadIdx.x % cw, cv = blockIdx.y * 256 + cl, rsub = threadIdx.x / cw;
  const bool active = cv < a.cpv && rsub < a.rows_pb;
  const int c = cv * VEC;
  const int Tn = a.KH * a.KW;
  float acc[MAXT][VEC];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[t][i] = 0.f;
  if (active) {
    const int M = a.B * a.OH * a.OW;
    const int m0 = blockIdx.x * (int)a.rows_per_block;
    const int m1 = m0 + (int)a.rows_per_block < M ? m0 + (int)a.rows_per_block : M;
    for (int m = m0 + rsub; m < m1; m += a.rows_pb) {
      const int q = m / a.OW, ox = m - q * a.OW;
      const int b = q / a.OH, oy = q - b * a.OH;
      float g[VEC];
      dvload<T, VEC>((const T*)a.y + (long)m * dy_ld + c, g);                                      // a.y = dy here
      float tv[MAXT][VEC];
      float tm[MAXT];
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int r = t / a.KW, s = t - r * a.KW;
        const int iy = oy * a.SH - a.PH + r * a.DH, ix = ox * a.SW - a.PW + s * a.DW;
        const bool in = t < Tn && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
        const int cy = min(max(iy, 0), a.IH - 1), cx = min(max(ix, 0), a.IW - 1);
        tm[t] = in ? 1.f : 0.f;
        dvload<T, VEC>((const T*)a.x + ((long)(b * a.IH + cy) * a.IW + cx) * a.x_ld + c, tv[t]);
      }
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[t][i] += (g[i] * tm[t]) * tv[t][i];
    }
  }
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[threadIdx.x][t * VEC + i] = acc[t][i];
  __syncthreads();
  int half = 1;
  while (half < a.rows_pb) half <<= 1;
  for (half >>= 1; half >= 1; half >>= 1) {
    if (rsub < half && rsub + half < a.rows_pb) {
#pragma unroll
      for (int e = 0; e < MAXT * VEC; ++e) red[threadIdx.x][e] += red[threadIdx.x + half * cw][e];
    }
    __syncthreads();
  }
  {
    const int nch = min(cw, a.cpv - (int)blockIdx.y * 256) * VEC;        // channels owned by this workgroup
    float* base = dw + (long)blockIdx.x * a.part_stride + (long)blockIdx.y * 256 * VEC * Tn;
    for (int e = threadIdx.x; e < nch * Tn; e += 256) {                  // consecutive lanes -> consecutive addresses
      const int ch = e / Tn, t = e - ch * Tn;
      atomicAdd(base + e, red[ch / VEC][t * VEC + (ch % VEC)]);
    }
  }
}

// f16, 3x3: the same sum with the per-pixel work cut to what it needs.  The generic kernel above spends ~2200 issue cycles
// per pixel and wave on 43 integer multiplies (nine independent tap addresses), 80 f16->f32 conversions and 36 mask
// multiplies; here the three row bases and three column offsets are formed once (6 multiplies), the pixel coordinates
// advance without divisions, out-of-image taps become zero VECTORS (16-byte selects on the loaded halves) and the
// products are written as (float)h * (float)h + f32 so the compiler emits v_fma_mix_f32 — no conversions at all.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void dwconv_wgrad3x3_f16_kernel(const DwArgs a, int dy_ld, float* dw) {
  constexpr int VEC = 8, MAXT = 9;
  __shared__ float red[256][MAXT * VEC + 1];
  const int cw = a.cpv < 256 ? a.cpv : 256;
  const int cl = threadIdx.x % cw, cv = blockIdx.y * 256 + cl, rsub = threadIdx.x / cw;
  const bool active = cv < a.cpv && rsub < a.rows_pb;
  const int c = cv * VEC;
  float acc[MAXT][VEC];
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[t][i] = 0.f;
  if (active) {
    const int M = a.B * a.OH * a.OW;
    const int m0 = blockIdx.x * (int)a.rows_per_block;
    const int m1 = m0 + (int)a.rows_per_block < M ? m0 + (int)a.rows_per_block : M;
    int m = m0 + rsub;
    int q = m / a.OW, ox = m - q * a.OW;
    int b = q / a.OH, oy = q - b * a.OH;
    const _Float16* xb = (const _Float16*)a.x + c;
    const _Float16* dyb = (const _Float16*)a.y + c;                   // a.y = dy here
    const long row_elems = (long)a.IW * a.x_ld;
    const h16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (; m < m1; m += a.rows_pb) {
      const h16x8 g = *(const h16x8*)(dyb + (long)m * dy_ld);
      const _Float16* rowp[3];
      bool rok[3], cok[3];
      int coff[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int iy = oy * a.SH - a.PH + r * a.DH;
        rok[r] = (unsigned)iy < (unsigned)a.IH;
        const int cy = min(max(iy, 0), a.IH - 1);
        rowp[r] = xb + (long)(b * a.IH + cy) * row_elems;
      }
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        const int ix = ox * a.SW - a.PW + s2 * a.DW;
        cok[s2] = (unsigned)ix < (unsigned)a.IW;
        coff[s2] = min(max(ix, 0), a.IW - 1) * a.x_ld;
      }
      h16x8 tv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) tv[t] = *(const h16x8*)(rowp[t / 3] + coff[t % 3]);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const h16x8 v = (rok[t / 3] && cok[t % 3]) ? tv[t] : zero;
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[t][i] += (float)g[i] * (float)v[i];
      }
      // advance rows_pb pixels without dividing (rows_pb <= 256 may span several short rows)
      ox += a.rows_pb;
      while (ox >= a.OW) { ox -= a.OW; ++oy; }
      while (oy >= a.OH) { oy -= a.OH; ++b; }
    }
  }
#pragma unroll
  for (int t = 0; t < MAXT; ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) red[threadIdx.x][t * VEC + i] = acc[t][i];
  __syncthreads();
  int half = 1;
  while (half < a.rows_pb) half <<= 1;
  for (half >>= 1; half >= 1; half >>= 1) {
    if (rsub < half && rsub + half < a.rows_pb) {
#pragma unroll
      for (int e = 0; e < MAXT * VEC; ++e) red[threadIdx.x][e] += red[threadIdx.x + half * cw][e];
    }
    __syncthreads();
  }
  const int nch = min(cw, a.cpv - (int)blockIdx.y * 256) * VEC;
  float* base = dw + (long)blockIdx.x * a.part_stride + (long)blockIdx.y * 256 * VEC * 9;
  for (int e = threadIdx.x; e < nch * 9; e += 256) {
    const int ch = e / 9, t = e - ch * 9;
    atomicAdd(base + e, red[ch / VEC][t * VEC + (ch % VEC)]);
  }
}

// ---- 3x3 / stride 1 / pad 1 / dilation 1 depthwise (every depthwise layer of the YOLO11 graphs): a thread owns a channel
// vector and walks RUN consecutive pixels of one image row with a sliding 3x3 window held in registers (16-bit packed):
// 3 new 16-byte loads per output pixel instead of 9, no integer division in the pixel loop.  One routine serves
//   MODE 0 forward  (y = sum_t x[p + t - 1] w[t]; BN statistics),
//   MODE 1 dgrad    (dx = sum_t dy[p + 1 - t] w[t] = the same walk with the taps flipped; optional accumulate),
//   MODE 2 wgrad    (dw[t] += dy[p] * x[p + t - 1]).
template <typename T, int VEC> struct PackedVec { typedef T type __attribute__((ext_vector_type(VEC))); };

template <typename T, int VEC, int MODE>
__global__ __launch_bounds__(256) void dw3x3_kernel(const DwArgs a, int aux_ld, float* dw, int run, int runs_per_row, const BnTailDev tail) {
  typedef typename PackedVec<T, VEC>::type pv;
  extern __shared__ float red[];                              // MODE 0: [2][C] stats; MODE 2: [9][C] filter gradient
  const int C = a.C, cpv = a.cpv;
  const int H = a.IH, W = a.IW;                               // stride 1, pad 1: input and output have the same size
  const int nred = MODE == 0 ? 2 * C : (MODE == 2 ? 9 * C : 0);
  for (int i = threadIdx.x; i < nred; i += 256) red[i] = 0.f;
  if (nred) __syncthreads();
  const long total = (long)a.B * H * runs_per_row * cpv;
  float wr[9][VEC];
  float s1[VEC], s2[VEC], gacc[MODE == 2 ? 9 : 1][VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) s1[i] = s2[i] = 0.f;
#pragma unroll
  for (int t = 0; t < (MODE == 2 ? 9 : 1); ++t)
#pragma unroll
    for (int i = 0; i < VEC; ++i) gacc[t][i] = 0.f;
  int cur_c = -1;
  const T* src = (const T*)a.x;                               // walked tensor: x (fwd / wgrad) or dy (dgrad)
  const int src_ld = MODE == 1 ? aux_ld : a.x_ld;
  for (long gid = (long)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (long)gridDim.x * 256) {
    const int cv = (int)(gid % cpv);
    const long rid = gid / cpv;
    const int rr = (int)(rid % runs_per_row);
    const long rowid = rid / runs_per_row;
    const int oy = (int)(rowid % H), b = (int)(rowid / H);
    const int c = cv * VEC, x0 = rr * run, x1 = min(W, x0 + run);
    if (MODE != 2 && c != cur_c) {                            // the channel vector is fixed when gridDim.x*256 % cpv == 0
      float tmp[9][VEC];
      load_taps<T, VEC, 9>((const T*)a.w, c, 9, tmp);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < VEC; ++i) wr[t][i] = MODE == 1 ? tmp[8 - t][i] : tmp[t][i];
      cur_c = c;
    }
    const T* rowp[3];
    bool rok[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int iy = oy + r - 1;
      rok[r] = (unsigned)iy < (unsigned)H;
      rowp[r] = src + ((long)(b * H + (rok[r] ? iy : oy)) * W) * src_ld + c;
    }
    pv win[3][3];                                             // [row][column]: columns x-1, x, x+1
    const pv zero = (pv)(T)0;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      win[r][0] = zero;
      // loads are unconditional from a clamped (always legal) address and masked afterwards: a predicated load compiles to a branch with
      // a full wait at its join, i.e. three serialised memory latencies per pixel step (r02, tools/dw_micro.py: 84.7 -> 79 us forward,
      // 59 -> 55 us input gradient on 80x80x128)
      const pv l1 = *(const pv*)(rowp[r] + (long)max(x0 - 1, 0) * src_ld);
      const pv l2 = *(const pv*)(rowp[r] + (long)x0 * src_ld);
      win[r][1] = (rok[r] && x0 - 1 >= 0) ? l1 : zero;
      win[r][2] = rok[r] ? l2 : zero;
    }
    for (int x = x0; x < x1; ++x) {
      pv nxt[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) nxt[r] = *(const pv*)(rowp[r] + (long)min(x + 1, W - 1) * src_ld);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        win[r][0] = win[r][1];
        win[r][1] = win[r][2];
        win[r][2] = (rok[r] && x + 1 < W) ? nxt[r] : zero;
      }
      const long m = (long)(b * H + oy) * W + x;
      if (MODE == 2) {
        float g[VEC];
        dvload<T, VEC>((const T*)a.y + m * aux_ld + c, g);   // a.y = dy
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
            for (int i = 0; i < VEC; ++i) gacc[r * 3 + s_][i] += g[i] * ElemTraits<T>::to_f(win[r][s_][i]);
      } else {
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] += ElemTraits<T>::to_f(win[r][s_][i]) * wr[r * 3 + s_][i];
        T* op = (T*)a.y + m * (MODE == 1 ? a.x_ld : a.y_ld) + c;
        if (MODE == 0) {
          const bool silu = a.flags & SY11_EPI_SILU;
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            s1[i] += acc[i];
            s2[i] += acc[i] * acc[i];
            const float v = acc[i] + (a.bias ? a.bias[c + i] : 0.f);
            acc[i] = silu ? silu_f(v) : v;
          }
        } else if (a.flags & SY11_EPI_ACCUM) {
          float o[VEC];
          dvload<T, VEC>(op, o);
#pragma unroll
          for (int i = 0; i < VEC; ++i) acc[i] += o[i];
        }
        dvstore<T, VEC>(op, acc);
      }
    }
  }
  // the thread's channel vector never changes (gridDim.x * 256 is a multiple of cpv): partial sums stay in registers for
  // the whole walk and are folded once — LDS atomics per run were 3x the cost of the walk itself
  if (MODE == 0 && a.stat_sum) {
    // every thread parks its 2 x VEC partial sums in LDS, then one thread per (sum, channel) adds the 256 / cpv rows that share the
    // channel: LDS atomics serialise the 16 threads of a channel (r02: 79 -> 64 us on 80x80x128, 33 -> 28 on 20x20x512)
    float* park = red + 2 * C;                                  // [256][2 * VEC]
#pragma unroll
    for (int i = 0; i < VEC; ++i) { park[threadIdx.x * 2 * VEC + i] = s1[i]; park[threadIdx.x * 2 * VEC + VEC + i] = s2[i]; }
    __syncthreads();
    const int share = 256 / cpv;                                // threads per channel vector (256 % cpv == 0)
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
      const int which = e / C, ch = e - which * C, cvv = ch / VEC, i = ch - cvv * VEC;
      float tsum = 0.f;                                         // thread t holds channel vector t % cpv (256 % cpv == 0)
      for (int k = 0; k < share; ++k) tsum += park[(cvv + k * cpv) * 2 * VEC + which * VEC + i];
      red[e] = tsum;
    }
  }
  if (MODE == 2) {
    // three taps at a time through a [256][3 * VEC] parking area (see MODE 0): red[t][c] = sum over the threads of channel c
    float* park = red + 9 * C;
    const int share = 256 / cpv;
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
      if (t3) __syncthreads();
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int i = 0; i < VEC; ++i) park[threadIdx.x * 3 * VEC + tt * VEC + i] = gacc[t3 * 3 + tt][i];
      __syncthreads();
      for (int e = threadIdx.x; e < 3 * C; e += 256) {
        const int tt = e / C, ch = e - tt * C, cvv = ch / VEC, i = ch - cvv * VEC;
        float tsum = 0.f;
        for (int k = 0; k < share; ++k) tsum += park[(cvv + k * cpv) * 3 * VEC + tt * VEC + i];
        red[(t3 * 3 + tt) * C + ch] = tsum;
      }
    }
  }
  if (MODE == 0 && a.stat_sum) {
    __syncthreads();
    const long so = (long)(blockIdx.x % a.stat_slots) * C;
    for (int i = threadIdx.x; i < C; i += 256) {
      atomicAdd(a.stat_sum + so + i, red[i]);
      atomicAdd(a.stat_sq + so + i, red[C + i]);
    }
    if (tail.ticket) bn_tail_run(tail, a.stat_sum, a.stat_sq, a.stat_slots, C, gridDim.x);
  }
  if (MODE == 2) {
    __syncthreads();
    for (int e = threadIdx.x; e < 9 * C; e += 256) {          // dw layout [c][t]: consecutive lanes -> consecutive addresses
      const int ch = e / 9, t = e - ch * 9;
      atomicAdd(dw + (long)blockIdx.x * a.part_stride + e, red[t * C + ch]);
    }
  }
}

// geometry of the sliding-window kernels: run length such that a row splits evenly-ish, grid such that every thread keeps one
// channel vector (gridDim.x * 256 a multiple of cpv) and walks >= 2 runs
static bool dw3x3_ok(const sy11_conv_desc* d, bool vec) {
  return vec && d->KH == 3 && d->KW == 3 && d->SH == 1 && d->SW == 1 && d->PH == 1 && d->PW == 1 && d->DH == 1 && d->DW == 1 &&
         d->IH == d->OH && d->IW == d->OW && d->C <= 1536 && 256 % (d->C / (16 / dtype_size(d->dtype))) == 0;
}
static void dw3x3_geom(const sy11_conv_desc* d, int cpv, int* run, int* rpr, unsigned* grid, long max_grid = 4096) {
  const int W = d->IW;
  int r = W <= 10 ? W : (W % 10 == 0 ? 10 : (W % 8 == 0 ? 8 : 10));
  *run = r;
  *rpr = cdiv(W, r);
  const long total = (long)d->B * d->IH * (*rpr) * cpv;
  long g = (total + 511) / 512;                               // ~2 runs per thread
  if (g > max_grid) g = max_grid;
  if (g < 1) g = 1;
  *grid = (unsigned)g;
}

static int dw_setup(const sy11_conv_desc* d, DwArgs& a, bool& vec, const void* p0, int ld0, const void* p1, int ld1, long M,
                    int min_rows_per_thread, dim3& grid) {
  SY11_REQUIRE(d->KH * d->KW <= 9, "depthwise: only up to 9 taps (3x3) are supported");
  const int esz = dtype_size(d->dtype), ve = 16 / esz;
  vec = (d->C % ve == 0) && (ld0 % ve == 0) && (ld1 % ve == 0) && !(((uintptr_t)p0 | (uintptr_t)p1) & 15);
  const int v = vec ? ve : 1;
  a.B = d->B; a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW; a.C = d->C; a.x_ld = d->x_ld; a.y_ld = d->y_ld;
  a.KH = d->KH; a.KW = d->KW; a.SH = d->SH; a.SW = d->SW; a.PH = d->PH; a.PW = d->PW; a.DH = d->DH; a.DW = d->DW;
  a.flags = d->flags;
  a.stat_slots = d->stat_slots > 1 ? d->stat_slots : 1;
  a.cpv = d->C / v;
  const int cw = a.cpv < 256 ? a.cpv : 256;
  a.rows_pb = 256 / cw;
  long nblk = (M + (long)a.rows_pb * min_rows_per_thread - 1) / ((long)a.rows_pb * min_rows_per_thread);
  if (nblk > 512) nblk = 512;       // bounds the same-address stat / dw atomics per channel
  if (nblk < 1) nblk = 1;
  a.rows_per_block = ((M + nblk - 1) / nblk + a.rows_pb - 1) / a.rows_pb * a.rows_pb;
  nblk = (M + a.rows_per_block - 1) / a.rows_per_block;
  grid = dim3((unsigned)nblk, cdiv(a.cpv, 256));
  return SY11_OK;
}

static int dwconv_fwd_tail(const sy11_conv_desc* d, const void* x, const void* w, const float* bias, void* y, float* stat_sum, float* stat_sq,
                           hipStream_t st, const BnTailDev& tail, bool* tail_done);

int sy11_dwconv_fwd_impl(const sy11_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                         float* stat_sum, float* stat_sq, hipStream_t st) {
  bool done;
  return dwconv_fwd_tail(d, x, w, bias, y, stat_sum, stat_sq, st, BnTailDev{}, &done);
}

// depthwise Conv.forward in train mode: statistics finalised in the kernel tail where the kernel supports it, else by the
// stand-alone finalize launch right behind it (same stream)
int sy11_dwconv_fwd_bn_impl(const sy11_conv_desc* d, const void* x, const void* w, void* y, float* stat_sum, float* stat_sq,
                            const sy11_bn_tail* bn, hipStream_t st) {
  bool done = false;
  int rc = dwconv_fwd_tail(d, x, w, nullptr, y, stat_sum, stat_sq, st, bn_tail_dev(bn, d->N, 0, 0), &done);
  if (rc || done) return rc;
  return sy11_bn_finalize(d->N, d->stat_slots > 1 ? d->stat_slots : 1, bn->count, stat_sum, stat_sq, bn->gamma, bn->beta, bn->eps, bn->momentum,
                          bn->running_mean, bn->running_var, bn->mean, bn->rstd, bn->scale, bn->shift, (void*)st);
}

static int dwconv_fwd_tail(const sy11_conv_desc* d, const void* x, const void* w, const float* bias, void* y, float* stat_sum, float* stat_sq,
                           hipStream_t st, const BnTailDev& tail, bool* tail_done) {
  *tail_done = false;
  SY11_REQUIRE(x && w && y, "dwconv_fwd: null pointer");
  SY11_REQUIRE(!(d->flags & (SY11_EPI_ACCUM | SY11_EPI_OUT_F32)), "dwconv_fwd: unsupported epilogue flag");
  DwArgs a{};
  bool vec;
  dim3 grid;
  int rc = dw_setup(d, a, vec, x, d->x_ld, y, d->y_ld, (long)d->B * d->OH * d->OW, 16, grid);
  if (rc) return rc;
  a.x = x; a.w = w; a.y = y; a.bias = bias; a.stat_sum = stat_sum; a.stat_sq = stat_sq;
  DetPartials dp;                                             // ordered mode (det.h): one partial statistics row per workgroup
  const bool det = stat_sum && sy11_det(2) && !tail.ticket;
  auto det_begin = [&](long rows) -> bool {
    if (!dp.acquire(st, 2, rows, d->C)) return false;
    a.stat_sum = dp.buf(0); a.stat_sq = dp.buf(1); a.stat_slots = (int)rows;
    return true;
  };
  auto det_end = [&]() -> int { return dp.fold01(stat_sum, stat_sq, d->stat_slots > 1 ? d->stat_slots : 1, d->C); };
  if (dw3x3_ok(d, vec)) {
    int run, rpr; unsigned g;
    dw3x3_geom(d, a.cpv, &run, &rpr, &g, 1024);               // every workgroup ends with 2*C statistic atomics
    if (det && !det_begin(g)) SY11_FAIL(SY11_ELAUNCH, "dwconv_fwd: ordered-reduction workspace unavailable");
    SY11_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw3x3_kernel<T, 16 / (int)sizeof(T), 0>), dim3(g), dim3(256), (2 * d->C + 256 * 2 * (16 / (int)sizeof(T))) * sizeof(float), st, a, 0,
                                                         (float*)nullptr, run, rpr, tail));
    SY11_LAUNCH_CHECK("dwconv_fwd");
    *tail_done = tail.ticket != nullptr && stat_sum != nullptr;
    return det ? det_end() : SY11_OK;
  }
  if (det && !det_begin(grid.x)) SY11_FAIL(SY11_ELAUNCH, "dwconv_fwd: ordered-reduction workspace unavailable");
  SY11_DISPATCH_DTYPE(d->dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (vec) hipLaunchKernelGGL((dwconv_fwd_kernel<T, VE, 9>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((dwconv_fwd_kernel<T, 1, 9>), grid, dim3(256), 0, st, a);
  });
  SY11_LAUNCH_CHECK("dwconv_fwd");
  return det ? det_end() : SY11_OK;
}

int sy11_dwconv_dgrad_impl(const sy11_conv_desc* d, const void* dy, int dy_ld, const void* w, void* dx, hipStream_t st) {
  SY11_REQUIRE(dy && w && dx && dy_ld >= d->N, "dwconv_dgrad: bad argument");
  DwArgs a{};
  bool vec;
  dim3 grid;
  int rc = dw_setup(d, a, vec, dy, dy_ld, dx, d->x_ld, (long)d->B * d->IH * d->IW, 16, grid);
  if (rc) return rc;
  a.x = dy; a.w = w; a.y = dx;
  if (dw3x3_ok(d, vec)) {
    int run, rpr; unsigned g;
    dw3x3_geom(d, a.cpv, &run, &rpr, &g);
    SY11_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw3x3_kernel<T, 16 / (int)sizeof(T), 1>), dim3(g), dim3(256), 0, st, a, dy_ld, (float*)nullptr,
                                                         run, rpr, BnTailDev{}));
    SY11_LAUNCH_CHECK("dwconv_dgrad");
    return SY11_OK;
  }
  SY11_DISPATCH_DTYPE(d->dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (vec) hipLaunchKernelGGL((dwconv_dgrad_kernel<T, VE, 9>), grid, dim3(256), 0, st, a, dy_ld);
    else hipLaunchKernelGGL((dwconv_dgrad_kernel<T, 1, 9>), grid, dim3(256), 0, st, a, dy_ld);
  });
  SY11_LAUNCH_CHECK("dwconv_dgrad");
  return SY11_OK;
}

extern "C" int sy11_conv2d_wgrad_dw(const sy11_conv_desc* d, const void* x, const void* dy, int dy_ld, float* dw, hipStream_t st) {
  SY11_REQUIRE(x && dy && dw && dy_ld >= d->N && d->x_ld >= d->C, "dwconv_wgrad: bad argument");
  DwArgs a{};
  bool vec;
  dim3 grid;
  int rc = dw_setup(d, a, vec, x, d->x_ld, dy, dy_ld, (long)d->B * d->OH * d->OW, 32, grid);
  if (rc) return rc;
  a.x = x; a.y = (void*)dy;
  DetPartials dp;                                             // ordered mode (det.h): one partial dW row per workgroup column x
  const bool det = sy11_det(2);
  float* const dw_out = dw;
  const int ncol = d->C * d->KH * d->KW;
  auto det_begin = [&](long rows) -> bool {
    if (rows <= 1) return true;
    if (!dp.acquire(st, 1, rows, ncol)) return false;
    dw = dp.buf(0); a.part_stride = ncol;
    return true;
  };
  // the windowed walk (3 loads per pixel instead of 9) also wins here since its 9 x VEC partial sums per thread are folded through a
  // parking area instead of LDS atomics (r02, tools/dw_micro.py: 75 -> 64 us on 80x80x128, 51 -> 42 on 40x40x256, 32 -> 22 on 20x20x256)
  static int win_wgrad = -1;
  if (win_wgrad < 0) { const char* e = getenv("SY11_DW_WINDOW_WGRAD"); win_wgrad = e ? atoi(e) : 1; }
  if (win_wgrad && dw3x3_ok(d, vec)) {
    int run, rpr; unsigned g;
    dw3x3_geom(d, a.cpv, &run, &rpr, &g, 512);                // every workgroup ends with 9*C filter-gradient atomics
    if (det && !det_begin(g)) SY11_FAIL(SY11_ELAUNCH, "dwconv_wgrad: ordered-reduction workspace unavailable");
    SY11_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((dw3x3_kernel<T, 16 / (int)sizeof(T), 2>), dim3(g), dim3(256), (9 * d->C + 256 * 3 * (16 / (int)sizeof(T))) * sizeof(float), st, a, dy_ld, dw,
                                                         run, rpr, BnTailDev{}));
    SY11_LAUNCH_CHECK("dwconv_wgrad");
    return dp.base ? dp.fold(0, dw_out) : SY11_OK;
  }
  if (det && !det_begin(grid.x)) SY11_FAIL(SY11_ELAUNCH, "dwconv_wgrad: ordered-reduction workspace unavailable");
  SY11_DISPATCH_DTYPE(d->dtype, T, {
    constexpr int VE = 16 / (int)sizeof(T);
    if (vec && std::is_same<T, _Float16>::value && d->KH == 3 && d->KW == 3) hipLaunchKernelGGL(dwconv_wgrad3x3_f16_kernel, grid, dim3(256), 0, st, a, dy_ld, dw);
    else if (vec) hipLaunchKernelGGL((dwconv_wgrad_kernel<T, VE, 9>), grid, dim3(256), 0, st, a, dy_ld, dw);
    else hipLaunchKernelGGL((dwconv_wgrad_kernel<T, 1, 9>), grid, dim3(256), 0, st, a, dy_ld, dw);
  });
  SY11_LAUNCH_CHECK("dwconv_wgrad");
  return dp.base ? dp.fold(0, dw_out) : SY11_OK;
}

// ------------------------------------------------------------------------------------------------ stem (Cin = 3, NCHW f32 in)
// One thread = one output pixel x all N (<= 64) output channels; the 27 x N filter sits in LDS (broadcast reads).
struct StemArgs {
  const float* x; const void* w; void* y; const float* bias; float* stat_sum; float* stat_sq;
  int B, IH, IW, OH, OW, N, y_ld, SH, SW, PH, PW;
  unsigned flags;
  int stat_slots;
  unsigned mag_ow, mag_oh;     // ceil(2^20 / OW), ceil(2^20 / OH) for the multiply-shift pixel walk (wgrad)
};

template <typename T, int NMAX>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const StemArgs a) {
  constexpr int ROWB = NMAX * (int)sizeof(T);       // bytes of one output pixel when N == NMAX
  constexpr int ROWS = ROWB + 16;                   // padded LDS row (keeps 16-byte stores off one bank group)
  constexpr int CP = ROWB / 16;                     // 16-byte chunks per pixel
  __shared__ float sw[27][NMAX];
  __shared__ float red[4][2][NMAX];                 // [wave][sum | sumsq][channel]: folded in wave order (no LDS atomics)
  __shared__ __attribute__((aligned(16))) unsigned char stage[256 * ROWS];
  for (int i = threadIdx.x; i < 27 * NMAX; i += 256) {
    const int k = i / NMAX, n = i - k * NMAX;          // k = (r*3+s)*3 + c  (filter layout [n][r][s][c])
    sw[k][n] = n < a.N ? ElemTraits<T>::to_f(((const T*)a.w)[n * 27 + k]) : 0.f;
  }
  __syncthreads();
  const int M = a.B * a.OH * a.OW;
  const int m = blockIdx.x * 256 + threadIdx.x;
  float acc[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
  const bool ok = m < M;
  if (ok) {
    const int q = m / a.OW, ox = m - q * a.OW;
    const int b = q / a.OH, oy = q - b * a.OH;
    const long plane = (long)a.IH * a.IW;
    const float* xb = a.x + (long)b * 3 * plane;
#pragma unroll 1
    for (int rs = 0; rs < 9; ++rs) {                    // NOT unrolled: 9 x (3 x NMAX) FMAs keeps the filter reads transient
      const int r = rs / 3, s_ = rs - r * 3;
      const int iy = oy * a.SH - a.PH + r, ix = ox * a.SW - a.PW + s_;
      const bool in = (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      const long off = (long)iy * a.IW + ix;
      const float v0 = in ? xb[off] : 0.f, v1 = in ? xb[plane + off] : 0.f, v2 = in ? xb[2 * plane + off] : 0.f;
      const float* w0 = sw[rs * 3];
#pragma unroll
      for (int n = 0; n < NMAX; ++n) acc[n] += v0 * w0[n] + v1 * w0[NMAX + n] + v2 * w0[2 * NMAX + n];
    }
  }
  const bool silu = a.flags & SY11_EPI_SILU;
  if (a.stat_sum) {
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      float s1 = ok ? acc[n] : 0.f, s2 = s1 * s1;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
      if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0][n] = s1; red[threadIdx.x >> 6][1][n] = s2; }
    }
    __syncthreads();
    if (threadIdx.x < a.N) {
      const long so = (long)(blockIdx.x % a.stat_slots) * a.N;
      const int n = threadIdx.x;
      atomicAdd(a.stat_sum + so + n, ((red[0][0][n] + red[1][0][n]) + red[2][0][n]) + red[3][0][n]);
      atomicAdd(a.stat_sq + so + n, ((red[0][1][n] + red[1][1][n]) + red[2][1][n]) + red[3][1][n]);
    }
  }
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    const float v = acc[n] + ((a.bias && n < a.N) ? a.bias[n] : 0.f);
    acc[n] = silu ? silu_f(v) : v;
  }
  if (a.N == NMAX && a.y_ld == NMAX) {
    // the 256 pixels of this block are one contiguous run of the NHWC output: transpose through LDS so that
    // consecutive lanes store consecutive 16-byte chunks (a thread's own pixel is 64-256 bytes apart from its neighbour's)
    T* row = (T*)(stage + threadIdx.x * ROWS);
#pragma unroll
    for (int n = 0; n < NMAX; ++n) row[n] = ElemTraits<T>::from_f(acc[n]);
    __syncthreads();
    unsigned char* yb = (unsigned char*)a.y + (long)blockIdx.x * 256 * ROWB;
    const int nvalid = min(256, M - blockIdx.x * 256);
#pragma unroll
    for (int i = 0; i < CP; ++i) {
      const int id = i * 256 + threadIdx.x;
      const int p = id / CP, cc = id - p * CP;
      if (p < nvalid) *(uint4*)(yb + (long)id * 16) = *(const uint4*)(stage + p * ROWS + cc * 16);
    }
  } else if (ok) {
    T* yp = (T*)a.y + (long)m * a.y_ld;
#pragma unroll
    for (int n = 0; n < NMAX; ++n)
      if (n < a.N) yp[n] = ElemTraits<T>::from_f(acc[n]);
  }
}

// dW[n][k] += sum_m dy[m][n] * patch[m][k]: stage 64 pixels of dy (64 x N) and patches (64 x 27) in LDS, each thread
// owns up to 4 (n,k) outputs.
template <typename T, int NMAX>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const StemArgs a, int dy_ld, float* dw, long pix_per_block, long part_stride) {
  constexpr int BP = 64;
  constexpr int NOUT = (27 * NMAX + 255) / 256;
  __shared__ float sdy[BP][NMAX + 1];
  __shared__ float sp[BP][28];
  const long M = (long)a.B * a.OH * a.OW;
  const long m0 = (long)blockIdx.x * pix_per_block;
  const long m1 = m0 + pix_per_block < M ? m0 + pix_per_block : M;
  float acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) acc[o] = 0.f;
  const long plane = (long)a.IH * a.IW;
  for (long mb = m0; mb < m1; mb += BP) {
    __syncthreads();
    for (int i = threadIdx.x; i < BP * NMAX; i += 256) {
      const int p = i / NMAX, n = i - p * NMAX;
      const long m = mb + p;
      sdy[p][n] = (m < m1 && n < a.N) ? ElemTraits<T>::to_f(((const T*)a.y)[m * dy_ld + n]) : 0.f;   // a.y = dy
    }
    for (int i = threadIdx.x; i < BP * 27; i += 256) {
      const int p = i / 27, k = i - p * 27;
      const long m = mb + p;
      float v = 0.f;
      if (m < m1) {
        const int ox = (int)(m % a.OW);
        const long q = m / a.OW;
        const int oy = (int)(q % a.OH), b = (int)(q / a.OH);
        const int c = k % 3, rs = k / 3, r = rs / 3, s = rs - r * 3;
        const int iy = oy * a.SH - a.PH + r, ix = ox * a.SW - a.PW + s;
        if ((unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW) v = a.x[((long)b * 3 + c) * plane + (long)iy * a.IW + ix];
      }
      sp[p][k] = v;
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const int id = threadIdx.x + o * 256;
      if (id < 27 * NMAX) {
        const int n = id / 27, k = id - n * 27;
        float s = 0.f;
#pragma unroll 8
        for (int p = 0; p < BP; ++p) s += sdy[p][n] * sp[p][k];
        acc[o] += s;
      }
    }
  }
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
    const int id = threadIdx.x + o * 256;
    if (id < 27 * NMAX) {
      const int n = id / 27, k = id - n * 27;
      if (n < a.N) atomicAdd(dw + (long)blockIdx.x * part_stride + n * 27 + k, acc[o]);
    }
  }
}

typedef short as16x4_t __attribute__((ext_vector_type(4)));
// ---- MFMA stem (f16 / bf16, N = 32 or 64, dense NHWC output).  The 27-tap patch is the GEMM reduction (padded to 32):
// lane (pixel, half) gathers its 16 reduction slots straight from the f32 NCHW image (L1-resident 3x3 neighbourhoods),
// packs them to 16 bit and feeds v_mfma_f32_32x32x16; the filter sits in registers as the B operand.  The accumulator
// tile has col = output channel (lane), rows = pixels (registers): BN statistics are in-lane sums + one shuffle, and
// the tile is transposed through LDS so that the NHWC output leaves in 16-byte coalesced stores.
typedef short ss16x8 __attribute__((ext_vector_type(8)));
template <typename T> struct StemMma;
template <> struct StemMma<_Float16> {
  static __device__ __forceinline__ f32x16 run(ss16x8 a, ss16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ short bits(float v) { return __builtin_bit_cast(short, (_Float16)v); }
};
template <> struct StemMma<__bf16> {
  static __device__ __forceinline__ f32x16 run(ss16x8 a, ss16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ short bits(float v) { return __builtin_bit_cast(short, (__bf16)v); }
};
template <> struct StemMma<float> {
  static __device__ __forceinline__ f32x16 run(ss16x8, ss16x8, f32x16 c) { return c; }
  static __device__ __forceinline__ short bits(float) { return 0; }
};

// reduction slot k = (r*3+s)*3 + c  ->  element offset inside the image of batch b relative to pixel (oy*SH, ox*SW)
struct StemTap { int off, dy, dx; bool ok; };
__device__ __forceinline__ StemTap stem_tap(int k, const StemArgs& a) {
  StemTap t;
  t.ok = k < 27;
  const int kk = t.ok ? k : 0;
  const int rs = kk / 3, c = kk - rs * 3, r = rs / 3, s_ = rs - r * 3;
  t.dy = r - a.PH;
  t.dx = s_ - a.PW;
  t.off = c * a.IH * a.IW + t.dy * a.IW + t.dx;
  return t;
}

template <typename T, int NT>
__global__ __launch_bounds__(256) void stem_fwd_mma(const StemArgs a, const BnTailDev tail) {
  constexpr int N = 32 * NT, ROWB = N * 2, ROWS = ROWB + 16, CP = ROWB / 16;
  __shared__ float red[4][2][N];                    // [wave][sum | sumsq][channel]: folded in wave order (no LDS atomics)
  __shared__ __attribute__((aligned(16))) unsigned char stage[256 * ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  ss16x8 wf[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 8 * half + 16 * g + i;
        wf[nt][g][i] = k < 27 ? StemMma<T>::bits(ElemTraits<T>::to_f(((const T*)a.w)[(nt * 32 + col) * 27 + k])) : (short)0;
      }
  // per reduction slot: element offset of the tap and a packed (dy + 1) | (dx + 1) << 2 | ok << 4 — two registers per slot, so that
  // BOTH 32-pixel groups of a wave can have their 16 gathers in flight at once (r02: one group at a time = two exposed memory
  // round trips per wave at 3 waves per SIMD)
  int toff[16], tpk[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const StemTap t = stem_tap(8 * half + 16 * (j >> 3) + (j & 7), a);
    toff[j] = t.off;
    tpk[j] = (t.dy + 1) | ((t.dx + 1) << 2) | ((t.ok ? 1 : 0) << 4);
  }
  const int M = a.B * a.OH * a.OW;
  const long plane3 = 3L * a.IH * a.IW;
  const bool silu = a.flags & SY11_EPI_SILU;
  float s1[NT], s2[NT], bias_v[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { s1[nt] = s2[nt] = 0.f; bias_v[nt] = a.bias ? a.bias[nt * 32 + col] : 0.f; }
  float raw[2][16];
  unsigned inmask[2];
#pragma unroll
  for (int gq = 0; gq < 2; ++gq) {
    const int gi = wave * 2 + gq;
    const int m = blockIdx.x * 256 + gi * 32 + col;
    const bool ok = m < M;
    // the group's first pixel is wave-uniform: decode it on the scalar unit, then walk `col` pixels with multiply-shift wraps
    const int base = __builtin_amdgcn_readfirstlane(min(blockIdx.x * 256 + gi * 32, M - 1));
    const int q0 = base / a.OW, b0 = q0 / a.OH;
    int ox = base - q0 * a.OW + (ok ? col : 0);
    const int wq = (int)(((unsigned)ox * a.mag_ow) >> 20);
    ox -= wq * a.OW;
    int oy = q0 - b0 * a.OH + wq;
    const int wq2 = (int)(((unsigned)oy * a.mag_oh) >> 20);
    oy -= wq2 * a.OH;
    const int b = min(b0 + wq2, a.B - 1);
    const int iy0 = oy * a.SH, ix0 = ox * a.SW;
    const float* xb = a.x + b * plane3 + (long)iy0 * a.IW + ix0;
    unsigned msk = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      // unconditional load from a clamped (always legal) address + select: a predicated load per slot compiles to 16
      // branch-and-wait sequences, i.e. 16 serialised memory latencies per 32 pixels (r01 ISA inspection)
      const int dy = (tpk[j] & 3) - 1, dx = ((tpk[j] >> 2) & 3) - 1;
      const bool in = ok && (tpk[j] & 16) && (unsigned)(iy0 + dy) < (unsigned)a.IH && (unsigned)(ix0 + dx) < (unsigned)a.IW;
      raw[gq][j] = xb[in ? toff[j] : 0];
      msk |= (in ? 1u : 0u) << j;
    }
    inmask[gq] = msk;
  }
#pragma unroll
  for (int gq = 0; gq < 2; ++gq) {
    const int gi = wave * 2 + gq;
    ss16x8 af[2];
#pragma unroll
    for (int j = 0; j < 16; ++j) af[j >> 3][j & 7] = StemMma<T>::bits(((inmask[gq] >> j) & 1u) ? raw[gq][j] : 0.f);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      acc = StemMma<T>::run(af[0], wf[nt][0], acc);
      acc = StemMma<T>::run(af[1], wf[nt][1], acc);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v0 = acc[e];
        s1[nt] += v0;
        s2[nt] += v0 * v0;
        float v = v0 + bias_v[nt];
        if (silu) v = silu_f(v);
        const int pl = gi * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
        *(T*)(stage + pl * ROWS + (nt * 32 + col) * 2) = ElemTraits<T>::from_f(v);
      }
    }
  }
  __syncthreads();                                   // red zeroed (top) and stage complete
  if (a.stat_sum) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float t1 = s1[nt] + __shfl_xor(s1[nt], 32), t2 = s2[nt] + __shfl_xor(s2[nt], 32);
      if (half == 0) { red[wave][0][nt * 32 + col] = t1; red[wave][1][nt * 32 + col] = t2; }
    }
  }
  {
    unsigned char* yb = (unsigned char*)a.y + (long)blockIdx.x * 256 * ROWB;
    const int nvalid = min(256, M - (int)blockIdx.x * 256);
#pragma unroll
    for (int i = 0; i < CP; ++i) {
      const int id = i * 256 + tid;
      const int p = id / CP, cc = id - p * CP;
      if (p < nvalid) *(uint4*)(yb + (long)id * 16) = *(const uint4*)(stage + p * ROWS + cc * 16);
    }
  }
  if (a.stat_sum) {
    __syncthreads();
    if (tid < N) {
      const long so = (long)(blockIdx.x % a.stat_slots) * N;
      atomicAdd(a.stat_sum + so + tid, ((red[0][0][tid] + red[1][0][tid]) + red[2][0][tid]) + red[3][0][tid]);
      atomicAdd(a.stat_sq + so + tid, ((red[0][1][tid] + red[1][1][tid]) + red[2][1][tid]) + red[3][1][tid]);
    }
    if (tail.ticket) bn_tail_run(tail, a.stat_sum, a.stat_sq, a.stat_slots, N, gridDim.x);
  }
}

// dW[n][k] += sum_p dy[p][n] * patch[p][k]: pixels are the MFMA reduction.  A = dy^T through the transposing LDS read
// (32-pixel x N tile per wave, row stride = 64 mod 256 bytes), B = patch column k = lane, gathered from the image.
template <typename T, int NT>
__global__ __launch_bounds__(256) void stem_wgrad_mma(const StemArgs a, float* dw, int pix_per_block, long part_stride) {
  constexpr int N = 32 * NT, ROW = (N * 2) % 256 == 64 ? N * 2 : N * 2 + 64, CPR = N / 8;
  __shared__ __attribute__((aligned(16))) unsigned char sdy[4 * 32 * ROW];
  __shared__ float sacc[NT][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int i = tid; i < NT * 32 * 33; i += 256) ((float*)sacc)[i] = 0.f;
  const int M = a.B * a.OH * a.OW;
  const int m0 = blockIdx.x * pix_per_block, m1 = min(M, m0 + pix_per_block);
  const StemTap tap = stem_tap(col, a);
  const long plane3 = 3L * a.IH * a.IW;
  const long tap_plane = (long)tap.off - ((long)tap.dy * a.IW + tap.dx);     // = c * IH * IW of this lane's reduction column
  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
  unsigned char* my = sdy + wave * 32 * ROW;
  const int trq = (lane & 15) >> 2, trp = lane & 3, trg = (lane >> 4) & 1;
  const int tr_off = (8 * half + trq) * ROW + (trg * 16 + trp * 4) * 2;
  const T* dyg = (const T*)a.y;                       // a.y carries dy, a.y_ld its pixel stride
  for (int c0 = m0; c0 < m1; c0 += 128) {
    const int p0 = c0 + wave * 32;                    // this wave's 32 pixels
    __syncthreads();
    for (int i = lane; i < 32 * CPR; i += 64) {
      const int pr = i / CPR, ch = i - pr * CPR;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (p0 + pr < m1) v = *(const uint4*)(dyg + (long)(p0 + pr) * a.y_ld + ch * 8);
      *(uint4*)(my + pr * ROW + ch * 16) = v;
    }
    // patch operand: 16 pixels of reduction column `col` for this lane (8*half + 16g + i)
    ss16x8 bf[2];
    {
      const int pw = __builtin_amdgcn_readfirstlane(min(p0, M - 1));        // wave-uniform: scalar division
      const int q = pw / a.OW;
      int ox = pw - q * a.OW + 8 * half, oy = q % a.OH, b = q / a.OH;
      // all 16 gathers are issued before the first value is used (two loops: addresses + loads, then packing).  Each load is
      // unconditional, from CLAMPED coordinates (always a legal address), and only the value is selected: `x[in ? idx : 0]`
      // compiled to 16 exec-masked branches per chunk, and packing inside the load loop to a full wait after every load (r02 ISA)
      float raw[16];
      unsigned inmask = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j == 8) ox += 16;                         // slots 8..15 are pixels 16 + 8*half + (0..7)
        int oxx = ox + (j & 7);
        const int q = (int)(((unsigned)oxx * a.mag_ow) >> 20);      // row wraps (branch-free: exact for these small ranges)
        oxx -= q * a.OW;
        int oyy = oy + q;
        const int q2 = (int)(((unsigned)oyy * a.mag_oh) >> 20);
        oyy -= q2 * a.OH;
        const int bb = min(b + q2, a.B - 1);
        const int pj = p0 + 8 * half + 16 * (j >> 3) + (j & 7);
        const int iy = oyy * a.SH + tap.dy, ix = oxx * a.SW + tap.dx;
        const bool in = pj < m1 && tap.ok && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
        const int cy = min(max(iy, 0), a.IH - 1), cx = min(max(ix, 0), a.IW - 1);
        raw[j] = a.x[bb * plane3 + tap_plane + (long)cy * a.IW + cx];
        inmask |= (in ? 1u : 0u) << j;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) bf[j >> 3][j & 7] = StemMma<T>::bits(((inmask >> j) & 1u) ? raw[j] : 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const unsigned char* pa = my + 16 * g * ROW + tr_off + nt * 64;
        const as16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) as16x4_t*)pa);
        const as16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) as16x4_t*)(pa + 4 * ROW));
        const ss16x8 af = ss16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[nt] = StemMma<T>::run(af, bf[g], acc[nt]);
      }
  }
  __syncthreads();
  for (int w = 0; w < 4; ++w) {                         // the four waves add their tiles one after the other: a fixed order
    if (wave == w) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[nt][(e & 3) + 8 * (e >> 2) + 4 * half][col] += acc[nt][e];
    }
    __syncthreads();
  }
  for (int i = tid; i < N * 27; i += 256) {
    const int n = i / 27, k = i - n * 27;
    atomicAdd(dw + (long)blockIdx.x * part_stride + i, sacc[n >> 5][n & 31][k]);
  }
}

// ---- LDS-tiled MFMA stem (r03; strides 1 and 2).  The gather kernels above spend their time on ADDRESSES: 27 scalar f32 gathers
// per pixel, each with clamped coordinates and a validity bit (stem_fwd_mma 274 us, stem_wgrad_mma 390 us for 64 x 3 x 640 x 640
// against 133 us of image + output bytes).  Here a workgroup owns a 4 x 64 block of output pixels of one image: the image rows under
// the block (<= 9 x 132 x 3 values) are read ONCE with aligned 16-byte loads, rounded to the MFMA type and laid out as
// [channel][row][column] in LDS with the zero padding already in place; an operand slot is then one ds_read_u16 at
// (lane's pixel offset + the slot's constant offset) — no coordinates, no masks.  Values and MFMA order are those of the gather
// kernels: results are bit-identical.
constexpr int STILE_H = 4, STILE_W = 64, STILE_RH = 9, STILE_RWP = 136, STILE_IT = 4;     // IT: 16-byte groups per thread per tile

// launch constants of the tiling (host: stem_tiling): the 16-byte groups per tile row and multiply-shift reciprocals for the
// per-thread group index and the (scalar) tile index
struct StemTiling { int tiles_x, tiles_y, tiles, RH, shift, NV; unsigned mag_nv, mag_tx, mag_ty; };

struct StemTile { int b, oy0, ox0; };
__device__ __forceinline__ StemTile stem_tile_at(int t, const StemTiling& g) {
  StemTile ti;                                         // workgroup-uniform: s_mul_hi_u32 instead of three emulated divisions
  const int q = g.tiles_x == 1 ? t : (int)__umulhi((unsigned)t, g.mag_tx), tx = t - q * g.tiles_x;      // (2^32 / 1 does not fit)
  ti.b = g.tiles_y == 1 ? q : (int)__umulhi((unsigned)q, g.mag_ty);
  ti.oy0 = (q - ti.b * g.tiles_y) * STILE_H;
  ti.ox0 = tx * STILE_W;
  return ti;
}

// a thread's share of the image rows under tile `ti` (global -> registers; zeros outside the image), and registers -> LDS as
// [channel][row][column] 16-bit patterns of T.  Group `id` = (channel, row, 16-byte group v); the tile's first column sits at
// element `shift` of its LDS row.  VEC (IW % 4 == 0, 16-byte aligned image): a group is wholly inside or wholly outside the
// image row, so it is ONE unconditional 16-byte load from a clamped address plus a select — no branches, all loads in flight.
struct StemRows { float f[STILE_IT][4]; };
template <bool VEC>
__device__ __forceinline__ void stem_rows_load(const StemArgs& a, const StemTiling& g, const StemTile& ti, StemRows& r) {
  const int gy0 = ti.oy0 * a.SH - a.PH, gxa = ti.ox0 * a.SW - a.PW - g.shift;
  const int plane = a.IH * a.IW;                       // B * 3 * plane < 2^31 (host)
  const float* xb = a.x + (long)ti.b * 3 * plane;
#pragma unroll
  for (int it = 0; it < STILE_IT; ++it) {
    const int id = threadIdx.x + 256 * it;
    const int rc = (int)(((unsigned)id * g.mag_nv) >> 16), v = id - rc * g.NV;
    const int c = (rc >= g.RH) + (rc >= 2 * g.RH), ry = rc - c * g.RH;
    const int gy = gy0 + ry, gx = gxa + 4 * v;
    const bool rowin = rc < 3 * g.RH && (unsigned)gy < (unsigned)a.IH;
    const int cc = min(c, 2), cy = min(max(gy, 0), a.IH - 1);
    const float* row = xb + cc * plane + cy * a.IW;
    if (VEC) {
      const bool in = rowin && gx >= 0 && gx < a.IW;
      const float4 q4 = *(const float4*)(row + min(max(gx, 0), a.IW - 4));
      r.f[it][0] = in ? q4.x : 0.f; r.f[it][1] = in ? q4.y : 0.f; r.f[it][2] = in ? q4.z : 0.f; r.f[it][3] = in ? q4.w : 0.f;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float q1 = row[min(max(gx + i, 0), a.IW - 1)];
        r.f[it][i] = (rowin && (unsigned)(gx + i) < (unsigned)a.IW) ? q1 : 0.f;
      }
    }
  }
}
template <typename T>
__device__ __forceinline__ void stem_rows_store(const StemTiling& g, const StemRows& r, short* xt) {
  typedef short s4 __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int it = 0; it < STILE_IT; ++it) {
    const int id = threadIdx.x + 256 * it;
    const int rc = (int)(((unsigned)id * g.mag_nv) >> 16), v = id - rc * g.NV;
    if (rc < 3 * g.RH)
      *(s4*)(xt + rc * STILE_RWP + 4 * v) = s4{StemMma<T>::bits(r.f[it][0]), StemMma<T>::bits(r.f[it][1]), StemMma<T>::bits(r.f[it][2]),
                                               StemMma<T>::bits(r.f[it][3])};
  }
}

// element offset of reduction slot k = (r*3+s)*3 + c inside the tile (0 for the padding slots 27..31: masked by the caller)
__device__ __forceinline__ int stem_tile_slot(int k, int RH) {
  if (k >= 27) return 0;
  const int rs = k / 3, c = k - rs * 3, r = rs / 3, s_ = rs - r * 3;
  return (c * RH + r) * STILE_RWP + s_;
}

// Persistent: workgroup w takes tiles w, w + gridDim.x, ...; the image rows of the NEXT tile are in flight (registers) while the
// current one is multiplied, the filter fragments and the statistics live in registers for the whole launch.  The MFMA computes
// the TRANSPOSED tile (A = filter: rows = output channels, B = patches: columns = pixels), so a lane ends up with 4 x 4
// consecutive channels of one pixel: the tile goes to the staging buffer in 8-byte writes (r03 first cut, pixels in registers:
// 32 two-byte LDS writes per lane and tile, a quarter of the kernel), and the per-channel statistics are per-register partial sums
// that are folded across lanes once per launch.
template <typename T, int NT, bool VEC>
__global__ __launch_bounds__(256) void stem_fwd_tile(const StemArgs a, const BnTailDev tail, const StemTiling g) {
  constexpr int N = 32 * NT, ROWB = N * 2, ROWS = ROWB + 16, CP = ROWB / 16;
  __shared__ float red[4][2][N];
  __shared__ __attribute__((aligned(16))) unsigned char stage[256 * ROWS];
  __shared__ __attribute__((aligned(16))) short xt[3 * STILE_RH * STILE_RWP];      // the image tile as 16-bit patterns of T
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  ss16x8 wf[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int gg = 0; gg < 2; ++gg)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = 8 * half + 16 * gg + i;
        wf[nt][gg][i] = k < 27 ? StemMma<T>::bits(ElemTraits<T>::to_f(((const T*)a.w)[(nt * 32 + col) * 27 + k])) : (short)0;
      }
  int toff[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) toff[j] = stem_tile_slot(8 * half + 16 * (j >> 3) + (j & 7), g.RH);
  // slots 27..31 = elements 3..7 of the second fragment of the upper lane half: cleared with three ANDs on the packed words
  const unsigned keep_lo = half ? 0x0000ffffu : 0xffffffffu, keep_hi = half ? 0u : 0xffffffffu;
  const bool silu = a.flags & SY11_EPI_SILU;
  const bool plain = !silu && !a.bias;
  float s1[NT][16], s2[NT][16], bias_v[NT][16];      // lane: pixel `col`; register e: channel (e&3) + 8*(e>>2) + 4*half (+ 32 nt)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s1[nt][e] = s2[nt][e] = 0.f;
      bias_v[nt][e] = a.bias ? a.bias[nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half] : 0.f;
    }
  StemRows rows;
  int t = blockIdx.x;
  StemTile ti = stem_tile_at(min(t, g.tiles - 1), g);
  if (t < g.tiles) stem_rows_load<VEC>(a, g, ti, rows);
  for (; t < g.tiles; t += gridDim.x) {
    __syncthreads();                                  // previous tile: xt read, stage drained
    stem_rows_store<T>(g, rows, xt);
    const StemTile cur = ti;
    if (t + (int)gridDim.x < g.tiles) {
      ti = stem_tile_at(t + gridDim.x, g);
      stem_rows_load<VEC>(a, g, ti, rows);
    }
    __syncthreads();
    const bool row_ok = cur.oy0 + wave < a.OH;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const int oxl = gq * 32 + col;
      const bool ok = row_ok && cur.ox0 + oxl < a.OW;
      const short* px = xt + wave * a.SH * STILE_RWP + g.shift + oxl * a.SW;
      ss16x8 af[2];
#pragma unroll
      for (int j = 0; j < 16; ++j) af[j >> 3][j & 7] = px[toff[j]];
      typedef unsigned u4v __attribute__((ext_vector_type(4)));
      u4v a0 = __builtin_bit_cast(u4v, af[0]), a1 = __builtin_bit_cast(u4v, af[1]);
      a1[1] &= keep_lo; a1[2] &= keep_hi; a1[3] &= keep_hi;
      if (!ok) { a0 = u4v{0, 0, 0, 0}; a1 = u4v{0, 0, 0, 0}; }
      af[0] = __builtin_bit_cast(ss16x8, a0);
      af[1] = __builtin_bit_cast(ss16x8, a1);
      unsigned char* srow = stage + (wave * 64 + oxl) * ROWS + half * 8;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        acc = StemMma<T>::run(wf[nt][0], af[0], acc);
        acc = StemMma<T>::run(wf[nt][1], af[1], acc);
        typedef T t4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          s1[nt][e] += acc[e];
          s2[nt][e] += acc[e] * acc[e];
        }
        if (plain) {                                  // training: no branch inside the unrolled element loops
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            t4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = ElemTraits<T>::from_f(acc[e4 * 4 + i]);
            *(t4*)(srow + nt * 64 + e4 * 16) = o;     // channels nt*32 + 8*e4 + 4*half + (0..3)
          }
        } else {
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            t4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float v = acc[e4 * 4 + i] + bias_v[nt][e4 * 4 + i];
              if (silu) v = silu_f(v);
              o[i] = ElemTraits<T>::from_f(v);
            }
            *(t4*)(srow + nt * 64 + e4 * 16) = o;
          }
        }
      }
    }
    __syncthreads();
    unsigned char* yimg = (unsigned char*)a.y + (long)cur.b * a.OH * a.OW * ROWB;
#pragma unroll
    for (int i = 0; i < CP; ++i) {
      const int id = i * 256 + tid;
      const int p = id / CP, cc = id - p * CP;
      const int oy = cur.oy0 + (p >> 6), ox = cur.ox0 + (p & 63);
      if (oy < a.OH && ox < a.OW) *(uint4*)(yimg + ((long)oy * a.OW + ox) * ROWB + cc * 16) = *(const uint4*)(stage + p * ROWS + cc * 16);
    }
  }
  if (a.stat_sum) {
    // fold the 32 pixel lanes of each half (a fixed butterfly), lane 0 / 32 then hold the wave's sums of their 16 channels per nt
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float u1 = s1[nt][e], u2 = s2[nt][e];
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) { u1 += __shfl_xor(u1, m); u2 += __shfl_xor(u2, m); }
        if (col == 0) {
          const int ch = nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
          red[wave][0][ch] = u1;
          red[wave][1][ch] = u2;
        }
      }
    __syncthreads();
    if (tid < N) {
      const long so = (long)(blockIdx.x % a.stat_slots) * N;
      atomicAdd(a.stat_sum + so + tid, ((red[0][0][tid] + red[1][0][tid]) + red[2][0][tid]) + red[3][0][tid]);
      atomicAdd(a.stat_sq + so + tid, ((red[0][1][tid] + red[1][1][tid]) + red[2][1][tid]) + red[3][1][tid]);
    }
    if (tail.ticket) bn_tail_run(tail, a.stat_sum, a.stat_sq, a.stat_slots, N, gridDim.x);
  }
}

// dW[n][k] += sum_p dy[p][n] * patch[p][k] over tiles blockIdx.x, + gridDim.x, ...: A = dy^T through the transposing LDS read
// (each wave stages the dy rows of ITS 32 pixels: wave-private, no workgroup barrier), B = reduction column k = lane out of the
// tile.  The next tile's image rows and dy rows are in flight while the current tile is multiplied.
template <typename T, int NT, bool VEC>
__global__ __launch_bounds__(256) void stem_wgrad_tile(const StemArgs a, float* dw, long part_stride, const StemTiling g) {
  constexpr int N = 32 * NT, ROW = (N * 2) % 256 == 64 ? N * 2 : N * 2 + 64, CPR = N / 8, DU = (32 * CPR) / 64;
  __shared__ __attribute__((aligned(16))) unsigned char sdy[4 * 32 * ROW];
  __shared__ float sacc[NT][32][33];
  __shared__ __attribute__((aligned(16))) short xt[3 * STILE_RH * STILE_RWP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int i = tid; i < NT * 32 * 33; i += 256) ((float*)sacc)[i] = 0.f;
  f32x16 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
  unsigned char* my = sdy + wave * 32 * ROW;
  const int trq = (lane & 15) >> 2, trp = lane & 3, trg = (lane >> 4) & 1;
  const int tr_off = (8 * half + trq) * ROW + (trg * 16 + trp * 4) * 2;
  const T* dyg = (const T*)a.y;                       // a.y carries dy, a.y_ld its pixel stride
  const bool kcol = col < 27;
  const short* pk = xt + wave * a.SH * STILE_RWP + g.shift + stem_tile_slot(col, g.RH);
  StemRows rows;
  uint4 dv[2][DU];
  auto dy_load = [&](const StemTile& ti) {
    const int oy = ti.oy0 + wave;
    const long mrow = ((long)ti.b * a.OH + min(oy, a.OH - 1)) * a.OW;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int i = lane + 64 * u, pr = i / CPR, ch = i - pr * CPR;
        const int ox = ti.ox0 + sub * 32 + pr;
        const uint4 q4 = *(const uint4*)(dyg + (mrow + min(ox, a.OW - 1)) * a.y_ld + ch * 8);     // clamped, then selected
        dv[sub][u] = (oy < a.OH && ox < a.OW) ? q4 : make_uint4(0, 0, 0, 0);
      }
  };
  int t = blockIdx.x;
  StemTile ti = stem_tile_at(min(t, g.tiles - 1), g);
  if (t < g.tiles) { stem_rows_load<VEC>(a, g, ti, rows); dy_load(ti); }
  for (; t < g.tiles; t += gridDim.x) {
    __syncthreads();                                  // every wave is done with the previous tile
    stem_rows_store<T>(g, rows, xt);
    uint4 dc[2][DU];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int u = 0; u < DU; ++u) dc[sub][u] = dv[sub][u];
    if (t + (int)gridDim.x < g.tiles) {
      ti = stem_tile_at(t + gridDim.x, g);
      stem_rows_load<VEC>(a, g, ti, rows);
      dy_load(ti);
    }
    __syncthreads();                                  // tile complete
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int i = lane + 64 * u, pr = i / CPR, ch = i - pr * CPR;
        *(uint4*)(my + pr * ROW + ch * 16) = dc[sub][u];
      }
      ss16x8 bf[2];
#pragma unroll
      for (int j = 0; j < 16; ++j) bf[j >> 3][j & 7] = pk[(sub * 32 + 8 * half + 16 * (j >> 3) + (j & 7)) * a.SW];
      if (!kcol) { bf[0] = ss16x8{0, 0, 0, 0, 0, 0, 0, 0}; bf[1] = bf[0]; }
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const unsigned char* pa = my + 16 * gg * ROW + tr_off + nt * 64;
          const as16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) as16x4_t*)pa);
          const as16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) as16x4_t*)(pa + 4 * ROW));
          const ss16x8 af = ss16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          acc[nt] = StemMma<T>::run(af, bf[gg], acc[nt]);
        }
    }
  }
  __syncthreads();
  for (int w = 0; w < 4; ++w) {                         // the four waves add their tiles one after the other: a fixed order
    if (wave == w) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int e = 0; e < 16; ++e) sacc[nt][(e & 3) + 8 * (e >> 2) + 4 * half][col] += acc[nt][e];
    }
    __syncthreads();
  }
  for (int i = tid; i < N * 27; i += 256) {
    const int n = i / 27, k = i - n * 27;
    atomicAdd(dw + (long)blockIdx.x * part_stride + i, sacc[n >> 5][n & 31][k]);
  }
}

// tiling of a launch; false: strides other than 1 / 2 or shapes outside the reciprocals' exact range (the gather kernels take those)
static bool stem_tiling(const sy11_conv_desc* d, StemTiling* g) {
  if (d->SH < 1 || d->SH > 2 || d->SW < 1 || d->SW > 2 || d->PW < 0 || d->PW > 3 || d->IW < 4) return false;
  g->tiles_x = (d->OW + STILE_W - 1) / STILE_W;
  g->tiles_y = (d->OH + STILE_H - 1) / STILE_H;
  const long tiles = (long)d->B * g->tiles_y * g->tiles_x;
  // q = umulhi(t, ceil(2^32 / d)) is exact while t * d < 2^32
  if (tiles * g->tiles_x >= (1L << 31) || tiles * g->tiles_y >= (1L << 31)) return false;
  g->tiles = (int)tiles;
  g->mag_tx = (unsigned)(((1UL << 32) + g->tiles_x - 1) / g->tiles_x);     // (a divisor of 1 is special-cased in stem_tile_at)
  g->mag_ty = (unsigned)(((1UL << 32) + g->tiles_y - 1) / g->tiles_y);
  g->RH = (STILE_H - 1) * d->SH + 3;
  g->shift = (-d->PW) & 3;                             // tile origins are multiples of 64 * SW columns: one shift for all tiles
  g->NV = (g->shift + (STILE_W - 1) * d->SW + 3 + 3) >> 2;
  g->mag_nv = (65536u + g->NV - 1) / g->NV;            // exact quotients for id < 1024 (checked for NV <= 39)
  return 3 * g->RH * g->NV <= 256 * STILE_IT;
}

static int stem_check(const sy11_conv_desc* d, const char* who) {
  SY11_REQUIRE(d && dtype_ok(d->dtype), "%s: bad desc", who);
  SY11_REQUIRE(d->C == 3 && d->KH == 3 && d->KW == 3 && d->DH == 1 && d->DW == 1 && d->groups == 1, "%s: stem kernel is 3x3, Cin=3, dilation 1", who);
  SY11_REQUIRE(d->N > 0 && d->N <= 64, "%s: N must be <= 64", who);
  SY11_REQUIRE(d->OH == (d->IH + 2 * d->PH - 3) / d->SH + 1 && d->OW == (d->IW + 2 * d->PW - 3) / d->SW + 1, "%s: OH/OW mismatch", who);
  SY11_REQUIRE((long)d->B * d->OH * d->OW < (1L << 31), "%s: too many pixels", who);
  return SY11_OK;
}

static int stem_fwd_tail(const sy11_conv_desc* d, const float* x_nchw, const void* w, const float* bias, void* y, float* stat_sum,
                         float* stat_sq, void* stream, const BnTailDev& tail, bool* tail_done);

extern "C" int sy11_stem_conv_fwd(const sy11_conv_desc* d, const float* x_nchw, const void* w, const float* bias, void* y,
                                  float* stat_sum, float* stat_sq, void* stream) {
  bool done;
  return stem_fwd_tail(d, x_nchw, w, bias, y, stat_sum, stat_sq, stream, BnTailDev{}, &done);
}

extern "C" int sy11_stem_conv_fwd_bn(const sy11_conv_desc* d, const float* x_nchw, const void* w, void* y, float* stat_sum, float* stat_sq,
                                     const sy11_bn_tail* bn, void* stream) {
  SY11_REQUIRE(d && stat_sum && stat_sq && bn && bn->gamma && bn->beta && bn->mean && bn->rstd && bn->scale && bn->shift && bn->ticket &&
                   bn->count > 0, "stem_conv_fwd_bn: statistics rows and a complete sy11_bn_tail are required");
  bool done = false;
  int rc = stem_fwd_tail(d, x_nchw, w, nullptr, y, stat_sum, stat_sq, stream, bn_tail_dev(bn, d->N, 0, 0), &done);
  if (rc || done) return rc;
  return sy11_bn_finalize(d->N, d->stat_slots > 1 ? d->stat_slots : 1, bn->count, stat_sum, stat_sq, bn->gamma, bn->beta, bn->eps, bn->momentum,
                          bn->running_mean, bn->running_var, bn->mean, bn->rstd, bn->scale, bn->shift, stream);
}

static int stem_fwd_tail(const sy11_conv_desc* d, const float* x_nchw, const void* w, const float* bias, void* y, float* stat_sum,
                         float* stat_sq, void* stream, const BnTailDev& tail, bool* tail_done) {
  *tail_done = false;
  int rc = stem_check(d, "stem_conv_fwd");
  if (rc) return rc;
  SY11_REQUIRE(x_nchw && w && y && d->y_ld >= d->N, "stem_conv_fwd: bad argument");
  StemArgs a{x_nchw, w, y, bias, stat_sum, stat_sq, d->B, d->IH, d->IW, d->OH, d->OW, d->N, d->y_ld, d->SH, d->SW, d->PH, d->PW, d->flags, d->stat_slots > 1 ? d->stat_slots : 1};
  const long M = (long)d->B * d->OH * d->OW;
  dim3 grid((unsigned)((M + 255) / 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  const bool mma = d->dtype != SY11_F32 && (d->N == 32 || d->N == 64) && d->y_ld == d->N && ((uintptr_t)y & 15) == 0 &&
                   (long)d->B * 3 * d->IH * d->IW < (1L << 31) && (long)(d->OW + 64) * d->OW < (1L << 20) && (long)(d->OH + 64) * d->OH < (1L << 20);
  static int tile_env = -1, tile_wg = 0;
  if (tile_env < 0) {
    const char* e = getenv("SY11_STEM_TILE"); tile_env = e ? atoi(e) : 1;
    const char* w = getenv("SY11_STEM_FWD_WG"); tile_wg = w ? atoi(w) : 1024;      // persistent workgroups (4 per CU)
  }
  StemTiling tg{};
  const bool tiled = mma && tile_env && stem_tiling(d, &tg);
  if (tiled) grid = dim3((unsigned)(tg.tiles < tile_wg ? tg.tiles : tile_wg));
  DetPartials dp;                                             // ordered mode (det.h): one partial statistics row per workgroup
  const bool det = stat_sum && sy11_det(2) && !tail.ticket;
  if (det) {
    if (!dp.acquire(st, 2, grid.x, d->N)) SY11_FAIL(SY11_ELAUNCH, "stem_conv_fwd: ordered-reduction workspace unavailable");
    a.stat_sum = dp.buf(0); a.stat_sq = dp.buf(1); a.stat_slots = (int)grid.x;
  }
  auto det_end = [&]() -> int { return dp.fold01(stat_sum, stat_sq, d->stat_slots > 1 ? d->stat_slots : 1, d->N); };
  if (mma) {
    a.mag_ow = (unsigned)(((1u << 20) + d->OW - 1) / d->OW);
    a.mag_oh = (unsigned)(((1u << 20) + d->OH - 1) / d->OH);
    if (tiled) {
      SY11_DISPATCH_DTYPE(d->dtype, T, {
        const bool vec = (d->IW & 3) == 0 && ((uintptr_t)x_nchw & 15) == 0;
        if (d->N == 32) { if (vec) hipLaunchKernelGGL((stem_fwd_tile<T, 1, true>), grid, block, 0, st, a, tail, tg); else hipLaunchKernelGGL((stem_fwd_tile<T, 1, false>), grid, block, 0, st, a, tail, tg); }
        else { if (vec) hipLaunchKernelGGL((stem_fwd_tile<T, 2, true>), grid, block, 0, st, a, tail, tg); else hipLaunchKernelGGL((stem_fwd_tile<T, 2, false>), grid, block, 0, st, a, tail, tg); }
      });
    } else {
      SY11_DISPATCH_DTYPE(d->dtype, T, {
        if (d->N == 32) hipLaunchKernelGGL((stem_fwd_mma<T, 1>), grid, block, 0, st, a, tail);
        else hipLaunchKernelGGL((stem_fwd_mma<T, 2>), grid, block, 0, st, a, tail);
      });
    }
    SY11_LAUNCH_CHECK("stem_conv_fwd");
    *tail_done = tail.ticket != nullptr && stat_sum != nullptr;
    return det ? det_end() : SY11_OK;
  }
  if (d->N <= 16) { SY11_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((stem_fwd_kernel<T, 16>), grid, block, 0, st, a)); }
  else if (d->N <= 32) { SY11_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((stem_fwd_kernel<T, 32>), grid, block, 0, st, a)); }
  else { SY11_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((stem_fwd_kernel<T, 64>), grid, block, 0, st, a)); }
  SY11_LAUNCH_CHECK("stem_conv_fwd");
  return det ? det_end() : SY11_OK;
}

extern "C" int sy11_stem_conv_wgrad(const sy11_conv_desc* d, const float* x_nchw, const void* dy, int32_t dy_ld, float* dw,
                                    void* stream) {
  int rc = stem_check(d, "stem_conv_wgrad");
  if (rc) return rc;
  SY11_REQUIRE(x_nchw && dy && dw && dy_ld >= d->N, "stem_conv_wgrad: bad argument");
  StemArgs a{x_nchw, nullptr, (void*)dy, nullptr, nullptr, nullptr, d->B, d->IH, d->IW, d->OH, d->OW, d->N, 0, d->SH, d->SW, d->PH, d->PW, 0, 1};
  const long M = (long)d->B * d->OH * d->OW;
  long nblk = (M + 1023) / 1024;
  if (nblk > 2048) nblk = 2048;
  const long ppb = ((M + nblk - 1) / nblk + 63) / 64 * 64;
  nblk = (M + ppb - 1) / ppb;
  dim3 grid((unsigned)nblk), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype != SY11_F32 && (d->N == 32 || d->N == 64) && dy_ld % 8 == 0 && ((uintptr_t)dy & 15) == 0 &&
      (long)d->B * 3 * d->IH * d->IW < (1L << 31)) {
    StemArgs am = a;
    am.y_ld = dy_ld;
    SY11_REQUIRE((long)(d->OW + 64) * d->OW < (1L << 20) && (long)(d->OH + 64) * d->OH < (1L << 20), "stem_conv_wgrad: output map larger than 960 pixels per side");
    am.mag_ow = (unsigned)(((1u << 20) + d->OW - 1) / d->OW);
    am.mag_oh = (unsigned)(((1u << 20) + d->OH - 1) / d->OH);
    long nb = (M + 2047) / 2048;                       // >= 16 chunks of 128 pixels per workgroup, at most 2 workgroups per CU
    if (nb > 512) nb = 512;
    const int ppbm = (int)(((M + nb - 1) / nb + 127) / 128 * 128);
    dim3 gm((unsigned)((M + ppbm - 1) / ppbm));
    static int tile_env = -1, tile_wg = 0;
    if (tile_env < 0) {
      const char* e = getenv("SY11_STEM_TILE"); tile_env = e ? atoi(e) : 1;
      const char* g = getenv("SY11_STEM_WGRAD_WG"); tile_wg = g ? atoi(g) : 512;
    }
    StemTiling tg{};
    const bool tiled = tile_env && stem_tiling(d, &tg);
    if (tiled) {                                        // >= 8 tiles (2048 pixels) per workgroup, at most `tile_wg` workgroups
      long nbt = (tg.tiles + 7) / 8;
      if (nbt > tile_wg) nbt = tile_wg;
      gm = dim3((unsigned)nbt);
    }
    DetPartials dp;                                           // ordered mode (det.h): one partial dW row per workgroup
    float* dwk = dw;
    long pstride = 0;
    if (sy11_det(2) && gm.x > 1) {
      if (!dp.acquire(st, 1, gm.x, d->N * 27)) SY11_FAIL(SY11_ELAUNCH, "stem_conv_wgrad: ordered-reduction workspace unavailable");
      dwk = dp.buf(0); pstride = d->N * 27;
    }
    if (tiled) {
      SY11_DISPATCH_DTYPE(d->dtype, T, {
        const bool vec = (d->IW & 3) == 0 && ((uintptr_t)x_nchw & 15) == 0;
        if (d->N == 32) { if (vec) hipLaunchKernelGGL((stem_wgrad_tile<T, 1, true>), gm, block, 0, st, am, dwk, pstride, tg); else hipLaunchKernelGGL((stem_wgrad_tile<T, 1, false>), gm, block, 0, st, am, dwk, pstride, tg); }
        else { if (vec) hipLaunchKernelGGL((stem_wgrad_tile<T, 2, true>), gm, block, 0, st, am, dwk, pstride, tg); else hipLaunchKernelGGL((stem_wgrad_tile<T, 2, false>), gm, block, 0, st, am, dwk, pstride, tg); }
      });
    } else {
      SY11_DISPATCH_DTYPE(d->dtype, T, {
        if (d->N == 32) hipLaunchKernelGGL((stem_wgrad_mma<T, 1>), gm, block, 0, st, am, dwk, ppbm, pstride);
        else hipLaunchKernelGGL((stem_wgrad_mma<T, 2>), gm, block, 0, st, am, dwk, ppbm, pstride);
      });
    }
    SY11_LAUNCH_CHECK("stem_conv_wgrad");
    return dp.base ? dp.fold(0, dw) : SY11_OK;
  }
  DetPartials dp;
  float* dwk = dw;
  long pstride = 0;
  if (sy11_det(2) && grid.x > 1) {
    if (!dp.acquire(st, 1, grid.x, d->N * 27)) SY11_FAIL(SY11_ELAUNCH, "stem_conv_wgrad: ordered-reduction workspace unavailable");
    dwk = dp.buf(0); pstride = d->N * 27;
  }
  if (d->N <= 16) { SY11_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((stem_wgrad_kernel<T, 16>), grid, block, 0, st, a, dy_ld, dwk, ppb, pstride)); }
  else if (d->N <= 32) { SY11_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((stem_wgrad_kernel<T, 32>), grid, block, 0, st, a, dy_ld, dwk, ppb, pstride)); }
  else { SY11_DISPATCH_DTYPE(d->dtype, T, hipLaunchKernelGGL((stem_wgrad_kernel<T, 64>), grid, block, 0, st, a, dy_ld, dwk, ppb, pstride)); }
  SY11_LAUNCH_CHECK("stem_conv_wgrad");
  return dp.base ? dp.fold(0, dw) : SY11_OK;
}
