// attention.hip — C2PSA attention core on gfx950 (N = H*W tokens, small: 400 at 640x640 input).
//
// qkv: NHWC pixels (B, N, heads*(2*kd+hd)); per head h the channels are [q(kd) | k(kd) | v(hd)]
// (the reference's qkv.view(B, heads, 2*kd+hd, N).split(...), nn/modules/block.py:1925-1927).
//   P[b,h,i,j] = softmax_j( scale * sum_d q[i,d] k[j,d] ),   o[b,i,h*hd+e] = sum_j P[i,j] v[j,e]
// 0.6 % of the network FLOPs: kept on the VALU with LDS-resident score rows (QT = 32 queries x N keys, f32);
// P is written out for the backward pass exactly as autograd keeps the softmax output.
#include "common.h"

#define ATT_QT 32
constexpr size_t ATT_LDS_MAX = 160 * 1024;     // score rows (QT x N f32) stay in LDS up to here

// GS (r04): the QT x N score rows live in the workgroup's own rows of P (global; same [row][N] layout) instead of LDS — the path for
// token counts whose rows do not fit 160 KB (f32 at N > ~1 200; the 16-bit MFMA kernels never hold score rows).  Rows past N are
// never touched in that form.
template <typename T, bool GS = false>
__global__ __launch_bounds__(256) void attention_fwd_kernel(int B, int N, int heads, int kd, int hd, const T* __restrict__ qkv, int qkv_ld,
                                                            T* __restrict__ o, int o_ld, float* __restrict__ p, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * ATT_QT;
  float* sS = GS ? p + (((long)b * heads + h) * N + q0) * N : sm;       // [QT][N]
  float* sQ = GS ? sm : sm + ATT_QT * N;                                 // [QT][kd]
  float* sK = sQ + ATT_QT * kd;                                          // [64][kd+1]
  const int hc = 2 * kd + hd;
  const T* base = qkv + (long)b * N * qkv_ld + h * hc;
  for (int i = tid; i < ATT_QT * kd; i += 256) {
    const int qi = i / kd, d = i - qi * kd;
    sQ[i] = (q0 + qi < N) ? ElemTraits<T>::to_f(base[(long)(q0 + qi) * qkv_ld + d]) * scale : 0.f;
  }
  for (int j0 = 0; j0 < N; j0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * kd; i += 256) {
      const int j = i / kd, d = i - j * kd;
      sK[j * (kd + 1) + d] = (j0 + j < N) ? ElemTraits<T>::to_f(base[(long)(j0 + j) * qkv_ld + kd + d]) : 0.f;
    }
    __syncthreads();
    const int j = tid & 63, qg = tid >> 6;      // 4 query groups of 8
    if (j0 + j < N) {
#pragma unroll
      for (int qq = 0; qq < ATT_QT / 4; ++qq) {
        const int qi = qg * (ATT_QT / 4) + qq;
        if (GS && q0 + qi >= N) continue;
        float s = 0.f;
        for (int d = 0; d < kd; ++d) s += sQ[qi * kd + d] * sK[j * (kd + 1) + d];
        sS[qi * N + j0 + j] = s;
      }
    }
  }
  __syncthreads();
  // softmax: 8 lanes per query row
  {
    const int qi = tid >> 3, l = tid & 7;
    const bool live = q0 + qi < N;
    const int n_row = (GS && !live) ? 0 : N;                 // global rows past N do not exist
    float mx = -INFINITY;
    for (int j = l; j < n_row; j += 8) mx = fmaxf(mx, sS[qi * N + j]);
#pragma unroll
    for (int o_ = 4; o_ >= 1; o_ >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o_));
    float sum = 0.f;
    for (int j = l; j < n_row; j += 8) {
      const float e = __expf(sS[qi * N + j] - mx);
      sS[qi * N + j] = e;
      sum += e;
    }
#pragma unroll
    for (int o_ = 4; o_ >= 1; o_ >>= 1) sum += __shfl_xor(sum, o_);
    const float inv = 1.f / sum;
    float* prow = p + (((long)b * heads + h) * N + (q0 + qi)) * N;
    for (int j = l; j < n_row; j += 8) {
      const float v = sS[qi * N + j] * inv;
      sS[qi * N + j] = v;
      if (!GS && live) prow[j] = v;
    }
  }
  __syncthreads();
  // o = P v
  for (int idx = tid; idx < ATT_QT * hd; idx += 256) {
    const int qi = idx / hd, e = idx - qi * hd;
    if (q0 + qi >= N) continue;
    float acc = 0.f;
    const T* vp = base + 2 * kd + e;
    for (int j = 0; j < N; ++j) acc += sS[qi * N + j] * ElemTraits<T>::to_f(vp[(long)j * qkv_ld]);
    o[((long)b * N + q0 + qi) * o_ld + h * hd + e] = ElemTraits<T>::from_f(acc);
  }
}

// backward A: per query tile.  dP = dO v^T, delta_i = sum_j dP P, dS = P (dP - delta) -> ws;  dq = scale * dS k
template <typename T, bool GS = false>
__global__ __launch_bounds__(256) void attention_bwd_q_kernel(int B, int N, int heads, int kd, int hd, const T* __restrict__ qkv, int qkv_ld,
                                                              const float* __restrict__ p, const T* __restrict__ d_o, int do_ld,
                                                              T* __restrict__ dqkv, int dqkv_ld, float* __restrict__ ds, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * ATT_QT;
  float* sS = GS ? ds + (((long)b * heads + h) * N + q0) * N : sm;       // [QT][N] dP then dS (GS: the workgroup's own rows of the dS workspace)
  float* sO = GS ? sm : sm + ATT_QT * N;                                  // [QT][hd] dO tile
  const int hc = 2 * kd + hd;
  const T* base = qkv + (long)b * N * qkv_ld + h * hc;
  for (int i = tid; i < ATT_QT * hd; i += 256) {
    const int qi = i / hd, e = i - qi * hd;
    sO[i] = (q0 + qi < N) ? ElemTraits<T>::to_f(d_o[((long)b * N + q0 + qi) * do_ld + h * hd + e]) : 0.f;
  }
  __syncthreads();
  for (int idx = tid; idx < (GS ? min(ATT_QT, N - q0) : ATT_QT) * N; idx += 256) {
    const int qi = idx / N, j = idx - qi * N;
    float s = 0.f;
    const T* vp = base + (long)j * qkv_ld + 2 * kd;
    for (int e = 0; e < hd; ++e) s += sO[qi * hd + e] * ElemTraits<T>::to_f(vp[e]);
    sS[idx] = s;
  }
  __syncthreads();
  {
    const int qi = tid >> 3, l = tid & 7;
    const bool live = q0 + qi < N;
    const float* prow = p + (((long)b * heads + h) * N + (live ? q0 + qi : 0)) * N;
    float dl = 0.f;
    if (live) for (int j = l; j < N; j += 8) dl += sS[qi * N + j] * prow[j];
#pragma unroll
    for (int o_ = 4; o_ >= 1; o_ >>= 1) dl += __shfl_xor(dl, o_);
    float* dsrow = ds + (((long)b * heads + h) * N + (q0 + qi)) * N;
    for (int j = l; j < N; j += 8) {
      if (GS && !live) break;
      const float v = live ? prow[j] * (sS[qi * N + j] - dl) : 0.f;
      sS[qi * N + j] = v;
      if (!GS && live) dsrow[j] = v;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < ATT_QT * kd; idx += 256) {
    const int qi = idx / kd, d = idx - qi * kd;
    if (q0 + qi >= N) continue;
    float acc = 0.f;
    const T* kp = base + kd + d;
    for (int j = 0; j < N; ++j) acc += sS[qi * N + j] * ElemTraits<T>::to_f(kp[(long)j * qkv_ld]);
    dqkv[((long)b * N + q0 + qi) * dqkv_ld + h * hc + d] = ElemTraits<T>::from_f(acc * scale);
  }
}

// backward B: per key tile of 32.  dv[j,e] = sum_i P[i,j] dO[i,e];  dk[j,d] = scale * sum_i dS[i,j] q[i,d]
template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_kv_kernel(int B, int N, int heads, int kd, int hd, const T* __restrict__ qkv, int qkv_ld,
                                                               const float* __restrict__ p, const float* __restrict__ ds,
                                                               const T* __restrict__ d_o, int do_ld, T* __restrict__ dqkv, int dqkv_ld, float scale) {
  __shared__ float sP[64][33], sD[64][33];
  const int tid = threadIdx.x;
  const int kt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int j0 = kt * 32;
  const int hc = 2 * kd + hd;
  const T* base = qkv + (long)b * N * qkv_ld + h * hc;
  const float* pb = p + ((long)b * heads + h) * N * N;
  const float* db = ds + ((long)b * heads + h) * N * N;
  // each thread owns fixed (key, column) outputs; columns [0,hd) are dv, [hd, hd+kd) are dk
  const int ncol = hd + kd;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int i0 = 0; i0 < N; i0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * 32; i += 256) {
      const int qi = i >> 5, j = i & 31;
      const bool ok = (i0 + qi < N) && (j0 + j < N);
      sP[qi][j] = ok ? pb[(long)(i0 + qi) * N + j0 + j] : 0.f;
      sD[qi][j] = ok ? db[(long)(i0 + qi) * N + j0 + j] : 0.f;
    }
    __syncthreads();
    const int imax = min(64, N - i0);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int id = tid + u * 256;
      const int j = id / ncol, cidx = id - j * ncol;
      float a = 0.f;
      if (id >= 32 * ncol) {
      } else if (cidx < hd) {
        const T* dp = d_o + ((long)b * N + i0) * do_ld + h * hd + cidx;
        for (int qi = 0; qi < imax; ++qi) a += sP[qi][j] * ElemTraits<T>::to_f(dp[(long)qi * do_ld]);
      } else {
        const T* qp = base + (long)i0 * qkv_ld + (cidx - hd);
        for (int qi = 0; qi < imax; ++qi) a += sD[qi][j] * ElemTraits<T>::to_f(qp[(long)qi * qkv_ld]);
      }
      acc[u] += a;
    }
  }
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    const int id = tid + u * 256;
    const int j = id / ncol, cidx = id - j * ncol;
    if (id >= 32 * ncol || j0 + j >= N) continue;
    T* out = dqkv + ((long)b * N + j0 + j) * dqkv_ld + h * hc;
    if (cidx < hd) out[2 * kd + cidx] = ElemTraits<T>::from_f(acc[u]);
    else out[kd + (cidx - hd)] = ElemTraits<T>::from_f(acc[u] * scale);
  }
}


// ================================================================================================ fast path
// KD / HD compile-time (yolo11: kd = 32, hd = 64 at every scale).  Every inner loop reads LDS only: K / V / dO / q
// chunks of 64 rows are staged once per block-iteration with coalesced global loads; score rows are wave-uniform
// (broadcast) reads, the staged operand is read conflict-free.
template <typename T, int KD, int HD>
__global__ __launch_bounds__(256) void attention_fwd_fast(int B, int N, int heads, const T* __restrict__ qkv, int qkv_ld,
                                                          T* __restrict__ o, int o_ld, float* __restrict__ p, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int CH = 64;                          // keys per staged chunk
  constexpr int G = 256 / HD, QPG = ATT_QT / G;   // PV: query groups / queries per group
  float* sS = sm;                                 // [QT][N]
  float* sQ = sS + ATT_QT * N;                    // [QT][KD]
  float* sX = sQ + ATT_QT * KD;                   // [CH][max(KD+1, HD)]  K chunk, then V chunk
  const int tid = threadIdx.x;
  const int q0 = blockIdx.x * ATT_QT, h = blockIdx.y, b = blockIdx.z;
  constexpr int HC = 2 * KD + HD;
  const T* base = qkv + (long)b * N * qkv_ld + h * HC;
  for (int i = tid; i < ATT_QT * KD; i += 256) {
    const int qi = i / KD, d = i - qi * KD;
    sQ[i] = (q0 + qi < N) ? ElemTraits<T>::to_f(base[(long)(q0 + qi) * qkv_ld + d]) * scale : 0.f;
  }
  for (int j0 = 0; j0 < N; j0 += CH) {
    __syncthreads();
    for (int i = tid; i < CH * KD; i += 256) {
      const int j = i / KD, d = i - j * KD;
      sX[j * (KD + 1) + d] = (j0 + j < N) ? ElemTraits<T>::to_f(base[(long)(j0 + j) * qkv_ld + KD + d]) : 0.f;
    }
    __syncthreads();
    const int j = tid & 63, qg = tid >> 6;
    if (j0 + j < N) {
      float kr[KD];
#pragma unroll
      for (int d = 0; d < KD; ++d) kr[d] = sX[j * (KD + 1) + d];
#pragma unroll
      for (int qq = 0; qq < ATT_QT / 4; ++qq) {
        const int qi = qg * (ATT_QT / 4) + qq;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < KD; ++d) s += sQ[qi * KD + d] * kr[d];
        sS[qi * N + j0 + j] = s;
      }
    }
  }
  __syncthreads();
  {
    const int qi = tid >> 3, l = tid & 7;
    const bool live = q0 + qi < N;
    const int n_row = N;
    float mx = -INFINITY;
    for (int j = l; j < n_row; j += 8) mx = fmaxf(mx, sS[qi * N + j]);
#pragma unroll
    for (int o_ = 4; o_ >= 1; o_ >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o_));
    float sum = 0.f;
    for (int j = l; j < n_row; j += 8) {
      const float e = __expf(sS[qi * N + j] - mx);
      sS[qi * N + j] = e;
      sum += e;
    }
#pragma unroll
    for (int o_ = 4; o_ >= 1; o_ >>= 1) sum += __shfl_xor(sum, o_);
    const float inv = 1.f / sum;
    float* prow = p + (((long)b * heads + h) * N + (q0 + qi)) * N;
    for (int j = l; j < n_row; j += 8) {
      const float v = sS[qi * N + j] * inv;
      sS[qi * N + j] = v;
      if (live) prow[j] = v;
    }
  }
  // o = P v, V staged in CH-key chunks
  const int e = tid % HD, g = tid / HD;
  float acc[QPG];
#pragma unroll
  for (int i = 0; i < QPG; ++i) acc[i] = 0.f;
  for (int j0 = 0; j0 < N; j0 += CH) {
    __syncthreads();
    for (int i = tid; i < CH * HD; i += 256) {
      const int j = i / HD, ee = i - j * HD;
      sX[j * HD + ee] = (j0 + j < N) ? ElemTraits<T>::to_f(base[(long)(j0 + j) * qkv_ld + 2 * KD + ee]) : 0.f;
    }
    __syncthreads();
    const int jn = min(CH, N - j0);
    for (int j = 0; j < jn; ++j) {
      const float v = sX[j * HD + e];
#pragma unroll
      for (int i = 0; i < QPG; ++i) acc[i] += sS[(g * QPG + i) * N + j0 + j] * v;
    }
  }
#pragma unroll
  for (int i = 0; i < QPG; ++i) {
    const int qi = g * QPG + i;
    if (q0 + qi < N) o[((long)b * N + q0 + qi) * o_ld + h * HD + e] = ElemTraits<T>::from_f(acc[i]);
  }
}

template <typename T, int KD, int HD>
__global__ __launch_bounds__(256) void attention_bwd_q_fast(int B, int N, int heads, const T* __restrict__ qkv, int qkv_ld,
                                                            const float* __restrict__ p, const T* __restrict__ d_o, int do_ld,
                                                            T* __restrict__ dqkv, int dqkv_ld, float* __restrict__ ds, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int CH = 64;
  constexpr int G = 256 / KD, QPG = ATT_QT / G;
  float* sS = sm;                        // [QT][N]
  float* sO = sS + ATT_QT * N;           // [QT][HD]
  float* sX = sO + ATT_QT * HD;          // [CH][HD+1] V chunk, then [CH][KD] K chunk
  const int tid = threadIdx.x;
  const int q0 = blockIdx.x * ATT_QT, h = blockIdx.y, b = blockIdx.z;
  constexpr int HC = 2 * KD + HD;
  const T* base = qkv + (long)b * N * qkv_ld + h * HC;
  for (int i = tid; i < ATT_QT * HD; i += 256) {
    const int qi = i / HD, e = i - qi * HD;
    sO[i] = (q0 + qi < N) ? ElemTraits<T>::to_f(d_o[((long)b * N + q0 + qi) * do_ld + h * HD + e]) : 0.f;
  }
  for (int j0 = 0; j0 < N; j0 += CH) {           // dP = dO v^T
    __syncthreads();
    for (int i = tid; i < CH * HD; i += 256) {
      const int j = i / HD, e = i - j * HD;
      sX[j * (HD + 1) + e] = (j0 + j < N) ? ElemTraits<T>::to_f(base[(long)(j0 + j) * qkv_ld + 2 * KD + e]) : 0.f;
    }
    __syncthreads();
    const int j = tid & 63, qg = tid >> 6;
    if (j0 + j < N) {
      float s[ATT_QT / 4];
#pragma unroll
      for (int qq = 0; qq < ATT_QT / 4; ++qq) s[qq] = 0.f;
      for (int e = 0; e < HD; ++e) {
        const float v = sX[j * (HD + 1) + e];
#pragma unroll
        for (int qq = 0; qq < ATT_QT / 4; ++qq) s[qq] += sO[(qg * (ATT_QT / 4) + qq) * HD + e] * v;
      }
#pragma unroll
      for (int qq = 0; qq < ATT_QT / 4; ++qq) sS[(qg * (ATT_QT / 4) + qq) * N + j0 + j] = s[qq];
    }
  }
  __syncthreads();
  {
    const int qi = tid >> 3, l = tid & 7;
    const bool live = q0 + qi < N;
    const float* prow = p + (((long)b * heads + h) * N + (live ? q0 + qi : 0)) * N;
    float dl = 0.f;
    if (live) for (int j = l; j < N; j += 8) dl += sS[qi * N + j] * prow[j];
#pragma unroll
    for (int o_ = 4; o_ >= 1; o_ >>= 1) dl += __shfl_xor(dl, o_);
    float* dsrow = ds + (((long)b * heads + h) * N + (q0 + qi)) * N;
    for (int j = l; j < N; j += 8) {
      const float v = live ? prow[j] * (sS[qi * N + j] - dl) : 0.f;
      sS[qi * N + j] = v;
      if (live) dsrow[j] = v;
    }
  }
  const int d = tid % KD, g = tid / KD;             // dq = scale * dS k
  float acc[QPG];
#pragma unroll
  for (int i = 0; i < QPG; ++i) acc[i] = 0.f;
  for (int j0 = 0; j0 < N; j0 += CH) {
    __syncthreads();
    for (int i = tid; i < CH * KD; i += 256) {
      const int j = i / KD, dd = i - j * KD;
      sX[j * KD + dd] = (j0 + j < N) ? ElemTraits<T>::to_f(base[(long)(j0 + j) * qkv_ld + KD + dd]) : 0.f;
    }
    __syncthreads();
    const int jn = min(CH, N - j0);
    for (int j = 0; j < jn; ++j) {
      const float v = sX[j * KD + d];
#pragma unroll
      for (int i = 0; i < QPG; ++i) acc[i] += sS[(g * QPG + i) * N + j0 + j] * v;
    }
  }
#pragma unroll
  for (int i = 0; i < QPG; ++i) {
    const int qi = g * QPG + i;
    if (q0 + qi < N) dqkv[((long)b * N + q0 + qi) * dqkv_ld + h * HC + d] = ElemTraits<T>::from_f(acc[i] * scale);
  }
}

template <typename T, int KD, int HD>
__global__ __launch_bounds__(256) void attention_bwd_kv_fast(int B, int N, int heads, const T* __restrict__ qkv, int qkv_ld,
                                                             const float* __restrict__ p, const float* __restrict__ ds,
                                                             const T* __restrict__ d_o, int do_ld, T* __restrict__ dqkv, int dqkv_ld, float scale) {
  constexpr int CH = 64, KT = 32;
  constexpr int GV = 256 / HD, KPV = KT / GV;       // dv: key groups / keys per thread
  constexpr int GK = 256 / KD, KPK = KT / GK;       // dk
  __shared__ float sP[CH][KT + 1], sD[CH][KT + 1], sO[CH][HD], sQ[CH][KD];
  const int tid = threadIdx.x;
  const int j0 = blockIdx.x * KT, h = blockIdx.y, b = blockIdx.z;
  constexpr int HC = 2 * KD + HD;
  const T* base = qkv + (long)b * N * qkv_ld + h * HC;
  const float* pb = p + ((long)b * heads + h) * N * N;
  const float* db = ds + ((long)b * heads + h) * N * N;
  const int e = tid % HD, gv = tid / HD, d = tid % KD, gk = tid / KD;
  float av[KPV], ak[KPK];
#pragma unroll
  for (int i = 0; i < KPV; ++i) av[i] = 0.f;
#pragma unroll
  for (int i = 0; i < KPK; ++i) ak[i] = 0.f;
  for (int i0 = 0; i0 < N; i0 += CH) {
    __syncthreads();
    for (int i = tid; i < CH * KT; i += 256) {
      const int qi = i / KT, j = i - qi * KT;
      const bool ok = (i0 + qi < N) && (j0 + j < N);
      sP[qi][j] = ok ? pb[(long)(i0 + qi) * N + j0 + j] : 0.f;
      sD[qi][j] = ok ? db[(long)(i0 + qi) * N + j0 + j] : 0.f;
    }
    for (int i = tid; i < CH * HD; i += 256) {
      const int qi = i / HD, ee = i - qi * HD;
      sO[qi][ee] = (i0 + qi < N) ? ElemTraits<T>::to_f(d_o[((long)b * N + i0 + qi) * do_ld + h * HD + ee]) : 0.f;
    }
    for (int i = tid; i < CH * KD; i += 256) {
      const int qi = i / KD, dd = i - qi * KD;
      sQ[qi][dd] = (i0 + qi < N) ? ElemTraits<T>::to_f(base[(long)(i0 + qi) * qkv_ld + dd]) : 0.f;
    }
    __syncthreads();
    for (int qi = 0; qi < CH; ++qi) {
      const float vo = sO[qi][e], vq = sQ[qi][d];
#pragma unroll
      for (int i = 0; i < KPV; ++i) av[i] += sP[qi][gv * KPV + i] * vo;
#pragma unroll
      for (int i = 0; i < KPK; ++i) ak[i] += sD[qi][gk * KPK + i] * vq;
    }
  }
#pragma unroll
  for (int i = 0; i < KPV; ++i) {
    const int j = j0 + gv * KPV + i;
    if (j < N) dqkv[((long)b * N + j) * dqkv_ld + h * HC + 2 * KD + e] = ElemTraits<T>::from_f(av[i]);
  }
#pragma unroll
  for (int i = 0; i < KPK; ++i) {
    const int j = j0 + gk * KPK + i;
    if (j < N) dqkv[((long)b * N + j) * dqkv_ld + h * HC + KD + d] = ElemTraits<T>::from_f(ak[i] * scale);
  }
}


// ================================================================================================ MFMA path
// f16 / bf16, KD = 32, HD = 64, N % 4 == 0.  One WAVE owns a 32-token tile and walks the other token axis in tiles of
// 32; a score tile lives in the 32x32 MFMA accumulator layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)),
// is soft-maxed in registers, and is fed straight back as the B operand of the next product: the reduction index of
// an MFMA may be enumerated in any order as long as A and B agree, so k-slot (half, i) of step m is bound to tile row
// 16m + 8*(i>>2) + 4*half + (i&3) — exactly the rows a lane already holds in regs 8m..8m+7.  The matching A operand
// (V^T, K^T, dO^T, Q^T: "8 rows of one column per lane") comes from row-major LDS images through the transposing
// ds_read_b64_tr_b16, two reads of 4 rows.  No score / dS matrix ever goes through LDS or the workspace.
typedef short as16x4 __attribute__((ext_vector_type(4)));
typedef short as16x8 __attribute__((ext_vector_type(8)));
template <typename T> struct AttMma;
template <> struct AttMma<_Float16> {
  static __device__ __forceinline__ f32x16 run(as16x8 a, as16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ short bits(float v) { return __builtin_bit_cast(short, (_Float16)v); }
};
template <> struct AttMma<__bf16> {
  static __device__ __forceinline__ f32x16 run(as16x8 a, as16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ short bits(float v) { return __builtin_bit_cast(short, (__bf16)v); }
};
template <> struct AttMma<float> {   // never instantiated on the MFMA path; keeps the dtype dispatch macro compiling
  static __device__ __forceinline__ f32x16 run(as16x8, as16x8, f32x16 c) { return c; }
  static __device__ __forceinline__ short bits(float) { return 0; }
};

__device__ __forceinline__ as16x8 att_tr_pair(const unsigned char* lo_addr, int hi_delta) {
  const as16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) as16x4*)lo_addr);
  const as16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) as16x4*)(lo_addr + hi_delta));
  return as16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// regs 8m..8m+7 of an accumulator tile -> B operand of MFMA step m
template <typename T>
__device__ __forceinline__ as16x8 att_pack(const float* v) {
  as16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = AttMma<T>::bits(v[i]);
  return r;
}
#define ATT_ZERO16(acc) _Pragma("unroll") for (int e_ = 0; e_ < 16; ++e_) (acc)[e_] = 0.f

// forward: wave = 32 queries.  S^T = K Q^T per key tile; pass 1 online (max, sum), pass 2 P -> global (f32) and
// O^T += V^T P^T with V staged [64 keys][64] (row stride 192 B) in a double-buffered LDS chunk.
template <typename T>
__global__ __launch_bounds__(256) void attention_fwd_mma(int B, int N, int heads, const T* __restrict__ qkv, int qkv_ld,
                                                         T* __restrict__ o, int o_ld, float* __restrict__ p, float scale) {
  constexpr int KD = 32, HD = 64, HC = 128, VROW = 192, VCH = 64 * VROW;
  __shared__ __attribute__((aligned(16))) unsigned char sV[2 * VCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const bool wave_live = q0 < N;
  const int qi = min(q0 + col, N - 1);
  const bool q_ok = q0 + col < N;
  const T* base = qkv + (long)b * N * qkv_ld + h * HC;
  const int ntile = (N + 31) / 32;
  as16x8 qf[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) qf[g] = *(const as16x8*)(base + (long)qi * qkv_ld + 8 * half + 16 * g);

  auto score_tile = [&](int jt, float* sc) {
    const int kj = min(jt * 32 + col, N - 1);
    const T* kp = base + (long)kj * qkv_ld + KD + 8 * half;
    const as16x8 k0 = *(const as16x8*)kp, k1 = *(const as16x8*)(kp + 16);
    f32x16 acc;
    ATT_ZERO16(acc);
    acc = AttMma<T>::run(k0, qf[0], acc);
    acc = AttMma<T>::run(k1, qf[1], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = acc[r] * scale;
  };
  // ---- pass 1: running max / sum per query (each lane sees half of the keys of its query column)
  float mrun = -3.0e38f, lrun = 0.f;
  if (wave_live)
    for (int jt = 0; jt < ntile; ++jt) {
      float sc[16];
      score_tile(jt, sc);
      float tmax = -3.0e38f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < N;
        tmax = ok ? fmaxf(tmax, sc[r]) : tmax;
      }
      const float mnew = fmaxf(mrun, tmax);
      float add = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool ok = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < N;
        add += ok ? __expf(sc[r] - mnew) : 0.f;
      }
      lrun = lrun * __expf(mrun - mnew) + add;
      mrun = mnew;
    }
  {
    const float m2 = __shfl_xor(mrun, 32), l2 = __shfl_xor(lrun, 32);
    const float mm = fmaxf(mrun, m2);
    lrun = lrun * __expf(mrun - mm) + l2 * __expf(m2 - mm);
    mrun = mm;
  }
  const float inv_l = lrun > 0.f ? 1.f / lrun : 0.f;

  // ---- pass 2
  f32x16 oacc[2];
  ATT_ZERO16(oacc[0]);
  ATT_ZERO16(oacc[1]);
  const int nchunk = (N + 63) / 64;
  const int st_row = tid >> 2, st_c = tid & 3;                // staging: thread -> (key row, 2 x 16-byte chunks)
  uint4 v0, v1;
  auto load_v = [&](int c) {
    const int kj = c * 64 + st_row;
    v0 = v1 = make_uint4(0, 0, 0, 0);
    if (kj < N) {
      const T* vp = base + (long)kj * qkv_ld + 2 * KD + st_c * 16;
      v0 = *(const uint4*)vp;
      v1 = *(const uint4*)(vp + 8);
    }
  };
  auto store_v = [&](int buf) {
    unsigned char* d = sV + buf * VCH + st_row * VROW + st_c * 32;
    *(uint4*)d = v0;
    *(uint4*)(d + 16) = v1;
  };
  load_v(0);
  store_v(0);
  __syncthreads();
  const int trq = (lane & 15) >> 2, trp = lane & 3, trg = (lane >> 4) & 1;
  const int tr_off = (4 * half + trq) * VROW + (trg * 16 + trp * 4) * 2;   // + (tile-in-chunk*32 + 16m) rows, + et*64 bytes
  float* prow = p + (((long)b * heads + h) * N + qi) * N;
  for (int c = 0; c < nchunk; ++c) {
    const bool more = c + 1 < nchunk;
    if (more) load_v(c + 1);
    const unsigned char* sv = sV + (c & 1) * VCH;
    if (wave_live) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) {
        const int jt = c * 2 + t2;
        if (jt < ntile) {
          float sc[16];
          score_tile(jt, sc);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = jt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half < N;
            sc[r] = ok ? __expf(sc[r] - mrun) * inv_l : 0.f;
          }
          if (q_ok) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int kj = jt * 32 + 8 * g + 4 * half;
              if (kj < N) *(f32x4*)(prow + kj) = f32x4{sc[4 * g], sc[4 * g + 1], sc[4 * g + 2], sc[4 * g + 3]};
            }
          }
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const as16x8 pb = att_pack<T>(sc + 8 * m);
            const unsigned char* a0 = sv + (t2 * 32 + 16 * m) * VROW + tr_off;
#pragma unroll
            for (int et = 0; et < 2; ++et) oacc[et] = AttMma<T>::run(att_tr_pair(a0 + et * 64, 8 * VROW), pb, oacc[et]);
          }
        }
      }
    }
    if (more) store_v((c + 1) & 1);
    __syncthreads();
  }
  if (wave_live && q_ok) {
    T* op = o + ((long)b * N + qi) * o_ld + h * HD;
#pragma unroll
    for (int et = 0; et < 2; ++et)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef T t4 __attribute__((ext_vector_type(4)));
        t4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ElemTraits<T>::from_f(oacc[et][4 * g + i]);
        *(t4*)(op + et * 32 + 8 * g + 4 * half) = v;
      }
  }
}

// backward A: wave = 32 queries.  dP^T = V dO^T per key tile (registers), delta_q = sum_j dP P (pass 1, also written
// to ws for kernel B), dS^T = P^T (dP^T - delta); dQ^T += K^T dS^T with all of K staged [N][32] (row stride 64 B).
// HAVE_O (r04): delta_q = sum_j dP_qj P_qj = dO_q . (sum_j P_qj V_j) = dO_q . O_q — with the forward output at hand pass 1 (every P row
// and every dP tile a second time) is one 64-element dot product per query.
template <typename T, bool HAVE_O = false>
__global__ __launch_bounds__(256) void attention_bwd_q_mma(int B, int N, int heads, const T* __restrict__ qkv, int qkv_ld,
                                                           const float* __restrict__ p, const T* __restrict__ d_o, int do_ld,
                                                           T* __restrict__ dqkv, int dqkv_ld, float* __restrict__ delta_ws, float scale,
                                                           const T* __restrict__ o_fwd, int o_ld) {
  constexpr int KD = 32, HD = 64, HC = 128, KROW = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char sK[];       // [ntile*32][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  const bool wave_live = q0 < N;
  const int qi = min(q0 + col, N - 1);
  const bool q_ok = q0 + col < N;
  const T* base = qkv + (long)b * N * qkv_ld + h * HC;
  const int ntile = (N + 31) / 32;
  for (int i = tid; i < ntile * 32 * 4; i += 256) {                       // 4 x 16-byte chunks per key row
    const int kj = i >> 2, ch = i & 3;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (kj < N) v = *(const uint4*)(base + (long)kj * qkv_ld + KD + ch * 8);
    *(uint4*)(sK + kj * KROW + ch * 16) = v;
  }
  as16x8 dof[4];
  {
    const T* dp = d_o + ((long)b * N + qi) * do_ld + h * HD + 8 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) dof[g] = *(const as16x8*)(dp + 16 * g);
  }
  const float* prow = p + (((long)b * heads + h) * N + qi) * N;
  auto dp_tile = [&](int jt, float* dpv, float* pv) {
    const int kj = min(jt * 32 + col, N - 1);
    const T* vp = base + (long)kj * qkv_ld + 2 * KD + 8 * half;
    f32x16 acc;
    ATT_ZERO16(acc);
#pragma unroll
    for (int g = 0; g < 4; ++g) acc = AttMma<T>::run(*(const as16x8*)(vp + 16 * g), dof[g], acc);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int kk = jt * 32 + 8 * g + 4 * half;
      f32x4 pp = f32x4{0.f, 0.f, 0.f, 0.f};
      if (kk < N) pp = *(const f32x4*)(prow + kk);                       // N % 4 == 0: a group of 4 keys is all-or-nothing
#pragma unroll
      for (int i = 0; i < 4; ++i) { pv[4 * g + i] = pp[i]; dpv[4 * g + i] = acc[4 * g + i]; }
    }
  };
  float delta = 0.f;
  if constexpr (HAVE_O && sizeof(T) == 2) {        // (the f32 instantiation only keeps the dtype dispatch compiling)
    const T* orow = o_fwd + ((long)b * N + qi) * o_ld + h * HD + 8 * half;      // the lane's half of the query's 64 outputs, as dof holds dO
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      typedef T t8 __attribute__((ext_vector_type(8)));
      const t8 ov = *(const t8*)(orow + 16 * g);
      const t8 dv = __builtin_bit_cast(t8, dof[g]);
#pragma unroll
      for (int i = 0; i < 8; ++i) delta += ElemTraits<T>::to_f(ov[i]) * ElemTraits<T>::to_f(dv[i]);
    }
  } else if (wave_live) {
    for (int jt = 0; jt < ntile; ++jt) {
      float dpv[16], pv[16];
      dp_tile(jt, dpv, pv);
#pragma unroll
      for (int r = 0; r < 16; ++r) delta += dpv[r] * pv[r];
    }
  }
  delta += __shfl_xor(delta, 32);
  if (wave_live && q_ok && half == 0) delta_ws[((long)b * heads + h) * N + qi] = delta;
  __syncthreads();                                                          // sK staged
  f32x16 qacc;
  ATT_ZERO16(qacc);
  const int trq = (lane & 15) >> 2, trp = lane & 3, trg = (lane >> 4) & 1;
  const int tr_off = (4 * half + trq) * KROW + (trg * 16 + trp * 4) * 2;
  if (wave_live)
    for (int jt = 0; jt < ntile; ++jt) {
      float dpv[16], pv[16];
      dp_tile(jt, dpv, pv);
#pragma unroll
      for (int r = 0; r < 16; ++r) dpv[r] = pv[r] * (dpv[r] - delta);
#pragma unroll
      for (int m = 0; m < 2; ++m)
        qacc = AttMma<T>::run(att_tr_pair(sK + (jt * 32 + 16 * m) * KROW + tr_off, 8 * KROW), att_pack<T>(dpv + 8 * m), qacc);
    }
  if (wave_live && q_ok) {
    T* dq = dqkv + ((long)b * N + qi) * dqkv_ld + h * HC;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      typedef T t4 __attribute__((ext_vector_type(4)));
      t4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = ElemTraits<T>::from_f(qacc[4 * g + i] * scale);
      *(t4*)(dq + 8 * g + 4 * half) = v;
    }
  }
}

// backward B: wave = 32 keys, loop over query tiles.  dP = dO V^T (registers; tile rows = queries, cols = keys),
// P read coalesced, dS = P (dP - delta_q);  dV^T += dO^T P,  dK^T += Q^T dS  with the dO / Q query tile staged in LDS.
template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_kv_mma(int B, int N, int heads, const T* __restrict__ qkv, int qkv_ld,
                                                            const float* __restrict__ p, const float* __restrict__ delta_ws,
                                                            const T* __restrict__ d_o, int do_ld, T* __restrict__ dqkv, int dqkv_ld, float scale) {
  constexpr int KD = 32, HD = 64, HC = 128, OROW = 192, QROW = 64, OB = 32 * OROW, QB = 32 * QROW;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm_kv[];    // [2][dO tile | Q tile] then delta[ntile*32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int j0 = (blockIdx.x * 4 + wave) * 32;
  const bool wave_live = j0 < N;
  const int kj = min(j0 + col, N - 1);
  const bool k_ok = j0 + col < N;
  const T* base = qkv + (long)b * N * qkv_ld + h * HC;
  const int ntile = (N + 31) / 32;
  float* sDelta = (float*)(sm_kv + 2 * (OB + QB));
  for (int i = tid; i < ntile * 32; i += 256) sDelta[i] = i < N ? delta_ws[((long)b * heads + h) * N + i] : 0.f;
  as16x8 vf[4];
  {
    const T* vp = base + (long)kj * qkv_ld + 2 * KD + 8 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) vf[g] = *(const as16x8*)(vp + 16 * g);
  }
  // staging of one query tile: dO 32 x 64 (256 chunks: one per thread), Q 32 x 32 (128 chunks: threads 0..127)
  const int so_row = tid >> 3, so_ch = tid & 7, sq_row = (tid & 127) >> 2, sq_ch = tid & 3;
  uint4 ro, rq;
  auto load_t = [&](int it) {
    ro = rq = make_uint4(0, 0, 0, 0);
    const int qo = it * 32 + so_row, qq = it * 32 + sq_row;
    if (qo < N) ro = *(const uint4*)(d_o + ((long)b * N + qo) * do_ld + h * HD + so_ch * 8);
    if (tid < 128 && qq < N) rq = *(const uint4*)(base + (long)qq * qkv_ld + sq_ch * 8);
  };
  auto store_t = [&](int buf) {
    unsigned char* d = sm_kv + buf * (OB + QB);
    *(uint4*)(d + so_row * OROW + so_ch * 16) = ro;
    if (tid < 128) *(uint4*)(d + OB + sq_row * QROW + sq_ch * 16) = rq;
  };
  load_t(0);
  store_t(0);
  __syncthreads();
  f32x16 vacc[2], kacc;
  ATT_ZERO16(vacc[0]);
  ATT_ZERO16(vacc[1]);
  ATT_ZERO16(kacc);
  const int trq = (lane & 15) >> 2, trp = lane & 3, trg = (lane >> 4) & 1;
  const int tro_off = (4 * half + trq) * OROW + (trg * 16 + trp * 4) * 2;
  const int trk_off = (4 * half + trq) * QROW + (trg * 16 + trp * 4) * 2;
  const float* pcol = p + ((long)b * heads + h) * N * N + kj;
  for (int it = 0; it < ntile; ++it) {
    const bool more = it + 1 < ntile;
    if (more) load_t(it + 1);
    const unsigned char* sd = sm_kv + (it & 1) * (OB + QB);
    if (wave_live) {
      // dP tile: A = dO rows (queries) straight from global, B = this wave's V rows
      const int qa = min(it * 32 + col, N - 1);
      const T* dop = d_o + ((long)b * N + qa) * do_ld + h * HD + 8 * half;
      f32x16 acc;
      ATT_ZERO16(acc);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc = AttMma<T>::run(*(const as16x8*)(dop + 16 * g), vf[g], acc);
      float pv[16], dsv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ql = (r & 3) + 8 * (r >> 2) + 4 * half, qg = it * 32 + ql;
        const bool ok = k_ok && qg < N;
        pv[r] = ok ? pcol[(long)qg * N] : 0.f;
        dsv[r] = pv[r] * (acc[r] - sDelta[it * 32 + ql]);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const as16x8 pb = att_pack<T>(pv + 8 * m), db = att_pack<T>(dsv + 8 * m);
        const unsigned char* ao = sd + 16 * m * OROW + tro_off;
#pragma unroll
        for (int et = 0; et < 2; ++et) vacc[et] = AttMma<T>::run(att_tr_pair(ao + et * 64, 8 * OROW), pb, vacc[et]);
        kacc = AttMma<T>::run(att_tr_pair(sd + OB + 16 * m * QROW + trk_off, 8 * QROW), db, kacc);
      }
    }
    if (more) store_t((it + 1) & 1);
    __syncthreads();
  }
  if (wave_live && k_ok) {
    T* dk = dqkv + ((long)b * N + kj) * dqkv_ld + h * HC + KD;
    T* dv = dk + KD;
    typedef T t4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      t4 v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = ElemTraits<T>::from_f(kacc[4 * g + i] * scale);
      *(t4*)(dk + 8 * g + 4 * half) = v;
#pragma unroll
      for (int et = 0; et < 2; ++et) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ElemTraits<T>::from_f(vacc[et][4 * g + i]);
        *(t4*)(dv + et * 32 + 8 * g + 4 * half) = v;
      }
    }
  }
}

static bool att_mma_ok(int dtype, int N, int kd, int hd, const void* a, int a_ld, const void* b2, int b_ld) {
  return dtype != SY11_F32 && kd == 32 && hd == 64 && N % 4 == 0 && N >= 32 && a_ld % 8 == 0 && b_ld % 8 == 0 &&
         (((uintptr_t)a | (uintptr_t)b2) & 15) == 0;
}

static int att_check(int dtype, int B, int N, int heads, int kd, int hd, const char* who, bool mma) {
  SY11_REQUIRE(dtype_ok(dtype) && B > 0 && N > 0 && heads > 0 && kd > 0 && hd > 0, "%s: bad dims", who);
  SY11_REQUIRE(kd <= 64 && hd <= 128 && 32 * (kd + hd) <= 16 * 256, "%s: kd<=64, hd<=128 supported", who);
  SY11_REQUIRE(heads <= 65535 && B <= 65535, "%s: heads/B exceed grid limits", who);
  (void)mma;                                     // every path takes any N: the VALU kernels keep their score rows in LDS while they fit, else in P / dS
  return SY11_OK;
}

extern "C" int sy11_attention_fwd(int32_t dtype, int32_t B, int32_t N, int32_t heads, int32_t kd, int32_t hd, const void* qkv,
                                  int32_t qkv_ld, void* o, int32_t o_ld, float* p, void* stream) {
  const bool use_mma = att_mma_ok(dtype, N, kd, hd, qkv, qkv_ld, o, o_ld) && ((uintptr_t)p & 15) == 0;
  int rc = att_check(dtype, B, N, heads, kd, hd, "attention_fwd", use_mma);
  if (rc) return rc;
  SY11_REQUIRE(qkv && o && p && qkv_ld >= heads * (2 * kd + hd) && o_ld >= heads * hd, "attention_fwd: bad pointer/stride");
  const size_t lds = (size_t)(ATT_QT * N + ATT_QT * kd + 64 * (kd + 1)) * 4;
  dim3 grid(cdiv(N, ATT_QT), heads, B), block(256);
  const float scale = 1.0f / sqrtf((float)kd);
  hipStream_t st = (hipStream_t)stream;
  if (use_mma) {
    dim3 gm(cdiv(cdiv(N, 32), 4), heads, B);
    SY11_DISPATCH_DTYPE(dtype, T, {
      hipLaunchKernelGGL((attention_fwd_mma<T>), gm, block, 0, st, B, N, heads, (const T*)qkv, qkv_ld, (T*)o, o_ld, p, scale);
    });
    SY11_LAUNCH_CHECK("attention_fwd");
    return SY11_OK;
  }
  if ((size_t)(ATT_QT * N + ATT_QT * 32 + 64 * 64) * 4 > ATT_LDS_MAX || lds > ATT_LDS_MAX) {     // score rows beyond LDS: keep them in the workgroup's rows of P
    const size_t lg = (size_t)(ATT_QT * kd + 64 * (kd + 1)) * 4;
    SY11_DISPATCH_DTYPE(dtype, T, {
      hipLaunchKernelGGL((attention_fwd_kernel<T, true>), grid, block, lg, st, B, N, heads, kd, hd, (const T*)qkv, qkv_ld, (T*)o, o_ld, p, scale);
    });
    SY11_LAUNCH_CHECK("attention_fwd");
    return SY11_OK;
  }
  if (kd == 32 && hd == 64) {
    const size_t l2 = (size_t)(ATT_QT * N + ATT_QT * 32 + 64 * 64) * 4;
    SY11_DISPATCH_DTYPE(dtype, T, {
      if (l2 > 64 * 1024) (void)hipFuncSetAttribute((const void*)attention_fwd_fast<T, 32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
      hipLaunchKernelGGL((attention_fwd_fast<T, 32, 64>), grid, block, l2, st, B, N, heads, (const T*)qkv, qkv_ld, (T*)o, o_ld, p, scale);
    });
    SY11_LAUNCH_CHECK("attention_fwd");
    return SY11_OK;
  }
  SY11_DISPATCH_DTYPE(dtype, T, {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attention_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attention_fwd_kernel<T>), grid, block, lds, st, B, N, heads, kd, hd, (const T*)qkv, qkv_ld, (T*)o, o_ld, p, scale);
  });
  SY11_LAUNCH_CHECK("attention_fwd");
  return SY11_OK;
}

extern "C" size_t sy11_attention_workspace_bytes(int32_t B, int32_t N, int32_t heads) {
  return (B <= 0 || N <= 0 || heads <= 0) ? 0 : (size_t)B * heads * N * N * sizeof(float);
}

extern "C" int sy11_attention_bwd_o(int32_t dtype, int32_t B, int32_t N, int32_t heads, int32_t kd, int32_t hd, const void* qkv,
                                    int32_t qkv_ld, const float* p, const void* o, int32_t o_ld, const void* d_o, int32_t do_ld, void* dqkv,
                                    int32_t dqkv_ld, float* workspace, void* stream);
extern "C" int sy11_attention_bwd(int32_t dtype, int32_t B, int32_t N, int32_t heads, int32_t kd, int32_t hd, const void* qkv,
                                  int32_t qkv_ld, const float* p, const void* d_o, int32_t do_ld, void* dqkv, int32_t dqkv_ld,
                                  float* workspace, void* stream) {
  return sy11_attention_bwd_o(dtype, B, N, heads, kd, hd, qkv, qkv_ld, p, nullptr, 0, d_o, do_ld, dqkv, dqkv_ld, workspace, stream);
}

extern "C" int sy11_attention_bwd_o(int32_t dtype, int32_t B, int32_t N, int32_t heads, int32_t kd, int32_t hd, const void* qkv,
                                    int32_t qkv_ld, const float* p, const void* o, int32_t o_ld, const void* d_o, int32_t do_ld, void* dqkv,
                                    int32_t dqkv_ld, float* workspace, void* stream) {
  SY11_REQUIRE(!o || o_ld >= heads * hd, "attention_bwd: bad stride of the forward output");
  const bool use_mma = att_mma_ok(dtype, N, kd, hd, qkv, qkv_ld, d_o, do_ld) && dqkv_ld % 8 == 0 && (((uintptr_t)dqkv | (uintptr_t)p) & 15) == 0 &&
                       (size_t)cdiv(N, 32) * 32 * 64 <= 150 * 1024;
  int rc = att_check(dtype, B, N, heads, kd, hd, "attention_bwd", use_mma);
  if (rc) return rc;
  SY11_REQUIRE(qkv && p && d_o && dqkv && workspace, "attention_bwd: null pointer");
  SY11_REQUIRE(qkv_ld >= heads * (2 * kd + hd) && dqkv_ld >= heads * (2 * kd + hd) && do_ld >= heads * hd, "attention_bwd: bad stride");
  const size_t lds = (size_t)(ATT_QT * N + ATT_QT * hd) * 4;
  const float scale = 1.0f / sqrtf((float)kd);
  hipStream_t st = (hipStream_t)stream;
  dim3 gq(cdiv(N, ATT_QT), heads, B), gk(cdiv(N, 32), heads, B), block(256);
  if (use_mma) {
    const int ntile = cdiv(N, 32);
    dim3 gm(cdiv(ntile, 4), heads, B);
    const size_t lq = (size_t)ntile * 32 * 64, lk = (size_t)2 * (32 * 192 + 32 * 64) + (size_t)ntile * 32 * 4;
    SY11_DISPATCH_DTYPE(dtype, T, {
      if (lq > 64 * 1024) (void)hipFuncSetAttribute((const void*)attention_bwd_q_mma<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lq);
      if (lk > 64 * 1024) (void)hipFuncSetAttribute((const void*)attention_bwd_kv_mma<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lk);
      const bool have_o = o != nullptr && o_ld % 8 == 0 && ((uintptr_t)o & 15) == 0;
      if (have_o) {
        if (lq > 64 * 1024) (void)hipFuncSetAttribute((const void*)attention_bwd_q_mma<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lq);
        hipLaunchKernelGGL((attention_bwd_q_mma<T, true>), gm, block, lq, st, B, N, heads, (const T*)qkv, qkv_ld, p, (const T*)d_o, do_ld, (T*)dqkv, dqkv_ld, workspace, scale, (const T*)o, o_ld);
      } else
        hipLaunchKernelGGL((attention_bwd_q_mma<T, false>), gm, block, lq, st, B, N, heads, (const T*)qkv, qkv_ld, p, (const T*)d_o, do_ld, (T*)dqkv, dqkv_ld, workspace, scale, (const T*)nullptr, 0);
      hipLaunchKernelGGL((attention_bwd_kv_mma<T>), gm, block, lk, st, B, N, heads, (const T*)qkv, qkv_ld, p, (const float*)workspace, (const T*)d_o, do_ld, (T*)dqkv, dqkv_ld, scale);
    });
    SY11_LAUNCH_CHECK("attention_bwd");
    return SY11_OK;
  }
  if ((size_t)(ATT_QT * N + ATT_QT * 64 + 64 * 65) * 4 > ATT_LDS_MAX || lds > ATT_LDS_MAX) {     // dP / dS rows beyond LDS: in the dS workspace
    const size_t lg = (size_t)(ATT_QT * hd) * 4;
    SY11_DISPATCH_DTYPE(dtype, T, {
      hipLaunchKernelGGL((attention_bwd_q_kernel<T, true>), gq, block, lg, st, B, N, heads, kd, hd, (const T*)qkv, qkv_ld, p, (const T*)d_o, do_ld, (T*)dqkv, dqkv_ld, workspace, scale);
      hipLaunchKernelGGL((attention_bwd_kv_kernel<T>), gk, block, 0, st, B, N, heads, kd, hd, (const T*)qkv, qkv_ld, p, (const float*)workspace, (const T*)d_o, do_ld, (T*)dqkv, dqkv_ld, scale);
    });
    SY11_LAUNCH_CHECK("attention_bwd");
    return SY11_OK;
  }
  if (kd == 32 && hd == 64) {
    const size_t l2 = (size_t)(ATT_QT * N + ATT_QT * 64 + 64 * 65) * 4;
    SY11_DISPATCH_DTYPE(dtype, T, {
      if (l2 > 64 * 1024) (void)hipFuncSetAttribute((const void*)attention_bwd_q_fast<T, 32, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
      hipLaunchKernelGGL((attention_bwd_q_fast<T, 32, 64>), gq, block, l2, st, B, N, heads, (const T*)qkv, qkv_ld, p, (const T*)d_o, do_ld, (T*)dqkv, dqkv_ld, workspace, scale);
      hipLaunchKernelGGL((attention_bwd_kv_fast<T, 32, 64>), gk, block, 0, st, B, N, heads, (const T*)qkv, qkv_ld, p, (const float*)workspace, (const T*)d_o, do_ld, (T*)dqkv, dqkv_ld, scale);
    });
    SY11_LAUNCH_CHECK("attention_bwd");
    return SY11_OK;
  }
  SY11_DISPATCH_DTYPE(dtype, T, {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)attention_bwd_q_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((attention_bwd_q_kernel<T>), gq, block, lds, st, B, N, heads, kd, hd, (const T*)qkv, qkv_ld, p, (const T*)d_o, do_ld, (T*)dqkv, dqkv_ld, workspace, scale);
    hipLaunchKernelGGL((attention_bwd_kv_kernel<T>), gk, block, 0, st, B, N, heads, kd, hd, (const T*)qkv, qkv_ld, p, (const float*)workspace, (const T*)d_o, do_ld, (T*)dqkv, dqkv_ld, scale);
  });
  SY11_LAUNCH_CHECK("attention_bwd");
  return SY11_OK;
}
