// attention.hip — C2PSA attention core on gfx950 (N = H*W tokens, small: 400 at 640x640 input).
//
// qkv: NHWC pixels (B, N, heads*(2*kd+hd)); per head h the channels are [q(kd) | k(kd) | v(hd)]
// (the reference's qkv.view(B, heads, 2*kd+hd, N).split(...), nn/modules/block.py:1925-1927).
//   P[b,h,i,j] = softmax_j( scale * sum_d q[i,d] k[j,d] ),   o[b,i,h*hd+e] = sum_j P[i,j] v[j,e]
// 0.6 % of the network FLOPs: kept on the VALU with LDS-resident score rows (QT = 32 queries x N keys, f32);
// P is written out for the backward pass exactly as autograd keeps the softmax output.
#include "common.h"

#define ATT_QT 32

template <typename T>
__global__ __launch_bounds__(256) void attention_fwd_kernel(int B, int N, int heads, int kd, int hd, const T* __restrict__ qkv, int qkv_ld,
                                                            T* __restrict__ o, int o_ld, float* __restrict__ p, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sS = sm;                       // [QT][N]
  float* sQ = sS + ATT_QT * N;          // [QT][kd]
  float* sK = sQ + ATT_QT * kd;         // [64][kd+1]
  const int tid = threadIdx.x;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * ATT_QT;
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
    float mx = -INFINITY;
    for (int j = l; j < N; j += 8) mx = fmaxf(mx, sS[qi * N + j]);
#pragma unroll
    for (int o_ = 4; o_ >= 1; o_ >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o_));
    float sum = 0.f;
    for (int j = l; j < N; j += 8) {
      const float e = __expf(sS[qi * N + j] - mx);
      sS[qi * N + j] = e;
      sum += e;
    }
#pragma unroll
    for (int o_ = 4; o_ >= 1; o_ >>= 1) sum += __shfl_xor(sum, o_);
    const float inv = 1.f / sum;
    const bool live = q0 + qi < N;
    float* prow = p + (((long)b * heads + h) * N + (q0 + qi)) * N;
    for (int j = l; j < N; j += 8) {
      const float v = sS[qi * N + j] * inv;
      sS[qi * N + j] = v;
      if (live) prow[j] = v;
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
template <typename T>
__global__ __launch_bounds__(256) void attention_bwd_q_kernel(int B, int N, int heads, int kd, int hd, const T* __restrict__ qkv, int qkv_ld,
                                                              const float* __restrict__ p, const T* __restrict__ d_o, int do_ld,
                                                              T* __restrict__ dqkv, int dqkv_ld, float* __restrict__ ds, float scale) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sS = sm;                        // [QT][N] dP then dS
  float* sO = sS + ATT_QT * N;           // [QT][hd] dO tile
  const int tid = threadIdx.x;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int q0 = qt * ATT_QT;
  const int hc = 2 * kd + hd;
  const T* base = qkv + (long)b * N * qkv_ld + h * hc;
  for (int i = tid; i < ATT_QT * hd; i += 256) {
    const int qi = i / hd, e = i - qi * hd;
    sO[i] = (q0 + qi < N) ? ElemTraits<T>::to_f(d_o[((long)b * N + q0 + qi) * do_ld + h * hd + e]) : 0.f;
  }
  __syncthreads();
  for (int idx = tid; idx < ATT_QT * N; idx += 256) {
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
      const float v = live ? prow[j] * (sS[qi * N + j] - dl) : 0.f;
      sS[qi * N + j] = v;
      if (live) dsrow[j] = v;
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
    float mx = -INFINITY;
    for (int j = l; j < N; j += 8) mx = fmaxf(mx, sS[qi * N + j]);
#pragma unroll
    for (int o_ = 4; o_ >= 1; o_ >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o_));
    float sum = 0.f;
    for (int j = l; j < N; j += 8) {
      const float e = __expf(sS[qi * N + j] - mx);
      sS[qi * N + j] = e;
      sum += e;
    }
#pragma unroll
    for (int o_ = 4; o_ >= 1; o_ >>= 1) sum += __shfl_xor(sum, o_);
    const float inv = 1.f / sum;
    const bool live = q0 + qi < N;
    float* prow = p + (((long)b * heads + h) * N + (q0 + qi)) * N;
    for (int j = l; j < N; j += 8) {
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

static int att_check(int dtype, int B, int N, int heads, int kd, int hd, const char* who) {
  SY11_REQUIRE(dtype_ok(dtype) && B > 0 && N > 0 && heads > 0 && kd > 0 && hd > 0, "%s: bad dims", who);
  SY11_REQUIRE(kd <= 64 && hd <= 128 && 32 * (kd + hd) <= 16 * 256, "%s: kd<=64, hd<=128 supported", who);
  SY11_REQUIRE(heads <= 65535 && B <= 65535, "%s: heads/B exceed grid limits", who);
  const size_t lds = (size_t)(ATT_QT * N + ATT_QT * (kd > hd ? kd : hd) + 64 * ((kd > hd ? kd : hd) + 1)) * 4;
  SY11_REQUIRE(lds <= 160 * 1024, "%s: N=%d needs %zu bytes of LDS (>160 KiB)", who, N, lds);
  return SY11_OK;
}

extern "C" int sy11_attention_fwd(int32_t dtype, int32_t B, int32_t N, int32_t heads, int32_t kd, int32_t hd, const void* qkv,
                                  int32_t qkv_ld, void* o, int32_t o_ld, float* p, void* stream) {
  int rc = att_check(dtype, B, N, heads, kd, hd, "attention_fwd");
  if (rc) return rc;
  SY11_REQUIRE(qkv && o && p && qkv_ld >= heads * (2 * kd + hd) && o_ld >= heads * hd, "attention_fwd: bad pointer/stride");
  const size_t lds = (size_t)(ATT_QT * N + ATT_QT * kd + 64 * (kd + 1)) * 4;
  dim3 grid(cdiv(N, ATT_QT), heads, B), block(256);
  const float scale = 1.0f / sqrtf((float)kd);
  hipStream_t st = (hipStream_t)stream;
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

extern "C" int sy11_attention_bwd(int32_t dtype, int32_t B, int32_t N, int32_t heads, int32_t kd, int32_t hd, const void* qkv,
                                  int32_t qkv_ld, const float* p, const void* d_o, int32_t do_ld, void* dqkv, int32_t dqkv_ld,
                                  float* workspace, void* stream) {
  int rc = att_check(dtype, B, N, heads, kd, hd, "attention_bwd");
  if (rc) return rc;
  SY11_REQUIRE(qkv && p && d_o && dqkv && workspace, "attention_bwd: null pointer");
  SY11_REQUIRE(qkv_ld >= heads * (2 * kd + hd) && dqkv_ld >= heads * (2 * kd + hd) && do_ld >= heads * hd, "attention_bwd: bad stride");
  const size_t lds = (size_t)(ATT_QT * N + ATT_QT * hd) * 4;
  const float scale = 1.0f / sqrtf((float)kd);
  hipStream_t st = (hipStream_t)stream;
  dim3 gq(cdiv(N, ATT_QT), heads, B), gk(cdiv(N, 32), heads, B), block(256);
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
