// fusion.hip — Fusion('ESChannel') of the fusion model variant on gfx950 (HBM-bound elementwise / reduction kernels).
//
// Reference (ultralytics/nn/modules/conv.py): Fusion.forward ESChannel branch :2087-2127,
// WeightedSpatialAttention :1839-1852, GCT :2284-2301.  With n inputs x_i (B, C, H, W):
//     a   = cat(x_i, dim=1)                                                   (n*C channels)
//     GCT : s[b,c] = sum_hw a^2 ; e = sqrt(s + eps) * alpha ; nrm = gamma / sqrt(mean_c(e^2) + eps)
//           gate[b,c] = 1 + tanh(e * nrm + beta)
//     SAB : S_i[b,h,w] = sigmoid(conv3x3_{2->1, pad 1, no bias}([mean_c x_i, max_c x_i]))
//     out = sum_i ( chunk_i(a * gate) + x_i * S_i ) = sum_i x_i * (gate[b, i*C + c] + S_i[b,h,w])
// so the forward is: one statistics pass per input (channel mean / max / argmax per pixel + per-(b,c) sum of squares),
// two tiny kernels (3x3 map + sigmoid, the gate vector), and one combine pass; the backward mirrors it.
// Layout: NHWC views (pointer + pixel stride), a group of LP = C / VEC lanes owns one pixel (VEC = 16 bytes of channels).
#include "common.h"
#include "det.h"

template <typename T, int VEC>
__device__ __forceinline__ void fload(const T* p, float* f) {
  typedef T vt __attribute__((ext_vector_type(VEC)));
  const vt v = *(const vt*)p;
#pragma unroll
  for (int i = 0; i < VEC; ++i) f[i] = ElemTraits<T>::to_f(v[i]);
}
template <typename T, int VEC>
__device__ __forceinline__ void fstore(T* p, const float* f) {
  typedef T vt __attribute__((ext_vector_type(VEC)));
  vt v;
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = ElemTraits<T>::from_f(f[i]);
  *(vt*)p = v;
}

static int fusion_geom(int dtype, int C, const char* who, int* lp) {
  const int vec = 16 / dtype_size(dtype);
  SY11_REQUIRE(C > 0 && C % vec == 0, "%s: C=%d must be a multiple of %d", who, C, vec);
  const int l = C / vec;
  SY11_REQUIRE(l <= 64 && (l & (l - 1)) == 0, "%s: C/%d = %d must be a power of two <= 64", who, vec, l);
  *lp = l;
  return SY11_OK;
}
static bool fusion_view_ok(const void* p, int ld, int dtype) {
  const int vec = 16 / dtype_size(dtype);
  return p && ((uintptr_t)p & 15) == 0 && ld % vec == 0;
}

// ------------------------------------------------------------------------------------------------ pass 1: statistics
template <typename T, int VEC>
__global__ __launch_bounds__(256) void fusion_stats_kernel(int HW, int C, int LP, const T* __restrict__ x, int x_ld, float* __restrict__ mm,
                                                           unsigned short* __restrict__ amax, float* __restrict__ sq, int sq_ld) {
  extern __shared__ float red[];                       // [C] sum of squares of this block, then a [256][VEC] parking area
  const int b = blockIdx.y, tid = threadIdx.x;
  const int cl = tid % LP, sub = tid / LP, ppb = 256 / LP;
  float sacc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) sacc[i] = 0.f;
  const float invC = 1.0f / (float)C;
  for (int p = blockIdx.x * ppb + sub; p < HW; p += gridDim.x * ppb) {
    const long pix = (long)b * HW + p;
    float v[VEC];
    fload<T, VEC>(x + pix * x_ld + cl * VEC, v);
    float s = 0.f, mx = v[0];
    int mi = cl * VEC;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      s += v[i];
      sacc[i] += v[i] * v[i];
      if (v[i] > mx) { mx = v[i]; mi = cl * VEC + i; }   // strict: first maximum wins (torch.max tie rule on CPU)
    }
    for (int o = LP >> 1; o >= 1; o >>= 1) {
      s += __shfl_xor(s, o);
      const float omx = __shfl_xor(mx, o);
      const int omi = __shfl_xor(mi, o);
      if (omx > mx || (omx == mx && omi < mi)) { mx = omx; mi = omi; }
    }
    if (cl == 0) {
      mm[pix * 2] = s * invC;
      mm[pix * 2 + 1] = mx;
      amax[pix] = (unsigned short)mi;
    }
  }
  // every thread parks its VEC partial sums; thread c then adds the ppb rows of channel c in row order (no LDS atomics)
  float* park = red + C;
#pragma unroll
  for (int i = 0; i < VEC; ++i) park[tid * VEC + i] = sacc[i];
  __syncthreads();
  for (int i = tid; i < C; i += 256) {
    const int cv = i / VEC, j = i - cv * VEC;
    float t = 0.f;
    for (int k = 0; k < ppb; ++k) t += park[(k * LP + cv) * VEC + j];
    atomicAdd(sq + (long)b * sq_ld + i, t);
  }
}

extern "C" int sy11_fusion_stats(int32_t dtype, int32_t B, int32_t HW, int32_t C, const void* x, int32_t x_ld, float* mm,
                                 uint16_t* amax, float* sq, int32_t sq_ld, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && B > 0 && HW > 0 && mm && amax && sq && sq_ld >= C && x_ld >= C, "fusion_stats: bad argument");
  SY11_REQUIRE(C <= 65535, "fusion_stats: C exceeds the uint16 argmax");
  int lp, rc;
  if ((rc = fusion_geom(dtype, C, "fusion_stats", &lp))) return rc;
  SY11_REQUIRE(fusion_view_ok(x, x_ld, dtype), "fusion_stats: x view not 16-byte addressable");
  const int ppb = 256 / lp;
  int gx = cdiv(HW, ppb * 8);
  if (gx > 512) gx = 512;
  if (sy11_det(32)) gx = 1;                              // ordered mode: one workgroup per image owns sq[b][:] (det.h)
  dim3 grid(gx, B), block(256);
  SY11_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((fusion_stats_kernel<T, 16 / (int)sizeof(T)>), grid, block, (C + 256 * (16 / (int)sizeof(T))) * sizeof(float), (hipStream_t)stream, HW,
                                                   C, lp, (const T*)x, x_ld, mm, amax, sq, sq_ld));
  SY11_LAUNCH_CHECK("fusion_stats");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ SAB 3x3 map
// S = sigmoid( sum_{r,s,ch} w[r][s][ch] * mm[(y+r-1, x+s-1)][ch] ),  w in [KH][KW][I] memory (the KRSC filter of cv1)
__global__ __launch_bounds__(256) void sab_map_fwd_kernel(int B, int H, int W, const float* __restrict__ mm, const float* __restrict__ w,
                                                          float* __restrict__ S) {
  const long n = (long)B * H * W;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= n) return;
  const int xw = (int)(pix % W), yh = (int)((pix / W) % H);
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int yy = yh + r - 1, xx = xw + s - 1;
      if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
        const long q = pix + (long)(r - 1) * W + (s - 1);
        acc += w[(r * 3 + s) * 2] * mm[q * 2] + w[(r * 3 + s) * 2 + 1] * mm[q * 2 + 1];
      }
    }
  S[pix] = 1.0f / (1.0f + __expf(-acc));
}

extern "C" int sy11_sab_map_fwd(int32_t B, int32_t H, int32_t W, const float* mm, const float* w, float* S, void* stream) {
  SY11_REQUIRE(B > 0 && H > 0 && W > 0 && mm && w && S, "sab_map_fwd: bad argument");
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(sab_map_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, H, W, mm, w, S);
  SY11_LAUNCH_CHECK("sab_map_fwd");
  return SY11_OK;
}

// dpre = dS * S * (1 - S);  dmm[p][ch] = sum_taps dpre[p - tap] * w[tap][ch];  dw[tap][ch] += sum_p dpre[p] * mm[p + tap][ch]
__global__ __launch_bounds__(256) void sab_map_bwd_kernel(int B, int H, int W, const float* __restrict__ dS, const float* __restrict__ S,
                                                          const float* __restrict__ mm, const float* __restrict__ w, float* __restrict__ dmm,
                                                          float* dw, long part_stride) {
  __shared__ float red[4 * 18];
  const long n = (long)B * H * W;
  const long pix = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = pix < n;
  float g0 = 0.f, g1 = 0.f, part[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) part[i] = 0.f;
  if (live) {
    const int xw = (int)(pix % W), yh = (int)((pix / W) % H);
    const float sp = S[pix], dpre = dS[pix] * sp * (1.f - sp);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int t = r * 3 + s;
        // forward tap (r,s) of output pixel q = p - (r-1, s-1) read input pixel p
        const int qy = yh - (r - 1), qx = xw - (s - 1);
        if ((unsigned)qy < (unsigned)H && (unsigned)qx < (unsigned)W) {
          const long q = pix - (long)(r - 1) * W - (s - 1);
          const float sq_ = S[q], dq = dS[q] * sq_ * (1.f - sq_);
          g0 += dq * w[t * 2];
          g1 += dq * w[t * 2 + 1];
        }
        const int iy = yh + r - 1, ix = xw + s - 1;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
          const long q = pix + (long)(r - 1) * W + (s - 1);
          part[t * 2] = dpre * mm[q * 2];
          part[t * 2 + 1] = dpre * mm[q * 2 + 1];
        }
      }
    dmm[pix * 2] = g0;
    dmm[pix * 2 + 1] = g1;
  }
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    float v = part[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[(threadIdx.x >> 6) * 18 + i] = v;       // [wave][18], folded in wave order below
  }
  __syncthreads();
  if (threadIdx.x < 18)
    atomicAdd(dw + (long)blockIdx.x * part_stride + threadIdx.x,
              ((red[threadIdx.x] + red[18 + threadIdx.x]) + red[36 + threadIdx.x]) + red[54 + threadIdx.x]);
}

extern "C" int sy11_sab_map_bwd(int32_t B, int32_t H, int32_t W, const float* dS, const float* S, const float* mm, const float* w,
                                float* dmm, float* dw, void* stream) {
  SY11_REQUIRE(B > 0 && H > 0 && W > 0 && dS && S && mm && w && dmm && dw, "sab_map_bwd: bad argument");
  const long n = (long)B * H * W;
  const long nb = (n + 255) / 256;
  DetPartials dp;                                      // ordered mode (det.h): one partial dw[18] row per workgroup
  float* dwk = dw;
  long pstride = 0;
  if (sy11_det(32) && nb > 1) {
    if (!dp.acquire((hipStream_t)stream, 1, nb, 18)) SY11_FAIL(SY11_ELAUNCH, "sab_map_bwd: ordered-reduction workspace unavailable");
    dwk = dp.buf(0); pstride = 18;
  }
  hipLaunchKernelGGL(sab_map_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, B, H, W, dS, S, mm, w, dmm, dwk, pstride);
  SY11_LAUNCH_CHECK("sab_map_bwd");
  return dp.base ? dp.fold(0, dw) : SY11_OK;
}

// ------------------------------------------------------------------------------------------------ GCT gate vector
__device__ __forceinline__ float block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

__global__ __launch_bounds__(256) void gct_gate_fwd_kernel(int Ct, const float* __restrict__ sq, const float* __restrict__ alpha,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float* __restrict__ G) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  float part = 0.f;
  for (int c = threadIdx.x; c < Ct; c += 256) {
    const float e = sqrtf(sq[(long)b * Ct + c] + eps) * alpha[c];
    part += e * e;
  }
  const float m = block_sum(part, sh) / (float)Ct;
  const float r = 1.0f / sqrtf(m + eps);
  for (int c = threadIdx.x; c < Ct; c += 256) {
    const float e = sqrtf(sq[(long)b * Ct + c] + eps) * alpha[c];
    G[(long)b * Ct + c] = 1.0f + tanhf(e * gamma[c] * r + beta[c]);
  }
}

extern "C" int sy11_gct_gate_fwd(int32_t B, int32_t Ct, const float* sq, const float* alpha, const float* gamma, const float* beta,
                                 float eps, float* G, void* stream) {
  SY11_REQUIRE(B > 0 && Ct > 0 && sq && alpha && gamma && beta && G, "gct_gate_fwd: bad argument");
  hipLaunchKernelGGL(gct_gate_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, Ct, sq, alpha, gamma, beta, eps, G);
  SY11_LAUNCH_CHECK("gct_gate_fwd");
  return SY11_OK;
}

// given dG = d loss / d gate: q[b][c] (dx += x * q) and the parameter gradients (accumulated over the batch)
__global__ __launch_bounds__(256) void gct_gate_bwd_kernel(int Ct, const float* __restrict__ sq, const float* __restrict__ alpha,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           const float* __restrict__ dG, float* __restrict__ q, float* dalpha, float* dgamma,
                                                           float* dbeta, long part_stride) {
  __shared__ float sh[4];
  const int b = blockIdx.x;
  const long po = (long)b * part_stride;                 // ordered mode: image b adds into its own partial row
  float part = 0.f;
  for (int c = threadIdx.x; c < Ct; c += 256) {
    const float e = sqrtf(sq[(long)b * Ct + c] + eps) * alpha[c];
    part += e * e;
  }
  const float m = block_sum(part, sh) / (float)Ct;
  const float r = 1.0f / sqrtf(m + eps);
  float dr_part = 0.f;
  for (int c = threadIdx.x; c < Ct; c += 256) {
    const float e = sqrtf(sq[(long)b * Ct + c] + eps) * alpha[c];
    const float t = tanhf(e * gamma[c] * r + beta[c]);
    const float dz = dG[(long)b * Ct + c] * (1.f - t * t);
    dr_part += dz * e * gamma[c];                       // dn_c = dz * e,  n_c = gamma_c * r
  }
  const float dr = block_sum(dr_part, sh);
  const float dm = -0.5f * dr * r * r * r;              // r = (m + eps)^-1/2
  for (int c = threadIdx.x; c < Ct; c += 256) {
    const float root = sqrtf(sq[(long)b * Ct + c] + eps);
    const float e = root * alpha[c];
    const float t = tanhf(e * gamma[c] * r + beta[c]);
    const float dz = dG[(long)b * Ct + c] * (1.f - t * t);
    const float de = dz * gamma[c] * r + dm * 2.f * e / (float)Ct;
    q[(long)b * Ct + c] = de * alpha[c] / root;          // ds = de * alpha / (2 root),  dx = 2 x ds
    atomicAdd(dbeta + po + c, dz);
    atomicAdd(dgamma + po + c, dz * e * r);
    atomicAdd(dalpha + po + c, de * root);
  }
}

extern "C" int sy11_gct_gate_bwd(int32_t B, int32_t Ct, const float* sq, const float* alpha, const float* gamma, const float* beta,
                                 float eps, const float* dG, float* q, float* dalpha, float* dgamma, float* dbeta, void* stream) {
  SY11_REQUIRE(B > 0 && Ct > 0 && sq && alpha && gamma && beta && dG && q && dalpha && dgamma && dbeta, "gct_gate_bwd: bad argument");
  DetPartials dp;                                      // ordered mode (det.h): one partial row per image for each of the three gradients
  float *pa = dalpha, *pg = dgamma, *pb = dbeta;
  long pstride = 0;
  if (sy11_det(32) && B > 1) {
    if (!dp.acquire((hipStream_t)stream, 3, B, Ct)) SY11_FAIL(SY11_ELAUNCH, "gct_gate_bwd: ordered-reduction workspace unavailable");
    pa = dp.buf(0); pg = dp.buf(1); pb = dp.buf(2); pstride = Ct;
  }
  hipLaunchKernelGGL(gct_gate_bwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, Ct, sq, alpha, gamma, beta, eps, dG, q, pa, pg, pb, pstride);
  SY11_LAUNCH_CHECK("gct_gate_bwd");
  if (dp.base) {
    int rc = dp.fold(0, dalpha);
    if (!rc) rc = dp.fold(1, dgamma);
    return rc ? rc : dp.fold(2, dbeta);
  }
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ combine
struct FusionIn {
  const void* x[3];
  const float* S[3];
  int ld[3];
  int n;
};

template <typename T, int VEC>
__global__ __launch_bounds__(256) void fusion_combine_kernel(int HW, int C, int LP, FusionIn in, const float* __restrict__ G, T* __restrict__ out,
                                                             int out_ld) {
  const int b = blockIdx.y, tid = threadIdx.x;
  const int cl = tid % LP, sub = tid / LP, ppb = 256 / LP;
  float g[3][VEC];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < VEC; ++j) g[i][j] = i < in.n ? G[(long)b * in.n * C + i * C + cl * VEC + j] : 0.f;
  for (int p = blockIdx.x * ppb + sub; p < HW; p += gridDim.x * ppb) {
    const long pix = (long)b * HW + p;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < in.n) {
        float v[VEC];
        fload<T, VEC>((const T*)in.x[i] + pix * in.ld[i] + cl * VEC, v);
        const float s = in.S[i][pix];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] += v[j] * (g[i][j] + s);
      }
    fstore<T, VEC>(out + pix * out_ld + cl * VEC, acc);
  }
}

extern "C" int sy11_fusion_combine(int32_t dtype, int32_t B, int32_t HW, int32_t C, int32_t n_in, const void* x0, int32_t ld0, const float* S0,
                                   const void* x1, int32_t ld1, const float* S1, const void* x2, int32_t ld2, const float* S2, const float* G,
                                   void* out, int32_t out_ld, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && B > 0 && HW > 0 && (n_in == 2 || n_in == 3) && G && out_ld >= C, "fusion_combine: bad argument");
  int lp, rc;
  if ((rc = fusion_geom(dtype, C, "fusion_combine", &lp))) return rc;
  FusionIn in{{x0, x1, x2}, {S0, S1, S2}, {ld0, ld1, ld2}, n_in};
  for (int i = 0; i < n_in; ++i) SY11_REQUIRE(fusion_view_ok(in.x[i], in.ld[i], dtype) && in.S[i] && in.ld[i] >= C, "fusion_combine: input %d view bad", i);
  SY11_REQUIRE(fusion_view_ok(out, out_ld, dtype), "fusion_combine: out view not 16-byte addressable");
  const int ppb = 256 / lp;
  int gx = cdiv(HW, ppb * 4);
  if (gx > 1024) gx = 1024;
  dim3 grid(gx, B), block(256);
  SY11_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((fusion_combine_kernel<T, 16 / (int)sizeof(T)>), grid, block, 0, (hipStream_t)stream, HW, C, lp, in, G,
                                                   (T*)out, out_ld));
  SY11_LAUNCH_CHECK("fusion_combine");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ backward passes
// reduce (per input): dG[b][c] += sum_pix dout * x ;  dS[pix] = sum_c dout * x
template <typename T, int VEC>
__global__ __launch_bounds__(256) void fusion_bwd_reduce_kernel(int HW, int C, int LP, const T* __restrict__ dout, int dout_ld, const T* __restrict__ x,
                                                                int x_ld, float* __restrict__ dG, int dg_ld, float* __restrict__ dS) {
  extern __shared__ float red[];                       // [C] (unused head) + [256][VEC] parking area
  const int b = blockIdx.y, tid = threadIdx.x;
  const int cl = tid % LP, sub = tid / LP, ppb = 256 / LP;
  float gacc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) gacc[i] = 0.f;
  for (int p = blockIdx.x * ppb + sub; p < HW; p += gridDim.x * ppb) {
    const long pix = (long)b * HW + p;
    float v[VEC], d[VEC];
    fload<T, VEC>(x + pix * x_ld + cl * VEC, v);
    fload<T, VEC>(dout + pix * dout_ld + cl * VEC, d);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float t = v[i] * d[i];
      s += t;
      gacc[i] += t;
    }
    for (int o = LP >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (cl == 0) dS[pix] = s;
  }
  float* park = red + C;
#pragma unroll
  for (int i = 0; i < VEC; ++i) park[tid * VEC + i] = gacc[i];
  __syncthreads();
  for (int i = tid; i < C; i += 256) {
    const int cv = i / VEC, j = i - cv * VEC;
    float t = 0.f;
    for (int k = 0; k < ppb; ++k) t += park[(k * LP + cv) * VEC + j];
    atomicAdd(dG + (long)b * dg_ld + i, t);
  }
}

extern "C" int sy11_fusion_bwd_reduce(int32_t dtype, int32_t B, int32_t HW, int32_t C, const void* dout, int32_t dout_ld, const void* x,
                                      int32_t x_ld, float* dG, int32_t dg_ld, float* dS, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && B > 0 && HW > 0 && dG && dS && dg_ld >= C && x_ld >= C && dout_ld >= C, "fusion_bwd_reduce: bad argument");
  int lp, rc;
  if ((rc = fusion_geom(dtype, C, "fusion_bwd_reduce", &lp))) return rc;
  SY11_REQUIRE(fusion_view_ok(x, x_ld, dtype) && fusion_view_ok(dout, dout_ld, dtype), "fusion_bwd_reduce: view not 16-byte addressable");
  const int ppb = 256 / lp;
  int gx = cdiv(HW, ppb * 8);
  if (gx > 512) gx = 512;
  if (sy11_det(32)) gx = 1;                              // ordered mode: one workgroup per image owns dG[b][:] (det.h)
  dim3 grid(gx, B), block(256);
  SY11_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((fusion_bwd_reduce_kernel<T, 16 / (int)sizeof(T)>), grid, block, (C + 256 * (16 / (int)sizeof(T))) * sizeof(float), (hipStream_t)stream,
                                                   HW, C, lp, (const T*)dout, dout_ld, (const T*)x, x_ld, dG, dg_ld, dS));
  SY11_LAUNCH_CHECK("fusion_bwd_reduce");
  return SY11_OK;
}

// apply (per input): dx (=|+=) dout * (G + S) + x * q + dmean / C + [c == argmax] dmax
template <typename T, int VEC>
__global__ __launch_bounds__(256) void fusion_bwd_apply_kernel(int HW, int C, int LP, const T* __restrict__ dout, int dout_ld, const T* __restrict__ x,
                                                               int x_ld, const float* __restrict__ G, const float* __restrict__ q, int g_ld,
                                                               const float* __restrict__ S, const float* __restrict__ dmm,
                                                               const unsigned short* __restrict__ amax, T* __restrict__ dx, int dx_ld, int accumulate) {
  const int b = blockIdx.y, tid = threadIdx.x;
  const int cl = tid % LP, sub = tid / LP, ppb = 256 / LP;
  float g[VEC], qq[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { g[j] = G[(long)b * g_ld + cl * VEC + j]; qq[j] = q[(long)b * g_ld + cl * VEC + j]; }
  const float invC = 1.0f / (float)C;
  for (int p = blockIdx.x * ppb + sub; p < HW; p += gridDim.x * ppb) {
    const long pix = (long)b * HW + p;
    float v[VEC], d[VEC], r[VEC];
    fload<T, VEC>(x + pix * x_ld + cl * VEC, v);
    fload<T, VEC>(dout + pix * dout_ld + cl * VEC, d);
    const float s = S[pix], dmean = dmm[pix * 2] * invC, dmax = dmm[pix * 2 + 1];
    const int am = amax[pix];
    if (accumulate) fload<T, VEC>(dx + pix * dx_ld + cl * VEC, r);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float t = d[j] * (g[j] + s) + v[j] * qq[j] + dmean + ((cl * VEC + j) == am ? dmax : 0.f);
      if (accumulate) t += r[j];
      r[j] = t;
    }
    fstore<T, VEC>(dx + pix * dx_ld + cl * VEC, r);
  }
}

extern "C" int sy11_fusion_bwd_apply(int32_t dtype, int32_t B, int32_t HW, int32_t C, const void* dout, int32_t dout_ld, const void* x,
                                     int32_t x_ld, const float* G, const float* q, int32_t g_ld, const float* S, const float* dmm,
                                     const uint16_t* amax, void* dx, int32_t dx_ld, int32_t accumulate, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && B > 0 && HW > 0 && G && q && S && dmm && amax && g_ld >= C && x_ld >= C && dout_ld >= C && dx_ld >= C,
               "fusion_bwd_apply: bad argument");
  int lp, rc;
  if ((rc = fusion_geom(dtype, C, "fusion_bwd_apply", &lp))) return rc;
  SY11_REQUIRE(fusion_view_ok(x, x_ld, dtype) && fusion_view_ok(dout, dout_ld, dtype) && fusion_view_ok(dx, dx_ld, dtype),
               "fusion_bwd_apply: view not 16-byte addressable");
  const int ppb = 256 / lp;
  int gx = cdiv(HW, ppb * 4);
  if (gx > 1024) gx = 1024;
  dim3 grid(gx, B), block(256);
  SY11_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((fusion_bwd_apply_kernel<T, 16 / (int)sizeof(T)>), grid, block, 0, (hipStream_t)stream, HW, C, lp,
                                                   (const T*)dout, dout_ld, (const T*)x, x_ld, G, q, g_ld, S, dmm, amax, (T*)dx, dx_ld, accumulate));
  SY11_LAUNCH_CHECK("fusion_bwd_apply");
  return SY11_OK;
}
