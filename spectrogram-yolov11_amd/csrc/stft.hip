// stft.hip — IQ -> STFT -> |X|^2 -> triangular "mel" bank -> 10*log10 producer for gfx950.
//
// The reference has no implementation of this stage (README.md:7 is prose); the spec is the build's own (DESIGN.md,
// oracle/stft_ref.py).  One 256-thread workgroup per (image, frame): the windowed frame (n_fft complex samples) is
// bit-reverse scattered into LDS, transformed by an in-place radix-2 DIT FFT (log2(n_fft) LDS stages, twiddles from an
// LDS table built with sincospi), fftshift-ed on read, reduced by the sparse filter bank (<= mel_taps bins per filter,
// gather form: no atomics) and written frame-major (B, frames, n_mel) so every store is a full coalesced run; the
// per-image min/max needed by the normalisation is folded in with order-preserving integer atomics.
// HBM-bound by design: 8 B/sample in (each sample re-read by n_fft/hop overlapping frames out of L2) + 4 B/bin out.
#include "common.h"
#include <stdlib.h>

__device__ __forceinline__ void atomic_min_f(float* p, float v) {
  if (v >= 0.f) atomicMin((int*)p, __float_as_int(v)); else atomicMax((unsigned*)p, __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f(float* p, float v) {
  if (v >= 0.f) atomicMax((int*)p, __float_as_int(v)); else atomicMin((unsigned*)p, __float_as_uint(v));
}

__global__ __launch_bounds__(256) void stft_logmel_kernel(int L, int n_fft, int log2n, int hop, int n_frames, int n_mel, const float2* __restrict__ iq,
                                                          const float* __restrict__ window, const int* __restrict__ mel_start,
                                                          const float* __restrict__ mel_w, int mel_taps, float* __restrict__ db, float* __restrict__ minmax) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* re = sm;                  // [n_fft]
  float* im = re + n_fft;          // [n_fft]
  float* twc = im + n_fft;         // [n_fft/2]
  float* tws = twc + n_fft / 2;    // [n_fft/2]
  __shared__ float rmin[4], rmax[4];
  const int tid = threadIdx.x;
  const int frame = blockIdx.x, b = blockIdx.y;
  const float2* src = iq + (long)b * L + (long)frame * hop;
  for (int k = tid; k < n_fft / 2; k += 256) {
    float s, c;
    sincospif(-2.0f * (float)k / (float)n_fft, &s, &c);
    twc[k] = c; tws[k] = s;
  }
  for (int n = tid; n < n_fft; n += 256) {
    const float2 v = src[n];
    const float w = window[n];
    const int r = __brev((unsigned)n) >> (32 - log2n);
    re[r] = v.x * w; im[r] = v.y * w;
  }
  __syncthreads();
  for (int s = 1; s <= log2n; ++s) {
    const int half = 1 << (s - 1);
    const int tstep = n_fft >> s;
    for (int t = tid; t < n_fft / 2; t += 256) {
      const int j = t & (half - 1);
      const int i0 = ((t >> (s - 1)) << s) + j, i1 = i0 + half;
      const float c = twc[j * tstep], sn = tws[j * tstep];
      const float ur = re[i0], ui = im[i0];
      const float xr = re[i1], xi = im[i1];
      const float vr = xr * c - xi * sn, vi = xr * sn + xi * c;
      re[i0] = ur + vr; im[i0] = ui + vi;
      re[i1] = ur - vr; im[i1] = ui - vi;
    }
    __syncthreads();
  }
  // power spectrum in place (re), fftshift applied by the reader: shifted bin k <- fft bin (k + n/2) mod n
  for (int n = tid; n < n_fft; n += 256) re[n] = re[n] * re[n] + im[n] * im[n];
  __syncthreads();
  float lmin = INFINITY, lmax = -INFINITY;
  float* out = db + ((long)b * n_frames + frame) * n_mel;
  const int hmask = n_fft - 1, hshift = n_fft >> 1;
  for (int j = tid; j < n_mel; j += 256) {
    const int k0 = mel_start[j];
    float acc = 0.f;
    for (int t = 0; t < mel_taps; ++t) {
      const float w = mel_w[j * mel_taps + t];
      const int k = k0 + t;
      if (w != 0.f && k < n_fft) acc += w * re[(k + hshift) & hmask];
    }
    const float v = 10.0f * log10f(acc + 1e-10f);
    out[j] = v;
    lmin = fminf(lmin, v); lmax = fmaxf(lmax, v);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { lmin = fminf(lmin, __shfl_xor(lmin, o)); lmax = fmaxf(lmax, __shfl_xor(lmax, o)); }
  if ((tid & 63) == 0) { rmin[tid >> 6] = lmin; rmax[tid >> 6] = lmax; }
  __syncthreads();
  if (tid == 0) {
    const float mn = fminf(fminf(rmin[0], rmin[1]), fminf(rmin[2], rmin[3]));
    const float mx = fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
    atomic_min_f(minmax + 2 * b, mn);
    atomic_max_f(minmax + 2 * b + 1, mx);
  }
}

// n_fft = 1024 fast path: Stockham autosort radix-4, one butterfly per thread per pass (5 passes, natural order in and
// out, no bit reversal), values in registers, ping-pong float2 LDS images, twiddles by sincospi (w2 = w1^2, w3 = w1 w2).
// LDS traffic per frame: 4 exchanges of 8 KB each way instead of 10 radix-2 stages over 3 arrays; 5 barriers instead of 11.
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ __launch_bounds__(256) void stft_logmel1024_kernel(int L, int hop, int n_frames, int n_mel, int fpb, const float2* __restrict__ iq,
                                                              const float* __restrict__ window, const int* __restrict__ mel_start,
                                                              const float* __restrict__ mel_w, int mel_taps, float* __restrict__ db,
                                                              float* __restrict__ minmax) {
  constexpr int N = 1024, Q = N / 4;
  __shared__ float2 bufA[N], bufB[N];
  __shared__ float rmin[4], rmax[4];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  // per-thread constants shared by the `fpb` frames of this workgroup: window taps and the twiddles of passes 1..4
  float wnd[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) wnd[r] = window[tid + r * Q];
  float2 tw[4][3];
#pragma unroll
  for (int pass = 1; pass < 5; ++pass) {
    const int Ns = 1 << (2 * pass);
    float sn, cs;
    sincospif(-2.0f * (float)(tid & (Ns - 1)) / (float)(4 * Ns), &sn, &cs);
    tw[pass - 1][0] = make_float2(cs, sn);
    tw[pass - 1][1] = cmul(tw[pass - 1][0], tw[pass - 1][0]);
    tw[pass - 1][2] = cmul(tw[pass - 1][0], tw[pass - 1][1]);
  }
  float lmin = INFINITY, lmax = -INFINITY;
  for (int fi = 0; fi < fpb; ++fi) {
    const int frame = blockIdx.x * fpb + fi;
    if (frame >= n_frames) break;
    const float2* src = iq + (long)b * L + (long)frame * hop;
    float2 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float2 x = src[tid + r * Q];
      v[r] = make_float2(x.x * wnd[r], x.y * wnd[r]);
    }
    float2* in = bufA;
    float2* out = bufB;
#pragma unroll
    for (int pass = 0; pass < 5; ++pass) {
      const int Ns = 1 << (2 * pass);
      const int k = tid & (Ns - 1);
      if (pass > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = in[tid + r * Q];
        v[1] = cmul(v[1], tw[pass - 1][0]); v[2] = cmul(v[2], tw[pass - 1][1]); v[3] = cmul(v[3], tw[pass - 1][2]);
      }
      // 4-point DFT (forward: e^{-2 pi i t u / 4})
      const float2 a0 = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), a1 = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
      const float2 a2 = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), a3 = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
      float2 y[4];
      y[0] = make_float2(a0.x + a2.x, a0.y + a2.y);
      y[2] = make_float2(a0.x - a2.x, a0.y - a2.y);
      y[1] = make_float2(a1.x + a3.y, a1.y - a3.x);          // a1 - i a3
      y[3] = make_float2(a1.x - a3.y, a1.y + a3.x);          // a1 + i a3
      const int base = ((tid - k) << 2) + k;                 // (j / Ns) * 4 Ns + k
      if (pass < 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) out[base + u * Ns] = y[u];
        __syncthreads();
        float2* t = in; in = out; out = t;
      } else {
        float* pwr = (float*)out;                            // power spectrum, natural order
#pragma unroll
        for (int u = 0; u < 4; ++u) pwr[base + u * Ns] = y[u].x * y[u].x + y[u].y * y[u].y;
        __syncthreads();
      }
    }
    const float* pw = (const float*)out;
    float* o = db + ((long)b * n_frames + frame) * n_mel;
    for (int j = tid; j < n_mel; j += 256) {
      const int k0 = mel_start[j];
      float acc = 0.f;
      if (mel_taps == 8) {                                   // the build's bank: all 8 weights in two 16-byte loads (one latency)
        const f32x4 wa = *(const f32x4*)(mel_w + j * 8), wb = *(const f32x4*)(mel_w + j * 8 + 4);
        const float wv[8] = {wa[0], wa[1], wa[2], wa[3], wb[0], wb[1], wb[2], wb[3]};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const int kk = k0 + t;
          acc += (kk < N ? wv[t] : 0.f) * pw[(kk + N / 2) & (N - 1)];
        }
      } else {
        for (int t = 0; t < mel_taps; ++t) {
          const float w = mel_w[j * mel_taps + t];
          const int kk = k0 + t;
          if (w != 0.f && kk < N) acc += w * pw[(kk + N / 2) & (N - 1)];
        }
      }
      const float val = 10.0f * log10f(acc + 1e-10f);
      o[j] = val;
      lmin = fminf(lmin, val); lmax = fmaxf(lmax, val);
    }
    __syncthreads();                                         // the power image is overwritten by the next frame's pass 0/1
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { lmin = fminf(lmin, __shfl_xor(lmin, off)); lmax = fmaxf(lmax, __shfl_xor(lmax, off)); }
  if ((tid & 63) == 0) { rmin[tid >> 6] = lmin; rmax[tid >> 6] = lmax; }
  __syncthreads();
  if (tid == 0) {
    atomic_min_f(minmax + 2 * b, fminf(fminf(rmin[0], rmin[1]), fminf(rmin[2], rmin[3])));
    atomic_max_f(minmax + 2 * b + 1, fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3])));
  }
}

__global__ void stft_minmax_init_kernel(int B, float* minmax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { minmax[2 * i] = INFINITY; minmax[2 * i + 1] = -INFINITY; }
}

// img[b][c][f][t] = (db[b][t][f] - min) / (max - min): 32x32 LDS transpose, three channel planes written per tile
__global__ __launch_bounds__(256) void stft_normalize_kernel(int n_mel, int n_frames, const float* __restrict__ db, const float* __restrict__ minmax,
                                                             float* __restrict__ img) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, f0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float mn = minmax[2 * b], mx = minmax[2 * b + 1];
  const float inv = 1.0f / fmaxf(mx - mn, 1e-12f);
  for (int j = ty; j < 32; j += 8) {
    const int t = t0 + j, f = f0 + tx;
    tile[j][tx] = (t < n_frames && f < n_mel) ? (db[((long)b * n_frames + t) * n_mel + f] - mn) * inv : 0.f;
  }
  __syncthreads();
  const long plane = (long)n_mel * n_frames;
  for (int j = ty; j < 32; j += 8) {
    const int f = f0 + j, t = t0 + tx;
    if (f < n_mel && t < n_frames) {
      const float v = tile[tx][j];
      float* o = img + (long)b * 3 * plane + (long)f * n_frames + t;
      o[0] = v; o[plane] = v; o[2 * plane] = v;
    }
  }
}

extern "C" int sy11_stft_minmax_init(int32_t B, float* minmax, void* stream) {
  SY11_REQUIRE(B > 0 && minmax, "stft_minmax_init: bad argument");
  hipLaunchKernelGGL(stft_minmax_init_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, B, minmax);
  SY11_LAUNCH_CHECK("stft_minmax_init");
  return SY11_OK;
}

extern "C" int sy11_stft_logmel(int32_t B, int32_t L, int32_t n_fft, int32_t hop, int32_t n_frames, int32_t n_mel,
                                const float* iq, const float* window, const int32_t* mel_start, const float* mel_w,
                                int32_t mel_taps, float* db, float* minmax, void* stream) {
  SY11_REQUIRE(B > 0 && L > 0 && hop > 0 && n_frames > 0 && n_mel > 0 && mel_taps > 0, "stft_logmel: non-positive dims");
  SY11_REQUIRE(iq && window && mel_start && mel_w && db && minmax, "stft_logmel: null pointer");
  int log2n = 0;
  while ((1 << log2n) < n_fft) ++log2n;
  SY11_REQUIRE((1 << log2n) == n_fft && n_fft >= 64 && n_fft <= 8192, "stft_logmel: n_fft must be a power of two in [64, 8192]");
  SY11_REQUIRE((long)n_fft + (long)(n_frames - 1) * hop <= L, "stft_logmel: %d frames of %d with hop %d do not fit in L=%d", n_frames, n_fft, hop, L);
  SY11_REQUIRE(B <= 65535, "stft_logmel: B too large");
  SY11_REQUIRE(((uintptr_t)iq & 7) == 0, "stft_logmel: iq must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  static int radix2 = -1;
  if (radix2 < 0) { const char* e = getenv("SY11_STFT_RADIX2"); radix2 = e ? atoi(e) : 0; }
  if (n_fft == 1024 && !radix2) {
    const int fpb = n_frames >= 256 ? 4 : 1;               // frames per workgroup (window taps / twiddles computed once)
    SY11_REQUIRE(((uintptr_t)mel_w & 15) == 0 || mel_taps != 8, "stft_logmel: mel_w must be 16-byte aligned");
    hipLaunchKernelGGL(stft_logmel1024_kernel, dim3(cdiv(n_frames, fpb), B), dim3(256), 0, st, L, hop, n_frames, n_mel, fpb, (const float2*)iq, window,
                       mel_start, mel_w, mel_taps, db, minmax);
    SY11_LAUNCH_CHECK("stft_logmel");
    return SY11_OK;
  }
  const size_t lds = (size_t)3 * n_fft * sizeof(float);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)stft_logmel_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(stft_logmel_kernel, dim3(n_frames, B), dim3(256), lds, st, L, n_fft, log2n, hop, n_frames, n_mel, (const float2*)iq, window,
                     mel_start, mel_w, mel_taps, db, minmax);
  SY11_LAUNCH_CHECK("stft_logmel");
  return SY11_OK;
}

extern "C" int sy11_stft_normalize(int32_t B, int32_t n_mel, int32_t n_frames, const float* db, const float* minmax,
                                   float* img_nchw, void* stream) {
  SY11_REQUIRE(B > 0 && B <= 65535 && n_mel > 0 && n_frames > 0 && db && minmax && img_nchw, "stft_normalize: bad argument");
  hipLaunchKernelGGL(stft_normalize_kernel, dim3(cdiv(n_frames, 32), cdiv(n_mel, 32), B), dim3(256), 0, (hipStream_t)stream, n_mel, n_frames, db,
                     minmax, img_nchw);
  SY11_LAUNCH_CHECK("stft_normalize");
  return SY11_OK;
}
