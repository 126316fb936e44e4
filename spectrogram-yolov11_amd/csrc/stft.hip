// stft.hip — IQ -> STFT -> |X|^2 -> triangular "mel" bank -> 10*log10 producer for gfx950.
//
// The reference has no implementation of this stage (README.md:7 is prose); the spec is the build's own (DESIGN.md,
// oracle/stft_ref.py).  One 256-thread workgroup per (image, frame): the windowed frame (n_fft complex samples) is
// bit-reverse scattered into LDS, transformed by an in-place radix-2 DIT FFT (log2(n_fft) LDS stages, twiddles from an
// LDS table built with sincospi), fftshift-ed on read, reduced by the sparse filter bank (<= mel_taps bins per filter,
// gather form: no atomics) and written frame-major (B, frames, n_mel) so every store is a full coalesced run; the
// per-image min/max needed by the normalisation is folded in with order-preserving integer atomics.
// HBM-bound by design: 8 B/sample in (each sample re-read by n_fft/hop overlapping frames out of L2) + 4 B/bin out.
// Three forms: the generic radix-2 kernel below (any power-of-two n_fft), the r01 radix-4 Stockham kernel for n_fft = 1024 (any hop / bank),
// and — the one the product shape runs — the r04 wave-per-frame kernel (n_fft 1024, hop 256, 8-tap bank): 180 -> 122 us at 64 x 640
// frames.  Ablations of that kernel (tools/stft_micro.py, r04): without the filter bank 80 us, without the two radix-16 butterflies
// 107 us — the butterflies are 13 us; what is left is latency (four LDS round trips per frame at 3 waves per SIMD) and the bank.
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

// n_fft = 1024, hop = 256, 8-tap bank (r04): ONE WAVE PER FRAME, no workgroup barrier inside the frame loop.
//   1024 = 16 x 16 x 4.  A lane holds x[64 r + lane], r = 0..15 (coalesced loads), so the radix-16 butterfly over r runs in registers;
//   ONE exchange through LDS (lane n' -> lane (k1, c), n' = 4 a + c) puts the 16 inputs of the second radix-16 butterfly (over a) into
//   one lane; the closing radix-4 (over c = lane & 3) runs across the four lanes of a quad with DPP broadcasts.  Twiddle tables
//   (W_1024^{n' k1}, W_64^{c p}) live in LDS, built once per workgroup.  r03 form above: 5 radix-4 passes through ping-pong LDS images
//   with a workgroup barrier each (10 trips through LDS, 6 barriers per frame).
//   A wave walks `run` CONSECUTIVE frames: frame f + 1 is frame f shifted by hop = 256 samples = 4 register slots, so 12 of the 16
//   raw samples a lane holds carry over and every IQ sample is loaded once per run instead of once per overlapping frame (4x).
//   The power spectrum goes to LDS fftshift-ed, its first 8 words repeated behind the end, and is read back by the gather-form filter
//   bank, 10 filters per lane, 8 consecutive words each.
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {   // forward: y_u = sum_t a_t e^{-2 pi i t u / 4}
  const float2 s02 = cadd(a0, a2), d02 = csub(a0, a2), s13 = cadd(a1, a3), d13 = csub(a1, a3);
  a0 = cadd(s02, s13);
  a2 = csub(s02, s13);
  a1 = make_float2(d02.x + d13.y, d02.y - d13.x);           // d02 - i d13
  a3 = make_float2(d02.x - d13.y, d02.y + d13.x);           // d02 + i d13
}
// in-place 16-point forward DFT of v[n]; the result X[k] is left at v[4 (k & 3) + (k >> 2)]  (digit-reversed: see DIG())
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
  constexpr float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R2 = 0.70710678118654752f;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) dft4(v[nb], v[4 + nb], v[8 + nb], v[12 + nb]);      // over n_a: v[4 k_a + n_b]
  // twiddles W_16^{n_b k_a} = cos(pi m / 8) - i sin(pi m / 8), m = n_b k_a
  v[4 + 1] = cmul(v[4 + 1], make_float2(C1, -S1));   v[8 + 1] = cmul(v[8 + 1], make_float2(R2, -R2));    v[12 + 1] = cmul(v[12 + 1], make_float2(S1, -C1));
  v[4 + 2] = cmul(v[4 + 2], make_float2(R2, -R2));   v[8 + 2] = make_float2(v[8 + 2].y, -v[8 + 2].x);    v[12 + 2] = cmul(v[12 + 2], make_float2(-R2, -R2));
  v[4 + 3] = cmul(v[4 + 3], make_float2(S1, -C1));   v[8 + 3] = cmul(v[8 + 3], make_float2(-R2, -R2));   v[12 + 3] = cmul(v[12 + 3], make_float2(-C1, S1));
#pragma unroll
  for (int ka = 0; ka < 4; ++ka) dft4(v[4 * ka], v[4 * ka + 1], v[4 * ka + 2], v[4 * ka + 3]);   // over n_b: v[4 k_a + k_b] = X[k_a + 4 k_b]
}
#define SY11_DIG(k) ((((k) & 3) << 2) | ((k) >> 2))
template <int CTRL> __device__ __forceinline__ float quad_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

template <int WAVES, int MINB, int MELU>
__global__ __launch_bounds__(WAVES * 64, MINB) void stft_logmel1024w_kernel(int L, int n_frames, int n_mel, int run, const float2* __restrict__ iq,
                                                               const float* __restrict__ window, const int* __restrict__ mel_start,
                                                               const float* __restrict__ mel_w, float* __restrict__ db, float* __restrict__ minmax) {
  constexpr int N = 1024, EXS = 68;                         // exchange row stride in float2 (544 B: a quad's 32 bytes land 8 banks apart)
  __shared__ float2 s_twA[16 * 64];                         // [k1][n'] = W_1024^{n' k1}
  __shared__ float2 s_twB[4 * 16];                          // [c][p]   = W_64^{c p}
  __shared__ __attribute__((aligned(16))) float s_ex[WAVES][16 * EXS * 2];   // per wave: the exchange image, then the padded power spectrum
  __shared__ float rmin[WAVES], rmax[WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  for (int i = tid; i < 1024; i += WAVES * 64) {
    float sn, cs;
    sincospif(-2.0f * (float)((i & 63) * (i >> 6)) / 1024.0f, &sn, &cs);
    s_twA[i] = make_float2(cs, sn);
  }
  if (tid < 64) {
    float sn, cs;
    sincospif(-2.0f * (float)((tid >> 4) * (tid & 15)) / 64.0f, &sn, &cs);
    s_twB[tid] = make_float2(cs, sn);
  }
  float wnd[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) wnd[r] = window[64 * r + lane];
  const int k1 = lane >> 2, q = lane & 3;
  // closing radix-4 across the quad as two butterfly steps (partner lane ^ 2, then lane ^ 1): per-lane signs, and lane 3 pre-rotates by -i
  const float sg1 = q < 2 ? 1.f : -1.f, sg2 = (q & 1) ? -1.f : 1.f;
  const bool rot3 = q == 3;
  const int qo = ((q & 1) << 1) | (q >> 1);                 // lane c ends up with output index {0, 2, 1, 3}[c]
  __syncthreads();
  float2* ex = (float2*)s_ex[wave];
  float* pw = s_ex[wave];
  float lmin = INFINITY, lmax = -INFINITY;
  const int f0 = (blockIdx.x * WAVES + wave) * run;
  float2 raw[16];
  if (f0 < n_frames) {
    const float2* src = iq + (long)b * L + (long)f0 * 256;
#pragma unroll
    for (int r = 0; r < 16; ++r) raw[r] = src[64 * r + lane];
  }
  for (int fi = 0; fi < run; ++fi) {
    const int frame = f0 + fi;
    if (frame >= n_frames) break;                            // wave-uniform
    float2 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = make_float2(raw[r].x * wnd[r], raw[r].y * wnd[r]);
    if (fi + 1 < run && frame + 1 < n_frames) {              // next frame = this one shifted by 256 samples = 4 register slots
      const float2* nsrc = iq + (long)b * L + (long)(frame + 1) * 256;
#pragma unroll
      for (int r = 0; r < 12; ++r) raw[r] = raw[r + 4];
#pragma unroll
      for (int r = 12; r < 16; ++r) raw[r] = nsrc[64 * r + lane];
    }
    dft16(v);                                                // over r: A[n' = lane][k1] at v[DIG(k1)]
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) ex[kk * EXS + lane] = cmul(v[SY11_DIG(kk)], s_twA[kk * 64 + lane]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // one wave: its LDS operations complete in order; no barrier needed
#pragma unroll
    for (int aa = 0; aa < 16; ++aa) v[aa] = ex[k1 * EXS + 4 * aa + q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    dft16(v);                                                // over a: C[c = q][p] at v[DIG(p)]
    float pwr[16];
#pragma unroll
    for (int pp = 0; pp < 16; ++pp) {
      const float2 c = cmul(v[SY11_DIG(pp)], s_twB[q * 16 + pp]);
      // closing radix-4 over the quad, Y[p + 16 u] = sum_c C_c W_4^{c u}:  lanes 0..3 hold C_0..C_3
      //   step 1 (partner = lane ^ 2): lanes 0, 1: t0 = C0 + C2, t2 = C1 + C3;  lanes 2, 3: t1 = C0 - C2, t3 = C1 - C3 (lane 3 keeps -i t3)
      //   step 2 (partner = lane ^ 1): lane 0: Y0 = t0 + t2, lane 1: Y2 = t0 - t2, lane 2: Y1 = t1 - i t3, lane 3: Y3 = t1 + i t3
      const float ax = fmaf(c.x, sg1, quad_bcast<0x4E>(c.x)), ay = fmaf(c.y, sg1, quad_bcast<0x4E>(c.y));
      const float rx = rot3 ? ay : ax, ry = rot3 ? -ax : ay;
      const float yx = fmaf(rx, sg2, quad_bcast<0xB1>(rx)), yy = fmaf(ry, sg2, quad_bcast<0xB1>(ry));
      pwr[pp] = yx * yx + yy * yy;                           // bin k = k1 + 16 p + 256 qo
    }
    // the spectrum is stored fftshift-ed (bin k at (k + 512) mod 1024 = k1 + 16 p + 256 (qo ^ 2)) with its first 8 entries repeated behind
    // the end, so a filter's 8 taps are 8 consecutive words at immediate offsets — no per-tap index arithmetic, no wrap test
#pragma unroll
    for (int pp = 0; pp < 16; ++pp) pw[k1 + 16 * pp + 256 * (qo ^ 2)] = pwr[pp];
    if (qo == 2 && k1 < 8) pw[N + k1] = pwr[0];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* o = db + ((long)b * n_frames + frame) * n_mel;
#pragma unroll MELU
    for (int j = lane; j < n_mel; j += 64) {
      const int k0 = mel_start[j];
      const int s0 = k0 & (N - 1);                           // the image is already shifted: shifted bin k0 IS word k0
      f32x4 wa = *(const f32x4*)(mel_w + j * 8), wb = *(const f32x4*)(mel_w + j * 8 + 4);
      if (k0 > N - 8) {                                      // taps past bin N - 1 do not exist (the generic kernel's `k < n_fft`); rare
#pragma unroll
        for (int t = 0; t < 4; ++t) { if (k0 + t >= N) wa[t] = 0.f; if (k0 + 4 + t >= N) wb[t] = 0.f; }
      }
      const float* pj = pw + s0;
      float acc = wa[0] * pj[0];
      acc += wa[1] * pj[1]; acc += wa[2] * pj[2]; acc += wa[3] * pj[3];
      acc += wb[0] * pj[4]; acc += wb[1] * pj[5]; acc += wb[2] * pj[6]; acc += wb[3] * pj[7];
      const float val = 3.0102999566398120f * __log2f(acc + 1e-10f);     // 10 log10(x) = 10 log10(2) log2(x)
      o[j] = val;
      lmin = fminf(lmin, val); lmax = fmaxf(lmax, val);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the spectrum has been read: the next frame's exchange may overwrite it
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { lmin = fminf(lmin, __shfl_xor(lmin, off)); lmax = fmaxf(lmax, __shfl_xor(lmax, off)); }
  if (lane == 0) { rmin[wave] = lmin; rmax[wave] = lmax; }
  __syncthreads();
  if (tid == 0) {
    float mn = rmin[0], mx = rmax[0];
#pragma unroll
    for (int i = 1; i < WAVES; ++i) { mn = fminf(mn, rmin[i]); mx = fmaxf(mx, rmax[i]); }
    if (mn <= mx) {                                          // a workgroup past the last frame contributes nothing
      atomic_min_f(minmax + 2 * b, mn);
      atomic_max_f(minmax + 2 * b + 1, mx);
    }
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
  static int wave_fft = -1;
  if (wave_fft < 0) { const char* e = getenv("SY11_STFT_WAVE"); wave_fft = e ? atoi(e) : 1; }
  if (n_fft == 1024 && hop == 256 && mel_taps == 8 && wave_fft && !radix2 && ((uintptr_t)mel_w & 15) == 0) {
    // one wave per frame, `run` consecutive frames per wave (shared samples stay in registers): 5 keeps 2 048 workgroups for 64 x 640 frames
    const int run = n_frames >= 320 ? 5 : (n_frames >= 16 ? 2 : 1);
    // 3 waves per SIMD (r04): the register allocator's free choice is 170 VGPRs = 2 waves per SIMD; bounded to 168 the kernel keeps a third
    // wave to cover the LDS exchanges (130 -> 116 us).  128 VGPRs (4 waves) spills 74 registers and loses (169 us); 8-wave workgroups change nothing.
    hipLaunchKernelGGL((stft_logmel1024w_kernel<4, 3, 5>), dim3(cdiv(n_frames, 4 * run), B), dim3(256), 0, st, L, n_frames, n_mel, run, (const float2*)iq, window,
                       mel_start, mel_w, db, minmax);
    SY11_LAUNCH_CHECK("stft_logmel");
    return SY11_OK;
  }
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
