// wgrad.hip — filter gradient of the NHWC convolution on MFMA (gfx950).
//
// dW[n][t][c] += sum_m dy[m][n] * x[pix(m,t)][c]        (m = output pixel, t = tap)
// GEMM view: rows = n (output channel), cols = kk = (t, c), reduction over the B*OH*OW pixels.
// Both operands are "pixel-major" in HBM (NHWC), i.e. the reduction index is the SLOW index of both tiles, so
// the tiles are staged as [pixel][n] and [pixel][kk] f32 images in LDS and the one-float-per-lane fragments of
// v_mfma_f32_32x32x2_f32 (A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31]) are read with conflict-free
// ds_read_b32 (32 consecutive floats per lane half).  f16/bf16 inputs are widened to f32 on the way into LDS
// (round 1: exact products, f32 MFMA rate; the 16-bit MFMA path with ds_read_b64_tr_b16 is the next step).
// The pixel range is split over gridDim.y; partial tiles are added to dW with f32 atomics whose 32-lane groups
// cover 128 contiguous bytes.
#include "common.h"
#include "tune.h"
#include "det.h"
#include <stdlib.h>

struct WgradArgs {
  const void* x;
  const void* dy;
  float* dw;
  int M, N, K, C, T;
  int x_ld, dy_ld;
  int IH, IW, OH, OW, sy, sx;
  int tiles_k;
  unsigned x_bytes, dy_bytes;   // extents of the x / dy views (buffer descriptors: out-of-range offset = zero fill)
  unsigned mag_ow, mag_oh;      // ceil(2^20 / OW), ceil(2^20 / OH): exact quotients for the small ranges of the pixel walk
  int tiles, total;      // tiles per pixel split, tiles * splits
  int pix_per_split;     // multiple of BP
  long dw_split_stride;  // 0: every pixel split adds into the one dW (atomics); ordered mode: split s owns dw + s * stride (det.h)
  signed char tap_dy[64];
  signed char tap_dx[64];
};

// dW element of a finished tile: f32 atomic add (pixel splits meet in dW), or — ordered mode, where split s owns its own partial dW
// and every element of it has exactly ONE writer — a plain store (no zero fill of the partials, no atomics)
template <bool STORE>
__device__ __forceinline__ void dw_out(float* p, float v) {
  if (STORE) *p = v; else atomicAdd(p, v);
}

template <typename T>
__device__ __forceinline__ void load_chunk_f32(const T* p, float* out);
template <>
__device__ __forceinline__ void load_chunk_f32<float>(const float* p, float* out) {
  const f32x4 v = *(const f32x4*)p;
  out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}
template <>
__device__ __forceinline__ void load_chunk_f32<_Float16>(const _Float16* p, float* out) {
  const f16x8 v = *(const f16x8*)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}
template <>
__device__ __forceinline__ void load_chunk_f32<__bf16>(const __bf16* p, float* out) {
  const bf16x8 v = *(const bf16x8*)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}

// TILE = 128 (2x2 waves, 2x2 MFMA tiles each) or 64 (2x2 waves, 1 MFMA tile each)
template <typename T, int TILE>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs a) {
  constexpr int BP = 32;                          // pixels per stage
  constexpr int EPC = 16 / (int)sizeof(T);        // elements per 16-byte global chunk
  constexpr int CPR = TILE / EPC;                 // chunks per tile row
  constexpr int RPP = 256 / CPR;                  // pixel rows covered per pass
  constexpr int NPASS = BP / RPP;
  constexpr int FI = TILE / 64;                   // MFMA tiles per wave per dim
  constexpr int LDW = TILE + 0;                   // f32 words per LDS row
  static_assert(NPASS >= 1, "tile too wide");
  __shared__ __attribute__((aligned(16))) float smem[2][2][BP][LDW];   // [stage][dy|x][pixel][col]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order: hardware deals linear workgroup ids round-robin to the 8 XCDs; give each XCD a contiguous run of
  // (split, tile) pairs so all tiles of one pixel split — which re-read the same dy / x rows — share one L2.
  const int per_xcd = gridDim.x >> 3;
  const int vid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (vid >= a.total) return;
  const int split = vid / a.tiles, tile = vid - split * a.tiles;
  const int tile_k = tile % a.tiles_k, tile_n = tile / a.tiles_k;
  const int n0 = tile_n * TILE, k0 = tile_k * TILE;
  const int m_begin = split * a.pix_per_split;
  const int m_end = min(a.M, m_begin + a.pix_per_split);
  float* const dwp = a.dw + (long)split * a.dw_split_stride;
  if (m_begin >= m_end) return;

  const int chunk = tid % CPR, prow = tid / CPR;
  // dy chunk: channels n0 + chunk*EPC ..; x chunk: kk = k0 + chunk*EPC -> (t, c) fixed for the whole pixel loop
  const int ncol = n0 + chunk * EPC;
  const bool n_ok = ncol < a.N;                     // N % EPC == 0 is required by the host
  const int kk = k0 + chunk * EPC;
  const bool k_ok = kk < a.K;
  const int kt = k_ok ? kk / a.C : 0, kc = k_ok ? kk - kt * a.C : 0;
  const int tdy = a.tap_dy[kt], tdx = a.tap_dx[kt];
  const int ohw = a.OH * a.OW;
  const T* __restrict__ xg = (const T*)a.x;
  const T* __restrict__ dyg = (const T*)a.dy;

  float rdy[NPASS][EPC], rx[NPASS][EPC];
  auto load_stage = [&](int m0) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int m = m0 + prow + p * RPP;
      const bool ok = m < m_end;
#pragma unroll
      for (int e = 0; e < EPC; ++e) rdy[p][e] = rx[p][e] = 0.f;
      if (ok && n_ok) load_chunk_f32<T>(dyg + (long)m * a.dy_ld + ncol, rdy[p]);
      if (ok && k_ok) {
        const int b = m / ohw, r = m - b * ohw, oy = r / a.OW, ox = r - oy * a.OW;
        const int iy = oy * a.sy + tdy, ix = ox * a.sx + tdx;
        if ((unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW)
          load_chunk_f32<T>(xg + (long)((b * a.IH + iy) * a.IW + ix) * a.x_ld + kc, rx[p]);
      }
    }
  };
  auto store_stage = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int r = prow + p * RPP;
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        *(f32x4*)&smem[buf][0][r][chunk * EPC + e] = f32x4{rdy[p][e], rdy[p][e + 1], rdy[p][e + 2], rdy[p][e + 3]};
        *(f32x4*)&smem[buf][1][r][chunk * EPC + e] = f32x4{rx[p][e], rx[p][e + 1], rx[p][e + 2], rx[p][e + 3]};
      }
    }
  };

  f32x16 acc[FI][FI];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int wr = wave >> 1, wc = wave & 1;
  const int fcol = lane & 31, fh = lane >> 5;

  const int nstage = (m_end - m_begin + BP - 1) / BP;
  load_stage(m_begin);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < nstage; ++s) {
    const bool more = s + 1 < nstage;
    if (more) load_stage(m_begin + (s + 1) * BP);
    const int buf = s & 1;
#pragma unroll
    for (int pp = 0; pp < BP / 2; ++pp) {
      float fa[FI], fb[FI];
#pragma unroll
      for (int i = 0; i < FI; ++i) fa[i] = smem[buf][0][2 * pp + fh][wr * (TILE / 2) + i * 32 + fcol];
#pragma unroll
      for (int j = 0; j < FI; ++j) fb[j] = smem[buf][1][2 * pp + fh][wc * (TILE / 2) + j * 32 + fcol];
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_stage(buf ^ 1);
    __syncthreads();
  }
  // D map: col = lane&31 (kk), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (n)
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wr * (TILE / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        const int k = k0 + wc * (TILE / 2) + j * 32 + fcol;
        if (n < a.N && k < a.K) { if (a.dw_split_stride) dw_out<true>(dwp + (long)n * a.K + k, acc[i][j][e]); else dw_out<false>(dwp + (long)n * a.K + k, acc[i][j][e]); }
      }
}


// ------------------------------------------------------------------------------------------------ 16-bit MFMA path
// Same GEMM, but the tiles stay f16/bf16 in LDS ([pixel][col], row stride TILE*2 + 64 bytes) and the MFMA operands —
// 8 consecutive PIXELS of one column per lane — come from the hardware-transposing ds_read_b64_tr_b16 (two reads of
// 4 pixels x 16 columns per 16-lane group).  The +64-byte row pad puts the 4 rows one read touches on 4 disjoint
// 16-bank ranges (row stride = 64 mod 256 bytes): conflict-free for both 32-lane halves.  v_mfma_f32_32x32x16_{f16,bf16}.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Mma16;
template <> struct Mma16<_Float16> {
  static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma16<__bf16> {
  static __device__ __forceinline__ f32x16 run(s16x8 a, s16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

// PSPLIT (TILE = 64 only): instead of giving each wave a quarter of the output tile, every wave owns the WHOLE 64 x 64 tile
// for a quarter of the pixels of each stage (k-step = wave).  Per stage the block then reads each operand fragment from LDS
// once instead of twice (16 KB instead of 32 KB for the same 16 MFMAs): the tile is LDS-read bound, not MFMA bound.  The
// four partial tiles are folded through the (then free) staging buffers: wave w keeps fragment w and parks its other three.
template <typename T, int TILE, bool PSPLIT = false>
__global__ __launch_bounds__(256) void wgrad16_kernel(const WgradArgs a) {
  static_assert(!PSPLIT || TILE == 64, "PSPLIT is the 64-wide variant");
  constexpr int BP = 64;                          // pixels per stage (4 MFMA k-steps of 16)
  constexpr int ROWB = TILE * 2 + 64;             // LDS row stride in bytes
  constexpr int CPR = TILE / 8;                   // 16-byte chunks per tile row
  constexpr int RPP = 256 / CPR;                  // pixel rows per pass
  constexpr int NPASS = BP / RPP;
  constexpr int FI = PSPLIT ? TILE / 32 : TILE / 64;
  constexpr int OPB = BP * ROWB;                  // bytes of one operand tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * OPB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order: hardware deals linear workgroup ids round-robin to the 8 XCDs; give each XCD a contiguous run of
  // (split, tile) pairs so all tiles of one pixel split — which re-read the same dy / x rows — share one L2.
  const int per_xcd = gridDim.x >> 3;
  const int vid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (vid >= a.total) return;
  const int split = vid / a.tiles, tile = vid - split * a.tiles;
  const int tile_k = tile % a.tiles_k, tile_n = tile / a.tiles_k;
  const int n0 = tile_n * TILE, k0 = tile_k * TILE;
  const int m_begin = split * a.pix_per_split;
  const int m_end = min(a.M, m_begin + a.pix_per_split);
  float* const dwp = a.dw + (long)split * a.dw_split_stride;
  if (m_begin >= m_end) return;

  const int chunk = tid % CPR, prow = tid / CPR;
  const int ncol = n0 + chunk * 8;
  const bool n_ok = ncol < a.N;
  const int kk = k0 + chunk * 8;
  const bool k_ok = kk < a.K;
  const int kt = k_ok ? kk / a.C : 0, kc = k_ok ? kk - kt * a.C : 0;
  const int tdy = a.tap_dy[kt], tdx = a.tap_dx[kt];
  const int ohw = a.OH * a.OW;

  // Staging is branch-free: both operands are read through buffer descriptors, so a row past the pixel range, a column
  // past N / K or a tap outside the image simply gets an out-of-range offset (hardware returns zeros) instead of a
  // divergent branch; the pixel coordinates advance by BP per stage with two multiply-shift quotients (no loops).
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  int pb[NPASS], py[NPASS], px[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const int m = m_begin + prow + p * RPP;
    const int mm = m < a.M ? m : 0;
    pb[p] = mm / ohw;
    const int r = mm - pb[p] * ohw;
    py[p] = r / a.OW;
    px[p] = r - py[p] * a.OW;
  }
  const int dy_col = n_ok ? ncol * 2 : -1, x_col = k_ok ? kc * 2 : -1;       // byte offsets of this thread's fixed chunk
  const int dy_rowb = a.dy_ld * 2, x_pixb = a.x_ld * 2;
  uint4 rdy[NPASS], rx[NPASS];
  auto load_stage = [&](int m0) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int m = m0 + prow + p * RPP;
      const bool ok = m < m_end;
      const unsigned od = (ok && dy_col >= 0) ? (unsigned)(m * dy_rowb + dy_col) : OOB;
      const int iy = py[p] * a.sy + tdy, ix = px[p] * a.sx + tdx;
      const bool in = ok && x_col >= 0 && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      const unsigned ox_ = in ? (unsigned)(((pb[p] * a.IH + iy) * a.IW + ix) * x_pixb + x_col) : OOB;
      const u4 vd = __builtin_amdgcn_raw_buffer_load_b128(dr, od, 0, 0);
      const u4 vx = __builtin_amdgcn_raw_buffer_load_b128(xr, ox_, 0, 0);
      rdy[p] = make_uint4(vd[0], vd[1], vd[2], vd[3]);
      rx[p] = make_uint4(vx[0], vx[1], vx[2], vx[3]);
      // advance BP pixels: q rows wrap, then whole images wrap (ranges are tiny: px < OW + BP, py < OH + BP/OW + 1)
      const int nx = px[p] + BP;
      const int q = (int)(((unsigned)nx * a.mag_ow) >> 20);
      px[p] = nx - q * a.OW;
      const int ny = py[p] + q;
      const int q2 = (int)(((unsigned)ny * a.mag_oh) >> 20);
      py[p] = ny - q2 * a.OH;
      pb[p] += q2;
    }
  };
  auto store_stage = [&](int buf) {
    unsigned char* sd = smem + buf * 2 * OPB;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int r = prow + p * RPP;
      *(uint4*)(sd + r * ROWB + chunk * 16) = rdy[p];
      *(uint4*)(sd + OPB + r * ROWB + chunk * 16) = rx[p];
    }
  };

  f32x16 acc[FI][FI];
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int wr = PSPLIT ? 0 : wave >> 1, wc = PSPLIT ? 0 : wave & 1;
  const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
  // per-lane byte offset inside an operand tile for k-step 0, first 4 rows: row = 8*(grp>>1) + q, col = 16*(grp&1) + 4*p4
  const int lane_off = ((grp >> 1) * 8 + q) * ROWB + ((grp & 1) * 16 + p4 * 4) * 2;

  const int nstage = (m_end - m_begin + BP - 1) / BP;
  load_stage(m_begin);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < nstage; ++s) {
    const bool more = s + 1 < nstage;
    if (more) load_stage(m_begin + (s + 1) * BP);
    const unsigned char* sd = smem + (s & 1) * 2 * OPB;
#pragma unroll
    for (int kq = 0; kq < (PSPLIT ? 1 : BP / 16); ++kq) {
      const int ks = PSPLIT ? wave : kq;
      s16x8 fa[FI], fb[FI];
#pragma unroll
      for (int i = 0; i < FI; ++i) {
        const unsigned char* pa = sd + ks * 16 * ROWB + lane_off + (wr * (TILE / 2) + i * 32) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * ROWB));
        fa[i] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < FI; ++j) {
        const unsigned char* pbp = sd + OPB + ks * 16 * ROWB + lane_off + (wc * (TILE / 2) + j * 32) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pbp);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pbp + 4 * ROWB));
        fb[j] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int j = 0; j < FI; ++j) acc[i][j] = Mma16<T>::run(fa[i], fb[j], acc[i][j]);
    }
    if (more) store_stage((s + 1) & 1);
    __syncthreads();
  }
  const int fcol = lane & 31, fh = lane >> 5;
  if constexpr (PSPLIT) {
    // fold the four partial tiles: wave w finishes fragment w = (i, j) = (w >> 1, w & 1); the staging buffers (48 KB, idle
    // after the last barrier) hold the 3 fragments each wave gives away: [src wave][slot][e][lane] floats, 3 * 4 KB per wave
    float* red = (float*)smem;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (f != wave) {
        const int slot = f - (f > wave ? 1 : 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((wave * 3 + slot) * 16 + e) * 64 + lane] = acc[f >> 1][f & 1][e];
      }
    }
    __syncthreads();
    f32x16 tot;
#pragma unroll
    for (int e = 0; e < 16; ++e) tot[e] = 0.f;
#pragma unroll
    for (int f = 0; f < 4; ++f)                               // select this wave's own fragment without dynamic register indexing
      if (f == wave) tot = acc[f >> 1][f & 1];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w != wave) {
        const int slot = wave - (wave > w ? 1 : 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) tot[e] += red[((w * 3 + slot) * 16 + e) * 64 + lane];
      }
    }
    const int fi = wave >> 1, fj = wave & 1;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = n0 + fi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
      const int k = k0 + fj * 32 + fcol;
      if (n < a.N && k < a.K) { if (a.dw_split_stride) dw_out<true>(dwp + (long)n * a.K + k, tot[e]); else dw_out<false>(dwp + (long)n * a.K + k, tot[e]); }
    }
    return;
  }
  if (n0 + TILE <= a.N && k0 + TILE <= a.K) {            // interior tile (the common case): no per-element bounds branches
    float* base = dwp + (long)(n0 + wr * (TILE / 2) + 4 * fh) * a.K + k0 + wc * (TILE / 2) + fcol;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
      for (int j = 0; j < FI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float* p = base + (long)(i * 32 + (e & 3) + 8 * (e >> 2)) * a.K + j * 32;
          if (a.dw_split_stride) dw_out<true>(p, acc[i][j][e]); else dw_out<false>(p, acc[i][j][e]);
        }
    return;
  }
#pragma unroll
  for (int i = 0; i < FI; ++i)
#pragma unroll
    for (int j = 0; j < FI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wr * (TILE / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        const int k = k0 + wc * (TILE / 2) + j * 32 + fcol;
        if (n < a.N && k < a.K) { if (a.dw_split_stride) dw_out<true>(dwp + (long)n * a.K + k, acc[i][j][e]); else dw_out<false>(dwp + (long)n * a.K + k, acc[i][j][e]); }
      }
}


// ------------------------------------------------------------------------------------------------ wide tile (r04)
// wgrad16_kernel<128> is bound by the L2 -> LDS fill (r04 stamps: ~40-55 GB/s per CU whatever the schedule): 64 pixels x (128 + 128)
// columns per stage = 32 MAC per staged byte.  This variant owns a 128-filter x 256-column block of dW with EIGHT waves (2 x 4, 64 x 64
// each, the same fragment code): 64 x (128 + 256) per stage = 42.7 MAC per byte, a third fewer staged bytes per FLOP; 113 KB of LDS
// (two stages), one workgroup per CU.  Staging is the same branch-free register path; the dy and x tiles have different chunk
// geometries (16 / 32 chunks per row), so each thread keeps two (row, chunk) roles.
template <typename T>
__global__ __launch_bounds__(512) void wgrad16w_kernel(const WgradArgs a) {
  constexpr int TN = 128, TK = 256, BP = 64;
  constexpr int ROWN = TN * 2 + 64, ROWK = TK * 2 + 64;      // LDS row strides in bytes (64 mod 128: the transposing reads stay conflict-free)
  constexpr int OPN = BP * ROWN, OPK = BP * ROWK;
  constexpr int NPN = 2, NPK = 4;                            // passes: 512 threads cover 32 dy rows / 16 x rows at a time
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per_xcd = gridDim.x >> 3;
  const int vid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (vid >= a.total) return;
  const int split = vid / a.tiles, tile = vid - split * a.tiles;
  const int tile_k = tile % a.tiles_k, tile_n = tile / a.tiles_k;
  const int n0 = tile_n * TN, k0 = tile_k * TK;
  const int m_begin = split * a.pix_per_split;
  const int m_end = min(a.M, m_begin + a.pix_per_split);
  float* const dwp = a.dw + (long)split * a.dw_split_stride;
  if (m_begin >= m_end) return;

  const int chunk_n = tid & 15, prow_n = tid >> 4;           // dy role
  const int chunk_k = tid & 31, prow_k = tid >> 5;           // x role
  const int ncol = n0 + chunk_n * 8;
  const int kk = k0 + chunk_k * 8;
  const bool k_ok = kk < a.K;
  const int kt = k_ok ? kk / a.C : 0, kc = k_ok ? kk - kt * a.C : 0;
  const int tdy = a.tap_dy[kt], tdx = a.tap_dx[kt];
  const int ohw = a.OH * a.OW;

  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  int pb[NPK], py[NPK], px[NPK];
#pragma unroll
  for (int p = 0; p < NPK; ++p) {
    const int m = m_begin + prow_k + p * 16;
    const int mm = m < a.M ? m : 0;
    pb[p] = mm / ohw;
    const int r = mm - pb[p] * ohw;
    py[p] = r / a.OW;
    px[p] = r - py[p] * a.OW;
  }
  const int dy_col = ncol < a.N ? ncol * 2 : -1, x_col = k_ok ? kc * 2 : -1;
  const int dy_rowb = a.dy_ld * 2, x_pixb = a.x_ld * 2;
  uint4 rdy[NPN], rx[NPK];
  auto load_stage = [&](int m0) {
#pragma unroll
    for (int p = 0; p < NPN; ++p) {
      const int m = m0 + prow_n + p * 32;
      const unsigned od = (m < m_end && dy_col >= 0) ? (unsigned)(m * dy_rowb + dy_col) : OOB;
      const u4 vd = __builtin_amdgcn_raw_buffer_load_b128(dr, od, 0, 0);
      rdy[p] = make_uint4(vd[0], vd[1], vd[2], vd[3]);
    }
#pragma unroll
    for (int p = 0; p < NPK; ++p) {
      const int m = m0 + prow_k + p * 16;
      const int iy = py[p] * a.sy + tdy, ix = px[p] * a.sx + tdx;
      const bool in = m < m_end && x_col >= 0 && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      const unsigned ox_ = in ? (unsigned)(((pb[p] * a.IH + iy) * a.IW + ix) * x_pixb + x_col) : OOB;
      const u4 vx = __builtin_amdgcn_raw_buffer_load_b128(xr, ox_, 0, 0);
      rx[p] = make_uint4(vx[0], vx[1], vx[2], vx[3]);
      const int nx = px[p] + BP;
      const int q = (int)(((unsigned)nx * a.mag_ow) >> 20);
      px[p] = nx - q * a.OW;
      const int ny = py[p] + q;
      const int q2 = (int)(((unsigned)ny * a.mag_oh) >> 20);
      py[p] = ny - q2 * a.OH;
      pb[p] += q2;
    }
  };
  auto store_stage = [&](int buf) {
    unsigned char* sd = smem_w + buf * (OPN + OPK);
#pragma unroll
    for (int p = 0; p < NPN; ++p) *(uint4*)(sd + (prow_n + p * 32) * ROWN + chunk_n * 16) = rdy[p];
#pragma unroll
    for (int p = 0; p < NPK; ++p) *(uint4*)(sd + OPN + (prow_k + p * 16) * ROWK + chunk_k * 16) = rx[p];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int wr = wave >> 2, wc = wave & 3;
  const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
  const int row_in = (grp >> 1) * 8 + q, col_in = ((grp & 1) * 16 + p4 * 4) * 2;
  const int off_n = row_in * ROWN + col_in + wr * 128, off_k = row_in * ROWK + col_in + wc * 128;

  const int nstage = (m_end - m_begin + BP - 1) / BP;
  load_stage(m_begin);
  store_stage(0);
  __syncthreads();
  for (int s = 0; s < nstage; ++s) {
    const bool more = s + 1 < nstage;
    if (more) load_stage(m_begin + (s + 1) * BP);
    const unsigned char* sd = smem_w + (s & 1) * (OPN + OPK);
#pragma unroll
    for (int ks = 0; ks < BP / 16; ++ks) {
      s16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned char* pa = sd + ks * 16 * ROWN + off_n + i * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * ROWN));
        fa[i] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned char* pbp = sd + OPN + ks * 16 * ROWK + off_k + j * 64;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pbp);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pbp + 4 * ROWK));
        fb[j] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Mma16<T>::run(fa[i], fb[j], acc[i][j]);
    }
    if (more) store_stage((s + 1) & 1);
    __syncthreads();
  }
  const int fcol = lane & 31, fh = lane >> 5;
  const bool interior = n0 + TN <= a.N && k0 + TK <= a.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int n = n0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        const int k = k0 + wc * 64 + j * 32 + fcol;
        if (interior || (n < a.N && k < a.K)) { if (a.dw_split_stride) dw_out<true>(dwp + (long)n * a.K + k, acc[i][j][e]); else dw_out<false>(dwp + (long)n * a.K + k, acc[i][j][e]); }
      }
}

// ------------------------------------------------------------------------------------------------ 3x3 stride-1: patch kernel (r03)
// The GEMM kernels above fetch an input pixel once per TAP (the nine (tap, channel) column tiles are different workgroups or
// different stages): 32-64 FLOP per byte through the CU's 64 B/clk address path, which is what bounds them.  Here a workgroup
// owns a 64-channel x 64-filter block of dW for ALL nine taps: a stage is 80 output pixels (one row of an 80-wide map, two rows
// of a 40-wide, four of a 20-wide one) and its (rows + 2) x (width + 2) input patch, staged ONCE as [patch pixel][channel] in LDS;
// the nine taps are nine row offsets into that patch for the transposing fragment read.  142 FLOP per staged byte.
// Waves 2 x 2 over (filter block, channel block) of 32 x 32, nine accumulator tiles each (144 registers).
// One tap ROW (ty) per workgroup: the patch is the stage's own rows shifted by ty - 1, two columns wider.  (All nine taps per
// workgroup — 142 FLOP per staged byte — was measured first and lost: 147 KB of f32 atomics per workgroup, 38 MB per layer at one
// workgroup per CU; a tap row has a third of that per workgroup and three times the workgroups.)  SP = pixels per stage;
// CB x NB = channels x filters per workgroup (32 or 64 each; partial blocks are zero-filled on the way in and masked on the way
// out).  The (NB/32) x (CB/32) blocks of 32 x 32 go to the four waves; with fewer than four blocks the waves of a block share its
// k-steps (pixels) and each adds its partial sums.
// (Nine taps per workgroup was also tried again for the <= 32-channel layers, whose dW is only 18-74 KB: 99-115 us against 68.)
template <typename T, int WS, int SP, int CB, int NB>
__global__ __launch_bounds__(256, SP == 80 ? 4 : 2) void wgrad3x3p_kernel(const WgradArgs a) {
  constexpr int NTY = 1, RS = SP / WS, PW = WS + 2, NPX = (RS + NTY - 1) * PW, NT = 3 * NTY;
  constexpr int ROWX = CB * 2 + (CB == 64 ? 64 : 0), ROWD = NB * 2 + (NB == 64 ? 64 : 0);    // row strides = 64 mod 256 bytes
  constexpr int XB = NPX * ROWX, DB = SP * ROWD;
  constexpr int CCH = CB / 8, NCH = NB / 8;                        // 16-byte chunks per pixel
  constexpr int XCH = (NPX * CCH + 255) / 256, DCH = (SP * NCH + 255) / 256;
  constexpr int NBLK = (CB / 32) * (NB / 32), KH = 4 / NBLK, KS = SP / 16, KPW = (KS + KH - 1) / KH;
  constexpr int FOLD = KH > 1 ? NBLK * 3 * 16 * 64 * 4 : 0;        // epilogue: the partial tiles of the waves that share a block
  __shared__ __attribute__((aligned(16))) unsigned char smem[(XB + DB) > FOLD ? (XB + DB) : FOLD];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per_xcd = gridDim.x >> 3;
  const int vid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (vid >= a.total) return;
  const int split = vid / a.tiles, tile = vid - split * a.tiles;
  // tile = ((tn * tiles_k) + tc) * 3 + ty: the tap rows of one block are neighbours (they read the same dy rows)
  constexpr int TYT = 3 / NTY;
  const int ty = tile % TYT, tcn = tile / TYT;
  const int tc = tcn % a.tiles_k, tn = tcn / a.tiles_k;
  const int c0 = tc * CB, n0 = tn * NB;
  const int sps = a.pix_per_split / SP, stages = a.M / SP;
  const int s_begin = split * sps, s_end = min(stages, s_begin + sps);
  float* const dwp = a.dw + (long)split * a.dw_split_stride;
  if (s_begin >= s_end) return;
  const int SX = a.OW / WS, spi = (a.OH / RS) * SX;            // stages per row band, per image
  const int row0 = NTY == 3 ? -1 : ty - 1;                     // first patch row relative to the stage's first output row

  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, a.dy_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  const int x_pixb = a.x_ld * 2, dy_rowb = a.dy_ld * 2;
  // this thread's 16-byte chunks of a stage: fixed patch position / stage pixel, only the stage origin moves
  int xconst[XCH], xrc[XCH], dconst[DCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    const int cidx = tid + 256 * i, px = cidx / CCH, ch = cidx - px * CCH;
    const int pr = px / PW, pc = px - pr * PW;
    xconst[i] = ((pr + row0) * a.IW + (pc - 1)) * x_pixb + (c0 + ch * 8) * 2;
    xrc[i] = (px < NPX && c0 + ch * 8 < a.C) ? (pr | (pc << 8)) : -1;
  }
#pragma unroll
  for (int i = 0; i < DCH; ++i) {
    const int cidx = tid + 256 * i, p = cidx / NCH, ch = cidx - p * NCH;
    const int oyl = p / WS, oxl = p - oyl * WS;
    dconst[i] = (p < SP && n0 + ch * 8 < a.N) ? (oyl * a.OW + oxl) * dy_rowb + (n0 + ch * 8) * 2 : -1;
  }
  uint4 rx[XCH], rd[DCH];
  auto load_stage = [&](int g) {
    const int b = g / spi, j = g - b * spi, jy = j / SX;
    const int oy0 = jy * RS, ox0 = (j - jy * SX) * WS;
    const int xbase = ((b * a.IH + oy0) * a.IW + ox0) * x_pixb;                  // host: the x / dy views are < 2^31 bytes
    const int dbase = ((b * a.OH + oy0) * a.OW + ox0) * dy_rowb;
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int iy = oy0 + row0 + (xrc[i] & 255), ix = ox0 - 1 + (xrc[i] >> 8);
      const bool in = xrc[i] >= 0 && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      const u4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, in ? (unsigned)(xbase + xconst[i]) : OOB, 0, 0);
      rx[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
#pragma unroll
    for (int i = 0; i < DCH; ++i) {
      const u4 v = __builtin_amdgcn_raw_buffer_load_b128(dr, dconst[i] >= 0 ? (unsigned)(dbase + dconst[i]) : OOB, 0, 0);
      rd[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int cidx = tid + 256 * i, px = cidx / CCH, ch = cidx - px * CCH;
      if (cidx < NPX * CCH) *(uint4*)(smem + px * ROWX + ch * 16) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < DCH; ++i) {
      const int cidx = tid + 256 * i, p = cidx / NCH, ch = cidx - p * NCH;
      if (cidx < SP * NCH) *(uint4*)(smem + XB + p * ROWD + ch * 16) = rd[i];
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const int blk = wave % NBLK, kh = wave / NBLK;                // this wave's 32 x 32 block and its share of the k-steps
  const int wr = blk % (NB / 32), wc = blk / (NB / 32);         // filter block (rows of dW), channel block
  const int grp = lane >> 4, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
  const int colb = ((grp & 1) * 16 + p4 * 4) * 2;
  // fragment rows of this lane, per k-step of this wave: dy pixel row / patch row at tx = 0, pixels ks*16 + 8*(grp>>1) + q (+4)
  int xoff[KPW][2];
#pragma unroll
  for (int k = 0; k < KPW; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = min((k * KH + kh) * 16, SP - 16) + (grp >> 1) * 8 + q + 4 * h;
      const int oyl = p / WS, oxl = p - oyl * WS;
      xoff[k][h] = (oyl * PW + oxl) * ROWX + colb + wc * 64;
    }
  const int dlane = XB + ((grp >> 1) * 8 + q) * ROWD + colb + wr * 64;

  load_stage(s_begin);
  for (int s = s_begin; s < s_end; ++s) {
    if (s > s_begin) __syncthreads();                 // every wave is done reading the previous stage
    store_stage();
    __syncthreads();
    if (s + 1 < s_end) load_stage(s + 1);             // in flight while this stage is multiplied
#pragma unroll
    for (int k = 0; k < KPW; ++k) {
      const int ks = k * KH + kh;
      if (KS % KH != 0 && ks >= KS) break;            // (wave-uniform) the last round of k-steps is not full
      s16x8 fa;
      {
        const unsigned char* pa = smem + dlane + ks * 16 * ROWD;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)pa);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(pa + 4 * ROWD));
        fa = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tapc = ((t / 3) * PW + (t % 3)) * ROWX;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + xoff[k][0] + tapc));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + xoff[k][1] + tapc));
        const s16x8 fb = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        acc[t] = Mma16<T>::run(fa, fb, acc[t]);
      }
    }
  }
  if constexpr (KH > 1) {
    // the waves of one block fold their partial tiles in wave order through the (now idle) staging memory: one set of atomics per
    // block instead of KH, and a fixed summation order inside the workgroup (the ordered mode relies on it)
    float* fold = (float*)smem + blk * (3 * 16 * 64);
    for (int r = 1; r < KH; ++r) {
      __syncthreads();
      if (kh == r) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) fold[(t * 16 + e) * 64 + lane] = acc[t][e];
      }
      __syncthreads();
      if (kh == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[t][e] += fold[(t * 16 + e) * 64 + lane];
      }
    }
    if (kh != 0) return;
  }
  // dW[n][t][c]: D rows = filters (dy), columns = channels (x)
  const int fcol = lane & 31, fh = lane >> 5;
  const int cc = c0 + wc * 32 + fcol, nr = n0 + wr * 32 + 4 * fh;
  float* base = dwp + (long)nr * a.K + (NTY == 3 ? 0 : ty * 3) * a.C + cc;
  if (cc < a.C) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int dn = (e & 3) + 8 * (e >> 2);
        if (nr + dn < a.N) { if (a.dw_split_stride) dw_out<true>(base + (long)dn * a.K + t * a.C, acc[t][e]); else dw_out<false>(base + (long)dn * a.K + t * a.C, acc[t][e]); }
      }
  }
}

// shapes the patch kernel takes: 3x3, stride 1, padding 1, dilation 1 (taps in row-major order), 64-channel / 64-filter blocks,
// output width 20, 40 or a multiple of 80, whole stages per image
static int wgrad3x3p_ws(const WgradArgs& a) {
  if (a.T != 9 || a.sy != 1 || a.sx != 1 || a.C % 8 || a.N % 8 || a.IH != a.OH || a.IW != a.OW) return 0;
  for (int t = 0; t < 9; ++t)
    if (a.tap_dy[t] != t / 3 - 1 || a.tap_dx[t] != t % 3 - 1) return 0;
  const int ws = a.OW % 80 == 0 ? 80 : (a.OW == 40 ? 40 : (a.OW == 20 ? 20 : 0));
  if (!ws || a.OH % (80 / ws)) return 0;
  return ws;
}
// pixels per stage: 160 where the map has whole 160-pixel stages (two barriers per 30 MFMAs instead of per 15), else 80
static int wgrad3x3p_sp(const WgradArgs& a, int ws) { return (ws >= 40 && a.OH % (160 / ws) == 0) ? 160 : 80; }

static int cu_count() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return n;
}

// cfg = 4 * kind + r.  kind 0: 64-wide tile, 1: 128-wide tile, 2: 64-wide pixel-split variant (16-bit only); r = 0..3:
// a quarter, half, one, two rounds of resident workgroups (every workgroup ends with tile x tile f32 atomics and the chip
// adds only ~1.3 TB/s of them: one round of 128-wide tiles is 33.5 MB = 26 us, so small layers want FEWER workgroups).  A "round" = CUs x workgroups that fit one CU (LDS: 2 for
// the 128-wide tile, 3 for the 64-wide); the pixel-split count is rounded DOWN so the grid never spills a nearly empty
// extra round (1048 workgroups on 512 slots ran three rounds for two rounds of work).
// kind 3 (16-bit, 3x3 stride 1 shapes of wgrad3x3p_ws): the patch kernel, one tap row per workgroup, r rounds of resident workgroups
// kind 4 (16-bit): the wide tile, 128 filters x 256 columns, eight waves, one workgroup per CU
constexpr int WGRAD_NCFG = SY11_WGRAD_NCFG;
static_assert(WGRAD_NCFG == 20, "configuration table and its size (tune.h) out of step");
constexpr int WGRAD_WIDE_LDS = 2 * 64 * ((128 * 2 + 64) + (256 * 2 + 64));
static int wgrad_launch_cfg(WgradArgs a, int dtype, hipStream_t st, int cfg) {
  const int kind = cfg >> 2;
  const bool psplit = kind == 2;
  const bool patch = kind == 3;
  const bool wide = kind == 4;
  if (wide && dtype == SY11_F32) SY11_FAIL(SY11_EUNSUPPORTED, "conv2d_wgrad: configuration %d is 16-bit only", cfg);
  const int patch_ws = patch ? wgrad3x3p_ws(a) : 0;
  if (patch && (!patch_ws || dtype == SY11_F32)) SY11_FAIL(SY11_EUNSUPPORTED, "conv2d_wgrad: configuration %d does not take this shape", cfg);
  const int patch_sp = patch ? wgrad3x3p_sp(a, patch_ws) : 0;
  const int tile = kind == 1 ? 128 : 64;
  const int slots = cu_count() * (wide ? 1 : patch ? (patch_sp == 80 ? 4 : 2) : (tile == 128 ? 2 : 3));
  const int target_wg = (slots << (cfg & 3)) >> 2;
  const int pcb = a.C > 32 ? 64 : 32, pnb = a.N > 32 ? 64 : 32;          // patch kernel: channels x filters per workgroup
  a.tiles_k = patch ? cdiv(a.C, pcb) : cdiv(a.K, wide ? 256 : tile);
  const int tiles = patch ? a.tiles_k * cdiv(a.N, pnb) * 3 : a.tiles_k * cdiv(a.N, wide ? 128 : tile);
  int splits = target_wg / tiles;
  const int stage_px = patch ? patch_sp : 64;
  const int max_splits = cdiv(a.M, 8 * stage_px);      // >= 8 stages per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.pix_per_split = cdiv(cdiv(a.M, splits), stage_px) * stage_px;
  splits = cdiv(a.M, a.pix_per_split);
  a.tiles = tiles; a.total = tiles * splits;
  // ordered mode: one partial dW per pixel split, folded in split order afterwards (a single split adds straight into dW: one
  // add per element onto whatever the accumulation window already holds — exact)
  DetPartials dp;
  float* const dw_out = a.dw;
  const bool det = sy11_det(8) && splits > 1;
  if (det) {
    // every element of a split's partial dW has exactly one writer (plain stores in the kernels): no zero fill
    if (!dp.acquire(st, 1, splits, a.N * a.K, false)) SY11_FAIL(SY11_ELAUNCH, "conv2d_wgrad: ordered-reduction workspace unavailable (%d x %d floats)", splits, a.N * a.K);
    a.dw = dp.buf(0);
    a.dw_split_stride = (long)a.N * a.K;
  }
  a.mag_ow = (unsigned)(((1u << 20) + a.OW - 1) / a.OW);
  a.mag_oh = (unsigned)(((1u << 20) + a.OH - 1) / a.OH);
  dim3 grid(cdiv(a.total, 8) * 8), block(256);
  if (wide) {
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute((const void*)wgrad16w_kernel<_Float16>, hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_WIDE_LDS) != hipSuccess ||
          hipFuncSetAttribute((const void*)wgrad16w_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_WIDE_LDS) != hipSuccess)
        SY11_FAIL(SY11_ELAUNCH, "conv2d_wgrad: cannot raise the LDS limit to %d bytes", WGRAD_WIDE_LDS);
      attr_set = true;
    }
    if (dtype == SY11_F16) hipLaunchKernelGGL((wgrad16w_kernel<_Float16>), grid, dim3(512), WGRAD_WIDE_LDS, st, a);
    else hipLaunchKernelGGL((wgrad16w_kernel<__bf16>), grid, dim3(512), WGRAD_WIDE_LDS, st, a);
  } else if (dtype == SY11_F32) {
    if (tile == 128) hipLaunchKernelGGL((wgrad_kernel<float, 128>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wgrad_kernel<float, 64>), grid, block, 0, st, a);
  } else if (patch) {
#define SY11_WG3B(TT, CBB, NBB)                                                                                       \
    do {                                                                                                              \
      if (patch_ws == 80 && patch_sp == 160) hipLaunchKernelGGL((wgrad3x3p_kernel<TT, 80, 160, CBB, NBB>), grid, block, 0, st, a); \
      else if (patch_ws == 80) hipLaunchKernelGGL((wgrad3x3p_kernel<TT, 80, 80, CBB, NBB>), grid, block, 0, st, a);   \
      else if (patch_ws == 40 && patch_sp == 160) hipLaunchKernelGGL((wgrad3x3p_kernel<TT, 40, 160, CBB, NBB>), grid, block, 0, st, a); \
      else if (patch_ws == 40) hipLaunchKernelGGL((wgrad3x3p_kernel<TT, 40, 80, CBB, NBB>), grid, block, 0, st, a);   \
      else hipLaunchKernelGGL((wgrad3x3p_kernel<TT, 20, 80, CBB, NBB>), grid, block, 0, st, a);                       \
    } while (0)
#define SY11_WG3(TT)                                                   \
    do {                                                               \
      if (pcb == 64 && pnb == 64) SY11_WG3B(TT, 64, 64);               \
      else if (pcb == 64) SY11_WG3B(TT, 64, 32);                       \
      else if (pnb == 64) SY11_WG3B(TT, 32, 64);                       \
      else SY11_WG3B(TT, 32, 32);                                      \
    } while (0)
    if (dtype == SY11_F16) SY11_WG3(_Float16); else SY11_WG3(__bf16);
#undef SY11_WG3
#undef SY11_WG3B
  } else if (dtype == SY11_F16) {
    if (psplit) hipLaunchKernelGGL((wgrad16_kernel<_Float16, 64, true>), grid, block, 0, st, a);
    else if (tile == 128) hipLaunchKernelGGL((wgrad16_kernel<_Float16, 128>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wgrad16_kernel<_Float16, 64>), grid, block, 0, st, a);
  } else {
    if (psplit) hipLaunchKernelGGL((wgrad16_kernel<__bf16, 64, true>), grid, block, 0, st, a);
    else if (tile == 128) hipLaunchKernelGGL((wgrad16_kernel<__bf16, 128>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((wgrad16_kernel<__bf16, 64>), grid, block, 0, st, a);
  }
  SY11_LAUNCH_CHECK("conv2d_wgrad");
  if (det) return dp.fold(0, dw_out);
  return SY11_OK;
}

extern "C" int sy11_conv2d_wgrad_dw(const sy11_conv_desc* d, const void* x, const void* dy, int dy_ld, float* dw, hipStream_t st);
extern "C" int sy11_conv2d_wgrad(const sy11_conv_desc* d, const void* x, const void* dy, int32_t dy_ld, float* dw,
                                 void* stream) {
  SY11_REQUIRE(d && x && dy && dw, "conv2d_wgrad: null argument");
  SY11_REQUIRE(dtype_ok(d->dtype), "conv2d_wgrad: bad dtype");
  hipStream_t st = (hipStream_t)stream;
  if (d->groups != 1 && d->groups == d->C && d->C == d->N) return sy11_conv2d_wgrad_dw(d, x, dy, dy_ld, dw, st);
  if (d->groups != 1) {                 // grouped: independent dense problems on channel slices; dw = [N][KH*KW][C/g]
    SY11_REQUIRE(d->groups > 1 && d->C % d->groups == 0 && d->N % d->groups == 0, "conv2d_wgrad: channels not divisible by groups=%d", d->groups);
    const int cg = d->C / d->groups, ng = d->N / d->groups, es = dtype_size(d->dtype);
    sy11_conv_desc dg = *d;
    dg.groups = 1; dg.C = cg; dg.N = ng;
    for (int g = 0; g < d->groups; ++g) {
      const int rc = sy11_conv2d_wgrad(&dg, (const char*)x + (long)g * cg * es, (const char*)dy + (long)g * ng * es, dy_ld,
                                       dw + (long)g * ng * d->KH * d->KW * cg, stream);
      if (rc) return rc;
    }
    return SY11_OK;
  }
  const int esz = dtype_size(d->dtype), epc = 16 / esz;
  SY11_REQUIRE(d->KH * d->KW <= 64 && d->KH > 0 && d->KW > 0, "conv2d_wgrad: <=64 taps");
  SY11_REQUIRE(d->C % epc == 0 && d->N % epc == 0, "conv2d_wgrad: C and N must be multiples of %d", epc);
  SY11_REQUIRE(d->x_ld % epc == 0 && dy_ld % epc == 0 && d->x_ld >= d->C && dy_ld >= d->N, "conv2d_wgrad: bad pixel strides");
  SY11_REQUIRE((((uintptr_t)x | (uintptr_t)dy) & 15) == 0 && ((uintptr_t)dw & 3) == 0, "conv2d_wgrad: misaligned pointer");
  SY11_REQUIRE((long)d->B * d->IH * d->IW < (1L << 31) && (long)d->B * d->OH * d->OW < (1L << 31), "conv2d_wgrad: pixel count overflows int32");
  WgradArgs a{};
  a.x = x; a.dy = dy; a.dw = dw;
  {
    const long xb = ((long)d->B * d->IH * d->IW - 1) * d->x_ld * esz + (long)d->C * esz;
    const long db = ((long)d->B * d->OH * d->OW - 1) * dy_ld * esz + (long)d->N * esz;
    SY11_REQUIRE(xb < (1L << 31) && db < (1L << 31), "conv2d_wgrad: operand view larger than 2 GiB");
    SY11_REQUIRE((long)(d->OW + 64) * d->OW < (1L << 20) && (long)(d->OH + 64) * d->OH < (1L << 20), "conv2d_wgrad: output map larger than 960 pixels per side");
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)db;
  }
  a.T = d->KH * d->KW; a.C = d->C; a.K = a.T * a.C; a.N = d->N;
  a.M = d->B * d->OH * d->OW;
  a.x_ld = d->x_ld; a.dy_ld = dy_ld;
  a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW; a.sy = d->SH; a.sx = d->SW;
  for (int r = 0; r < d->KH; ++r)
    for (int s = 0; s < d->KW; ++s) {
      a.tap_dy[r * d->KW + s] = (signed char)(r * d->DH - d->PH);
      a.tap_dx[r * d->KW + s] = (signed char)(s * d->DW - d->PW);
    }
  // static heuristic (r01 sweeps): 128-wide tiles unless the map is small, one round of workgroups, >= 512 pixels per split
  int cfg = ((a.N > 64 && a.K > 64 && a.M > 30000) ? 4 : 0) + 2;
  // 3x3 stride 1 on the large maps: the patch kernel (r03 probe: it wins every such layer at >= 80x80 x 64 images, never at 20x20)
  if (d->dtype != SY11_F32 && a.M >= 200000 && wgrad3x3p_ws(a)) cfg = 14;
  const int forced = sy11_opt(OPT_WGRAD_CFG);
  const int ncfg = d->dtype == SY11_F32 ? 8 : WGRAD_NCFG;
  auto patch_ok = [&](int c) { return c < 12 || (c < 16 ? wgrad3x3p_ws(a) != 0 : (a.N > 64 && a.K > 128)); };
  if (forced >= 0 && forced < ncfg && patch_ok(forced)) return wgrad_launch_cfg(a, d->dtype, st, forced);
  {                                  // a recorded / imported pick is honoured even with measuring off
    sy11tune::Cache& cache = sy11tune::cache(1);
    static float* scratch = nullptr;
    static size_t scratch_elems = 0;
    const int key[] = {d->dtype, a.M, a.N, a.K, a.C, a.sy, a.sx, a.IW, a.OW, a.x_ld, a.dy_ld};
    const uint64_t h = sy11tune::hash(key, (int)(sizeof(key) / sizeof(int)));
    int hit;
    if (cache.get(h, &hit)) {
      if (hit >= 0 && hit < ncfg && patch_ok(hit)) cfg = hit;      // an imported record from another build / a corrupt file: keep the heuristic
    } else if (sy11tune::enabled() && !sy11tune::capturing(st)) {
      const size_t need = (size_t)a.N * a.K;
      if (need > scratch_elems) {                       // candidates accumulate with atomics: measure into a scratch dW
        if (scratch) (void)hipFree(scratch);
        scratch = nullptr; scratch_elems = 0;
        if (hipMalloc((void**)&scratch, need * sizeof(float)) == hipSuccess) scratch_elems = need; else (void)hipGetLastError();
      }
      if (scratch) {
        WgradArgs t = a;
        t.dw = scratch;
        int cands[WGRAD_NCFG], nc = 0;
        for (int c = 0; c < ncfg; ++c)
          if (patch_ok(c)) cands[nc++] = c;
        const int best = sy11tune::pick(cands, nc, [&](int c) { return wgrad_launch_cfg(t, d->dtype, st, c); }, st, "wgrad", key,
                                        (int)(sizeof(key) / sizeof(int)));
        if (best >= 0) { cache.put(h, best); cfg = best; }
      }
    }
  }
  return wgrad_launch_cfg(a, d->dtype, st, cfg);
}

