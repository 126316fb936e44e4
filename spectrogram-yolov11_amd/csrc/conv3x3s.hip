// conv3x3s.hip — 3x3 stride-1 convolution with FEW channels (C = 16 / 32, N <= 32) on MFMA for gfx950: forward and
// (through the same argument block, with the transposed filter and flipped taps) the stride-1 input gradient.
//
// These layers (160x160 16<->32 of yolo11s) are streaming problems: 4.6 K filter values against 1.6 M pixels.
// The GEMM kernels stage filter rows and pixel rows per (tap, channel slab) and run them at 3x their byte floor (r03 sweep: 81-89 us
// for 157 MB).  Here the WHOLE filter of a 32-filter block sits in a wave's registers as MFMA A operands (9 taps x C/16 k-steps),
// a workgroup walks 4 x 64-pixel output tiles (persistent, next tile's rows in flight), the 6 x 66-pixel input patch of a tile is
// staged ONCE as [pixel][channel] in LDS with the zero padding in place, and a B operand (8 channels of one pixel at one tap) is one
// ds_read_b128.  The MFMA computes the transposed tile (rows = filters, columns = pixels), so a lane holds 4 x 4 consecutive
// filters of one pixel: 8-byte writes into the staging buffer, 16-byte coalesced stores out, BN statistics as per-register partial
// sums folded across lanes once per launch (the layout of direct.hip's stem_fwd_tile).
#include "det.h"
#include "igemm_args.h"
#include <type_traits>


struct SmallCTiling { int tiles_x, tiles_y, tiles; unsigned mag_tx, mag_ty; };

template <typename T> struct ScMma;
template <> struct ScMma<_Float16> {
  static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct ScMma<__bf16> {
  static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

// C: input channels (16, 32, 64); NFB: 32-filter blocks per workgroup (1: N <= 32, 2: N <= 64); SC_TW: tile width (64 x 4 or 32 x 8
// output pixels: 32 where the map width is a multiple of 32 but not of 64 — a 160-wide map would leave every third 64-wide tile half empty)
template <typename T, int C, int NFB, int SC_TW>
__global__ __launch_bounds__(256, 2) void smallc3x3_kernel(const IgemmArgs a, const SmallCTiling g) {
  constexpr int SC_TH = 256 / SC_TW, SC_PH = SC_TH + 2, SC_PW = SC_TW + 2, GPR = SC_TW / 32;
  constexpr int PIXB = C * 2, CCH = C / 8, NPX = SC_PH * SC_PW, XT = NPX * PIXB;
  constexpr int XCH = (NPX * CCH + 255) / 256;
  constexpr int KS = 9 * C / 16, CS = C / 16;
  constexpr int NW = NFB * 32, ROWB = NW * 2, ROWS = ROWB + 16;      // staging row of one pixel (+16: 8-byte writes of 32 lanes off one bank group)
  constexpr int GPW = 2 * NFB;                                         // 32-pixel groups per wave and tile
  constexpr unsigned OOB = 0x80000000u;
  constexpr bool PREFETCH = C <= 32;                                   // next tile's rows in registers while this one is multiplied (C = 64: 52 registers too many)
  __shared__ __attribute__((aligned(16))) unsigned char xt[XT];
  __shared__ __attribute__((aligned(16))) unsigned char stage[256 * ROWS];
  __shared__ float red[4][2][32];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  const int fb = wave % NFB, gset = wave / NFB;
  const bool has_stats = a.stat_sum != nullptr, silu = a.flags & SY11_EPI_SILU, accum = a.flags & SY11_EPI_ACCUM;
  const bool plain = !silu && !a.bias;

  // filter fragments of this wave's block: row = filter fb*32 + col, k-step (tap, 16-channel slice), 8 channels per lane half
  uint4 wf[KS];
  {
    const int n = fb * 32 + col;
    const T* wrow = (const T*)a.w + (long)min(n, a.N - 1) * a.wK;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int cs = 0; cs < CS; ++cs) {
        const uint4 v = *(const uint4*)(wrow + a.tap_w[t] * a.C + cs * 16 + 8 * half);
        wf[t * CS + cs] = n < a.N ? v : make_uint4(0, 0, 0, 0);
      }
  }
  int tapoff[9];                                                       // byte offset of tap t inside the patch (workgroup-uniform)
#pragma unroll
  for (int t = 0; t < 9; ++t) tapoff[t] = ((1 + a.tap_dy[t]) * SC_PW + 1 + a.tap_dx[t]) * PIXB;

  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, a.x_bytes, 0x00020000);
  const int x_pixb = a.x_ld * 2;
  // this thread's 16-byte chunks of a patch: fixed position, only the tile origin moves
  int xconst[XCH], xrc[XCH];
#pragma unroll
  for (int i = 0; i < XCH; ++i) {
    const int cidx = tid + 256 * i, px = cidx / CCH, ch = cidx - px * CCH;
    const int pr = px / SC_PW, pc = px - pr * SC_PW;
    xconst[i] = ((pr - 1) * a.IW + (pc - 1)) * x_pixb + ch * 16;
    xrc[i] = px < NPX ? (pr | (pc << 8)) : -1;
  }
  auto tile_at = [&](int t, int& b, int& oy0, int& ox0) {
    const int q = g.tiles_x == 1 ? t : (int)__umulhi((unsigned)t, g.mag_tx), tx = t - q * g.tiles_x;
    b = g.tiles_y == 1 ? q : (int)__umulhi((unsigned)q, g.mag_ty);
    oy0 = (q - b * g.tiles_y) * SC_TH;
    ox0 = tx * SC_TW;
  };
  uint4 rx[XCH];
  auto load_tile = [&](int b, int oy0, int ox0) {
    const int xbase = ((b * a.IH + oy0) * a.IW + ox0) * x_pixb;      // host: the x view is < 2^31 bytes
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int iy = oy0 - 1 + (xrc[i] & 255), ix = ox0 - 1 + (xrc[i] >> 8);
      const bool in = xrc[i] >= 0 && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      const u4 v = __builtin_amdgcn_raw_buffer_load_b128(xr, in ? (unsigned)(xbase + xconst[i]) : OOB, 0, 0);
      rx[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
      const int cidx = tid + 256 * i;
      if (cidx < NPX * CCH) *(uint4*)(xt + cidx * 16) = rx[i];         // [pixel][channel]: chunk index = byte offset / 16
    }
  };

  float s1[16], s2[16];                  // register e: filter fb*32 + (e&3) + 8*(e>>2) + 4*half
#pragma unroll
  for (int e = 0; e < 16; ++e) s1[e] = s2[e] = 0.f;
  const int ncpr = a.N / 8;                                            // 16-byte chunks per output pixel
  int t = blockIdx.x, b, oy0, ox0;
  tile_at(min(t, g.tiles - 1), b, oy0, ox0);
  if (PREFETCH && t < g.tiles) load_tile(b, oy0, ox0);
  for (; t < g.tiles; t += gridDim.x) {
    __syncthreads();                                  // previous tile: patch read, staging buffer drained
    if (!PREFETCH) load_tile(b, oy0, ox0);
    store_tile();
    const int cb = b, coy = oy0, cox = ox0;
    if (t + (int)gridDim.x < g.tiles) {
      tile_at(t + gridDim.x, b, oy0, ox0);
      if (PREFETCH) load_tile(b, oy0, ox0);
    }
    __syncthreads();
#pragma unroll 1
    for (int gi = 0; gi < GPW; ++gi) {                                 // one group at a time: unrolled, the groups' fragments and accumulators pile up
      const int gq = gset * GPW + gi;                                  // 32-pixel group, row-major inside the tile
      const int prow = gq / GPR, pcol = (gq % GPR) * 32 + col;
      const bool ok = coy + prow < a.OH && cox + pcol < a.OW;
      const unsigned char* px = xt + (prow * SC_PW + pcol) * PIXB + half * 16;
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int tt = 0; tt < 9; ++tt)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
          const uint4 xf = *(const uint4*)(px + tapoff[tt] + cs * 32);
          acc = ScMma<T>::run(wf[tt * CS + cs], xf, acc);
        }
      unsigned char* srow = stage + (gq * 32 + col) * ROWS + fb * 64 + half * 8;
      typedef T t4 __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float v0 = ok ? acc[e] : 0.f;
        acc[e] = v0;
        s1[e] += v0;
        s2[e] += v0 * v0;
      }
      if (plain) {                                    // training: statistics only — no branch inside the unrolled element loops
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          t4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = ElemTraits<T>::from_f(acc[e4 * 4 + i]);
          *(t4*)(srow + e4 * 16) = o;                 // filters fb*32 + 8*e4 + 4*half + (0..3)
        }
      } else {                                        // inference: bias (from memory, not from 16 more registers) and SiLU
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          t4 o;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int e = e4 * 4 + i, n = fb * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
            float v = acc[e];
            if (a.bias) v += a.bias[min(n, a.N - 1)];
            if (silu) v = silu_f(v);
            o[i] = ElemTraits<T>::from_f(v);
          }
          *(t4*)(srow + e4 * 16) = o;
        }
      }
    }
    __syncthreads();
    {
      T* yimg = (T*)a.y + (long)cb * a.OH * a.OW * a.y_ld;
      for (int id = tid; id < 256 * ncpr; id += 256) {
        const int p = id / ncpr, cc = id - p * ncpr;
        const int oy = coy + p / SC_TW, ox = cox + p % SC_TW;
        if (oy < a.OH && ox < a.OW && a.debug != 5) {
          uint4 v = *(const uint4*)(stage + p * ROWS + cc * 16);
          T* gp = yimg + ((long)oy * a.OW + ox) * a.y_ld + cc * 8;
          if (accum) {
            typedef T vt8 __attribute__((ext_vector_type(8)));
            vt8 xv = __builtin_bit_cast(vt8, v), yv = __builtin_bit_cast(vt8, *(const uint4*)gp);
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[q] = ElemTraits<T>::from_f(ElemTraits<T>::to_f(xv[q]) + ElemTraits<T>::to_f(yv[q]));
            v = __builtin_bit_cast(uint4, xv);
          }
          *(uint4*)gp = v;
        }
      }
    }
  }
  if (has_stats) {
    // fold the 32 pixel lanes of each half (a fixed butterfly); waves of one filter block are then added in wave order
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float u1 = s1[e], u2 = s2[e];
#pragma unroll
      for (int m = 1; m < 32; m <<= 1) { u1 += __shfl_xor(u1, m); u2 += __shfl_xor(u2, m); }
      if (col == 0) {
        const int ch = (e & 3) + 8 * (e >> 2) + 4 * half;
        red[wave][0][ch] = u1;
        red[wave][1][ch] = u2;
      }
    }
    __syncthreads();
    if (tid < NW && tid < a.N) {
      const int fbk = tid >> 5, ch = tid & 31;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (w % NFB == fbk) { t1 += red[w][0][ch]; t2 += red[w][1][ch]; }
      const long so = (long)(blockIdx.x % a.stat_slots) * a.stat_stride;
      atomicAdd(a.stat_sum + so + tid, t1);
      atomicAdd(a.stat_sq + so + tid, t2);
    }
  }
}

static int smallc_epi(const IgemmArgs& a) {
  return (a.stat_sum ? 1 : 0) | (a.bias ? 2 : 0) | ((a.flags & SY11_EPI_SILU) ? 4 : 0) | ((a.flags & SY11_EPI_ACCUM) ? 8 : 0) |
         ((a.flags & SY11_EPI_OUT_F32) ? 16 : 0);
}

static int smallc_tw(int OW) { return (OW % 64 != 0 && OW % 32 == 0) ? 32 : 64; }

bool sy11_smallc3x3_legal(const IgemmArgs& a) {
  if (a.T != 9 || a.K != 9 * a.C || a.sy != 1 || a.sx != 1 || !a.dense_out || !a.vec_out || a.tail.ticket) return false;
  if (a.C != 16 && a.C != 32) return false;              // (measured r03: with 64 channels or 64 filters — 36 fragments or two blocks per
  if (a.N > 32) return false;                            //  wave — the halo / GEMM tiles are 20-60 % faster: not instantiated)
  if ( a.N % 8 || a.x_ld % 8 || a.y_ld % 8 || a.wK % 8) return false;
  if (a.debug != 0 && a.debug != 5) return false;
  if (smallc_epi(a) & 16) return false;
  unsigned seen = 0;
  for (int t = 0; t < 9; ++t) {
    const int dy = a.tap_dy[t] + 1, dx = a.tap_dx[t] + 1;
    if (dy < 0 || dy > 2 || dx < 0 || dx > 2 || a.tap_w[t] < 0 || a.tap_w[t] > 8) return false;
    seen |= 1u << (dy * 3 + dx);
  }
  if (seen != 0x1ff) return false;
  const int tw = smallc_tw(a.OW), th = 256 / tw;
  const long tiles = (long)(a.M / (a.OH * a.OW)) * cdiv(a.OH, th) * cdiv(a.OW, tw);
  const int tmax = cdiv(a.OH, th) > cdiv(a.OW, tw) ? cdiv(a.OH, th) : cdiv(a.OW, tw);
  return a.IH == a.OH && a.IW == a.OW && tiles * tmax < (1L << 31);
}

int sy11_smallc3x3_launch(const IgemmArgs& a_in, int dtype, hipStream_t st) {
  if (!sy11_smallc3x3_legal(a_in) || dtype == SY11_F32) SY11_FAIL(SY11_EUNSUPPORTED, "smallc3x3: problem not covered by the few-channel kernel");
  IgemmArgs a = a_in;
  SmallCTiling g{};
  const int tw = smallc_tw(a.OW), th = 256 / tw;
  g.tiles_x = cdiv(a.OW, tw);
  g.tiles_y = cdiv(a.OH, th);
  g.tiles = (a.M / (a.OH * a.OW)) * g.tiles_y * g.tiles_x;
  g.mag_tx = (unsigned)(((1UL << 32) + g.tiles_x - 1) / g.tiles_x);     // (a divisor of 1 is special-cased in the kernel)
  g.mag_ty = (unsigned)(((1UL << 32) + g.tiles_y - 1) / g.tiles_y);
  static int wgs = -1;
  if (wgs < 0) { const char* e = getenv("SY11_SMALLC_WG"); wgs = e ? atoi(e) : 512; }      // persistent workgroups (2 per CU)
  const int nwg = g.tiles < wgs ? g.tiles : wgs;
  dim3 grid((unsigned)nwg), block(256);
  DetPartials dp;                                   // ordered mode (det.h): one partial statistics row per workgroup
  const bool det = a.stat_sum && sy11_det(1);
  if (det) {
    if (!dp.acquire(st, 2, nwg, a.N)) SY11_FAIL(SY11_ELAUNCH, "smallc3x3: ordered-reduction workspace unavailable (%d x %d floats)", nwg, a.N);
    a.stat_sum = dp.buf(0); a.stat_sq = dp.buf(1); a.stat_slots = nwg; a.stat_stride = a.N;
  }
#define SY11_SCW(TT, CC, NF)                                                                              \
  do {                                                                                                   \
    if (tw == 64) hipLaunchKernelGGL((smallc3x3_kernel<TT, CC, NF, 64>), grid, block, 0, st, a, g);      \
    else hipLaunchKernelGGL((smallc3x3_kernel<TT, CC, NF, 32>), grid, block, 0, st, a, g);               \
  } while (0)
#define SY11_SCT(TT)                                     \
  do {                                                   \
    if (a.C == 16) SY11_SCW(TT, 16, 1);                  \
    else SY11_SCW(TT, 32, 1);                            \
  } while (0)
  if (dtype == SY11_F16) SY11_SCT(_Float16); else SY11_SCT(__bf16);
#undef SY11_SCT
#undef SY11_SCW
  SY11_LAUNCH_CHECK("smallc3x3");
  if (det) {
    return dp.fold01(a_in.stat_sum, a_in.stat_sq, a_in.stat_slots, a_in.stat_stride);
  }
  return SY11_OK;
}
