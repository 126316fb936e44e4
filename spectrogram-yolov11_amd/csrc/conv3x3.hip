// conv3x3.hip — halo-tiled 3x3 convolution (stride 1 and 2, f16) on MFMA for gfx950: forward and stride-1 data gradient.
//
// Why a second conv kernel.  The generic implicit GEMM (igemm.hip) gathers the A operand tap by tap: a 128-pixel tile pulls
// 9 x 128 pixel rows per channel slab through the CU's L1 although they are (TH+2) x (TW+2) distinct pixels.  r02 ablations
// showed what bounds those kernels: not HBM, not the MFMA pipe, not DMA latency (deeper rings change nothing) but the bytes a CU
// can take in from L2 into LDS per unit time (~50-70 GB/s per CU, MI355X_MICROARCH.md "gather into LDS").  So the lever is bytes
// per FLOP.  Here a workgroup owns a 2-D tile of TH x TW output pixels (8x16, or whole rows of a 20- / 40-wide map) and loads,
// per 32-channel slab, the input PATCH that tile needs — ((TH-1)S+3) x ((TW-1)S+3) pixels, zero-filled outside the image — ONCE;
// the nine taps read it at nine row offsets.  A bytes per tile and slab: 180 rows instead of 1152 (stride 1), 561 instead of 1152
// (stride 2).  The filter rows still stream per (slab, tap) through a 4-deep ring.
//
// LDS image of the patch: one 64-byte row per input pixel (32 channels), rows XOR-swizzled on the source side exactly as in
// igemm.hip.  For stride 2 the columns of a patch row are stored de-interleaved (even x first, then odd x): a tap reads x = 2*tx + dx,
// i.e. always one parity, so the 16 lanes of a fragment read hit 16 CONSECUTIVE rows again (conflict-free) instead of every other one.
// Padding needs no masks in the MFMA loop: out-of-image patch pixels are fetched with an out-of-range buffer offset (hardware zeros).
//
// Waves are specialised for the copies only (every wave does MFMAs): waves 0-1 issue the filter stages, waves 2-3 the patch of the
// NEXT slab in six slices during taps 0..5 — vmcnt counts per wave, so each role waits for its own copies with a trivial count and the
// stage barrier publishes them.
#include "det.h"
#include "igemm_args.h"
#include <type_traits>

typedef int h_rsrc_t __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void h_dma16(unsigned lds_addr, unsigned voff, h_rsrc_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
static __device__ __forceinline__ h_rsrc_t h_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  h_rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32) & 0xffff);
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
static __device__ __forceinline__ void h_mma(const uint4& a, const uint4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int N> static __device__ __forceinline__ void h_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// EPI bits as in igemm.hip: 1 BN statistics, 2 bias, 4 SiLU, 8 accumulate into y
template <int S, int TH, int TW, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1))) void halo3x3_kernel(const IgemmArgs a, const int tiles_x, const int tiles_y) {
  typedef _Float16 T;
  constexpr int BM = TH * TW > 128 ? 256 : 128;    // MFMA rows per workgroup; tiles of up to 256 pixels halve the filter bytes per FLOP
  constexpr int KB = 64, RPI = 16;
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  constexpr int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
  constexpr int PWh = S == 2 ? (PW + 1) / 2 : PW, PWp = S == 2 ? 2 * PWh : PW;    // row pitch of the patch image (pixels)
  constexpr int PR = PH * PWp, NIA = (PR + RPI - 1) / RPI;
  constexpr int NAW = (NIA + 1) / 2;              // patch copies (1 KiB each) per patch-loader wave and slab
  constexpr int A_BYTES = 2 * NAW * RPI * KB;     // both loader waves issue NAW copies: with NIA odd the last one is a zero-filled pad row block
  constexpr int CH = (NAW + 5) / 6;               // ... issued per tap, during taps 0..5
  constexpr int NSTB = 4, B_BYTES = BN * KB;
  constexpr int BPW = (BN / RPI + 1) / 2;          // filter copies per filter-loader wave and stage
  constexpr int NVALID = TH * TW;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(WM * WN == 4 && MI >= 1 && NI >= 1 && NVALID <= BM, "tile layout");
  static_assert(BM * BN * 2 <= 2 * A_BYTES + NSTB * B_BYTES, "epilogue staging must fit the dead operand buffers");
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES + NSTB * B_BYTES];
  __shared__ float s_red[WM * 2 * BN];           // [wave row wm][sum | sumsq][channel]: ordered fold, no LDS atomics

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool is_b = wave_u < 2;                   // waves 0-1: filter stages; waves 2-3: patches
  const int wr = wave_u & 1;
  // XCD-aware order (workgroups b, b+8, ... share an XCD): a contiguous run of neighbouring tiles per XCD shares halo lines in its L2
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % a.tiles_n;
  int tq = bid / a.tiles_n;
  const int tile_x = tq % tiles_x;
  tq /= tiles_x;
  const int tile_y = tq % tiles_y;
  const int img = tq / tiles_y;
  const int oy0 = tile_y * TH, ox0 = tile_x * TW, bn0 = tile_n * BN;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;


  // per-tap constants (wave-uniform): row offset inside the patch image, byte offset of the tap's filter columns
  int toff[9], wtap[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int dyp = a.tap_dy[t] + 1, dxp = a.tap_dx[t] + 1;
    toff[t] = dyp * PWp + (S == 1 ? dxp : (dxp == 1 ? PWh : (dxp == 2 ? 1 : 0)));
    wtap[t] = a.tap_w[t] * a.C * 2;
  }
  const h_rsrc_t xr = h_make_rsrc(a.x, a.x_bytes), wr_ = h_make_rsrc(a.w, a.w_bytes);
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned sB_base = smem_base + 2 * A_BYTES;
  const int lrow = lane >> 2, lphys = lane & 3;

  // ---- copy descriptors, hoisted.  Filter loader: BPW copies per stage; patch loader: NAW copies per slab.
  int b_off[BPW];
  int a_off[NAW];
  if (is_b) {
#pragma unroll
    for (int k = 0; k < BPW; ++k) {
      const int rl = (wr * BPW + k) * RPI + lrow, n = bn0 + rl;
      const int logical = lphys ^ ((rl >> 2) & 3);
      b_off[k] = (rl < BN && n < a.N) ? n * a.wK * 2 + logical * 16 : (int)OOB;
    }
#pragma unroll
    for (int k = 0; k < NAW; ++k) a_off[k] = (int)OOB;
  } else {
#pragma unroll
    for (int k = 0; k < BPW; ++k) b_off[k] = (int)OOB;
#pragma unroll
    for (int k = 0; k < NAW; ++k) {
      const int j = wr + 2 * k, p = j * RPI + lrow;
      const int logical = lphys ^ ((p >> 2) & 3);
      const int py = p / PWp, rem = p - py * PWp;
      int px;
      if (S == 1) {
        px = rem;
      } else {
        const int par = rem >= PWh ? 1 : 0;
        px = 2 * (rem - par * PWh) + par;
      }
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = j < NIA && p < PR && px < PW && (unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW;
      a_off[k] = ok ? ((img * a.IH + iy) * a.IW + ix) * a.x_ld * 2 + logical * 16 : (int)OOB;
    }
  }

  // ---- fragment addressing
  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 31, fh = lane >> 5;
  int arow0[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = wm * (BM / WM) + i * 32 + frow;
    const int ty = r / TW, tx = r - ty * TW;
    arow0[i] = r < NVALID ? ty * S * PWp + tx : 0;      // rows past the tile compute on pixel 0 and are never stored
  }
  int fb_off[NI][2];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int r = wn * (BN / WN) + j * 32 + frow;
      fb_off[j][g] = r * KB + (((2 * g + fh) ^ ((r >> 2) & 3)) << 4);
    }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nslab = a.C >> 5, nstage = nslab * 9;

  auto issue_b = [&](int bufq, int tapbytes, int slab) {      // one filter stage: BN rows x 64 bytes
    const unsigned dst = sB_base + bufq * B_BYTES + (wr * BPW) * (RPI * KB);
    const int ko = tapbytes + slab * KB;
#pragma unroll
    for (int k = 0; k < BPW; ++k) h_dma16(dst + k * (RPI * KB), b_off[k] >= 0 ? (unsigned)(b_off[k] + ko) : OOB, wr_);
  };
  auto issue_a = [&](int k0, int k1, int slab) {               // copies k0..k1-1 of this wave's share of patch(slab)
    const unsigned dst = smem_base + (slab & 1) * A_BYTES;
#pragma unroll
    for (int k = 0; k < NAW; ++k)
      if (k >= k0 && k < k1) h_dma16(dst + (wr + 2 * k) * (RPI * KB), a_off[k] >= 0 ? (unsigned)(a_off[k] + slab * KB) : OOB, xr);
  };

  __syncthreads();                                   // s_red zeroed before anybody can reach the epilogue
  if (is_b) {
#pragma unroll
    for (int q = 0; q < NSTB - 1; ++q)
      if (q < nstage) issue_b(q & 3, wtap[q], 0);    // stages 0..2 are taps 0..2 of slab 0
  } else {
    issue_a(0, NAW, 0);
  }

  for (int slab = 0; slab < nslab; ++slab) {
    const unsigned char* sAc = smem + (slab & 1) * A_BYTES;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int q = slab * 9 + t;
      if (is_b) {                                    // stage q landed; the (up to two) younger stages may stay in flight
        const int later = nstage - 1 - q;
        if (later >= 2) h_wait_vm<2 * BPW>();
        else if (later == 1) h_wait_vm<BPW>();
        else h_wait_vm<0>();
      } else if (t == 0) {
        h_wait_vm<0>();                              // the whole patch of this slab (issued during the previous slab)
      }
      __builtin_amdgcn_s_barrier();
      if (is_b) {
        const int t3 = (t + 3) % 9, s3 = slab + (t + 3) / 9;
        if (q + 3 < nstage) issue_b((s3 + t3) & 3, wtap[t3], s3);
      } else if (t < 6 && slab + 1 < nslab) {
        issue_a(t * CH, (t + 1) * CH < NAW ? (t + 1) * CH : NAW, slab + 1);
      }
      const unsigned char* sBc = smem + 2 * A_BYTES + ((slab + t) & 3) * B_BYTES;
      // all fragment reads of the stage (both k-groups) before its MFMAs: read - wait - MFMA per fragment exposed the LDS round
      // trip 3-4 times per stage of only 2 MI NI MFMAs (r03 ISA inspection; the order is pinned, the allocator would fold the sets)
      uint4 fa[2][MI], fb[2][NI];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = arow0[i] + toff[t];
          fa[g][i] = *(const uint4*)(sAc + row * KB + (((2 * g + fh) ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[g][j] = *(const uint4*)(sBc + fb_off[j][g]);
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) h_mma(fa[g][i], fb[g][j], acc[i][j]);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * MI * NI, 0);
    }
  }
  __syncthreads();                                   // all fragment reads done before the epilogue reuses the operand buffers

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  constexpr bool do_stats = EPI & 1, has_bias = EPI & 2, silu = EPI & 4, accum = EPI & 8;
  constexpr int ROWB = BN * 2, CPR = ROWB / 16;
  float ssum[NI], ssq[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) ssum[j] = ssq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
      const int ty = rl / TW, tx = rl - ty * TW;
      const bool rok = rl < NVALID && oy0 + ty < a.OH && ox0 + tx < a.OW;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int cl = wn * (BN / WN) + j * 32 + frow;
        const int n = bn0 + cl;
        float v = acc[i][j][e];
        if (do_stats && rok && n < a.N) { ssum[j] += v; ssq[j] += v * v; }
        if (has_bias && n < a.N) v += a.bias[n];
        if (silu) v = silu_f(v);
        *(T*)(smem + rl * ROWB + cl * 2) = (T)v;
      }
    }
  __syncthreads();
  // four chunks per thread per trip: an accumulating epilogue reads its four y chunks before the first add (one round trip, not four)
  constexpr int U = 4;
  for (int base = tid; base < NVALID * CPR; base += 256 * U) {
    unsigned char* gp[U];
    uint4 o[U];
    int lofs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * 256;
      const int rl = idx / CPR, ch = idx - rl * CPR;
      const int ty = rl / TW, tx = rl - ty * TW;
      const int oy = oy0 + ty, ox = ox0 + tx, n = bn0 + ch * 8;
      const bool ok = idx < NVALID * CPR && oy < a.OH && ox < a.OW && n < a.N && a.debug != 5;      // debug 5 = tuner dry run of an accumulating epilogue
      gp[u] = ok ? (unsigned char*)a.y + (((long)(img * a.OH + oy) * a.OW + ox) * a.y_ld + n) * 2 : nullptr;
      lofs[u] = rl * ROWB + ch * 16;
      if (accum && ok) o[u] = *(const uint4*)gp[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!gp[u]) continue;
      uint4 v = *(const uint4*)(smem + lofs[u]);
      if (accum) {
        f16x8 x = __builtin_bit_cast(f16x8, v), y = __builtin_bit_cast(f16x8, o[u]);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (_Float16)((float)x[k] + (float)y[k]);
        v = __builtin_bit_cast(uint4, x);
      }
      *(uint4*)gp[u] = v;
    }
  }
  if (do_stats) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const float s1 = ssum[j] + __shfl_xor(ssum[j], 32);
      const float s2 = ssq[j] + __shfl_xor(ssq[j], 32);
      if (fh == 0) {
        const int col = wn * (BN / WN) + j * 32 + frow;
        s_red[wm * 2 * BN + col] = s1;            // one row per wave row: the fold below adds them in index order (bit-reproducible)
        s_red[wm * 2 * BN + BN + col] = s2;
      }
    }
    __syncthreads();
    if (tid < BN && bn0 + tid < a.N) {
      const long so = (long)(blockIdx.x % a.stat_slots) * a.stat_stride;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int r = 0; r < WM; ++r) { t1 += s_red[r * 2 * BN + tid]; t2 += s_red[r * 2 * BN + BN + tid]; }
      atomicAdd(a.stat_sum + so + bn0 + tid, t1);
      atomicAdd(a.stat_sq + so + bn0 + tid, t2);
    }
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int halo_epi(const IgemmArgs& a) {
  return (a.stat_sum ? 1 : 0) | (a.bias ? 2 : 0) | ((a.flags & SY11_EPI_SILU) ? 4 : 0) | ((a.flags & SY11_EPI_ACCUM) ? 8 : 0) |
         ((a.flags & SY11_EPI_OUT_F32) ? 16 : 0);
}
static bool halo_tile(int OW, int OH, int big, int* th, int* tw) {
  if (OW % 16 == 0) { *tw = 16; *th = (big && OH % 16 == 0) ? 16 : 8; return !big || *th == 16; }
  if (OW == 20) { *tw = 20; *th = big ? 12 : 6; return true; }
  if (OW == 40) { *tw = 40; *th = big ? 6 : 3; return true; }
  return false;
}

// bn = channels per workgroup (128 / 64 / 32); +1000 = the 256-pixel tiles (stride 1 only)
bool sy11_halo3x3_legal(const IgemmArgs& a, int bn_code) {
  const int big = bn_code >= 1000, bn = bn_code % 1000;
  if (big && (a.sy != 1 || bn < 64)) return false;
  if (a.T != 9 || a.K != 9 * a.C || a.C % 32 || a.N % 8 || !a.dense_out || !a.vec_out || a.tail.ticket || (a.debug != 0 && a.debug != 5)) return false;
  if (!((a.sy == 1 && a.sx == 1) || (a.sy == 2 && a.sx == 2))) return false;
  const int epi = halo_epi(a);
  if (a.sy == 1 ? !(epi == 0 || epi == 1 || epi == 8 || epi == 6) : !(epi == 0 || epi == 1 || epi == 6)) return false;
  unsigned seen = 0;
  for (int t = 0; t < 9; ++t) {
    const int dy = a.tap_dy[t] + 1, dx = a.tap_dx[t] + 1;
    if (dy < 0 || dy > 2 || dx < 0 || dx > 2) return false;
    seen |= 1u << (dy * 3 + dx);
  }
  if (seen != 0x1ff) return false;
  int th, tw;
  if (!halo_tile(a.OW, a.OH, big, &th, &tw)) return false;
  if (bn != 128 && bn != 64 && bn != 32) return false;
  if (bn > 32 && bn >= 2 * a.N) return false;          // a tile more than twice the channel count is pure waste
  return true;
}

template <int S, int TH, int TW, int BN, int WM, int WN>
static void halo_launch_epi(const IgemmArgs& a, int epi, dim3 grid, hipStream_t st, int tx, int ty) {
  dim3 block(256);
  switch (epi) {
    case 0: hipLaunchKernelGGL((halo3x3_kernel<S, TH, TW, BN, WM, WN, 0>), grid, block, 0, st, a, tx, ty); break;
    case 1: hipLaunchKernelGGL((halo3x3_kernel<S, TH, TW, BN, WM, WN, 1>), grid, block, 0, st, a, tx, ty); break;
    case 6: hipLaunchKernelGGL((halo3x3_kernel<S, TH, TW, BN, WM, WN, 6>), grid, block, 0, st, a, tx, ty); break;
    case 8:
      if constexpr (S == 1) hipLaunchKernelGGL((halo3x3_kernel<S, TH, TW, BN, WM, WN, 8>), grid, block, 0, st, a, tx, ty);
      break;
    default: break;
  }
}
template <int S, int TH, int TW>
static void halo_launch_bn(const IgemmArgs& a, int bn, int epi, dim3 grid, hipStream_t st, int tx, int ty) {
  if (bn == 128) halo_launch_epi<S, TH, TW, 128, 2, 2>(a, epi, grid, st, tx, ty);
  else if (bn == 64) halo_launch_epi<S, TH, TW, 64, 4, 1>(a, epi, grid, st, tx, ty);
  else if constexpr (TH * TW <= 128) halo_launch_epi<S, TH, TW, 32, 4, 1>(a, epi, grid, st, tx, ty);
}
template <int S>
static void halo_launch_tile(const IgemmArgs& a, int bn, int epi, int th, int tw, dim3 grid, hipStream_t st, int tx, int ty) {
  if (tw == 16 && th == 8) halo_launch_bn<S, 8, 16>(a, bn, epi, grid, st, tx, ty);
  else if (tw == 20 && th == 6) halo_launch_bn<S, 6, 20>(a, bn, epi, grid, st, tx, ty);
  else if (tw == 40 && th == 3) halo_launch_bn<S, 3, 40>(a, bn, epi, grid, st, tx, ty);
  else if constexpr (S == 1) {
    if (tw == 16) halo_launch_bn<1, 16, 16>(a, bn, epi, grid, st, tx, ty);
    else if (tw == 20) halo_launch_bn<1, 12, 20>(a, bn, epi, grid, st, tx, ty);
    else halo_launch_bn<1, 6, 40>(a, bn, epi, grid, st, tx, ty);
  }
}

int sy11_halo3x3_launch(const IgemmArgs& a_in, int bn_code, hipStream_t st) {
  if (!sy11_halo3x3_legal(a_in, bn_code)) SY11_FAIL(SY11_EUNSUPPORTED, "halo3x3: problem not covered by the halo-tiled kernel");
  const int big = bn_code >= 1000, bn = bn_code % 1000;
  IgemmArgs a = a_in;
  int th, tw;
  halo_tile(a.OW, a.OH, big, &th, &tw);
  const int B = a.M / (a.OH * a.OW);
  const int tx = cdiv(a.OW, tw), ty = cdiv(a.OH, th);
  a.tiles_n = cdiv(a.N, bn);
  const long nwg = (long)B * tx * ty * a.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffL) SY11_FAIL(SY11_EINVAL, "halo3x3: bad grid %ld", nwg);
  dim3 grid((unsigned)nwg);
  const int epi = halo_epi(a);
  DetPartials dp;                                   // ordered mode (det.h): one partial statistics row per workgroup
  const bool det = a.stat_sum && sy11_det(1);
  if (det) {
    if (!dp.acquire(st, 2, nwg, a.N)) SY11_FAIL(SY11_ELAUNCH, "halo3x3: ordered-reduction workspace unavailable (%ld x %d floats)", nwg, a.N);
    a.stat_sum = dp.buf(0); a.stat_sq = dp.buf(1); a.stat_slots = (int)nwg; a.stat_stride = a.N;
  }
  if (a.sy == 1) halo_launch_tile<1>(a, bn, epi, th, tw, grid, st, tx, ty);
  else halo_launch_tile<2>(a, bn, epi, th, tw, grid, st, tx, ty);
  SY11_LAUNCH_CHECK("halo3x3");
  if (det) {
    return dp.fold01(a_in.stat_sum, a_in.stat_sq, a_in.stat_slots, a_in.stat_stride);
  }
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ stride-2 input gradient
// dx of a 3x3 / stride 2 / pad 1 convolution, all four output-parity classes in ONE pass over dy.
//
// The generic path (igemm.hip) runs one launch per parity class of dx (1, 2, 2 and 4 taps): dy is read four times, every class writes
// every other pixel of every other row (half cache lines), and each class is a short-K GEMM.  Here a workgroup owns TH x TW QUADS
// (2x2 blocks of dx pixels = TH x TW positions of the dy grid): the dy patch it needs — (TH+1) x (TW+1) pixels — is loaded once per
// 32-channel slab, the nine taps run as nine MFMA stages into FOUR accumulator sets (one per parity class: a tap belongs to exactly
// one class and reads the patch at a row offset in {0,1} x {0,1}), and the epilogue interleaves the four classes in LDS so that dx is
// written as whole contiguous pixel rows of the 2TH x 2TW block.  Same machinery as halo3x3_kernel otherwise.
//   class (py, px) of dx pixel (2Y+py, 2X+px):  py = 0: filter row r = 1 (dy row Y);  py = 1: r = 0 (dy row Y+1) and r = 2 (dy row Y).
template <int TH, int TW, int BN, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void halo_dgrad_s2_kernel(const IgemmArgs a, const int tiles_x, const int tiles_y) {
  typedef _Float16 T;
  constexpr int BM = 128, KB = 64, RPI = 16, NI = BN / 32;
  constexpr int PH = TH + 1, PW = TW + 1, PR = PH * PW, NIA = (PR + RPI - 1) / RPI;
  constexpr int NAW = (NIA + 1) / 2, A_BYTES = 2 * NAW * RPI * KB, CH = (NAW + 5) / 6;
  constexpr int NSTB = 4, B_BYTES = BN * KB, BPW = (BN / RPI + 1) / 2;
  constexpr int NVALID = TH * TW;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int LOOP_BYTES = 2 * A_BYTES + NSTB * B_BYTES, OUT_BYTES = 2 * BM * BN * 2;   // epilogue: one dx row parity at a time
  static_assert(NVALID <= BM && BN % 32 == 0, "tile layout");
  __shared__ __attribute__((aligned(16))) unsigned char smem[LOOP_BYTES > OUT_BYTES ? LOOP_BYTES : OUT_BYTES];
  // stage t -> (parity class, patch row offset, patch column offset, filter tap r*3+s)
  constexpr int CLS[9] = {0, 1, 1, 2, 2, 3, 3, 3, 3};
  constexpr int AY[9] = {0, 0, 0, 1, 0, 1, 1, 0, 0};
  constexpr int AX[9] = {0, 1, 0, 0, 0, 1, 0, 1, 0};
  constexpr int WT[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool is_b = wave_u < 2;
  const int wr = wave_u & 1;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % a.tiles_n;
  int tq = bid / a.tiles_n;
  const int tile_x = tq % tiles_x;
  tq /= tiles_x;
  const int tile_y = tq % tiles_y;
  const int img = tq / tiles_y;
  const int Y0 = tile_y * TH, X0 = tile_x * TW, bn0 = tile_n * BN;
  // here: a.x = dy (B, a.IH, a.IW, a.C) with a.IH/IW = the conv's OUTPUT size; a.y = dx (B, a.OHF, a.OWF, a.N) = the conv's input
  const h_rsrc_t xr = h_make_rsrc(a.x, a.x_bytes), wr_ = h_make_rsrc(a.w, a.w_bytes);
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned sB_base = smem_base + 2 * A_BYTES;
  const int lrow = lane >> 2, lphys = lane & 3;

  int b_off[BPW], a_off[NAW];
#pragma unroll
  for (int k = 0; k < BPW; ++k) {
    const int rl = (wr * BPW + k) * RPI + lrow, n = bn0 + rl;
    b_off[k] = (is_b && rl < BN && n < a.N) ? n * a.wK * 2 + (lphys ^ ((rl >> 2) & 3)) * 16 : (int)OOB;
  }
#pragma unroll
  for (int k = 0; k < NAW; ++k) {
    const int j = wr + 2 * k, p = j * RPI + lrow;
    const int py = p / PW, px = p - py * PW;
    const int oy = Y0 + py, ox = X0 + px;
    const bool ok = !is_b && j < NIA && p < PR && oy < a.IH && ox < a.IW;
    a_off[k] = ok ? ((img * a.IH + oy) * a.IW + ox) * a.x_ld * 2 + (lphys ^ ((p >> 2) & 3)) * 16 : (int)OOB;
  }
  const int frow = lane & 31, fh = lane >> 5;
  const int r0 = wave * 32 + frow;
  const int arow0 = r0 < NVALID ? (r0 / TW) * PW + (r0 % TW) : 0;
  int fb_off[NI][2];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int r = j * 32 + frow;
      fb_off[j][g] = r * KB + (((2 * g + fh) ^ ((r >> 2) & 3)) << 4);
    }
  f32x16 acc[4][NI];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[c][j][e] = 0.f;

  const int nslab = a.C >> 5, nstage = nslab * 9;
  auto issue_b = [&](int bufq, int tap, int slab) {
    const unsigned dst = sB_base + bufq * B_BYTES + (wr * BPW) * (RPI * KB);
    const int ko = (tap * a.C + slab * 32) * 2;
#pragma unroll
    for (int k = 0; k < BPW; ++k) h_dma16(dst + k * (RPI * KB), b_off[k] >= 0 ? (unsigned)(b_off[k] + ko) : OOB, wr_);
  };
  auto issue_a = [&](int k0, int k1, int slab) {
    const unsigned dst = smem_base + (slab & 1) * A_BYTES;
#pragma unroll
    for (int k = 0; k < NAW; ++k)
      if (k >= k0 && k < k1) h_dma16(dst + (wr + 2 * k) * (RPI * KB), a_off[k] >= 0 ? (unsigned)(a_off[k] + slab * KB) : OOB, xr);
  };
  if (is_b) {
#pragma unroll
    for (int q = 0; q < NSTB - 1; ++q)
      if (q < nstage) issue_b(q & 3, WT[q], 0);
  } else {
    issue_a(0, NAW, 0);
  }
  for (int slab = 0; slab < nslab; ++slab) {
    const unsigned char* sAc = smem + (slab & 1) * A_BYTES;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int q = slab * 9 + t;
      if (is_b) {
        const int later = nstage - 1 - q;
        if (later >= 2) h_wait_vm<2 * BPW>();
        else if (later == 1) h_wait_vm<BPW>();
        else h_wait_vm<0>();
      } else if (t == 0) {
        h_wait_vm<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (is_b) {
        const int t3 = (t + 3) % 9, s3 = slab + (t + 3) / 9;
        if (q + 3 < nstage) issue_b((s3 + t3) & 3, WT[t3], s3);
      } else if (t < 6 && slab + 1 < nslab) {
        issue_a(t * CH, (t + 1) * CH < NAW ? (t + 1) * CH : NAW, slab + 1);
      }
      const unsigned char* sBc = smem + 2 * A_BYTES + ((slab + t) & 3) * B_BYTES;
      const int row = arow0 + AY[t] * PW + AX[t];
      uint4 fa[2], fb[2][NI];                         // all fragment reads of the stage first (see halo3x3_kernel)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        fa[g] = *(const uint4*)(sAc + row * KB + (((2 * g + fh) ^ ((row >> 2) & 3)) << 4));
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[g][j] = *(const uint4*)(sBc + fb_off[j][g]);
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < NI; ++j) h_mma(fa[g], fb[g][j], acc[CLS[t]][j]);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * (1 + NI), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * NI, 0);
    }
  }
  __syncthreads();
  // ---- epilogue, one dx row parity (py) at a time: the two classes (py, 0) and (py, 1) interleaved into a [TH][2TW][BN] image in
  // LDS = the dx rows 2*(Y0+ty)+py of the block, then written as whole contiguous pixel rows
  constexpr bool accum = EPI & 8;
  constexpr int ROWB = BN * 2, CPR = ROWB / 16, DW = 2 * TW;
#pragma unroll
  for (int py = 0; py < 2; ++py) {
    if (py) __syncthreads();                          // the previous parity's rows have been read out
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;         // quad index
        if (rl < NVALID) {
          const int ty = rl / TW, tx = rl - ty * TW;
          const int prow = ty * DW + 2 * tx + px;
#pragma unroll
          for (int j = 0; j < NI; ++j) *(T*)(smem + prow * ROWB + (j * 32 + frow) * 2) = (T)acc[2 * py + px][j][e];
        }
      }
    __syncthreads();
    constexpr int U = 4;                               // as in halo3x3_kernel: the reads of an accumulating epilogue go out four at a time
    for (int base = tid; base < 2 * NVALID * CPR; base += 256 * U) {
      unsigned char* gp[U];
      uint4 o[U];
      int lofs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * 256;
        const int prow = idx / CPR, ch = idx - prow * CPR;
        const int ty = prow / DW, xx = prow - ty * DW;
        const int iy = 2 * (Y0 + ty) + py, ix = 2 * X0 + xx, n = bn0 + ch * 8;
        const bool ok = idx < 2 * NVALID * CPR && iy < a.OHF && ix < a.OWF && n < a.N && a.debug != 5;
        gp[u] = ok ? (unsigned char*)a.y + (((long)(img * a.OHF + iy) * a.OWF + ix) * a.y_ld + n) * 2 : nullptr;
        lofs[u] = prow * ROWB + ch * 16;
        if (accum && ok) o[u] = *(const uint4*)gp[u];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!gp[u]) continue;
        uint4 v = *(const uint4*)(smem + lofs[u]);
        if (accum) {
          f16x8 x = __builtin_bit_cast(f16x8, v), y = __builtin_bit_cast(f16x8, o[u]);
#pragma unroll
          for (int k = 0; k < 8; ++k) x[k] = (_Float16)((float)x[k] + (float)y[k]);
          v = __builtin_bit_cast(uint4, x);
        }
        *(uint4*)gp[u] = v;
      }
    }
  }
}

// a: x = dy, IH/IW = dy's (the conv's output) size, C = dy channels, w = tap-transposed filter [Cin][9][C], wK = 9*C,
//    y = dx, OHF/OWF = dx's (the conv's input) size, N = Cin, flags & ACCUM.  Legal: f16, C % 32 == 0, N % 8 == 0, tile table.
bool sy11_halo_dgrad_s2_legal(const IgemmArgs& a) {
  int th, tw;
  if (a.C % 32 || a.N % 8 || a.wK != 9 * a.C || !halo_tile(a.IW, a.IH, 0, &th, &tw)) return false;
  if (((uintptr_t)a.y & 15) || (a.y_ld * 2) % 16 || ((uintptr_t)a.x & 15) || (a.x_ld * 2) % 16) return false;
  return (a.OHF == 2 * a.IH || a.OHF == 2 * a.IH - 1) && (a.OWF == 2 * a.IW || a.OWF == 2 * a.IW - 1);
}

template <int TH, int TW>
static void halo_dgrad_s2_tile(const IgemmArgs& a, int bn, bool accum, dim3 grid, hipStream_t st, int tx, int ty) {
  dim3 block(256);
  if (bn == 64) {
    if (accum) hipLaunchKernelGGL((halo_dgrad_s2_kernel<TH, TW, 64, 8>), grid, block, 0, st, a, tx, ty);
    else hipLaunchKernelGGL((halo_dgrad_s2_kernel<TH, TW, 64, 0>), grid, block, 0, st, a, tx, ty);
  } else {
    if (accum) hipLaunchKernelGGL((halo_dgrad_s2_kernel<TH, TW, 32, 8>), grid, block, 0, st, a, tx, ty);
    else hipLaunchKernelGGL((halo_dgrad_s2_kernel<TH, TW, 32, 0>), grid, block, 0, st, a, tx, ty);
  }
}

int sy11_halo_dgrad_s2_launch(const IgemmArgs& a_in, hipStream_t st) {
  if (!sy11_halo_dgrad_s2_legal(a_in)) SY11_FAIL(SY11_EUNSUPPORTED, "halo_dgrad_s2: problem not covered");
  IgemmArgs a = a_in;
  int th, tw;
  halo_tile(a.IW, a.IH, 0, &th, &tw);
  // 64-wide channel tiles even for 128+ channels: 196 VGPRs = TWO workgroups per CU, which beats one 128-wide tile per CU that
  // loads the dy patch half as often (r02, tools/dgrad_s2_micro.py: 160x160x128 <- 80x80x128 266 -> 218 us, 80x80x256 <- 40x40x256 227 -> 186)
  const int bn = a.N > 32 ? 64 : 32;
  const int B = a.M;                                  // the caller passes the batch size in M
  const int tx = cdiv(a.IW, tw), ty = cdiv(a.IH, th);
  a.tiles_n = cdiv(a.N, bn);
  const long nwg = (long)B * tx * ty * a.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffL) SY11_FAIL(SY11_EINVAL, "halo_dgrad_s2: bad grid %ld", nwg);
  dim3 grid((unsigned)nwg);
  const bool accum = a.flags & SY11_EPI_ACCUM;
  if (tw == 16) halo_dgrad_s2_tile<8, 16>(a, bn, accum, grid, st, tx, ty);
  else if (tw == 20) halo_dgrad_s2_tile<6, 20>(a, bn, accum, grid, st, tx, ty);
  else halo_dgrad_s2_tile<3, 40>(a, bn, accum, grid, st, tx, ty);
  SY11_LAUNCH_CHECK("halo_dgrad_s2");
  return SY11_OK;
}
