// igemm.hip — NHWC implicit-GEMM convolution on MFMA for gfx950 (forward and data-gradient).
//
// GEMM view:  D[m][n] = sum_k A[m][k] * Bw[n][k]
//   m = output pixel of a (B, OH, OW) grid, n = output channel, k = (tap t, channel c).
//   A is never materialised: row m / column k is gathered from the NHWC input at
//   (b, oy*sy + tap_dy[t], ox*sx + tap_dx[t], c), zero outside the image (padding) — im2col on the fly.
//   Bw row n is the filter [n][tap_w[t]][c] (K-contiguous), so both operands are "K-major".
//
// Tiling (one 256-thread workgroup = 4 wave64):  BM=128 pixels x BN in {128,64,32} channels x 128 BYTES of K
// per stage (32 f32 / 64 f16 elements), two LDS stages, register-staged global->LDS copies issued one stage
// ahead (cdna guide T14), 16-byte ds_read_b128 fragment reads from an XOR-swizzled image
// (phys_chunk = chunk ^ ((row>>1)&7): conflict-free for the four 16-lane groups of ds_read_b128 on 128-B rows).
// MFMA: v_mfma_f32_32x32x2_f32 (exact f32, 4 per 16-byte K group) or v_mfma_f32_32x32x16_{f16,bf16}.
// For f32 the k order inside a 16-byte group is permuted identically for A and B (lane half h owns chunk 2g+h),
// which leaves the dot product unchanged.
//
// The same kernel computes the data gradient: input := dy, filter := tap-transposed weights, taps := the
// (P - r*D)/S offsets of one output-parity class, output written with a pixel stride (oy*oy_mul+oy_add).
#include "common.h"
#include "tune.h"
#include "det.h"
#include "bn_tail.h"
#include "igemm_args.h"
#include <stdlib.h>
#include <type_traits>

__device__ uint4 g_zero16;   // zero page: source of padded / out-of-range 16-byte chunks (zero-initialised)

template <typename T> struct Mma;
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), c, 0, 0, 0);
  }
};
template <> struct Mma<_Float16> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<__bf16> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

// One wave-wide LDS-DMA: 64 lanes x 16 bytes from (buffer descriptor + per-lane byte offset) to LDS at
// (wave-uniform lds_addr + lane*16); out-of-range offsets deliver zeros.  Issued through inline asm ON PURPOSE: when the
// compiler sees the LDS-DMA builtin it must assume every later ds_read may alias the DMA destination and puts
// s_waitcnt vmcnt(0) in front of the MFMA fragment reads and the tap-table reads, i.e. it serialises the copy of stage
// s+1 behind the math of stage s (r01 ISA inspection).  Here the ring is synchronised by hand (counted vmcnt + s_barrier),
// and the asm is also never duplicated into divergent branches, so "loads per stage" is an exact count.
typedef int dma_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_dma16(unsigned lds_addr, unsigned voff, dma_rsrc_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
// cache-policy variants of the same copy: sc1 = served from L2 without allocating in this CU's L1; nt = non-temporal
__device__ __forceinline__ void lds_dma16_sc1(unsigned lds_addr, unsigned voff, dma_rsrc_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen sc1 lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void lds_dma16_nt(unsigned lds_addr, unsigned voff, dma_rsrc_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ dma_rsrc_t make_dma_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  dma_rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32) & 0xffff);    // stride 0, no swizzle
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);                            // num_records (raw buffer: bytes)
  r[3] = 0x00020000;
  return r;
}

// KB = bytes of K per LDS row / stage (128 or 64), NST = LDS ring depth (2 or 3).
template <int KB> __device__ __forceinline__ int swz(int r) { return KB == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3); }

// EPI: compile-time epilogue (bit 0 stats, 1 bias, 2 SiLU, 3 accumulate, 4 f32 output) or -1 = decide from runtime flags.
// Small-K layers (1x1 convs, K = 64..192) spend most of their instructions in the epilogue, so its dead branches matter.
template <typename T, int BM, int BN, int WM, int WN, int KB, int NST, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BM == 256 ? 2 : 1))) void igemm_kernel(const IgemmArgs a) {
  constexpr int EPC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int CPRW = KB / 16;             // 16-byte chunks per LDS row
  constexpr int BK = CPRW * EPC;            // elements of K per stage
  constexpr int RPI = 64 / CPRW;            // tile rows moved by one wave-wide LDS-DMA instruction (1 KiB)
  constexpr int RPP = 4 * RPI;              // rows per pass of the 4 waves
  constexpr int MI = BM / WM / 32;
  constexpr int NI = BN / WN / 32;
  constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB, STAGE = A_BYTES + B_BYTES;
  static_assert(WM * WN == 4 && MI >= 1 && NI >= 1, "wave layout");
  static_assert(BN % RPP == 0 || BN < RPP, "pass geometry");
  __shared__ __attribute__((aligned(16))) unsigned char smem[NST * STAGE + 192 + 768];
  __shared__ float s_red[WM * 2 * BN];           // BN batch statistics of this tile, [wave row wm][sum | sumsq][channel]: ordered fold, no LDS atomics

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware tile order: workgroups b and b+8 share an XCD (and its L2); give each XCD a contiguous run of tiles
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;


  constexpr int APASS = BM / RPP, BPASS = (BN + RPP - 1) / RPP;
  constexpr int ESZ = (int)sizeof(T);
  constexpr unsigned OOB = 0x80000000u;      // any offset >= num_records makes the buffer load return zeros (padding / tails)
  // ---- staging by LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... offen lds): 64 lanes x 16 bytes land at
  // (wave-uniform LDS base + lane*16) = RPI consecutive tile rows; no VGPR round trip, no ds_write.  The bank swizzle sits
  // on the SOURCE side (the lane landing on physical chunk p of row r fetches logical chunk p ^ swz(r)).  All per-stage
  // address work is hoisted: a row's byte offset at tap (0,0) and a 64-bit "tap is inside the image" mask are computed
  // once; a stage adds one per-tap delta (LDS table) and turns invalid lanes into an out-of-range offset (hardware zero fill).
  const int ld_row = tid / CPRW;
  const int ld_chunk = (tid % CPRW) ^ swz<KB>(ld_row);           // LOGICAL chunk this lane fetches (swizzle is pass-invariant)
  int* s_tapoff = (int*)(smem + NST * STAGE + 192);              // [64] byte delta of tap t in x, [64] byte delta in w, [64] packed (dy, dx)
  if (tid < 64) {
    const int t = tid < a.T ? tid : 0;
    const int dy = a.tap_dy[t], dx = a.tap_dx[t];
    s_tapoff[tid] = (dy * a.IW + dx) * a.x_ld * ESZ;
    s_tapoff[64 + tid] = a.tap_w[t] * a.C * ESZ;
    s_tapoff[128 + tid] = (dy & 0xffff) | (dx << 16);
  }
  const int ohw = a.OH * a.OW;
  int a_off[APASS], a_iy[APASS], a_ix[APASS];
  unsigned long long a_mask[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = bm0 + ld_row + RPP * i;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    const int b = mm / ohw, r = mm - b * ohw, oy = r / a.OW, ox = r - oy * a.OW;
    const int iy0 = oy * a.sy, ix0 = ox * a.sx;
    a_off[i] = ((b * a.IH + iy0) * a.IW + ix0) * a.x_ld * ESZ;       // host guarantees < 2^31 bytes
    a_iy[i] = ok ? iy0 : -0x4000;                                    // a row past M: no tap is inside the image
    a_ix[i] = ix0;
    a_mask[i] = 0;
  }
  int b_off[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int rl = ld_row + RPP * i, n = bn0 + rl;
    b_off[i] = (n < a.N && rl < BN) ? n * a.wK * ESZ : (int)OOB;
  }
  // (tap, channel-in-tap) of this thread's chunk in the stage about to be issued.  Carried BY VALUE through issue_stage:
  // captured by reference next to the "memory"-clobbering DMA asm they were kept in scratch memory (r02 ISA inspection:
  // two scratch_load_dword + s_waitcnt vmcnt(0) at the top of every stage — a ~400-cycle stall in front of each DMA issue
  // that also drained every LDS-DMA still in flight, so deeper rings could not help)
  // xo / wo: the tap's byte deltas (LDS table) of THIS position, read one stage ahead — at the end of the previous issue — so that the
  // table's LDS round trip hides behind a stage of MFMAs instead of sitting in front of every stage's DMA issue (r03)
  struct KPos { int kt, kc, xo, wo; };
  KPos kp;
  kp.kt = (ld_chunk * EPC) / a.C;
  kp.kc = (ld_chunk * EPC) - kp.kt * a.C;
  kp.xo = kp.wo = 0;
  const dma_rsrc_t xr = make_dma_rsrc(a.x, a.x_bytes), wr_ = make_dma_rsrc(a.w, a.w_bytes);
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const bool simple_k = a.C >= BK;
  // K order.  Tap-major (all channel slabs of tap 0, then tap 1 ...) re-reads an input pixel after C/BK stages — far beyond
  // what the 32 KB L1 keeps — so every tap goes to L2 again.  Channel-major (the T taps of one slab back to back) makes the
  // next stage read the rows the previous one just fetched, shifted by one pixel: most of them are still in L1.
  const bool chan_major = a.chan_major && a.T > 1 && a.C % BK == 0;
  const bool b_issue = (BN >= RPP) || (wave_u * RPI < BN);         // BN < rows-per-pass: only waves covering real rows issue
  __syncthreads();  // tap tables visible
  // "tap t of row i lies inside the image", one bit per tap, from the LDS tap table (one broadcast read per tap for all of a lane's rows).
  // r01-r03 read the taps from the kernel arguments inside this loop: every tap of every row was a dependent scalar-memory round trip,
  // 16 000 cycles per workgroup on a 3x3 layer (in-kernel stamps of the 8-wave pipeline, r04) — now ~2 000
#pragma unroll 1
  for (int t = 0; t < a.T; ++t) {
    const int v = s_tapoff[128 + t];
    const int dy = (short)(v & 0xffff), dx = v >> 16;
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      if ((unsigned)(a_iy[i] + dy) < (unsigned)a.IH && (unsigned)(a_ix[i] + dx) < (unsigned)a.IW) a_mask[i] |= 1ull << t;
  }
  {
    const int tt0 = kp.kt < a.T ? kp.kt : 0;
    kp.xo = s_tapoff[tt0];
    kp.wo = s_tapoff[64 + tt0];
  }

  auto issue_stage = [&](int stage_idx, KPos k) -> KPos {
    int kt = k.kt, kc = k.kc;
    const unsigned sa = smem_base + stage_idx * STAGE + wave_u * (RPI * KB);
    const unsigned sb = sa + A_BYTES;
    const bool kvalid = kt < a.T;
    const int tt = kvalid ? kt : 0;
    const int xo = k.xo + kc * ESZ, wo = k.wo + kc * ESZ;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const bool ok = kvalid && ((a_mask[i] >> tt) & 1ull);
      lds_dma16(sa + i * (RPP * KB), ok ? (unsigned)(a_off[i] + xo) : OOB, xr);
    }
    if (b_issue) {
      // filter rows: every workgroup of the launch streams the same rows once per tile — with the default policy they also
      // pass through (and evict from) the 32 KB L1 that could otherwise keep the tile's input rows across the taps of a slab
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        const unsigned off = (kvalid && b_off[i] >= 0) ? (unsigned)(b_off[i] + wo) : OOB;
        // (r02 tried sc1 / nt cache policies for the filter rows behind a run-time switch: no effect — and the switch put a branch
        // chain around every filter copy of every stage; r03: one plain copy)
        lds_dma16(sb + i * (RPP * KB), off, wr_);
      }
    }
    if (chan_major) {                             // all taps of one channel slab back to back (see chan_major above)
      const bool wrap = kt + 1 >= a.T;
      kt = wrap ? 0 : kt + 1;
      kc += wrap ? BK : 0;
    } else {
    kc += BK;
    if (simple_k) {                               // C >= BK: at most one tap boundary per stage, branch-free
      const bool wrap = kc >= a.C;
      kc -= wrap ? a.C : 0;
      kt += wrap ? 1 : 0;
    } else {
      while (kc >= a.C) { kc -= a.C; ++kt; }
    }
    }
    KPos r;
    r.kt = kt; r.kc = kc;
    const int tn = kt < a.T ? kt : 0;
    r.xo = s_tapoff[tn];
    r.wo = s_tapoff[64 + tn];
    return r;
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 31, fh = lane >> 5;
  const int nstage = a.debug == 4 ? 0 : (a.K + BK - 1) / BK;   // debug 4 = ablation: epilogue only
  // fragment read addresses (bytes inside a stage), hoisted: one per (tile row set, k-group)
  int fa_off[MI][CPRW / 2], fb_off[NI][CPRW / 2];
#pragma unroll
  for (int g = 0; g < CPRW / 2; ++g) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r = wm * (BM / WM) + i * 32 + frow;
      fa_off[i][g] = r * KB + (((2 * g + fh) ^ swz<KB>(r)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int r = wn * (BN / WN) + j * 32 + frow;
      fb_off[j][g] = A_BYTES + r * KB + (((2 * g + fh) ^ swz<KB>(r)) << 4);
    }
  }

  // NST-deep LDS ring, ONE raw barrier per stage, counted vmcnt; the ring walk is unrolled NST times so every LDS
  // address is (hoisted VGPR + compile-time stage offset).
  auto consume = [&](const unsigned char* st) {
    constexpr int G = CPRW / 2;
    if constexpr (BM == 256) {
      // 1-2 waves per SIMD: nothing else hides an LDS round trip, so ALL fragment reads of the stage are issued before the first
      // MFMA (G x (MI + NI) x 16 bytes per lane) and the MFMAs drain them in order behind counted waits.  (r02 read one k-group
      // ahead with two register sets; the register allocator folded the sets into one and the second k-group ran as
      // read - wait - 4 MFMAs - read - wait ...: three exposed LDS latencies per stage, r03 ISA inspection.)
      uint4 fa[G][MI], fb[G][NI];
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[g][i] = *(const uint4*)(st + fa_off[i][g]);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[g][j] = *(const uint4*)(st + fb_off[j][g]);
      }
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) Mma<T>::run(fa[g][i], fb[g][j], acc[i][j]);
      // pin the order the source states: the reads first, then the MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, G * (MI + NI), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, G * MI * NI, 0);
    } else {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        uint4 fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = *(const uint4*)(st + fa_off[i][g]);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[j] = *(const uint4*)(st + fb_off[j][g]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) Mma<T>::run(fa[i], fb[j], acc[i][j]);
      }
    }
  };
  // NST - 1 stages are kept in flight: on the small maps (20x20 / 40x40: 200..800 workgroups, 1..3 per CU) a stage is one
  // L2 / fabric round trip (~1 us) that nothing else on the CU hides, so the kernel time is (stages x latency) / depth.
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < nstage) kp = issue_stage(p, kp);
  for (int s0 = 0; s0 < nstage; s0 += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int s = s0 + u;
      if (s < nstage) {
        // stage s must have landed; the (up to NST - 2) stages issued after it may stay outstanding
        const int later = nstage - 1 - s;
        if (NST >= 4 && later >= 2) {
          if (b_issue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (APASS + BPASS)) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * APASS) : "memory");
        } else if (NST >= 3 && later >= 1) {
          if (b_issue) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APASS + BPASS) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(APASS) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (s + NST - 1 < nstage && a.debug != 1) kp = issue_stage((u + NST - 1) % NST, kp);
        if (a.debug != 2) consume(smem + u * STAGE);
      }
    }
  }
  __syncthreads();                               // all fragment reads done before the epilogue reuses the ring
  if (a.debug == 3) return;                      // ablation: no epilogue

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  constexpr bool RT = EPI < 0;
  const bool do_stats = RT ? (a.stat_sum != nullptr) : bool(EPI & 1);
  const bool has_bias = RT ? (a.bias != nullptr) : bool(EPI & 2);
  const bool silu = RT ? bool(a.flags & SY11_EPI_SILU) : bool(EPI & 4);
  const bool accum = RT ? bool(a.flags & SY11_EPI_ACCUM) : bool(EPI & 8);
  const bool out32 = RT ? bool(a.flags & SY11_EPI_OUT_F32) : bool(EPI & 16);
  float ssum[NI], ssq[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) ssum[j] = ssq[j] = 0.f;
  constexpr int OSZ_C = (EPI >= 0 && (EPI & 16)) ? 4 : (int)sizeof(T);
  if (!RT && a.vec_out && BM * BN * OSZ_C <= NST * STAGE && bm0 + BM <= a.M && bn0 + BN <= a.N && a.debug != 5) {
    // Interior tile with a compile-time epilogue (the common case): same LDS transposition as below, but no per-value
    // row / channel bounds tests, compile-time row pitch (shifts instead of divisions) — about half the instructions
    constexpr int ROWB = BN * OSZ_C, CPR = ROWB / 16, EPC_O = 16 / OSZ_C;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int cl = wn * (BN / WN) + j * 32 + frow;
          float v = acc[i][j][e];
          if (EPI & 1) { ssum[j] += v; ssq[j] += v * v; }
          if (EPI & 2) v += a.bias[bn0 + cl];
          if (EPI & 4) v = silu_f(v);
          if (EPI & 16) *(float*)(smem + rl * ROWB + cl * 4) = v;
          else *(T*)(smem + rl * ROWB + cl * (int)sizeof(T)) = ElemTraits<T>::from_f(v);
        }
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < BM * CPR / 256; ++it) {
      const int idx = tid + it * 256;
      const int rl = idx / CPR, ch = idx % CPR;
      const int m = bm0 + rl;
      long obase;
      if (a.dense_out) {
        obase = (long)m * a.y_ld;
      } else {
        const int b = m / ohw, r = m - b * ohw, oy = r / a.OW, ox = r - oy * a.OW;
        obase = ((long)(b * a.OHF + oy * a.oy_mul + a.oy_add) * a.OWF + ox * a.ox_mul + a.ox_add) * a.y_ld;
      }
      uint4 v = *(const uint4*)(smem + rl * ROWB + ch * 16);
      unsigned char* gp = (unsigned char*)a.y + (obase + bn0 + ch * EPC_O) * OSZ_C;
      if (EPI & 8) {
        const uint4 o = *(const uint4*)gp;
        if (EPI & 16) {
          f32x4 x = __builtin_bit_cast(f32x4, v), y = __builtin_bit_cast(f32x4, o);
          v = __builtin_bit_cast(uint4, x + y);
        } else {
          typedef T vt8 __attribute__((ext_vector_type(16 / sizeof(T))));
          vt8 x = __builtin_bit_cast(vt8, v), y = __builtin_bit_cast(vt8, o);
#pragma unroll
          for (int q = 0; q < (int)(16 / sizeof(T)); ++q) x[q] = ElemTraits<T>::from_f(ElemTraits<T>::to_f(x[q]) + ElemTraits<T>::to_f(y[q]));
          v = __builtin_bit_cast(uint4, x);
        }
      }
      *(uint4*)gp = v;
    }
  } else if (a.vec_out && BM * BN * ((a.flags & SY11_EPI_OUT_F32) ? 4 : (int)sizeof(T)) <= NST * STAGE) {
    // Wide-store path: the tile goes through LDS (row-major [128][BN] in the OUTPUT type, reusing the dead stage
    // buffers) so that every lane stores 16 contiguous bytes of one pixel row instead of 64 scattered 2-byte stores.
    const int osz = out32 ? 4 : (int)sizeof(T);
    const int rowb = BN * osz;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        const bool rok = bm0 + rl < a.M;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int cl = wn * (BN / WN) + j * 32 + frow;
          const int n = bn0 + cl;
          float v = acc[i][j][e];
          if (do_stats && rok && n < a.N) { ssum[j] += v; ssq[j] += v * v; }
          if (has_bias && n < a.N) v += a.bias[n];
          if (silu) v = silu_f(v);
          if (out32) *(float*)(smem + rl * rowb + cl * 4) = v;
          else *(T*)(smem + rl * rowb + cl * (int)sizeof(T)) = ElemTraits<T>::from_f(v);
        }
      }
    __syncthreads();
    const int cpr = rowb / 16;                       // 16-byte chunks per tile row
    const int epc_o = 16 / osz;
    // four 16-byte chunks per thread per trip; an accumulating epilogue issues its four reads of y before the first add, so a
    // thread waits for ONE memory round trip per trip instead of one per chunk
    constexpr int U = 4;
    for (int base = tid; base < BM * cpr; base += 256 * U) {
      unsigned char* gp[U];
      uint4 o[U];
      int lofs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * 256;
        const int rl = idx / cpr, ch = idx - rl * cpr;
        const int m = bm0 + rl, n = bn0 + ch * epc_o;
        // N % epc_o == 0 is guaranteed by the host; debug 5 = tuner dry run
        const bool ok = idx < BM * cpr && m < a.M && n < a.N && a.debug != 5;
        long obase;
        if (a.dense_out) {
          obase = (long)m * a.y_ld;
        } else {
          const int b = m / ohw, r = m - b * ohw, oy = r / a.OW, ox = r - oy * a.OW;
          obase = ((long)(b * a.OHF + oy * a.oy_mul + a.oy_add) * a.OWF + ox * a.ox_mul + a.ox_add) * a.y_ld;
        }
        gp[u] = ok ? (unsigned char*)a.y + (obase + n) * osz : nullptr;
        lofs[u] = rl * rowb + ch * 16;
        if (accum && ok) o[u] = *(const uint4*)gp[u];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!gp[u]) continue;
        uint4 v = *(const uint4*)(smem + lofs[u]);
        if (accum) {
          if (out32) {
            f32x4 x = __builtin_bit_cast(f32x4, v), y = __builtin_bit_cast(f32x4, o[u]);
            v = __builtin_bit_cast(uint4, x + y);
          } else {
            typedef T vt8 __attribute__((ext_vector_type(16 / sizeof(T))));
            vt8 x = __builtin_bit_cast(vt8, v), y = __builtin_bit_cast(vt8, o[u]);
#pragma unroll
            for (int q = 0; q < (int)(16 / sizeof(T)); ++q) x[q] = ElemTraits<T>::from_f(ElemTraits<T>::to_f(x[q]) + ElemTraits<T>::to_f(y[q]));
            v = __builtin_bit_cast(uint4, x);
          }
        }
        *(uint4*)gp[u] = v;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = bm0 + wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (m >= a.M || a.debug == 5) continue;
        long obase;
        if (a.dense_out) {
          obase = (long)m * a.y_ld;
        } else {
          const int b = m / ohw, r = m - b * ohw, oy = r / a.OW, ox = r - oy * a.OW;
          obase = ((long)(b * a.OHF + oy * a.oy_mul + a.oy_add) * a.OWF + ox * a.ox_mul + a.ox_add) * a.y_ld;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = bn0 + wn * (BN / WN) + j * 32 + frow;
          if (n >= a.N) continue;
          float v = acc[i][j][e];
          if (do_stats) { ssum[j] += v; ssq[j] += v * v; }
          if (has_bias) v += a.bias[n];
          if (silu) v = silu_f(v);
          if (out32) {
            float* yp = (float*)a.y + obase + n;
            if (accum) v += *yp;
            *yp = v;
          } else {
            T* yp = (T*)a.y + obase + n;
            if (accum) v += ElemTraits<T>::to_f(*yp);
            *yp = ElemTraits<T>::from_f(v);
          }
        }
      }
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
    if (a.tail.ticket) bn_tail_run(a.tail, a.stat_sum, a.stat_sq, a.stat_slots, a.stat_stride, gridDim.x);
  }
}

// ------------------------------------------------------------------------------------------------ host side
// cfg: 0 = 128x128, 1 = 128x64, 2 = 128x32, 3 = 256x128 (pixels x channels per workgroup), all with 64-byte K stages;
// 4..6 = the first three with 128-byte K stages (half the barriers per K, 2 workgroups per CU instead of 4)
// 7, 8 = persistent 1x1 kernel (igemm1x1.hip) with 128 / 64 channels per workgroup
// 9..11 = the first three with a 4-deep ring of 64-byte stages (3 stages in flight); 12..14 = 128-byte stages, 3-deep ring
// 15, 16 = halo-tiled 3x3 kernel (conv3x3.hip) with the widest / the next narrower channel tile
// 17, 18 = the same with 256-pixel tiles (stride 1): half the filter bytes per FLOP
// 19 = few-channel 3x3 stride-1 kernel (conv3x3s.hip): C = 16 / 32, N <= 32, the filter resident in registers
// 20, 21 = 8-wave software pipeline (igemm8.hip): 256 pixels x 128 / 64 channels, 128-byte K stages in 3 LDS slots, one workgroup per CU
// 22, 23 = the same with 64-byte K stages in 6 slots (twice the prefetch distance in cycles)
// 24 = 256 pixels x 256 channels (wave tile 128 x 64): 64 FLOP per filled byte instead of 43
// 25 = configuration 20 as PERSISTENT workgroups (igemm8p.hip): the next tile's first stages in flight under the current tile's epilogue
constexpr int IGEMM_NCFG = SY11_IGEMM_NCFG;
static_assert(IGEMM_NCFG == 26, "configuration table and its size (tune.h) out of step");
bool sy11_igemm8_legal(const IgemmArgs& a, int bn, int kb, int epi);
int sy11_igemm8_launch(const IgemmArgs& a, int bn, int kb, int epi, hipStream_t st);
bool sy11_igemm8p_legal(const IgemmArgs& a, int epi);
int sy11_igemm8p_launch(const IgemmArgs& a, int epi, hipStream_t st);
static int halo_bn(const IgemmArgs& a, int cfg) {
  const int wide = a.N > 64 ? 128 : (a.N > 32 ? 64 : 32);
  const int bn = (cfg == 15 || cfg == 17) ? wide : (wide > 32 ? wide / 2 : 0);
  return bn == 0 ? 0 : (cfg >= 17 ? 1000 + bn : bn);
}
int sy11_igemm1x1p_launch(int dtype, const void* x, const void* w, void* y, float* stat_sum, float* stat_sq, int M, int N, int K, int x_ld,
                          int y_ld, int stat_slots, int stat_stride, unsigned x_bytes, unsigned w_bytes, int epi, int nostore, int bn,
                          hipStream_t st, const float* bias);
static int epi_code(const IgemmArgs& a) {
  return (a.stat_sum ? 1 : 0) | (a.bias ? 2 : 0) | ((a.flags & SY11_EPI_SILU) ? 4 : 0) | ((a.flags & SY11_EPI_ACCUM) ? 8 : 0) |
         ((a.flags & SY11_EPI_OUT_F32) ? 16 : 0);
}
template <typename T>
static bool cfg_legal(const IgemmArgs& a, int cfg) {
  if (cfg < 0 || cfg >= IGEMM_NCFG) return false;
  if (cfg >= 15 && cfg <= 18) {
    const int bn = halo_bn(a, cfg);
    return std::is_same<T, _Float16>::value && bn > 0 && sy11_halo3x3_legal(a, bn);
  }
  if (cfg == 19) return !std::is_same<T, float>::value && sy11_smallc3x3_legal(a);
  if (cfg == 25) return std::is_same<T, _Float16>::value && sy11_igemm8p_legal(a, epi_code(a));
  if (cfg >= 20 && cfg <= 24) return std::is_same<T, _Float16>::value && sy11_igemm8_legal(a, cfg == 24 ? 256 : ((cfg & 1) ? 64 : 128), cfg < 22 ? 128 : 64, epi_code(a));
  if (cfg == 7 || cfg == 8) {
    const int epi = epi_code(a), bn = cfg == 7 ? 128 : 64;
    if (std::is_same<T, float>::value || a.T != 1 || a.tap_dy[0] || a.tap_dx[0] || a.sy != 1 || a.sx != 1 || !a.dense_out || !a.vec_out) return false;
    if ((epi != 0 && epi != 1 && epi != 8 && epi != 6) || a.K % 32 || a.K != a.C || a.wK != a.K || a.N % 8 || a.debug) return false;
    if (cfg == 7 && a.N <= 64) return false;                 // 128 channels per workgroup only when there are that many
    if (cfg == 8 && a.N <= 32) return false;
    return (size_t)(a.K / 32) * 64 * (bn + 256) + (size_t)128 * bn * 2 <= 150 * 1024;
  }
  if (cfg >= 9) {                                      // deep rings: f16, the training / inference epilogues only (code size)
    const int epi = epi_code(a);
    if (!std::is_same<T, _Float16>::value || (epi != 0 && epi != 1 && epi != 8 && epi != 6)) return false;
    return cfg >= 12 ? a.K >= 384 : a.K >= 128;        // at least as many stages as the ring is deep
  }
  if (cfg >= 4) return a.K >= 256;                     // long stages only pay with enough K to amortise them
  if (cfg != 3) return true;
  const int epi = (a.stat_sum ? 1 : 0) | (a.bias ? 2 : 0) | ((a.flags & SY11_EPI_SILU) ? 4 : 0) | ((a.flags & SY11_EPI_ACCUM) ? 8 : 0) |
                  ((a.flags & SY11_EPI_OUT_F32) ? 16 : 0);
  return std::is_same<T, _Float16>::value && (epi == 0 || epi == 1 || epi == 8 || epi == 6) && a.N > 64 && a.M >= 256;   // 6 = the fused inference conv (bias + SiLU)
}

template <typename T>
static int launch_cfg(IgemmArgs a, hipStream_t st, int cfg) {
  if (cfg >= 15 && cfg <= 18) return sy11_halo3x3_launch(a, halo_bn(a, cfg), st);
  if (cfg == 19) return sy11_smallc3x3_launch(a, ElemTraits<T>::code, st);
  if (cfg == 7 || cfg == 8)
    return sy11_igemm1x1p_launch(ElemTraits<T>::code, a.x, a.w, a.y, a.stat_sum, a.stat_sq, a.M, a.N, a.K, a.x_ld, a.y_ld, a.stat_slots,
                                 a.stat_stride, a.x_bytes, a.w_bytes, epi_code(a), a.debug == 5 ? 1 : 0, cfg == 7 ? 128 : 64, st, a.bias);
  const bool wave8 = cfg >= 20 && cfg <= 25;
  const int bm = (cfg == 3 || wave8) ? 256 : 128;
  const int tile = cfg == 25 ? 0 : wave8 ? (cfg & 1) : (cfg >= 12 ? cfg - 12 : (cfg >= 9 ? cfg - 9 : (cfg >= 4 ? cfg - 4 : cfg)));
  const int bn = cfg == 24 ? 256 : (tile == 1 ? 64 : (tile == 2 ? 32 : 128));
  a.tiles_n = cdiv(a.N, bn);
  const long nwg = (long)cdiv(a.M, bm) * a.tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffL) SY11_FAIL(SY11_EINVAL, "igemm: bad grid %ld", nwg);
  dim3 grid((unsigned)nwg), block(256);
  // ordered mode (det.h): statistics go to one partial row per workgroup (slot = workgroup index), folded in index order afterwards
  DetPartials dp;
  float* const stat_sum_out = a.stat_sum;
  float* const stat_sq_out = a.stat_sq;
  const int stat_slots_out = a.stat_slots, stat_stride_out = a.stat_stride;
  const bool det = a.stat_sum && sy11_det(1) && !a.tail.ticket;
  if (det) {
    if (!dp.acquire(st, 2, nwg, a.N)) SY11_FAIL(SY11_ELAUNCH, "igemm: ordered-reduction workspace unavailable (%ld x %d floats)", nwg, a.N);
    a.stat_sum = dp.buf(0); a.stat_sq = dp.buf(1); a.stat_slots = (int)nwg; a.stat_stride = a.N;
  }
  // 0 = 128-byte stages x2 (2 WG/CU), 1 = 64-byte stages x2 (4 WG/CU), 2 = 64-byte stages x4, 3 = 128-byte stages x3
  const int variant = cfg >= 12 ? 3 : (cfg >= 9 ? 2 : (cfg >= 4 ? 0 : 1));
  // epilogue specialisation: the common flag sets get branch-free code, anything else the runtime-flag build (EPI = -1)
  int epi = (a.stat_sum ? 1 : 0) | (a.bias ? 2 : 0) | ((a.flags & SY11_EPI_SILU) ? 4 : 0) | ((a.flags & SY11_EPI_ACCUM) ? 8 : 0) |
            ((a.flags & SY11_EPI_OUT_F32) ? 16 : 0);
  const int epi_pre = epi;
  if (sizeof(T) == 2 && !std::is_same<T, _Float16>::value) epi = -1;          // bf16: generic build only
#define SY11_IGV(BNN, WMM, WNN, EE)                                                                                  \
  do {                                                                                                               \
    if (variant == 0) hipLaunchKernelGGL((igemm_kernel<T, 128, BNN, WMM, WNN, 128, 2, EE>), grid, block, 0, st, a);  \
    else if (variant == 1) hipLaunchKernelGGL((igemm_kernel<T, 128, BNN, WMM, WNN, 64, 2, EE>), grid, block, 0, st, a); \
    else if constexpr (std::is_same<T, _Float16>::value && (EE == 0 || EE == 1 || EE == 8 || EE == 6)) {            \
      if (variant == 2) hipLaunchKernelGGL((igemm_kernel<T, 128, BNN, WMM, WNN, 64, 4, EE>), grid, block, 0, st, a);  \
      else hipLaunchKernelGGL((igemm_kernel<T, 128, BNN, WMM, WNN, 128, 3, EE>), grid, block, 0, st, a);             \
    }                                                                                                                \
  } while (0)
#define SY11_IG(BNN, WMM, WNN)                                   \
  do {                                                           \
    switch (epi) {                                               \
      case 0: SY11_IGV(BNN, WMM, WNN, 0); break;                 \
      case 1: SY11_IGV(BNN, WMM, WNN, 1); break;                 \
      case 8: SY11_IGV(BNN, WMM, WNN, 8); break;                 \
      case 2: SY11_IGV(BNN, WMM, WNN, 2); break;                 \
      case 6: SY11_IGV(BNN, WMM, WNN, 6); break;                 \
      case 18: SY11_IGV(BNN, WMM, WNN, 18); break;               \
      default: SY11_IGV(BNN, WMM, WNN, -1); break;               \
    }                                                            \
  } while (0)
  if (wave8) {
    const int rc8 = cfg == 25 ? sy11_igemm8p_launch(a, epi_pre, st) : sy11_igemm8_launch(a, bn, cfg < 22 ? 128 : 64, epi_pre, st);
    if (rc8) return rc8;
  } else if (bm == 256) {
    if constexpr (std::is_same<T, _Float16>::value) {
      if (epi_pre == 0) hipLaunchKernelGGL((igemm_kernel<T, 256, 128, 2, 2, 64, 3, 0>), grid, block, 0, st, a);
      else if (epi_pre == 1) hipLaunchKernelGGL((igemm_kernel<T, 256, 128, 2, 2, 64, 3, 1>), grid, block, 0, st, a);
      else if (epi_pre == 6) hipLaunchKernelGGL((igemm_kernel<T, 256, 128, 2, 2, 64, 3, 6>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((igemm_kernel<T, 256, 128, 2, 2, 64, 3, 8>), grid, block, 0, st, a);
    }
  } else if (bn == 128) SY11_IG(128, 2, 2);
  else if (bn == 64) SY11_IG(64, 4, 1);
  else SY11_IG(32, 4, 1);
#undef SY11_IGV
#undef SY11_IG
  SY11_LAUNCH_CHECK("igemm");
  if (det) {
    return dp.fold01(stat_sum_out, stat_sq_out, stat_slots_out, stat_stride_out);
  }
  return SY11_OK;
}

template <typename T>
static int select_and_launch(IgemmArgs& a, hipStream_t st) {
  static int dbg = -1;
  if (dbg < 0) { const char* e = getenv("SY11_IGEMM_DEBUG"); dbg = e ? atoi(e) : 0; }
  const int forced = sy11_opt(OPT_IGEMM_CFG);
  a.debug = dbg;
  a.chan_major = sy11_opt(OPT_IGEMM_KORDER);
  a.bpol = sy11_opt(OPT_IGEMM_BPOL);
  // static heuristic: widest channel tile the layer fills; small maps (20x20 / 40x40) narrow it until the grid covers the chip
  int bn = a.N > 64 ? 128 : (a.N > 32 ? 64 : 32);
  while (bn > 32 && (long)cdiv(a.M, 128) * cdiv(a.N, bn) < 512) bn >>= 1;
  int cfg = bn == 128 ? 0 : (bn == 64 ? 1 : 2);
  // 3x3 stride 1: the halo-tiled kernel moves 1.7-2.3x fewer bytes into LDS and won every such layer of yolo11s in the r02 sweeps
  // (-20..45 %); stride 2 is a wash (the tuner decides).  Narrower channel tile when the wide one leaves CUs without a workgroup.
  if (a.sy == 1 && a.T == 9 && cfg_legal<T>(a, 15)) {
    int th = 8, tw = 16;
    if (a.OW == 20) { th = 6; tw = 20; } else if (a.OW == 40) { th = 3; tw = 40; }
    const long wgs = (long)(a.M / (a.OH * a.OW)) * cdiv(a.OH, th) * cdiv(a.OW, tw) * cdiv(a.N, halo_bn(a, 15) % 1000);
    cfg = (wgs < 400 && cfg_legal<T>(a, 16)) ? 16 : 15;
  }
  // few channels on both sides (160x160 16 <-> 32): the register-resident-filter kernel won every such layer (r03: 50-61 us against 88-97)
  if (cfg_legal<T>(a, 19)) cfg = 19;
  if (forced >= 0 && cfg_legal<T>(a, forced)) return launch_cfg<T>(a, st, forced);
  if (dbg == 0) {                    // a recorded / imported pick is honoured even with measuring off ("tune" 0 only stops NEW measurements)
    sy11tune::Cache& cache = sy11tune::cache(0);
    const int key[] = {(int)sizeof(T), a.M, a.N, a.K, a.C, a.T, a.sy, a.sx, a.IW, a.OW, a.x_ld, a.y_ld, a.dense_out,
                       (int)(a.flags & SY11_EPI_OUT_F32)};
    const uint64_t h = sy11tune::hash(key, (int)(sizeof(key) / sizeof(int)));
    int hit;
    if (cache.get(h, &hit)) {
      if (cfg_legal<T>(a, hit)) cfg = hit;
    } else if (sy11tune::enabled() && !sy11tune::capturing(st)) {
      int cands[IGEMM_NCFG], nc = 0;
      for (int c = 0; c < IGEMM_NCFG; ++c) {
        const int ct = c == 25 ? 0 : c >= 20 ? (c & 1) : (c >= 12 ? c - 12 : (c >= 9 ? c - 9 : (c >= 7 ? (c == 7 ? 0 : 1) : (c >= 4 ? c - 4 : c))));
        const int cbn = c == 24 ? 256 : (ct == 1 ? 64 : (ct == 2 ? 32 : 128));
        if ((c < 15 || c >= 20) && cbn > 32 && cbn >= 2 * a.N) continue;              // tile more than twice the channel count: pure waste
        if (cfg_legal<T>(a, c)) cands[nc++] = c;
      }
      // measuring must leave no trace: no BN statistics; an accumulating epilogue runs with its global stores disabled
      IgemmArgs t = a;
      t.stat_sum = t.stat_sq = nullptr;
      t.tail.ticket = nullptr;
      if (t.flags & SY11_EPI_ACCUM) t.debug = 5;
      const int best = sy11tune::pick(cands, nc, [&](int c) { return launch_cfg<T>(t, st, c); }, st, "igemm", key,
                                      (int)(sizeof(key) / sizeof(int)));
      if (best >= 0) { cache.put(h, best); cfg = best; }
    }
  }
  // ring depth.  The tuner times a problem back to back on hot caches, where a stage's DMA returns from the local L2 in a few
  // hundred cycles and ring depth does not matter; inside the model every layer reads what ANOTHER kernel (other XCDs) just
  // wrote, a stage is a fabric round trip, and the small maps (1..3 workgroups per CU) have nothing else to hide it with.
  const int deep = sy11_opt(OPT_IGEMM_DEEP);
  if (deep > 0) {
    const long nwg = (long)cdiv(a.M, 128) * cdiv(a.N, cfg % 4 == 1 ? 64 : (cfg % 4 == 2 ? 32 : 128));
    if (deep >= 2 || nwg <= 1024) {
      const int up = cfg <= 2 ? cfg + 9 : ((cfg >= 4 && cfg <= 6) ? cfg + 8 : -1);
      if (up >= 0 && cfg_legal<T>(a, up)) cfg = up;
    }
  }
  return launch_cfg<T>(a, st, cfg);
}

template <typename T>
static int launch_igemm(IgemmArgs& a, hipStream_t st) {
  {
    const long esz = (long)sizeof(T);
    const long rows_in = (long)(a.M / (a.OH * a.OW)) * a.IH * a.IW;
    const long xb = ((rows_in - 1) * a.x_ld + a.C) * esz, wb = (long)a.N * a.wK * esz;
    if (xb >= (1L << 31) || wb >= (1L << 31)) SY11_FAIL(SY11_EUNSUPPORTED, "igemm: operand view larger than 2 GiB (%ld / %ld bytes)", xb, wb);
    a.x_bytes = (unsigned)xb;
    a.w_bytes = (unsigned)wb;
  }
  {
    const int osz = (a.flags & SY11_EPI_OUT_F32) ? 4 : (int)sizeof(T);
    a.vec_out = (((uintptr_t)a.y & 15) == 0) && (((long)a.y_ld * osz) % 16 == 0) && (((long)a.N * osz) % 16 == 0);
  }
  return select_and_launch<T>(a, st);
}

static int validate_conv(const sy11_conv_desc* d, const char* who) {
  SY11_REQUIRE(d != nullptr, "%s: null desc", who);
  SY11_REQUIRE(dtype_ok(d->dtype), "%s: bad dtype %d", who, d->dtype);
  SY11_REQUIRE(d->B > 0 && d->IH > 0 && d->IW > 0 && d->C > 0 && d->N > 0, "%s: non-positive dims", who);
  SY11_REQUIRE(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= 64, "%s: kernel %dx%d unsupported (<=64 taps)", who, d->KH, d->KW);
  SY11_REQUIRE(d->SH > 0 && d->SW > 0 && d->DH > 0 && d->DW > 0 && d->PH >= 0 && d->PW >= 0, "%s: bad stride/dilation/pad", who);
  const int oh = (d->IH + 2 * d->PH - d->DH * (d->KH - 1) - 1) / d->SH + 1;
  const int ow = (d->IW + 2 * d->PW - d->DW * (d->KW - 1) - 1) / d->SW + 1;
  SY11_REQUIRE(oh == d->OH && ow == d->OW, "%s: OH/OW (%d,%d) do not match conv arithmetic (%d,%d)", who, d->OH, d->OW, oh, ow);
  SY11_REQUIRE(d->x_ld >= d->C && d->y_ld >= d->N, "%s: pixel stride smaller than channel count", who);
  SY11_REQUIRE((long)d->B * d->IH * d->IW < (1L << 31) && (long)d->B * d->OH * d->OW < (1L << 31), "%s: pixel count overflows int32", who);
  SY11_REQUIRE(d->PH * 1 < 120 && d->DH * (d->KH - 1) < 120 && d->PW < 120 && d->DW * (d->KW - 1) < 120, "%s: tap offset exceeds int8", who);
  return SY11_OK;
}

static int check_align(const void* p, int ld, int c, int esz, const char* who, const char* what) {
  SY11_REQUIRE(p != nullptr, "%s: null %s", who, what);
  SY11_REQUIRE(((uintptr_t)p & 15) == 0, "%s: %s pointer not 16-byte aligned", who, what);
  SY11_REQUIRE(((long)ld * esz) % 16 == 0, "%s: %s pixel stride %d not a multiple of 16 bytes", who, what, ld);
  SY11_REQUIRE(((long)c * esz) % 16 == 0, "%s: %s channel count %d not a multiple of 16 bytes", who, what, c);
  return SY11_OK;
}

int sy11_dwconv_fwd_impl(const sy11_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                         float* stat_sum, float* stat_sq, hipStream_t st);
int sy11_dwconv_dgrad_impl(const sy11_conv_desc* d, const void* dy, int dy_ld, const void* w, void* dx, hipStream_t st);

// grouped convolution (1 < groups < C, e.g. the fusion variant's DDWConv g = 8, conv.py:694-710): every group is an
// independent dense convolution on a channel slice of x / y (pixel strides unchanged) and a row block of the filter
static int group_dims(const sy11_conv_desc* d, const char* who, int* cg, int* ng) {
  SY11_REQUIRE(d->groups > 1 && d->C % d->groups == 0 && d->N % d->groups == 0, "%s: channels not divisible by groups=%d", who, d->groups);
  *cg = d->C / d->groups;
  *ng = d->N / d->groups;
  return SY11_OK;
}

static int conv2d_fwd_impl(const sy11_conv_desc* d, const void* x, const void* w, const float* bias, void* y, float* stat_sum,
                           float* stat_sq, int stat_stride, hipStream_t st, const BnTailDev& tail = BnTailDev{});

extern "C" int sy11_conv2d_fwd(const sy11_conv_desc* d, const void* x, const void* w, const float* bias, void* y,
                               float* stat_sum, float* stat_sq, void* stream) {
  int rc = validate_conv(d, "conv2d_fwd");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (d->groups == 1) return conv2d_fwd_impl(d, x, w, bias, y, stat_sum, stat_sq, d->N, st);
  if (d->groups == d->C && d->C == d->N) return sy11_dwconv_fwd_impl(d, x, w, bias, y, stat_sum, stat_sq, st);
  int cg, ng;
  if ((rc = group_dims(d, "conv2d_fwd", &cg, &ng))) return rc;
  SY11_REQUIRE(x && w && y, "conv2d_fwd: null pointer");
  const int esz = dtype_size(d->dtype), osz = (d->flags & SY11_EPI_OUT_F32) ? 4 : esz;
  sy11_conv_desc dg = *d;
  dg.groups = 1; dg.C = cg; dg.N = ng;
  for (int g = 0; g < d->groups; ++g) {
    rc = conv2d_fwd_impl(&dg, (const char*)x + (long)g * cg * esz, (const char*)w + (long)g * ng * d->KH * d->KW * cg * esz,
                         bias ? bias + g * ng : nullptr, (char*)y + (long)g * ng * osz, stat_sum ? stat_sum + g * ng : nullptr,
                         stat_sq ? stat_sq + g * ng : nullptr, d->N, st);
    if (rc) return rc;
  }
  return SY11_OK;
}

int sy11_dwconv_fwd_bn_impl(const sy11_conv_desc* d, const void* x, const void* w, void* y, float* stat_sum, float* stat_sq,
                            const sy11_bn_tail* bn, hipStream_t st);

extern "C" int sy11_conv2d_fwd_bn(const sy11_conv_desc* d, const void* x, const void* w, void* y, float* stat_sum, float* stat_sq,
                                  const sy11_bn_tail* bn, void* stream) {
  int rc = validate_conv(d, "conv2d_fwd_bn");
  if (rc) return rc;
  SY11_REQUIRE(stat_sum && stat_sq && bn && bn->gamma && bn->beta && bn->mean && bn->rstd && bn->scale && bn->shift && bn->ticket &&
                   bn->count > 0, "conv2d_fwd_bn: statistics rows and a complete sy11_bn_tail are required");
  SY11_REQUIRE((bn->running_mean == nullptr) == (bn->running_var == nullptr), "conv2d_fwd_bn: running stats must both be given or both NULL");
  hipStream_t st = (hipStream_t)stream;
  if (d->groups == 1) return conv2d_fwd_impl(d, x, w, nullptr, y, stat_sum, stat_sq, d->N, st, bn_tail_dev(bn, d->N, 0, 0));
  if (d->groups == d->C && d->C == d->N) return sy11_dwconv_fwd_bn_impl(d, x, w, y, stat_sum, stat_sq, bn, st);
  int cg, ng;
  if ((rc = group_dims(d, "conv2d_fwd_bn", &cg, &ng))) return rc;
  SY11_REQUIRE(x && w && y, "conv2d_fwd_bn: null pointer");
  const int esz = dtype_size(d->dtype);
  sy11_conv_desc dg = *d;
  dg.groups = 1; dg.C = cg; dg.N = ng;
  for (int g = 0; g < d->groups; ++g) {
    rc = conv2d_fwd_impl(&dg, (const char*)x + (long)g * cg * esz, (const char*)w + (long)g * ng * d->KH * d->KW * cg * esz, nullptr,
                         (char*)y + (long)g * ng * esz, stat_sum + g * ng, stat_sq + g * ng, d->N, st, bn_tail_dev(bn, ng, g * ng, g));
    if (rc) return rc;
  }
  return SY11_OK;
}

static int conv2d_fwd_impl(const sy11_conv_desc* d, const void* x, const void* w, const float* bias, void* y, float* stat_sum,
                           float* stat_sq, int stat_stride, hipStream_t st, const BnTailDev& tail) {
  int rc;
  const int esz = dtype_size(d->dtype);
  if ((rc = check_align(x, d->x_ld, d->C, esz, "conv2d_fwd", "x"))) return rc;
  SY11_REQUIRE(w && y, "conv2d_fwd: null w/y");
  SY11_REQUIRE(((uintptr_t)w & 15) == 0, "conv2d_fwd: w not 16-byte aligned");
  SY11_REQUIRE((stat_sum == nullptr) == (stat_sq == nullptr), "conv2d_fwd: stat_sum and stat_sq must both be given or both NULL");
  IgemmArgs a{};
  a.x = x; a.w = w; a.y = y; a.bias = bias; a.stat_sum = stat_sum; a.stat_sq = stat_sq;
  a.T = d->KH * d->KW; a.C = d->C; a.K = a.T * a.C; a.wK = a.K;
  a.M = d->B * d->OH * d->OW; a.N = d->N;
  a.x_ld = d->x_ld; a.y_ld = d->y_ld;
  a.IH = d->IH; a.IW = d->IW; a.OH = d->OH; a.OW = d->OW;
  a.sy = d->SH; a.sx = d->SW;
  a.OHF = d->OH; a.OWF = d->OW; a.oy_mul = a.ox_mul = 1; a.oy_add = a.ox_add = 0; a.dense_out = 1;
  a.flags = d->flags;
  a.stat_slots = d->stat_slots > 1 ? d->stat_slots : 1;
  a.stat_stride = stat_stride;
  a.tail = tail;
  for (int r = 0; r < d->KH; ++r)
    for (int s = 0; s < d->KW; ++s) {
      const int t = r * d->KW + s;
      a.tap_dy[t] = (signed char)(r * d->DH - d->PH);
      a.tap_dx[t] = (signed char)(s * d->DW - d->PW);
      a.tap_w[t] = (signed char)t;
    }
  SY11_DISPATCH_DTYPE(d->dtype, T, return launch_igemm<T>(a, st));
}

extern "C" int sy11_conv2d_dgrad(const sy11_conv_desc* d, const void* dy, int32_t dy_ld, const void* wt, void* dx,
                                 void* stream) {
  int rc = validate_conv(d, "conv2d_dgrad");
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (d->groups != 1 && d->groups == d->C && d->C == d->N) return sy11_dwconv_dgrad_impl(d, dy, dy_ld, wt, dx, st);
  if (d->groups != 1) {                 // grouped: wt = [groups][C/g][KH*KW][N/g] (each group's own tap-transposed block)
    int cg, ng;
    if ((rc = group_dims(d, "conv2d_dgrad", &cg, &ng))) return rc;
    SY11_REQUIRE(dy && wt && dx, "conv2d_dgrad: null pointer");
    const int esz = dtype_size(d->dtype);
    sy11_conv_desc dg = *d;
    dg.groups = 1; dg.C = cg; dg.N = ng;
    for (int g = 0; g < d->groups; ++g) {
      rc = sy11_conv2d_dgrad(&dg, (const char*)dy + (long)g * ng * esz, dy_ld, (const char*)wt + (long)g * cg * d->KH * d->KW * ng * esz,
                             (char*)dx + (long)g * cg * esz, stream);
      if (rc) return rc;
    }
    return SY11_OK;
  }
  const int esz = dtype_size(d->dtype);
  SY11_REQUIRE(dy_ld >= d->N, "conv2d_dgrad: dy_ld < N");
  if ((rc = check_align(dy, dy_ld, d->N, esz, "conv2d_dgrad", "dy"))) return rc;
  SY11_REQUIRE(wt && dx, "conv2d_dgrad: null wt/dx");
  SY11_REQUIRE(((uintptr_t)wt & 15) == 0, "conv2d_dgrad: wt not 16-byte aligned");
  SY11_REQUIRE(!(d->flags & (SY11_EPI_SILU | SY11_EPI_OUT_F32)), "conv2d_dgrad: only SY11_EPI_ACCUM is meaningful");
  // 3x3 / stride 2 / pad 1 in f16: all four parity classes in ONE pass over dy (conv3x3.hip halo_dgrad_s2_kernel), unless switched off
  if (d->dtype == SY11_F16 && d->KH == 3 && d->KW == 3 && d->SH == 2 && d->SW == 2 && d->PH == 1 && d->PW == 1 && d->DH == 1 && d->DW == 1 &&
      sy11_opt(OPT_IGEMM_CFG) < 0 && sy11_opt(OPT_DGRAD_S2_HALO) != 0) {
    // yolo11s at batch 64, plain / accumulating (tools/dgrad_s2_micro.py): 320x320x32 <- 160x160x64 394 -> 168 / 519 -> 220 us,
    // 160x160x128 <- 80x80x128 278 -> 218 / 427 -> 312, 80x80x256 <- 40x40x256 219 -> 186 / 261 -> 209, 40x40x256 <- 20x20x512
    // 104 -> 94 / 112 -> 101, the head's 80x80x128 87 -> 62 / 103 -> 77 and 40x40x256 72 -> 55 / 83 -> 63.  0 = the four igemm launches.
    IgemmArgs a{};
    a.x = dy; a.w = wt; a.y = dx;
    a.C = d->N; a.wK = 9 * d->N; a.N = d->C; a.K = 9 * d->N; a.T = 9;
    a.x_ld = dy_ld; a.y_ld = d->x_ld;
    a.IH = d->OH; a.IW = d->OW; a.OHF = d->IH; a.OWF = d->IW;
    a.M = d->B;
    a.flags = d->flags & SY11_EPI_ACCUM;
    const long xb = (((long)d->B * d->OH * d->OW - 1) * dy_ld + d->N) * 2, wb = (long)d->C * 9 * d->N * 2;
    if (xb < (1L << 31) && wb < (1L << 31)) {
      a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
      if (sy11_halo_dgrad_s2_legal(a)) return sy11_halo_dgrad_s2_launch(a, st);
    }
  }
  // one launch per output-parity class (ph, pw): pixels i = SH*i' + ph use only taps with (ph + PH - r*DH) % SH == 0
  for (int ph = 0; ph < d->SH; ++ph)
    for (int pw = 0; pw < d->SW; ++pw) {
      if (ph >= d->IH || pw >= d->IW) continue;
      IgemmArgs a{};
      a.x = dy; a.w = wt; a.y = dx; a.bias = nullptr;
      a.C = d->N; a.wK = d->KH * d->KW * d->N;
      a.N = d->C;
      a.x_ld = dy_ld; a.y_ld = d->x_ld;
      a.IH = d->OH; a.IW = d->OW;
      a.OH = (d->IH - ph + d->SH - 1) / d->SH; a.OW = (d->IW - pw + d->SW - 1) / d->SW;
      a.M = d->B * a.OH * a.OW;
      a.sy = 1; a.sx = 1;
      a.OHF = d->IH; a.OWF = d->IW; a.oy_mul = d->SH; a.oy_add = ph; a.ox_mul = d->SW; a.ox_add = pw;
      a.dense_out = (d->SH == 1 && d->SW == 1);
      a.flags = d->flags & SY11_EPI_ACCUM;
      a.stat_slots = 1;
      a.stat_stride = a.N;
      int t = 0;
      for (int r = 0; r < d->KH; ++r) {
        const int ny = ph + d->PH - r * d->DH;
        if (((ny % d->SH) + d->SH) % d->SH) continue;
        for (int s = 0; s < d->KW; ++s) {
          const int nx = pw + d->PW - s * d->DW;
          if (((nx % d->SW) + d->SW) % d->SW) continue;
          const int oy_off = (ny >= 0 ? ny / d->SH : -((-ny) / d->SH));   // exact: ny divisible by SH
          const int ox_off = (nx >= 0 ? nx / d->SW : -((-nx) / d->SW));
          a.tap_dy[t] = (signed char)oy_off;
          a.tap_dx[t] = (signed char)ox_off;
          a.tap_w[t] = (signed char)(r * d->KW + s);
          ++t;
        }
      }
      a.T = t;
      a.K = t * a.C;
      if (t == 0) {
        // no tap reaches this parity class: gradient is zero there
        if (!(d->flags & SY11_EPI_ACCUM)) SY11_FAIL(SY11_EUNSUPPORTED, "conv2d_dgrad: stride leaves untouched input pixels; zero dx and pass SY11_EPI_ACCUM");
        continue;
      }
      SY11_DISPATCH_DTYPE(d->dtype, T, rc = launch_igemm<T>(a, st));
      if (rc) return rc;
    }
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ weight transpose
template <typename T>
__global__ void weight_transpose_kernel(int N, int T_, int C, const T* __restrict__ w, T* __restrict__ wt) {
  // wt[c][t][n] = w[n][t][c]; 32x32 LDS tile transpose over (n, c) for each tap
  __shared__ T tile[32][33];
  const int t = blockIdx.z;
  const int n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int n = n0 + j, c = c0 + tx;
    tile[j][tx] = (n < N && c < C) ? w[((long)n * T_ + t) * C + c] : (T)0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, n = n0 + tx;
    if (c < C && n < N) wt[((long)c * T_ + t) * N + n] = tile[tx][j];
  }
}

extern "C" int sy11_weight_transpose(int32_t dtype, int32_t N, int32_t T_, int32_t C, const void* w, void* wt, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype), "weight_transpose: bad dtype");
  SY11_REQUIRE(N > 0 && T_ > 0 && C > 0 && T_ <= 65535, "weight_transpose: bad dims");
  SY11_REQUIRE(w && wt, "weight_transpose: null pointer");
  dim3 grid(cdiv(C, 32), cdiv(N, 32), T_), block(256);
  SY11_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((weight_transpose_kernel<T>), grid, block, 0, (hipStream_t)stream, N, T_, C, (const T*)w, (T*)wt));
  SY11_LAUNCH_CHECK("weight_transpose");
  return SY11_OK;
}

// ------------------------------------------------------------------------------------------------ batched weight transpose
// All dgrad filters of a model in ONE launch: desc[l] = {N, T, C, src_off, dst_off, first_tile} over flat buffers.
struct WtDesc { int N, T, C, pad; long src_off, dst_off; int first_tile, tiles_c, tiles_n, pad2; };

template <typename T>
__global__ __launch_bounds__(256) void weight_transpose_multi_kernel(int nlayers, const WtDesc* __restrict__ desc, const T* __restrict__ src,
                                                                     T* __restrict__ dst) {
  __shared__ T tile[32][33];
  const int bid = blockIdx.x;
  int lo = 0, hi = nlayers - 1;                   // last layer whose first_tile <= bid
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].first_tile <= bid) lo = mid; else hi = mid - 1;
  }
  const WtDesc d = desc[lo];
  int rel = bid - d.first_tile;
  const int per_tap = d.tiles_c * d.tiles_n;
  const int t = rel / per_tap;
  rel -= t * per_tap;
  const int n0 = (rel / d.tiles_c) * 32, c0 = (rel % d.tiles_c) * 32;
  const T* w = src + d.src_off;
  T* wt = dst + d.dst_off;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int n = n0 + j, c = c0 + tx;
    tile[j][tx] = (n < d.N && c < d.C) ? w[((long)n * d.T + t) * d.C + c] : (T)0;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, n = n0 + tx;
    if (c < d.C && n < d.N) wt[((long)c * d.T + t) * d.N + n] = tile[tx][j];
  }
}

extern "C" int sy11_weight_transpose_multi(int32_t dtype, int32_t nlayers, int32_t total_tiles, const void* desc, const void* src,
                                           void* dst, void* stream) {
  SY11_REQUIRE(dtype_ok(dtype) && nlayers > 0 && total_tiles > 0 && desc && src && dst, "weight_transpose_multi: bad argument");
  SY11_DISPATCH_DTYPE(dtype, T, hipLaunchKernelGGL((weight_transpose_multi_kernel<T>), dim3(total_tiles), dim3(256), 0, (hipStream_t)stream,
                                                   nlayers, (const WtDesc*)desc, (const T*)src, (T*)dst));
  SY11_LAUNCH_CHECK("weight_transpose_multi");
  return SY11_OK;
}
