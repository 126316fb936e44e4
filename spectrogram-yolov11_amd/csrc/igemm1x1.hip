// igemm1x1.hip — persistent-tile kernel for the memory-bound 1x1 convolutions (and their input gradients) on gfx950.
//
// A 1x1 / stride-1 conv is the plain GEMM  y[M x N] = x[M x K] * w[N x K]^T  with K = Cin <= 192: one or two K stages per
// 128-pixel tile.  In the generic igemm kernel such a tile is prologue -> 1..3 DMA stages -> 8..24 MFMAs -> epilogue, with
// nothing to overlap inside a workgroup, and every tile re-reads the filter.  Here a workgroup is PERSISTENT: it loads its
// filter slice (BN x K) into LDS once, then walks pixel tiles t, t + G, t + 2G, ... with the activations double-buffered:
// the LDS-DMA of tile t+1 is issued before the MFMAs of tile t and overlaps them and tile t's epilogue (LDS transposition,
// 16-byte stores); BatchNorm statistics are accumulated in registers over all tiles and reduced once per workgroup.
// LDS image = K/32 "planes" of [rows][64 bytes] (the stage layout of igemm.hip, same source-side XOR swizzle, same fragment
// reads), so the MFMA core is identical.  Selected per layer by the first-call autotuner (tune.h) as one more candidate.
#include "common.h"
#include "det.h"

typedef int p_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void p_lds_dma16(unsigned lds_addr, unsigned voff, p_rsrc_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ p_rsrc_t p_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  p_rsrc_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32) & 0xffff);
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
template <typename T> struct PMma;
template <> struct PMma<_Float16> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct PMma<__bf16> {
  static __device__ __forceinline__ void run(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

struct P1x1Args {
  const void* x;
  const void* w;          // [N][K] rows (dgrad: the tap-transposed filter [C][N])
  void* y;
  float* stat_sum;
  float* stat_sq;
  int M, N, K;
  int x_ld, y_ld;
  int tiles_m;
  int stat_slots, stat_stride;
  unsigned x_bytes, w_bytes;
  int nostore;            // tuner dry run of an accumulating epilogue
  const float* bias;      // EPI 6: y = silu(acc + bias) — the inference conv with BatchNorm folded in
};

// EPI: 0 plain store, 1 + BN statistics, 8 accumulate into y, 6 bias + SiLU
template <typename T, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) void igemm1x1p_kernel(const P1x1Args a) {
  constexpr int BM = 128, KB = 64, RPI = 16, RPP = 64;
  constexpr int MI = BM / WM / 32, NI = BN / WN / 32;
  constexpr int A_PLANE = BM * KB, B_PLANE = BN * KB;
  constexpr int APASS = BM / RPP, BPASS = (BN + RPP - 1) / RPP;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int ROWB = BN * 2, CPR = ROWB / 16;
  static_assert(WM * WN == 4 && MI >= 1 && NI >= 1, "wave layout");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float s_red[WM * 2 * BN];           // [wave row wm][sum | sumsq][channel]: ordered fold, no LDS atomics
  const int np = a.K >> 5;                                   // planes of 32 elements (64 bytes) of K
  unsigned char* sB = smem;                                  // [np][BN][64]
  unsigned char* sA = sB + np * B_PLANE;                     // [2][np][BM][64]
  unsigned char* sT = sA + 2 * np * A_PLANE;                 // [BM][BN] output staging (16-bit)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int bn0 = blockIdx.y * BN;

  const p_rsrc_t xr = p_make_rsrc(a.x, a.x_bytes), wr_ = p_make_rsrc(a.w, a.w_bytes);
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int ld_row = tid >> 2;
  const int ld_chunk = (tid & 3) ^ ((ld_row >> 2) & 3);      // logical chunk fetched by this lane (swizzle on the source side)
  const bool b_issue = (BN >= RPP) || (wave_u * RPI < BN);
  // ---- the filter slice, once
  for (int pl = 0; pl < np; ++pl) {
    if (b_issue) {
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        const int rl = ld_row + RPP * i, n = bn0 + rl;
        const unsigned off = (n < a.N && rl < BN) ? (unsigned)((n * a.K + pl * 32) * 2 + ld_chunk * 16) : OOB;
        p_lds_dma16(smem_base + pl * B_PLANE + wave_u * (RPI * KB) + i * (RPP * KB), off, wr_);
      }
    }
  }
  const int x_rowb = a.x_ld * 2;
  auto issue_a = [&](int tile, int buf) {
    const unsigned base = smem_base + np * B_PLANE + buf * np * A_PLANE + wave_u * (RPI * KB);
    unsigned ro[APASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int m = tile * BM + ld_row + RPP * i;
      ro[i] = m < a.M ? (unsigned)(m * x_rowb + ld_chunk * 16) : OOB;
    }
    for (int pl = 0; pl < np; ++pl) {
#pragma unroll
      for (int i = 0; i < APASS; ++i) p_lds_dma16(base + pl * A_PLANE + i * (RPP * KB), ro[i] == OOB ? OOB : ro[i] + pl * 64, xr);
    }
  };

  const int wm = wave / WN, wn = wave % WN;
  const int frow = lane & 31, fh = lane >> 5;
  int fa_off[MI][2], fb_off[NI][2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r = wm * (BM / WM) + i * 32 + frow;
      fa_off[i][g] = r * KB + (((2 * g + fh) ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int r = wn * (BN / WN) + j * 32 + frow;
      fb_off[j][g] = r * KB + (((2 * g + fh) ^ ((r >> 2) & 3)) << 4);
    }
  }
  float ssum[NI], ssq[NI];
  float bias_r[NI];                                    // EPI 6: this lane's output channels never change across tiles
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = bn0 + wn * (BN / WN) + j * 32 + frow;
    bias_r[j] = (EPI == 6 && n < a.N) ? a.bias[n] : 0.f;
  }
#pragma unroll
  for (int j = 0; j < NI; ++j) ssum[j] = ssq[j] = 0.f;

  int tile = blockIdx.x;
  if (tile < a.tiles_m) issue_a(tile, 0);
  for (int it = 0; tile < a.tiles_m; tile += gridDim.x, ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this tile's planes (and, first time, the filter) have landed
    __builtin_amdgcn_s_barrier();                            // ... for every wave; everyone is also done with the previous epilogue
    const int nxt = tile + gridDim.x;
    if (nxt < a.tiles_m) issue_a(nxt, (it + 1) & 1);         // overlaps the math and the epilogue below
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const unsigned char* pa = sA + (it & 1) * np * A_PLANE;
    for (int pl = 0; pl < np; ++pl) {
      const unsigned char* sa = pa + pl * A_PLANE;
      const unsigned char* sb = sB + pl * B_PLANE;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        uint4 fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[i] = *(const uint4*)(sa + fa_off[i][g]);
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[j] = *(const uint4*)(sb + fb_off[j][g]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) PMma<T>::run(fa[i], fb[j], acc[i][j]);
      }
    }
    // ---- epilogue of this tile: accumulator layout -> row-major 16-bit image in LDS -> 16-byte stores
    const int bm0 = tile * BM;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rl = wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int cl = wn * (BN / WN) + j * 32 + frow;
          float v = acc[i][j][e];
          if (EPI == 6) v = silu_f(v + bias_r[j]);
          if (EPI & 1) { ssum[j] += v; ssq[j] += v * v; }    // rows >= M and channels >= N are exact zeros (zero-filled operands)
          *(T*)(sT + rl * ROWB + cl * 2) = ElemTraits<T>::from_f(v);
        }
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BM * CPR / 256; ++k) {
      const int idx = tid + k * 256;
      const int rl = idx / CPR, ch = idx % CPR;
      const int m = bm0 + rl, n = bn0 + ch * 8;
      if (m < a.M && n < a.N && !a.nostore) {
        uint4 v = *(const uint4*)(sT + rl * ROWB + ch * 16);
        T* gp = (T*)a.y + (long)m * a.y_ld + n;
        if (EPI & 8) {
          typedef T vt8 __attribute__((ext_vector_type(8)));
          vt8 x = __builtin_bit_cast(vt8, v), y = *(const vt8*)gp;
#pragma unroll
          for (int q = 0; q < 8; ++q) x[q] = ElemTraits<T>::from_f(ElemTraits<T>::to_f(x[q]) + ElemTraits<T>::to_f(y[q]));
          v = __builtin_bit_cast(uint4, x);
        }
        *(uint4*)gp = v;
      }
    }
  }
  if (EPI & 1) {
    __syncthreads();
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
      const long so = (long)((blockIdx.x + blockIdx.y) % a.stat_slots) * a.stat_stride;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int r = 0; r < WM; ++r) { t1 += s_red[r * 2 * BN + tid]; t2 += s_red[r * 2 * BN + BN + tid]; }
      atomicAdd(a.stat_sum + so + bn0 + tid, t1);
      atomicAdd(a.stat_sq + so + bn0 + tid, t2);
    }
  }
}

// LDS bytes of a configuration (bn channels per workgroup, K elements)
static size_t p1x1_lds(int bn, int K) { return (size_t)(K / 32) * 64 * (bn + 256) + (size_t)128 * bn * 2; }

// returns SY11_EUNSUPPORTED when the problem does not fit this kernel (the caller then uses the generic igemm)
int sy11_igemm1x1p_launch(int dtype, const void* x, const void* w, void* y, float* stat_sum, float* stat_sq, int M, int N, int K, int x_ld,
                          int y_ld, int stat_slots, int stat_stride, unsigned x_bytes, unsigned w_bytes, int epi, int nostore, int bn,
                          hipStream_t st, const float* bias) {
  if (dtype == SY11_F32 || (epi != 0 && epi != 1 && epi != 8 && epi != 6) || (epi == 6 && !bias) || K % 32 || K < 32 || N % 8 || (bn != 128 && bn != 64 && bn != 32))
    SY11_FAIL(SY11_EUNSUPPORTED, "igemm1x1p: unsupported problem");
  const size_t lds = p1x1_lds(bn, K);
  if (lds > 150 * 1024) SY11_FAIL(SY11_EUNSUPPORTED, "igemm1x1p: K=%d needs %zu bytes of LDS", K, lds);
  P1x1Args a{x, w, y, stat_sum, stat_sq, M, N, K, x_ld, y_ld, cdiv(M, 128), stat_slots > 1 ? stat_slots : 1, stat_stride, x_bytes, w_bytes, nostore, bias};
  const int tiles_n = cdiv(N, bn);
  int gx = a.tiles_m < 256 ? a.tiles_m : 256;
  if (lds <= 72 * 1024 && a.tiles_m >= 1024) gx = 512;       // two resident workgroups per CU when LDS allows
  dim3 grid(gx, tiles_n), block(256);
  DetPartials dp;                                   // ordered mode (det.h): slot = x + y is unique per (workgroup row, channel tile)
  const bool det = stat_sum && (epi & 1) && sy11_det(1);
  if (det) {
    if (!dp.acquire(st, 2, gx + tiles_n, N)) SY11_FAIL(SY11_ELAUNCH, "igemm1x1p: ordered-reduction workspace unavailable (%d x %d floats)", gx + tiles_n, N);
    a.stat_sum = dp.buf(0); a.stat_sq = dp.buf(1); a.stat_slots = gx + tiles_n; a.stat_stride = N;
  }
#define SY11_P1(TT, BNN, WMM, WNN, EE)                                                                                           \
  do {                                                                                                                            \
    if (lds > 64 * 1024)                                                                                                          \
      (void)hipFuncSetAttribute((const void*)igemm1x1p_kernel<TT, BNN, WMM, WNN, EE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((igemm1x1p_kernel<TT, BNN, WMM, WNN, EE>), grid, block, lds, st, a);                                        \
  } while (0)
#define SY11_P1E(TT, BNN, WMM, WNN)                   \
  do {                                                \
    if (epi == 0) SY11_P1(TT, BNN, WMM, WNN, 0);      \
    else if (epi == 1) SY11_P1(TT, BNN, WMM, WNN, 1); \
    else if (epi == 6) SY11_P1(TT, BNN, WMM, WNN, 6); \
    else SY11_P1(TT, BNN, WMM, WNN, 8);               \
  } while (0)
#define SY11_P1T(TT)                              \
  do {                                            \
    if (bn == 128) SY11_P1E(TT, 128, 2, 2);       \
    else if (bn == 64) SY11_P1E(TT, 64, 4, 1);    \
    else SY11_P1E(TT, 32, 4, 1);                  \
  } while (0)
  if (dtype == SY11_F16) SY11_P1T(_Float16);
  else SY11_P1T(__bf16);
#undef SY11_P1T
#undef SY11_P1E
#undef SY11_P1
  SY11_LAUNCH_CHECK("igemm1x1p");
  if (det) {
    return dp.fold01(stat_sum, stat_sq, stat_slots > 1 ? stat_slots : 1, stat_stride);
  }
  return SY11_OK;
}
