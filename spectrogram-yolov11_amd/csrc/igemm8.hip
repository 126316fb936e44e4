// igemm8.hip — the implicit-GEMM convolution of igemm.hip as an 8-wave, two-waves-per-SIMD, phase-staggered pipeline (r04).
//
// Why a second main loop.  The 4-wave tiles of igemm.hip run every wave through  barrier -> issue LDS-DMA -> read fragments -> MFMAs:
// a wave that is issuing its copies is not issuing MFMAs, the workgroups of a CU run in phase, and the r03 counters put the matrix pipe
// at 50-60 % busy even with the copies compiled out.  Here a workgroup is 512 threads = two wave GROUPS (waves 0-3 / 4-7: wave w and
// w + 4 share a SIMD).  Group 1 enters the loop one barrier late, and every phase is
//        L: fragment reads of this phase + this phase's share of the LDS-DMA pieces two stages ahead + (counted) waits
//        s_barrier
//        M: 8 x v_mfma_f32_32x32x16 between s_setprio 1 / 0
//        s_barrier
// so that while one group sits in M the other sits in L: each SIMD always has one wave feeding the matrix pipe and one wave feeding
// the address / LDS paths (cdna_hip_programming.md, "The 256^2 8-phase template": same synchronisation skeleton, with this kernel's
// im2col gather as the copy).  One workgroup per CU (148 KB of LDS), so the stagger is the ONLY overlap and it is deterministic.
//
// Geometry: 256 pixels x BN channels per workgroup (BN = 128: wave tile 64 x 64, or 64: wave tile 64 x 32), stages of 128 BYTES of K
// per row (64 f16: whole 128-byte lines from HBM / L2), a ring of three stages, two phases (K = 32 each) per stage.
//
// Ring discipline (global barrier numbers: group 0 runs L_p between B(2p-1) and B(2p), M_p between B(2p) and B(2p+1); group 1 one later):
//   RAW  stage s+1 is retired by EVERY wave's own counted s_waitcnt vmcnt in L_{2s+1} (leaving only stage s+2's pieces in flight) and is
//        first read in L_{2s+2}: at least one barrier after the last wave's wait, for both groups.
//   WAR  every wave drains its fragment reads (lgkmcnt(0)) BEFORE the barrier that ends its L section, so the last reads of stage s-1
//        (phase 2s-1) are complete before B(4s-1); its buffer is refilled by pieces issued in L_{2s} / L_{2s+1}, i.e. after B(4s-1).
// Each stage is exactly NPC pieces per wave (out-of-range rows are issued as zero-fill pieces), so the counts are exact.
#include "common.h"
#include "tune.h"
#include "det.h"
#include "bn_tail.h"
#include "igemm_args.h"

namespace {

typedef int rsrc8_t __attribute__((ext_vector_type(4)));
// one wave-wide LDS-DMA: 64 lanes x 16 bytes from (buffer descriptor + per-lane byte offset) to LDS at (wave-uniform address + lane * 16);
// out-of-range offsets deliver zeros.  Inline asm on purpose (see igemm.hip: the builtin makes the compiler drain vmcnt before LDS reads).
__device__ __forceinline__ void dma16(unsigned lds_addr, unsigned voff, rsrc8_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ rsrc8_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  rsrc8_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32) & 0xffff);
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// diagnostic build only (SY11_IGEMM_DEBUG=9): where a phase spends its cycles.  Wave 0 and wave 4 of workgroup 0 add up, over all
// phases, the s_memtime deltas of [fragment-read issue | piece issue | counted waits | barrier 1 | MFMA issue | barrier 2] + the phase
// count + the whole kernel; read back with sy11_debug_stamps().  Nothing else in the kernel reads this memory.
__device__ unsigned long long g_i8_stamp[2][8];
__device__ __forceinline__ unsigned long long stamp() { return __builtin_amdgcn_s_memtime(); }

}  // namespace

// EPI bits as in igemm.hip: 1 statistics, 2 bias, 4 SiLU, 8 accumulate.  Instantiated: 0, 1, 8, 6.
template <int BN, int EPI>
__global__ __launch_bounds__(512) void igemm8_kernel(const IgemmArgs a) {
  typedef _Float16 T;
  constexpr int BM = 256, KB = 128, ESZ = 2, EPC = 8, CPRW = 8, BK = 64, NST = 3;
  constexpr int RPI = 8, RPP = 64;                   // rows per wave-instruction, rows per pass of the 8 waves
  constexpr int APASS = BM / RPP, BPASS = BN / RPP;
  constexpr int MI = 2, NI = BN / 64;                // 32 x 32 blocks of a wave tile (64 rows x BN / 2 columns)
  constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB, STAGE = A_BYTES + B_BYTES;
  constexpr int NPC = APASS + BPASS;                 // LDS-DMA pieces per wave and stage
  constexpr int A_H0 = APASS / 2, B_H0 = (BPASS + 1) / 2;   // passes issued in phase 0 of a stage (the rest in phase 1)
  constexpr unsigned OOB = 0x80000000u;
  constexpr int TAB = NST * STAGE;                   // tap tables: [64] x delta, [64] w delta
  constexpr int RED = TAB + 512;                     // statistics fold: [4 row groups][sum | sumsq][BN]
  __shared__ __attribute__((aligned(16))) unsigned char smem[RED + 4 * 2 * BN * 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wm2 = (wave >> 1) & 1, wn = wave & 1;
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {                                                  // XCD-aware bijective tile order (workgroups b and b + 8 share an L2)
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tile_n = bid % a.tiles_n, tile_m = bid / a.tiles_n;
  const int bm0 = tile_m * BM, bn0 = tile_n * BN;

  // ---- copy side: lane -> (row within a pass, 16-byte chunk); the bank swizzle sits on the SOURCE side
  const int ld_row = tid >> 3;
  const int ld_chunk = (tid & 7) ^ ((ld_row >> 1) & 7);
  int* s_tapoff = (int*)(smem + TAB);
  if (tid < 64) {
    const int t = tid < a.T ? tid : 0;
    s_tapoff[tid] = (a.tap_dy[t] * a.IW + a.tap_dx[t]) * a.x_ld * ESZ;
    s_tapoff[64 + tid] = a.tap_w[t] * a.C * ESZ;
  }
  const int ohw = a.OH * a.OW;
  int a_off[APASS];
  unsigned long long a_mask[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int m = bm0 + ld_row + RPP * i;
    const bool ok = m < a.M;
    const int mm = ok ? m : 0;
    const int b = mm / ohw, r = mm - b * ohw, oy = r / a.OW, ox = r - oy * a.OW;
    const int iy0 = oy * a.sy, ix0 = ox * a.sx;
    a_off[i] = ((b * a.IH + iy0) * a.IW + ix0) * a.x_ld * ESZ;
    unsigned long long mk = 0;
    if (ok)
#pragma unroll 1
      for (int t = 0; t < a.T; ++t) {
        const int iy = iy0 + a.tap_dy[t], ix = ix0 + a.tap_dx[t];
        if ((unsigned)iy < (unsigned)a.IH && (unsigned)ix < (unsigned)a.IW) mk |= 1ull << t;
      }
    a_mask[i] = mk;
  }
  int b_off[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int n = bn0 + ld_row + RPP * i;
    b_off[i] = n < a.N ? n * a.wK * ESZ : (int)OOB;
  }
  struct KPos { int kt, kc, xo, wo; };               // (tap, channel) of this lane's chunk in the stage about to be issued + that tap's deltas
  KPos kp;
  kp.kt = (ld_chunk * EPC) / a.C;
  kp.kc = (ld_chunk * EPC) - kp.kt * a.C;
  const rsrc8_t xr = make_rsrc(a.x, a.x_bytes), wr = make_rsrc(a.w, a.w_bytes);
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const bool simple_k = a.C >= BK;
  const bool chan_major = a.chan_major && a.T > 1 && a.C % BK == 0;
  __syncthreads();                                   // tap tables visible
  {
    const int tt0 = kp.kt < a.T ? kp.kt : 0;
    kp.xo = s_tapoff[tt0];
    kp.wo = s_tapoff[64 + tt0];
  }
  const unsigned piece0 = smem_base + wave * (RPI * KB);

  // half H of the pieces of one stage into ring slot `slot`, at K position k
  auto issue_half = [&](int slot, int half, const KPos& k) {
    const unsigned sa = piece0 + slot * STAGE, sb = sa + A_BYTES;
    const bool kvalid = k.kt < a.T;
    const int tt = kvalid ? k.kt : 0;
    const int xo = k.xo + k.kc * ESZ, wo = k.wo + k.kc * ESZ;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      if ((i < A_H0) != (half == 0)) continue;
      const bool ok = kvalid && ((a_mask[i] >> tt) & 1ull);
      if (a.debug == 6) continue;                    // ablation: no pixel-row pieces
      dma16(sa + i * (RPP * KB), ok ? (unsigned)(a_off[i] + xo) : OOB, xr);
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      if ((i < B_H0) != (half == 0)) continue;
      const unsigned off = (kvalid && b_off[i] >= 0) ? (unsigned)(b_off[i] + wo) : OOB;
      if (a.debug == 7) continue;                    // ablation: no filter-row pieces
      dma16(sb + i * (RPP * KB), off, wr);
    }
  };
  auto advance = [&](KPos k) -> KPos {               // by value (igemm.hip: by reference next to the asm it lived in scratch)
    int kt = k.kt, kc = k.kc;
    if (chan_major) {
      const bool wrap = kt + 1 >= a.T;
      kt = wrap ? 0 : kt + 1;
      kc += wrap ? BK : 0;
    } else {
      kc += BK;
      if (simple_k) {
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

  // ---- math side
  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int frow = lane & 31, fh = lane >> 5;
  const int wrow0 = grp * 128 + wm2 * 64, wcol0 = wn * (BN / 2);
  // fragment byte offsets inside a stage for k-group g (16 k = two 16-byte chunks: lane half fh owns chunk 2g + fh); rows r and r + 32
  // share the swizzle, so block i / j adds a compile-time 32 * KB
  int fa_off[4], fb_off[4];
  {
    const int ra = wrow0 + frow, rb = wcol0 + frow;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      fa_off[g] = ra * KB + (((2 * g + fh) ^ ((ra >> 1) & 7)) << 4);
      fb_off[g] = A_BYTES + rb * KB + (((2 * g + fh) ^ ((rb >> 1) & 7)) << 4);
    }
  }
  const int nstage = (a.K + BK - 1) / BK;
  const bool timing = a.debug == 9 && blockIdx.x == 0 && (wave & 3) == 0;
  unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = timing ? stamp() : 0;

  // ---- prologue: stages 0 and 1 in flight, stage 0 landed and visible
  issue_half(0, 0, kp);
  issue_half(0, 1, kp);
  kp = advance(kp);
  if (nstage > 1) {
    issue_half(1, 0, kp);
    issue_half(1, 1, kp);
    kp = advance(kp);
    wait_vm<NPC>();
  } else {
    wait_vm<0>();
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();        // the stagger: group 1 runs one barrier behind group 0

  for (int s0 = 0; s0 < nstage; s0 += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int s = s0 + u;
      if (s < nstage) {
        const unsigned char* st = smem + u * STAGE;
        const bool more = s + 2 < nstage && a.debug != 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
          if (timing) c0 = stamp();
          // ---- L: this phase's fragments (k-groups 2h, 2h + 1), then the pieces of stage s + 2
          uint4 fa[2][MI], fb[2][NI];
          if (a.debug != 8)                          // ablation 8: copies and barriers only
#pragma unroll
          for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[g][i] = *(const uint4*)(st + fa_off[2 * h + g] + i * (32 * KB));
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[g][j] = *(const uint4*)(st + fb_off[2 * h + g] + j * (32 * KB));
          }
          __builtin_amdgcn_sched_barrier(0);
          if (timing) c1 = stamp();
          if (more) issue_half((u + 2) % NST, h, kp);
          if (timing) c2 = stamp();
          if (h == 1) {
            if (more) { kp = advance(kp); wait_vm<NPC>(); }      // stage s + 1 has landed (this wave's pieces); stage s + 2 stays in flight
            else wait_vm<0>();
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // fragments in registers BEFORE the barrier: the slot may be refilled after it
          __builtin_amdgcn_sched_barrier(0);
          if (timing) c3 = stamp();
          __builtin_amdgcn_s_barrier();
          if (timing) c4 = stamp();
          // ---- M
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
          if (a.debug != 2 && a.debug != 8) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
              for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[g][i]), __builtin_bit_cast(f16x8, fb[g][j]), acc[i][j], 0, 0, 0);
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          if (timing) c5 = stamp();
          __builtin_amdgcn_s_barrier();
          if (timing) {
            const unsigned long long c6 = stamp();
            tacc[0] += c1 - c0; tacc[1] += c2 - c1; tacc[2] += c3 - c2; tacc[3] += c4 - c3; tacc[4] += c5 - c4; tacc[5] += c6 - c5; tacc[6] += 1;
          }
        }
      }
    }
  }
  if (timing && lane == 0) {
#pragma unroll
    for (int q = 0; q < 7; ++q) g_i8_stamp[grp][q] = tacc[q];
    g_i8_stamp[grp][7] = stamp() - t_begin;
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();        // balance the stagger
  __syncthreads();                                   // every fragment read done: the ring becomes the output staging tile

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  constexpr int ROWB = BN * ESZ, CPR = ROWB / 16;
  float ssum[NI], ssq[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) ssum[j] = ssq[j] = 0.f;
  float bias_v[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = bn0 + wcol0 + j * 32 + frow;
    bias_v[j] = ((EPI & 2) && n < a.N) ? a.bias[n] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int rl = wrow0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int cl = wcol0 + j * 32 + frow;
        float v = acc[i][j][e];
        if (EPI & 1) { ssum[j] += v; ssq[j] += v * v; }          // rows >= M and channels >= N were zero-filled: they add nothing
        if (EPI & 2) v += bias_v[j];
        if (EPI & 4) v = silu_f(v);
        *(T*)(smem + rl * ROWB + cl * ESZ) = (T)v;
      }
    }
  __syncthreads();
  constexpr int U = 4;
  static_assert((BM * CPR) % (512 * U) == 0, "store loop geometry");
#pragma unroll 1
  for (int base = tid; base < BM * CPR; base += 512 * U) {
    unsigned char* gp[U];
    uint4 o[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * 512;
      const int rl = idx / CPR, ch = idx % CPR;
      const int m = bm0 + rl, n = bn0 + ch * EPC;
      const bool ok = m < a.M && n < a.N && a.debug != 5;        // N % 8 == 0 (host); debug 5 = the tuner's dry run of an accumulating launch
      long obase;
      if (a.dense_out) {
        obase = (long)m * a.y_ld;
      } else {
        const int b = m / ohw, r = m - b * ohw, oy = r / a.OW, ox = r - oy * a.OW;
        obase = ((long)(b * a.OHF + oy * a.oy_mul + a.oy_add) * a.OWF + ox * a.ox_mul + a.ox_add) * a.y_ld;
      }
      gp[u] = ok ? (unsigned char*)a.y + (obase + n) * ESZ : nullptr;
      if ((EPI & 8) && ok) o[u] = *(const uint4*)gp[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!gp[u]) continue;
      const int idx = base + u * 512;
      uint4 v = *(const uint4*)(smem + (idx / CPR) * ROWB + (idx % CPR) * 16);
      if (EPI & 8) {
        f16x8 x = __builtin_bit_cast(f16x8, v), y = __builtin_bit_cast(f16x8, o[u]);
#pragma unroll
        for (int q = 0; q < 8; ++q) x[q] = (T)((float)x[q] + (float)y[q]);
        v = __builtin_bit_cast(uint4, x);
      }
      *(uint4*)gp[u] = v;
    }
  }
  if (EPI & 1) {
    float* s_red = (float*)(smem + RED);
    const int rg = grp * 2 + wm2;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const float s1 = ssum[j] + __shfl_xor(ssum[j], 32);
      const float s2 = ssq[j] + __shfl_xor(ssq[j], 32);
      if (fh == 0) {
        const int col = wcol0 + j * 32 + frow;
        s_red[rg * 2 * BN + col] = s1;               // one row per wave row group: folded below in index order (bit-reproducible)
        s_red[rg * 2 * BN + BN + col] = s2;
      }
    }
    __syncthreads();
    if (tid < BN && bn0 + tid < a.N) {
      const long so = (long)(blockIdx.x % a.stat_slots) * a.stat_stride;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { t1 += s_red[r * 2 * BN + tid]; t2 += s_red[r * 2 * BN + BN + tid]; }
      atomicAdd(a.stat_sum + so + bn0 + tid, t1);
      atomicAdd(a.stat_sq + so + bn0 + tid, t2);
    }
  }
}

// f16 only, 16-byte addressable output rows, epilogues 0 / 1 / 8 / 6, no BN tail ticket; bn = 128 or 64
bool sy11_igemm8_legal(const IgemmArgs& a, int bn, int epi) {
  if (bn != 128 && bn != 64) return false;
  if (epi != 0 && epi != 1 && epi != 8 && epi != 6) return false;
  if (!a.vec_out || a.tail.ticket || a.M < 256 || a.K < 128) return false;
  if (a.C % 8) return false;                         // a 16-byte chunk never straddles two taps
  return bn == 128 ? a.N > 64 : (a.N > 32 && a.N <= 64);
}

int sy11_igemm8_launch(const IgemmArgs& a, int bn, int epi, hipStream_t st) {
  const long nwg = (long)cdiv(a.M, 256) * cdiv(a.N, bn);
  if (nwg <= 0 || nwg > 0x7fffffffL) SY11_FAIL(SY11_EINVAL, "igemm8: bad grid %ld", nwg);
  dim3 grid((unsigned)nwg), block(512);
#define SY11_I8(BNN)                                                                                   \
  do {                                                                                                 \
    if (epi == 0) hipLaunchKernelGGL((igemm8_kernel<BNN, 0>), grid, block, 0, st, a);                  \
    else if (epi == 1) hipLaunchKernelGGL((igemm8_kernel<BNN, 1>), grid, block, 0, st, a);             \
    else if (epi == 8) hipLaunchKernelGGL((igemm8_kernel<BNN, 8>), grid, block, 0, st, a);             \
    else hipLaunchKernelGGL((igemm8_kernel<BNN, 6>), grid, block, 0, st, a);                           \
  } while (0)
  if (bn == 128) SY11_I8(128);
  else SY11_I8(64);
#undef SY11_I8
  SY11_LAUNCH_CHECK("igemm8");
  return SY11_OK;
}

// diagnostic: the 16 counters of the last SY11_IGEMM_DEBUG=9 launch ([wave group][6 section sums | phases | kernel cycles])
extern "C" int sy11_debug_stamps(uint64_t* out16) {
  SY11_REQUIRE(out16 != nullptr, "debug_stamps: null pointer");
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_i8_stamp), sizeof(unsigned long long) * 16) != hipSuccess) {
    (void)hipGetLastError();
    SY11_FAIL(SY11_ELAUNCH, "debug_stamps: copy failed");
  }
  return SY11_OK;
}
