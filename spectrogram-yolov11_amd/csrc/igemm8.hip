// igemm8.hip — the implicit-GEMM convolution of igemm.hip as an 8-wave (two waves per SIMD) software pipeline (r04).
//
// Why a second main loop.  The 4-wave tiles of igemm.hip run every wave through  barrier -> issue LDS-DMA -> read fragments -> MFMAs:
// a wave that is issuing its copies (~65 cycles per 1 KiB piece) is not issuing MFMAs, the workgroups of a CU run in phase, and the
// matrix pipe sat at 50-60 % busy (r03 counters).  Here a workgroup is 512 threads = two wave GROUPS (waves 0-3 / 4-7: wave w and
// w + 4 share a SIMD), one workgroup per CU, and a STEP (one 64-deep K stage) is, per wave,
//        group 1:  [6 pieces of stage s+3]  [16 MFMAs of stage s from registers, interleaved with the 16 fragment reads of stage s+1]
//        group 0:  [16 MFMAs ... reads ...]  [6 pieces of stage s+3]
//        both:     counted vmcnt (stage s+2 landed), lgkmcnt(0), ONE s_barrier
// i.e. the two waves of a SIMD run the same work in opposite order: while one feeds the matrix pipe the other feeds the address path.
// Fragments are double-buffered in REGISTERS (a whole stage per set), so a stage's LDS slot is free one step early and three slots
// carry a prefetch distance of three stages.
//
// r04 history: the first form followed the guide's 8-phase template literally (two barriers per 8-MFMA phase, group 1 one barrier
// behind): parity-green and exactly as fast as the 4-wave 256x128 tile (187 vs 186 us on 80x80 256->256 s2).  In-kernel stamps
// (SY11_IGEMM_DEBUG=9) showed why: per phase the load section (8 reads + 3 pieces + waits) took ~490 cycles against 256 of MFMA, so
// the computing group waited ~450 cycles at its second barrier — the barriers paced the pipeline at the slower section twice per
// phase.  This form has one barrier per 512 MFMA cycles and hides the reads under the wave's own MFMAs.
//
// Geometry: 256 pixels x BN channels per workgroup (BN = 128: wave tile 64 x 64, or 64: wave tile 64 x 32), stages of 128 BYTES of K
// per row (64 f16: whole 128-byte lines from HBM / L2), a ring of three LDS slots.
//
// Ring discipline (one barrier B(s) ends step s):
//   RAW  stage s+2 is retired by EVERY wave's own counted s_waitcnt vmcnt before B(s) (only stage s+3's pieces stay in flight) and is
//        first read during step s+1.
//   WAR  stage s+1's fragment reads (issued during step s) are drained by lgkmcnt(0) before B(s); its slot is refilled by the pieces
//        of stage s+4, issued during step s+1.
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
// wait until at most min(max(stages, 0), MAXS) x NPC of this wave's pieces are outstanding (wave-uniform `stages`; immediates need a switch)
template <int NPC, int MAXS> __device__ __forceinline__ void wait_stages(int stages) {
  static_assert(MAXS <= 5, "wait_stages: switch depth");
  const int n = stages < 0 ? 0 : (stages > MAXS ? MAXS : stages);
  if (n == 0) wait_vm<0>();
  else if (n == 1) wait_vm<NPC>();
  else if (n == 2) wait_vm<2 * NPC>();
  else if (n == 3) wait_vm<3 * NPC>();
  else if (n == 4) wait_vm<4 * NPC>();
  else wait_vm<5 * NPC>();
}
// LDS bank swizzle of a tile row (applied on the SOURCE side of the copies and on the fragment reads): 128-byte rows XOR the 16-byte chunk
// index with (row >> 1) & 7, 64-byte rows with (row >> 2) & 3 — conflict-free ds_read_b128 fragments (igemm.hip)
template <int KB> __device__ __forceinline__ int swz8(int r) { return KB == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3); }

// diagnostic build only (SY11_IGEMM_DEBUG=9): where a phase spends its cycles.  Wave 0 and wave 4 of workgroup 0 add up, over all
// steps, the s_memtime deltas of [group 1's piece issue | MFMAs + fragment reads | group 0's piece issue + counted waits | barrier] + the step
// count + the whole kernel; read back with sy11_debug_stamps().  Nothing else in the kernel reads this memory.
__device__ unsigned long long g_i8_stamp[2][8];
__device__ unsigned long long g_i8_wg[2048][8];     // per workgroup (first 2048): entry / exit in s_memrealtime ticks (100 MHz), XCC id, then shader cycles entry -> addresses set up -> first step -> last step done -> epilogue done
__device__ __forceinline__ unsigned long long stamp() { return __builtin_amdgcn_s_memtime(); }

}  // namespace

// EPI bits as in igemm.hip: 1 statistics, 2 bias, 4 SiLU, 8 accumulate.  Instantiated: 0, 1, 8, 6.  DBG = 1: the diagnostic build (honours
// SY11_IGEMM_DEBUG 2 = no MFMAs and 9 = cycle stamps; instantiated for <128, 1> only) — the product build carries none of those branches.
template <int BN, int KB, int EPI, int DBG = 0>
__global__ __launch_bounds__(512) void igemm8_kernel(const IgemmArgs a) {
  typedef _Float16 T;
  constexpr int BM = 256, ESZ = 2, EPC = 8, BK = KB / ESZ;
  constexpr bool ROLL = BN == 256;                   // 256 x 256 tile: k-group-granular register double buffering (see the main loop)
  constexpr int NST = ROLL ? 4 : (KB == 128 ? 3 : 6);   // LDS slots: 3 x 48 KB / 6 x 24 KB (BN = 128), 4 x 32 KB (BN = 256)
  static_assert(!ROLL || KB == 64, "the 256-channel tile takes 64-byte stages");
  constexpr int CPRW = KB / 16, G = KB / 32;         // 16-byte chunks per row; k-groups (16 k = one MFMA) per stage
  constexpr int RPI = 64 / CPRW, RPP = 8 * RPI;      // rows per wave-instruction, rows per pass of the 8 waves
  constexpr int APASS = BM / RPP, BPASS = BN / RPP;
  static_assert(BPASS >= 1, "a pass of the 8 waves must not cover more filter rows than the tile has");
  constexpr int MI = ROLL ? 4 : 2, NI = ROLL ? 2 : BN / 64;   // 32 x 32 blocks of a wave tile: 64 rows x BN / 2 columns, or 128 x 64 (BN = 256)
  constexpr int RG = ROLL ? 2 : 4;                   // wave rows of the tile (statistics fold)
  constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB, STAGE = A_BYTES + B_BYTES;
  constexpr int NPC = APASS + BPASS;                 // LDS-DMA pieces per wave and stage
  constexpr unsigned OOB = 0x80000000u;
  constexpr int TAB = NST * STAGE;                   // tap tables: [64] x delta, [64] w delta, [64] packed (dy, dx)
  constexpr int RED = TAB + 768;                     // statistics fold: [RG wave rows][sum | sumsq][BN]
  __shared__ __attribute__((aligned(16))) unsigned char smem[RED + RG * 2 * BN * 4];

  const unsigned long long t_entry = (DBG && a.debug == 9) ? stamp() : 0;
  const unsigned long long r_begin = (DBG && a.debug == 9) ? __builtin_amdgcn_s_memrealtime() : 0;
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
  const int ld_row = tid / CPRW;
  const int ld_chunk = (tid % CPRW) ^ swz8<KB>(ld_row);
  int* s_tapoff = (int*)(smem + TAB);
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
    a_off[i] = ((b * a.IH + iy0) * a.IW + ix0) * a.x_ld * ESZ;
    a_iy[i] = ok ? iy0 : -0x4000;                    // a row past M: no tap is inside the image
    a_ix[i] = ix0;
    a_mask[i] = 0;
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
  // "tap t of row i lies inside the image", one bit per tap.  The taps come from the LDS table (one broadcast read per tap for all of
  // a lane's rows): read from the kernel arguments inside this loop (r03 form, still in igemm.hip's 4-wave tiles until r04) every tap of
  // every row was a dependent scalar-memory round trip — 16 000 cycles per workgroup on a 3x3 layer, a quarter of a K = 1152 tile's life
  // (in-kernel stamps, tools/igemm8_stamps.py).
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
  const unsigned piece0 = smem_base + wave * (RPI * KB);

  // the NPC pieces of one stage into ring slot `slot`, at K position k
  auto issue_stage = [&](int slot, const KPos& k) {
    const unsigned sa = piece0 + slot * STAGE, sb = sa + A_BYTES;
    const bool kvalid = k.kt < a.T;
    const int tt = kvalid ? k.kt : 0;
    const int xo = k.xo + k.kc * ESZ, wo = k.wo + k.kc * ESZ;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const bool ok = kvalid && ((a_mask[i] >> tt) & 1ull);
      dma16(sa + i * (RPP * KB), ok ? (unsigned)(a_off[i] + xo) : OOB, xr);
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const unsigned off = (kvalid && b_off[i] >= 0) ? (unsigned)(b_off[i] + wo) : OOB;
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
  const int wrow0 = ROLL ? grp * 128 : grp * 128 + wm2 * 64, wcol0 = ROLL ? (wave & 3) * 64 : wn * (BN / 2);
  // fragment byte offsets inside a stage for k-group g (16 k = two 16-byte chunks: lane half fh owns chunk 2g + fh); rows r and r + 32
  // share the swizzle, so block i / j adds a compile-time 32 * KB
  int fa_off[G], fb_off[G];
  {
    const int ra = wrow0 + frow, rb = wcol0 + frow;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      fa_off[g] = ra * KB + (((2 * g + fh) ^ swz8<KB>(ra)) << 4);
      fb_off[g] = A_BYTES + rb * KB + (((2 * g + fh) ^ swz8<KB>(rb)) << 4);
    }
  }
  const int nstage = (a.K + BK - 1) / BK;
  const bool timing = DBG && a.debug == 9 && blockIdx.x == 0 && (wave & 3) == 0;
  unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = (DBG && a.debug == 9) ? stamp() : 0;
  unsigned long long t_loop = 0;

  if constexpr (ROLL) {
    // ---- 256 x 256 tile: 128 accumulator registers leave room for TWO k-groups of fragments (2 x 24 registers), not two stages: the
    // reads of k-group g + 1 (the next stage's k-group 0 behind the last one) run under the 8 MFMAs of k-group g.  A slot is then read
    // during its own step (and its k-group 0 during the step before), so it is free one step LATER than in the whole-stage scheme:
    // four slots = stage s (being read), s + 1 (k-group 0 being read), s + 2 (landed by the end of the step), s + 3 (in flight, in the slot
    // stage s - 1 left).   RAW: stage s + 2 retired before B(s), first read (k-group 0) during step s + 1.   WAR: slot of stage s - 1:
    // last read during step s - 1, drained before B(s - 1); refilled by pieces issued during step s.
    static_assert(G == 2, "rolling scheme: two k-groups per stage");
    const int npro = nstage < NST - 1 ? nstage : NST - 1;
#pragma unroll
    for (int q = 0; q < NST - 1; ++q)
      if (q < nstage) { issue_stage(q, kp); kp = advance(kp); }
    wait_stages<NPC, NST - 2>(npro - 1);               // stage 0 has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    uint4 rA[2][MI], rB[2][NI];                        // [register set = k-group parity][32-row block]
#pragma unroll
    for (int i = 0; i < MI; ++i) rA[0][i] = *(const uint4*)(smem + fa_off[0] + i * (32 * KB));
#pragma unroll
    for (int j = 0; j < NI; ++j) rB[0][j] = *(const uint4*)(smem + fb_off[0] + j * (32 * KB));
    wait_stages<NPC, NST - 2>(npro - 2);               // stage 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (DBG) t_loop = stamp();
    for (int s0 = 0; s0 < nstage; s0 += NST) {
#pragma unroll
      for (int u = 0; u < NST; ++u) {
        const int s = s0 + u;
        if (s < nstage) {
          const unsigned char* cu = smem + u * STAGE;                 // stage s (its k-group 1)
          const unsigned char* nx = smem + ((u + 1) % NST) * STAGE;   // stage s + 1 (its k-group 0)
          const bool more = s + NST - 1 < nstage && a.debug != 1;     // stage s + 3 goes into the slot stage s - 1 has left
          if (grp == 1 && more) issue_stage((u + NST - 1) % NST, kp);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const unsigned char* src = g + 1 < G ? cu : nx;
            const int gn = g + 1 < G ? g + 1 : 0;
#pragma unroll
            for (int i = 0; i < MI; ++i) rA[gn & 1][i] = *(const uint4*)(src + fa_off[gn] + i * (32 * KB));
#pragma unroll
            for (int j = 0; j < NI; ++j) rB[gn & 1][j] = *(const uint4*)(src + fb_off[gn] + j * (32 * KB));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, rA[g & 1][i]), __builtin_bit_cast(f16x8, rB[g & 1][j]), acc[i][j], 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < G * MI * NI; ++q) {      // one fragment read per MFMA gap; k-group g's MFMAs wait only for k-group g's reads
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q % (MI * NI) < MI + NI) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          if (grp == 0 && more) issue_stage((u + NST - 1) % NST, kp);
          if (more) kp = advance(kp);
          wait_stages<NPC, 1>(a.debug == 1 ? 0 : nstage - 3 - s);     // stage s + 2 has landed; stage s + 3 stays in flight
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
        }
      }
    }
  } else {
    // ---- prologue: stages 0 .. NST-1 in flight; stage 0's fragments in register set 0; stage 1 landed and visible
    const int npro = nstage < NST ? nstage : NST;
  #pragma unroll
    for (int q = 0; q < NST; ++q)
      if (q < nstage) { issue_stage(q, kp); kp = advance(kp); }
    wait_stages<NPC, NST - 1>(npro - 1);               // stage 0 has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                      // ... and is visible to every wave
    uint4 fA[2][G][MI], fB[2][G][NI];                  // [register set][k-group][32-row block]
  #pragma unroll
    for (int g = 0; g < G; ++g) {
  #pragma unroll
      for (int i = 0; i < MI; ++i) fA[0][g][i] = *(const uint4*)(smem + fa_off[g] + i * (32 * KB));
  #pragma unroll
      for (int j = 0; j < NI; ++j) fB[0][g][j] = *(const uint4*)(smem + fb_off[g] + j * (32 * KB));
    }
    wait_stages<NPC, NST - 1>(npro - 2);               // stage 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // stage 1 visible; slot 0 free (its fragments are in registers)

    if (DBG) t_loop = stamp();
    for (int s0 = 0; s0 < nstage; s0 += 2 * NST) {
  #pragma unroll
      for (int u = 0; u < 2 * NST; ++u) {
        const int s = s0 + u;
        if (s < nstage) {
          const int cur = u & 1, nxt = cur ^ 1;                      // compile-time after unrolling (2 * NST is even and a multiple of NST)
          const unsigned char* nx = smem + ((u + 1) % NST) * STAGE;  // stage s + 1: read into the other register set during this step's MFMAs
          const bool more = s + NST < nstage && a.debug != 1;        // stage s + NST goes into the slot stage s has just left
          unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
          if (timing) c0 = stamp();
          if (grp == 1 && more) issue_stage(u % NST, kp);
          if (timing) c1 = stamp();
          __builtin_amdgcn_sched_barrier(0);
#ifndef I8_NO_PRIO
          __builtin_amdgcn_s_setprio(1);
#endif
  #pragma unroll
          for (int g = 0; g < G; ++g) {
  #pragma unroll
            for (int i = 0; i < MI; ++i) fA[nxt][g][i] = *(const uint4*)(nx + fa_off[g] + i * (32 * KB));
  #pragma unroll
            for (int j = 0; j < NI; ++j) fB[nxt][g][j] = *(const uint4*)(nx + fb_off[g] + j * (32 * KB));
            if (!DBG || a.debug != 2) {
  #pragma unroll
              for (int i = 0; i < MI; ++i)
  #pragma unroll
                for (int j = 0; j < NI; ++j)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fA[cur][g][i]), __builtin_bit_cast(f16x8, fB[cur][g][j]), acc[i][j], 0, 0, 0);
            }
          }
          // one fragment read per MFMA gap (left alone the compiler issues all the reads in a row behind the first MFMA: ~100 idle pipe cycles)
#ifndef I8_NO_INTERLEAVE
  #pragma unroll
          for (int q = 0; q < G * MI * NI; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q < G * (MI + NI)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
#endif
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          if (timing) c2 = stamp();
          if (grp == 0 && more) issue_stage(u % NST, kp);
          if (more) kp = advance(kp);
          // stage s + 2 has landed (this wave's pieces); the stages issued after it — up to NST - 2 of them — stay in flight
          wait_stages<NPC, NST - 2>(a.debug == 1 ? 0 : nstage - 3 - s);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // stage s + 1 is in registers: its slot may be refilled after the barrier
          __builtin_amdgcn_sched_barrier(0);
          if (timing) c3 = stamp();
          __builtin_amdgcn_s_barrier();
          if (timing) {
            const unsigned long long c4 = stamp();
            tacc[0] += c1 - c0; tacc[1] += c2 - c1; tacc[2] += c3 - c2; tacc[3] += c4 - c3; tacc[6] += 1;
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
  if (DBG && a.debug == 9 && tid == 0 && blockIdx.x < 2048) {
    g_i8_wg[blockIdx.x][0] = r_begin;
    g_i8_wg[blockIdx.x][2] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);       // HW_REG_XCC_ID (id 20), bits 3:0
    g_i8_wg[blockIdx.x][3] = t_begin - t_entry;
    g_i8_wg[blockIdx.x][4] = t_loop - t_begin;
    g_i8_wg[blockIdx.x][5] = stamp() - t_loop;
  }
  const unsigned long long t_epi = (DBG && a.debug == 9) ? stamp() : 0;
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
    const int rg = ROLL ? grp : grp * 2 + wm2;
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
      for (int r = 0; r < RG; ++r) { t1 += s_red[r * 2 * BN + tid]; t2 += s_red[r * 2 * BN + BN + tid]; }
      atomicAdd(a.stat_sum + so + bn0 + tid, t1);
      atomicAdd(a.stat_sq + so + bn0 + tid, t2);
    }
  }
  if (DBG && a.debug == 9 && tid == 0 && blockIdx.x < 2048) {
    g_i8_wg[blockIdx.x][6] = stamp() - t_epi;
    g_i8_wg[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
  }
}

// f16 only, 16-byte addressable output rows, epilogues 0 / 1 / 8 / 6, no BN tail ticket; bn = 256, 128 or 64; kb = 128 or 64 bytes of K per stage
bool sy11_igemm8_legal(const IgemmArgs& a, int bn, int kb, int epi) {
  if (bn != 256 && bn != 128 && bn != 64) return false;
  if (kb != 128 && kb != 64) return false;
  if (bn == 64 && kb == 64) return false;           // a pass of the 8 waves would cover 128 filter rows: not instantiated
  if (bn == 256 && kb != 64) return false;          // 256 x 256: 64-byte stages in four slots
  if (epi != 0 && epi != 1 && epi != 8 && epi != 6) return false;
  if (!a.vec_out || a.tail.ticket || a.M < 256 || a.K < 128) return false;
  if (a.C % 8) return false;                         // a 16-byte chunk never straddles two taps
  return bn == 256 ? a.N > 128 : (bn == 128 ? a.N > 64 : (a.N > 32 && a.N <= 64));
}

int sy11_igemm8_launch(const IgemmArgs& a, int bn, int kb, int epi, hipStream_t st) {
  const long nwg = (long)cdiv(a.M, 256) * cdiv(a.N, bn);
  if (nwg <= 0 || nwg > 0x7fffffffL) SY11_FAIL(SY11_EINVAL, "igemm8: bad grid %ld", nwg);
  dim3 grid((unsigned)nwg), block(512);
#define SY11_I8(BNN, KBB)                                                                                  \
  do {                                                                                                     \
    if (epi == 0) hipLaunchKernelGGL((igemm8_kernel<BNN, KBB, 0>), grid, block, 0, st, a);                 \
    else if (epi == 1) hipLaunchKernelGGL((igemm8_kernel<BNN, KBB, 1>), grid, block, 0, st, a);            \
    else if (epi == 8) hipLaunchKernelGGL((igemm8_kernel<BNN, KBB, 8>), grid, block, 0, st, a);            \
    else hipLaunchKernelGGL((igemm8_kernel<BNN, KBB, 6>), grid, block, 0, st, a);                          \
  } while (0)
  if (bn == 128 && epi == 1 && (a.debug == 2 || a.debug == 9)) {
    if (kb == 128) hipLaunchKernelGGL((igemm8_kernel<128, 128, 1, 1>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((igemm8_kernel<128, 64, 1, 1>), grid, block, 0, st, a);
  } else if (bn == 256) SY11_I8(256, 64);
  else if (bn == 128 && kb == 128) SY11_I8(128, 128);
  else if (bn == 128) SY11_I8(128, 64);
  else if (kb == 128) SY11_I8(64, 128);
  else SY11_FAIL(SY11_EINVAL, "igemm8: 64-channel tiles take 128-byte stages only");
#undef SY11_I8
  SY11_LAUNCH_CHECK("igemm8");
  return SY11_OK;
}

// diagnostic: the 16 counters of the last SY11_IGEMM_DEBUG=9 launch ([wave group][6 section sums | phases | kernel cycles])
extern "C" int sy11_debug_stamps(uint64_t* out16) {
  SY11_REQUIRE(out16 != nullptr, "debug_stamps: null pointer");
  // out16[16 ...]: callers that pass a (16 + 4 * 2048)-element array and set out16[0] = 1 also get the per-workgroup records
  const bool want_wg = out16[0] == 1;
  if (want_wg && hipMemcpyFromSymbol(out16 + 16, HIP_SYMBOL(g_i8_wg), sizeof(unsigned long long) * 8 * 2048) != hipSuccess) {
    (void)hipGetLastError();
    SY11_FAIL(SY11_ELAUNCH, "debug_stamps: copy failed");
  }
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_i8_stamp), sizeof(unsigned long long) * 16) != hipSuccess) {
    (void)hipGetLastError();
    SY11_FAIL(SY11_ELAUNCH, "debug_stamps: copy failed");
  }
  return SY11_OK;
}
