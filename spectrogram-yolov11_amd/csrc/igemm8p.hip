// igemm8p.hip — PERSISTENT form of the 8-wave convolution pipeline of igemm8.hip (256 pixels x 128 channels, 128-byte K stages).
//
// What the in-kernel stamps of igemm8.hip showed (tools/igemm8_stamps.py, r04): with one 512-thread workgroup per CU a tile's life is
//     address setup ~2 500-4 600 cycles | first stages in flight ~4 500-12 000 | main loop 1 650 per stage | epilogue ~5 800
// i.e. on a K = 384..1152 layer a third to a half of the time is NOT the main loop, and because every workgroup of a round starts at the
// same moment the "first stages" wait is a chip-wide burst (256 x 144 KB) served at the HBM rate.  The 4-wave tiles of igemm.hip hide the
// same costs behind a second workgroup on the CU.  Here a workgroup stays on its CU and walks tiles:
//     [loop of tile t] -> addresses of tile t+1 -> ISSUE stages 0 and 1 of tile t+1 -> epilogue of tile t -> issue stage 2 -> [loop of t+1]
// so the copies of the next tile's first stages land while the current tile is transposed and stored, and the CUs drift apart instead of
// bursting together.  The output tile is staged through ONE ring slot (48 KB) in two 128-row passes, the other two slots already
// belong to the next tile.  Main loop, ring discipline and epilogue arithmetic are igemm8.hip's (whole-stage register double buffering,
// one barrier per stage, skewed issue order) — see there.
//
// vmcnt across tiles: stores of the epilogue count in vmcnt like the copies.  When the next tile's loop waits for its stage 0 with
// vmcnt(2 x NPC), the 12 youngest operations are the 6 pieces of stage 2 and 6 epilogue stores: stages 0 and 1 are older than every
// store of the epilogue, so they have landed whatever the stores do.
#include "common.h"
#include "tune.h"
#include "det.h"
#include "bn_tail.h"
#include "igemm_args.h"

namespace {

typedef int rsrcp_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pdma16(unsigned lds_addr, unsigned voff, rsrcp_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ rsrcp_t pmake_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  rsrcp_t r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(p >> 32) & 0xffff);
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
template <int N> __device__ __forceinline__ void pwait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int NPC, int MAXS> __device__ __forceinline__ void pwait_stages(int stages) {
  const int n = stages < 0 ? 0 : (stages > MAXS ? MAXS : stages);
  if (n == 0) pwait_vm<0>();
  else if (n == 1) pwait_vm<NPC>();
  else pwait_vm<2 * NPC>();
}

}  // namespace

// diagnostic (SY11_IGEMM_DEBUG=9): per workgroup, summed over its tiles, the shader cycles of [wait for stages 0 / 1 + first fragment reads |
// main loop | next tile's addresses + its first two stages issued | epilogue] and the tile count; read back by sy11_debug_stamps (mode 2)
__device__ unsigned long long g_i8p_wg[256][8];

template <int EPI>
__global__ __launch_bounds__(512) void igemm8p_kernel(const IgemmArgs a, const int ntiles) {
  typedef _Float16 T;
  constexpr int BM = 256, BN = 128, KB = 128, ESZ = 2, EPC = 8, BK = 64, NST = 3, G = 4;
  constexpr int RPI = 8, RPP = 64, APASS = BM / RPP, BPASS = BN / RPP, NPC = APASS + BPASS;
  constexpr int MI = 2, NI = 2;
  constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB, STAGE = A_BYTES + B_BYTES;
  constexpr unsigned OOB = 0x80000000u;
  constexpr int TAB = NST * STAGE, RED = TAB + 768;
  __shared__ __attribute__((aligned(16))) unsigned char smem[RED + 4 * 2 * BN * 4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wm2 = (wave >> 1) & 1, wn = wave & 1;
  // tiles of this workgroup: XCD x (workgroups w with w % 8 == x share an L2) owns a contiguous run of tiles; its workgroups take them
  // round-robin, so the tiles in flight on one L2 at any moment are neighbours (shared filter rows, adjacent pixel rows)
  int tile, tile_end, tile_step;
  {
    const int W = gridDim.x, x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int lo = x * q + (x < r ? x : r);
    tile_end = lo + q + (x < r ? 1 : 0);
    tile_step = (W - x + 7) >> 3;                      // workgroups on this XCD
    tile = lo + j;
  }
  if (tile >= tile_end) return;                        // whole workgroup, before any barrier

  const int ld_row = tid >> 3;
  const int ld_chunk = (tid & 7) ^ ((ld_row >> 1) & 7);
  int* s_tapoff = (int*)(smem + TAB);
  if (tid < 64) {
    const int t = tid < a.T ? tid : 0;
    const int dy = a.tap_dy[t], dx = a.tap_dx[t];
    s_tapoff[tid] = (dy * a.IW + dx) * a.x_ld * ESZ;
    s_tapoff[64 + tid] = a.tap_w[t] * a.C * ESZ;
    s_tapoff[128 + tid] = (dy & 0xffff) | (dx << 16);
  }
  const int ohw = a.OH * a.OW;
  struct KPos { int kt, kc, xo, wo; };
  struct TileAddr { int bm0, bn0; int a_off[APASS]; unsigned long long a_mask[APASS]; int b_off[BPASS]; };
  const rsrcp_t xr = pmake_rsrc(a.x, a.x_bytes), wr = pmake_rsrc(a.w, a.w_bytes);
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const bool simple_k = a.C >= BK;
  const bool chan_major = a.chan_major && a.T > 1 && a.C % BK == 0;
  __syncthreads();                                     // tap tables visible (they are per layer: every tile of the walk uses them)
  KPos kp_init;
  kp_init.kt = (ld_chunk * EPC) / a.C;
  kp_init.kc = (ld_chunk * EPC) - kp_init.kt * a.C;
  {
    const int tt0 = kp_init.kt < a.T ? kp_init.kt : 0;
    kp_init.xo = s_tapoff[tt0];
    kp_init.wo = s_tapoff[64 + tt0];
  }
  const unsigned piece0 = smem_base + wave * (RPI * KB);

  auto setup = [&](int t) -> TileAddr {
    TileAddr ta;
    const int tile_n = t % a.tiles_n, tile_m = t / a.tiles_n;
    ta.bm0 = tile_m * BM; ta.bn0 = tile_n * BN;
    int iy0[APASS], ix0[APASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int m = ta.bm0 + ld_row + RPP * i;
      const bool ok = m < a.M;
      const int mm = ok ? m : 0;
      const int b = mm / ohw, r = mm - b * ohw, oy = r / a.OW, ox = r - oy * a.OW;
      const int y0 = oy * a.sy, x0 = ox * a.sx;
      ta.a_off[i] = ((b * a.IH + y0) * a.IW + x0) * a.x_ld * ESZ;
      iy0[i] = ok ? y0 : -0x4000;
      ix0[i] = x0;
      ta.a_mask[i] = 0;
    }
#pragma unroll 1
    for (int t2 = 0; t2 < a.T; ++t2) {
      const int v = s_tapoff[128 + t2];
      const int dy = (short)(v & 0xffff), dx = v >> 16;
#pragma unroll
      for (int i = 0; i < APASS; ++i)
        if ((unsigned)(iy0[i] + dy) < (unsigned)a.IH && (unsigned)(ix0[i] + dx) < (unsigned)a.IW) ta.a_mask[i] |= 1ull << t2;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int n = ta.bn0 + ld_row + RPP * i;
      ta.b_off[i] = n < a.N ? n * a.wK * ESZ : (int)OOB;
    }
    return ta;
  };
  auto issue_stage = [&](int slot, const KPos& k, const TileAddr& ta) {
    const unsigned sa = piece0 + slot * STAGE, sb = sa + A_BYTES;
    const bool kvalid = k.kt < a.T;
    const int tt = kvalid ? k.kt : 0;
    const int xo = k.xo + k.kc * ESZ, wo = k.wo + k.kc * ESZ;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const bool ok = kvalid && ((ta.a_mask[i] >> tt) & 1ull);
      pdma16(sa + i * (RPP * KB), ok ? (unsigned)(ta.a_off[i] + xo) : OOB, xr);
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const unsigned off = (kvalid && ta.b_off[i] >= 0) ? (unsigned)(ta.b_off[i] + wo) : OOB;
      pdma16(sb + i * (RPP * KB), off, wr);
    }
  };
  auto advance = [&](KPos k) -> KPos {
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

  const int frow = lane & 31, fh = lane >> 5;
  const int wrow0 = grp * 128 + wm2 * 64, wcol0 = wn * (BN / 2);
  int fa_off[G], fb_off[G];
  {
    const int ra = wrow0 + frow, rb = wcol0 + frow;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      fa_off[g] = ra * KB + (((2 * g + fh) ^ ((ra >> 1) & 7)) << 4);
      fb_off[g] = A_BYTES + rb * KB + (((2 * g + fh) ^ ((rb >> 1) & 7)) << 4);
    }
  }
  const int nstage = (a.K + BK - 1) / BK;
  const int npro = nstage < NST ? nstage : NST;

  TileAddr cur = setup(tile);
  KPos kp = kp_init;
#pragma unroll
  for (int q = 0; q < NST; ++q)
    if (q < nstage) { issue_stage(q, kp, cur); kp = advance(kp); }

  const bool timing = a.debug == 9 && tid == 0 && blockIdx.x < 256;
  unsigned long long tsum[5] = {0, 0, 0, 0, 0};
  const unsigned long long r0 = timing ? __builtin_amdgcn_s_memrealtime() : 0;
  for (;;) {
    const unsigned long long c0 = timing ? __builtin_amdgcn_s_memtime() : 0;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    pwait_stages<NPC, NST - 1>(npro - 1);              // stage 0 has landed (see the header for the stores in between)
    __builtin_amdgcn_s_barrier();
    uint4 fA[2][G][MI], fB[2][G][NI];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int i = 0; i < MI; ++i) fA[0][g][i] = *(const uint4*)(smem + fa_off[g] + i * (32 * KB));
#pragma unroll
      for (int j = 0; j < NI; ++j) fB[0][g][j] = *(const uint4*)(smem + fb_off[g] + j * (32 * KB));
    }
    pwait_stages<NPC, NST - 1>(npro - 2);              // stage 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned long long c1 = timing ? __builtin_amdgcn_s_memtime() : 0;

    for (int s0 = 0; s0 < nstage; s0 += 2 * NST) {
#pragma unroll
      for (int u = 0; u < 2 * NST; ++u) {
        const int s = s0 + u;
        if (s < nstage) {
          const int cb = u & 1, nb = cb ^ 1;
          const unsigned char* nx = smem + ((u + 1) % NST) * STAGE;
          const bool more = s + NST < nstage;
          if (grp == 1 && more) issue_stage(u % NST, kp, cur);
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int g = 0; g < G; ++g) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fA[nb][g][i] = *(const uint4*)(nx + fa_off[g] + i * (32 * KB));
#pragma unroll
            for (int j = 0; j < NI; ++j) fB[nb][g][j] = *(const uint4*)(nx + fb_off[g] + j * (32 * KB));
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
              for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fA[cb][g][i]), __builtin_bit_cast(f16x8, fB[cb][g][j]), acc[i][j], 0, 0, 0);
          }
#pragma unroll
          for (int q = 0; q < G * MI * NI; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (q < G * (MI + NI)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          if (grp == 0 && more) issue_stage(u % NST, kp, cur);
          if (more) kp = advance(kp);
          pwait_stages<NPC, NST - 2>(nstage - 3 - s);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
        }
      }
    }
    // every slot has been read for the last time (the loop's last barrier): the next tile may take slots 0 and 1, the epilogue slot 2

    const unsigned long long c2 = timing ? __builtin_amdgcn_s_memtime() : 0;
    const int next_tile = tile + tile_step;
    const bool has_next = next_tile < tile_end;
    TileAddr nxt = cur;
    KPos kpn = kp_init;
    if (has_next) {
      nxt = setup(next_tile);
      issue_stage(0, kpn, nxt); kpn = advance(kpn);
      if (nstage > 1) { issue_stage(1, kpn, nxt); kpn = advance(kpn); }
    }

    const unsigned long long c3 = timing ? __builtin_amdgcn_s_memtime() : 0;
    // ---- epilogue of the current tile through slot 2, 128 rows at a time
    const int bm0 = cur.bm0, bn0 = cur.bn0;
    unsigned char* stg = smem + 2 * STAGE;
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
    for (int half = 0; half < 2; ++half) {
      if (half == 1) __syncthreads();                  // the first half has been read out of the staging slot
      if (grp == half) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int rl = wm2 * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;      // row inside this 128-row half
#pragma unroll
            for (int j = 0; j < NI; ++j) {
              const int cl = wcol0 + j * 32 + frow;
              float v = acc[i][j][e];
              if (EPI & 1) { ssum[j] += v; ssq[j] += v * v; }
              if (EPI & 2) v += bias_v[j];
              if (EPI & 4) v = silu_f(v);
              *(T*)(stg + rl * ROWB + cl * ESZ) = (T)v;
            }
          }
      }
      __syncthreads();
      constexpr int U = 4;
      static_assert(128 * CPR == 512 * U, "store geometry: one trip of four chunks per thread and half");
      unsigned char* gp[U];
      uint4 o[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = tid + u * 512;
        const int rl = idx / CPR, ch = idx % CPR;
        const int m = bm0 + half * 128 + rl, n = bn0 + ch * EPC;
        const bool ok = m < a.M && n < a.N && a.debug != 5;
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
        const int idx = tid + u * 512;
        uint4 v = *(const uint4*)(stg + (idx / CPR) * ROWB + (idx % CPR) * 16);
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
          s_red[rg * 2 * BN + col] = s1;
          s_red[rg * 2 * BN + BN + col] = s2;
        }
      }
      __syncthreads();
      if (tid < BN && bn0 + tid < a.N) {
        const long so = (long)(tile % a.stat_slots) * a.stat_stride;      // a row per TILE (ordered mode: tiles = rows, each written once)
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { t1 += s_red[r * 2 * BN + tid]; t2 += s_red[r * 2 * BN + BN + tid]; }
        atomicAdd(a.stat_sum + so + bn0 + tid, t1);
        atomicAdd(a.stat_sq + so + bn0 + tid, t2);
      }
    }
    if (timing) {
      const unsigned long long c4 = __builtin_amdgcn_s_memtime();
      tsum[0] += c1 - c0; tsum[1] += c2 - c1; tsum[2] += c3 - c2; tsum[3] += c4 - c3; tsum[4] += 1;
      g_i8p_wg[blockIdx.x][0] = tsum[0]; g_i8p_wg[blockIdx.x][1] = tsum[1]; g_i8p_wg[blockIdx.x][2] = tsum[2]; g_i8p_wg[blockIdx.x][3] = tsum[3];
      g_i8p_wg[blockIdx.x][4] = tsum[4]; g_i8p_wg[blockIdx.x][5] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    if (!has_next) break;
    __syncthreads();                                   // staging slot (and the statistics fold) read out: stage 2 of the next tile may land in it
    if (nstage > 2) { issue_stage(2, kpn, nxt); kpn = advance(kpn); }
    cur = nxt; kp = kpn; tile = next_tile;
  }
}

bool sy11_igemm8p_legal(const IgemmArgs& a, int epi) {
  if (epi != 0 && epi != 1 && epi != 8 && epi != 6) return false;
  if (!a.vec_out || a.tail.ticket || a.M < 256 || a.K < 128 || a.N <= 64) return false;
  return a.C % 8 == 0;
}

int sy11_igemm8p_launch(const IgemmArgs& a, int epi, hipStream_t st) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); cus = 256; }
    else cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const long ntiles = (long)cdiv(a.M, 256) * cdiv(a.N, 128);
  if (ntiles <= 0 || ntiles > 0x7fffffffL) SY11_FAIL(SY11_EINVAL, "igemm8p: bad tile count %ld", ntiles);
  const int wgs = ntiles < cus ? (int)ntiles : cus;    // one workgroup per CU (152 KB of LDS each)
  dim3 grid((unsigned)wgs), block(512);
  if (epi == 0) hipLaunchKernelGGL((igemm8p_kernel<0>), grid, block, 0, st, a, (int)ntiles);
  else if (epi == 1) hipLaunchKernelGGL((igemm8p_kernel<1>), grid, block, 0, st, a, (int)ntiles);
  else if (epi == 8) hipLaunchKernelGGL((igemm8p_kernel<8>), grid, block, 0, st, a, (int)ntiles);
  else hipLaunchKernelGGL((igemm8p_kernel<6>), grid, block, 0, st, a, (int)ntiles);
  SY11_LAUNCH_CHECK("igemm8p");
  return SY11_OK;
}

// diagnostic readback (see g_i8p_wg): 256 x 8 counters to the host array
extern "C" int sy11_debug_stamps_persistent(uint64_t* out2048) {
  SY11_REQUIRE(out2048 != nullptr, "debug_stamps_persistent: null pointer");
  if (hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_i8p_wg), sizeof(unsigned long long) * 2048) != hipSuccess) {
    (void)hipGetLastError();
    SY11_FAIL(SY11_ELAUNCH, "debug_stamps_persistent: copy failed");
  }
  return SY11_OK;
}
