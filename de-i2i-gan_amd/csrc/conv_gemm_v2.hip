// gather GEMM v2 (bf16 hot path): 256 x {128,64} output tiles, 8 wavefronts (512 threads), both operands streamed
// straight into LDS with LDS-DMA (global_load_lds, 16 B per lane, no VGPR staging, no ds_write), a 3-stage LDS ring
// with ONE raw s_barrier per k-step and counted vmcnt so two stages of loads stay in flight under the MFMAs.
//
// Why this shape on MI355X: a 128x128 tile needs (128+128)*128 B of operands per 2*128*128*64 FLOP = 64 FLOP/B, i.e.
// ~150 GB/s per CU at the MFMA peak -- more than a CU can pull from L2, and the VGPR->LDS store path (ds_write_b128,
// ~79 B/clk/CU) alone costs 80% of the MFMA time.  256x128 tiles raise the intensity to 85 FLOP/B and LDS-DMA takes
// the store path out of the picture.
//
// Requirements (checked by the launcher; everything else runs on the v1 kernel): bf16, source channel stride a
// multiple of 64 (every 64-element k-step then lies inside one tap: tap / channel offset are wave-uniform scalars).
//
// LDS image: [row][128 B] with the 16-byte chunk c of row r stored at slot c ^ ((r>>1)&7) (conflict-free
// ds_read_b128 fragments).  LDS-DMA writes lane l of a wave-instruction at base + 16*l, i.e. row (l>>3), slot (l&7) of
// an 8-row group, so each lane FETCHES chunk (l&7) ^ ((r>>1)&7) of its row: the swizzle lives on the source address.
// Rows that contribute zeros (M / N edge, zero padding) fetch from a zero page.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "common.h"
#include "geom.h"
#include "launch.h"

namespace dei2i {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page[256];   // zero-initialised device memory

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

DEI2I_D void glds16(const void* gptr, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_t*)gptr, (lds_void_t*)lds_wave_base, 16, 0, 0);
}

DEI2I_D int xcd_remap2(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

template <int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(512) void gather_gemm_v2_kernel(const DescPack pack, const bf16_t* __restrict__ src,
                                                             const bf16_t* __restrict__ wgt_base, const int wrows,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                             float* __restrict__ ws, const int ldc, const int act,
                                                             const int tiles_n, const int ksteps_per_split, const int ablate,
                                                             unsigned long long* __restrict__ dbg) {
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  static_assert(BM % 64 == 0 && (BM / WM) % 32 == 0, "tile rows");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LA = BM / 64, LB = BN / 64;         // LDS-DMA instructions per wave per stage
  static_assert(WM * WN == 8, "8 waves per workgroup");
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 block");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned long long t0 = ablate == 5 ? __builtin_amdgcn_s_memtime() : 0ull;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const GatherDesc& g = pack.d[blockIdx.y];
  const bf16_t* __restrict__ wgt = wgt_base + pack.woff[blockIdx.y];
  const int bid = xcd_remap2(blockIdx.x, gridDim.x);
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (m0 >= g.M) return;

  const int nk_total = g.K >> 6;
  const int kbeg = blockIdx.z * ksteps_per_split;
  const int kend = min(nk_total, kbeg + ksteps_per_split);
  if (kbeg >= kend) return;
  const int nk = kend - kbeg;

  // ---- per-lane load roles ----
  const int lrow = lane >> 3, lslot = lane & 7;
  int a_chunk[LA];
#pragma unroll
  for (int j = 0; j < LA; ++j) {
    const int r = j * 64 + wave * 8 + lrow;
    a_chunk[j] = (lslot ^ ((r >> 1) & 7)) * 8;      // element offset of the 16-byte chunk this lane fetches
  }
  const bf16_t* b_ptr[LB];
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int r = j * 64 + wave * 8 + lrow;
    const int n = n0 + r;
    const int chunk = (lslot ^ ((r >> 1) & 7)) * 8;
    b_ptr[j] = n < wrows ? wgt + (size_t)n * g.K + chunk : nullptr;
  }
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);

  // Source-pixel offset table in LDS: tab[tap][row] = element offset of the row's source pixel for that tap (or -1
  // for a zero contribution), built once per workgroup.  The steady-state k-step then costs a handful of VALU
  // instructions per LDS-DMA: address generation (reflect / bounds / upsample math) is off the critical path.
  int* tab = reinterpret_cast<int*>(smem + STAGES * STAGE_BYTES);
  const int ntaps = g.th * g.tw;
  for (int idx = tid; idx < ntaps * BM; idx += 512) {
    const int t = idx / BM, r = idx - t * BM;
    const int m = m0 + r;
    int off = -1;
    if (m < g.M) {
      int n, oy, ox;
      decode_m(g, m, n, oy, ox);
      const int ty = (int)fd_div((uint32_t)t, g.fd_tw);
      const int tx = t - ty * g.tw;
      const int y = bound_coord(oy * g.sh + g.by0 + ty * g.ys, g.Hl, g.pad_mode);
      const int x = bound_coord(ox * g.sw + g.bx0 + tx * g.xs, g.Wl, g.pad_mode);
      if ((y | x) >= 0) off = ((n * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up)) * g.Cs;
    }
    tab[idx] = off;
  }
  __syncthreads();

  const int steps_per_tap = __builtin_amdgcn_readfirstlane(g.Cs >> 6);
  int cur_tap = -1;
  int a_off[LA];
  // k-step order: tap outer, 64-channel chunk inner (matches the packed weight layout [Cout][tap][Cs])
  auto issue = [&](int stage, int it) {
    const int kstep = kbeg + it;
    const int tap = kstep / steps_per_tap;                   // wave-uniform (scalar unit)
    const int ci0 = (kstep - tap * steps_per_tap) << 6;
    if (tap != cur_tap) {
      cur_tap = tap;
#pragma unroll
      for (int j = 0; j < LA; ++j) a_off[j] = tab[tap * BM + j * 64 + wave * 8 + lrow];
    }
    unsigned char* sbase = smem + stage * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < LA; ++j) {
      const bf16_t* p = a_off[j] >= 0 ? src + ((size_t)(unsigned)a_off[j] + (unsigned)(a_chunk[j] + ci0)) : zero;
      glds16(p, sbase + (j * 64 + wave * 8) * 128);
    }
    const int kb = kstep << 6;
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const bf16_t* p = b_ptr[j] != nullptr ? b_ptr[j] + kb : zero;
      glds16(p, sbase + BM * 128 + (j * 64 + wave * 8) * 128);
    }
  };

  // 16x16x32 MFMAs (same cycles per FLOP as 32x32x16; the chip holds a higher clock under them -- mfma_peak.hip), in the
  // transposed product (A = weights, B = pixels): D[channel][pixel], four consecutive channels of one pixel in the
  // four registers of a 16x16 block, so the epilogue stages 8-byte packs.
  constexpr int PB = WTM / 16, CB = WTN / 16;       // 16-pixel / 16-channel blocks per wave
  f32x4 acc[PB][CB];
#pragma unroll
  for (int i = 0; i < PB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int l16 = lane & 15, kg = lane >> 4;        // lane roles: row of the block / 8-channel k-group

  // Fragment reads are software-pipelined against the MFMAs through two register sets: the ds_reads of k-block
  // ks+1 are issued BEFORE the MFMAs of ks.  (With one set, a ds_read that overwrites the source registers of a
  // queued MFMA cannot issue until that MFMA has read them, so LDS latency and MFMA time add up instead of overlapping.)
  auto compute = [&](int stage) {
    const unsigned char* ab = smem + stage * STAGE_BYTES;
    const unsigned char* bb = ab + BM * 128;
    u32x4 af[2][PB], bf[2][CB];
    auto load_frags = [&](int ks, u32x4 (&a)[PB], u32x4 (&b)[CB]) {
      const int chunk = ks * 4 + kg;                 // this lane's 16-byte chunk of the 128-byte k-row
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int row = wm * WTM + i * 16 + l16;
        a[i] = *reinterpret_cast<const u32x4*>(ab + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const int row = wn * WTN + j * 16 + l16;
        b[j] = *reinterpret_cast<const u32x4*>(bb + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
      }
    };
    load_frags(0, af[0], bf[0]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks + 1 < 2) load_frags(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);   // in flight under the MFMAs of ks
      __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of the MFMAs (else the two register sets are merged)
#pragma unroll
      for (int i = 0; i < PB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bf[ks & 1][j]),
                                                               __builtin_bit_cast(bf16x8, af[ks & 1][i]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- LDS ring, one barrier per k-step ----
  //   iteration it:  wait(stage it landed) ; barrier ; issue(stage it+STAGES-1) ; compute(stage it)
  // barrier(it) is passed only after every wave finished compute(it-1), so the stage that held it-1 is free to refill.
  // STAGES = 3: two stages of loads stay in flight under the MFMAs; STAGES = 2 (256x256 tiles): one.
  constexpr int AHEAD = STAGES - 1;
  issue(0, 0);
  if (AHEAD > 1 && nk > 1) issue(1, 1);
  const unsigned long long t1 = ablate == 5 ? __builtin_amdgcn_s_memtime() : 0ull;
  for (int it = 0; it < nk; ++it) {
    if (AHEAD > 1 && it + 1 < nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA + LB) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (it + AHEAD < nk && ablate != 1 && ablate != 3) issue((it + AHEAD) % STAGES, it + AHEAD);
    if (ablate != 2 && ablate != 3) compute(it % STAGES);
  }
  const unsigned long long t2 = ablate == 5 ? __builtin_amdgcn_s_memtime() : 0ull;
  if (ablate == 4) return;
  // ---- epilogue ----
  // D row = channel 4*kg + e of its 16-block, col = pixel l16.  The tile is staged through the (now idle)
  // LDS ring as bf16 [pixel][BN (+8 pad)] with 8-byte writes (row stride 16*odd bytes: 2-way instead of 32-way bank
  // conflicts) and written back with 16-byte stores, one 2*BN-byte row per BN/8 lanes.
  __syncthreads();                       // every wave is done reading the ring
  constexpr int CROW = BN * 2 + 16;
  unsigned char* ctile = smem;
  const float slope = act_slope(act);      // branch-free activation + padded-channel bit masks: see common.h
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int col0 = wn * WTN + j * 16 + 4 * kg;                       // first of this lane's four channels
    float bq[4];
    uint32_t m01, m23;
    epi_col_consts(bias, n0 + col0, wrows, bq, m01, m23);
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int row = wm * WTM + i * 16 + l16;
      *reinterpret_cast<u32x2*>(ctile + row * CROW + col0 * 2) =
          epi_finish4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3], bq, slope, m01, m23);
    }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;             // 16-byte chunks per tile row
  constexpr int RPP = 512 / CPR;          // rows per pass
  const int chunk = tid % CPR, rsub = tid / CPR;
  const int ncol = n0 + chunk * 8;
  if (ncol < ldc) {
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
      const int row = p * RPP + rsub;
      const int m = m0 + row;
      if (m >= g.M) continue;
      size_t opix;
      if (g.out_identity) {
        opix = (size_t)m;
      } else {
        int n, oy, ox;
        decode_m(g, m, n, oy, ox);
        opix = (size_t)out_pixel(g, n, oy, ox);
      }
      *reinterpret_cast<u32x4*>(out + opix * ldc + ncol) = *reinterpret_cast<const u32x4*>(ctile + row * CROW + chunk * 16);
    }
  }
  if (ablate == 5 && dbg != nullptr && tid == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    unsigned long long* d = dbg + (size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 4;
    d[0] = t0; d[1] = t1; d[2] = t2; d[3] = t3;
  }
}

int g_v2_ablate = 0;
unsigned long long* g_v2_dbg = nullptr;

template <int BM, int BN, int WM, int WN, int STAGES>
static hipError_t launch_v2(const DescPack& pack, const void* src, const void* wgt, int wrows, const float* bias, void* out,
                            float* ws, int ldc, int act, int splits, hipStream_t st) {
  int tiles_m = 0, Kmax = 0;
  double flops = 0.0;
  for (int i = 0; i < pack.n; ++i) {
    const GatherDesc& g = pack.d[i];
    tiles_m = std::max(tiles_m, (g.M + BM - 1) / BM);
    Kmax = std::max(Kmax, g.K);
    flops += 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)wrows;
  }
  const int tiles_n = (ldc + BN - 1) / BN;
  const int nk = Kmax >> 6;
  const int kps = (nk + splits - 1) / splits;
  const int zs = (nk + kps - 1) / kps;
  const size_t lds = STAGES * (size_t)(BM + BN) * 128 + 16 * BM * sizeof(int);
  auto kern = gather_gemm_v2_kernel<BM, BN, WM, WN, STAGES>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  count_launch(K_GATHER_V2);
  prof_begin(PROF_GATHER_GEMM, flops, st);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n, pack.n, zs), dim3(512), lds, st, pack, (const bf16_t*)src,
                     (const bf16_t*)wgt, wrows, bias, (bf16_t*)out, zs > 1 ? ws : (float*)nullptr, ldc, act, tiles_n, kps, g_v2_ablate, g_v2_dbg);
  prof_end(PROF_GATHER_GEMM, st);
  return hipGetLastError();
}

// returns hipErrorNotSupported when the shape does not qualify (caller falls through to the v1 kernel)
hipError_t gather_gemm_v2(const DescPack& pack, const void* src, const void* wgt, int wrows, const float* bias, void* out,
                          float* ws, size_t ws_bytes, int ldc, int act, int num_cu, hipStream_t st) {
  int tiles256 = 0, tiles192 = 0, nk = 0;
  for (int i = 0; i < pack.n; ++i) {
    const GatherDesc& g = pack.d[i];
    if (g.Cs % 64 != 0 || g.K % 64 != 0 || g.th * g.tw > 16) return hipErrorNotSupported;
    if (g.wK != g.K || g.wtw != g.tw) return hipErrorNotSupported;        // sub-tap views: v1 only
    if ((long long)g.N * g.Hs * g.Ws * g.Cs >= (1ll << 31)) return hipErrorNotSupported;      // 32-bit offset table
    tiles256 += (g.M + 255) / 256;
    tiles192 += (g.M + 191) / 192;
    nk = std::max(nk, g.K >> 6);
  }
  if (ldc <= 32) return hipErrorNotSupported;
  (void)ws_bytes;
  // One 8-wave workgroup owns a CU (LDS), every tile of a launch takes the same time, so cost ~ rounds x tile size.
  if (ldc >= 256 && ldc % 256 == 0) {
    // measured: 256x256 wins (~1.2x) when all tiles run in ONE round; with 2+ rounds the exposed 128 KB epilogue and the
    // shallower 2-stage prefetch eat the gain (res-block dgrad 189 vs 185 us, dec0 dgrad 536 vs 350 us)
    const int t256 = tiles256 * (ldc / 256);
    if (t256 >= num_cu / 2 && t256 <= num_cu)
      return launch_v2<256, 256, 2, 4, 2>(pack, src, wgt, wrows, bias, out, nullptr, ldc, act, 1, st);
  }
  const int BN = ldc > 64 ? 128 : 64;
  const int ntn = (ldc + BN - 1) / BN;
  if (tiles256 * ntn < num_cu / 2) return hipErrorNotSupported;   // small-M layers: v1 (128-row tiles, split-K) fills the chip better
  if (BN == 128) {
    // ragged M (dgrad on the padded frame: 69696 rows -> 546 tiles = 2.13 rounds): 192-row tiles waste less of the last round
    const long long c256 = (long long)((tiles256 * ntn + num_cu - 1) / num_cu) * 256;
    const long long c192 = (long long)((tiles192 * ntn + num_cu - 1) / num_cu) * 192;
    if (c192 < c256) return launch_v2<192, 128, 2, 4, 3>(pack, src, wgt, wrows, bias, out, nullptr, ldc, act, 1, st);
    return launch_v2<256, 128, 4, 2, 3>(pack, src, wgt, wrows, bias, out, nullptr, ldc, act, 1, st);
  }
  return launch_v2<256, 64, 4, 2, 3>(pack, src, wgt, wrows, bias, out, nullptr, ldc, act, 1, st);
}

}  // namespace dei2i
