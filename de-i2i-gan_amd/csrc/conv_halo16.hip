// Halo-resident conv, 16 x 32 pixel tiles with 32-channel slices (bf16, stride 1, 3x3 taps): the large-batch forward
// convs and zero-boundary dgrads of the 3x3 layers.
//
// conv_halo.hip's loop is bound by the CU's global->LDS issue rate, not by the MFMA pipe: one 16-byte-per-lane LDS-DMA
// instruction occupies the address path for ~64 cycles, its k-step (256 pixels x 128 channels x 64 input channels,
// 4.2 MFLOP) moves 16 KB of weights + 4.8 KB of halo = 20.8 instructions ~ 1 300 cycles against 1 024 cycles of MFMA
// issue, and measures 1 650.  Weight bytes per FLOP only depend on the PIXEL tile, so this kernel doubles it:
//
//   tile      : 16 x 32 = 512 output pixels of one image x BN output channels per workgroup
//   k-step    : one tap x 32 input channels (one 16x16x32 MFMA k-block) -- the same 4.2 MFLOP per k-step for BN = 128
//   LDS       : halo[2] of (16+2) x (32+2) pixels x 32 channels (2 x 39 KB) | weight ring STAGES x BN x 64 B | offset table
//   DMA       : 8 KB of weights + 4.3 KB of halo per k-step = 12.3 instructions ~ 790 cycles: under the MFMA issue time
//   fragments : 8 pixel blocks + 4 channel blocks per wave and k-step (12 KB of LDS reads per 32 MFMAs; 16 KB before)
//   LDS image : 64-byte rows; 16-byte chunk c of row r sits at slot c ^ (((r >> 2) & 1) << 1) -- every ds_read_b128 of 16
//               consecutive rows x 4 k-groups is conflict-free at ANY starting row (all lane groups of the instruction
//               see 16 distinct (row & 3, slot) pairs), so the tap-shifted pixel fragments never conflict (the 128-byte
//               rows of conv_halo.hip lose 5.9 % of their cycles there)
//   grid      : the res-block shape at batch 16 (16 x 64 x 64 x 256 -> 256) is 128 tiles x 2 channel tiles = 256
//               workgroups: ONE round on the 256 CUs, and half as many prologues / epilogues per FLOP
//
// Waves, phases and the LDS-DMA protocol are those of conv_halo.hip (two wave groups in anti-phase: M(j) = fragment reads
// of k-step j, C(j) = its 32 MFMAs with the DMA issue in the gaps, one workgroup barrier per phase).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <type_traits>

#include "common.h"
#include "geom.h"
#include "launch.h"

namespace dei2i {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_h16[256];

typedef __attribute__((address_space(3))) void lds_void_h16;
typedef __attribute__((address_space(1))) const void gbl_void_h16;

DEI2I_D void glds16x(const void* gptr, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_h16*)gptr, (lds_void_h16*)lds_wave_base, 16, 0, 0);
}

DEI2I_D int xcd_remap_h16(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

constexpr int H16_TH = 16, H16_TW = 32;
constexpr int H16_GROUPS = 39;                        // 16-pixel LDS-DMA groups per halo slice (624 >= 18*34 pixels)
constexpr int H16_HBYTES = H16_GROUPS * 1024;         // 39,936
constexpr int H16_HL = 5;                             // halo LDS-DMA instructions per wave per slice (40 >= 39 groups)

template <int N> DEI2I_D void wait_vm16() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
DEI2I_D int sw16(int row) { return ((row >> 2) & 1) << 1; }

// DIAG: a diagnostic build (option "v2_ablate" = 6) that accumulates s_memtime stamps per wave into `dbg`:
//   [0] loop cycles  [1] loop s_memrealtime ticks  [2] k-steps  [3] whole-kernel cycles
//   [4] M: fragment reads issued  [5] M: vmcnt wait  [6] M: lgkmcnt wait  [7] M: barrier  [8] C: MFMA + DMA issue  [9] C: barrier
// FOLD: this launch is the input gradient of a REFLECT-padded conv (architecture.py:51-56: pad 1).  The zero-boundary dgrad
// on the input grid misses what the padded frame's ring (row -1 / H, column -1 / W) reflects back onto rows 1 / H-2 and
// columns 1 / W-2 -- conv_api.hip computes that ring as a second small GEMM + split-K finalize + border fold (three
// launches, ~30 us per conv).  Here the tiles that touch the image border compute their piece of the ring themselves:
// ring row -1 over the tile's 32 columns is dx[-1][x] = sum_tx W[ky=0][kx=tx]^T dy[0][x+1-tx] -- the tile's own halo,
// addressed one tile row above row 0, with the ONE tap row that still reads inside dy -- as two extra 16-pixel blocks
// (waves wm = 0, 1), ring column -1 / W over the tile's 16 rows as one extra block of 16 rows x 1 column (waves wm = 2 /
// 3); each is 4 extra MFMAs per wave in 3 of the 9 taps, accumulated apart and added to rows 1 / 14 or columns 1 / 30 of
// the LDS-staged tile before it is written.  The four frame CORNERS (one pixel each) are left to reflect_corner_kernel.
template <int BN, int STAGES, bool DIAG = false, bool FOLD = false>
__global__ __launch_bounds__(512) void halo16_conv_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                          const bf16_t* __restrict__ wgt, const int wrows,
                                                          const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                          const int ldc, const int act, const int tiles_n,
                                                          float* __restrict__ stats, unsigned long long* __restrict__ dbg,
                                                          const bf16_t* __restrict__ zring, const int ring_pix) {
  // zring (optional): the input tensor is the SOURCE-resolution z of a SPADE -> upsample -> conv block whose 2-pixel frame
  // (at the logical, upsampled resolution) has per-pixel gamma / beta classes: those halo pixels come from the compact
  // ring tensor [N][ring_pix][Cs] (geom.h) instead of from src[y >> up][x >> up]
  const unsigned long long kt0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned long long dg[6] = {0, 0, 0, 0, 0, 0};
  auto now = [&]() -> unsigned long long {
    if (!DIAG) return 0ull;
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
  };
  constexpr int WTN = BN / 2;                         // channels per wave (2 waves along N)
  constexpr int PB = 8, CB = WTN / 16;                // 16-pixel blocks / 16-channel blocks per wave
  constexpr int B_STAGE = BN * 64;
  constexpr int LB = 1;                               // weight LDS-DMA instructions per wave per stage
  static_assert(STAGES >= 4 && STAGES <= 8 && (BN == 128 || BN == 64), "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const halo = smem;
  unsigned char* const ring = smem + 2 * H16_HBYTES;
  int* const htab = reinterpret_cast<int*>(ring + STAGES * B_STAGE);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  const int bid = xcd_remap_h16(blockIdx.x, gridDim.x);
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int tiles_x = g.Wo / H16_TW, tiles_y = g.Ho / H16_TH;
  const int img = tile_m / (tiles_x * tiles_y);
  const int trem = tile_m - img * (tiles_x * tiles_y);
  const int y0 = (trem / tiles_x) * H16_TH, x0 = (trem % tiles_x) * H16_TW;
  const int n0 = tile_n * BN;

  const int hwd = H16_TW + g.tw - 1;                  // halo width in pixels
  const int npix = (H16_TH + g.th - 1) * hwd;
  const int hy0 = y0 + g.by0 + (g.ys < 0 ? -(g.th - 1) : 0);
  const int hx0 = x0 + g.bx0 + (g.xs < 0 ? -(g.tw - 1) : 0);

  // ---- halo source-offset table (element offset of each halo pixel's channel 0, -1 = contributes zero) ----
  for (int p = tid; p < H16_GROUPS * 16; p += 512) {
    int off = -1;
    if (p < npix) {
      const int hy = p / hwd, hx = p - hy * hwd;
      const int y = bound_coord(hy0 + hy, g.Hl, g.pad_mode);
      const int x = bound_coord(hx0 + hx, g.Wl, g.pad_mode);
      if ((y | x) >= 0) {
        if (zring != nullptr && !(ring_interior(y, g.Hl) && ring_interior(x, g.Wl)))
          off = -2 - (img * ring_pix + ring_index(y, x, g.Hl, g.Wl)) * g.Cs;       // <= -2: element offset -2 - off into zring
        else
          off = ((img * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up)) * g.Cs;
      }
    }
    htab[p] = off;
  }
  __syncthreads();

  // ---- per-lane LDS-DMA roles: one instruction = 16 rows x 64 B; lane l -> row (l >> 2), slot (l & 3) ----
  const int lrow = lane >> 2, lslot = lane & 3;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_h16);
  int h_off[H16_HL], h_group[H16_HL];
#pragma unroll
  for (int j = 0; j < H16_HL; ++j) {
    int grp = j * 8 + wave;                            // 40 slots for 39 groups: the surplus re-fetches group 0
    if (grp >= H16_GROUPS) grp -= H16_GROUPS;          // (identical bytes land twice; keeps vmcnt uniform across waves)
    const int pix = grp * 16 + lrow;
    const int o = htab[pix];
    h_group[j] = grp;
    const int so = (lslot ^ sw16(pix)) << 3;
    h_off[j] = o >= 0 ? o + so : (o == -1 ? -1 : o - so);
  }
  const int rg = BN == 128 ? wave : (wave & 3);        // 16-row group of the weight stage this wave fetches
  const bf16_t* b_ptr;
  {
    const int r = rg * 16 + lrow;
    const int n = n0 + r;
    b_ptr = n < wrows ? wgt + (size_t)n * g.K + ((lslot ^ sw16(r)) << 3) : nullptr;
  }

  const int ntaps = __builtin_amdgcn_readfirstlane(g.th * g.tw);
  const int nslices = __builtin_amdgcn_readfirstlane(g.Cs >> 5);
  const int nk = ntaps * nslices;

  auto issue_halo = [&](int slice) {
    unsigned char* hb = halo + (slice & 1) * H16_HBYTES;
    const int ci0 = slice << 5;
#pragma unroll
    for (int j = 0; j < H16_HL; ++j) {
      const bf16_t* p = h_off[j] >= 0 ? src + ((size_t)(unsigned)h_off[j] + (unsigned)ci0) : zero;
      if (h_off[j] < -1) p = zring + ((size_t)(unsigned)(-2 - h_off[j]) + (unsigned)ci0);
      glds16x(p, hb + h_group[j] * 1024);
    }
  };
  int is_tap = 0, is_slice = 0;                        // (tap, slice) of the next weight k-step to issue
  auto issue_b = [&](int stage) {
    const int kb = is_tap * g.Cs + (is_slice << 5);
    const bf16_t* p = b_ptr != nullptr ? b_ptr + kb : zero;
    glds16x(p, ring + stage * B_STAGE + rg * 1024);
    if (++is_tap == ntaps) { is_tap = 0; ++is_slice; }
  };

  // transposed product (A = weights, B = pixels): D[channel 4*kg + e of its 16-block][pixel l16]
  typedef __attribute__((ext_vector_type(4))) float acc_t;
  acc_t acc[PB][CB];
#pragma unroll
  for (int i = 0; i < PB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  const int l16 = lane & 15, kg = lane >> 4;
  struct Frags { u32x4 a[PB]; u32x4 b[CB]; };
  int a_pix0[PB];                                      // halo pixel of this lane's pixel for tap offset 0
#pragma unroll
  for (int i = 0; i < PB; ++i) a_pix0[i] = (wm * 4 + (i >> 1)) * hwd + (i & 1) * 16 + l16;
  int b_addr[CB];                                      // byte address of this lane's weight chunk within a stage
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int row = wn * WTN + j * 16 + l16;
    b_addr[j] = row * 64 + ((kg ^ sw16(row)) << 4);
  }
  // FOLD: this wave's ring block (wave-uniform kind: 0 none, 1 top row, 2 bottom row, 3 left column, 4 right column)
  int ring_kind = 0, ring_pix0 = 0;
  acc_t racc[CB];
  u32x4 ring_frag;
  bool ring_live = false;
  if constexpr (FOLD) {
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) racc[j][e] = 0.f;
    if (wm < 2) {
      if (y0 == 0) { ring_kind = 1; ring_pix0 = -hwd + wm * 16 + l16; }
      else if (y0 + H16_TH == g.Ho) { ring_kind = 2; ring_pix0 = H16_TH * hwd + wm * 16 + l16; }
    } else if (wm == 2) {
      if (x0 == 0) { ring_kind = 3; ring_pix0 = l16 * hwd - 1; }
    } else if (x0 + H16_TW == g.Wo) {
      ring_kind = 4;
      ring_pix0 = l16 * hwd + H16_TW;
    }
    ring_kind = __builtin_amdgcn_readfirstlane(ring_kind);
  }
  int ld_tx = 0, ld_ty = 0, ld_slice = 0, ld_stage = 0;
  const int step_x = g.xs > 0 ? 1 : -1, step_y = g.ys > 0 ? hwd : -hwd;
  int ld_toff = (g.ys > 0 ? 0 : (g.th - 1) * hwd) + (g.xs > 0 ? 0 : g.tw - 1);     // halo pixel offset of tap (0,0)
  const int toff_row_wrap = step_y - (g.tw - 1) * step_x;
  const int toff_origin = ld_toff;
  auto read_frags = [&](Frags& f) {
    const unsigned char* hb = halo + (ld_slice & 1) * H16_HBYTES;
    const unsigned char* sb = ring + ld_stage * B_STAGE;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int pix = a_pix0[i] + ld_toff;
      f.a[i] = *reinterpret_cast<const u32x4*>(hb + pix * 64 + ((kg ^ sw16(pix)) << 4));
    }
    if constexpr (FOLD) {
      // the ring row / column only sees the tap row / column that still reads inside dy (dgrad taps: dy row = oy + 1 - ty)
      ring_live = ring_kind == 1 ? ld_ty == 0 : (ring_kind == 2 ? ld_ty == 2 : (ring_kind == 3 ? ld_tx == 0 : (ring_kind == 4 && ld_tx == 2)));
      if (ring_live) {
        const int pix = ring_pix0 + ld_toff;
        ring_frag = *reinterpret_cast<const u32x4*>(hb + pix * 64 + ((kg ^ sw16(pix)) << 4));
      }
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) f.b[j] = *reinterpret_cast<const u32x4*>(sb + b_addr[j]);
    if (++ld_tx == g.tw) {
      ld_tx = 0;
      if (++ld_ty == g.th) { ld_ty = 0; ++ld_slice; ld_toff = toff_origin; }
      else ld_toff += toff_row_wrap;
    } else {
      ld_toff += step_x;
    }
    if (++ld_stage == STAGES) ld_stage = 0;
  };

  // ---- main loop: two wave groups in anti-phase (see conv_halo.hip) ----
  //   C(j) issues weights(j-1+STAGES) into stage (j-1) % STAGES (last read by the trailing group's M(j-1), two phases
  //        earlier) and, at tap 1 of a slice, the NEXT slice's halo (into the buffer of the previous slice).
  //   M(j) ends with "my share of weights(j+1) has landed" + barrier.  weights(j+1) was issued in C(j+2-STAGES); issued
  //        after it: the weights of C(j+3-STAGES .. j-1) = (STAGES-3) stages, and the halo of C(tap 1) while
  //        2 <= tap <= STAGES-1 -- exactly those may stay in flight.  The last STAGES-2 k-steps drain everything.
  const int grp = wave >> 2;
  int tap = 0, slice = 0;                              // k-step of this wave's current M / C phase
  Frags f;
  auto phase_m = [&](int j) {
    const unsigned long long q0 = now();
    read_frags(f);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long q1 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    const bool halo_young = tap >= 2 && tap <= STAGES - 1 && slice + 1 < nslices;
    if (j + STAGES - 2 >= nk) wait_vm16<0>();
    else if (halo_young) wait_vm16<(STAGES - 3) * LB + H16_HL>();
    else wait_vm16<(STAGES - 3) * LB>();
    const unsigned long long q2 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long q3 = now();
    __builtin_amdgcn_s_barrier();
    if (DIAG) {
      const unsigned long long q4 = now();
      dg[0] += q1 - q0; dg[1] += q2 - q1; dg[2] += q3 - q2; dg[3] += q4 - q3;
    }
  };
  auto phase_c = [&](int j) {
    const unsigned long long q0 = now();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < CB; ++jj)
        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.b[jj]), __builtin_bit_cast(bf16x8, f.a[i]),
                                                             acc[i][jj], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (j - 1 + STAGES < nk) issue_b((j + STAGES - 1) % STAGES);
    __builtin_amdgcn_sched_barrier(0);
    if (tap == 1 && slice + 1 < nslices) issue_halo(slice + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 2; i < PB; ++i)
#pragma unroll
      for (int jj = 0; jj < CB; ++jj)
        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.b[jj]), __builtin_bit_cast(bf16x8, f.a[i]),
                                                             acc[i][jj], 0, 0, 0);
    if constexpr (FOLD) {
      if (ring_live) {
#pragma unroll
        for (int jj = 0; jj < CB; ++jj)
          racc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.b[jj]), __builtin_bit_cast(bf16x8, ring_frag),
                                                             racc[jj], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long q1 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    __builtin_amdgcn_s_barrier();
    if (DIAG) {
      const unsigned long long q2 = now();
      dg[4] += q1 - q0; dg[5] += q2 - q1;
    }
    if (++tap == ntaps) { tap = 0; ++slice; }
  };

  const unsigned long long st0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long sr0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0ull;
  // prologue: halo slice 0 and weights(0 .. STAGES-2); weights(STAGES-1) is issued by C(0).  nk >= 9 > STAGES-1.
  issue_halo(0);
  for (int s2 = 0; s2 < STAGES - 1; ++s2) issue_b(s2);
  wait_vm16<(STAGES - 2) * LB>();                       // everything older than weights(1): halo 0 and weights(0)
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();           // group 1 starts one phase late
#pragma unroll 1
  for (int j = 0; j < nk; ++j) {
    phase_m(j);
    phase_c(j);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();           // pairs with group 1's last phase
  unsigned long long* drec = nullptr;
  if (DIAG && dbg != nullptr && lane == 0) {
    drec = dbg + ((size_t)blockIdx.x * 8 + wave) * 10;
    drec[0] = __builtin_amdgcn_s_memtime() - st0;
    drec[1] = __builtin_amdgcn_s_memrealtime() - sr0;
    drec[2] = (unsigned long long)nk;
    for (int q = 0; q < 6; ++q) drec[4 + q] = dg[q];
  }

  // ---- epilogue, in two halves of 8 tile rows: stage the half as bf16 [pixel][BN (+8 pad)] (8-byte writes: a lane
  //      holds 4 consecutive channels of its pixel), write back with 16-byte stores ----
  constexpr int CROW = BN * 2 + 16;
  unsigned char* ctile = smem;
  const float slope = act_slope(act);
  constexpr int CPR = BN / 8;             // 16-byte chunks per tile row
  constexpr int RPP = 512 / CPR;          // rows per pass
  const int chunk = tid % CPR, rsub = tid / CPR;
  const int ncol = n0 + chunk * 8;
  // statistics: one record per 8 x 32 half tile, i.e. the record grid of the 8 x 32 tile kernel (conv_halo.hip) and of
  // dei2i_conv2d_stats_chunks: record (2 * tile_row + half, tile_col) of the image
  const int rec0 = img * (2 * tiles_y * tiles_x) + (2 * (trem / tiles_x)) * tiles_x + (trem % tiles_x);
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    float st8[16];                        // sum (0..7) / sum of squares (8..15) of this thread's 8 channels
#pragma unroll
    for (int k = 0; k < 16; ++k) st8[k] = 0.f;
    __syncthreads();
    if ((wm >> 1) == half) {
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const int col0 = wn * WTN + j * 16 + 4 * kg;
        float bq[4];
        uint32_t m01, m23;
        epi_col_consts(bias, n0 + col0, wrows, bq, m01, m23);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          const int row = ((wm & 1) * 4 + (i >> 1)) * 32 + (i & 1) * 16 + l16;
          *reinterpret_cast<u32x2*>(ctile + row * CROW + col0 * 2) =
              epi_finish4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3], bq, slope, m01, m23);
        }
      }
    }
    __syncthreads();
    if constexpr (FOLD) {
      // the ring's reflection: row ring -> tile row 1 (top) / 14 (bottom), then -- after a barrier: pixel (1,1) takes both --
      // column ring -> column 1 (left) / 30 (right) of each row
      auto ring_add = [&](int row) {
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          u32x2* p = reinterpret_cast<u32x2*>(ctile + row * CROW + (wn * WTN + j * 16 + 4 * kg) * 2);
          const u32x2 v = *p;
          u32x2 o;
          o.x = (uint32_t)f32_to_bf16(__uint_as_float(v.x << 16) + racc[j][0]) |
                ((uint32_t)f32_to_bf16(__uint_as_float(v.x & 0xffff0000u) + racc[j][1]) << 16);
          o.y = (uint32_t)f32_to_bf16(__uint_as_float(v.y << 16) + racc[j][2]) |
                ((uint32_t)f32_to_bf16(__uint_as_float(v.y & 0xffff0000u) + racc[j][3]) << 16);
          *p = o;
        }
      };
      if ((ring_kind == 1 && half == 0) || (ring_kind == 2 && half == 1)) ring_add((ring_kind == 1 ? 1 : 6) * 32 + wm * 16 + l16);
      __syncthreads();
      if (ring_kind >= 3 && (l16 >> 3) == half) ring_add((l16 & 7) * 32 + (ring_kind == 3 ? 1 : H16_TW - 2));
      __syncthreads();
    }
    if (ncol < ldc) {
#pragma unroll
      for (int p = 0; p < 256 / RPP; ++p) {
        const int row = p * RPP + rsub;
        const size_t opix = (size_t)out_pixel(g, img, y0 + half * 8 + (row >> 5), x0 + (row & 31));
        const u32x4 v = *reinterpret_cast<const u32x4*>(ctile + row * CROW + chunk * 16);
        *reinterpret_cast<u32x4*>(out + opix * ldc + ncol) = v;
        if (stats != nullptr) {
          float fv[8];
          Elem<bf16_t>::unpack(v, fv);
#pragma unroll
          for (int e = 0; e < 8; ++e) { st8[e] += fv[e]; st8[8 + e] = fmaf(fv[e], fv[e], st8[8 + e]); }
        }
      }
    }
    if (stats != nullptr) {                              // kernel-uniform: per-thread partials -> LDS -> ordered sums
      float* red = reinterpret_cast<float*>(smem + 256 * CROW);
      float* mine = red + ((size_t)rsub * CPR + chunk) * 16;
#pragma unroll
      for (int k = 0; k < 16; k += 4) {
        f32x4 t;
        t.x = st8[k]; t.y = st8[k + 1]; t.z = st8[k + 2]; t.w = st8[k + 3];
        *reinterpret_cast<f32x4*>(mine + k) = t;
      }
      __syncthreads();
      for (int o = tid; o < CPR * 16; o += 512) {
        const int ch = o >> 4, k = o & 15;
        float sum = 0.f;
        for (int r = 0; r < RPP; ++r) sum += red[((size_t)r * CPR + ch) * 16 + k];
        const int c = n0 + ch * 8 + (k & 7);
        if (c < ldc) stats[((size_t)(rec0 + half * tiles_x) * 2 + (k >> 3)) * ldc + c] = sum;
      }
    }
  }
  if (DIAG && drec != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    drec[3] = __builtin_amdgcn_s_memtime() - kt0;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same tile with FOUR waves, one per SIMD, each software-pipelined (option "halo16" = 2).
//
// Stamps of the 8-wave kernel above (tools/diag_halo16_stamps.py, res-block shape): a C phase of 32 MFMAs takes 820-1010
// cycles against 512 of bare issue, an M phase of 12 fragment reads 420-460 cycles to issue -- the two waves of a SIMD
// take issue slots from each other (the MFMA holds the vector issue 8 of its 16 cycles, the partner's address math, LDS
// reads and LDS-DMA go through the same port), so the anti-phase pairing overlaps much less than it intends.  Here a
// wave owns its SIMD: 4 tile rows (128 pixels) x all BN channels, 32x32x16 MFMAs (32 per k-step of 32 channels = 1 024
// cycles, 256 accumulator registers), and between those MFMAs it issues the 16 fragment reads of the NEXT k-step into a
// second register set plus its share of the LDS-DMA (2 weight instructions per k-step, 10 halo instructions per slice):
// one s_barrier per k-step, nothing else serialises.  LDS image: 64-byte rows, chunk c of row r at slot c ^ ((r >> 2) & 3)
// (conflict-free for 32 consecutive rows at any start).
// ---------------------------------------------------------------------------------------------------------------------
extern int g_v2_ablate;
extern unsigned long long* g_v2_dbg;
DEI2I_D int sw32(int row) { return (row >> 2) & 3; }
constexpr int W4_HL = 10;                              // halo LDS-DMA instructions per wave per slice (40 >= 39 groups)

template <int BN, int STAGES, bool DIAG = false>
__global__ __launch_bounds__(256) void halo16w4_conv_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                            const bf16_t* __restrict__ wgt, const int wrows,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                            const int ldc, const int act, const int tiles_n,
                                                            float* __restrict__ stats, unsigned long long* __restrict__ dbg) {
  const unsigned long long kt0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned long long dg[6] = {0, 0, 0, 0, 0, 0};          // DIAG: [half 0 | address prep | half 1 | halo issue | waits | barrier]
  auto now = [&]() -> unsigned long long {
    if (!DIAG) return 0ull;
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
  };
  constexpr int PB = 4, CB = BN / 32;                  // 32-pixel blocks (the wave's 4 tile rows) / 32-channel blocks
  constexpr int B_STAGE = BN * 64;
  constexpr int LB = BN / 64;                          // weight LDS-DMA instructions per wave per stage (16 rows each)
  static_assert(STAGES >= 4 && STAGES <= 8 && (BN == 128 || BN == 64), "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const halo = smem;
  unsigned char* const ring = smem + 2 * H16_HBYTES;
  int* const htab = reinterpret_cast<int*>(ring + STAGES * B_STAGE);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int bid = xcd_remap_h16(blockIdx.x, gridDim.x);
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int tiles_x = g.Wo / H16_TW, tiles_y = g.Ho / H16_TH;
  const int img = tile_m / (tiles_x * tiles_y);
  const int trem = tile_m - img * (tiles_x * tiles_y);
  const int y0 = (trem / tiles_x) * H16_TH, x0 = (trem % tiles_x) * H16_TW;
  const int n0 = tile_n * BN;

  const int hwd = H16_TW + g.tw - 1;
  const int npix = (H16_TH + g.th - 1) * hwd;
  const int hy0 = y0 + g.by0 + (g.ys < 0 ? -(g.th - 1) : 0);
  const int hx0 = x0 + g.bx0 + (g.xs < 0 ? -(g.tw - 1) : 0);

  for (int p = tid; p < H16_GROUPS * 16; p += 256) {
    int off = -1;
    if (p < npix) {
      const int hy = p / hwd, hx = p - hy * hwd;
      const int y = bound_coord(hy0 + hy, g.Hl, g.pad_mode);
      const int x = bound_coord(hx0 + hx, g.Wl, g.pad_mode);
      if ((y | x) >= 0) off = ((img * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up)) * g.Cs;
    }
    htab[p] = off;
  }
  __syncthreads();

  const int lrow = lane >> 2, lslot = lane & 3;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_h16);
  const bf16_t* b_ptr[LB];
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int r = (j * 4 + wave) * 16 + lrow;
    const int n = n0 + r;
    b_ptr[j] = n < wrows ? wgt + (size_t)n * g.K + ((lslot ^ sw32(r)) << 3) : nullptr;
  }

  const int ntaps = __builtin_amdgcn_readfirstlane(g.th * g.tw);
  const int nslices = __builtin_amdgcn_readfirstlane(g.Cs >> 5);
  const int nk = ntaps * nslices;

  // piece q (0..9) of a halo slice: 16-pixel group (4q + wave) mod 39 (40 slots for 39 groups: one is fetched twice)
  auto issue_halo_part = [&](int slice, int q) {
    unsigned char* hb = halo + (slice & 1) * H16_HBYTES;
    int grp = q * 4 + wave;
    if (grp >= H16_GROUPS) grp -= H16_GROUPS;
    const int pix = grp * 16 + lrow;
    const int o = htab[pix];
    const bf16_t* p = o >= 0 ? src + ((size_t)(unsigned)(o + ((lslot ^ sw32(pix)) << 3)) + (unsigned)(slice << 5)) : zero;
    glds16x(p, hb + grp * 1024);
  };
  int is_tap = 0, is_slice = 0;
  auto issue_b_part = [&](int stage, int j) {           // piece j of the k-step (is_tap, is_slice); the last piece advances it
    const int kb = is_tap * g.Cs + (is_slice << 5);
    const bool live = is_slice < nslices;               // past the last k-step: refill from the zero page (nobody reads it)
    const bf16_t* p = (b_ptr[j] != nullptr && live) ? b_ptr[j] + kb : zero;
    glds16x(p, ring + stage * B_STAGE + (j * 4 + wave) * 1024);
    if (j == LB - 1) {
      if (++is_tap == ntaps) { is_tap = 0; ++is_slice; }
    }
  };
  auto issue_b = [&](int stage) {
#pragma unroll
    for (int j = 0; j < LB; ++j) issue_b_part(stage, j);
  };

  f32x16 acc[PB][CB];
#pragma unroll
  for (int i = 0; i < PB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  // Two fragment register sets, one per K16 half of a k-step: while the MFMAs of one half run, the other set is loaded --
  // R1 <- half 1 of k-step j during the MFMAs of half 0, R0 <- half 0 of k-step j+1 during the MFMAs of half 1 -- so the loop
  // body exists ONCE (no second copy of the accumulator-updating code: the register allocator keeps the 256 accumulators
  // in place) and every read has half a k-step (512 cycles of MFMA issue) to land.
  struct Half { u32x4 a[PB]; u32x4 b[CB]; };
  int ld_stage_issue = STAGES - 1;                      // stage the next issue_b fills: (j - 1) mod STAGES in k-step j
  int a_pix0[PB];
#pragma unroll
  for (int i = 0; i < PB; ++i) a_pix0[i] = (wave * 4 + i) * hwd + lr;
  int b_addr[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int row = j * 32 + lr;
    b_addr[j] = row * 64 + ((lh ^ sw32(row)) << 4);
  }
  int ld_tx = 0, ld_ty = 0, ld_slice = 0, ld_stage = 0;
  const int step_x = g.xs > 0 ? 1 : -1, step_y = g.ys > 0 ? hwd : -hwd;
  int ld_toff = (g.ys > 0 ? 0 : (g.th - 1) * hwd) + (g.xs > 0 ? 0 : g.tw - 1);
  const int toff_row_wrap = step_y - (g.tw - 1) * step_x;
  const int toff_origin = ld_toff;
  // byte offsets (into smem) of this lane's chunk of each fragment of the k-step being loaded; K16 half 1 = the same
  // address with chunk bit 1 flipped (^ 32).  la / lbase serve the reads in flight, la_n / lbase_n are computed for the
  // next k-step one piece per MFMA row (vector ALU work between MFMAs is free; as a block in front of them it is not).
  int la[PB], la_n[PB];
  int lbase, lbase_n;
  int pn_hb = 0, pn_toff = 0;                          // scalar state of the k-step being prepared
  auto prep_scalar = [&]() {
    pn_hb = (ld_slice & 1) * H16_HBYTES;
    pn_toff = ld_toff;
    lbase_n = 2 * H16_HBYTES + ld_stage * B_STAGE;
    if (++ld_tx == g.tw) {
      ld_tx = 0;
      if (++ld_ty == g.th) { ld_ty = 0; ++ld_slice; ld_toff = toff_origin; }
      else ld_toff += toff_row_wrap;
    } else {
      ld_toff += step_x;
    }
    if (++ld_stage == STAGES) ld_stage = 0;
  };
  auto prep_piece = [&](int i) {
    const int pix = a_pix0[i] + pn_toff;
    la_n[i] = pn_hb + pix * 64 + ((lh ^ sw32(pix)) << 4);
  };
  auto prep_commit = [&]() {
#pragma unroll
    for (int i = 0; i < PB; ++i) la[i] = la_n[i];
    lbase = lbase_n;
  };
  auto rd_a = [&](Half& f, int h, int i) { f.a[i] = *reinterpret_cast<const u32x4*>(smem + (la[i] ^ (h << 5))); };
  auto rd_b = [&](Half& f, int h, int j) { f.b[j] = *reinterpret_cast<const u32x4*>(smem + lbase + (b_addr[j] ^ (h << 5))); };
  // LDS-DMA issue: the weight pieces in the row gaps 1.. of half 0, the next slice's halo after the MFMAs of tap 1.
  // (Tried: staggering the pieces over the 8 row gaps by wave -- every wave in its own gaps, the halo two pieces per k-step
  // over taps 1..5 -- so that the four waves' 1 KB instructions do not queue behind each other on the CU's address path:
  // the per-gap scalar branches and the offset-table reads it needs inside the MFMA stream cost more than the queueing,
  // 108 us against 85 us on the res-block shape.)
  int tap = 0, slice = 0;
  auto dma_gap = [&](int gidx) {
    if (gidx >= 1 && gidx - 1 < LB) issue_b_part(ld_stage_issue, gidx - 1);
  };
  // the 16 (BN = 128) MFMAs of one K16 half on `cur`, with the PB + CB fragment reads of `nxt` (half h_next of the k-step
  // whose addresses are in la / lbase) after the first rows -- the last row keeps 4 MFMAs of slack before `nxt` is used.
  // FIRST (half 0 of a k-step): the next k-step's addresses are prepared between the rows.
  auto half_step = [&](const Half& cur, Half& nxt, int h_next, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
#pragma unroll
      for (int jj = 0; jj < CB; ++jj)
        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.b[jj]), __builtin_bit_cast(bf16x8, cur.a[i]),
                                                             acc[i][jj], 0, 0, 0);
      if (i == 0) {
        rd_a(nxt, h_next, 0); rd_b(nxt, h_next, 0); rd_a(nxt, h_next, 1);
      } else if (i == 1) {
        if (CB > 1) rd_b(nxt, h_next, 1);
        rd_a(nxt, h_next, 2);
        if (CB > 2) rd_b(nxt, h_next, 2);
      } else if (i == 2) {
        rd_a(nxt, h_next, 3);
        if (CB > 3) rd_b(nxt, h_next, 3);
      }
      if constexpr (FIRST) prep_piece(i);
      dma_gap((FIRST ? 0 : 4) + i);
      __builtin_amdgcn_sched_barrier(0);
    }
    // `nxt` is complete (its last read was issued 4 MFMAs = 128 cycles ago): the next half never waits in mid-stream
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  const unsigned long long st0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long sr0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0ull;
  // prologue: halo 0, weights(0 .. STAGES-2); then half 0 of k-step 0
  for (int q = 0; q < W4_HL; ++q) issue_halo_part(0, q);
  for (int s2 = 0; s2 < STAGES - 1; ++s2) issue_b(s2);
  wait_vm16<(STAGES - 3) * LB>();                       // halo 0, weights(0), weights(1) have landed (my share)
  __builtin_amdgcn_s_barrier();
  Half r0, r1;
  prep_scalar();                                        // addresses of k-step 0
#pragma unroll
  for (int i = 0; i < PB; ++i) prep_piece(i);
  prep_commit();
#pragma unroll
  for (int i = 0; i < PB; ++i) rd_a(r0, 0, i);
#pragma unroll
  for (int jj = 0; jj < CB; ++jj) rd_b(r0, 0, jj);
#pragma unroll 1
  for (int j = 0; j < nk; ++j) {
    // k-step j.  The weight ring keeps being refilled past the last k-step (from the zero page, into stages nobody reads),
    // and the reads of "k-step nk" run on valid LDS addresses with unused values: the bookkeeping is the same every k-step.
    const unsigned long long q0 = now();
    prep_scalar();                                      // k-step j+1: scalar state now, vector addresses between the MFMA rows
    half_step(r0, r1, 1, std::true_type{});             // MFMAs of half 0; R1 <- half 1 of k-step j; weights(j-1+STAGES)
    const unsigned long long q1 = now();
    prep_commit();
    const unsigned long long q2 = now();
    half_step(r1, r0, 0, std::false_type{});            // MFMAs of half 1; R0 <- half 0 of k-step j+1
    const unsigned long long q3 = now();
    if (++ld_stage_issue == STAGES) ld_stage_issue = 0;
    if (tap == 1 && slice + 1 < nslices) {              // the next slice's halo (workgroup-uniform branch, outside the MFMA stream)
      for (int q = 0; q < W4_HL; ++q) issue_halo_part(slice + 1, q);
    }
    const unsigned long long q4 = now();
    // my share of weights(j+2) [issued in k-step j+3-STAGES] must have landed before the barrier: issued after it are
    // (STAGES-3) weight stages and, while 1 <= tap <= STAGES-2, the next slice's halo
    const bool halo_young = tap >= 1 && tap <= STAGES - 2 && slice + 1 < nslices;
    if (halo_young) wait_vm16<(STAGES - 3) * LB + W4_HL>();
    else wait_vm16<(STAGES - 3) * LB>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long q5 = now();
    __builtin_amdgcn_s_barrier();
    if (DIAG) {
      const unsigned long long q6 = now();
      dg[0] += q1 - q0; dg[1] += q2 - q1; dg[2] += q3 - q2; dg[3] += q4 - q3; dg[4] += q5 - q4; dg[5] += q6 - q5;
    }
    if (++tap == ntaps) { tap = 0; ++slice; }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the ring refills issued past the last k-step
  unsigned long long* drec = nullptr;
  if (DIAG && dbg != nullptr && lane == 0) {
    drec = dbg + ((size_t)blockIdx.x * 8 + wave) * 10;
    drec[0] = __builtin_amdgcn_s_memtime() - st0;
    drec[1] = __builtin_amdgcn_s_memrealtime() - sr0;
    drec[2] = (unsigned long long)nk;
    for (int q = 0; q < 6; ++q) drec[4 + q] = dg[q];
  }

  // ---- epilogue: D row = channel (e&3) + 8(e>>2) + 4lh of the 32-block, col = pixel lr; stage the tile through LDS as
  //      bf16 [pixel][BN (+8 pad)], write back with 16-byte stores ----
  __syncthreads();
  constexpr int CROW = BN * 2 + 16;
  unsigned char* ctile = smem;
  const float slope = act_slope(act);
#pragma unroll
  for (int j = 0; j < CB; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col0 = j * 32 + 8 * q + 4 * lh;
      float bq[4];
      uint32_t m01, m23;
      epi_col_consts(bias, n0 + col0, wrows, bq, m01, m23);
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int row = (wave * 4 + i) * 32 + lr;
        *reinterpret_cast<u32x2*>(ctile + row * CROW + col0 * 2) =
            epi_finish4(acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3], bq, slope, m01, m23);
      }
    }
  __syncthreads();
  constexpr int CPR = BN / 8;
  constexpr int RPP = 256 / CPR;
  const int chunk = tid % CPR, rsub = tid / CPR;
  const int ncol = n0 + chunk * 8;
  const int rec0 = img * (2 * tiles_y * tiles_x) + (2 * (trem / tiles_x)) * tiles_x + (trem % tiles_x);
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {                 // one statistics record per 8 x 32 half tile (see the 8-wave kernel)
    float st8[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) st8[k] = 0.f;
    if (ncol < ldc) {
#pragma unroll 4
      for (int p = 0; p < 256 / RPP; ++p) {
        const int row = half * 256 + p * RPP + rsub;
        const size_t opix = (size_t)out_pixel(g, img, y0 + (row >> 5), x0 + (row & 31));
        const u32x4 v = *reinterpret_cast<const u32x4*>(ctile + row * CROW + chunk * 16);
        *reinterpret_cast<u32x4*>(out + opix * ldc + ncol) = v;
        if (stats != nullptr) {
          float fv[8];
          Elem<bf16_t>::unpack(v, fv);
#pragma unroll
          for (int e = 0; e < 8; ++e) { st8[e] += fv[e]; st8[8 + e] = fmaf(fv[e], fv[e], st8[8 + e]); }
        }
      }
    }
    if (stats != nullptr) {
      float* red = reinterpret_cast<float*>(smem + 512 * CROW);
      float* mine = red + ((size_t)rsub * CPR + chunk) * 16;
      __syncthreads();                                   // the previous half's sums have been read
#pragma unroll
      for (int k = 0; k < 16; k += 4) {
        f32x4 t;
        t.x = st8[k]; t.y = st8[k + 1]; t.z = st8[k + 2]; t.w = st8[k + 3];
        *reinterpret_cast<f32x4*>(mine + k) = t;
      }
      __syncthreads();
      for (int o = tid; o < CPR * 16; o += 256) {
        const int ch = o >> 4, k = o & 15;
        float sum = 0.f;
        for (int r = 0; r < RPP; ++r) sum += red[((size_t)r * CPR + ch) * 16 + k];
        const int c = n0 + ch * 8 + (k & 7);
        if (c < ldc) stats[((size_t)(rec0 + half * tiles_x) * 2 + (k >> 3)) * ldc + c] = sum;
      }
    }
  }
  if (DIAG && drec != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    drec[3] = __builtin_amdgcn_s_memtime() - kt0;
  }
}

template <int BN, int STAGES>
static hipError_t launch_halo16w4(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out,
                                  int ldc, int act, hipStream_t st, float* stats) {
  const int tiles_m = g.N * (g.Ho / H16_TH) * (g.Wo / H16_TW);
  const int tiles_n = (ldc + BN - 1) / BN;
  constexpr size_t loop_lds = 2 * (size_t)H16_HBYTES + (size_t)STAGES * BN * 64 + H16_GROUPS * 16 * sizeof(int);
  constexpr size_t epi_lds = 512 * (size_t)(BN * 2 + 16) + 256 * 16 * sizeof(float);
  const size_t lds = std::max(loop_lds, epi_lds);
  const bool diag = g_v2_ablate == 6 && g_v2_dbg != nullptr;
  auto kern = diag ? halo16w4_conv_kernel<BN, STAGES, true> : halo16w4_conv_kernel<BN, STAGES, false>;
  static bool attr_done[2] = {false, false};
  if (!attr_done[diag]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done[diag] = true;
  }
  count_launch(K_HALO16_CONV);
  prof_begin(PROF_HALO_CONV, 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)wrows, st);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), lds, st, g, (const bf16_t*)src, (const bf16_t*)wgt, wrows, bias,
                     (bf16_t*)out, ldc, act, tiles_n, stats, g_v2_dbg);
  prof_end(PROF_HALO_CONV, st);
  return hipGetLastError();
}

// The four corners of the padded frame of a reflect-padded 3x3 conv's input gradient: frame pixel (-1,-1) is the image of
// input pixel (1,1) and only output (0,0) reads it, through kernel tap (0,0): dx[1][1] += W[:,:,0,0]^T dy[0][0]; likewise
// (-1,W) -> dx[1][W-2] += W[..,0,2]^T dy[0][W-1], (H,-1) -> dx[H-2][1] += W[..,2,0]^T dy[H-1][0], (H,W) -> dx[H-2][W-2] +=
// W[..,2,2]^T dy[H-1][W-1].  wd: the dgrad-packed weights [Cin][9][CoutS].  grid (4, N); runs after the FOLD launch.
__global__ __launch_bounds__(256) void reflect_corner_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ wd,
                                                             bf16_t* __restrict__ dx, int H, int W, int CoutS, int Cin, int CinS) {
  extern __shared__ float dyv[];
  const int corner = blockIdx.x, n = blockIdx.y;
  const int ky = corner >> 1 ? 2 : 0, kx = corner & 1 ? 2 : 0;
  const int sy = ky ? H - 1 : 0, sx = kx ? W - 1 : 0;             // dy pixel read
  const int ty = ky ? H - 2 : 1, tx = kx ? W - 2 : 1;             // dx pixel written
  const bf16_t* src = dy + ((size_t)(n * H + sy) * W + sx) * CoutS;
  for (int c = threadIdx.x; c < CoutS; c += 256) dyv[c] = bf16_to_f32(src[c]);
  __syncthreads();
  bf16_t* dst = dx + ((size_t)(n * H + ty) * W + tx) * CinS;
  for (int ci = threadIdx.x; ci < Cin; ci += 256) {
    const bf16_t* wrow = wd + ((size_t)ci * 9 + ky * 3 + kx) * CoutS;
    float s = 0.f;
    for (int c = 0; c < CoutS; c += 8) {
      float w8[8];
      Elem<bf16_t>::unpack(*reinterpret_cast<const u32x4*>(wrow + c), w8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(w8[e], dyv[c + e], s);
    }
    dst[ci] = f32_to_bf16(bf16_to_f32(dst[ci]) + s);
  }
}

extern int g_halo_bn, g_halo_stages;
int g_halo16 = 1;
int g_halo16_fold = 1;         // A/B option "halo16_fold": 0 = ring GEMM + finalize + border fold as separate launches              // A/B option "halo16": 0 = always the 8 x 32 tile kernel (conv_halo.hip)
int g_halo16_stages = 8;

template <int BN, int STAGES>
static hipError_t launch_halo16(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out,
                                int ldc, int act, hipStream_t st, float* stats, const void* ring = nullptr, bool fold = false) {
  const int tiles_m = g.N * (g.Ho / H16_TH) * (g.Wo / H16_TW);
  const int tiles_n = (ldc + BN - 1) / BN;
  constexpr size_t loop_lds = 2 * (size_t)H16_HBYTES + (size_t)STAGES * BN * 64 + H16_GROUPS * 16 * sizeof(int);
  constexpr size_t epi_lds = 256 * (size_t)(BN * 2 + 16) + 512 * 16 * sizeof(float);
  const size_t lds = std::max(loop_lds, epi_lds);
  const bool diag = g_v2_ablate == 6 && g_v2_dbg != nullptr && STAGES == 8 && !fold;
  auto kern = diag ? halo16_conv_kernel<BN, STAGES, STAGES == 8> : halo16_conv_kernel<BN, STAGES, false>;
  if (fold) kern = halo16_conv_kernel<BN, STAGES, false, STAGES == 8>;        // (instantiated for the shipped ring depth only)
  if (fold && STAGES != 8) return hipErrorNotSupported;
  static bool attr_done[3] = {false, false, false};
  const int which = fold ? 2 : (diag ? 1 : 0);
  if (!attr_done[which]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done[which] = true;
  }
  count_launch(K_HALO16_CONV);
  prof_begin(PROF_HALO_CONV, 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)wrows, st);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, st, g, (const bf16_t*)src, (const bf16_t*)wgt, wrows, bias,
                     (bf16_t*)out, ldc, act, tiles_n, stats, g_v2_dbg, (const bf16_t*)ring,
                     ring != nullptr ? ring_pixels(g.Hl, g.Wl) : 0);
  prof_end(PROF_HALO_CONV, st);
  return hipGetLastError();
}

// returns hipErrorNotSupported when the shape does not qualify (the caller goes on to the 8 x 32 tile kernel)
// ring: see the kernel (SPADE -> upsample -> conv with z kept at the source resolution); only the 8-wave kernel takes it
// fold: the launch is the interior input gradient of a reflect-padded 3x3 conv and also folds the frame's ring (not its
// corners: reflect_corners) into the border rows / columns -- see the kernel
hipError_t halo16_conv(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out, int ldc,
                       int act, int num_cu, hipStream_t st, float* stats, const void* ring, bool fold) {
  if (fold && (!g_halo16_fold || g.ys >= 0 || g.xs >= 0 || g.pad_mode != PAD_ZERO || g.up || g.Ho < 2 * H16_TH || g.Wo < 2 * H16_TW ||
               bias != nullptr || act != ACT_NONE || stats != nullptr || ring != nullptr || g_halo16 == 2))
    return hipErrorNotSupported;
  if (!g_halo16 || g_halo_bn != 0 || g_halo_stages != 0) return hipErrorNotSupported;   // (the tile sweep is the 8 x 32 kernel's)
  if (g.sh != 1 || g.sw != 1 || (g.ys != 1 && g.ys != -1) || (g.xs != 1 && g.xs != -1)) return hipErrorNotSupported;
  if (g.th != 3 || g.tw != 3 || g.wK != g.K || g.wtw != g.tw) return hipErrorNotSupported;
  if (g.Cs % 32 != 0 || g.Ho % H16_TH != 0 || g.Wo % H16_TW != 0 || g.M != g.N * g.Ho * g.Wo) return hipErrorNotSupported;
  if ((long long)g.N * g.Hs * g.Ws * g.Cs >= (1ll << 31)) return hipErrorNotSupported;   // 32-bit offset table
  if (ldc < 64 || ldc % 8 != 0) return hipErrorNotSupported;
  const int tiles_m = g.N * (g.Ho / H16_TH) * (g.Wo / H16_TW);
  const int tn = ldc >= 128 ? (ldc + 127) / 128 : 1;
  // the double-size tile needs a grid that still covers the chip: at least ~7/8 of a round (fewer: the 8 x 32 tiles spread better)
  if (tiles_m * tn < (num_cu * 7) / 8) return hipErrorNotSupported;
  if (ring != nullptr && (g.Hl < 4 || g.Wl < 4)) return hipErrorNotSupported;
  if (g_halo16 == 2 && ring == nullptr) {          // the four-wave software-pipelined variant
    if (ldc >= 128) return launch_halo16w4<128, 8>(g, src, wgt, wrows, bias, out, ldc, act, st, stats);
    return launch_halo16w4<64, 8>(g, src, wgt, wrows, bias, out, ldc, act, st, stats);
  }
  if (fold && g_halo16_stages != 8) return hipErrorNotSupported;
  if (ldc >= 128) {
    if (g_halo16_stages == 4) return launch_halo16<128, 4>(g, src, wgt, wrows, bias, out, ldc, act, st, stats);
    if (g_halo16_stages == 6) return launch_halo16<128, 6>(g, src, wgt, wrows, bias, out, ldc, act, st, stats);
    return launch_halo16<128, 8>(g, src, wgt, wrows, bias, out, ldc, act, st, stats, ring, fold);
  }
  if (g_halo16_stages == 4) return launch_halo16<64, 4>(g, src, wgt, wrows, bias, out, ldc, act, st, stats);
  return launch_halo16<64, 8>(g, src, wgt, wrows, bias, out, ldc, act, st, stats, ring, fold);
}

hipError_t reflect_corners(const void* dy, const void* wd_packed, void* dx, int N, int H, int W, int CoutS, int Cin, int CinS,
                           hipStream_t st) {
  hipLaunchKernelGGL(reflect_corner_kernel, dim3(4, N), dim3(256), (size_t)CoutS * sizeof(float), st, (const bf16_t*)dy,
                     (const bf16_t*)wd_packed, (bf16_t*)dx, H, W, CoutS, Cin, CinS);
  return hipGetLastError();
}

}  // namespace dei2i
