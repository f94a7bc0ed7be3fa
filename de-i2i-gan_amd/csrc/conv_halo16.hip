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
template <int N> DEI2I_D void lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// LDS reads the compiler does not track (no automatic s_waitcnt: every use needs a hand-placed lgkm_wait)
template <int OFF, bool SKIP = false> DEI2I_D u32x4 lds_read128_asm(int addr) {
  u32x4 v;
  if constexpr (SKIP) asm volatile("; no read %0 %1" : "=v"(v) : "v"(addr));        // (timing-only ablation)
  else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
DEI2I_D int lds_read32_asm(int addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
DEI2I_D int sw16(int row) { return ((row >> 2) & 1) << 1; }

typedef float f32x2 __attribute__((ext_vector_type(2)));
DEI2I_D f32x2 bf16x2_unpack(uint32_t u) { return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }

// Column sums of the epilogues: thread tid of the 512 holds NV partial values for the 8-channel chunk (tid % CPR), taken over the
// rows it wrote back; the sum over the 512 / CPR threads of a chunk goes to out(q, channel) with q = k / 8, channel = 8 * chunk +
// k % 8.  Lanes of a wave that share a chunk are CPR apart: butterfly over those lane bits (ds_bpermute), then the 8 waves'
// sums through `scratch` (8 * CPR * NV floats) in wave order -- ONE barrier, ~NV / 4 LDS writes and 8 reads per thread (the rounds
// of 32-row column sums this replaces: 4 NV-float writes + 32 reads per output and two barriers per 16 values).  Fixed order:
// the result does not depend on timing.  `store(q, c, sum)` writes one output; the caller syncs before `scratch` is reused.
template <int NV, int CPR, typename Store>
DEI2I_D void reduce_rows16(float (&v)[NV], float* __restrict__ scratch, int tid, Store store) {
  static_assert(NV % 8 == 0 && (CPR == 16 || CPR == 8), "layout");
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float t = v[k];
    if constexpr (CPR == 8) t += __shfl_xor(t, 8, 64);
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    v[k] = t;
  }
  if (lane < CPR) {
    float* mine = scratch + ((size_t)wave * CPR + lane) * NV;
#pragma unroll
    for (int k = 0; k < NV; k += 4) *reinterpret_cast<f32x4*>(mine + k) = f32x4{v[k], v[k + 1], v[k + 2], v[k + 3]};
  }
  __syncthreads();
  for (int o = tid; o < CPR * NV; o += 512) {
    const int q = o / (CPR * 8), c = o % (CPR * 8);
    const float* col = scratch + (size_t)(c >> 3) * NV + q * 8 + (c & 7);
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += col[(size_t)w * CPR * NV];
    store(q, c, sum);
  }
}

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
// the LDS-staged tile before it is written; the four frame CORNERS (one pixel each) are dot products in the epilogue.
// DIAG 2 / 3 / 4: stamps + a TIMING-ONLY ablation (results are wrong): 2 = no LDS-DMA inside the loop, 3 = no fragment reads,
// 4 = no MFMAs (options v2_ablate = 7 / 8 / 9)
// ABL (options v2_ablate = 10 + ABL, no stamps): the same as a bit mask -- 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no MFMAs
// PIPE: the software-pipelined main loop (see there) instead of the two-phase one
// EPIN: the epilogue also takes the backward reductions of the normalisation layer in front of this conv (geom.h: EpiNorm) from
// the finished dz tile -- FOLD launches only (every conv of the generator is reflect-padded)
// S2: a 4x4 STRIDE-2 conv (generator.py:107-116, discriminator.py:60-77) on the same tile.  Output pixel (oy, ox), tap (ky, kx)
// reads input (2 oy + ky - pad, 2 ox + kx - pad): with ky = 2 ty + dy, kx = 2 tx + dx that is pixel (oy + ty, ox + tx) of the
// parity plane (dy, dx) of the padded input -- a 2x2 STRIDE-1 conv over 4 planes.  The planes exist only in LDS: a slice is
// (plane, 32 channels), its halo is 17 x 33 plane pixels gathered by the LDS-DMA through a per-plane source-offset table (reflect
// padding is in the table), its 4 taps read tap-shifted fragments exactly like the 3x3 kernel -- so the input is fetched ONCE per
// tile and channel tile (the gather GEMM streams every input pixel four times, once per tap that reads it: 48 KB per k-step
// against 8 KB of weights + 9 KB of halo here).  HBM layout and the packed weights ([Cout][4x4][CinS]) are the generic ones.
template <int BN, int STAGES, int DIAG = 0, bool FOLD = false, int ABL = 0, bool PIPE = false, bool EPIN = false, bool S2 = false>
__global__ __launch_bounds__(512) void halo16_conv_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                          const bf16_t* __restrict__ wgt, const int wrows,
                                                          const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                          const int ldc, const int act, const int tiles_n,
                                                          float* __restrict__ stats, unsigned long long* __restrict__ dbg,
                                                          const bf16_t* __restrict__ zring, const int ring_pix,
                                                          const EpiNorm en) {
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

  static_assert(!S2 || (!FOLD && !PIPE && !EPIN && DIAG == 0 && ABL == 0 && STAGES == 4), "the stride-2 form runs the two-phase loop on a 4-stage ring");
  const int lth = S2 ? 2 : g.th, ltw = S2 ? 2 : g.tw;  // taps of the loop (S2: per parity plane)
  const int hwd = H16_TW + ltw - 1;                   // halo width in pixels
  const int npix = (H16_TH + lth - 1) * hwd;
  const int hy0 = S2 ? y0 : y0 + g.by0 + (g.ys < 0 ? -(g.th - 1) : 0);
  const int hx0 = S2 ? x0 : x0 + g.bx0 + (g.xs < 0 ? -(g.tw - 1) : 0);
  constexpr int HTAB = H16_GROUPS * 16;               // entries of one source-offset table

  // ---- halo source-offset table (element offset of each halo pixel's channel 0, -1 = contributes zero); S2: one per parity plane ----
  for (int p = tid; p < (S2 ? 4 : 1) * HTAB; p += 512) {
    int off = -1;
    const int par = S2 ? p / HTAB : 0, q = S2 ? p - par * HTAB : p;
    if (q < npix) {
      const int hy = q / hwd, hx = q - hy * hwd;
      const int y = bound_coord(S2 ? 2 * (hy0 + hy) + (par >> 1) + g.by0 : hy0 + hy, g.Hl, g.pad_mode);
      const int x = bound_coord(S2 ? 2 * (hx0 + hx) + (par & 1) + g.bx0 : hx0 + hx, g.Wl, g.pad_mode);
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

  const int ntaps = __builtin_amdgcn_readfirstlane(lth * ltw);
  const int ncb = __builtin_amdgcn_readfirstlane(g.Cs >> 5);                    // 32-channel blocks of the input
  const int nslices = S2 ? 4 * ncb : ncb;              // S2: slice = (parity plane, channel block), plane-major
  const int nk = ntaps * nslices;

  auto issue_halo = [&](int slice) {
    unsigned char* hb = halo + (slice & 1) * H16_HBYTES;
    if constexpr (S2) {
      const int par = slice / ncb, ci0 = (slice - par * ncb) << 5;
#pragma unroll
      for (int j = 0; j < H16_HL; ++j) {
        const int pix = h_group[j] * 16 + lrow;
        const int o = htab[par * HTAB + pix];
        const bf16_t* p = o >= 0 ? src + ((size_t)(unsigned)(o + ((lslot ^ sw16(pix)) << 3)) + (unsigned)ci0) : zero;
        glds16x(p, hb + h_group[j] * 1024);
      }
    } else {
      const int ci0 = slice << 5;
#pragma unroll
      for (int j = 0; j < H16_HL; ++j) {
        const bf16_t* p = h_off[j] >= 0 ? src + ((size_t)(unsigned)h_off[j] + (unsigned)ci0) : zero;
        if (h_off[j] < -1) p = zring + ((size_t)(unsigned)(-2 - h_off[j]) + (unsigned)ci0);
        glds16x(p, hb + h_group[j] * 1024);
      }
    }
  };
  int is_tap = 0, is_slice = 0;                        // (tap, slice) of the next weight k-step to issue
  auto issue_b = [&](int stage) {
    int kb;
    if constexpr (S2) {                                // plane (dy, dx), loop tap (ty, tx) -> kernel tap (2 ty + dy, 2 tx + dx) of the 4x4 grid
      const int par = is_slice / ncb, cb = is_slice - par * ncb;
      kb = ((2 * (is_tap >> 1) + (par >> 1)) * 4 + 2 * (is_tap & 1) + (par & 1)) * g.Cs + (cb << 5);
    } else {
      kb = is_tap * g.Cs + (is_slice << 5);
    }
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
  // byte address (tap offset 0, before the swizzle) of this lane's chunk of its pixel in tile row 4 * wm + r, columns 0..15;
  // columns 16..31 sit 1024 bytes further in the same slot (16 rows on: the swizzle bit (row >> 2) & 1 repeats)
  int a_base[PB / 2];
#pragma unroll
  for (int r = 0; r < PB / 2; ++r) a_base[r] = ((wm * 4 + r) * hwd + l16) * 64 + kg * 16;
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
  int ld_toff = (g.ys > 0 ? 0 : (lth - 1) * hwd) + (g.xs > 0 ? 0 : ltw - 1);       // halo pixel offset of tap (0,0)
  const int toff_row_wrap = step_y - (ltw - 1) * step_x;
  const int toff_origin = ld_toff;
  auto read_frags = [&](Frags& f) {
    const unsigned char* hb = halo + (ld_slice & 1) * H16_HBYTES;
    const unsigned char* sb = ring + ld_stage * B_STAGE;
    const int s_off = ld_toff * 64 + (ld_slice & 1) * H16_HBYTES;     // wave-uniform; H16_HBYTES is a multiple of 1024
#pragma unroll
    for (int r = 0; r < PB / 2; ++r) {
      const int t = a_base[r] + s_off;
      const int addr = t ^ ((t >> 3) & 32);             // slot = chunk ^ (((row >> 2) & 1) << 1): row bit 2 = address bit 8
      f.a[2 * r] = *reinterpret_cast<const u32x4*>(halo + addr);
      f.a[2 * r + 1] = *reinterpret_cast<const u32x4*>(halo + addr + 1024);
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
    if (++ld_tx == ltw) {
      ld_tx = 0;
      if (++ld_ty == lth) { ld_ty = 0; ++ld_slice; ld_toff = toff_origin; }
      else ld_toff += toff_row_wrap;
    } else {
      ld_toff += step_x;
    }
    if (++ld_stage == STAGES) ld_stage = 0;
  };

  unsigned long long* drec = nullptr;
  unsigned long long st0 = 0;
  if constexpr (PIPE) {
    // ---- main loop, software-pipelined (option "halo16" = 1): every wave runs MFMAs all the time -----------------------------
    // The anti-phase loop below gives a wave EITHER its 12 fragment reads OR its 32 MFMAs per phase; a DMA piece among the
    // MFMAs stalls the in-order wave ~100-300 cycles behind the other waves' pieces and the SIMD's matrix pipe idles with it
    // (its partner wave is in a read phase).  Timing-only ablations of that loop, res-block shape, kernel wall: full 71.7 us |
    // MFMA only 51.3 | reads only 40.9 | DMA only 35.8 | empty loop 18.0 -- the three streams run one after the other much more
    // than beside each other.  Here a k-step (one tap x 32 channels) is two halves of 16 MFMAs:
    //   H0(j): issue the reads of pixel rows 2,3 of k-step j                      | MFMAs of rows 0,1
    //   H1(j): issue the reads of rows 0,1 and of the weights of k-step j+1       | 4 MFMAs | DMA slot j | 12 MFMAs (rows 2,3)
    // so both waves of a SIMD always hold MFMAs for the pipe and one wave's read latency or DMA issue stall is the other's
    // issue time.  Waves 0-3 put the k-step's ONE workgroup barrier between H0 and H1, waves 4-7 after H1 (half a k-step
    // apart: the two waves of a SIMD do not meet the barrier, or their read bursts, at the same moment).
    //   DMA slot j = halo piece `tap` of the NEXT slice (taps 0..4: this wave's 5 pieces, into the buffer of the previous
    //        slice, last read in H0 of that slice's tap 8) then weights(j+7) into stage (j+7) % 8 (last read in H1(j-2)).
    //   barrier #j needs every wave's piece of weights(j+2) landed (waves 4-7 read stage j+2 in H1(j+1) with no barrier in
    //        between) and, at tap 7, all of the next slice's halo (read from H1(tap 8) on).  weights(j+2) went out in slot j-5;
    //        younger at barrier #j: for waves 0-3 (slot j not issued yet) the weights of slots j-4..j-1 and the halo pieces
    //        of the taps tap-4..tap-1 that lie in 0..4; for waves 4-7 those of slots j-4..j / taps tap-4..tap.
    static_assert(!PIPE || (STAGES == 8 && BN == 128), "the read schedule and wait counts below are written for 8 stages and 4 channel blocks");
    constexpr int HWD = H16_TW + 2;
    const int grp = wave >> 2;
    const bool flip_y = g.ys < 0, flip_x = g.xs < 0;
    // The fragment reads are inline asm with HAND-COUNTED lgkmcnt waits: the compiler's own wait insertion falls back to
    // lgkmcnt(0) at every branch merge of this body, i.e. it would wait for the reads it has just issued.  LDS operations
    // return in issue order; per k-step a wave issues
    //   H0: ah x4                                  H1: [piece offset (taps 0..4)] al' x4 [ring' (FOLD)] | bc'[0] | bc'[1] | bc'[2] | bc'[3]
    // (the ring and piece-offset reads are issued whether or not they are used, so that the counts are static).
    u32x4 al[4], ah[4], bc[CB];
    auto tap_off = [&](int t) -> int {                  // byte offset of tap t's halo pixel (ty, tx of the LOOP order)
      const int ty = t / 3, tx = t % 3;
      return ((flip_y ? 2 - ty : ty) * HWD + (flip_x ? 2 - tx : tx)) * 64;
    };
    auto read_a = [&](u32x4* dst, int r0, int soff) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int t = a_base[r0 + r] + soff;
        const int addr = t ^ ((t >> 3) & 32);
        dst[2 * r] = lds_read128_asm<0, (ABL & 2) != 0>(addr);
        dst[2 * r + 1] = lds_read128_asm<1024, (ABL & 2) != 0>(addr);
      }
    };
    const int b_addr0 = b_addr[0] + 2 * H16_HBYTES;     // (block jj sits 16 rows = 1024 bytes after block jj-1, same slot)
    auto read_b1 = [&](int jj, int stage) -> u32x4 {
      const int a = b_addr0 + stage * B_STAGE;
      return jj == 0 ? lds_read128_asm<0, (ABL & 2) != 0>(a) : jj == 1 ? lds_read128_asm<1024, (ABL & 2) != 0>(a) : jj == 2 ? lds_read128_asm<2048, (ABL & 2) != 0>(a) : lds_read128_asm<3072, (ABL & 2) != 0>(a);
    };
    auto read_ring = [&](int t, int par) {              // FOLD: the ring block's fragment of loop tap t (read whether or not it is live)
      if constexpr (FOLD) {
        const int ty = t / 3, tx = t % 3;
        ring_live = (ring_kind == 1 && ty == 0) || (ring_kind == 2 && ty == 2) || (ring_kind == 3 && tx == 0) || (ring_kind == 4 && tx == 2);
        const int pix = ring_live ? ring_pix0 + (tap_off(t) >> 6) : 0;
        ring_frag = lds_read128_asm<0, (ABL & 2) != 0>(par + pix * 64 + ((kg ^ sw16(pix)) << 4));
      }
    };
    // (the opaque asm statements below keep per-tap copies of loop-invariant addresses -- 9 weight pointers, 5 table
    //  addresses -- from being hoisted into registers the loop does not have: they were spilled, and a scratch reload is a
    //  vector-memory operation with an s_waitcnt vmcnt(0) behind it, i.e. a drain of the whole LDS-DMA queue)
    const bf16_t* b_src = b_ptr != nullptr ? b_ptr : zero;
    const int b_live = b_ptr != nullptr ? 1 : 0;
    auto issue_w = [&](int tapw, int slicew, int stage) {
      const bf16_t* bp = b_src;
      asm volatile("" : "+v"(bp));
      const int kb = (tapw * g.Cs + (slicew << 5)) * b_live;
      glds16x(bp + kb, ring + stage * B_STAGE + rg * 1024);
    };
    // a wave's halo piece jp = 16-pixel group 8 * jp + wave (mod 39): the pixel's source offset is read back from the offset
    // table when the piece is issued (five per-lane offsets -- or the 64-bit pointers the compiler makes of them -- would
    // not fit beside the accumulators and two fragment sets)
    const int hl_addr = (int)(reinterpret_cast<unsigned char*>(htab) - smem) + lrow * 4;
    const int so_lane = (lslot ^ sw16(lrow)) << 3;      // (group bases are multiples of 16 pixels: the swizzle bit is the lane's)
    auto piece_group = [&](int jp) -> int {
      int gq = jp * 8 + wave;
      if (gq >= H16_GROUPS) gq -= H16_GROUPS;
      return gq;
    };
    auto read_piece_off = [&](int jp) -> int {
      int a = hl_addr;
      asm volatile("" : "+v"(a));
      return lds_read32_asm(a + piece_group(jp) * 64);
    };
    auto issue_halo_piece = [&](int slicew, int jp, int o) {
      const int ci0 = slicew << 5;
      const bf16_t* p = o >= 0 ? src + ((size_t)(unsigned)(o + so_lane) + (unsigned)ci0) : zero;
      if (o < -1) p = zring + ((size_t)(unsigned)(-2 - o + so_lane) + (unsigned)ci0);
      glds16x(p, halo + (slicew & 1) * H16_HBYTES + piece_group(jp) * 1024);
    };
    // 4 MFMAs: pixel blocks a[0..3] (accumulators i0..i0+3) x channel block jj
    auto mfma4 = [&](const u32x4* a, int i0, int jj) {
      if constexpr (!(ABL & 4)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i0 + i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[jj]), __builtin_bit_cast(bf16x8, a[i]),
                                                                    acc[i0 + i][jj], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(a[i]));
        asm volatile("" ::"v"(bc[jj]));
      }
    };
    constexpr int NF = FOLD ? 1 : 0;

    // (cycle count of the loop for every build of this path, when a debug buffer is set: tools/diag_halo16_stamps.py)
    st0 = dbg != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long sr0p = dbg != nullptr ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // prologue: halo slice 0 and weights(0..6); halo 0, weights(0) and weights(1) landed (waves 4-7 read stage 1 in H1(0))
    {
      int po[H16_HL];
#pragma unroll
      for (int jp = 0; jp < H16_HL; ++jp) po[jp] = read_piece_off(jp);
      lgkm_wait<0>();
#pragma unroll
      for (int jp = 0; jp < H16_HL; ++jp) issue_halo_piece(0, jp, po[jp]);
    }
#pragma unroll
    for (int s2 = 0; s2 < STAGES - 1; ++s2) issue_w(s2, 0, s2);
    wait_vm16<STAGES - 3>();
    __builtin_amdgcn_s_barrier();
    read_a(al, 0, tap_off(0));
    read_ring(0, 0);
#pragma unroll
    for (int jj = 0; jj < CB; ++jj) bc[jj] = read_b1(jj, 0);
    __builtin_amdgcn_sched_barrier(0);

    // one fragment read of rows r0.. : row r (0/1 of the pair), column half c
    auto read_a1 = [&](u32x4* dst, int r0, int q, int soff) {      // q = 2 * r + c
      const int t = a_base[r0 + (q >> 1)] + soff;
      const int addr = t ^ ((t >> 3) & 32);
      dst[q] = (q & 1) ? lds_read128_asm<1024, (ABL & 2) != 0>(addr) : lds_read128_asm<0, (ABL & 2) != 0>(addr);
    };
    auto kstep = [&](auto tap_c, int sl) {
      constexpr int T = decltype(tap_c)::value;
      const int j = sl * 9 + T;
      const int par = (sl & 1) * H16_HBYTES;
      const bool has_halo = sl + 1 < nslices;
      const int soff = tap_off(T) + par;
      // The reads are spread between the MFMA groups (4 MFMAs each), at most three in a row: eight waves that each issue 5-6
      // reads at once right after a barrier fill the LDS queue, every wave stalls at its next read and the MFMAs behind it
      // wait (reads issued up front: MFMAs and reads took the SUM of their times).  Issue order per k-step:
      //   H0: ah0 G0 ah1 G1 ah2 G2 ah3 G3      H1: [P] al'0 al'1 G0 bc'0 al'2 (DMA) G1 bc'1 al'3 G2 bc'2 [ring'] G3 bc'3
      // ---- H0(j): rows 0,1 multiply (groups by channel block jj), rows 2,3 of this k-step are fetched
      read_a1(ah, 2, 0, soff);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(ABL & 16)) lgkm_wait<2>();          // all of al', ring' and bc'[0..2] are in; bc'[3] and ah0 may be in flight
#pragma unroll
      for (int jj = 0; jj < CB; ++jj) {
        if (jj == CB - 1) {
          if constexpr (!(ABL & 16)) lgkm_wait<4>();      // bc'[3] is in (ah0..3 may be in flight)
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma4(al, 0, jj);
        if constexpr (FOLD) {
          if (ring_live)
            racc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[jj]), __builtin_bit_cast(bf16x8, ring_frag),
                                                               racc[jj], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (jj < 3) read_a1(ah, 2, jj + 1, soff);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (grp == 0) {
        if constexpr (!(ABL & 1)) {
          constexpr int NH = T == 7 ? 3 : 4 + (T == 0 ? 0 : T == 1 ? 1 : T == 2 ? 2 : T == 3 ? 3 : T == 4 ? 4 : T == 5 ? 4 : T == 6 ? 3 : 1);
          if (j + STAGES - 2 >= nk) wait_vm16<0>();
          else if (has_halo) wait_vm16<NH>();
          else wait_vm16<4>();
        }
        // (tap 8: these were the last reads of this slice's halo buffer, which the DMA refills after the next barrier)
        if constexpr (T == 8 && !(ABL & 16)) lgkm_wait<0>();
        if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- H1(j): rows 2,3 multiply; rows 0,1 of k-step j+1 are fetched, and its weight block jj as soon as this k-step's
      //      MFMAs with block jj are out (after the last k-step these reads fetch bytes nobody uses)
      constexpr int TN = (T + 1) % 9;
      constexpr int NP = T <= 4 ? 1 : 0;
      const int par_n = T == 8 ? ((sl + 1) & 1) * H16_HBYTES : par;
      const int soff_n = tap_off(TN) + par_n;
      const int stage_n = (j + 1) & (STAGES - 1);
      int piece_off = -1;
      if constexpr (NP) piece_off = read_piece_off(T);
      read_a1(al, 0, 0, soff_n);
      read_a1(al, 0, 1, soff_n);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(ABL & 16)) lgkm_wait<NP + 2>();     // ah0..3 are in
#pragma unroll
      for (int jj = 0; jj < CB; ++jj) {
        __builtin_amdgcn_sched_barrier(0);
        mfma4(ah, 4, jj);
        __builtin_amdgcn_sched_barrier(0);
        bc[jj] = read_b1(jj, stage_n);
        if (jj == 0) read_a1(al, 0, 2, soff_n);
        if (jj == 1) read_a1(al, 0, 3, soff_n);
        if (jj == 2) read_ring(TN, par_n);
        __builtin_amdgcn_sched_barrier(0);
        if (jj == 0) {
          if constexpr (!(ABL & 1)) {
            if constexpr (NP) {
              if constexpr (!(ABL & 16)) lgkm_wait<4>();  // the piece offset (oldest of this half's reads) is in
              if (has_halo) issue_halo_piece(sl + 1, T, piece_off);
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr int TW = (T + STAGES - 1) % 9, DS = (T + STAGES - 1) / 9;
            if (sl + DS < nslices) issue_w(TW, sl + DS, (j + STAGES - 1) & (STAGES - 1));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (grp == 1) {
        if constexpr (!(ABL & 1)) {
          constexpr int NH = T == 7 ? 4 : 5 + (T == 0 ? 1 : T == 1 ? 2 : T == 2 ? 3 : T == 3 ? 4 : T == 4 ? 5 : T == 5 ? 4 : T == 6 ? 3 : 1);
          if (j + STAGES - 1 >= nk) wait_vm16<0>();
          else if (has_halo) wait_vm16<NH>();
          else wait_vm16<5>();
        }
        if constexpr (!(ABL & 8)) __builtin_amdgcn_s_barrier();
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int sl = 0; sl < nslices; ++sl) {
      // (opaque: without it the 9 taps' read addresses -- a_base + tap offset, 36 registers -- are hoisted out of the loop)
#pragma unroll
      for (int r = 0; r < PB / 2; ++r) asm volatile("" : "+v"(a_base[r]));
      kstep(std::integral_constant<int, 0>{}, sl);
      kstep(std::integral_constant<int, 1>{}, sl);
      kstep(std::integral_constant<int, 2>{}, sl);
      kstep(std::integral_constant<int, 3>{}, sl);
      kstep(std::integral_constant<int, 4>{}, sl);
      kstep(std::integral_constant<int, 5>{}, sl);
      kstep(std::integral_constant<int, 6>{}, sl);
      kstep(std::integral_constant<int, 7>{}, sl);
      kstep(std::integral_constant<int, 8>{}, sl);
      // (register copies at the loop's back-edge must not meet a fragment register whose read is still in flight)
      lgkm_wait<0>();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (dbg != nullptr && lane == 0) {
      unsigned long long* rec = dbg + ((size_t)blockIdx.x * 8 + wave) * 10;
      rec[0] = __builtin_amdgcn_s_memtime() - st0;
      rec[1] = __builtin_amdgcn_s_memrealtime() - sr0p;
      rec[2] = (unsigned long long)nk;
    }
  } else {
    // ---- main loop: two wave groups in anti-phase (see conv_halo.hip) ----
    //   M(j) reads its fragments, then issues weights(j-2+STAGES) into stage (j-2) % STAGES (last read by the trailing group's
    //        M(j-2), three phases earlier) and, at tap 2 of a slice, the NEXT slice's halo (into the buffer of the previous
    //        slice, last read at tap 8 of that slice), and ends with "my share of weights(j+1) has landed" + barrier.
    //        weights(j+1) was issued in M(j+3-STAGES); issued after it: the weights of M(j+4-STAGES .. j) = (STAGES-3) pieces,
    //        and the halo of M(tap 2) while 2 <= tap <= STAGES-1 -- exactly those may stay in flight.  The last STAGES-2
    //        k-steps drain everything.
    //   C(j) is the 32 MFMAs and nothing else.
    const int grp = wave >> 2;
    int tap = 0, slice = 0;                              // k-step of this wave's current M / C phase
    Frags f;
    if constexpr (DIAG == 3 || (ABL & 2)) {
      read_frags(f);
      ld_tx = 0; ld_ty = 0; ld_slice = 0; ld_stage = 0; ld_toff = toff_origin;
    }
    // DMA slot of k-step jj: weights(jj-2+STAGES) and, when k-step jj is tap 2 of its slice, the next slice's halo.  It sits
    // after the first 8 MFMAs of C(jj-1).  Measured alternatives (same box, res-block shape, kernel wall 69.4-70.7 us as is):
    // the slot in M(jj) after the fragment reads 74.5-75.9 us, before them 77.0-77.5, weights in M / halo in C 72.9-73.3, half
    // of a group's waves before and half after the reads 75.8-78.4 -- a piece costs the wave that issues it ~150-300 cycles
    // wherever it sits, and the M phase has less room for it than the C phase.
    auto slot_w = [&](int jj) {
      if (jj - 2 + STAGES < nk) issue_b((jj + STAGES - 2) % STAGES);
    };
    auto slot_h = [&](bool tap2) {
      if (tap2 && slice + 1 < nslices) issue_halo(slice + 1);
    };
    auto phase_m = [&](int j) {
      const unsigned long long q0 = now();
      if constexpr (DIAG != 3 && !(ABL & 2)) read_frags(f);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long q1 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
      // the next slice's halo went out in C(tap 1): while it is younger than weights(j+1) it may stay in flight -- but not at the
      // slice's last tap, whose barrier is the last one before M(next slice, tap 0) reads it (short tap grids: S2)
      const bool halo_young = tap >= 2 && tap <= STAGES - 1 && tap <= ntaps - 2 && slice + 1 < nslices;
      if constexpr (DIAG != 2 && !(ABL & 1)) {
        if (j + STAGES - 2 >= nk) wait_vm16<0>();
        else if (halo_young) wait_vm16<(STAGES - 3) * LB + H16_HL>();
        else wait_vm16<(STAGES - 3) * LB>();
      }
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
      if constexpr (DIAG != 4 && !(ABL & 4)) {
  #pragma unroll
        for (int i = 0; i < 2; ++i)
  #pragma unroll
          for (int jj = 0; jj < CB; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.b[jj]), __builtin_bit_cast(bf16x8, f.a[i]),
                                                                 acc[i][jj], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DIAG != 2 && !(ABL & 1)) {
        slot_w(j + 1);
        __builtin_amdgcn_sched_barrier(0);
        slot_h(tap == 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DIAG != 4 && !(ABL & 4)) {
  #pragma unroll
        for (int i = 2; i < PB; ++i)
  #pragma unroll
          for (int jj = 0; jj < CB; ++jj)
            acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.b[jj]), __builtin_bit_cast(bf16x8, f.a[i]),
                                                                 acc[i][jj], 0, 0, 0);
      } else {
        // keep the fragments live so that the reads are not removed
  #pragma unroll
        for (int i = 0; i < PB; ++i) asm volatile("" ::"v"(f.a[i]));
  #pragma unroll
        for (int jj = 0; jj < CB; ++jj) asm volatile("" ::"v"(f.b[jj]));
      }
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

    st0 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long sr0 = DIAG ? __builtin_amdgcn_s_memrealtime() : 0ull;
    // prologue: halo slice 0 and weights(0 .. STAGES-3); weights(STAGES-2) is issued by M(0).  nk >= 9 > STAGES-1.
    issue_halo(0);
    for (int s2 = 0; s2 < STAGES - 2; ++s2) issue_b(s2);
    issue_b(STAGES - 2);                                  // (the slot of k-step 0 would sit in "C(-1)")
    wait_vm16<(STAGES - 2) * LB>();                       // everything older than weights(1): halo 0 and weights(0)
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();           // group 1 starts one phase late
  #pragma unroll 1
    for (int j = 0; j < nk; ++j) {
      phase_m(j);
      phase_c(j);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();           // pairs with group 1's last phase
    if (DIAG && dbg != nullptr && lane == 0) {
      drec = dbg + ((size_t)blockIdx.x * 8 + wave) * 10;
      drec[0] = __builtin_amdgcn_s_memtime() - st0;
      drec[1] = __builtin_amdgcn_s_memrealtime() - sr0;
      drec[2] = (unsigned long long)nk;
      for (int q = 0; q < 6; ++q) drec[4 + q] = dg[q];
    }

  }

  // ---- epilogue: the whole 16 x 32 tile is staged as bf16 [pixel][BN (+8 pad)] (8-byte writes: a lane holds 4 consecutive
  //      channels of its pixel) -- the loop's buffers are dead, 512 rows fit --, FOLD launches add the reflect ring and the frame
  //      corners to the staged rows, then 16 write-back passes of 16-byte stores run without a barrier between them (round 2 staged
  //      and wrote 8 rows at a time: three barriers per half, half of the waves idle while the other half staged) ----
  constexpr int CROW = BN * 2 + 16;
  constexpr int HALF_BYTES = 256 * CROW;
  unsigned char* const ctile = smem;
  float* const cred = reinterpret_cast<float*>(smem + 2 * HALF_BYTES);            // corner dot products: 4 x BN floats
  float* const rscr = cred + 4 * BN;                                              // reduce_rows16 scratch: 8 x (BN / 8) x 32 floats
  const float slope = act_slope(act);
  constexpr int CPR = BN / 8;             // 16-byte chunks per tile row
  constexpr int RPP = 512 / CPR;          // rows per pass
  constexpr int NPASS = 512 / RPP;        // write-back passes over the tile (8-row half h = passes [h * NPASS / 2, (h + 1) * NPASS / 2))
  const int chunk = tid % CPR, rsub = tid / CPR;
  const int ncol = n0 + chunk * 8;
  // statistics: one record per 8 x 32 half tile, i.e. the record grid of the 8 x 32 tile kernel (conv_halo.hip) and of
  // dei2i_conv2d_stats_chunks: record (2 * tile_row + half, tile_col) of the image
  const int rec0 = img * (2 * tiles_y * tiles_x) + (2 * (trem / tiles_x)) * tiles_x + (trem % tiles_x);
  // EPIN: the per-channel coefficients of this thread's 8 channels as 4 pairs (packed fp32 arithmetic below) -- kind 1: mean | rstd
  // of the image; kind 2: a | b | mean | rstd
  f32x2 nc[EPIN ? 4 : 1][4];
  u32x4 gmi = {0u, 0u, 0u, 0u}, bti = {0u, 0u, 0u, 0u};      // kind 1: gamma | beta of the interior class (2, 2), packed
  u32x4 xq[EPIN ? NPASS : 1];
  if constexpr (EPIN) {
    if (ncol < ldc) {
      auto ld8 = [&](const float* q, f32x2 (&d)[4]) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(q), hi = *reinterpret_cast<const f32x4*>(q + 4);
        d[0] = f32x2{lo.x, lo.y}; d[1] = f32x2{lo.z, lo.w}; d[2] = f32x2{hi.x, hi.y}; d[3] = f32x2{hi.z, hi.w};
      };
      if ((en.kind & 0xff) == 1) {
        ld8(en.mean + (size_t)img * ldc + ncol, nc[0]);
        ld8(en.rstd + (size_t)img * ldc + ncol, nc[1]);
        const bf16_t* gp = en.gb + ((size_t)(img * 5 + 2) * 5 + 2) * 2 * ldc + ncol;
        gmi = *reinterpret_cast<const u32x4*>(gp);
        bti = *reinterpret_cast<const u32x4*>(gp + ldc);
      } else {
        const size_t grow = en.group_images > 0 ? (size_t)(img / en.group_images) * ldc + ncol : (size_t)ncol;
        ld8(en.a + grow, nc[0]); ld8(en.b + grow, nc[1]);
        ld8(en.mean + grow, nc[2]); ld8(en.rstd + grow, nc[3]);
      }
    }
  }
  __syncthreads();                        // every wave is out of the loop's LDS buffers
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int col0 = wn * WTN + j * 16 + 4 * kg;
    float bq[4];
    uint32_t m01, m23;
    epi_col_consts(bias, n0 + col0, wrows, bq, m01, m23);
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int row = (wm * 4 + (i >> 1)) * 32 + (i & 1) * 16 + l16;
      *reinterpret_cast<u32x2*>(ctile + row * CROW + col0 * 2) =
          epi_finish4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3], bq, slope, m01, m23);
    }
  }
  if constexpr (EPIN) {
    // the thread's x rows (one 16-byte load per write-back pass), ALL of them, requested here -- the accumulators are dead --
    // ahead of the barriers and the ring / corner work on the staged tile: their HBM latency is off the write-back's path
    if (ncol < ldc) {
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int row = p * RPP + rsub;
        const int py = y0 + (row >> 5), px = x0 + (row & 31);
        const size_t xpix = ((size_t)(img * (g.Ho >> en.up) + (py >> en.up))) * (g.Wo >> en.up) + (px >> en.up);
        xq[p] = *reinterpret_cast<const u32x4*>(en.x + xpix * ldc + ncol);
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
    if (ring_kind == 1 || ring_kind == 2) ring_add((ring_kind == 1 ? 1 : H16_TH - 2) * 32 + wm * 16 + l16);
    __syncthreads();
    if (ring_kind >= 3) ring_add(l16 * 32 + (ring_kind == 3 ? 1 : H16_TW - 2));
    // the frame's CORNER pixels this tile holds the image of: frame (-1,-1) is read by output (0,0) only, through kernel tap
    // (0,0), and reflects onto input pixel (1,1): dx[1][1] += W[:,:,0,0]^T dy[0][0]; likewise (-1,W) -> dx[1][W-2] through
    // tap (0,2) with dy[0][W-1], (H,-1) -> dx[H-2][1] through (2,0) with dy[H-1][0], (H,W) -> dx[H-2][W-2] through (2,2)
    // with dy[H-1][W-1].  One dot product of length Cs per output channel: 4 threads per channel, summed through LDS.
    // (a tile is at most 16 rows of the >= 32-row image: it holds the top OR the bottom corner of a side, never both)
    const bool lef = x0 == 0, rig = x0 + H16_TW == g.Wo, top = y0 == 0, bot = y0 + H16_TH == g.Ho;
    const bool has = (top || bot) && (lef || rig);      // workgroup-uniform
    if (has) {
      const int ky = top ? 0 : 2, kx = lef ? 0 : 2;
      const int sy = ky ? g.Ho - 1 : 0, sx = kx ? g.Wo - 1 : 0;
      const int c = tid % BN, part = tid / BN;
      const int per = g.Cs >> 2;                            // Cs % 32 == 0: a multiple of 8
      float sum = 0.f;
      if (part < 4 && n0 + c < wrows) {
        const bf16_t* wrow = wgt + (size_t)(n0 + c) * g.K + (ky * 3 + kx) * g.Cs + part * per;
        const bf16_t* dyp = src + ((size_t)(img * g.Hs + sy) * g.Ws + sx) * g.Cs + part * per;
        for (int q = 0; q < per; q += 8) {
          float w8[8], d8[8];
          Elem<bf16_t>::unpack(*reinterpret_cast<const u32x4*>(wrow + q), w8);
          Elem<bf16_t>::unpack(*reinterpret_cast<const u32x4*>(dyp + q), d8);
#pragma unroll
          for (int e = 0; e < 8; ++e) sum = fmaf(w8[e], d8[e], sum);
        }
      }
      if (part < 4) cred[part * BN + c] = sum;
    }
    __syncthreads();
    if (has && tid < BN) {
      const int c = tid;
      const int crow = (top ? 1 : H16_TH - 2) * 32 + (lef ? 1 : H16_TW - 2);
      const float tot = (cred[c] + cred[BN + c]) + (cred[2 * BN + c] + cred[3 * BN + c]);
      bf16_t* p = reinterpret_cast<bf16_t*>(ctile + crow * CROW) + c;
      *p = f32_to_bf16(bf16_to_f32(*p) + tot);
    }
    if (has) __syncthreads();
  }
  // sums of this thread's 8 channels over the rows it writes back -- EPIN: 4 pairs x the (up to) 4 sums of the norm's backward
  // over the whole tile (one record per 16 x 32 tile: dei2i_conv2d_dgrad_norm_chunks); statistics: per 8-row half (below)
  f32x2 ep[EPIN ? 4 : 1][4];
#pragma unroll
  for (int q = 0; q < (EPIN ? 4 : 1); ++q)
#pragma unroll
    for (int k = 0; k < 4; ++k) ep[q][k] = f32x2{0.f, 0.f};
  constexpr int HP = NPASS / 2;           // passes per 8-row half
  // (the two halves are one rolled loop: the 16 passes unrolled are ~16 KB of code with the EPIN arithmetic, and every dispatch
  //  walks its code once from a cold instruction cache -- measured +0.7 ms per step)
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    float st8[16];                        // statistics: sum (0..7) / sum of squares (8..15) of this thread's 8 channels, this half
#pragma unroll
    for (int k = 0; k < 16; ++k) st8[k] = 0.f;
    if (ncol < ldc) {
#pragma unroll
      for (int p = 0; p < HP; ++p) {
        const int row = (half * HP + p) * RPP + rsub;
        const int py = y0 + (row >> 5), px = x0 + (row & 31);
        const size_t opix = (size_t)out_pixel(g, img, py, px);
        const u32x4 v = *reinterpret_cast<const u32x4*>(ctile + row * CROW + chunk * 16);
        *reinterpret_cast<u32x4*>(out + opix * ldc + ncol) = v;
        if constexpr (!EPIN) {
          if (stats != nullptr) {
            float fv[8];
            Elem<bf16_t>::unpack(v, fv);
#pragma unroll
            for (int e = 0; e < 8; ++e) { st8[e] += fv[e]; st8[8 + e] = fmaf(fv[e], fv[e], st8[8 + e]); }
          }
        }
        if constexpr (EPIN) {
          // v = dL/dz of 8 channels of pixel (py, px), as the streaming pass would read it back (bf16): the same per-element
          // arithmetic as spade_bwd_partial_kernel / bn_bwd_partial_kernel (reduce.hip), on channel PAIRS (v_pk_*_f32)
          const uint32_t dw[4] = {v.x, v.y, v.z, v.w}, xw[4] = {xq[p].x, xq[p].y, xq[p].z, xq[p].w};
          if (en.kind & 0x100) {                         // (timing only, tools/diag_epin.py: no arithmetic -- the loads stay live)
            ep[0][0] += bf16x2_unpack(xw[0]) + bf16x2_unpack(dw[0]);
          } else if ((en.kind & 0xff) == 1) {
            const int cy = border_class(py, g.Ho), cx = border_class(px, g.Wo);
            const bool interior = cy == 2 && cx == 2;
            u32x4 gq = gmi, bq2 = bti;                   // the interior class (kept in registers); the frame's pixels fetch theirs
            if (!interior) {
              const bf16_t* gp = en.gb + ((size_t)(img * 5 + cy) * 5 + cx) * 2 * ldc + ncol;
              gq = *reinterpret_cast<const u32x4*>(gp);
              bq2 = *reinterpret_cast<const u32x4*>(gp + ldc);
            }
            const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, bw[4] = {bq2.x, bq2.y, bq2.z, bq2.w};
            const float inm = interior ? 1.f : 0.f;      // the interior class's d gamma, d beta come from here; the frame's classes
            const f32x2 inm2 = {inm, inm};               // from spade_bwd_border_kernel
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 d2 = bf16x2_unpack(dw[e]);
              const f32x2 xh = (bf16x2_unpack(xw[e]) - nc[0][e]) * nc[1][e];
              const f32x2 g1 = bf16x2_unpack(gw[e]) + f32x2{1.f, 1.f};
              const f32x2 z = __builtin_elementwise_fma(xh, g1, bf16x2_unpack(bw[e]));
              const f32x2 gg = {z.x > 0.f ? d2.x : 0.f, z.y > 0.f ? d2.y : 0.f};
              const f32x2 dxh = gg * g1;
              ep[0][e] += dxh;
              ep[1][e] = __builtin_elementwise_fma(dxh, xh, ep[1][e]);
              const f32x2 gi = gg * inm2;
              ep[2][e] = __builtin_elementwise_fma(gi, xh, ep[2][e]);
              ep[3][e] += gi;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2 x2 = bf16x2_unpack(xw[e]);
              const f32x2 zz = __builtin_elementwise_fma(nc[0][e], x2, nc[1][e]);
              const f32x2 gg = bf16x2_unpack(dw[e]) * f32x2{act_grad_from_out(zz.x, en.act), act_grad_from_out(zz.y, en.act)};
              ep[0][e] += gg;
              ep[1][e] = __builtin_elementwise_fma(gg, (x2 - nc[2][e]) * nc[3][e], ep[1][e]);
            }
          }
        }
      }
    }
    if constexpr (EPIN) {                                // the second half's x rows move down (static register indices in the loop)
#pragma unroll
      for (int p = 0; p < HP; ++p) xq[p] = xq[HP + p];
    }
    // statistics: the 512 / CPR row groups' partials -> wave butterflies -> LDS -> the 8 waves' sums in order (reduce_rows16: one
    // barrier; each half has its own scratch)
    if constexpr (!EPIN) {
      if (stats != nullptr) {                            // kernel-uniform
        float* const srow = stats + (size_t)(rec0 + half * tiles_x) * 2 * ldc;
        reduce_rows16<16, CPR>(st8, rscr + half * (8 * CPR * 16), tid, [&](int q, int c, float sum) {
          if (n0 + c < ldc) srow[(size_t)q * ldc + n0 + c] = sum;
        });
      }
    }
  }
  if constexpr (EPIN) {
    const int nq = (en.kind & 0xff) == 1 ? 4 : 2;
    if (!(en.kind & 0x200)) {                            // (0x200: timing only, no reduction)
      float* const prow = en.partial + (size_t)(img * (tiles_y * tiles_x) + trem) * nq * ldc;
      float flat[32];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) { flat[q * 8 + 2 * k] = ep[q][k].x; flat[q * 8 + 2 * k + 1] = ep[q][k].y; }
      reduce_rows16<32, CPR>(flat, rscr, tid, [&](int q, int c, float sum) {
        if (q < nq && n0 + c < ldc) prow[(size_t)q * ldc + n0 + c] = sum;
      });
    }
  }
  if (DIAG && drec != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    drec[3] = (__builtin_amdgcn_s_memtime() - kt0) | ((st0 - kt0) << 32);     // whole kernel | prologue (cycles)
  }
}

// (A four-wave variant of this tile -- one wave per SIMD, 32x32x16 MFMAs, 256 accumulators, software-pipelined by hand -- was
//  built and measured in round 2: equal to or slower than the eight-wave two-phase loop (108 us on the res-block shape with the
//  DMA pieces staggered per wave, ~equal without): one wave per SIMD has nobody to cover its stalls.  Removed)

extern int g_v2_ablate;
extern unsigned long long* g_v2_dbg;
extern int g_halo_bn, g_halo_stages;
int g_halo16 = 3;
int g_halo16_fold = 1;         // A/B option "halo16_fold": 0 = ring GEMM + finalize + border fold as separate launches              // A/B option "halo16": 0 = always the 8 x 32 tile kernel (conv_halo.hip)
int g_halo16_stages = 8;

template <int BN, int STAGES>
static hipError_t launch_halo16(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out,
                                int ldc, int act, hipStream_t st, float* stats, const void* ring = nullptr, bool fold = false,
                                const EpiNorm* en = nullptr, bool s2 = false) {
  const int tiles_m = g.N * (g.Ho / H16_TH) * (g.Wo / H16_TW);
  const int tiles_n = (ldc + BN - 1) / BN;
  if (s2) {                                             // the 4x4 stride-2 form: two-phase loop, 4-stage weight ring
    if constexpr (STAGES == 4) {
      constexpr size_t lds2 = std::max(2 * (size_t)H16_HBYTES + 4 * (size_t)BN * 64 + 4 * H16_GROUPS * 16 * sizeof(int),
                                       2 * 256 * (size_t)(BN * 2 + 16) + 4 * BN * sizeof(float) + 8 * (BN / 8) * 32 * sizeof(float));
      auto k2 = halo16_conv_kernel<BN, 4, 0, false, 0, false, false, true>;
      static bool attr2 = false;
      if (!attr2) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        if (e != hipSuccess) return e;
        attr2 = true;
      }
      count_launch(K_HALO16_S2);
      prof_begin(PROF_GATHER_GEMM, 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)wrows, st);
      hipLaunchKernelGGL(k2, dim3(tiles_m * tiles_n), dim3(512), lds2, st, g, (const bf16_t*)src, (const bf16_t*)wgt, wrows, bias,
                         (bf16_t*)out, ldc, act, tiles_n, stats, (unsigned long long*)nullptr, (const bf16_t*)nullptr, 0, EpiNorm{});
      prof_end(PROF_GATHER_GEMM, st);
      return hipGetLastError();
    }
    return hipErrorNotSupported;
  }
  constexpr size_t loop_lds = 2 * (size_t)H16_HBYTES + (size_t)STAGES * BN * 64 + 4 * H16_GROUPS * 16 * sizeof(int);   // (4 tables: S2)
  constexpr size_t epi_lds = 2 * 256 * (size_t)(BN * 2 + 16) + 4 * BN * sizeof(float) + 8 * (BN / 8) * 32 * sizeof(float);   // tile | corners | reduce scratch
  const size_t lds = std::max(loop_lds, epi_lds);
  const bool diag = g_v2_ablate >= 6 && g_v2_ablate <= 9 && g_v2_dbg != nullptr && STAGES == 8 && !fold;
  constexpr bool S8 = STAGES == 8 && BN == 128;   // the diagnostic builds exist for the shipped ring depth and the 128-channel tile only
  auto kern = halo16_conv_kernel<BN, STAGES, 0>;
  if (diag && S8) {
    kern = g_v2_ablate == 6 ? halo16_conv_kernel<BN, STAGES, S8 ? 1 : 0> : g_v2_ablate == 7 ? halo16_conv_kernel<BN, STAGES, S8 ? 2 : 0>
         : g_v2_ablate == 8 ? halo16_conv_kernel<BN, STAGES, S8 ? 3 : 0> : halo16_conv_kernel<BN, STAGES, S8 ? 4 : 0>;
  }
  // (the FOLD build of the pipelined loop needs 20 more registers -- ring accumulators and fragment -- than a wave has: it
  //  spills inside the loop, and a scratch reload drains the LDS-DMA queue.  Those launches keep the two-phase loop.)
  // (and its read schedule and wait counts are written for 4 channel blocks per wave: the 64-channel tile keeps the two-phase loop too)
  const bool pipe = g_halo16 == 3 && STAGES == 8 && BN == 128 && !fold;
  if (pipe) {
    constexpr bool P8 = STAGES == 8 && BN == 128;
    kern = halo16_conv_kernel<BN, STAGES, 0, false, 0, P8>;
    if (!fold && BN == 128 && g_v2_ablate > 20 && g_v2_ablate <= 51) {
      constexpr bool Q = P8 && BN == 128;
      switch (g_v2_ablate - 20) {
        case 8: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 8 : 0, Q>; break;
        case 16: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 16 : 0, Q>; break;
        case 24: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 24 : 0, Q>; break;
        case 12: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 12 : 0, Q>; break;
        case 9: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 9 : 0, Q>; break;
        case 25: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 25 : 0, Q>; break;
        case 27: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 27 : 0, Q>; break;
        case 26: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 26 : 0, Q>; break;
        case 2: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 2 : 0, Q>; break;
        case 31: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 31 : 0, Q>; break;
        case 1: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 1 : 0, Q>; break;
        case 4: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 4 : 0, Q>; break;
        case 5: kern = halo16_conv_kernel<BN, STAGES, 0, false, Q ? 5 : 0, Q>; break;
        default: break;
      }
    }
    static const void* attr_set[16] = {};               // (per template instance of this launcher: the kernels it has prepared)
    bool seen = false;
    for (const void* q : attr_set) seen = seen || q == reinterpret_cast<const void*>(kern);
    if (!seen) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      for (const void*& q : attr_set)
        if (q == nullptr) { q = reinterpret_cast<const void*>(kern); break; }
    }
  }
  const bool abl = !pipe && g_v2_ablate > 10 && g_v2_ablate <= 17 && S8 && !fold;
  if (abl) {
    switch (g_v2_ablate - 10) {
      case 1: kern = halo16_conv_kernel<BN, STAGES, 0, false, S8 ? 1 : 0>; break;
      case 2: kern = halo16_conv_kernel<BN, STAGES, 0, false, S8 ? 2 : 0>; break;
      case 3: kern = halo16_conv_kernel<BN, STAGES, 0, false, S8 ? 3 : 0>; break;
      case 4: kern = halo16_conv_kernel<BN, STAGES, 0, false, S8 ? 4 : 0>; break;
      case 5: kern = halo16_conv_kernel<BN, STAGES, 0, false, S8 ? 5 : 0>; break;
      case 6: kern = halo16_conv_kernel<BN, STAGES, 0, false, S8 ? 6 : 0>; break;
      default: kern = halo16_conv_kernel<BN, STAGES, 0, false, S8 ? 7 : 0>; break;
    }
  }
  if (fold && !pipe) kern = halo16_conv_kernel<BN, STAGES, 0, STAGES == 8>;   // (instantiated for the shipped ring depth only)
  if (fold && !pipe && en != nullptr) kern = halo16_conv_kernel<BN, STAGES, 0, STAGES == 8, 0, false, STAGES == 8>;
  if ((fold && STAGES != 8) || (en != nullptr && !fold)) return hipErrorNotSupported;
  static bool attr_done[16] = {};
  const int which = fold ? (en != nullptr ? 15 : 2) : (abl ? 7 + (g_v2_ablate - 10) : (diag && S8 ? g_v2_ablate - 3 : 0));
  if (!pipe && !attr_done[which]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done[which] = true;
  }
  count_launch(K_HALO16_CONV);
  // (profiling families: the pipelined forward instance is a kernel of its own -- bench.py's roofline line --, the FOLD launches
  //  are another, the 64-channel tile counts with the other conv kernels)
  const ProfFamily fam = fold ? PROF_HALO_FOLD : (pipe ? PROF_HALO_CONV : PROF_GATHER_GEMM);
  prof_begin(fam, 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)wrows, st);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, st, g, (const bf16_t*)src, (const bf16_t*)wgt, wrows, bias,
                     (bf16_t*)out, ldc, act, tiles_n, stats, g_v2_dbg, (const bf16_t*)ring,
                     ring != nullptr ? ring_pixels(g.Hl, g.Wl) : 0, en != nullptr ? *en : EpiNorm{});
  prof_end(fam, st);
  return hipGetLastError();
}

// returns hipErrorNotSupported when the shape does not qualify (the caller goes on to the 8 x 32 tile kernel)
// ring: see the kernel (SPADE -> upsample -> conv with z kept at the source resolution); only the 8-wave kernel takes it
// fold: the launch is the interior input gradient of a reflect-padded 3x3 conv and also folds the frame's ring and corners
// into the border rows / columns -- see the kernel
int g_halo16_s2 = 1;           // A/B option "halo16_s2": 0 = the 4x4 stride-2 convs stay on the gather GEMM

// the 4x4 stride-2 forward convs (kernel template S2): reflect or zero padding 1, 64 input channels.  Measured against the LDS-DMA
// gather GEMM on the step's shapes, operands cold (tools/bench_s2.py): 64 -> 128 @256^2 x16 115 vs 135 us, x32 207 vs 235, D's
// 64 -> 128 @128^2 x64 109 vs 135 -- but 128 -> 256 @128^2 x16 98 vs 86 and D's 128 -> 256 @64^2 x64 83 vs 80: per k-step this form
// moves 8 KB of weights + 9 KB of halo (a plane's halo serves 4 taps, the 3x3 kernel's 9), and the gather GEMM's 128-channel
// k-steps are twice as efficient as its 64-channel ones -- so only the 64-channel inputs are taken.
bool halo16_s2_shape_ok(const GatherDesc& g, int ldc, int num_cu) {
  if (!g_halo16_s2 || !g_halo16 || g_halo_bn != 0 || g_halo_stages != 0) return false;
  if (g.sh != 2 || g.sw != 2 || g.th != 4 || g.tw != 4 || g.ys != 1 || g.xs != 1 || g.by0 != -1 || g.bx0 != -1 || g.up || g.wK != g.K || g.wtw != 4) return false;
  if (g.Cs != 64 && g_halo16_s2 != 2) return false;                                        // (option value 2: every channel count, for A/B)
  if (g.Cs % 32 != 0 || g.Cs < 64 || g.Ho % H16_TH != 0 || g.Wo % H16_TW != 0 || g.M != g.N * g.Ho * g.Wo || !g.out_identity) return false;
  if (g.Hl != 2 * g.Ho || g.Wl != 2 * g.Wo) return false;
  if ((long long)g.N * g.Hs * g.Ws * g.Cs >= (1ll << 31)) return false;                   // 32-bit offset table
  if (ldc < 128 || ldc % 8 != 0) return false;                                             // (the 128-channel tile)
  const int tiles_m = g.N * (g.Ho / H16_TH) * (g.Wo / H16_TW);
  return tiles_m * ((ldc + 127) / 128) >= (num_cu * 7) / 8;
}

static hipError_t halo16_conv_s2(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out,
                                 int ldc, int act, int num_cu, hipStream_t st, float* stats) {
  if (!halo16_s2_shape_ok(g, ldc, num_cu)) return hipErrorNotSupported;
  return launch_halo16<128, 4>(g, src, wgt, wrows, bias, out, ldc, act, st, stats, nullptr, false, nullptr, true);
}

hipError_t halo16_conv(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out, int ldc,
                       int act, int num_cu, hipStream_t st, float* stats, const void* ring, bool fold, const EpiNorm* en) {
  if (g.sh == 2 && g.sw == 2 && !fold && en == nullptr && ring == nullptr)
    return halo16_conv_s2(g, src, wgt, wrows, bias, out, ldc, act, num_cu, st, stats);
  if (en != nullptr && (!fold || ((en->kind & 0xff) != 1 && (en->kind & 0xff) != 2))) return hipErrorNotSupported;
  if (fold && (!g_halo16_fold || g.ys >= 0 || g.xs >= 0 || g.pad_mode != PAD_ZERO || g.up || g.Ho < 2 * H16_TH || g.Wo < 2 * H16_TW ||
               bias != nullptr || act != ACT_NONE || stats != nullptr || ring != nullptr))
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
  if (fold && g_halo16_stages != 8) return hipErrorNotSupported;
  if (ldc >= 128) {
    if (g_halo16_stages == 4) return launch_halo16<128, 4>(g, src, wgt, wrows, bias, out, ldc, act, st, stats);
    if (g_halo16_stages == 6) return launch_halo16<128, 6>(g, src, wgt, wrows, bias, out, ldc, act, st, stats);
    return launch_halo16<128, 8>(g, src, wgt, wrows, bias, out, ldc, act, st, stats, ring, fold, en);
  }
  if (g_halo16_stages == 4) return launch_halo16<64, 4>(g, src, wgt, wrows, bias, out, ldc, act, st, stats);
  return launch_halo16<64, 8>(g, src, wgt, wrows, bias, out, ldc, act, st, stats, ring, fold, en);
}

}  // namespace dei2i
