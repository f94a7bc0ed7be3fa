// Thin-input wgrad (bf16, stride 1, 8-channel inputs: the generator stem 3 -> 64, 7x7 reflect):
//   dw[co][ty][tx][ci] = sum over pixels p of dy[p][co] * x[p + (ty, tx)][ci]
//
// The generic wgrad kernel gathers one 16-byte pixel vector per (pixel, tap) from L2 -- 49 gathers per pixel for 49 x 8
// k-columns -- and ran this layer at 0.7 TB/s of its 150 MB (193 us).  Here, as in wgrad_halo.hip, a workgroup walks
// 4 x 32 pixel half-tiles of its pixel split and loads, per half-tile, the dy tile (128 px x 64 co, 16 KB) and the
// (4+6) x (32+6) input halo (380 px x 16 B, 6 KB) ONCE by LDS-DMA; every tap then reads its B fragment from the halo.
//
// MFMA view: D[co][col] += sum_px A[co][px] * B[px][col] with 32 columns = (4 adjacent taps tx..tx+3) x (8 channels):
// the halo is pixel-major [pixel][8 ch], so the 64 bytes that start at pixel (p + tx0) ARE those 32 columns of reduction
// row p -- the B operand is an overlapping window view, fetched with ds_read_b64_tr_b16 like any pixel-major tile.
// A 7-tap row is two such windows (tx 0..3, 4..7; the eighth tap is computed and thrown away), 14 "virtual taps" in all.
//
//   waves  : 8 = 2 (32-channel blocks of co) x 4 (virtual-tap groups: vt = tg, tg+4, tg+8, tg+12); 64 accumulators/thread
//   grid   : pixel splits only (one co tile, one channel slice); per-split fp32 slabs [Cout][49][8], plain stores, summed
//            and un-packed by wgrad_reduce_unpack (two-level: few (co, channel-block) workgroups, many slabs)
//   bound  : HBM / LDS-DMA (22 KB per half-tile against 32 MFMAs per wave)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "common.h"
#include "geom.h"
#include "launch.h"

namespace dei2i {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_wt[256];

typedef __attribute__((address_space(3))) void lds_void_wt;
typedef __attribute__((address_space(1))) const void gbl_void_wt;

DEI2I_D void glds16wt(const void* gptr, unsigned char* lds_wave_base) {
  glds16_asm(gptr, lds_wave_base);      // (common.h: hipcc must not see the LDS write, or it drains the ring)
}

constexpr int WT_TH = 4, WT_TW = 32;                 // half-tile: 128 pixels
constexpr int WT_BCO = 64;                           // output channels per workgroup
constexpr int WT_ROWB_A = WT_BCO * 2;                // 128-byte dy rows
constexpr int WT_A_BYTES = 128 * WT_ROWB_A;          // 16 KB

template <int KH, int KW>
__global__ __launch_bounds__(512) void wgrad_thin_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                         const bf16_t* __restrict__ dy, const int co_rows, const int ldy,
                                                         float* __restrict__ slabs, const int tiles_per_split,
                                                         const long long slab_elems) {
  constexpr int HH = WT_TH + KH - 1, HWD = WT_TW + KW - 1;      // halo rows / columns (10 x 38)
  constexpr int HPIX = HH * HWD;
  constexpr int HGROUPS = (HPIX + 4 + 63) / 64;                  // 64-pixel DMA instructions (+4: the window overrun)
  constexpr int B_BYTES = HGROUPS * 1024;
  constexpr int STAGE = WT_A_BYTES + B_BYTES;
  constexpr int TXG = (KW + 3) / 4, NVT = KH * TXG;              // virtual taps: (ty, group of 4 tx)
  constexpr int TPW = (NVT + 3) / 4;                             // per wave
  static_assert(HGROUPS <= 8, "one halo DMA instruction per wave");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave & 1, tg = wave >> 1;

  const int split = blockIdx.x;
  const int tiles_x = g.Wo / WT_TW, tiles_y = g.Ho / WT_TH;
  const int tiles_img = tiles_x * tiles_y;
  const int ntiles = g.N * tiles_img;
  const int tbeg = split * tiles_per_split;
  const int tend = min(ntiles, tbeg + tiles_per_split);
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_wt);

  // dy: one instruction = 8 pixel rows x 128 B: lane l -> row (l>>3), 16-byte slot (l&7), rows swizzled for the
  //     transposed reads (rows r and r+2 alias: flip the 64-byte half); 16 instructions per half-tile, 2 per wave
  // x : one instruction = 64 halo pixels x 16 B, no swizzle (the windows overlap); waves 0 .. HGROUPS-1 issue one each
  auto issue = [&](int stage, int t) {
    unsigned char* sa = smem + stage * STAGE;
    unsigned char* sb = sa + WT_A_BYTES;
    const int img = t / tiles_img;
    const int rem = t - img * tiles_img;
    const int tyi = rem / tiles_x;
    const int y0 = tyi * WT_TH, x0 = (rem - tyi * tiles_x) * WT_TW;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rgrp = (j * 8 + wave) * 8;
      const int r = rgrp + (lane >> 3);                                   // pixel of the half-tile: (r>>5, r&31)
      const int off = ((lane & 7) * 16) ^ (((r >> 1) & 1) << 6);
      const int c = off >> 1;
      const size_t pix = ((size_t)img * g.Ho + y0 + (r >> 5)) * g.Wo + x0 + (r & 31);
      const bf16_t* p = c < ldy ? dy + pix * ldy + c : zero;
      glds16wt(p, sa + rgrp * WT_ROWB_A);
    }
    if (wave < HGROUPS) {
      const int hp = wave * 64 + lane;
      const int hy = hp / HWD, hx = hp - hy * HWD;
      const bf16_t* p = zero;
      if (hp < HPIX) {
        const int y = bound_coord(y0 + g.by0 + hy, g.Hl, g.pad_mode);
        const int x = bound_coord(x0 + g.bx0 + hx, g.Wl, g.pad_mode);
        if ((y | x) >= 0) p = src + ((size_t)((img * g.Hs + y) * g.Ws + x)) * 8;
      }
      glds16wt(p, sb + wave * 1024);
    }
  };

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  // transposed-read lane roles (ds_read_b64_tr_b16): lane i = 4q+p of a 16-lane group supplies row q, cols 4p..4p+3
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_half = (lane >> 4) & 1;
  const int a_colb = (cb * 32 + 16 * tr_half + 4 * tr_p) * 2;           // byte column in a dy row
  const int b_colb = (16 * tr_half + 4 * tr_p) * 2;                     // byte offset in the 64-byte tap window

  auto tr_read = [&](const unsigned char* base, int o0, int o1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o1));
    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 r;
    r.x = l2.x; r.y = l2.y; r.z = h2.x; r.w = h2.y;
    return r;
  };

  auto compute = [&](int stage) {
    const unsigned char* ab = smem + stage * STAGE;
    const unsigned char* bb = ab + WT_A_BYTES;
#pragma unroll 2
    for (int kb = 0; kb < 8; ++kb) {                                      // 16-pixel reduction blocks of the half-tile
      const int ra = kb * 16 + 8 * lh + tr_q;                             // dy rows ra, ra+4
      const u32x4 af = tr_read(ab, ra * WT_ROWB_A + (a_colb ^ (((ra >> 1) & 1) << 6)),
                               (ra + 4) * WT_ROWB_A + (a_colb ^ ((((ra + 4) >> 1) & 1) << 6)));
      const int py = kb >> 1, px0 = (kb & 1) * 16;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int vt = tg + 4 * t;                                        // wave-uniform
        if (vt >= NVT) break;
        const int ty = vt / TXG, tx0 = (vt - ty * TXG) * 4;
        const int rb = (py + ty) * HWD + px0 + tx0 + 8 * lh + tr_q;       // halo pixels rb, rb+4 (window start)
        const u32x4 bf = tr_read(bb, rb * 16 + b_colb, (rb + 4) * 16 + b_colb);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[t], 0, 0, 0);
      }
    }
  };

  // two-stage ring over the half-tiles of this split (rotated: `issue` and `compute` exist once in the code)
  const int nt = tend - tbeg;
#pragma unroll 1
  for (int it = -1; it < nt; ++it) {
    if (it >= 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (it + 1 < nt) issue((it + 1) & 1, tbeg + it + 1);
    if (it >= 0) compute(it & 1);
  }

  // partial block -> this split's slab [Cout][KH*KW][8]: lane lr holds column (tx0 + lr/8, channel lr%8)
  float* slab = slabs + (size_t)split * slab_elems;
  const int co_w = __builtin_amdgcn_readfirstlane(cb * 32);
  const int txl = lr >> 3, ci = lr & 7;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int vt = tg + 4 * t;
    if (vt >= NVT) break;
    const int ty = vt / TXG, tx = (vt - ty * TXG) * 4 + txl;
    if (tx >= KW) continue;                                               // the padding tap of the second window
    float* pt = slab + (size_t)(co_w + 4 * lh) * g.K + (ty * KW + tx) * 8 + ci;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (co_w + 8 * q >= co_rows) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[(size_t)(8 * q + r) * g.K] = acc[t][q * 4 + r];
    }
  }
}

int g_wt_splits_per_cu = 0;      // A/B option "wgrad_thin_splits"
// returns hipErrorNotSupported when the shape does not qualify (the caller falls through to the other wgrad kernels)
hipError_t wgrad_thin(const GatherDesc& g, const void* src, const void* dy, int co_rows, int ldy, float* slabs,
                      size_t slab_capacity_elems, int num_cu, int* nsplit_out, hipStream_t st) {
  if (g.Cs != 8 || g.sh != 1 || g.sw != 1 || g.ys != 1 || g.xs != 1 || g.up != 0 || g.th != 7 || g.tw != 7)
    return hipErrorNotSupported;
  if (g.Ho % WT_TH != 0 || g.Wo % WT_TW != 0 || co_rows > WT_BCO || ldy < 8) return hipErrorNotSupported;
  if ((long long)g.N * g.Hs * g.Ws * g.Cs >= (1ll << 31)) return hipErrorNotSupported;
  const int ntiles = g.N * (g.Ho / WT_TH) * (g.Wo / WT_TW);
  if (ntiles < 64) return hipErrorNotSupported;
  int splits = std::min((g_wt_splits_per_cu > 0 ? g_wt_splits_per_cu : 2) * num_cu, ntiles / 4);   // 44 KB of LDS per workgroup
  const long long slab_elems = wgrad_slab_elems(co_rows, g.K);
  if ((size_t)slab_elems * splits > slab_capacity_elems) splits = (int)(slab_capacity_elems / (size_t)slab_elems);
  if (splits < 1) return hipErrorNotSupported;
  const int tps = (ntiles + splits - 1) / splits;
  const int zs = (ntiles + tps - 1) / tps;
  constexpr int HPIX = (WT_TH + 6) * (WT_TW + 6);
  const size_t lds = 2 * (size_t)(WT_A_BYTES + ((HPIX + 4 + 63) / 64) * 1024);
  auto kern = wgrad_thin_kernel<7, 7>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  count_launch(K_WGRAD_THIN);
  prof_begin(PROF_WGRAD, 2.0 * (double)g.M * 49.0 * (double)g.Clog * (double)co_rows, st);
  hipLaunchKernelGGL(kern, dim3(zs), dim3(512), lds, st, g, (const bf16_t*)src, (const bf16_t*)dy, co_rows, ldy, slabs, tps,
                     slab_elems);
  prof_end(PROF_WGRAD, st);
  *nsplit_out = zs;
  return hipGetLastError();
}

}  // namespace dei2i
