// Thin-input convs (bf16, source channel stride 8 = one 16-byte vector per pixel, <= 64 output channels):
//   the generator stem (3 -> 64, 7x7 reflect), the discriminator's first conv (3 -> 64, 4x4 stride 2) and the interior
//   dgrad of the 64 -> 4 heads (dY has 4(8) channels, dX 64).
//
// As implicit GEMMs these have K = taps * 8 (392 / 128 / 72) and an output of 64 channels per pixel: the work is the
// 134 MB output stream, and on the gather-GEMM kernels every 256-pixel tile pays a pipeline fill for 2..7 k-steps
// (measured 172 us for the stem against a 25 us HBM floor).  Here the whole weight matrix [64][K] lives in LDS for the
// lifetime of a persistent workgroup, which walks 8 x 32 pixel tiles with ONE barrier per tile: the input halo of tile
// t+1 (<= 19 KB) arrives by LDS-DMA while tile t computes.  A k-block of the MFMA (16 k) is two taps x 8 channels, so
// lane (row = pixel, half h) reads its A fragment -- the 16-byte pixel vector at tap 2*kb + h -- with one ds_read_b128
// straight from the halo; B fragments come from the LDS weight rows (row stride 16 * odd bytes: conflict-free).
// Each wave owns one tile row (32 pixels) x 64 channels, stages its 4 KB result in a wave-private LDS slab and writes
// full 128-byte pixel rows.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "common.h"
#include "geom.h"
#include "launch.h"

namespace dei2i {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_thin[256];

typedef __attribute__((address_space(3))) void lds_void_tc;
typedef __attribute__((address_space(1))) const void gbl_void_tc;

DEI2I_D void glds16tc(const void* gptr, unsigned char* lds_wave_base) {
  glds16_asm(gptr, lds_wave_base);      // (common.h: hipcc must not see the LDS write, or it drains the ring)
}

constexpr int TC_TH = 8, TC_TW = 32, TC_N = 64;
constexpr int TC_PAD_TAP = (int)0x80000000;      // tap-table sentinel (real offsets can be negative: dgrad walks taps backwards)
constexpr int TC_SROW = 80;                       // staging row stride: 32 channels x 2 B + 16 (2-way instead of 16-way conflicts)

// NKB = 16-wide k-blocks (two taps each): 25 (7x7), 8 (4x4), 5 (3x3).  Wave (cb, rp) owns output channels cb*32.. and
// tile rows 2rp, 2rp+1; its weight fragments (NKB x 4 registers) and tap offsets stay in registers for the whole launch,
// so a tile costs 2*NKB fragment reads and 2*NKB MFMAs per wave.  The MFMA runs "transposed" (A = weights, B = pixels):
// D[channel][pixel] puts 4 consecutive channels of one pixel in consecutive registers -> 8-byte staging writes.
template <int NKB>
__global__ __launch_bounds__(512) void thin_cin_conv_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                            const bf16_t* __restrict__ wgt, const int wrows,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                            const int ldc, const int act, const int ntiles, const int halo_bytes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const halo = smem;                                  // [2][halo_bytes]
  unsigned char* const stage = smem + 2 * halo_bytes;                // [8 waves][64 px][TC_SROW]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave & 1, rp = wave >> 1;
  const int lr = lane & 31, lh = lane >> 5;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_thin);
  const int ntaps = g.th * g.tw;

  const int hwd = (TC_TW - 1) * g.sw + g.tw, hht = (TC_TH - 1) * g.sh + g.th;
  const int npix = hwd * hht;
  const int ngroups = (npix + 63) >> 6;                              // 64-pixel LDS-DMA groups
  const int tiles_x = g.Wo / TC_TW, tiles_y = g.Ho / TC_TH;
  const int tiles_img = tiles_x * tiles_y;

  auto issue_halo = [&](int buf, int t) {
    const int img = t / tiles_img;
    const int rem = t - img * tiles_img;
    const int ty = rem / tiles_x;
    const int hy0 = ty * TC_TH * g.sh + g.by0 + (g.ys < 0 ? -(g.th - 1) : 0);
    const int hx0 = (rem - ty * tiles_x) * TC_TW * g.sw + g.bx0 + (g.xs < 0 ? -(g.tw - 1) : 0);
    for (int grp = wave; grp < ngroups; grp += 8) {                  // wave-uniform trip count
      const int p = grp * 64 + lane;
      const bf16_t* ptr = zero;
      if (p < npix) {
        const int hy = p / hwd, hx = p - hy * hwd;
        const int y = bound_coord(hy0 + hy, g.Hl, g.pad_mode);
        const int x = bound_coord(hx0 + hx, g.Wl, g.pad_mode);
        if ((y | x) >= 0) ptr = src + ((size_t)((img * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up))) * 8;
      }
      glds16tc(ptr, halo + buf * halo_bytes + grp * 1024);
    }
  };

  // ---- launch-invariant registers: weight fragments (A operand: row = channel cb*32 + lr, k = tap 2kb+lh) and the
  //      byte offset of that tap in the halo ----
  const int step_x = g.xs > 0 ? 1 : -1, step_y = g.ys > 0 ? hwd : -hwd;
  u32x4 wf[NKB];
  int toff[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int tap = 2 * kb + lh;
    const int co = cb * 32 + lr;
    u32x4 v = {0u, 0u, 0u, 0u};
    int o = TC_PAD_TAP;
    if (tap < ntaps) {
      const int ty = tap / g.tw, tx = tap - ty * g.tw;
      o = (ty * step_y + tx * step_x) * 16;
      if (co < wrows) v = *reinterpret_cast<const u32x4*>(wgt + (size_t)co * g.K + tap * 8);
    }
    wf[kb] = v;
    toff[kb] = o;
  }
  // halo byte address of this lane's pixel (tile row 2rp + i, column lr) for tap (0,0)
  int pixb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    pixb[i] = (((2 * rp + i) * g.sh + (g.ys < 0 ? g.th - 1 : 0)) * hwd + lr * g.sw + (g.xs < 0 ? g.tw - 1 : 0)) * 16;
  unsigned char* const wst = stage + wave * (64 * TC_SROW);
  float bv[16];                                                        // bias of this lane's 16 channels (D row order)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int co = cb * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
    bv[e] = (bias != nullptr && co < wrows) ? bias[co] : 0.f;
  }

  int t = blockIdx.x;
  if (t < ntiles) issue_halo(0, t);
  for (int it = 0; t < ntiles; ++it) {
    // my share of this tile's halo has landed.  vmcnt counts stores too, in issue order: the four output stores of the
    // previous tile are younger than the halo loads and stay in flight (waiting for them would serialise every tile
    // behind its own HBM write latency)
    if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();                                                 // ... everyone's; the other buffer is free
    const int tn = t + gridDim.x;
    if (tn < ntiles) issue_halo((it + 1) & 1, tn);
    const unsigned char* hb = halo + (it & 1) * halo_bytes;

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const bool pad = toff[kb] == TC_PAD_TAP;                        // padding tap of an odd tap count: zero weights
      const int o = pad ? 0 : toff[kb];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(hb + pixb[i] + o);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[kb]), __builtin_bit_cast(bf16x8, a), acc[i], 0, 0, 0);
      }
    }

    // ---- epilogue: D row = channel (e&3) + 8(e>>2) + 4lh, col = pixel lr: four consecutive channels per register
    //      quad -> one 8-byte staging write; then 16-byte reads and 64-byte-per-pixel global stores ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = q * 4 + k;
          const int co = cb * 32 + k + 8 * q + 4 * lh;
          v[k] = co < wrows ? apply_act(acc[i][e] + bv[e], act) : 0.f;
        }
        u32x2 pk;
        pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<u32x2*>(wst + (i * 32 + lr) * TC_SROW + (8 * q + 4 * lh) * 2) = pk;
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    {
      const int img = t / tiles_img;
      const int rem = t - img * tiles_img;
      const int tyi = rem / tiles_x;
      const int oy0 = tyi * TC_TH + 2 * rp, ox0 = (rem - tyi * tiles_x) * TC_TW;
      const int chunk = lane & 3;                                    // 16-byte chunk of the wave's 32 channels
      const bool ok = cb * 32 + chunk * 8 < ldc;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int pr = p * 16 + (lane >> 2);                         // pixel of the wave's 2 x 32 block
        const size_t opix = (size_t)out_pixel(g, img, oy0 + (pr >> 5), ox0 + (pr & 31));
        const u32x4 v = *reinterpret_cast<const u32x4*>(wst + pr * TC_SROW + chunk * 16);
        if (ok) *reinterpret_cast<u32x4*>(out + opix * ldc + cb * 32 + chunk * 8) = v;
      }
    }
    t = tn;
  }
}

template <int NKB>
static hipError_t launch_thin(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out, int ldc,
                              int act, int ntiles, int num_cu, hipStream_t st) {
  const int hwd = (TC_TW - 1) * g.sw + g.tw, hht = (TC_TH - 1) * g.sh + g.th;
  const int halo_bytes = ((hwd * hht + 63) / 64) * 1024;
  const size_t lds = 2 * (size_t)halo_bytes + 8 * 64 * TC_SROW;
  auto kern = thin_cin_conv_kernel<NKB>;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_set = lds;
  }
  count_launch(K_THIN_CIN);
  prof_begin(PROF_GATHER_GEMM, 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)wrows, st);
  hipLaunchKernelGGL(kern, dim3(std::min(ntiles, num_cu)), dim3(512), lds, st, g, (const bf16_t*)src, (const bf16_t*)wgt, wrows, bias,
                     (bf16_t*)out, ldc, act, ntiles, halo_bytes);
  prof_end(PROF_GATHER_GEMM, st);
  return hipGetLastError();
}

// returns hipErrorNotSupported when the shape does not qualify (the caller falls through to the gather GEMMs)
hipError_t thin_cin_conv(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out, int ldc,
                         int act, int num_cu, hipStream_t st) {
  if (g.Cs != 8 || wrows > TC_N || ldc > TC_N || ldc % 8 != 0 || g.wK != g.K || g.wtw != g.tw) return hipErrorNotSupported;
  if ((g.ys != 1 && g.ys != -1) || (g.xs != 1 && g.xs != -1) || g.sh < 1 || g.sh > 2 || g.sw != g.sh) return hipErrorNotSupported;
  if (g.Ho % TC_TH != 0 || g.Wo % TC_TW != 0 || g.M != g.N * g.Ho * g.Wo) return hipErrorNotSupported;
  const int ntiles = g.N * (g.Ho / TC_TH) * (g.Wo / TC_TW);
  if (ntiles < num_cu) return hipErrorNotSupported;
  const int nkb = (g.th * g.tw + 1) / 2;
  if (nkb == 25) return launch_thin<25>(g, src, wgt, wrows, bias, out, ldc, act, ntiles, num_cu, st);     // 7x7
  if (nkb == 8) return launch_thin<8>(g, src, wgt, wrows, bias, out, ldc, act, ntiles, num_cu, st);       // 4x4
  if (nkb == 5) return launch_thin<5>(g, src, wgt, wrows, bias, out, ldc, act, ntiles, num_cu, st);       // 3x3
  return hipErrorNotSupported;
}


// =====================================================================================================================
// Thin-OUTPUT convs (bf16, 64-channel source, <= 8 output channels): the generator heads (64 -> 4, 3x3) and the
// interior dgrad of the stem (dY 64 channels -> dX 3(8), 7x7).  The work is the 134 MB input stream plus, for the 7x7,
// 3 136 MACs per output value; on the gather GEMMs the 8-wide output wastes a 32-column tile and every 256-pixel tile
// pays its own pipeline fill (measured 130 us / 570 us).  Persistent workgroups walk 8 x 32 pixel tiles, ONE barrier per
// tile: the 64-channel input halo of tile t+1 ((8+th-1) x (32+tw-1) pixels x 128 B, double-buffered) arrives by LDS-DMA
// while tile t computes; the 8 weight rows live in LDS.  The MFMA runs transposed (A = weights, rows >= 8 read a zero
// row; B = pixels): D[channel][pixel] leaves each lane with 4 consecutive channels of its pixel in 4 registers, so the
// result goes straight to memory as 8-byte stores -- a wave writes 32 pixels x 16 B = 512 contiguous bytes.
// =====================================================================================================================
constexpr int TO_MAXCO = 8;

__global__ __launch_bounds__(512) void thin_cout_conv_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                             const bf16_t* __restrict__ wgt, const int wrows,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                             const int ldc, const int act, const int ntiles, const int halo_bytes,
                                                             const int nbuf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const halo = smem;                                  // [nbuf][halo_bytes], 128-byte pixel rows, XOR-swizzled
  unsigned char* const wl = smem + nbuf * halo_bytes;                   // [9 rows][K*2 B]: 8 weight rows + one zero row

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // tile row of this wave
  const int lr = lane & 31, lh = lane >> 5;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_thin);
  const int ntaps = g.th * g.tw;
  const int wrow_bytes = g.K * 2;

  for (int i = tid; i < (TO_MAXCO + 1) * (wrow_bytes >> 4); i += 512) {
    const int row = i / (wrow_bytes >> 4), ch = i - row * (wrow_bytes >> 4);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < wrows) v = *reinterpret_cast<const u32x4*>(wgt + (size_t)row * g.K + ch * 8);
    *reinterpret_cast<u32x4*>(wl + row * wrow_bytes + ch * 16) = v;
  }

  const int hwd = TC_TW + g.tw - 1, hht = TC_TH + g.th - 1;
  const int npix = hwd * hht;
  const int ngroups = (npix + 7) >> 3;                               // 8-pixel LDS-DMA groups
  const int tiles_x = g.Wo / TC_TW, tiles_y = g.Ho / TC_TH;
  const int tiles_img = tiles_x * tiles_y;

  // Per-lane constants of this wave's halo pieces (group wave + 8 i: pixel p = 8 grp + lane / 8), hoisted by hand: the halo row /
  // column of the pixel and its swizzled channel offset.  The address chain per piece was ~60 instructions (a division by the
  // halo width, two bounds checks, 64-bit products) against 36 MFMAs of compute per 3x3 tile: the issue phase, not the stream,
  // set the kernel's time (wgrad_halo.hip has the measurement of the same pattern).  Reflect-padded launches (the heads) issue a
  // piece as a wave-uniform image base + a 32-bit per-lane byte offset; zero-padded ones (the stem's interior dgrad) keep the
  // per-lane pointer with its zero-page select.
  constexpr int TO_MAXG = 9;                                           // 7x7: 67 groups -> 9 per wave
  int hyx[TO_MAXG], coff[TO_MAXG];
#pragma unroll
  for (int i = 0; i < TO_MAXG; ++i) {
    const int grp = wave + 8 * i;
    const int p = grp * 8 + (lane >> 3);
    const int hy = p / hwd, hx = p - hy * hwd;
    hyx[i] = (grp < ngroups && p < npix) ? ((hy << 16) | hx) : -1;
    coff[i] = ((lane & 7) ^ ((p >> 1) & 7)) << 3;
    asm volatile("" : "+v"(hyx[i]), "+v"(coff[i]));
  }
  const bool fast = g.pad_mode == PAD_REFLECT;
  const unsigned halo_lds = lds_addr_of(halo);
  auto issue_halo = [&](int buf, int t) {
    const int img = t / tiles_img;
    const int rem = t - img * tiles_img;
    const int ty = rem / tiles_x;
    const int hy0 = ty * TC_TH + g.by0 + (g.ys < 0 ? -(g.th - 1) : 0);
    const int hx0 = (rem - ty * tiles_x) * TC_TW + g.bx0 + (g.xs < 0 ? -(g.tw - 1) : 0);
    const bf16_t* ibase = src + (size_t)img * ((size_t)g.Hs * g.Ws * 64);                    // wave-uniform
#pragma unroll
    for (int i = 0; i < TO_MAXG; ++i) {
      const int grp = wave + 8 * i;
      if (grp >= ngroups) break;                                     // wave-uniform
      if (fast) {
        const int q = hyx[i] < 0 ? 0 : hyx[i];                       // padding slots of the last group: any valid pixel (never read)
        const int vy = hy0 + (q >> 16), vx = hx0 + (q & 0xffff);
        const int ay = max(vy, -vy), ax = max(vx, -vx);
        const int y = min(ay, 2 * g.Hl - 2 - ay), x = min(ax, 2 * g.Wl - 2 - ax);
        glds16_asm_s(ibase, (unsigned)((((y >> g.up) * g.Ws + (x >> g.up)) << 6) + coff[i]) * 2u, halo_lds + buf * halo_bytes + grp * 1024);
      } else {
        const bf16_t* ptr = zero;
        if (hyx[i] >= 0) {
          const int y = bound_coord(hy0 + (hyx[i] >> 16), g.Hl, g.pad_mode);
          const int x = bound_coord(hx0 + (hyx[i] & 0xffff), g.Wl, g.pad_mode);
          if ((y | x) >= 0) ptr = ibase + (unsigned)((((y >> g.up) * g.Ws + (x >> g.up)) << 6) + coff[i]);
        }
        glds16tc(ptr, halo + buf * halo_bytes + grp * 1024);
      }
    }
  };

  // weight fragment address of this lane: row lr (rows >= 8 -> the zero row), 8 channels at k = 16*kblock + 8*lh
  const unsigned char* const wbase = wl + (lr < TO_MAXCO ? lr : TO_MAXCO) * wrow_bytes + lh * 16;
  const int pix0 = (wave + (g.ys < 0 ? g.th - 1 : 0)) * hwd + lr + (g.xs < 0 ? g.tw - 1 : 0);
  const int step_x = g.xs > 0 ? 1 : -1, step_y = g.ys > 0 ? hwd : -hwd;
  float bv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) bv[k] = (bias != nullptr && 4 * lh + k < wrows) ? bias[4 * lh + k] : 0.f;

  int t = blockIdx.x;
  if (t < ntiles && nbuf == 2) issue_halo(0, t);
  for (int it = 0; t < ntiles; ++it) {
    const int tn = t + gridDim.x;
    const unsigned char* hb = halo;
    if (nbuf == 2) {                                                 // halo of tile t+1 streams in under tile t
      if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (one output store per tile stays in flight)
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      __syncthreads();
      if (tn < ntiles) issue_halo((it + 1) & 1, tn);
      hb = halo + (it & 1) * halo_bytes;
    } else {                                                         // 7x7: one 67 KB halo buffer next to 56 KB of weights
      __syncthreads();                                               // everyone is done with the previous tile
      issue_halo(0, t);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    int ty = 0, tx = 0;                                              // wave-uniform tap walk
    for (int tap = 0; tap < ntaps; ++tap) {
      const int pix = pix0 + ty * step_y + tx * step_x;
      const unsigned char* pa = hb + pix * 128;
      const int sw = (pix >> 1) & 7;
      const unsigned char* wa = wbase + tap * 128;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 w = *reinterpret_cast<const u32x4*>(wa + ks * 32);
        const u32x4 a = *reinterpret_cast<const u32x4*>(pa + (((ks * 2 + lh) ^ sw) << 4));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, a), acc, 0, 0, 0);
      }
      if (++tx == g.tw) { tx = 0; ++ty; }
    }

    // D row = channel (e&3) + 8(e>>2) + 4lh, col = pixel lr: registers 0..3 are channels 4lh .. 4lh+3 of pixel lr
    {
      const int img = t / tiles_img;
      const int rem = t - img * tiles_img;
      const int tyi = rem / tiles_x;
      const size_t opix = (size_t)out_pixel(g, img, tyi * TC_TH + wave, (rem - tyi * tiles_x) * TC_TW + lr);
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = 4 * lh + k < wrows ? apply_act(acc[k] + bv[k], act) : 0.f;
      u32x2 pk;
      pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
      pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
      *reinterpret_cast<u32x2*>(out + opix * ldc + 4 * lh) = pk;
    }
    t = tn;
  }
}

// returns hipErrorNotSupported when the shape does not qualify (the caller falls through to the gather GEMMs)
hipError_t thin_cout_conv(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out, int ldc,
                          int act, int num_cu, hipStream_t st) {
  if (g.Cs != 64 || wrows > TO_MAXCO || ldc != 8 || g.wK != g.K || g.wtw != g.tw) return hipErrorNotSupported;
  if ((g.ys != 1 && g.ys != -1) || (g.xs != 1 && g.xs != -1) || g.sh != 1 || g.sw != 1) return hipErrorNotSupported;
  if (g.Ho % TC_TH != 0 || g.Wo % TC_TW != 0 || g.M != g.N * g.Ho * g.Wo) return hipErrorNotSupported;
  if ((long long)g.N * g.Hs * g.Ws * 64 >= (1ll << 31)) return hipErrorNotSupported;
  const int ntiles = g.N * (g.Ho / TC_TH) * (g.Wo / TC_TW);
  if (ntiles < num_cu) return hipErrorNotSupported;
  const int hwd = TC_TW + g.tw - 1, hht = TC_TH + g.th - 1;
  const int halo_bytes = ((hwd * hht + 7) / 8) * 1024;
  const size_t wbytes = (size_t)(TO_MAXCO + 1) * g.K * 2;
  int nbuf = 2;
  if (2 * (size_t)halo_bytes + wbytes > 160 * 1024) nbuf = 1;
  const size_t lds = nbuf * (size_t)halo_bytes + wbytes;
  if (lds > 160 * 1024) return hipErrorNotSupported;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(thin_cout_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_set = lds;
  }
  count_launch(K_THIN_COUT);
  prof_begin(PROF_GATHER_GEMM, 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)wrows, st);
  hipLaunchKernelGGL(thin_cout_conv_kernel, dim3(std::min(ntiles, num_cu)), dim3(512), lds, st, g, (const bf16_t*)src,
                     (const bf16_t*)wgt, wrows, bias, (bf16_t*)out, ldc, act, ntiles, halo_bytes, nbuf);
  prof_end(PROF_GATHER_GEMM, st);
  return hipGetLastError();
}

}  // namespace dei2i
