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
constexpr int TO_MAXG = 9;                         // LDS-DMA pieces per wave and tile (7x7: 34 groups x 2 planes = 68 -> 9 per wave)

DEI2I_D int sw_to(int row) { return ((row >> 2) & 1) << 1; }

// Round 3 form: 16x16x32 MFMAs (A = 16 weight rows of which <= 8 are live, B = 16 pixels x 32 channels).  The 32x32x16 form this
// replaces spent a 32-row A operand on 8 live rows (75 % of the matrix work wasted: the 7x7 input gradient of the stem ran 196
// MFMAs of 32 cycles per 32 pixels) and re-read the weight fragment from LDS for every MFMA in every wave -- the 3x3 heads were
// LDS-bound at twice their MFMA time.  Here a tap of a 32-pixel tile row is 4 MFMAs of 16 cycles on 4 fragment reads, the <= 9-tap
// weights (18 fragments) live in registers for the whole launch (WREG), the 49-tap ones stream one fragment per (tap, k-block).
//   halo in LDS : two PLANES (channels 0..31 | 32..63) of [pixel][64 B] rows, 16-byte chunk c of row r at slot c ^ (((r >> 2) & 1)
//                 << 1) -- conv_halo16.hip's image: a ds_read_b128 of 16 consecutive pixels x 4 k-groups is conflict-free at any
//                 tap shift; one LDS-DMA piece = 16 pixels of one plane
//   D           : lane (px = l16, kg) holds channels 4kg .. 4kg+3 of its pixel: lanes kg < 2 store 8 bytes each
template <bool WREG>
__global__ __launch_bounds__(512) void thin_cout_conv_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                             const bf16_t* __restrict__ wgt, const int wrows,
                                                             const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                             const int ldc, const int act, const int ntiles, const int halo_bytes,
                                                             const int nbuf) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const halo = smem;                                  // [nbuf][2 planes][plane_bytes]
  unsigned char* const wl = smem + nbuf * halo_bytes;                // !WREG: [8 rows][K*2 B] weights

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // tile row of this wave
  const int l16 = lane & 15, kg = lane >> 4;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_thin);
  const int ntaps = g.th * g.tw;
  const int wrow_bytes = g.K * 2;
  const int plane_bytes = halo_bytes >> 1;

  if constexpr (!WREG) {
    for (int i = tid; i < TO_MAXCO * (wrow_bytes >> 4); i += 512) {
      const int row = i / (wrow_bytes >> 4), ch = i - row * (wrow_bytes >> 4);
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < wrows) v = *reinterpret_cast<const u32x4*>(wgt + (size_t)row * g.K + ch * 8);
      *reinterpret_cast<u32x4*>(wl + row * wrow_bytes + ch * 16) = v;
    }
  }

  const int hwd = TC_TW + g.tw - 1, hht = TC_TH + g.th - 1;
  const int npix = hwd * hht;
  const int ngroups = (npix + 15) >> 4;                              // 16-pixel LDS-DMA groups per plane
  const int npieces = 2 * ngroups;
  const int tiles_x = g.Wo / TC_TW, tiles_y = g.Ho / TC_TH;
  const int tiles_img = tiles_x * tiles_y;

  // Per-lane constants of this wave's halo pieces (piece q = wave + 8 i: plane q & 1, group q >> 1, pixel 16 grp + lane / 4), hoisted:
  // the halo row / column of the pixel and its swizzled channel offset (the address chain per piece was the issue phase's time).
  int hyx[TO_MAXG], coff[TO_MAXG];
#pragma unroll
  for (int i = 0; i < TO_MAXG; ++i) {
    const int q = wave + 8 * i;
    const int grp = q >> 1, p = grp * 16 + (lane >> 2);
    const int hy = p / hwd, hx = p - hy * hwd;
    hyx[i] = (q < npieces && p < npix) ? ((hy << 16) | hx) : -1;
    coff[i] = (q & 1) * 32 + (((lane & 3) ^ sw_to(p)) << 3);           // channel offset of this lane's 16-byte chunk
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
      const int q = wave + 8 * i;
      if (q >= npieces) break;                                       // wave-uniform
      const unsigned dst = (unsigned)(buf * halo_bytes + (q & 1) * plane_bytes + (q >> 1) * 1024);
      if (fast) {
        const int hq = hyx[i] < 0 ? 0 : hyx[i];                      // padding rows of the last group: any valid pixel (never read)
        const int vy = hy0 + (hq >> 16), vx = hx0 + (hq & 0xffff);
        const int ay = max(vy, -vy), ax = max(vx, -vx);
        const int y = min(ay, 2 * g.Hl - 2 - ay), x = min(ax, 2 * g.Wl - 2 - ax);
        glds16_asm_s(ibase, (unsigned)((((y >> g.up) * g.Ws + (x >> g.up)) << 6) + coff[i]) * 2u, halo_lds + dst);
      } else {
        const bf16_t* ptr = zero;
        if (hyx[i] >= 0) {
          const int y = bound_coord(hy0 + (hyx[i] >> 16), g.Hl, g.pad_mode);
          const int x = bound_coord(hx0 + (hyx[i] & 0xffff), g.Wl, g.pad_mode);
          if ((y | x) >= 0) ptr = ibase + (unsigned)((((y >> g.up) * g.Ws + (x >> g.up)) << 6) + coff[i]);
        }
        glds16tc(ptr, halo + dst);
      }
    }
  };

  // weight fragments: A row = output channel l16 (rows >= 8: zeros), k = 32 (kb) + 8 kg .. +7 of tap t
  constexpr int NWF = WREG ? 18 : 1;
  u32x4 wf[NWF];
  if constexpr (WREG) {
#pragma unroll
    for (int f = 0; f < 18; ++f) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (l16 < wrows && (f >> 1) < ntaps) v = *reinterpret_cast<const u32x4*>(wgt + (size_t)l16 * g.K + (f >> 1) * 64 + (f & 1) * 32 + kg * 8);
      wf[f] = v;
    }
  }
  const unsigned char* const wbase = wl + (l16 < TO_MAXCO ? l16 : 0) * wrow_bytes + kg * 16;
  const bool wlive = l16 < TO_MAXCO;
  // byte address (plane 0, before the swizzle) of this lane's chunk of pixel (tile row `wave`, column l16) at tap (0,0); columns
  // 16..31 are 16 rows = 1024 bytes further (same swizzle bit)
  const int pix0 = (wave + (g.ys < 0 ? g.th - 1 : 0)) * hwd + l16 + (g.xs < 0 ? g.tw - 1 : 0);
  const int step_x = g.xs > 0 ? 1 : -1, step_y = g.ys > 0 ? hwd : -hwd;
  float bv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) bv[k] = (bias != nullptr && 4 * kg + k < wrows) ? bias[4 * kg + k] : 0.f;

  int t = blockIdx.x;
  if (t < ntiles && nbuf == 2) issue_halo(0, t);
  for (int it = 0; t < ntiles; ++it) {
    const int tn = t + gridDim.x;
    const unsigned char* hb = halo;
    if (nbuf == 2) {                                                 // halo of tile t+1 streams in under tile t
      if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (two output stores per tile stay in flight)
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      __syncthreads();
      if (tn < ntiles) issue_halo((it + 1) & 1, tn);
      hb = halo + (it & 1) * halo_bytes;
    } else {                                                         // 7x7: one 68 KB halo buffer next to 50 KB of weights
      __syncthreads();                                               // everyone is done with the previous tile
      issue_halo(0, t);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }

    typedef __attribute__((ext_vector_type(4))) float acc4_t;
    acc4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto tap_mfma = [&](int pix, const u32x4& w0, const u32x4& w1) {
      const int ad = pix * 64 + ((kg ^ sw_to(pix)) << 4);             // columns 0..15 of the tile row; 16..31 at +1024 (pix + 16: same swizzle bit)
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(hb + ad + pb * 1024);
        const u32x4 a1 = *reinterpret_cast<const u32x4*>(hb + plane_bytes + ad + pb * 1024);
        acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), __builtin_bit_cast(bf16x8, a0), acc[pb], 0, 0, 0);
        acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), __builtin_bit_cast(bf16x8, a1), acc[pb], 0, 0, 0);
      }
    };
    if constexpr (WREG) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap < ntaps) {
          const int ty = tap / 3, tx = tap - 3 * ty;                  // (WREG launches are 3x3)
          tap_mfma(pix0 + ty * step_y + tx * step_x, wf[2 * tap], wf[2 * tap + 1]);
        }
      }
    } else {
      int ty = 0, tx = 0;                                            // wave-uniform tap walk
      for (int tap = 0; tap < ntaps; ++tap) {
        u32x4 w0 = {0u, 0u, 0u, 0u}, w1 = {0u, 0u, 0u, 0u};
        if (wlive) {
          w0 = *reinterpret_cast<const u32x4*>(wbase + tap * 128);
          w1 = *reinterpret_cast<const u32x4*>(wbase + tap * 128 + 64);
        }
        tap_mfma(pix0 + ty * step_y + tx * step_x, w0, w1);
        if (++tx == g.tw) { tx = 0; ++ty; }
      }
    }

    // D row = channel 4 kg + e, col = pixel l16: lanes kg < 2 hold the 8 output channels
    if (kg < 2) {
      const int img = t / tiles_img;
      const int rem = t - img * tiles_img;
      const int tyi = rem / tiles_x;
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const size_t opix = (size_t)out_pixel(g, img, tyi * TC_TH + wave, (rem - tyi * tiles_x) * TC_TW + pb * 16 + l16);
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = 4 * kg + k < wrows ? apply_act(acc[pb][k] + bv[k], act) : 0.f;
        u32x2 pk;
        pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<u32x2*>(out + opix * ldc + 4 * kg) = pk;
      }
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
  const int halo_bytes = 2 * ((hwd * hht + 15) / 16) * 1024;          // two channel planes of 16-pixel groups
  if (2 * ((hwd * hht + 15) / 16) > 8 * TO_MAXG) return hipErrorNotSupported;
  const bool wreg = g.th == 3 && g.tw == 3;                            // 18 weight fragments in registers
  const size_t wbytes = wreg ? 0 : (size_t)TO_MAXCO * g.K * 2;
  int nbuf = 2;
  if (2 * (size_t)halo_bytes + wbytes > 160 * 1024) nbuf = 1;
  const size_t lds = nbuf * (size_t)halo_bytes + wbytes;
  if (lds > 160 * 1024) return hipErrorNotSupported;
  auto kern = wreg ? thin_cout_conv_kernel<true> : thin_cout_conv_kernel<false>;
  static size_t lds_set[2] = {0, 0};
  if (lds > lds_set[wreg ? 1 : 0]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    lds_set[wreg ? 1 : 0] = lds;
  }
  count_launch(K_THIN_COUT);
  prof_begin(PROF_GATHER_GEMM, 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)wrows, st);
  hipLaunchKernelGGL(kern, dim3(std::min(ntiles, num_cu)), dim3(512), lds, st, g, (const bf16_t*)src,
                     (const bf16_t*)wgt, wrows, bias, (bf16_t*)out, ldc, act, ntiles, halo_bytes, nbuf);
  prof_end(PROF_GATHER_GEMM, st);
  return hipGetLastError();
}

}  // namespace dei2i
