// wgrad v2 (bf16 hot path): dw[co][k] = sum over pixels m of dy[m][co] * gather(x)[m][k]
//
// 256(co) x 128(k) output tiles, 8 wavefronts, both operands streamed pixel-major into a 3-stage LDS ring with LDS-DMA
// (no VGPR staging), fragments fetched with ds_read_b64_tr_b16 (hardware transpose: the reduction index -- the pixel --
// is the slow index of both operands).  The pixel range is split over blockIdx.z; each split writes its fp32 partial
// tile to its own slab with plain 128-byte-coalesced stores and a second kernel sums the slabs and un-packs to the
// OIHW gradient in one pass -- no float atomics (chip-wide atomic rate is ~1.3 TB/s, plain stores ~6 TB/s) and the
// result is bitwise reproducible.
//
// Requirements (launcher-checked, everything else runs on v1): bf16, Cs % 128 == 0 (a 128-column k-tile lies inside one
// tap), CoutS % 8 == 0.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "common.h"
#include "geom.h"
#include "launch.h"

namespace dei2i {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_w[256];   // zero-initialised device memory

typedef __attribute__((address_space(3))) void lds_void_t2;
typedef __attribute__((address_space(1))) const void gbl_void_t2;

DEI2I_D void glds16w(const void* gptr, unsigned char* lds_wave_base) {
  glds16_asm(gptr, lds_wave_base);      // (common.h: hipcc must not see the LDS write, or it drains the ring)
}

DEI2I_D int xcd_remap3(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// LDS image per stage: A = dy tile [64 px][512 B] (256 co), B = x tile [64 px][256 B] (128 k columns).
// tr-read swizzle: byte offset within the row ^= (row & 3) << 6 (both row lengths are multiples of 256 B).
template <int BM>
__global__ __launch_bounds__(512) void wgrad_v2_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                       const bf16_t* __restrict__ dy, const int co_rows, const int ldy,
                                                       float* __restrict__ slabs, const int tiles_k,
                                                       const int chunks_per_split, const long long slab_elems) {
  constexpr int BN = 128, BR = 64, STAGES = 3;
  constexpr int ROWB_A = BM * 2, ROWB_B = BN * 2;
  constexpr int RPI_A = 1024 / ROWB_A, SPR_A = ROWB_A / 16;      // rows per DMA instruction, 16-byte slots per row
  constexpr int A_BYTES = BR * ROWB_A, B_BYTES = BR * ROWB_B, STAGE_BYTES = A_BYTES + B_BYTES;   // 32 KB + 16 KB
  constexpr int LA = A_BYTES / 1024 / 8, LB = B_BYTES / 1024 / 8;                                  // 4, 2 DMA per wave
  constexpr int WTM = BM / 4, WTN = 64, TM = WTM / 32, TN = 2;                                     // 4 x 2 waves
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap3(blockIdx.x, gridDim.x);
  const int tile_k = bid % tiles_k, tile_c = bid / tiles_k;
  const int c0 = tile_c * BM, k0 = tile_k * BN;

  const int nchunks = (g.M + BR - 1) / BR;
  const int cbeg = blockIdx.z * chunks_per_split;
  const int cend = min(nchunks, cbeg + chunks_per_split);
  const int nc = cend - cbeg;

  // k-tile -> (tap, channel offset): uniform for the workgroup
  int tap, ci0;
  decode_k(g, k0, tap, ci0);
  const int ty = (int)fd_div((uint32_t)tap, g.fd_tw);
  const int dyy = ty * g.ys, dxx = (tap - ty * g.tw) * g.xs;
  const bool k_ok = k0 < g.K;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_w);

  // DMA roles.  A: one instruction = RPI_A rows x ROWB_A bytes (2 x 512 B or 4 x 256 B).
  //             B: one instruction = 4 rows x 256 B -> lane l: row (l>>4), slot (l&15).
  auto issue = [&](int stage, int chunk) {
    unsigned char* sbase = smem + stage * STAGE_BYTES;
    const int mb = chunk * BR;
#pragma unroll
    for (int j = 0; j < LA; ++j) {
      const int rgrp = (j * 8 + wave) * RPI_A;               // first row of this instruction's row group
      const int r = rgrp + lane / SPR_A;
      const int off = ((lane % SPR_A) * 16) ^ ((r & 3) << 6);  // source byte offset that belongs at this LDS slot
      const int m = mb + r;
      const int c = c0 + (off >> 1);
      const bf16_t* p = (m < g.M && c < ldy) ? dy + (size_t)m * ldy + c : zero;
      glds16w(p, sbase + rgrp * ROWB_A);
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const int rgrp = j * 32 + wave * 4;
      const int r = rgrp + (lane >> 4);
      const int off = ((lane & 15) * 16) ^ ((r & 3) << 6);
      const int m = mb + r;
      const bf16_t* p = zero;
      if (k_ok && m < g.M) {
        int n, oy, ox;
        decode_m(g, m, n, oy, ox);
        const int y = bound_coord(oy * g.sh + g.by0 + dyy, g.Hl, g.pad_mode);
        const int x = bound_coord(ox * g.sw + g.bx0 + dxx, g.Wl, g.pad_mode);
        if ((y | x) >= 0) p = src + ((size_t)((n * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up))) * g.Cs + ci0 + (off >> 1);
      }
      glds16w(p, sbase + A_BYTES + rgrp * ROWB_B);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_cohalf = (lane >> 4) & 1;

  auto compute = [&](int stage) {
    const unsigned char* ab = smem + stage * STAGE_BYTES;
    const unsigned char* bb = ab + A_BYTES;
    u32x4 af[2][TM], bf[2][TN];
    auto load_frags = [&](int kk, u32x4 (&a)[TM], u32x4 (&b)[TN]) {
      const int row_lo = kk * 16 + 8 * lh + tr_q;
      const int s0 = (row_lo & 3) << 6, s1 = ((row_lo + 4) & 3) << 6;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int colb = (wm * WTM + i * 32 + 16 * tr_cohalf + 4 * tr_p) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ab + row_lo * ROWB_A + (colb ^ s0)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ab + (row_lo + 4) * ROWB_A + (colb ^ s1)));
        u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        a[i].x = l2.x; a[i].y = l2.y; a[i].z = h2.x; a[i].w = h2.y;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int colb = (wn * WTN + j * 32 + 16 * tr_cohalf + 4 * tr_p) * 2;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(bb + row_lo * ROWB_B + (colb ^ s0)));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(bb + (row_lo + 4) * ROWB_B + (colb ^ s1)));
        u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
        b[j].x = l2.x; b[j].y = l2.y; b[j].z = h2.x; b[j].w = h2.y;
      }
    };
    load_frags(0, af[0], bf[0]);
#pragma unroll
    for (int kk = 0; kk < BR / 16; ++kk) {
      if (kk + 1 < BR / 16) load_frags(kk + 1, af[(kk + 1) & 1], bf[(kk + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[kk & 1][i]),
                                                               __builtin_bit_cast(bf16x8, bf[kk & 1][j]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (nc > 0) {
    issue(0, cbeg);
    if (nc > 1) issue(1, cbeg + 1);
    for (int it = 0; it < nc; ++it) {
      if (it + 1 < nc) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA + LB) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (it + 2 < nc) issue((it + 2) % STAGES, cbeg + it + 2);
      compute(it % STAGES);
    }
  }

  // partial tile -> this split's slab [co_rows][K] with plain stores (lanes run along k: 128-byte segments)
  float* slab = slabs + (size_t)blockIdx.z * slab_elems;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = c0 + wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      if (co >= co_rows) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int k = k0 + wn * WTN + j * 32 + lr;
        if (k < g.K) slab[(size_t)co * g.K + k] = acc[i][j][e];
      }
    }
}

// dw_oihw[co][ci][t] = sum_s slab[s][co][t][ci]  (slab reduce fused with the un-pack to the reference's OIHW layout)
// One workgroup per (co, 64-channel block): the taps x 64 block is read tap-major (ci fastest: every slab read is a
// coalesced 256-byte row, four slabs in flight), transposed through LDS and written ci-major -- the (ci0.., all taps)
// range of one co is CONTIGUOUS in OIHW, so the writes are coalesced too.  Slabs are added in index order: deterministic.
constexpr int RU_CB = 64;            // channels per block
constexpr int RU_MAX_TAPS = 64;      // 8x8 (cls_clf at 512x512) is the largest kernel of the path
__global__ __launch_bounds__(256) void wgrad_reduce_unpack_kernel(const float* __restrict__ slabs, int nsplit, long long slab_elems,
                                                                  float* __restrict__ dw, int Cout, int Cin, int CinS, int taps,
                                                                  int accumulate) {
  extern __shared__ float ru_lds[];                 // [taps][RU_CB + 1]
  const int co = blockIdx.x, ci0 = blockIdx.y * RU_CB;
  const int cb = min(RU_CB, CinS - ci0);
  const float* base = slabs + ((size_t)co * taps) * CinS + ci0;
  for (int i = threadIdx.x; i < taps * RU_CB; i += 256) {
    const int t = i / RU_CB, cl = i - t * RU_CB;
    if (cl >= cb) continue;
    const float* p = base + (size_t)t * CinS + cl;
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = 0.f;
    int s = 0;
    for (; s + 7 < nsplit; s += 8) {                // eight slab reads in flight
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] += p[(size_t)(s + u) * slab_elems];
    }
    for (; s < nsplit; ++s) v[s & 7] += p[(size_t)s * slab_elems];
    ru_lds[t * (RU_CB + 1) + cl] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  __syncthreads();
  const int live = min(cb, Cin - ci0);              // padded channels (ci >= Cin) are not part of the OIHW tensor
  float* o = dw + ((size_t)co * Cin + ci0) * taps;
  for (int j = threadIdx.x; j < live * taps; j += 256) {
    const int cl = j / taps, t = j - cl * taps;
    const float v = ru_lds[t * (RU_CB + 1) + cl];
    o[j] = accumulate ? o[j] + v : v;
  }
}

hipError_t wgrad_v2(const GatherDesc& g, const void* src, const void* dy, int co_rows, int ldy, float* slabs,
                    size_t slab_capacity_elems, int num_cu, int* nsplit_out, hipStream_t st) {
  if (g.Cs % 128 != 0 || co_rows < 96 || g.M < 4096) return hipErrorNotSupported;
  const int BM = co_rows >= 192 ? 256 : 128;
  constexpr int BN = 128, BR = 64;
  const int tiles_c = (co_rows + BM - 1) / BM, tiles_k = (g.K + BN - 1) / BN;
  const int tiles = tiles_c * tiles_k;
  const int nchunks = (g.M + BR - 1) / BR;
  int splits = std::max(1, num_cu / tiles);                  // one workgroup per CU (144 KB LDS), a single round
  if (splits > nchunks / 8) splits = std::max(1, nchunks / 8);
  const long long slab_elems = wgrad_slab_elems(co_rows, g.K);
  if ((size_t)slab_elems * splits > slab_capacity_elems) splits = (int)(slab_capacity_elems / (size_t)slab_elems);
  if (splits < 1) return hipErrorNotSupported;
  const int cps = (nchunks + splits - 1) / splits;
  const int zs = (nchunks + cps - 1) / cps;
  count_launch(K_WGRAD_V2);
  prof_begin(PROF_WGRAD, 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)co_rows, st);
  if (BM == 256) {
    const size_t lds = 3 * (size_t)(64 * 512 + 64 * 256);
    auto kern = wgrad_v2_kernel<256>;
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles, 1, zs), dim3(512), lds, st, g, (const bf16_t*)src, (const bf16_t*)dy, co_rows, ldy,
                       slabs, tiles_k, cps, slab_elems);
  } else {
    const size_t lds = 3 * (size_t)(64 * 256 + 64 * 256);
    auto kern = wgrad_v2_kernel<128>;
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      attr_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles, 1, zs), dim3(512), lds, st, g, (const bf16_t*)src, (const bf16_t*)dy, co_rows, ldy,
                       slabs, tiles_k, cps, slab_elems);
  }
  prof_end(PROF_WGRAD, st);
  *nsplit_out = zs;
  return hipGetLastError();
}

// first level of a two-level slab sum, used when the (co, channel-block) grid alone cannot fill the chip (thin layers
// with many slabs): slab z*k += slabs z*k+1 .. z*k+k-1, elementwise and in place, in index order
__global__ void slab_group_sum_kernel(float* __restrict__ slabs, int nsplit, long long slab_elems, int k, int groups) {
  const size_t n4 = (size_t)slab_elems / 4;
  const size_t total = n4 * groups;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int z = (int)(i / n4);
    const size_t e = (i - (size_t)z * n4) * 4;
    const int s0 = z * k, s1 = min(nsplit, s0 + k);
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0, v2 = v0, v3 = v0;
    int sidx = s0;
    for (; sidx + 3 < s1; sidx += 4) {
      v0 += *reinterpret_cast<const f32x4*>(slabs + (size_t)sidx * slab_elems + e);
      v1 += *reinterpret_cast<const f32x4*>(slabs + (size_t)(sidx + 1) * slab_elems + e);
      v2 += *reinterpret_cast<const f32x4*>(slabs + (size_t)(sidx + 2) * slab_elems + e);
      v3 += *reinterpret_cast<const f32x4*>(slabs + (size_t)(sidx + 3) * slab_elems + e);
    }
    for (; sidx < s1; ++sidx) v0 += *reinterpret_cast<const f32x4*>(slabs + (size_t)sidx * slab_elems + e);
    *reinterpret_cast<f32x4*>(slabs + (size_t)s0 * slab_elems + e) = (v0 + v1) + (v2 + v3);
  }
}

hipError_t wgrad_reduce_unpack(float* slabs, int nsplit, long long slab_elems, float* dw, int Cout, int Cin, int CinS,
                               int taps, int accumulate, hipStream_t st) {
  if (taps > RU_MAX_TAPS || Cout <= 0 || CinS <= 0) return hipErrorInvalidValue;
  const int cblocks = (CinS + RU_CB - 1) / RU_CB;
  long long stride = slab_elems;
  if (Cout * cblocks < 512 && nsplit > 8 && slab_elems % 4 == 0) {
    const int k = (nsplit + 7) / 8, groups = (nsplit + k - 1) / k;
    const size_t total = (size_t)slab_elems / 4 * groups;
    hipLaunchKernelGGL(slab_group_sum_kernel, dim3(grid_for(total, 256, 256u * 8u)), dim3(256), 0, st, slabs, nsplit, slab_elems,
                       k, groups);
    nsplit = groups;
    stride = slab_elems * k;
  }
  const size_t lds = (size_t)taps * (RU_CB + 1) * sizeof(float);
  hipLaunchKernelGGL(wgrad_reduce_unpack_kernel, dim3(Cout, cblocks), dim3(256), lds, st, (const float*)slabs, nsplit, stride, dw,
                     Cout, Cin, CinS, taps, accumulate);
  return hipGetLastError();
}

}  // namespace dei2i
