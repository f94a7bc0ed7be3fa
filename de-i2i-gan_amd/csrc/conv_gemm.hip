// Implicit-GEMM convolution kernels for gfx950 (CDNA4): forward / dgrad ("gather GEMM") and wgrad.
//
// Layout: activations NHWC with the channel count padded to a 16-byte vector (8 bf16 / 4 f32); packed
// weights [rows][K] with K = taps*Cs contiguous, so BOTH GEMM operands of the gather GEMM are K-contiguous
// and are staged as [row][128 B] LDS tiles with an XOR swizzle that makes the ds_read_b128 fragment reads
// conflict-free.  64-wide wavefronts, 4 waves per workgroup, 32x32 MFMA blocks:
//   bf16: v_mfma_f32_32x32x16_bf16 (one 16-byte fragment per lane = 8 k-values)
//   f32 : v_mfma_f32_32x32x2_f32, four per 16-byte fragment (exact f32 FMA chain; parity mode)
// The k order inside a 16-byte fragment pair is the same permutation for A and B, so the sum is unchanged.
//
// wgrad reduces over pixels, which is the slow index of both operands; the tiles are staged pixel-major
// (coalesced 16-byte loads) and the bf16 fragments are fetched with ds_read_b64_tr_b16 (hardware transpose),
// the f32 fragments with ds_read_b32.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>

#include "common.h"
#include "geom.h"
#include "launch.h"

namespace dei2i {

// ------------------------------------------------------------------------------------------------
// MFMA wrappers: one call consumes a 16-byte A fragment and a 16-byte B fragment per lane.
// ------------------------------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static DEI2I_D void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static DEI2I_D void run(const u32x4& a, const u32x4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

DEI2I_D u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
DEI2I_D u32x4 zero16() { u32x4 z = {0u, 0u, 0u, 0u}; return z; }

// bijective XCD-aware remap of the linear workgroup id (cdna_hip_programming.md T1)
DEI2I_D int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// ------------------------------------------------------------------------------------------------
// gather GEMM:  out[m][n] = act( sum_k  gather(src)[m][k] * wgt[n][k] + bias[n] )
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gather_gemm_kernel(const DescPack pack, const T* __restrict__ src,
                                                          const T* __restrict__ wgt_base, const int wrows,
                                                          const float* __restrict__ bias, T* __restrict__ out,
                                                          float* __restrict__ ws, const int ldc, const int act,
                                                          const int tiles_n, const int ksteps_per_split, const int slab_rows) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr int BKE = 128 / (int)sizeof(T);      // k elements per 128-byte LDS row
  constexpr int RA = BM / 32, RB = BN / 32;      // 16-byte vectors per thread per k-step
  constexpr int WTM = BM / WM, WTN = BN / WN;    // wave tile
  constexpr int TM = WTM / 32, TN = WTN / 32;    // 32x32 MFMA blocks per wave
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 block");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TILE_BYTES = (BM + BN) * 128;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const GatherDesc& g = pack.d[blockIdx.y];
  const T* __restrict__ wgt = wgt_base + pack.woff[blockIdx.y];
  const int nwg = gridDim.x;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tile_n = bid % tiles_n;
  const int tile_m = bid / tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  if (m0 >= g.M) return;

  const int nk_total = (g.K + BKE - 1) / BKE;
  const int kbeg = blockIdx.z * ksteps_per_split;
  const int kend = min(nk_total, kbeg + ksteps_per_split);
  const bool empty = kbeg >= kend;        // a split-K slice past this class's K: it still owns (zero) slab rows
  if (empty && ws == nullptr) return;

  // ---- per-thread load assignment ----
  const int kv = tid & 7;       // 16-byte chunk within the 128-byte k-row
  const int r0 = tid >> 3;      // 0..31
  int a_nb[RA], a_by[RA], a_bx[RA];
#pragma unroll
  for (int i = 0; i < RA; ++i) {
    const int m = m0 + r0 + 32 * i;
    if (m < g.M) {
      int n, oy, ox;
      decode_m(g, m, n, oy, ox);
      a_nb[i] = n;
      a_by[i] = oy * g.sh + g.by0;
      a_bx[i] = ox * g.sw + g.bx0;
    } else {
      a_nb[i] = -1; a_by[i] = 0; a_bx[i] = 0;
    }
  }
  int b_row[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int n = n0 + r0 + 32 * j;
    b_row[j] = n < wrows ? n : -1;
  }

  u32x4 areg[RA], breg[RB];
  int cur_tap = -1, b_woff = 0;
  int a_pix[RA];

  // Dead-tap skipping (reflect-ring launches: 2 of 3 tap rows / columns read outside dY for every row of a tile).
  // Needs every thread of the workgroup on the same tap within a k-step: channel stride a multiple of the k-step.
  const bool skip_dead = pack.skip_dead_taps != 0 && (g.Cs % BKE) == 0;
  bool tap_live = true;          // liveness of the tap of the most recently loaded k-step (workgroup-uniform)
  auto load_tile = [&](int kstep) {
    const int k = kstep * BKE + kv * VEC;
    const bool kval = k < g.K;
    int tap, ci;
    decode_k(g, k, tap, ci);
    if (tap != cur_tap) {
      cur_tap = tap;
      const int ty = (int)fd_div((uint32_t)tap, g.fd_tw);
      const int tx = tap - ty * g.tw;
      const int dy = ty * g.ys, dx = tx * g.xs;
      b_woff = weight_tap_offset(g, ty, tx);
      int any = 0;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        const int y = bound_coord(a_by[i] + dy, g.Hl, g.pad_mode);
        const int x = bound_coord(a_bx[i] + dx, g.Wl, g.pad_mode);
        const int pix = (a_nb[i] * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up);
        a_pix[i] = ((y | x | a_nb[i]) < 0) ? -1 : pix;
        any |= a_pix[i] >= 0;
      }
      if (skip_dead) tap_live = __syncthreads_or(any) != 0;
    }
    if (!tap_live) return;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      areg[i] = (kval && a_pix[i] >= 0) ? ld16(src + (size_t)a_pix[i] * g.Cs + ci) : zero16();
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      breg[j] = (kval && b_row[j] >= 0) ? ld16(wgt + (size_t)b_row[j] * g.wK + b_woff + ci) : zero16();
    }
  };

  auto store_tile = [&](int buf) {
    unsigned char* base = smem + buf * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int row = r0 + 32 * i;
      *reinterpret_cast<u32x4*>(base + row * 128 + ((kv ^ ((row >> 1) & 7)) << 4)) = areg[i];
    }
    unsigned char* bb = base + BM * 128;
#pragma unroll
    for (int j = 0; j < RB; ++j) {
      const int row = r0 + 32 * j;
      *reinterpret_cast<u32x4*>(bb + row * 128 + ((kv ^ ((row >> 1) & 7)) << 4)) = breg[j];
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

  // fragment reads software-pipelined against the MFMAs through two register sets (see conv_gemm_v2.hip)
  auto compute_tile = [&](int buf) {
    const unsigned char* ab = smem + buf * TILE_BYTES;
    const unsigned char* bb = ab + BM * 128;
    u32x4 af[2][TM], bf[2][TN];
    auto load_frags = [&](int ks, u32x4 (&a)[TM], u32x4 (&b)[TN]) {
      const int chunk = ks * 2 + lh;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * WTM + i * 32 + lr;
        a[i] = *reinterpret_cast<const u32x4*>(ab + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * WTN + j * 32 + lr;
        b[j] = *reinterpret_cast<const u32x4*>(bb + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
      }
    };
    load_frags(0, af[0], bf[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) load_frags(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(af[ks & 1][i], bf[ks & 1][j], acc[i][j]);
    }
  };

  // ---- main loop: register-staged, double-buffered LDS, one barrier per k-step ----
  // Rotated so that load / compute / store each appear ONCE in the instruction stream: a dispatch starts with a cold
  // instruction cache and walks its code at ~0.4 us per KB, which is the whole run time of the small split-K launches
  // (SPADE class tables, reflect rings) -- measured 19 us for the 43 KB unrolled version whatever the problem size.
  if (!empty) {
    bool live_cur = false;
    int buf = 1;                            // buffer of the tile to compute; the next tile is stored to buf ^ 1
#pragma unroll 1
    for (int ks = kbeg; ks <= kend; ++ks) {
      const bool more = ks < kend;
      if (more) load_tile(ks);              // global loads in flight under the MFMAs
      const bool live_next = more && tap_live;
      if (live_cur) compute_tile(buf);
      if (live_next) store_tile(buf ^ 1);
      __syncthreads();
      buf ^= 1;
      live_cur = live_next;
    }
  }

  // ---- epilogue: D[i][j], j = lane&31 (output channel), i = (e&3) + 8*(e>>2) + 4*(lane>>5) (pixel) ----
  // staged through LDS as fp32 [BM][BN + 4]; a rolled loop then moves 4-channel vectors (coalesced 8 / 16-byte stores)
  float* __restrict__ stg = reinterpret_cast<float*>(smem);
  constexpr int SROW = BN + 4;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
#pragma unroll
      for (int j = 0; j < TN; ++j) stg[row * SROW + wn * WTN + j * 32 + lr] = acc[i][j][e];
    }
  __syncthreads();
  constexpr int NV = BN / 4;
#pragma unroll 1
  for (int v = tid; v < BM * NV; v += 256) {
    const int row = v / NV, n = n0 + (v - row * NV) * 4;
    const int m = m0 + row;
    if (m >= g.M || n >= ldc) continue;
    const f32x4 q = *reinterpret_cast<const f32x4*>(stg + row * SROW + (n - n0));
    if (ws != nullptr) {         // split-K: this slice's fp32 slab, rows indexed by (class base + m); plain stores
      const size_t wrow = (size_t)(pack.m_base[blockIdx.y] + m);
      if (pack.ws_atomic) {
        const float qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < wrows) atomicAdd(ws + wrow * ldc + n + e, qq[e]);
      } else {
        // columns >= wrows are never read by the finalize kernel: the full vector is stored
        *reinterpret_cast<f32x4*>(ws + ((size_t)blockIdx.z * slab_rows + wrow) * ldc + n) = q;
      }
      continue;
    }
    size_t opix;
    if (g.out_identity) {
      opix = (size_t)m;
    } else {
      int ni, oy, ox;
      decode_m(g, m, ni, oy, ox);
      opix = (size_t)out_pixel(g, ni, oy, ox);
    }
    float o[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (n + e < wrows) {
        if (bias != nullptr) o[e] += bias[n + e];
        o[e] = apply_act(o[e], act);
      } else {
        o[e] = 0.f;
      }
    }
    T* dst = out + opix * ldc + n;
    if constexpr (sizeof(T) == 2) {
      u32x2 pk;
      pk.x = (uint32_t)f32_to_bf16(o[0]) | ((uint32_t)f32_to_bf16(o[1]) << 16);
      pk.y = (uint32_t)f32_to_bf16(o[2]) | ((uint32_t)f32_to_bf16(o[3]) << 16);
      *reinterpret_cast<u32x2*>(dst) = pk;
    } else {
      f32x4 pk = {o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>(dst) = pk;
    }
  }
}

// split-K finalize: out = act(sum over slices of slab[z] + bias).  Slab row r belongs to class k
// (m_base[k] <= r < m_base[k+1]), local row m = r - m_base[k]; only the output pixels the classes cover are written
// (all of them for a conv / dgrad frame; the reflect ring for a ring launch).  Padded channels are forced to 0.
template <typename T>
__global__ void splitk_finalize_kernel(const DescPack pack, const float* __restrict__ ws, const int zs, const size_t slab_elems,
                                       const float* __restrict__ bias, T* __restrict__ out, int ldc, int co, int act) {
  // four channels per thread (ldc is a multiple of the 16-byte vector: 8 bf16 / 4 f32), two slabs in flight
  size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t total4 = slab_elems >> 2;
  for (; q < total4; q += stride) {
    const size_t i = q << 2;
    const int c = (int)(i % (size_t)ldc);
    const int r = (int)(i / (size_t)ldc);
    int k = 0;
    while (k + 1 < pack.n && r >= pack.m_base[k + 1]) ++k;
    const GatherDesc& g = pack.d[k];
    int n, oy, ox;
    decode_m(g, r - pack.m_base[k], n, oy, ox);
    const size_t opix = (size_t)out_pixel(g, n, oy, ox);
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (c < co) {
      int z = 0;
      for (; z + 1 < zs; z += 2) {
        v0 += *reinterpret_cast<const f32x4*>(ws + (size_t)z * slab_elems + i);
        v1 += *reinterpret_cast<const f32x4*>(ws + (size_t)(z + 1) * slab_elems + i);
      }
      if (z < zs) v0 += *reinterpret_cast<const f32x4*>(ws + (size_t)z * slab_elems + i);
    }
    const f32x4 v = v0 + v1;
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float o = 0.f;
      if (c + e < co) {
        o = vv[e];
        if (bias != nullptr) o += bias[c + e];
        o = apply_act(o, act);
      }
      Elem<T>::store(out + opix * ldc + c + e, o);
    }
  }
}

// LDS swizzle for the pixel-major wgrad tiles read with ds_read_b64_tr_b16: a 32-lane half reads 4 consecutive
// rows x 64 bytes; spread the 4 rows over the four 64-byte quarters of the 256-byte bank row.
template <int ROWB> DEI2I_D int tr_swz(int row) {
  if constexpr (ROWB % 256 == 0) return (row & 3) << 6;
  else return ((row >> 1) & 1) << 6;    // 128-byte rows: rows r and r+2 alias, flip the 64-byte half
}

// ------------------------------------------------------------------------------------------------
// wgrad:  dw[co][k] = sum_m dy[m][co] * gather(src)[m][k]
// The pixel range is split over gridDim.z; split z stores its partial to slab z (dw + z*slab_elems, plain stores) and
// wgrad_reduce_unpack sums the slabs in a fixed order -- deterministic, no atomics.  Callers that hand in a bare packed
// buffer (dei2i_conv2d_wgrad) get a single split.
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void wgrad_kernel(const GatherDesc g, const T* __restrict__ src,
                                                    const T* __restrict__ dy, const int co_rows, const int ldy,
                                                    float* __restrict__ dw_base, const int tiles_k,
                                                    const int chunks_per_split, const long long slab_elems) {
  constexpr int VEC = Elem<T>::VEC;
  constexpr bool IS_BF16 = sizeof(T) == 2;
  float* __restrict__ dw = dw_base + (size_t)blockIdx.z * (size_t)slab_elems;
  constexpr int BR = 128 / (int)sizeof(T);           // pixels per reduction chunk (64 bf16 / 32 f32)
  constexpr int VPR_A = BM / VEC, VPR_B = BN / VEC;  // 16-byte vectors per LDS row
  constexpr int RPP_A = 256 / VPR_A, RPP_B = 256 / VPR_B;   // rows covered per pass
  constexpr int NA = BR / RPP_A, NB = BR / RPP_B;    // vectors per thread per chunk
  constexpr int ROWB_A = BM * (int)sizeof(T), ROWB_B = BN * (int)sizeof(T);
  constexpr int WTM = BM / 2, WTN = BN / 2;          // 2x2 waves
  constexpr int TM = WTM / 32, TN = WTN / 32;
  static_assert(NA >= 1 && NB >= 1, "tile too wide for 256 threads");
  static_assert(TM >= 1 && TN >= 1, "wave tile must hold at least one 32x32 block");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int A_BYTES = BR * ROWB_A, B_BYTES = BR * ROWB_B;
  constexpr int TILE_BYTES = A_BYTES + B_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = gridDim.x;
  const int bid = xcd_remap(blockIdx.x, nwg);
  const int tile_k = bid % tiles_k, tile_c = bid / tiles_k;
  const int c0 = tile_c * BM, k0 = tile_k * BN;

  const int nchunks = (g.M + BR - 1) / BR;
  const int cbeg = blockIdx.z * chunks_per_split;
  const int cend = min(nchunks, cbeg + chunks_per_split);
  if (cbeg >= cend) return;

  // A (dy) loader: fixed channel vector, rows strided
  const int va = tid % VPR_A, ra0 = tid / VPR_A;
  const int a_c = c0 + va * VEC;
  const bool a_cok = a_c < ldy;
  // B (gathered patches) loader: fixed k vector -> fixed (tap, ci)
  const int vb = tid % VPR_B, rb0 = tid / VPR_B;
  const int b_k = k0 + vb * VEC;
  const bool b_kok = b_k < g.K;
  int b_tap, b_ci;
  decode_k(g, b_kok ? b_k : 0, b_tap, b_ci);
  const int b_ty = (int)fd_div((uint32_t)b_tap, g.fd_tw);
  const int b_dy = b_ty * g.ys, b_dx = (b_tap - b_ty * g.tw) * g.xs;

  u32x4 areg[NA], breg[NB];
  auto load_chunk = [&](int chunk) {
    const int mb = chunk * BR;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int m = mb + ra0 + RPP_A * i;
      areg[i] = (a_cok && m < g.M) ? ld16(dy + (size_t)m * ldy + a_c) : zero16();
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int m = mb + rb0 + RPP_B * i;
      u32x4 v = zero16();
      if (b_kok && m < g.M) {
        int n, oy, ox;
        decode_m(g, m, n, oy, ox);
        const int y = bound_coord(oy * g.sh + g.by0 + b_dy, g.Hl, g.pad_mode);
        const int x = bound_coord(ox * g.sw + g.bx0 + b_dx, g.Wl, g.pad_mode);
        if ((y | x) >= 0) {
          const int pix = (n * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up);
          v = ld16(src + (size_t)pix * g.Cs + b_ci);
        }
      }
      breg[i] = v;
    }
  };
  auto store_chunk = [&](int buf) {
    unsigned char* ab = smem + buf * TILE_BYTES;
    unsigned char* bb = ab + A_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int row = ra0 + RPP_A * i;
      int off = va * 16;
      if (IS_BF16) off ^= tr_swz<ROWB_A>(row);
      *reinterpret_cast<u32x4*>(ab + row * ROWB_A + off) = areg[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = rb0 + RPP_B * i;
      int off = vb * 16;
      if (IS_BF16) off ^= tr_swz<ROWB_B>(row);
      *reinterpret_cast<u32x4*>(bb + row * ROWB_B + off) = breg[i];
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
  // transposed-read lane roles (ds_read_b64_tr_b16): 16-lane group grp, lane i = 4q+p supplies row q, cols 4p..4p+3
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_cohalf = (lane >> 4) & 1;

  auto compute_chunk = [&](int buf) {
    const unsigned char* ab = smem + buf * TILE_BYTES;
    const unsigned char* bb = ab + A_BYTES;
    if constexpr (IS_BF16) {
      u32x4 af[BR / 16][TM], bf[BR / 16][TN];
      auto load_frags = [&](int kk, u32x4 (&a)[TM], u32x4 (&b)[TN]) {
        const int row_lo = kk * 16 + 8 * lh + tr_q;     // rows row_lo (k 0..3) and row_lo+4 (k 4..7)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int colb = (wm * WTM + i * 32 + 16 * tr_cohalf + 4 * tr_p) * 2;
          const int o0 = row_lo * ROWB_A + (colb ^ tr_swz<ROWB_A>(row_lo));
          const int o1 = (row_lo + 4) * ROWB_A + (colb ^ tr_swz<ROWB_A>(row_lo + 4));
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ab + o0));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ab + o1));
          u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          a[i].x = l2.x; a[i].y = l2.y; a[i].z = h2.x; a[i].w = h2.y;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int colb = (wn * WTN + j * 32 + 16 * tr_cohalf + 4 * tr_p) * 2;
          const int o0 = row_lo * ROWB_B + (colb ^ tr_swz<ROWB_B>(row_lo));
          const int o1 = (row_lo + 4) * ROWB_B + (colb ^ tr_swz<ROWB_B>(row_lo + 4));
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(bb + o0));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(bb + o1));
          u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
          b[j].x = l2.x; b[j].y = l2.y; b[j].z = h2.x; b[j].w = h2.y;
        }
      };
#pragma unroll
      for (int kk = 0; kk < BR / 16; ++kk) load_frags(kk, af[kk], bf[kk]);     // every fragment read of the chunk in flight
#pragma unroll
      for (int kk = 0; kk < BR / 16; ++kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[kk][i]),
                                                                 __builtin_bit_cast(bf16x8, bf[kk][j]), acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int kk = 0; kk < BR / 2; ++kk) {
        float af[TM], bf[TN];
        const int row = kk * 2 + lh;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          af[i] = *reinterpret_cast<const float*>(ab + row * ROWB_A + (wm * WTM + i * 32 + lr) * 4);
#pragma unroll
        for (int j = 0; j < TN; ++j)
          bf[j] = *reinterpret_cast<const float*>(bb + row * ROWB_B + (wn * WTN + j * 32 + lr) * 4);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // rotated so that load / compute / store each appear once in the instruction stream (cold-start code walk)
  {
    int buf = 1;
#pragma unroll 1
    for (int c = cbeg; c <= cend; ++c) {
      const bool more = c < cend;
      if (more) load_chunk(c);
      if (c > cbeg) compute_chunk(buf);
      if (more) store_chunk(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }
  }

  // D[i = co][j = k column]; lanes run along k (contiguous in dw) -> 128-byte store segments.  co_rows is a multiple of
  // 4 (padded channel count): rows 8q .. 8q+3 (+4 lh) share one guard.
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = c0 + wm * WTM + i * 32 + 8 * q + 4 * lh;
      if (co >= co_rows) continue;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int k = k0 + wn * WTN + j * 32 + lr;
        if (k < g.K) {
          float* p = dw + (size_t)co * g.K + k;
#pragma unroll
          for (int r = 0; r < 4; ++r) p[(size_t)r * g.K] = acc[i][j][q * 4 + r];
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int g_num_cu = 256;
static int g_use_v2 = 1;
void set_use_v2(int on) { g_use_v2 = on; }
static int g_use_halo = 1;
void set_use_halo(int on) { g_use_halo = on; }
static int g_use_thin = 1;
void set_use_thin(int on) { g_use_thin = on; }
static int g_splitk_atomic = 0;
void set_splitk_atomic(int on) { g_splitk_atomic = on; }

template <typename T, int BM, int BN, int WM, int WN>
static hipError_t launch_gg(const DescPack& pack, const void* src, const void* wgt, int wrows, const float* bias,
                            void* out, float* ws, int ldc, int act, int splits, int slab_rows, int* zs_out, hipStream_t st) {
  constexpr int BKE = 128 / (int)sizeof(T);
  int tiles_m = 0, Kmax = 0;
  double flops = 0.0;
  for (int i = 0; i < pack.n; ++i) {
    const GatherDesc& g = pack.d[i];
    tiles_m = std::max(tiles_m, (g.M + BM - 1) / BM);
    Kmax = std::max(Kmax, g.K);
    flops += 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)wrows;
  }
  if (pack.skip_dead_taps) flops = 0.0;      // reflect-ring launches skip most taps: left out of the executed-FLOP count
  const int tiles_n = (ldc + BN - 1) / BN;
  const int nk = (Kmax + BKE - 1) / BKE;
  const int kps = (nk + splits - 1) / splits;
  const int zs = (nk + kps - 1) / kps;
  dim3 grid(tiles_m * tiles_n, pack.n, zs);
  const size_t lds = std::max<size_t>(2 * (BM + BN) * 128, (size_t)BM * (BN + 4) * sizeof(float));   // tiles | epilogue stage
  auto kern = gather_gemm_kernel<T, BM, BN, WM, WN>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  count_launch(K_GATHER_V1);
  prof_begin(PROF_GATHER_GEMM, flops, st);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, pack, (const T*)src, (const T*)wgt, wrows, bias, (T*)out,
                     zs > 1 ? ws : (float*)nullptr, ldc, act, tiles_n, kps, slab_rows);
  prof_end(PROF_GATHER_GEMM, st);
  *zs_out = zs;
  return hipGetLastError();
}

template <typename T>
static hipError_t gather_gemm_t(const DescPack& pack, const void* src, const void* wgt, int wrows, const float* bias,
                                void* out, float* ws, size_t ws_bytes, int ldc, int act, hipStream_t st) {
  constexpr int BKE = 128 / (int)sizeof(T);
  // tile choice by GEMM-N
  int BN = ldc > 64 ? 128 : (ldc > 32 ? 64 : 32);
  const int BM = 128;
  int tiles = 0, nk = 0;
  bool any = false;
  for (int i = 0; i < pack.n; ++i) {
    const GatherDesc& g = pack.d[i];
    if (g.M <= 0) continue;
    any = true;
    tiles += ((g.M + BM - 1) / BM) * ((ldc + BN - 1) / BN);
    nk = std::max(nk, (g.K + BKE - 1) / BKE);
  }
  if (!any) return hipSuccess;
  const GatherDesc& g0 = pack.d[0];
  // split-K when the grid cannot fill the chip: aim for >= 2 workgroups per CU, down to 2 k-steps per workgroup (small-M
  // layers -- deep D convs, the 5x5 SPADE table convs, reflect rings -- are a serial chain of k-steps on a handful of
  // CUs).  Every slice writes its own fp32 slab with plain stores (rows = class base + m) and the finalize kernel sums
  // the slabs: no memset, no float atomics (chip-wide atomic rate ~1.3 TB/s against ~6 TB/s of plain stores), and the
  // result does not depend on arrival order.
  const int slab_rows = pack.m_base[pack.n - 1] + pack.d[pack.n - 1].M;
  const size_t slab_elems = (size_t)slab_rows * ldc;
  int splits = 1;
  if (tiles < g_num_cu && nk >= 4 && ws != nullptr && ws_bytes >= 2 * slab_elems * sizeof(float)) {
    splits = (2 * g_num_cu + tiles - 1) / tiles;
    if (splits > nk / 2) splits = nk / 2;
    const size_t fit = ws_bytes / (slab_elems * sizeof(float));
    if ((size_t)splits > fit) splits = (int)fit;
    if (splits < 1) splits = 1;
  }
  if (splits > 1 && !pack.ws_compact && (long long)slab_rows < (long long)g0.N * g0.OH * g0.OW) {
    // classes that do not cover the output frame (no such conv on the reference path): uncovered pixels read as zero
    hipError_t e0 = hipMemsetAsync(out, 0, (size_t)g0.N * g0.OH * g0.OW * ldc * sizeof(T), st);
    if (e0 != hipSuccess) return e0;
  }
  if (splits > 1 && pack.ws_atomic) {
    hipError_t e0 = hipMemsetAsync(ws, 0, slab_elems * sizeof(float), st);
    if (e0 != hipSuccess) return e0;
  }
  hipError_t e;
  int zs = 1;
  if (BN == 128) e = launch_gg<T, 128, 128, 2, 2>(pack, src, wgt, wrows, bias, out, ws, ldc, act, splits, slab_rows, &zs, st);
  else if (BN == 64) e = launch_gg<T, 128, 64, 2, 2>(pack, src, wgt, wrows, bias, out, ws, ldc, act, splits, slab_rows, &zs, st);
  else e = launch_gg<T, 128, 32, 4, 1>(pack, src, wgt, wrows, bias, out, ws, ldc, act, splits, slab_rows, &zs, st);
  if (e != hipSuccess) return e;
  if (zs > 1) {
    const int threads = 256;
    size_t blocks = ((slab_elems >> 2) + threads - 1) / threads;
    if (blocks > 4096) blocks = 4096;
    count_launch(K_SPLITK_FINALIZE);
    hipLaunchKernelGGL(splitk_finalize_kernel<T>, dim3((unsigned)blocks), dim3(threads), 0, st, pack, (const float*)ws,
                       pack.ws_atomic ? 1 : zs, slab_elems,
                       bias, (T*)out, ldc, wrows, act);
    e = hipGetLastError();
  }
  return e;
}

hipError_t gather_gemm_multi(int dtype, const GatherDesc* descs, const long long* woffs, int n, const void* src,
                             const void* wgt, int wrows, const float* bias, void* out, float* ws, size_t ws_bytes, int ldc,
                             int act, hipStream_t st, bool compact_ws) {
  if (n < 1 || n > 4) return hipErrorInvalidValue;
  DescPack pack;
  pack.n = 0;
  pack.ws_compact = compact_ws ? 1 : 0;
  pack.ws_atomic = g_splitk_atomic;
  pack.skip_dead_taps = compact_ws ? 1 : 0;      // the reflect-ring launches are the compact ones
  for (int i = 0; i < n; ++i) {
    if (descs[i].M <= 0) continue;           // empty parity class
    pack.d[pack.n] = descs[i];
    pack.woff[pack.n] = woffs[i];
    pack.fd_taps[pack.n] = make_fastdiv((uint32_t)(descs[i].th * descs[i].tw));
    pack.m_base[pack.n] = pack.n == 0 ? 0 : pack.m_base[pack.n - 1] + pack.d[pack.n - 1].M;
    pack.n++;
  }
  if (pack.n == 0) return hipSuccess;
  for (int i = pack.n; i < 4; ++i) { pack.d[i] = pack.d[0]; pack.woff[i] = 0; pack.fd_taps[i] = pack.fd_taps[0]; pack.m_base[i] = 0; }
  if (dtype == DT_BF16) {
    if (g_use_thin && pack.n == 1) {      // 8-channel inputs (stem, D's first conv, heads dgrad): thin_conv.hip
      hipError_t e = thin_cin_conv(pack.d[0], src, (const bf16_t*)wgt + pack.woff[0], wrows, bias, out, ldc, act, g_num_cu, st);
      if (e != hipErrorNotSupported) return e;
      e = thin_cout_conv(pack.d[0], src, (const bf16_t*)wgt + pack.woff[0], wrows, bias, out, ldc, act, g_num_cu, st);
      if (e != hipErrorNotSupported) return e;
    }
    if (g_use_halo && pack.n == 1) {      // stride-1 3x3 layers: halo-resident kernels (conv_halo16.hip, conv_halo.hip)
      hipError_t e = halo16_conv(pack.d[0], src, (const bf16_t*)wgt + pack.woff[0], wrows, bias, out, ldc, act, g_num_cu, st);
      if (e != hipErrorNotSupported) return e;
      e = halo_conv(pack.d[0], src, (const bf16_t*)wgt + pack.woff[0],
                               wrows, bias, out, ldc, act, g_num_cu, st);
      if (e != hipErrorNotSupported) return e;
    }
    if (g_use_v2) {
      hipError_t e = gather_gemm_v2(pack, src, wgt, wrows, bias, out, ws, ws_bytes, ldc, act, g_num_cu, st);
      if (e != hipErrorNotSupported) return e;
    }
    return gather_gemm_t<bf16_t>(pack, src, wgt, wrows, bias, out, ws, ws_bytes, ldc, act, st);
  }
  return gather_gemm_t<float>(pack, src, wgt, wrows, bias, out, ws, ws_bytes, ldc, act, st);
}

hipError_t gather_gemm(int dtype, const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias,
                       void* out, float* ws, size_t ws_bytes, int ldc, int act, hipStream_t st) {
  const long long zero = 0;
  return gather_gemm_multi(dtype, &g, &zero, 1, src, wgt, wrows, bias, out, ws, ws_bytes, ldc, act, st);
}

template <typename T, int BM, int BN>
static hipError_t launch_wg(const GatherDesc& g, const void* src, const void* dy, int co_rows, int ldy, float* dw,
                            size_t capacity_elems, int* nsplit_out, hipStream_t st) {
  constexpr int BR = 128 / (int)sizeof(T);
  const int tiles_c = (co_rows + BM - 1) / BM;
  const int tiles_k = (g.K + BN - 1) / BN;
  const int nchunks = (g.M + BR - 1) / BR;
  const int tiles = tiles_c * tiles_k;
  const bool slabs = nsplit_out != nullptr;            // deterministic mode: one slab per split, reduced by the caller
  int splits = (2 * g_num_cu + tiles - 1) / tiles;     // every split costs a slab write + read
  if (splits > nchunks / 2) splits = nchunks / 2;
  const long long slab_elems = wgrad_slab_elems(co_rows, g.K);
  if (slabs && (size_t)slab_elems * (size_t)splits > capacity_elems) splits = (int)(capacity_elems / (size_t)slab_elems);
  if (splits < 1 || !slabs) splits = 1;                // a bare packed buffer: one split, plain stores
  const int cps = (nchunks + splits - 1) / splits;
  const int zs = (nchunks + cps - 1) / cps;
  const size_t lds = 2 * (size_t)BR * (BM + BN) * sizeof(T);
  auto kern = wgrad_kernel<T, BM, BN>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  if (slabs) *nsplit_out = zs;
  count_launch(K_WGRAD_V1);
  prof_begin(PROF_WGRAD, 2.0 * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)co_rows, st);
  hipLaunchKernelGGL(kern, dim3(tiles, 1, zs), dim3(256), lds, st, g, (const T*)src, (const T*)dy, co_rows, ldy, dw,
                     tiles_k, cps, slabs ? slab_elems : 0ll);
  prof_end(PROF_WGRAD, st);
  return hipGetLastError();
}

hipError_t wgrad_gemm(int dtype, const GatherDesc& g, const void* src, const void* dy, int co_rows, int ldy, float* dw,
                      size_t capacity_elems, int* nsplit_out, hipStream_t st) {
  if (g.M <= 0) {
    if (nsplit_out) *nsplit_out = 0;      // empty sum: the reduce kernel writes zeros
    return hipSuccess;
  }
  if (dtype == DT_BF16) {
    if (co_rows > 64) return launch_wg<bf16_t, 128, 128>(g, src, dy, co_rows, ldy, dw, capacity_elems, nsplit_out, st);
    return launch_wg<bf16_t, 64, 128>(g, src, dy, co_rows, ldy, dw, capacity_elems, nsplit_out, st);
  }
  if (co_rows > 64) return launch_wg<float, 128, 128>(g, src, dy, co_rows, ldy, dw, capacity_elems, nsplit_out, st);
  return launch_wg<float, 64, 128>(g, src, dy, co_rows, ldy, dw, capacity_elems, nsplit_out, st);
}

void set_num_cu(int n) { g_num_cu = n > 0 ? n : 256; }

}  // namespace dei2i
