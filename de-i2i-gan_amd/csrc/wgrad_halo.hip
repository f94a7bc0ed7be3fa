// Halo-resident wgrad (bf16, stride-1 3x3): dw[co][tap][ci] = sum over pixels p of dy[p][co] * x[p + tap][ci].
//
// wgrad_v2 streams, per 64-pixel chunk, a dy tile and ONE tap's gathered x tile: 48 KB of LDS-DMA for 4.2 MFLOP, and
// the CU's global->LDS path moves ~16 B/clk (one 16-byte-per-lane instruction per ~64 cycles), so its loop runs at
// 3 000 cycles per chunk against 1 024 cycles of MFMA issue -- and every k-tile of a pixel split re-reads dy
// (measured 576 MB of HBM/MALL traffic per launch against 67 MB algorithmic).  Here a workgroup owns a
// (128 output channels) x (64 input channels) x (all 9 taps) block of dw -- 144 accumulator registers per thread --
// and walks 4 x 32 pixel half-tiles of its pixel range: per half-tile it loads the dy tile (128 px x 128 co, 32 KB)
// and the 6 x 34 input halo of its 64-channel slice (26 KB) once and runs all nine taps from LDS: 58 KB per 18.9 MFLOP,
// 3 600 cycles of DMA under 4 600 cycles of MFMA issue.
//
//   grid     : (pixel splits, combos) with combos = ceil(Cout/128) * Cs/64; the combos of one split share an XCD's L2
//   LDS      : 2 stages x [dy 128 px x 256 B | halo 208 px x 128 B]; both pixel-major (the reduction index is the row),
//              fragments fetched with ds_read_b64_tr_b16; rows swizzled for its 2 x 32 lane groups (tr_swz)
//   waves    : 8 = 4 (32-channel blocks of co) x 2 (32-channel blocks of ci); wave (cb, ib) owns dw[cb][ib] for 9 taps
//   output   : each split's partial block goes to its fp32 slab [Cout][9][Cs] with plain 128-byte stores; the slabs are
//              summed and un-packed to OIHW by wgrad_reduce_unpack_kernel (wgrad_v2.hip) -- no atomics, deterministic
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "common.h"
#include "geom.h"
#include "launch.h"

namespace dei2i {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_hw[256];

typedef __attribute__((address_space(3))) void lds_void_hw;
typedef __attribute__((address_space(1))) const void gbl_void_hw;

DEI2I_D void glds16hw(const void* gptr, unsigned char* lds_wave_base) { glds16_asm(gptr, lds_wave_base); }      // (common.h)

constexpr int HW_TH = 4, HW_TW = 32;                  // half-tile: 128 pixels
constexpr int HW_HH = HW_TH + 2, HW_HWD = HW_TW + 2;  // 6 x 34 halo
constexpr int HW_HPIX = HW_HH * HW_HWD;               // 204
constexpr int HW_HGROUPS = 26;                        // 8-pixel DMA groups (208 >= 204)
constexpr int HW_HPAD = HW_HGROUPS * 8;                // 208 halo pixel rows per stage

// 256-byte rows: spread 4 consecutive rows over the four 64-byte quarters; 128-byte rows: rows r and r+2 alias, flip
// the 64-byte half (the tr-read swizzles of conv_gemm.hip)
template <int ROWB> DEI2I_D int hw_swz(int row) {
  if constexpr (ROWB == 256) return (row & 3) << 6;
  else return ((row >> 1) & 1) << 6;
}

// BCO x BCI = 128 x 64 (Cout >= 96) or 64 x 128 (Cout <= 64): 8 waves = (BCO/32) x (BCI/32) blocks, 9 taps each;
// 64 x 64 with the taps split over NTG = 2 wave groups for 64-channel inputs (the 64 -> 4 heads: one live co block)
// PRO: the conv's input was normalised + activated on the operand path in the forward pass (ConvPro, geom.h; the
// normalised tensor z was never written), so the same transform is applied here to every landed input halo, in place in
// LDS, before its nine taps are read: one extra pass over 26 KB and one extra barrier per half-tile.
template <int N> DEI2I_D void hw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BCO, int BCI, int NTG, bool PRO = false>
__global__ __launch_bounds__(512) void wgrad_halo_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                         const bf16_t* __restrict__ dy, const int co_rows, const int ldy,
                                                         float* __restrict__ slabs, const int nslices, const int tiles_per_split,
                                                         const long long slab_elems, const ConvPro pro) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NIB = BCI / 32, NCB = BCO / 32;
  constexpr int TPW = (9 + NTG - 1) / NTG;            // taps per wave
  static_assert(NCB * NIB * NTG == 8, "8 waves");
  constexpr int ROWB_A = BCO * 2, ROWB_B = BCI * 2;
  constexpr int HW_A_BYTES = 128 * ROWB_A, HW_B_BYTES = HW_HPAD * ROWB_B, HW_STAGE = HW_A_BYTES + HW_B_BYTES;
  constexpr int RPI_A = 1024 / ROWB_A, SPR_A = ROWB_A / 16;      // dy rows per DMA instruction, 16-byte slots per row
  constexpr int RPI_B = 1024 / ROWB_B, SPR_B = ROWB_B / 16;
  constexpr int NA = 128 / RPI_A / 8, NB = (HW_HPAD / RPI_B + 7) / 8;   // DMA instructions per wave per half-tile
  constexpr int GROUPS_B = HW_HPAD / RPI_B;
  const int tg = wave / (NCB * NIB);                  // tap group
  const int cb = (wave / NIB) % NCB, ib = wave % NIB; // 32-channel block of co / of ci
  const int tap0 = NTG == 1 ? 0 : tg * TPW;           // compile-time 0 for the 9-taps-per-wave variants

  const int combo = blockIdx.y, split = blockIdx.x;
  const int tile_c = combo / nslices, slice = combo - tile_c * nslices;
  const int c0 = tile_c * BCO, ci0 = slice * BCI;

  const int tiles_x = g.Wo / HW_TW, tiles_y = g.Ho / HW_TH;
  const int tiles_img = tiles_x * tiles_y;
  const int ntiles = g.N * tiles_img;
  const int tbeg = split * tiles_per_split;
  const int tend = min(ntiles, tbeg + tiles_per_split);
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_hw);

  // ---- LDS-DMA roles ----
  // dy: one instruction = 4 pixel rows x 256 B: lane l -> row (l>>4), 16-byte slot (l&15); 32 instructions per tile
  // x : one instruction = 8 halo pixels x 128 B: lane l -> pixel (l>>3), slot (l&7); 26 groups in 32 slots
  auto issue = [&](int stage, int t) {
    unsigned char* sa = smem + stage * HW_STAGE;
    unsigned char* sb = sa + HW_A_BYTES;
    const int img = t / tiles_img;
    const int rem = t - img * tiles_img;
    const int ty = rem / tiles_x;
    const int y0 = ty * HW_TH, x0 = (rem - ty * tiles_x) * HW_TW;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int rgrp = (j * 8 + wave) * RPI_A;
      const int r = rgrp + lane / SPR_A;                                  // pixel of the half-tile: (r>>5, r&31)
      const int off = ((lane % SPR_A) * 16) ^ hw_swz<ROWB_A>(r);          // source byte offset that belongs at this slot
      const int c = c0 + (off >> 1);
      const size_t pix = ((size_t)img * g.Ho + y0 + (r >> 5)) * g.Wo + x0 + (r & 31);
      const bf16_t* p = c < ldy ? dy + pix * ldy + c : zero;
      glds16hw(p, sa + rgrp * ROWB_A);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      int grp = j * 8 + wave;
      if (grp >= GROUPS_B) grp -= GROUPS_B;                               // surplus slots re-fetch the first groups
      const int hp = grp * RPI_B + lane / SPR_B;
      const int hy = hp / HW_HWD, hx = hp - hy * HW_HWD;
      const int off = ((lane % SPR_B) * 16) ^ hw_swz<ROWB_B>(hp);
      const bf16_t* p = zero;
      if (hp < HW_HPIX) {
        const int y = bound_coord(y0 + g.by0 + hy, g.Hl, g.pad_mode);
        const int x = bound_coord(x0 + g.bx0 + hx, g.Wl, g.pad_mode);
        if ((y | x) >= 0) {
          if (PRO && pro.ring != nullptr && !(ring_interior(y, g.Hl) && ring_interior(x, g.Wl)))
            p = pro.ring + ((size_t)(img * pro.ring_pix + ring_index(y, x, g.Hl, g.Wl))) * g.Cs + ci0 + (off >> 1);
          else
            p = src + ((size_t)((img * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up))) * g.Cs + ci0 + (off >> 1);
        }
      }
      glds16hw(p, sb + grp * 1024);
    }
  };
  // PRO: z = act(A*x + B) on the landed halo of tile t (ring pixels arrive normalised, padding stays zero)
  float* const pcoef = reinterpret_cast<float*>(smem + 2 * HW_STAGE);      // A[BCI] | B[BCI] of the current image's slice
  auto transform = [&](int stage, int t) {
    unsigned char* sb = smem + stage * HW_STAGE + HW_A_BYTES;
    const int img = t / tiles_img;
    const int rem = t - img * tiles_img;
    const int ty = rem / tiles_x;
    const int y0 = ty * HW_TH, x0 = (rem - ty * tiles_x) * HW_TW;
    for (int v = tid; v < HW_HPIX * SPR_B; v += 512) {
      const int hp = v / SPR_B, sl = v - hp * SPR_B;
      const int hy = hp / HW_HWD, hx = hp - hy * HW_HWD;
      const int y = bound_coord(y0 + g.by0 + hy, g.Hl, g.pad_mode);
      const int x = bound_coord(x0 + g.bx0 + hx, g.Wl, g.pad_mode);
      if ((y | x) < 0) continue;
      if (pro.ring != nullptr && !(ring_interior(y, g.Hl) && ring_interior(x, g.Wl))) continue;
      const int cofs = ((sl * 16) ^ hw_swz<ROWB_B>(hp)) >> 1;
      float A8[8], B8[8], f[8];
      ldcoef<8>(pcoef + cofs, A8);
      ldcoef<8>(pcoef + BCI + cofs, B8);
      u32x4* p = reinterpret_cast<u32x4*>(sb + hp * ROWB_B + sl * 16);
      Elem<bf16_t>::unpack(*p, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float w = fmaf(A8[e], f[e], B8[e]);
        f[e] = fmaf(pro.slope, fminf(w, 0.f), fmaxf(w, 0.f));
      }
      *p = Elem<bf16_t>::pack(f);
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
  const int b_colb = (ib * 32 + 16 * tr_half + 4 * tr_p) * 2;           // byte column in a halo row

  auto tr_read = [&](const unsigned char* base, int o0, int o1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o1));
    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 r;
    r.x = l2.x; r.y = l2.y; r.z = h2.x; r.w = h2.y;
    return r;
  };

  auto compute = [&](int stage) {
    const unsigned char* ab = smem + stage * HW_STAGE;
    const unsigned char* bb = ab + HW_A_BYTES;
#pragma unroll 2
    for (int kb = 0; kb < 8; ++kb) {                                      // 16-pixel reduction blocks of the half-tile
      const int ra = kb * 16 + 8 * lh + tr_q;                             // dy rows ra, ra+4
      const u32x4 af = tr_read(ab, ra * ROWB_A + (a_colb ^ hw_swz<ROWB_A>(ra)), (ra + 4) * ROWB_A + (a_colb ^ hw_swz<ROWB_A>(ra + 4)));
      const int py = kb >> 1, px0 = (kb & 1) * 16;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int tap = tap0 + t;                                         // wave-uniform
        if (NTG > 1 && tap >= 9) break;
        const int ty = tap / 3, tx = tap - ty * 3;
        const int rb = (py + ty) * HW_HWD + px0 + tx + 8 * lh + tr_q;     // halo pixels rb, rb+4
        const u32x4 bf = tr_read(bb, rb * ROWB_B + (b_colb ^ hw_swz<ROWB_B>(rb)), (rb + 4) * ROWB_B + (b_colb ^ hw_swz<ROWB_B>(rb + 4)));
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[t], 0, 0, 0);
      }
    }
  };

  // ---- two-stage ring over the half-tiles of this split: one barrier per half-tile (~4 600 cycles of MFMA) ----
  // (rotated: the trip it = -1 only issues stage 0, so `issue` and `compute` each exist once in the instruction stream)
  const int nt = tend - tbeg;
  int cur_img = -1;
#pragma unroll 1
  for (int it = -1; it < nt; ++it) {
    if (it >= 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // my share of stage `it` has landed
      __builtin_amdgcn_s_barrier();                                       // ... everyone's; compute(it-1) is done everywhere
    }
    bool reload = false;
    float cval = 0.f;
    if constexpr (PRO) {
      if (it >= 0 && pro.A != nullptr) {            // (A == nullptr: ring redirect only -- the tensor is already normalised)
        const int img = (tbeg + it) / tiles_img;
        reload = img != cur_img;                                          // workgroup-uniform: a new image's coefficients
        if (reload && tid < 2 * BCI)
          cval = (tid < BCI ? pro.A : pro.B)[(size_t)img * pro.n_stride + ci0 + (tid < BCI ? tid : tid - BCI)];
        cur_img = img;
      }
    }
    if (it + 1 < nt) issue((it + 1) & 1, tbeg + it + 1);
    if constexpr (PRO) {
      if (it >= 0 && pro.A != nullptr) {
        if (reload) {
          if (it + 1 < nt) hw_wait_vm<NA + NB>(); else hw_wait_vm<0>();   // the coefficient load is older than the DMA just issued
          if (tid < 2 * BCI) pcoef[tid] = cval;
          __syncthreads();
        }
        transform(it & 1, tbeg + it);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                     // the halo is normalised for every wave
      }
    }
    if (it >= 0) compute(it & 1);
  }

  // ---- partial block -> this split's slab [Cout][9][Cs] (lanes run along ci: 128-byte segments) ----
  // co_rows is a multiple of 8 (the padded channel count), and the 8 rows 8q .. 8q+7 of a 32-block are held by the
  // lanes' (e & 3, lh): one wave-uniform guard per row group instead of a per-element exec mask (the unrolled
  // per-element guards were 2/3 of this kernel's code, and a dispatch walks its code cold)
  float* slab = slabs + (size_t)split * slab_elems;
  const int co_w = __builtin_amdgcn_readfirstlane(c0 + cb * 32);
  float* p0 = slab + (size_t)(co_w + 4 * lh) * g.K + ci0 + ib * 32 + lr;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int tap = tap0 + t;
    if (NTG > 1 && tap >= 9) break;
    float* pt = p0 + tap * g.Cs;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (co_w + 8 * q >= co_rows) break;
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[(size_t)(8 * q + r) * g.K] = acc[t][q * 4 + r];
    }
  }
}

template <int BCO, int BCI, int NTG>
static hipError_t launch_wgrad_halo(const GatherDesc& g, const void* src, const void* dy, int co_rows, int ldy, float* slabs,
                                    size_t slab_capacity_elems, int num_cu, int* nsplit_out, bool force, hipStream_t st,
                                    const ConvPro* pro) {
  const int nslices = g.Cs / BCI, tiles_c = (co_rows + BCO - 1) / BCO;
  const int combos = nslices * tiles_c;
  const int ntiles = g.N * (g.Ho / HW_TH) * (g.Wo / HW_TW);
  int splits = std::max(1, num_cu / combos);                  // one workgroup per CU, a single round
  if (splits > 8) splits -= splits % 8;                       // the combos of a split land on one XCD (linear id % 8)
  if (splits > ntiles / 4) splits = std::max(1, ntiles / 4);
  const long long slab_elems = wgrad_slab_elems(co_rows, g.K);
  if ((size_t)slab_elems * splits > slab_capacity_elems) splits = (int)(slab_capacity_elems / (size_t)slab_elems);
  if (splits < 1) return hipErrorNotSupported;
  if (!force && combos * splits < (num_cu * 3) / 4) return hipErrorNotSupported;   // too little parallelism: wgrad_v2 / v1 fill the chip better
  const int tps = (ntiles + splits - 1) / splits;
  const int zs = (ntiles + tps - 1) / tps;
  const size_t lds = 2 * (size_t)(128 * BCO * 2 + HW_HPAD * BCI * 2) + (pro != nullptr ? 2 * BCI * sizeof(float) : 0);
  auto kern = pro != nullptr ? wgrad_halo_kernel<BCO, BCI, NTG, true> : wgrad_halo_kernel<BCO, BCI, NTG, false>;
  static bool attr_done[2] = {false, false};
  if (!attr_done[pro != nullptr]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done[pro != nullptr] = true;
  }
  const ConvPro pv = pro != nullptr ? *pro : ConvPro{nullptr, nullptr, 0, 0.f, nullptr, 0};
  count_launch(K_WGRAD_HALO);
  prof_begin(PROF_WGRAD, 2.0 * (double)g.M * 9.0 * (double)g.Clog * (double)co_rows, st);
  hipLaunchKernelGGL(kern, dim3(zs, combos), dim3(512), lds, st, g, (const bf16_t*)src, (const bf16_t*)dy, co_rows, ldy, slabs,
                     nslices, tps, slab_elems, pv);
  prof_end(PROF_WGRAD, st);
  *nsplit_out = zs;
  return hipGetLastError();
}

// returns hipErrorNotSupported when the shape does not qualify (the caller falls through to wgrad_v2 / v1)
hipError_t wgrad_halo(const GatherDesc& g, const void* src, const void* dy, int co_rows, int ldy, float* slabs,
                      size_t slab_capacity_elems, int num_cu, int* nsplit_out, bool force, hipStream_t st, const ConvPro* pro) {
  if (pro != nullptr && pro->ring != nullptr && (g.Hl < 4 || g.Wl < 4)) return hipErrorNotSupported;
  if (g.sh != 1 || g.sw != 1 || g.ys != 1 || g.xs != 1 || g.th != 3 || g.tw != 3) return hipErrorNotSupported;
  if (g.Ho % HW_TH != 0 || g.Wo % HW_TW != 0) return hipErrorNotSupported;
  if ((long long)g.N * g.Hs * g.Ws * g.Cs >= (1ll << 31)) return hipErrorNotSupported;
  if (co_rows <= 64) {
    if (g.Cs % 128 == 0 && co_rows >= 48)
      return launch_wgrad_halo<64, 128, 1>(g, src, dy, co_rows, ldy, slabs, slab_capacity_elems, num_cu, nsplit_out, force, st, pro);
    if (g.Cs % 64 == 0 && co_rows <= 32)        // thin heads: the work is the input stream; taps split over two wave groups
      return launch_wgrad_halo<64, 64, 2>(g, src, dy, co_rows, ldy, slabs, slab_capacity_elems, num_cu, nsplit_out, force, st, pro);
    return hipErrorNotSupported;
  }
  if (g.Cs % 64 != 0 || co_rows < 96) return hipErrorNotSupported;
  return launch_wgrad_halo<128, 64, 1>(g, src, dy, co_rows, ldy, slabs, slab_capacity_elems, num_cu, nsplit_out, force, st, pro);
}

}  // namespace dei2i
