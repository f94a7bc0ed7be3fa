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

// CBW: 32-channel co blocks per WAVE.  With CBW = 1 a wave reads one dy fragment and NINE input fragments per 16-pixel block for nine
// MFMAs, and the four co waves of an input block read the same input fragments: 640 KB of ds_read_b64_tr per 128-pixel half-tile
// against 4 600 cycles of MFMA issue -- at the 8-byte read rate (128 B/clk) the loop is LDS-read-bound.  CBW = 2 with the taps
// split over two wave groups (5 + 4) gives a wave 64 co x 32 ci x 5 taps: two dy fragments + five input fragments for ten MFMAs,
// 416 KB per half-tile.  Waves w and w + 4 (one SIMD) get the 5-tap and the 4-tap group: every SIMD still issues 18 MFMAs per block.
// ABL: TIMING-ONLY ablation builds (results are wrong; option "wgrad_halo_abl"): bit 1 no LDS-DMA after the first half-tile,
// 2 no fragment reads inside the loop, 4 no MFMAs, 8 no slab stores
template <int BCO, int BCI, int NTG, bool PRO = false, int CBW = 1, int ABL = 0>
__global__ __launch_bounds__(512) void wgrad_halo_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                         const bf16_t* __restrict__ dy, const int co_rows, const int ldy,
                                                         float* __restrict__ slabs, const int nslices, const int tiles_per_split,
                                                         const long long slab_elems, const ConvPro pro,
                                                         unsigned long long* __restrict__ dbg) {
  // dbg (dei2i_set_debug_buffer; tools/diag_wgrad_abl.py): per wave 8 words -- [0] s_memrealtime at entry, [1] cycles to the first
  // compute, [2] to the end of the loop, [3] to the last slab store's completion, [4] s_memrealtime at exit, [5] cycles waiting at the
  // top of the trips (vmcnt + barrier), [6] issuing LDS-DMA, [7] in compute()
  const unsigned long long t_in = dbg != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long r_in = dbg != nullptr ? __builtin_amdgcn_s_memrealtime() : 0ull;
  unsigned long long t_first = 0, c_wait = 0, c_dma = 0, c_comp = 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NIB = BCI / 32, NCB = BCO / (32 * CBW);
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
  int it_now = -1;                                                        // (the trip; ABL 16 keeps the first half-tile's DMA)
  struct TileAt { int img, y0, x0; };
  auto tile_at = [&](int t) {
    TileAt q;
    q.img = t / tiles_img;
    const int rem = t - q.img * tiles_img;
    const int ty = rem / tiles_x;
    q.y0 = ty * HW_TH;
    q.x0 = (rem - ty * tiles_x) * HW_TW;
    return q;
  };
  auto tile_next = [&](TileAt& q) {                                       // t -> t + 1 without the two divisions
    q.x0 += HW_TW;
    if (q.x0 >= g.Wo) {
      q.x0 = 0;
      q.y0 += HW_TH;
      if (q.y0 >= g.Ho) { q.y0 = 0; ++q.img; }
    }
  };
  // Per-lane constants of this wave's DMA instructions, hoisted by hand: the compiler recomputed the whole address chain (two
  // run-time divisions of the tile index, the lane's pixel / slot / swizzle, 64-bit products) for every instruction of every
  // half-tile -- ~365 cycles per instruction, 2 900 of a half-tile's 8 900 cycles with no MFMA issued (tools/diag_wgrad_abl.py:
  // the address arithmetic WITHOUT the instruction cost as much as with it).
  int a_off[NA];                 // element offset of the lane's 16 bytes from the half-tile's first pixel of dy; -1: beyond ldy -> zeros
  int b_hyx[NB], b_c[NB];        // (halo row << 16 | halo column), -1 for the padding slots; channel offset ci0 + the slot's swizzled 8 channels
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int rgrp = (j * 8 + wave) * RPI_A;
    const int r = rgrp + lane / SPR_A;                                    // pixel of the half-tile: (r>>5, r&31)
    const int off = ((lane % SPR_A) * 16) ^ hw_swz<ROWB_A>(r);            // source byte offset that belongs at this slot
    const int c = c0 + (off >> 1);
    a_off[j] = c < ldy ? ((r >> 5) * g.Wo + (r & 31)) * ldy + c : -1;
    asm volatile("" : "+v"(a_off[j]));
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    int grp = j * 8 + wave;
    if (grp >= GROUPS_B) grp -= GROUPS_B;                                 // surplus slots re-fetch the first groups
    const int hp = grp * RPI_B + lane / SPR_B;
    const int hy = hp / HW_HWD, hx = hp - hy * HW_HWD;
    b_hyx[j] = hp < HW_HPIX ? ((hy << 16) | hx) : -1;
    b_c[j] = ci0 + ((((lane % SPR_B) * 16) ^ hw_swz<ROWB_B>(hp)) >> 1);
    asm volatile("" : "+v"(b_hyx[j]), "+v"(b_c[j]));
  }
  // FAST path (reflect padding, the whole co tile inside dy, no operand-path transform: every conv of the generator): a wave-
  // uniform base + the lane's 32-bit byte offset per instruction -- ~10 instructions each instead of ~56 (a wave issues one
  // instruction per ~4-5 cycles: the 451 instructions of the general path's eight pieces WERE the 2 900 cycles)
  const bool fast = !PRO && g.pad_mode == PAD_REFLECT && c0 + BCO <= ldy && !(ABL & 16);
  if (fast) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      if (b_hyx[j] < 0) b_hyx[j] = 0;                                     // padding slots: any valid pixel (never read)
    }
  }
  const unsigned smem_lds = lds_addr_of(smem);
  struct FastBase { const bf16_t* a; const bf16_t* b; int vy0, vx0; };
  auto fast_base = [&](const TileAt& q) {                                 // wave-uniform, once per half-tile
    FastBase f;
    f.a = dy + (long long)((q.img * g.Ho + q.y0) * g.Wo + q.x0) * ldy;
    f.b = src + (size_t)q.img * ((size_t)g.Hs * g.Ws * g.Cs);
    f.vy0 = q.y0 + g.by0;
    f.vx0 = q.x0 + g.bx0;
    return f;
  };
  auto issue_a_fast = [&](int stage, const FastBase& f, int j) {
    const int rgrp = (j * 8 + wave) * RPI_A;
    glds16_asm_s(f.a, (unsigned)a_off[j] * 2u, smem_lds + stage * HW_STAGE + rgrp * ROWB_A);
  };
  auto issue_b_fast = [&](int stage, const FastBase& f, int j) {
    int grp = j * 8 + wave;
    if (grp >= GROUPS_B) grp -= GROUPS_B;
    const int vy = f.vy0 + (b_hyx[j] >> 16), vx = f.vx0 + (b_hyx[j] & 0xffff);
    const int ay = max(vy, -vy), ax = max(vx, -vx);
    const int y = min(ay, 2 * g.Hl - 2 - ay), x = min(ax, 2 * g.Wl - 2 - ax);
    const unsigned voff = (unsigned)(((y >> g.up) * g.Ws + (x >> g.up)) * g.Cs + b_c[j]) * 2u;
    glds16_asm_s(f.b, voff, smem_lds + stage * HW_STAGE + HW_A_BYTES + grp * 1024);
  };
  auto issue_a = [&](int stage, const TileAt& q, int j) {
    unsigned char* sa = smem + stage * HW_STAGE;
    const int rgrp = (j * 8 + wave) * RPI_A;
    const long long base = (long long)((q.img * g.Ho + q.y0) * g.Wo + q.x0) * ldy;      // wave-uniform
    const bf16_t* p = a_off[j] >= 0 ? dy + base + a_off[j] : zero;
    if ((ABL & 16) && it_now >= 0) { asm volatile("" :: "v"(p)); return; }          // timing only: the address arithmetic without the instruction
    glds16hw(p, sa + rgrp * ROWB_A);
  };
  auto issue_b = [&](int stage, const TileAt& q, int j) {
    unsigned char* sb = smem + stage * HW_STAGE + HW_A_BYTES;
    int grp = j * 8 + wave;
    if (grp >= GROUPS_B) grp -= GROUPS_B;
    const bf16_t* p = zero;
    if (b_hyx[j] >= 0) {
      const int y = bound_coord(q.y0 + g.by0 + (b_hyx[j] >> 16), g.Hl, g.pad_mode);
      const int x = bound_coord(q.x0 + g.bx0 + (b_hyx[j] & 0xffff), g.Wl, g.pad_mode);
      if ((y | x) >= 0) {
        if (PRO && pro.ring != nullptr && !(ring_interior(y, g.Hl) && ring_interior(x, g.Wl)))
          p = pro.ring + ((size_t)(q.img * pro.ring_pix + ring_index(y, x, g.Hl, g.Wl))) * g.Cs + b_c[j];
        else
          p = src + (unsigned)(((q.img * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up)) * g.Cs + b_c[j]);   // < 2^31 elements (checked at launch)
      }
    }
    if ((ABL & 16) && it_now >= 0) { asm volatile("" :: "v"(p)); return; }
    glds16hw(p, sb + grp * 1024);
  };
  auto issue = [&](int stage, const TileAt& q) {                          // the whole half-tile in one burst
    if (fast) {
      const FastBase f = fast_base(q);
#pragma unroll
      for (int j = 0; j < NA; ++j) issue_a_fast(stage, f, j);
#pragma unroll
      for (int j = 0; j < NB; ++j) issue_b_fast(stage, f, j);
      return;
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) issue_a(stage, q, j);
#pragma unroll
    for (int j = 0; j < NB; ++j) issue_b(stage, q, j);
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

  f32x16 acc[TPW][CBW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int j = 0; j < CBW; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][j][e] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;
  // transposed-read lane roles (ds_read_b64_tr_b16): lane i = 4q+p of a 16-lane group supplies row q, cols 4p..4p+3
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_half = (lane >> 4) & 1;
  const int a_colb = (cb * CBW * 32 + 16 * tr_half + 4 * tr_p) * 2;     // byte column in a dy row (of the wave's first co block)
  const int b_colb = (ib * 32 + 16 * tr_half + 4 * tr_p) * 2;           // byte column in a halo row

  auto tr_read = [&](const unsigned char* base, int o0, int o1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o1));
    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 r;
    r.x = l2.x; r.y = l2.y; r.z = h2.x; r.w = h2.y;
    return r;
  };

  u32x4 abl_frag = {0x3c003c00u + (unsigned)lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  auto compute = [&](int stage) {
    const unsigned char* ab = smem + stage * HW_STAGE;
    const unsigned char* bb = ab + HW_A_BYTES;
#pragma unroll 2
    for (int kb = 0; kb < 8; ++kb) {                                      // 16-pixel reduction blocks of the half-tile
      const int ra = kb * 16 + 8 * lh + tr_q;                             // dy rows ra, ra+4
      u32x4 af[CBW];
#pragma unroll
      for (int j = 0; j < CBW; ++j) {
        if (ABL & 2) { af[j] = abl_frag; asm volatile("" : "+v"(af[j])); continue; }
        af[j] = tr_read(ab, ra * ROWB_A + ((a_colb + 64 * j) ^ hw_swz<ROWB_A>(ra)), (ra + 4) * ROWB_A + ((a_colb + 64 * j) ^ hw_swz<ROWB_A>(ra + 4)));
      }
      const int py = kb >> 1, px0 = (kb & 1) * 16;
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int tap = tap0 + t;                                         // wave-uniform
        if (NTG > 1 && tap >= 9) break;
        const int ty = tap / 3, tx = tap - ty * 3;
        const int rb = (py + ty) * HW_HWD + px0 + tx + 8 * lh + tr_q;     // halo pixels rb, rb+4
        u32x4 bf;
        if (ABL & 2) { bf = abl_frag; asm volatile("" : "+v"(bf)); }
        else bf = tr_read(bb, rb * ROWB_B + (b_colb ^ hw_swz<ROWB_B>(rb)), (rb + 4) * ROWB_B + (b_colb ^ hw_swz<ROWB_B>(rb + 4)));
#pragma unroll
        for (int j = 0; j < CBW; ++j)
          if (ABL & 4) { acc[t][j][0] += __uint_as_float(af[j].x ^ bf.x); }
          else acc[t][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[j]), __builtin_bit_cast(bf16x8, bf), acc[t][j], 0, 0, 0);
      }
    }
  };

  // ---- two-stage ring over the half-tiles of this split: one barrier per half-tile (~4 600 cycles of MFMA) ----
  // (rotated: the trip it = -1 only issues stage 0, so `issue` and `compute` each exist once in the instruction stream)
  const int nt = tend - tbeg;
  int cur_img = -1;
  TileAt nxt = tile_at(tbeg);                                             // the half-tile the NEXT issue fetches
#pragma unroll 1
  for (int it = -1; it < nt; ++it) {
    it_now = it;
    unsigned long long s0 = dbg != nullptr ? __builtin_amdgcn_s_memtime() : 0ull;
    if (it >= 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // my share of stage `it` has landed
      __builtin_amdgcn_s_barrier();                                       // ... everyone's; compute(it-1) is done everywhere
    }
    if (dbg != nullptr) {
      const unsigned long long s1 = __builtin_amdgcn_s_memtime();
      c_wait += s1 - s0; s0 = s1;
      if (it == 0) t_first = s1 - t_in;
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
    if (it + 1 < nt && (!(ABL & 1) || it < 0)) issue((it + 1) & 1, nxt);
    if (dbg != nullptr) { const unsigned long long s1 = __builtin_amdgcn_s_memtime(); c_dma += s1 - s0; s0 = s1; }
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
    tile_next(nxt);
    if (dbg != nullptr) { asm volatile("s_nop 0" ::: "memory"); c_comp += __builtin_amdgcn_s_memtime() - s0; }
  }
  const unsigned long long t_loop = dbg != nullptr ? __builtin_amdgcn_s_memtime() - t_in : 0ull;

  // ---- partial block -> this split's slab [Cout][9][Cs] (lanes run along ci: 128-byte segments) ----
  // co_rows is a multiple of 8 (the padded channel count), and the 8 rows 8q .. 8q+7 of a 32-block are held by the
  // lanes' (e & 3, lh): one wave-uniform guard per row group instead of a per-element exec mask (the unrolled
  // per-element guards were 2/3 of this kernel's code, and a dispatch walks its code cold)
  float* slab = slabs + (size_t)split * slab_elems;
#pragma unroll
  for (int j = 0; j < CBW; ++j) {
    const int co_w = __builtin_amdgcn_readfirstlane(c0 + (cb * CBW + j) * 32);
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
        for (int r = 0; r < 4; ++r)
          if (!(ABL & 8) || acc[t][j][q * 4 + r] == 123.456f) pt[(size_t)(8 * q + r) * g.K] = acc[t][j][q * 4 + r];
      }
    }
  }
  if (dbg != nullptr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      unsigned long long* rec = dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 8;
      rec[0] = r_in; rec[1] = t_first; rec[2] = t_loop; rec[3] = __builtin_amdgcn_s_memtime() - t_in;
      rec[4] = __builtin_amdgcn_s_memrealtime(); rec[5] = c_wait; rec[6] = c_dma; rec[7] = c_comp;
    }
  }
}

extern unsigned long long* g_v2_dbg;
int g_wgrad_halo_abl = 0;      // timing-only ablation builds of the <128, 64> kernel (see ABL)
int g_wgrad_halo_cbw = 1;      // A/B option "wgrad_halo_cbw": 1 = one co block x nine taps per wave (the round-1 wave layout)

template <int BCO, int BCI, int NTG, int CBW = 1>
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
  auto kern = pro != nullptr ? wgrad_halo_kernel<BCO, BCI, NTG, true, CBW> : wgrad_halo_kernel<BCO, BCI, NTG, false, CBW>;
  constexpr bool AB = BCO == 128 && NTG == 1;                   // the ablation builds exist for the 128 x 64, nine-taps-per-wave layout
  if (AB && pro == nullptr && g_wgrad_halo_abl > 0) {
    switch (g_wgrad_halo_abl) {
      case 1: kern = wgrad_halo_kernel<BCO, BCI, NTG, false, CBW, AB ? 1 : 0>; break;
      case 2: kern = wgrad_halo_kernel<BCO, BCI, NTG, false, CBW, AB ? 2 : 0>; break;
      case 3: kern = wgrad_halo_kernel<BCO, BCI, NTG, false, CBW, AB ? 3 : 0>; break;
      case 4: kern = wgrad_halo_kernel<BCO, BCI, NTG, false, CBW, AB ? 4 : 0>; break;
      case 6: kern = wgrad_halo_kernel<BCO, BCI, NTG, false, CBW, AB ? 6 : 0>; break;
      case 7: kern = wgrad_halo_kernel<BCO, BCI, NTG, false, CBW, AB ? 7 : 0>; break;
      case 8: kern = wgrad_halo_kernel<BCO, BCI, NTG, false, CBW, AB ? 8 : 0>; break;
      case 12: kern = wgrad_halo_kernel<BCO, BCI, NTG, false, CBW, AB ? 12 : 0>; break;
      case 15: kern = wgrad_halo_kernel<BCO, BCI, NTG, false, CBW, AB ? 15 : 0>; break;
      case 16: kern = wgrad_halo_kernel<BCO, BCI, NTG, false, CBW, AB ? 16 : 0>; break;
      default: break;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  static bool attr_done[2] = {false, false};                    // (per template instance of this launcher)
  const int which = pro != nullptr ? 1 : 0;
  if (!attr_done[which]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done[which] = true;
  }
  const ConvPro pv = pro != nullptr ? *pro : ConvPro{nullptr, nullptr, 0, 0.f, nullptr, 0};
  count_launch(K_WGRAD_HALO);
  prof_begin(PROF_WGRAD, 2.0 * (double)g.M * 9.0 * (double)g.Clog * (double)co_rows, st);
  hipLaunchKernelGGL(kern, dim3(zs, combos), dim3(512), lds, st, g, (const bf16_t*)src, (const bf16_t*)dy, co_rows, ldy, slabs,
                     nslices, tps, slab_elems, pv, g_v2_dbg);
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
    if (g.Cs % 128 == 0 && co_rows >= 48 && g_wgrad_halo_cbw == 2)
      return launch_wgrad_halo<64, 128, 2, 2>(g, src, dy, co_rows, ldy, slabs, slab_capacity_elems, num_cu, nsplit_out, force, st, pro);
    if (g.Cs % 128 == 0 && co_rows >= 48)
      return launch_wgrad_halo<64, 128, 1>(g, src, dy, co_rows, ldy, slabs, slab_capacity_elems, num_cu, nsplit_out, force, st, pro);
    if (g.Cs % 64 == 0 && co_rows <= 32)        // thin heads: the work is the input stream; taps split over two wave groups
      return launch_wgrad_halo<64, 64, 2>(g, src, dy, co_rows, ldy, slabs, slab_capacity_elems, num_cu, nsplit_out, force, st, pro);
    return hipErrorNotSupported;
  }
  if (g.Cs % 64 != 0 || co_rows < 96) return hipErrorNotSupported;
  if (g_wgrad_halo_cbw == 2)
    return launch_wgrad_halo<128, 64, 2, 2>(g, src, dy, co_rows, ldy, slabs, slab_capacity_elems, num_cu, nsplit_out, force, st, pro);
  return launch_wgrad_halo<128, 64, 1>(g, src, dy, co_rows, ldy, slabs, slab_capacity_elems, num_cu, nsplit_out, force, st, pro);
}

}  // namespace dei2i
