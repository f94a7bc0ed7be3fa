// Halo-resident conv (bf16, stride 1, up to 3x3 taps): forward convs and zero-boundary dgrads of the 3x3 layers.
//
// The gather GEMM (conv_gemm_v2.hip) re-fetches the activation tile for every tap: 9 x (256 rows x 128 B) per
// 64-channel slice, so its k-step moves 48 KB L2->LDS for 4.2 MFLOP and the loop runs at the operand-delivery rate
// (~21 B/clk/CU), not at the MFMA rate.  Here a workgroup owns an 8 x 32 pixel output tile of ONE image and keeps the
// (8+th-1) x (32+tw-1) input halo of the current 64-channel slice in LDS (<= 340 pixels x 128 B = 43.5 KB, double
// buffered); all th*tw taps read their A fragments from that halo at shifted pixel addresses.  Per k-step only the
// weight tile (BN x 128 B) comes from L2: 16 KB + 4.8 KB of amortised halo for the same 4.2 MFLOP -- 2.3x less
// operand traffic, which also frees LDS for a 4-stage weight ring (three k-steps of prefetch; the tile sweep found deeper rings no better).
//
//   k-step order : 64-channel slice outer, tap inner (weights are packed [Cout][tap][Cs]: k = tap*Cs + slice*64)
//   LDS          : halo[2] (2 x 44,032 B) | weight ring STAGES x BN x 128 B | halo source-offset table
//   LDS image    : 128-byte rows, 16-byte chunk c of row r at slot c ^ ((r>>1)&7) (conflict-free ds_read_b128);
//                  LDS-DMA lands lane-linear, so the swizzle is applied to the SOURCE chunk each lane fetches
//   waves        : 8 (4 x 2): wave (wm, wn) owns tile rows 2wm, 2wm+1 (64 pixels) x BN/2 output channels
//   halo pixel   : output (py, px), tap (ty, tx) reads halo pixel (py + dy(ty), px + dx(tx)), dy = ty for ys = +1 and
//                  th-1-ty for ys = -1 (dgrad: the tap sign flip is the kernel flip, geom.h)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

#include "common.h"
#include "geom.h"
#include "launch.h"

namespace dei2i {

__device__ __attribute__((aligned(256))) unsigned char g_zero_page_halo[256];
extern int g_v2_ablate;
extern unsigned long long* g_v2_dbg;
int g_halo_mfma32 = 0;
int g_halo_bn = 0, g_halo_stages = 0;      // tile-size sweep options (0 = the shipped choice)

typedef __attribute__((address_space(3))) void lds_void_h;
typedef __attribute__((address_space(1))) const void gbl_void_h;

DEI2I_D void glds16h(const void* gptr, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gbl_void_h*)gptr, (lds_void_h*)lds_wave_base, 16, 0, 0);
}

DEI2I_D int xcd_remap_h(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

constexpr int HALO_TH = 8, HALO_TW = 32;
constexpr int HALO_GROUPS = 43;                       // 8-pixel LDS-DMA groups per halo slice (344 >= 10*34 pixels)
constexpr int HALO_BYTES = HALO_GROUPS * 8 * 128;     // 44,032
constexpr int HALO_HL = 6;                            // halo LDS-DMA instructions per wave per slice (48 >= 43 groups)

template <int N> DEI2I_D void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// FP8: the operands are e4m3 bytes and the descriptor / pointers describe them as PAIRS (a "bf16" tensor with half the
// channels), so every address, LDS image and DMA pattern is unchanged; a 16-byte fragment then holds 16 k-values and
// feeds two v_mfma_f32_16x16x32_fp8_fp8 (low / high 8 bytes -- the same k permutation for both operands), and the
// epilogue multiplies by *dequant = 1 / (activation scale * weight scale).  Half the LDS and DMA bytes per FLOP.
// PRO: the conv's input is normalised + activated on the operand path (ConvPro, geom.h): once a halo slice has landed in
// LDS every thread rewrites its share of it in place, z = act(A*x + B) with this image's per-channel coefficients (LDS
// table), spread over the M phases of taps 4..6 of the previous slice (slice 0: in the prologue); halo pixels on the
// image's 2-pixel frame come from the pre-normalised ring tensor and are left alone, zero padding stays zero.
// stats != nullptr: the epilogue also writes this tile's per-channel sum / sum of squares of the STORED (rounded) outputs,
// record (n * tiles_per_image + tile) of a (N, tiles, 2, ldc) fp32 tensor -- the layout of moments_partial (reduce.hip),
// so the BatchNorm / InstanceNorm finalize kernels read it unchanged and the separate statistics pass is gone.
template <int BN, int STAGES, bool DIAG, bool M16, bool FP8 = false, bool PRO = false>
__global__ __launch_bounds__(512) void halo_conv_kernel(const GatherDesc g, const bf16_t* __restrict__ src,
                                                        const bf16_t* __restrict__ wgt, const int wrows,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                        const int ldc, const int act, const int tiles_n, const int ablate,
                                                        unsigned long long* __restrict__ dbg,
                                                        const float* __restrict__ dequant, const ConvPro pro,
                                                        float* __restrict__ stats) {
  static_assert(!FP8 || M16, "the fp8 variant uses the 16x16x32 shape");
  static_assert(!PRO || (!FP8 && !DIAG), "operand-path normalisation: bf16 production variant only");
  constexpr int BM = HALO_TH * HALO_TW;             // 256 output pixels
  constexpr int WN = 2, WTN = BN / WN, TM = 2, TN = WTN / 32;
  constexpr int LB = BN / 64;                       // weight LDS-DMA instructions per wave per stage
  constexpr int B_STAGE = BN * 128;
  constexpr int AHEAD = STAGES - 1;
  // A 3-deep ring was tried in the tile sweep: it ran 3 % faster and produced WRONG results in a fraction of the launches
  // (tests/diag_ring_depth.py) -- with two wave groups in anti-phase a stage is re-issued while the trailing group can still
  // be reading it.  4 is the minimum the schedule below is correct for.
  static_assert(TN >= 1 && STAGES >= 4, "tile shape");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned long long kt0 = ablate == 5 ? __builtin_amdgcn_s_memtime() : 0ull;
  unsigned char* const halo = smem;
  unsigned char* const ring = smem + 2 * HALO_BYTES;
  int* const htab = reinterpret_cast<int*>(ring + STAGES * B_STAGE);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int bid = xcd_remap_h(blockIdx.x, gridDim.x);
  const int tile_n = bid % tiles_n, tile_m = bid / tiles_n;
  const int tiles_x = g.Wo / HALO_TW, tiles_y = g.Ho / HALO_TH;
  const int img = tile_m / (tiles_x * tiles_y);
  const int trem = tile_m - img * (tiles_x * tiles_y);
  const int y0 = (trem / tiles_x) * HALO_TH, x0 = (trem % tiles_x) * HALO_TW;
  const int n0 = tile_n * BN;

  const int hwd = HALO_TW + g.tw - 1;                // halo width in pixels
  const int npix = (HALO_TH + g.th - 1) * hwd;
  const int hy0 = y0 + g.by0 + (g.ys < 0 ? -(g.th - 1) : 0);
  const int hx0 = x0 + g.bx0 + (g.xs < 0 ? -(g.tw - 1) : 0);

  // ---- halo source-offset table (element offset of each halo pixel's channel 0, -1 = contributes zero;
  //      PRO: <= -2 = element offset -2 - off into the pre-normalised ring tensor) ----
  for (int p = tid; p < HALO_GROUPS * 8; p += 512) {
    int off = -1;
    if (p < npix) {
      const int hy = p / hwd, hx = p - hy * hwd;
      const int y = bound_coord(hy0 + hy, g.Hl, g.pad_mode);
      const int x = bound_coord(hx0 + hx, g.Wl, g.pad_mode);
      if ((y | x) >= 0) {
        if (PRO && pro.ring != nullptr && !(ring_interior(y, g.Hl) && ring_interior(x, g.Wl)))
          off = -2 - (img * pro.ring_pix + ring_index(y, x, g.Hl, g.Wl)) * g.Cs;
        else
          off = ((img * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up)) * g.Cs;
      }
    }
    htab[p] = off;
  }
  float* const pcoef = reinterpret_cast<float*>(htab + HALO_GROUPS * 8);     // PRO: A[Cs] | B[Cs] of this image
  if constexpr (PRO) {
    for (int i = tid; i < 2 * g.Cs; i += 512) {
      const int which = i >= g.Cs ? 1 : 0;
      pcoef[i] = (which ? pro.B : pro.A)[(size_t)img * pro.n_stride + (i - which * g.Cs)];
    }
  }
  __syncthreads();

  // ---- per-lane LDS-DMA roles ----
  const int lrow = lane >> 3, lslot = lane & 7;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_halo);
  int h_off[HALO_HL], h_group[HALO_HL];
#pragma unroll
  for (int j = 0; j < HALO_HL; ++j) {
    int grp = j * 8 + wave;                          // 48 slots for 43 groups: the surplus re-fetches groups 0..4
    if (grp >= HALO_GROUPS) grp -= HALO_GROUPS;      // (identical bytes land twice; keeps vmcnt uniform across waves)
    const int pix = grp * 8 + lrow;
    const int o = htab[pix];
    const int so = (lslot ^ ((pix >> 1) & 7)) << 3;
    h_group[j] = grp;
    h_off[j] = o >= 0 ? o + so : (o == -1 ? -1 : o - so);
  }
  const bf16_t* b_ptr[LB];
#pragma unroll
  for (int j = 0; j < LB; ++j) {
    const int r = j * 64 + wave * 8 + lrow;
    const int n = n0 + r;
    b_ptr[j] = n < wrows ? wgt + (size_t)n * g.K + ((lslot ^ ((r >> 1) & 7)) << 3) : nullptr;
  }

  const int ntaps = __builtin_amdgcn_readfirstlane(g.th * g.tw);
  const int nslices = __builtin_amdgcn_readfirstlane(g.Cs >> 6);
  const int nk = ntaps * nslices;

  auto issue_halo = [&](int slice) {
    unsigned char* hb = halo + (slice & 1) * HALO_BYTES;
    const int ci0 = slice << 6;
#pragma unroll
    for (int j = 0; j < HALO_HL; ++j) {
      const bf16_t* p = h_off[j] >= 0 ? src + ((size_t)(unsigned)h_off[j] + (unsigned)ci0) : zero;
      if constexpr (PRO) {
        if (h_off[j] < -1) p = pro.ring + ((size_t)(unsigned)(-2 - h_off[j]) + (unsigned)ci0);
      }
      glds16h(p, hb + h_group[j] * 1024);
    }
  };
  // PRO: in-place normalisation of a landed halo slice.  Vector v = tid + 512 j is LDS slot (tid & 7) of halo pixel
  // (tid >> 3) + 64 j, i.e. logical 16-byte chunk (tid & 7) ^ ((tid >> 4) & 7) for EVERY j: a thread's 8 channels -- and
  // its 16 coefficients -- are fixed within a slice.
  const int t_slot = tid & 7, t_chunk = t_slot ^ ((tid >> 4) & 7), t_pix0 = tid >> 3;
  auto transform = [&](int tslice, int jbeg, int jend) {
    unsigned char* hb = halo + (tslice & 1) * HALO_BYTES;
    const float* ca = pcoef + (tslice << 6) + t_chunk * 8;
    float A8[8], B8[8];
    ldcoef<8>(ca, A8);
    ldcoef<8>(ca + g.Cs, B8);
    for (int j = jbeg; j < jend; ++j) {
      const int pix = t_pix0 + 64 * j;
      if (pix < npix && htab[pix] >= 0) {
        u32x4* p = reinterpret_cast<u32x4*>(hb + pix * 128 + t_slot * 16);
        float f[8];
        Elem<bf16_t>::unpack(*p, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = fmaf(A8[e], f[e], B8[e]);
          f[e] = fmaf(pro.slope, fminf(v, 0.f), fmaxf(v, 0.f));
        }
        *p = Elem<bf16_t>::pack(f);
      }
    }
  };
  int is_tap = 0, is_slice = 0;                      // (tap, slice) of the next weight k-step to issue
  auto issue_b = [&](int stage) {
    unsigned char* sb = ring + stage * B_STAGE;
    const int kb = is_tap * g.Cs + (is_slice << 6);
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const bf16_t* p = b_ptr[j] != nullptr ? b_ptr[j] + kb : zero;
      glds16h(p, sb + (j * 64 + wave * 8) * 128);
    }
    if (++is_tap == ntaps) { is_tap = 0; ++is_slice; }
  };

  // MFMA shape: 32x32x16 (M16 = false) or 16x16x32 (M16 = true; same cycles per FLOP, but the chip holds a higher clock
  // under it -- mfma_peak.hip: 2.26 vs 1.94 PF, 2.2 vs 1.9 GHz).  Both run the transposed product (A = weights,
  // B = pixels): D[channel][pixel] leaves four consecutive channels of one pixel in consecutive registers.
  constexpr int PB = M16 ? TM * 2 : TM;              // pixel blocks per wave (16 or 32 pixels each)
  constexpr int CB = M16 ? WTN / 16 : TN;            // channel blocks per wave
  constexpr int NKS = M16 ? 2 : 4;                   // MFMA k-blocks per 64-channel k-step
  typedef __attribute__((ext_vector_type(M16 ? 4 : 16))) float acc_t;
  acc_t acc[PB][CB];
#pragma unroll
  for (int i = 0; i < PB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int e = 0; e < (M16 ? 4 : 16); ++e) acc[i][j][e] = 0.f;

  const int lr = lane & 31, lh = lane >> 5;          // 32x32x16 lane roles: row / k-half
  const int l16 = lane & 15, kg = lane >> 4;         // 16x16x32 lane roles: row / k-group (8 channels each)

  // ---- fragment reads: ALL k-blocks of a k-step live in registers (Frags); the reads of k-step j are issued in this
  //      wave's M phase, its MFMAs in the C phase ----
  struct Frags { u32x4 a[NKS][PB]; u32x4 b[NKS][CB]; };
  int a_pix0[PB];                                    // halo pixel of this lane's pixel row for tap offset 0
#pragma unroll
  for (int i = 0; i < PB; ++i)
    a_pix0[i] = M16 ? (wm * TM + (i >> 1)) * hwd + (i & 1) * 16 + l16 : (wm * TM + i) * hwd + lr;
  int b_lane[CB], b_rsw[CB];
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int row = M16 ? wn * WTN + j * 16 + l16 : wn * WTN + j * 32 + lr;
    b_lane[j] = row * 128;
    b_rsw[j] = (row >> 1) & 7;
  }
  // state of the k-step whose fragments are being LOADED
  int ld_tx = 0, ld_ty = 0, ld_slice = 0, ld_stage = 0;
  const int step_x = g.xs > 0 ? 1 : -1, step_y = g.ys > 0 ? hwd : -hwd;
  int ld_toff = (g.ys > 0 ? 0 : (g.th - 1) * hwd) + (g.xs > 0 ? 0 : g.tw - 1);     // halo pixel offset of tap (0,0)
  const int toff_row_wrap = step_y - (g.tw - 1) * step_x;                            // (ty,tw-1) -> (ty+1,0)
  const int toff_origin = ld_toff;
  const unsigned char* la_base[PB];
  int la_swz[PB];
  const unsigned char* lb_base;
  auto prep_load = [&]() {                            // addresses for the k-step (ld_tap, ld_slice, ld_stage)
    const unsigned char* hb = halo + (ld_slice & 1) * HALO_BYTES;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int pix = a_pix0[i] + ld_toff;
      la_base[i] = hb + pix * 128;
      la_swz[i] = (pix >> 1) & 7;
    }
    lb_base = ring + ld_stage * B_STAGE;
  };
  auto advance_load = [&]() {
    if (++ld_tx == g.tw) {
      ld_tx = 0;
      if (++ld_ty == g.th) { ld_ty = 0; ++ld_slice; ld_toff = toff_origin; }
      else ld_toff += toff_row_wrap;
    } else {
      ld_toff += step_x;
    }
    if (++ld_stage == STAGES) ld_stage = 0;
  };
  auto read_frags = [&](Frags& f, int ks) {
    const int chunk = M16 ? ks * 4 + kg : ks * 2 + lh;          // this lane's 16-byte chunk of the 128-byte k-row
#pragma unroll
    for (int i = 0; i < PB; ++i) f.a[ks][i] = *reinterpret_cast<const u32x4*>(la_base[i] + ((chunk ^ la_swz[i]) << 4));
#pragma unroll
    for (int j = 0; j < CB; ++j) f.b[ks][j] = *reinterpret_cast<const u32x4*>(lb_base + b_lane[j] + ((chunk ^ b_rsw[j]) << 4));
  };
  auto mfma_group = [&](const Frags& f, int ks) {
#pragma unroll
    for (int i = 0; i < PB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        if constexpr (FP8) {
          const u32x4 bq = f.b[ks][j], aq = f.a[ks][i];
          const long b0 = (long)(((unsigned long long)bq.y << 32) | bq.x), b1 = (long)(((unsigned long long)bq.w << 32) | bq.z);
          const long a0 = (long)(((unsigned long long)aq.y << 32) | aq.x), a1 = (long)(((unsigned long long)aq.w << 32) | aq.z);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b0, a0, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(b1, a1, acc[i][j], 0, 0, 0);
        } else if constexpr (M16)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.b[ks][j]),
                                                               __builtin_bit_cast(bf16x8, f.a[ks][i]), acc[i][j], 0, 0, 0);
        else
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.b[ks][j]),
                                                               __builtin_bit_cast(bf16x8, f.a[ks][i]), acc[i][j], 0, 0, 0);
      }
  };

  // ---- main loop: two wave groups in anti-phase ("ping-pong") ----
  // With one barrier per k-step all eight waves run in lockstep: both waves of a SIMD issue their LDS reads / address
  // math / LDS-DMA at the same time and the MFMA pipe idles meanwhile (measured: 1 990 cycles per k-step against 1 024
  // of MFMA issue).  Here every k-step of a wave is two phases -- M(j): issue LDS-DMA, read the 16 fragments of k-step j
  // into registers; C(j): 16 MFMAs, nothing else -- each closed by a workgroup barrier, and waves 4..7 run one phase
  // behind waves 0..3 (one extra barrier up front).  Every SIMD hosts one wave of each group, so in every phase one of
  // its waves owns the MFMA pipe while the other owns the LDS / VALU / DMA issue slots.
  //
  //   phase:      P0     P1     P2     P3     P4 ...
  //   group 0:    M(0)   C(0)   M(1)   C(1)   M(2)
  //   group 1:    -      M(0)   C(0)   M(1)   C(1)
  //
  // LDS-DMA protocol (every wave fetches its own share of each weight stage / halo slice):
  //   C(j) issues weights(j-1+STAGES) into stage (j-1) % STAGES: its last reader was group 1's M(j-1), two phases before
  //        group 0's C(j).  At tap 1 of a slice it also issues the NEXT slice's halo (into the buffer of the previous
  //        slice, last read by group 1 at that slice's last tap).  (Measured alternatives, res-block shape, cycles per
  //        k-step: DMA issue in M 2 695; in C after the first MFMA group 1 658; staggered per wave 2 048; halo spread one
  //        instruction per tap 1 953; one barrier per k-step without phases 1 990.)
  //   M(j) ends with "my share of weights(j+1) has landed" + barrier: the earliest reader of stage j+1 is group 0's
  //        M(j+1), which starts after the barrier that closes group 1's M(j).  Halo slices are issued >= 4 k-steps
  //        before that and are older than the weights waited for, so they have landed too.
  const int grp = wave >> 2;
  int tap = 0, slice = 0;                             // k-step of this wave's current M / C phase
  Frags f0, f1;
  unsigned long long dg[6] = {0, 0, 0, 0, 0, 0};      // DIAG: cycles in [issue | reads issued | vmcnt | lgkmcnt | barrier M | C+barrier]
  auto now = [&]() -> unsigned long long {
    if (!DIAG) return 0ull;
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return t;
  };
  auto phase_m = [&](Frags& f, int j) {
    const unsigned long long q0 = now();
    prep_load();
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) read_frags(f, ks);
    advance_load();
    if constexpr (PRO) {
      // the next slice's halo (issued in C(tap 1)) has landed for EVERY wave once the barrier of the trailing group's
      // M(tap 3) has passed; its first reader is the leading group's M(tap 0) of the next slice, after this group's
      // M(tap 6) barrier in both groups' phase order
      if (slice + 1 < nslices && tap >= 4 && tap <= 6) transform(slice + 1, (tap - 4) * 2, (tap - 4) * 2 + 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long q2 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    // my share of weights(j+1) [issued in C(j-2)] must have landed; younger: weights(j+2) [C(j-1)] and the halo slice
    // C(j-1) issued at tap 1 (this k-step is then tap 2)
    const bool halo_young = tap == 2 && slice + 1 < nslices;
    if (j + 1 >= nk) wait_vm<0>();
    else if (j + 2 < nk) { if (halo_young) wait_vm<LB + HALO_HL>(); else wait_vm<LB>(); }
    else { if (halo_young) wait_vm<HALO_HL>(); else wait_vm<0>(); }
    const unsigned long long q3 = DIAG ? __builtin_amdgcn_s_memtime() : 0ull;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long q4 = now();
    __builtin_amdgcn_s_barrier();
    if (DIAG) {
      const unsigned long long q5 = now();
      dg[1] += q2 - q0; dg[2] += q3 - q2; dg[3] += q4 - q3; dg[4] += q5 - q4;
    }
  };
  // C(j): the 16 MFMAs, with this wave's LDS-DMA issue in the gaps (an MFMA holds the vector issue 8 of its 32 cycles;
  // a 16-byte-per-lane load instruction occupies the CU's address path ~64 cycles, the hard floor of this kernel:
  // 20.8 KB per k-step = 1 300 cycles)
  auto phase_c = [&](const Frags& f, int j) {
    const unsigned long long q0 = now();
    mfma_group(f, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (j - 1 + STAGES < nk) issue_b((j + STAGES - 1) % STAGES);
    __builtin_amdgcn_sched_barrier(0);
    if (tap == 1 && slice + 1 < nslices) issue_halo(slice + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 1; ks < NKS; ++ks) mfma_group(f, ks);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (DIAG) dg[5] += now() - q0;
    if (++tap == ntaps) { tap = 0; ++slice; }
  };

  const unsigned long long st0 = ablate == 5 ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long sr0 = ablate == 5 ? __builtin_amdgcn_s_memrealtime() : 0ull;
  // prologue: halo slice 0 and weights(0 .. STAGES-2); weights(STAGES-1) is issued by C(0)
  issue_halo(0);
  for (int s2 = 0; s2 < STAGES - 1; ++s2)
    if (s2 < nk) issue_b(s2);
  {
    const int nb = min(STAGES - 2, max(nk - 1, 0));      // stages younger than weights(0)
    if (nb >= 2) wait_vm<2 * LB>(); else if (nb == 1) wait_vm<LB>(); else wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();                          // halo 0 and weights(0) complete for every wave
  if constexpr (PRO) {
    transform(0, 0, 6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // slice 0 normalised for every wave
  }
  if (grp == 1) __builtin_amdgcn_s_barrier();            // group 1 starts one phase late
  // two k-steps per trip (two fragment register sets); an odd last k-step leaves through the break, so the phase code
  // exists twice, not three times (a dispatch walks its code cold: code size is start-up latency)
#pragma unroll 1
  for (int j = 0; j < nk; j += 2) {
    phase_m(f0, j);
    phase_c(f0, j);
    if (j + 1 >= nk) break;
    phase_m(f1, j + 1);
    phase_c(f1, j + 1);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();            // pairs with group 1's last phase
  if (ablate == 5 && dbg != nullptr && lane == 0) {      // diagnostic: loop cycles and the clock held
    unsigned long long* d = dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
    d[0] = __builtin_amdgcn_s_memtime() - st0;
    d[1] = __builtin_amdgcn_s_memrealtime() - sr0;
    d[2] = (unsigned long long)nk;
    d[3] = st0;
    if (DIAG) {
      unsigned long long* e = dbg + (size_t)gridDim.x * 8 * 4 + ((size_t)blockIdx.x * 8 + wave) * 6;
      for (int q = 0; q < 6; ++q) e[q] = dg[q];
    }
  }

  // ---- epilogue: D row = channel (e&3) + 8(e>>2) + 4lh of the 32-block, col = pixel lr.  Stage the tile through LDS
  //      as bf16 [pixel][BN (+8 pad)] with 8-byte writes (row stride 16*odd bytes: 2-way instead of 32-way conflicts),
  //      write back with 16-byte stores ----
  __syncthreads();
  constexpr int CROW = BN * 2 + 16;       // staging row stride in bytes
  unsigned char* ctile = smem;
  // Branch-free and small: act(v) = max(v,0) + slope*min(v,0) (slope 0 / 0.2 / 1 for ReLU / LeakyReLU / none -- exact for
  // all three), padded output channels (n >= wrows) cleared with a bit mask on the packed pair.  The unrolled
  // per-element `act` switch this replaces was 60 % of the kernel's code, and a dispatch walks its code cold.
  const float slope = act_slope(act);
  const float dq = FP8 ? dequant[0] : 1.f;
  if constexpr (M16) {
    // D row = channel 4*kg + e of its 16-block, col = pixel l16
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int col0 = wn * WTN + j * 16 + 4 * kg;
      float bq[4];
      uint32_t m01, m23;
      epi_col_consts(bias, n0 + col0, wrows, bq, m01, m23);
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int row = (wm * TM + (i >> 1)) * 32 + (i & 1) * 16 + l16;
        *reinterpret_cast<u32x2*>(ctile + row * CROW + col0 * 2) =
            epi_finish4(acc[i][j][0] * dq, acc[i][j][1] * dq, acc[i][j][2] * dq, acc[i][j][3] * dq, bq, slope, m01, m23);
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col0 = wn * WTN + j * 32 + 8 * q + 4 * lh;             // first of this lane's four channels
        float bq[4];
        uint32_t m01, m23;
        epi_col_consts(bias, n0 + col0, wrows, bq, m01, m23);
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          const int row = (wm * TM + i) * 32 + lr;
          *reinterpret_cast<u32x2*>(ctile + row * CROW + col0 * 2) =
              epi_finish4(acc[i][j][q * 4], acc[i][j][q * 4 + 1], acc[i][j][q * 4 + 2], acc[i][j][q * 4 + 3], bq, slope, m01, m23);
        }
      }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;             // 16-byte chunks per tile row
  constexpr int RPP = 512 / CPR;          // rows per pass
  const int chunk = tid % CPR, rsub = tid / CPR;
  const int ncol = n0 + chunk * 8;
  float st8[16];                                        // stats: sum (0..7) / sum of squares (8..15) of this thread's 8 channels
#pragma unroll
  for (int k = 0; k < 16; ++k) st8[k] = 0.f;
  if (ncol < ldc) {
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
      const int row = p * RPP + rsub;
      const size_t opix = (size_t)out_pixel(g, img, y0 + (row >> 5), x0 + (row & 31));
      const u32x4 v = *reinterpret_cast<const u32x4*>(ctile + row * CROW + chunk * 16);
      *reinterpret_cast<u32x4*>(out + opix * ldc + ncol) = v;
      if (stats != nullptr) {
        float f[8];
        Elem<bf16_t>::unpack(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { st8[e] += f[e]; st8[8 + e] = fmaf(f[e], f[e], st8[8 + e]); }
      }
    }
  }
  if (stats != nullptr) {                                // kernel-uniform
    // per-thread partials -> the (free) weight-ring area [rsub][chunk][16] -> ordered sums over the RPP row groups
    float* red = reinterpret_cast<float*>(smem + 2 * HALO_BYTES);
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
      if (c < ldc) stats[((size_t)tile_m * 2 + (k >> 3)) * ldc + c] = sum;
    }
  }
  if (ablate == 5 && dbg != nullptr && lane == 0) {      // diagnostic: cycles from kernel entry to the last store issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    dbg[(size_t)gridDim.x * 8 * 4 + ((size_t)blockIdx.x * 8 + wave) * 6] = __builtin_amdgcn_s_memtime() - kt0;
  }
}

template <int BN, int STAGES, bool FULL = true>      // FULL = false: a sweep instance (tile-size report), default variant only
static hipError_t launch_halo(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out,
                              int ldc, int act, const float* dequant, hipStream_t st, const ConvPro* pro = nullptr,
                              float* stats = nullptr) {
  const int tiles_m = g.N * (g.Ho / HALO_TH) * (g.Wo / HALO_TW);
  const int tiles_n = (ldc + BN - 1) / BN;
  const size_t lds = 2 * (size_t)HALO_BYTES + (size_t)STAGES * BN * 128 + HALO_GROUPS * 8 * sizeof(int) +
                     (pro != nullptr ? 2 * (size_t)g.Cs * sizeof(float) : 0);
  if (lds > 160 * 1024) return hipErrorNotSupported;
  auto kern = halo_conv_kernel<BN, STAGES, false, true>;
  if constexpr (FULL) {
    if (g_halo_mfma32) kern = halo_conv_kernel<BN, STAGES, false, false>;     // A/B option: 32x32x16 MFMAs
    if (g_v2_ablate == 6) kern = halo_conv_kernel<BN, STAGES, true, true>;   // diagnostic build: per-phase cycle stamps
    if (dequant != nullptr) kern = halo_conv_kernel<BN, STAGES, false, true, true>;   // e4m3 operands
    if (pro != nullptr) {
      if (dequant != nullptr) return hipErrorNotSupported;
      kern = halo_conv_kernel<BN, STAGES, false, true, false, true>;           // operand-path normalisation
    }
  } else if (pro != nullptr) {
    return hipErrorNotSupported;
  }
  const ConvPro pv = pro != nullptr ? *pro : ConvPro{nullptr, nullptr, 0, 0.f, nullptr, 0};
  {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  // (fp8: g describes byte pairs, so Clog is half the channel count)
  count_launch(dequant != nullptr ? K_HALO_CONV_FP8 : K_HALO_CONV);
  prof_begin(PROF_HALO_CONV, (dequant != nullptr ? 4.0 : 2.0) * (double)g.M * (double)(g.th * g.tw) * (double)g.Clog * (double)wrows, st);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), lds, st, g, (const bf16_t*)src, (const bf16_t*)wgt, wrows, bias,
                     (bf16_t*)out, ldc, act, tiles_n, g_v2_ablate == 6 ? 5 : g_v2_ablate, g_v2_dbg, dequant, pv, stats);
  prof_end(PROF_HALO_CONV, st);
  return hipGetLastError();
}

// returns hipErrorNotSupported when the shape does not qualify (the caller falls through to the gather GEMMs)
// dequant != nullptr selects the fp8 variant (g describes the e4m3 operands as byte pairs: see the kernel)
// pro / stats: see the kernel (operand-path normalisation of the input, statistics of the output from the epilogue)
hipError_t halo_conv(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out, int ldc,
                     int act, int num_cu, hipStream_t st, const float* dequant, const ConvPro* pro, float* stats) {
  if (g.sh != 1 || g.sw != 1 || (g.ys != 1 && g.ys != -1) || (g.xs != 1 && g.xs != -1)) return hipErrorNotSupported;
  if (g.th != 3 || g.tw != 3 || g.wK != g.K || g.wtw != g.tw) return hipErrorNotSupported;       // tap count >= ring depth + 1 (halo issue at tap 1)
  if (g.Cs % 64 != 0 || g.Ho % HALO_TH != 0 || g.Wo % HALO_TW != 0 || g.M != g.N * g.Ho * g.Wo) return hipErrorNotSupported;
  if ((long long)g.N * g.Hs * g.Ws * g.Cs >= (1ll << 31)) return hipErrorNotSupported;   // 32-bit offset table
  if (ldc < 64 || ldc % 8 != 0) return hipErrorNotSupported;
  const int tiles_m = g.N * (g.Ho / HALO_TH) * (g.Wo / HALO_TW);
  if (pro != nullptr && (g.Cs > 512 || (pro->ring != nullptr && (g.Hl < 4 || g.Wl < 4)))) return hipErrorNotSupported;
  if (dequant == nullptr && pro == nullptr && stats == nullptr && (g_halo_bn != 0 || g_halo_stages != 0)) {
    // tile-size sweep (options halo_bn / halo_stages; profiles/sweep_tiles.py): output-channel tile width x weight-ring
    // depth.  LDS = 88,064 B of halo + STAGES x BN x 128 B of ring (+ the offset table): 128x4 = 153.6 KB is the
    // largest 128-wide one that fits the 160 KB, 64-wide tiles leave room for rings up to 8 deep.
    const int bn = (g_halo_bn == 64 || ldc < 128) ? 64 : 128;
    int stg = g_halo_stages != 0 ? g_halo_stages : 4;
    if (bn == 128 && stg > 4) stg = 4;          // deeper rings do not fit beside a 128-wide tile
    if (tiles_m * ((ldc + bn - 1) / bn) < num_cu / 2) return hipErrorNotSupported;
    if (bn == 128 && stg == 4) return launch_halo<128, 4>(g, src, wgt, wrows, bias, out, ldc, act, dequant, st);
    if (bn == 64 && stg == 4) return launch_halo<64, 4>(g, src, wgt, wrows, bias, out, ldc, act, dequant, st);
    if (bn == 64 && stg == 6) return launch_halo<64, 6, false>(g, src, wgt, wrows, bias, out, ldc, act, dequant, st);
    if (bn == 64 && stg == 8) return launch_halo<64, 8, false>(g, src, wgt, wrows, bias, out, ldc, act, dequant, st);
    return hipErrorInvalidValue;
  }
  if (ldc >= 128) {
    if (tiles_m * ((ldc + 127) / 128) < num_cu / 2) return hipErrorNotSupported;   // small grids: split-K v1 fills the chip better
    return launch_halo<128, 4>(g, src, wgt, wrows, bias, out, ldc, act, dequant, st, pro, stats);
  }
  if (tiles_m < num_cu / 2) return hipErrorNotSupported;
  return launch_halo<64, 4>(g, src, wgt, wrows, bias, out, ldc, act, dequant, st, pro, stats);
}

}  // namespace dei2i
