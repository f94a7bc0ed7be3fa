// SPADE's label path, second stage, for ALL modules of a generator in one launch per direction.
//
// normalization.py:17-37: every SPADE module turns the (nearest-resized) label map into gamma | beta with two 3x3 zero-padded convs,
//   actv = ReLU(conv(seg; mlp_shared)),  gamma = conv(actv; mlp_gamma),  beta = conv(actv; mlp_beta).
// With a constant label map the result takes 5 x 5 distinct values per (sample, channel) (position relative to the 2-pixel border:
// networks/architecture.py SPADE), so the convs run on a 5 x 5 "class image" -- 25 pixels per sample.  At that size a conv launch is
// all latency: the generic gather GEMM took ~18 us (+ a split-K finalize) for 1.9 GFLOP, ten modules, forward / input gradient /
// weight gradient / bias gradient each -- 0.9 ms of a 35 ms step for work that is worth a few microseconds.
//
// Here the first stage is one conv over the modules' concatenated filters (networks/generator.py prime_spade: same input), and the
// second stage -- hidden -> 2C per module, each module reading ITS channel slice of that one activation tensor -- is these kernels,
// every module in the same launch (a table of per-module pointers travels as a kernel argument):
//   label_gb_pack   gamma | beta filters (fp32 OIHW) -> bf16 [2C][9][hidden] (forward) and [hidden][9][2C] (input gradient)
//   label_gb_fwd    gb_m[n, p, :] = bias_m + sum_{tap, ci} actv[n, p + tap - 1, off_m + ci] * Wf_m[:, tap, ci]
//   label_gb_dgrad  dactv[n, q, off_m + ci] = sum_{tap, co} dgb_m[n, q - tap + 1, co] * Wd_m[ci, tap, co]
//   label_gb_wgrad  dW_m[co, ci, tap] = sum_{n, p} dgb_m[n, p, co] * actv[n, p + tap - 1, off_m + ci];  dbias_m[co] = sum dgb_m
// 16x16x32 MFMAs straight from global memory (everything is L2-resident: 25 pixels x 256 B per sample and module) for the first three;
// the weight gradient reduces over PIXELS, which are the strided dimension of an NHWC tensor: its tiles go through LDS and come back
// transposed (ds_read_b64_tr_b16) as 32x32x16 fragments, like csrc/wgrad_halo.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dei2i_hip.h"
#include "common.h"
#include "launch.h"

namespace dei2i {

constexpr int LP_MAXMOD = 16;
constexpr int LP_PIX = 25;          // the 5 x 5 class image

struct LabelMod {
  const float* gw;        // mlp_gamma.weight [C][hidden][3][3]
  const float* bw;        // mlp_beta.weight
  const float* gbias;     // [C]
  const float* bbias;
  bf16_t* wf;             // packed forward filters [2C][9][hidden]
  bf16_t* wd;             // packed input-gradient filters [hidden][9][2C]
  bf16_t* gb;             // forward output / (const) incoming gradient (N, 25, 2C)
  float* dgw;             // weight / bias gradients (fp32, the parameters' layout)
  float* dbw;
  float* dgbias;
  float* dbbias;
  int C;                  // norm_nc: gamma = channels [0, C), beta = [C, 2C)
  int in_off;             // this module's first channel in the activation tensor
  int blk0;               // first workgroup (x) of this module in the launch
  int live;               // dgrad / wgrad: the module has an incoming gradient (else zeros)
};

struct LabelPack {
  LabelMod m[LP_MAXMOD];
  int n;
  int hidden;             // channels per module in the activation tensor (multiple of 32, <= 128)
  int ctot;               // channel stride of the activation tensor
  int N;
};

DEI2I_D int lp_module_of(const LabelPack& pk, int bx) {
  int m = 0;
#pragma unroll 1
  for (int i = 1; i < pk.n; ++i)
    if (bx >= pk.m[i].blk0) m = i;
  return m;
}

DEI2I_D u32x4 lp_ld16(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }

// ---- filters -> the two bf16 layouts; one thread per 8 consecutive elements of either output --------------------------------
__global__ __launch_bounds__(256) void label_gb_pack_kernel(const LabelPack pk) {
  const int m = lp_module_of(pk, blockIdx.x);
  const LabelMod& md = pk.m[m];
  const int C = md.C, C2 = 2 * C, H = pk.hidden;
  const int per = C2 * 9 * H / 8;                      // vectors per layout
  const int v = (blockIdx.x - md.blk0) * 256 + threadIdx.x;
  if (v >= 2 * per) return;
  float f[8];
  if (v < per) {                                       // wf[co][tap][ci .. ci+7]
    const int e = v * 8;
    const int co = e / (9 * H), r = e - co * 9 * H, tap = r / H, ci = r - tap * H;
    const float* src = (co < C ? md.gw + (size_t)co * H * 9 : md.bw + (size_t)(co - C) * H * 9);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = src[(size_t)(ci + k) * 9 + tap];
    *reinterpret_cast<u32x4*>(md.wf + e) = Elem<bf16_t>::pack(f);
  } else {                                             // wd[ci][tap][co .. co+7]  (C is a multiple of 8: a vector stays in gamma or beta)
    const int e = (v - per) * 8;
    const int ci = e / (9 * C2), r = e - ci * 9 * C2, tap = r / C2, co = r - tap * C2;
    const float* src = (co < C ? md.gw + (size_t)co * H * 9 : md.bw + (size_t)(co - C) * H * 9);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = src[((size_t)k * H + ci) * 9 + tap];
    *reinterpret_cast<u32x4*>(md.wd + e) = Elem<bf16_t>::pack(f);
  }
}

// LDS image of the class images of LP_IMG samples WITH their zero ring (7 x 7 pixels each): pixel pitch LP_PITCH bytes (128 channels + 16:
// a 16-pixel fragment read at any tap shift spreads over the banks)
constexpr int LP_IMG = 4;
constexpr int LP_PITCH = 128 * 2 + 16;
constexpr int LP_RING = 49;

// stage `nch` channels (multiple of 8, <= 128) starting at `src + c0` (pixel stride `cstride` elements) of samples n0 .. n0 + LP_IMG - 1
template <int DUMMY = 0>
DEI2I_D void lp_stage_ring(unsigned char* lds, const bf16_t* src, size_t cstride, int c0, int nch, int n0, int N, int tid) {
  const int vpp = nch / 8;                               // vectors per pixel
  const int total = LP_IMG * LP_RING * vpp;
  for (int v0 = tid; v0 < total; v0 += 4 * 256) {        // four loads in flight per thread
    u32x4 val[4];
    int dst[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = v0 + u * 256;
      val[u] = u32x4{0u, 0u, 0u, 0u};
      dst[u] = -1;
      if (v < total) {
        const int pix = v / vpp, c = v - pix * vpp;
        const int i = pix / LP_RING, r = pix - i * LP_RING;
        const int ry = r / 7, rx = r - ry * 7;
        dst[u] = pix * LP_PITCH + c * 16;
        if (n0 + i < N && ry >= 1 && ry <= 5 && rx >= 1 && rx <= 5)
          val[u] = lp_ld16(src + ((size_t)(n0 + i) * LP_PIX + (ry - 1) * 5 + (rx - 1)) * cstride + c0 + c * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (dst[u] >= 0) *reinterpret_cast<u32x4*>(lds + dst[u]) = val[u];
  }
}

// ---- forward: workgroup = (module, 128 output channels) x LP_IMG samples; wave = 32 output channels x LP_IMG samples x 2 pixel blocks ----
__global__ __launch_bounds__(256) void label_gb_fwd_kernel(const LabelPack pk, const bf16_t* __restrict__ actv) {
  __shared__ __attribute__((aligned(16))) unsigned char img[LP_IMG * LP_RING * LP_PITCH];
  const int m = lp_module_of(pk, blockIdx.x);
  const LabelMod& md = pk.m[m];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kg = lane >> 4;
  const int C2 = 2 * md.C, H = pk.hidden;
  const int co0 = (blockIdx.x - md.blk0) * 128 + wave * 32;
  const int n0 = blockIdx.y * LP_IMG;
  lp_stage_ring(img, actv, (size_t)pk.ctot, md.in_off, H, n0, pk.N, tid);
  __syncthreads();
  if (co0 >= C2) return;                               // wave-uniform (after the barrier)
  f32x4 acc[2][LP_IMG][2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < LP_IMG; ++i)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[j][i][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  int rpix[2];                                         // ring-image pixel of this lane's output pixel (pixels 25..31: pixel 24, never stored)
#pragma unroll
  for (int b = 0; b < 2; ++b) { const int p = min(b * 16 + l16, LP_PIX - 1); rpix[b] = (p / 5 + 1) * 7 + (p % 5) + 1; }
  const bf16_t* wrow[2];
  bool wlive[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) { wlive[j] = co0 + j * 16 + l16 < C2; wrow[j] = md.wf + (size_t)(co0 + j * 16 + (wlive[j] ? l16 : 0)) * 9 * H + kg * 8; }
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    const int toff = (tap / 3 - 1) * 7 + (tap - (tap / 3) * 3 - 1);
    u32x4 a[4][2];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        a[k][j] = u32x4{0u, 0u, 0u, 0u};
        if (k * 32 < H && wlive[j]) a[k][j] = lp_ld16(wrow[j] + (size_t)tap * H + k * 32);
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k * 32 < H) {
#pragma unroll
        for (int i = 0; i < LP_IMG; ++i)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(img + (i * LP_RING + rpix[b] + toff) * LP_PITCH + k * 64 + kg * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[j][i][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[k][j]), __builtin_bit_cast(bf16x8, v), acc[j][i][b], 0, 0, 0);
          }
      }
    }
  }
  // D: lane (pixel = l16, kg) holds output channels 4 kg .. + 3 of its 16-channel block
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int co = co0 + j * 16 + 4 * kg;
    if (co >= C2) continue;
    float bias[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) bias[e] = co + e < md.C ? md.gbias[co + e] : md.bbias[co + e - md.C];
#pragma unroll
    for (int i = 0; i < LP_IMG; ++i) {
      const int n = n0 + i;
      if (n >= pk.N) continue;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int p = b * 16 + l16;
        if (p >= LP_PIX) continue;
        u32x2 o;
        o.x = (uint32_t)f32_to_bf16(acc[j][i][b][0] + bias[0]) | ((uint32_t)f32_to_bf16(acc[j][i][b][1] + bias[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(acc[j][i][b][2] + bias[2]) | ((uint32_t)f32_to_bf16(acc[j][i][b][3] + bias[3]) << 16);
        *reinterpret_cast<u32x2*>(md.gb + ((size_t)n * LP_PIX + p) * C2 + co) = o;
      }
    }
  }
}

// ---- input gradient: workgroup = (module, 64 activation channels) x LP_IMG samples; wave = 16 channels; dgb staged 128 channels at a time ----
__global__ __launch_bounds__(256) void label_gb_dgrad_kernel(const LabelPack pk, bf16_t* __restrict__ dactv) {
  __shared__ __attribute__((aligned(16))) unsigned char img[LP_IMG * LP_RING * LP_PITCH];
  const int m = lp_module_of(pk, blockIdx.x);
  const LabelMod& md = pk.m[m];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kg = lane >> 4;
  const int C2 = 2 * md.C, H = pk.hidden;
  const int ci0 = (blockIdx.x - md.blk0) * 64 + wave * 16;
  const int n0 = blockIdx.y * LP_IMG;
  const bool wave_live = ci0 < H;
  f32x4 acc[LP_IMG][2];
#pragma unroll
  for (int i = 0; i < LP_IMG; ++i)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[i][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (md.live) {                                       // workgroup-uniform
    int rpix[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) { const int q = min(b * 16 + l16, LP_PIX - 1); rpix[b] = (q / 5 + 1) * 7 + (q % 5) + 1; }
    const bool wlive = ci0 + l16 < H;
    const bf16_t* wrow = md.wd + (size_t)(ci0 + (wlive ? l16 : 0)) * 9 * C2 + kg * 8;
#pragma unroll 1
    for (int kc = 0; kc < C2; kc += 128) {
      const int nch = min(128, C2 - kc);
      __syncthreads();                                 // the previous chunk's reads are done
      lp_stage_ring(img, md.gb, (size_t)C2, kc, nch, n0, pk.N, tid);
      __syncthreads();
      if (!wave_live) continue;
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {              // gb[p] took actv[p + t]: actv[q] fed gb[q - t]
        const int toff = -((tap / 3 - 1) * 7 + (tap - (tap / 3) * 3 - 1));
        u32x4 a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          a[k] = u32x4{0u, 0u, 0u, 0u};
          if (k * 32 < nch && wlive) a[k] = lp_ld16(wrow + (size_t)tap * C2 + kc + k * 32);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k * 32 < nch) {
#pragma unroll
            for (int i = 0; i < LP_IMG; ++i)
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(img + (i * LP_RING + rpix[b] + toff) * LP_PITCH + k * 64 + kg * 16);
                acc[i][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[k]), __builtin_bit_cast(bf16x8, v), acc[i][b], 0, 0, 0);
              }
          }
        }
      }
    }
  }
  const int ci = ci0 + 4 * kg;
  if (!wave_live || ci >= H) return;
#pragma unroll
  for (int i = 0; i < LP_IMG; ++i) {
    const int n = n0 + i;
    if (n >= pk.N) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = b * 16 + l16;
      if (q >= LP_PIX) continue;
      u32x2 o;
      o.x = (uint32_t)f32_to_bf16(acc[i][b][0]) | ((uint32_t)f32_to_bf16(acc[i][b][1]) << 16);
      o.y = (uint32_t)f32_to_bf16(acc[i][b][2]) | ((uint32_t)f32_to_bf16(acc[i][b][3]) << 16);
      *reinterpret_cast<u32x2*>(dactv + ((size_t)n * LP_PIX + q) * pk.ctot + md.in_off + ci) = o;
    }
  }
}

// ---- weight + bias gradient: workgroup = (module, tap, 128 output channels); wave = 32 output channels x all hidden channels --------
// Per stage LP_G samples: the dgb tile [LP_G x 32 pixels][128 co] and the tap-shifted activation tile [LP_G x 32 pixels][hidden] in LDS
// (pixels 25..31 of a sample and out-of-image sources are zeros), then 2 LP_G reduction blocks of 16 pixels, fragments read transposed.
// The next stage's vectors are requested before the current stage is reduced (they wait in registers).
constexpr int LP_G = 3;          // (2 x 3 x 32 rows x 272 B = 51 KB of static LDS)
constexpr int LP_ROWB = 128 * 2 + 16;    // row pitch in bytes (+16: rows 4 apart land in different banks for the transposed reads)
constexpr int LP_VPT = LP_G * 32 * 32 / 256;             // staged vectors per thread at hidden = 128

__global__ __launch_bounds__(256) void label_gb_wgrad_kernel(const LabelPack pk, const bf16_t* __restrict__ actv) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * LP_G * 32 * LP_ROWB];
  unsigned char* const sa = smem;
  unsigned char* const sb = smem + LP_G * 32 * LP_ROWB;
  const int m = lp_module_of(pk, blockIdx.x);
  const LabelMod& md = pk.m[m];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = md.C, C2 = 2 * C, H = pk.hidden;
  const int local = blockIdx.x - md.blk0;
  const int tap = local % 9, cblk = local / 9;
  const int cob = cblk * 128;                           // first output channel of the workgroup
  const int ty = tap / 3 - 1, tx = tap - (tap / 3) * 3 - 1;
  const int NIB = H / 32;
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const int lr = lane & 31, lh = lane >> 5;
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_half = (lane >> 4) & 1;
  auto tr_read = [&](const unsigned char* base, int o0, int o1) {
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o1));
    u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    u32x4 r;
    r.x = l2.x; r.y = l2.y; r.z = h2.x; r.w = h2.y;
    return r;
  };
  const int a_colb = (wave * 32 + 16 * tr_half + 4 * tr_p) * 2;
  // the staging roles of this thread: vector v = tid + 256 u -> (row, column); fixed for the launch, only the sample changes
  const int va = 16, vb = H / 8, nvec = LP_G * 32 * (va + vb);
  int dst[LP_VPT], soff[LP_VPT], simg[LP_VPT];          // LDS byte offset (-1: none), source element offset within a sample (-1: zeros), sample
#pragma unroll
  for (int u = 0; u < LP_VPT; ++u) {
    const int v = tid + u * 256;
    dst[u] = soff[u] = -1;
    simg[u] = 0;
    if (v < nvec) {
      const int row = v / (va + vb), c = v - row * (va + vb);
      const int i = row / 32, p = row - i * 32;
      simg[u] = i;
      if (c < va) {
        dst[u] = row * LP_ROWB + c * 16;
        if (md.live && p < LP_PIX && cob + c * 8 < C2) soff[u] = p * C2 + cob + c * 8;
      } else {
        const int cc = c - va, py = p / 5, px = p - py * 5, sy = py + ty, sx = px + tx;
        dst[u] = LP_G * 32 * LP_ROWB + row * LP_ROWB + cc * 16;
        if (p < LP_PIX && sy >= 0 && sy < 5 && sx >= 0 && sx < 5) soff[u] = (sy * 5 + sx) * pk.ctot + md.in_off + cc * 8;
        simg[u] |= 0x100;                               // (flag: the activation tile)
      }
    }
  }
  u32x4 val[LP_VPT];
  auto request = [&](int n0) {
#pragma unroll
    for (int u = 0; u < LP_VPT; ++u) {
      val[u] = u32x4{0u, 0u, 0u, 0u};
      const int n = n0 + (simg[u] & 0xff);
      if (soff[u] >= 0 && n < pk.N)
        val[u] = (simg[u] & 0x100) ? lp_ld16(actv + (size_t)n * LP_PIX * pk.ctot + soff[u]) : lp_ld16(md.gb + (size_t)n * LP_PIX * C2 + soff[u]);
    }
  };
  float bs[4] = {0.f, 0.f, 0.f, 0.f};                   // tap 4 (no shift): column sums of dgb -- thread t sums channel cob + (t & 127), rows t >> 7 mod 2
  request(0);
#pragma unroll 1
  for (int n0 = 0; n0 < pk.N; n0 += LP_G) {
#pragma unroll
    for (int u = 0; u < LP_VPT; ++u)
      if (dst[u] >= 0) *reinterpret_cast<u32x4*>(smem + dst[u]) = val[u];
    __syncthreads();
    if (n0 + LP_G < pk.N) request(n0 + LP_G);            // in flight under this stage's reduction
    if (tap == 4) {
      const bf16_t* col = reinterpret_cast<const bf16_t*>(sa) + (tid & 127);
#pragma unroll 4
      for (int row = tid >> 7; row < LP_G * 32; row += 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[e] += bf16_to_f32(col[(row + 2 * e) * (LP_ROWB / 2)]);
      }
    }
    if (cob + wave * 32 < C2) {                         // wave-uniform
#pragma unroll 1
      for (int kb = 0; kb < LP_G * 2; ++kb) {
        const int ra = kb * 16 + 8 * lh + tr_q;         // rows ra, ra + 4 of the 16-pixel block's half lh
        const u32x4 af = tr_read(sa, ra * LP_ROWB + a_colb, (ra + 4) * LP_ROWB + a_colb);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j < NIB) {
            const int b_colb = (j * 32 + 16 * tr_half + 4 * tr_p) * 2;
            const u32x4 bf = tr_read(sb, ra * LP_ROWB + b_colb, (ra + 4) * LP_ROWB + b_colb);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bf), acc[j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
  // D[row = output channel][col = hidden channel]: lane (lr, lh) holds col lr, rows 8 q + 4 lh + r of acc element 4 q + r
  const int co_w = cob + wave * 32;
  if (co_w < C2) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j >= NIB) continue;
      const int ci = j * 32 + lr;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co_w + 8 * q + 4 * lh + r;
          if (co >= C2) continue;
          float* dstp = co < C ? md.dgw + ((size_t)co * H + ci) * 9 + tap : md.dbw + ((size_t)(co - C) * H + ci) * 9 + tap;
          *dstp = acc[j][4 * q + r];
        }
    }
  }
  if (tap == 4) {                                       // the two row halves of a channel meet through LDS (all reductions are done)
    float* red = reinterpret_cast<float*>(smem);
    red[tid] = (bs[0] + bs[1]) + (bs[2] + bs[3]);
    __syncthreads();
    if (tid < 128) {
      const float sum = red[tid] + red[tid + 128];
      const int co = cob + tid;
      if (co < C) md.dgbias[co] = sum;
      else if (co < C2) md.dbbias[co - C] = sum;
    }
  }
}

}  // namespace dei2i

using namespace dei2i;

namespace {

// the C-ABI table (include/dei2i_hip.h: dei2i_label_mod) -> the kernel argument
bool lp_build(const dei2i_label_mod* mods, int n, int hidden, int ctot, int N, int unit_of_module(const dei2i_label_mod&, int), LabelPack& pk,
              int& blocks) {
  if (!mods || n < 1 || n > LP_MAXMOD || hidden < 32 || hidden > 128 || hidden % 32 != 0 || ctot < hidden || ctot % 8 != 0 || N < 1) return false;
  pk.n = n; pk.hidden = hidden; pk.ctot = ctot; pk.N = N;
  blocks = 0;
  for (int i = 0; i < n; ++i) {
    const dei2i_label_mod& s = mods[i];
    if (s.C < 16 || s.C % 16 != 0 || s.in_off < 0 || s.in_off % 8 != 0 || s.in_off + hidden > ctot) return false;
    LabelMod& d = pk.m[i];
    d.gw = s.gamma_weight; d.bw = s.beta_weight; d.gbias = s.gamma_bias; d.bbias = s.beta_bias;
    d.wf = (bf16_t*)s.packed_fwd; d.wd = (bf16_t*)s.packed_dgrad; d.gb = (bf16_t*)s.gb;
    d.dgw = s.d_gamma_weight; d.dbw = s.d_beta_weight; d.dgbias = s.d_gamma_bias; d.dbbias = s.d_beta_bias;
    d.C = s.C; d.in_off = s.in_off; d.blk0 = blocks; d.live = s.live;
    blocks += unit_of_module(s, hidden);
  }
  return true;
}

}  // namespace

extern "C" {

size_t dei2i_label_gb_packed_elems(int C, int hidden) { return (size_t)2 * C * 9 * hidden; }

int dei2i_label_gb_pack(const dei2i_label_mod* mods, int n, int hidden, dei2i_stream s) {
  LabelPack pk;
  int blocks;
  if (!lp_build(mods, n, hidden, 1 << 20, 1, [](const dei2i_label_mod& md, int h) { return (2 * (2 * md.C * 9 * h / 8) + 255) / 256; }, pk, blocks))
    return DEI2I_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i)
    if (!mods[i].gamma_weight || !mods[i].beta_weight || !mods[i].packed_fwd || !mods[i].packed_dgrad) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(label_gb_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, pk);
  return (int)hipGetLastError();
}

int dei2i_label_gb_fwd(const dei2i_label_mod* mods, int n, int hidden, int ctot, int N, const void* actv, dei2i_stream s) {
  LabelPack pk;
  int blocks;
  if (!actv || !lp_build(mods, n, hidden, ctot, N, [](const dei2i_label_mod& md, int) { return (2 * md.C + 127) / 128; }, pk, blocks))
    return DEI2I_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i)
    if (!mods[i].packed_fwd || !mods[i].gamma_bias || !mods[i].beta_bias || !mods[i].gb) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(label_gb_fwd_kernel, dim3(blocks, (N + LP_IMG - 1) / LP_IMG), dim3(256), 0, (hipStream_t)s, pk, (const bf16_t*)actv);
  return (int)hipGetLastError();
}

int dei2i_label_gb_dgrad(const dei2i_label_mod* mods, int n, int hidden, int ctot, int N, void* dactv, dei2i_stream s) {
  LabelPack pk;
  int blocks;
  if (!dactv || !lp_build(mods, n, hidden, ctot, N, [](const dei2i_label_mod&, int h) { return (h + 63) / 64; }, pk, blocks)) return DEI2I_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i)
    if (mods[i].live && (!mods[i].packed_dgrad || !mods[i].gb)) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(label_gb_dgrad_kernel, dim3(blocks, (N + LP_IMG - 1) / LP_IMG), dim3(256), 0, (hipStream_t)s, pk, (bf16_t*)dactv);
  return (int)hipGetLastError();
}

int dei2i_label_gb_wgrad(const dei2i_label_mod* mods, int n, int hidden, int ctot, int N, const void* actv, dei2i_stream s) {
  LabelPack pk;
  int blocks;
  if (!actv || !lp_build(mods, n, hidden, ctot, N, [](const dei2i_label_mod& md, int) { return 9 * ((2 * md.C + 127) / 128); }, pk, blocks))
    return DEI2I_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i)
    if (!mods[i].d_gamma_weight || !mods[i].d_beta_weight || !mods[i].d_gamma_bias || !mods[i].d_beta_bias || (mods[i].live && !mods[i].gb))
      return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(label_gb_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, pk, (const bf16_t*)actv);
  return (int)hipGetLastError();
}

}  // extern "C"
