// C ABI for the convolution family: descriptor construction, weight (re)packing, fold of the dgrad halo.
#include <hip/hip_runtime.h>

#include "../../include/dei2i_hip.h"
#include "launch.h"

namespace dei2i {

int g_use_wgrad_v2 = 1;
int g_use_wgrad_halo = 1;
int g_use_wgrad_thin = 1;
int g_dgrad_s2_ring = 1;          // option "dgrad_s2_ring": stride-2 reflect dgrads decomposed (interior into dx + ring rectangles)
extern int g_halo_bn, g_halo_stages, g_halo16, g_halo16_stages, g_halo16_fold;

static ConvShape to_shape(const dei2i_conv* c) {
  ConvShape s;
  s.N = c->N; s.H = c->H; s.W = c->W; s.Cin = c->Cin; s.Cout = c->Cout;
  s.kh = c->kh; s.kw = c->kw; s.stride = c->stride; s.pad = c->pad; s.pad_mode = c->pad_mode; s.up = c->up;
  return s;
}

static bool valid_conv(const dei2i_conv* c) {
  if (!c) return false;
  const int vec = c->dtype == DT_BF16 ? 8 : 4;
  if (c->dtype != DT_BF16 && c->dtype != DT_F32) return false;
  if (c->N <= 0 || c->H <= 0 || c->W <= 0 || c->Cin <= 0 || c->Cout <= 0) return false;
  if (c->CinS < c->Cin || c->CoutS < c->Cout || c->CinS % vec || c->CoutS % vec) return false;
  if (c->kh <= 0 || c->kw <= 0 || c->stride <= 0 || c->pad < 0 || c->up < 0 || c->up > 1) return false;
  if (c->kh < c->stride || c->kw < c->stride) return false;          // every dgrad parity class needs a tap
  if (c->pad_mode == PAD_REFLECT && (c->pad >= (c->H << c->up) || c->pad >= (c->W << c->up))) return false;
  if (((c->H << c->up) + 2 * c->pad) < c->kh || ((c->W << c->up) + 2 * c->pad) < c->kw) return false;
  return true;
}

// ---- weight packing ------------------------------------------------------------------------------
template <typename T>
__global__ void pack_fwd_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int CinS, int kh, int kw) {
  const size_t total = (size_t)Cout * kh * kw * CinS;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const long long si = packed_fwd_src((long long)i, Cin, CinS, kh * kw);
    Elem<T>::store(dst + i, si >= 0 ? w[si] : 0.f);
  }
}

// class (ay,ax): rows = Cin, K = th*tw*CoutS, element [ci][(jy*tw+jx)][co] = w[co][ci][ay+s*jy][ax+s*jx]
template <typename T>
__global__ void pack_dgrad_kernel(const float* __restrict__ w, T* __restrict__ dst, int Cout, int Cin, int CoutS, int kh,
                                  int kw, int s, int ay, int ax, int th, int tw) {
  const size_t total = (size_t)Cin * th * tw * CoutS;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const long long si = packed_dgrad_src((long long)i, Cout, Cin, CoutS, kh, kw, s, ay, ax, th, tw);
    Elem<T>::store(dst + i, si >= 0 ? w[si] : 0.f);
  }
}

// forward layout and every dgrad parity class of one weight in ONE launch (a weight is re-packed once per optimizer step;
// with ~50 convolutions per net every saved launch is ~4.5 us of device time)
struct PackAll {
  long long fwd_total;
  long long cls_begin[17];       // element offsets of the dgrad classes in the packed dgrad buffer (+ total at [ncls])
  int ay[16], ax[16], th[16], tw[16];
  int ncls;
};
template <typename T>
__global__ void pack_all_kernel(const float* __restrict__ w, T* __restrict__ fwd, T* __restrict__ dgrad, const PackAll pa, int Cout,
                                int Cin, int CinS, int CoutS, int kh, int kw, int s) {
  const long long total = pa.fwd_total + pa.cls_begin[pa.ncls];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    if (i < pa.fwd_total) {
      const long long si = packed_fwd_src(i, Cin, CinS, kh * kw);
      Elem<T>::store(fwd + i, si >= 0 ? w[si] : 0.f);
    } else {
      const long long j = i - pa.fwd_total;
      int k = 0;
      while (k + 1 < pa.ncls && j >= pa.cls_begin[k + 1]) ++k;
      const long long si = packed_dgrad_src(j - pa.cls_begin[k], Cout, Cin, CoutS, kh, kw, s, pa.ay[k], pa.ax[k], pa.th[k], pa.tw[k]);
      Elem<T>::store(dgrad + j, si >= 0 ? w[si] : 0.f);
    }
  }
}

// The same for the weights that matter (the per-element kernel above reads w with a stride of `taps` floats per lane and
// stores 2 bytes per lane: 0.76 ms per step over ~43 weights, 6x its bytes at HBM speed).  A workgroup takes 32 output
// channels x 32 input channels x all taps: the OIHW rows of its tile are contiguous runs of 32 * taps floats (coalesced
// loads into LDS), both packed layouts go out as 16-byte stores -- forward rows [co][tap][ci0 .. ci0+31] (64-byte runs),
// dgrad rows [ci][class tap][co0 .. co0+31] (64-byte runs).  Padding channels are written as zeros by the tiles that
// reach past Cin / Cout.
constexpr int PT_CO = 16, PT_CI = 32, PT_MAXTAPS = 16, PT_ROW = PT_CI + 1;
template <typename T>
__global__ __launch_bounds__(256) void pack_tiled_kernel(const float* __restrict__ w, T* __restrict__ fwd, T* __restrict__ dgrad,
                                                         const PackAll pa, int Cout, int Cin, int CinS, int CoutS, int kh, int kw, int s) {
  constexpr int VEC = Elem<T>::VEC;
  // [co_l][tap][ci_l (+1 pad)]: the loads (tap fastest along a lane run) store at a stride of 33 floats, the forward rows read
  // 8 consecutive floats, the dgrad rows 8 floats at a stride of taps * 33
  extern __shared__ float tile[];
  const int taps = kh * kw;
  const int ci0 = blockIdx.x * PT_CI, co0 = blockIdx.y * PT_CO;
  const int run = PT_CI * taps;                                // floats of one output channel's piece of the tile
  // (run <= 512: a thread owns the positions r = tid and tid + 256 of every output channel's run -- no divisions in the
  //  loop, and the 32 loads of a thread are issued before the first LDS store)
  {
    const int r0 = threadIdx.x, r1 = threadIdx.x + 256;
    const int cil0 = r0 / taps, t0 = r0 - cil0 * taps, cil1 = r1 / taps, t1 = r1 - cil1 * taps;
    const bool in0 = r0 < run && ci0 + cil0 < Cin, in1 = r1 < run && ci0 + cil1 < Cin;
    float v0[PT_CO], v1[PT_CO];
#pragma unroll
    for (int col = 0; col < PT_CO; ++col) {
      const float* wr = w + ((size_t)(co0 + col) * Cin + ci0) * taps;
      const bool row = co0 + col < Cout;
      v0[col] = (row && in0) ? wr[r0] : 0.f;
      v1[col] = (row && in1) ? wr[r1] : 0.f;
    }
#pragma unroll
    for (int col = 0; col < PT_CO; ++col) {
      if (r0 < run) tile[(col * taps + t0) * PT_ROW + cil0] = v0[col];
      if (r1 < run) tile[(col * taps + t1) * PT_ROW + cil1] = v1[col];
    }
  }
  __syncthreads();
  // forward layout [co][tap][CinS]
  constexpr int FCH = PT_CI / VEC;                             // 16-byte chunks per (co, tap) run
  for (int i = threadIdx.x; i < PT_CO * taps * FCH; i += 256) {
    const int ch = i % FCH, r = i / FCH;                       // r = col * taps + t
    const int col = r / taps, t = r - col * taps;
    const int co = co0 + col, ci = ci0 + ch * VEC;
    if (co >= Cout || ci >= CinS) continue;                    // (the forward layout has Cout rows; CinS is a multiple of VEC)
    float v[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = tile[r * PT_ROW + ch * VEC + e];
    *reinterpret_cast<u32x4*>(fwd + ((size_t)co * taps + t) * CinS + ci) = Elem<T>::pack(v);
  }
  // dgrad layout, per parity class k: [Cin][th*tw][CoutS]
  constexpr int DCH = PT_CO / VEC;
  for (int k = 0; k < pa.ncls; ++k) {
    const int th = pa.th[k], tw = pa.tw[k], nt = th * tw;
    T* dst = dgrad + pa.cls_begin[k];
    for (int i = threadIdx.x; i < PT_CI * nt * DCH; i += 256) {
      const int ch = i % DCH, r = i / DCH;
      const int t = r % nt, cil = r / nt;
      const int ci = ci0 + cil, co = co0 + ch * VEC;
      if (ci >= Cin || co >= CoutS) continue;                  // (the dgrad layout has Cin rows)
      const int jy = t / tw, jx = t - jy * tw;
      const int tap = (pa.ay[k] + s * jy) * kw + (pa.ax[k] + s * jx);
      float v[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = tile[((ch * VEC + e) * taps + tap) * PT_ROW + cil];
      *reinterpret_cast<u32x4*>(dst + ((size_t)ci * nt + t) * CoutS + co) = Elem<T>::pack(v);
    }
  }
}

__global__ void unpack_wgrad_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int CinS,
                                    int taps, float beta) {
  const size_t total = (size_t)Cout * Cin * taps;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % taps);
    const size_t r = i / taps;
    const int ci = (int)(r % Cin);
    const int co = (int)(r / Cin);
    const float v = src[((size_t)co * taps + t) * CinS + ci];
    dst[i] = beta != 0.f ? beta * dst[i] + v : v;
  }
}

// ---- fold: gradient of reflect padding + nearest upsample ------------------------------------------
// dx[n,hs,ws,:] = sum over logical (hl,wl) in the 2^up x 2^up cell of sum over padded positions that map to it.
template <typename T>
__global__ void fold_pad_kernel(const T* __restrict__ ext, const T* __restrict__ addend, T* __restrict__ dx, int N, int H,
                                int W, int C, int pad, int reflect, int up) {
  constexpr int VEC = Elem<T>::VEC;
  const int Hl = H << up, Wl = W << up;
  const int off = reflect ? pad : 0;
  const int OH = Hl + 2 * off, OW = Wl + 2 * off;
  const int cv = C / VEC;
  const size_t total = (size_t)N * H * W * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    size_t r = i / cv;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if (addend != nullptr) {
      float t[VEC];
      Elem<T>::unpack(*reinterpret_cast<const u32x4*>(addend + i * VEC), t);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = t[e];
    }
    // candidate rows/cols in the extended frame
    int ys[6], xs[6];
    const int ny = fold_sources(h, up, Hl, pad, reflect, ys);
    const int nx = fold_sources(w, up, Wl, pad, reflect, xs);
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        float t[VEC];
        Elem<T>::unpack(*reinterpret_cast<const u32x4*>(ext + (((size_t)n * OH + ys[a]) * OW + xs[b]) * C + c), t);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += t[e];
      }
    *reinterpret_cast<u32x4*>(dx + i * VEC) = Elem<T>::pack(acc);
  }
}

// Border part of the reflect fold when the interior of the dgrad frame was written straight into dx: pixel (h, w) of
// the listed border rows / columns adds every folded ring source of ext except its own interior image (already in dx).
// Candidates per image: the 2*pad rows {1..pad, H-1-pad..H-2} at every column, then the 2*pad such columns at the
// remaining rows.
template <typename T>
__global__ void fold_border_kernel(const T* __restrict__ ext, T* __restrict__ dx, int N, int H, int W, int C, int pad) {
  constexpr int VEC = Elem<T>::VEC;
  const int cv = C / VEC;
  const int OW = W + 2 * pad, OH = H + 2 * pad;
  const int per_img = 2 * pad * W + 2 * pad * (H - 2 * pad);
  const size_t total = (size_t)N * per_img * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    size_t r = i / cv;
    const int q = (int)(r % per_img);
    const int n = (int)(r / per_img);
    int h, w;
    if (q < 2 * pad * W) {
      const int k = q / W;
      w = q - k * W;
      h = k < pad ? 1 + k : H - 1 - pad + (k - pad);
    } else {
      const int q2 = q - 2 * pad * W;
      const int k = q2 % (2 * pad);
      const int hr = q2 / (2 * pad);                       // index among the non-border rows
      w = k < pad ? 1 + k : W - 1 - pad + (k - pad);
      // non-border rows: 0, then pad+1 .. H-2-pad, then H-1   (rows 1..pad and H-1-pad..H-2 were handled above)
      h = hr == 0 ? 0 : (hr <= H - 2 - 2 * pad ? pad + hr : H - 1);
    }
    int ys[6], xs[6];
    const int ny = fold_sources(h, 0, H, pad, 1, ys);
    const int nx = fold_sources(w, 0, W, pad, 1, xs);
    if (ny == 1 && nx == 1) continue;
    T* dst = dx + (((size_t)n * H + h) * W + w) * C + c;
    float acc[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(dst), acc);
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        if (a == 0 && b == 0) continue;                    // the interior image, written by the interior GEMM
        float t[VEC];
        Elem<T>::unpack(*reinterpret_cast<const u32x4*>(ext + (((size_t)n * OH + ys[a]) * OW + xs[b]) * C + c), t);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += t[e];
      }
    *reinterpret_cast<u32x4*>(dst) = Elem<T>::pack(acc);
  }
}

}  // namespace dei2i

using namespace dei2i;

extern "C" {

size_t dei2i_packed_fwd_elems(const dei2i_conv* c) { return (size_t)c->Cout * c->kh * c->kw * c->CinS; }
size_t dei2i_wgrad_slab_elems(const dei2i_conv* c) { return (size_t)wgrad_slab_elems(c->Cout, c->kh * c->kw * c->CinS); }

size_t dei2i_packed_dgrad_elems(const dei2i_conv* c) {
  size_t n = 0;
  for (int ay = 0; ay < c->stride; ++ay)
    for (int ax = 0; ax < c->stride; ++ax)
      n += (size_t)c->Cin * dgrad_taps(c->kh, c->stride, ay) * dgrad_taps(c->kw, c->stride, ax) * c->CoutS;
  return n;
}

int dei2i_pack_weight_fwd(const dei2i_conv* c, const float* w, void* packed, dei2i_stream s) {
  if (!valid_conv(c) || !w || !packed) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const size_t total = dei2i_packed_fwd_elems(c);
  const unsigned grid = grid_for(total, 256);
  if (c->dtype == DT_BF16)
    hipLaunchKernelGGL(pack_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, w, (bf16_t*)packed, c->Cout, c->Cin, c->CinS, c->kh, c->kw);
  else
    hipLaunchKernelGGL(pack_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, w, (float*)packed, c->Cout, c->Cin, c->CinS, c->kh, c->kw);
  return (int)hipGetLastError();
}

int dei2i_pack_weight_dgrad(const dei2i_conv* c, const float* w, void* packed, dei2i_stream s) {
  if (!valid_conv(c) || !w || !packed) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const size_t esz = c->dtype == DT_BF16 ? 2 : 4;
  size_t off = 0;
  for (int ay = 0; ay < c->stride; ++ay)
    for (int ax = 0; ax < c->stride; ++ax) {
      const int th = dgrad_taps(c->kh, c->stride, ay), tw = dgrad_taps(c->kw, c->stride, ax);
      const size_t total = (size_t)c->Cin * th * tw * c->CoutS;
      if (total == 0) continue;
      const unsigned grid = grid_for(total, 256);
      void* dst = (char*)packed + off * esz;
      if (c->dtype == DT_BF16)
        hipLaunchKernelGGL(pack_dgrad_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, w, (bf16_t*)dst, c->Cout, c->Cin, c->CoutS,
                           c->kh, c->kw, c->stride, ay, ax, th, tw);
      else
        hipLaunchKernelGGL(pack_dgrad_kernel<float>, dim3(grid), dim3(256), 0, st, w, (float*)dst, c->Cout, c->Cin, c->CoutS,
                           c->kh, c->kw, c->stride, ay, ax, th, tw);
      off += total;
    }
  return (int)hipGetLastError();
}

int dei2i_pack_weight_both(const dei2i_conv* c, const float* w, void* packed_fwd, void* packed_dgrad, dei2i_stream s) {
  if (!valid_conv(c) || !w || !packed_fwd || !packed_dgrad || c->stride > 4) return DEI2I_ERR_BAD_ARG;
  PackAll pa;
  pa.fwd_total = (long long)dei2i_packed_fwd_elems(c);
  pa.ncls = 0;
  long long off = 0;
  for (int ay = 0; ay < c->stride; ++ay)
    for (int ax = 0; ax < c->stride; ++ax) {
      const int th = dgrad_taps(c->kh, c->stride, ay), tw = dgrad_taps(c->kw, c->stride, ax);
      const long long total = (long long)c->Cin * th * tw * c->CoutS;
      if (total == 0) continue;
      pa.ay[pa.ncls] = ay; pa.ax[pa.ncls] = ax; pa.th[pa.ncls] = th; pa.tw[pa.ncls] = tw;
      pa.cls_begin[pa.ncls++] = off;
      off += total;
    }
  pa.cls_begin[pa.ncls] = off;
  for (int k = pa.ncls + 1; k < 17; ++k) pa.cls_begin[k] = off;
  if (c->kh * c->kw <= PT_MAXTAPS && (long long)c->Cout * c->Cin * c->kh * c->kw >= (1 << 16) && c->CoutS % 8 == 0) {
    const dim3 tg((c->CinS + PT_CI - 1) / PT_CI, (c->CoutS + PT_CO - 1) / PT_CO);
    const size_t plds = (size_t)PT_CO * c->kh * c->kw * PT_ROW * sizeof(float);       // <= 67.6 KB
    static bool attr_done = false;
    if (!attr_done) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(pack_tiled_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          PT_CO * PT_MAXTAPS * PT_ROW * (int)sizeof(float));
      hipFuncSetAttribute(reinterpret_cast<const void*>(pack_tiled_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          PT_CO * PT_MAXTAPS * PT_ROW * (int)sizeof(float));
      attr_done = true;
    }
    if (c->dtype == DT_BF16)
      hipLaunchKernelGGL(pack_tiled_kernel<bf16_t>, tg, dim3(256), plds, (hipStream_t)s, w, (bf16_t*)packed_fwd, (bf16_t*)packed_dgrad, pa,
                         c->Cout, c->Cin, c->CinS, c->CoutS, c->kh, c->kw, c->stride);
    else
      hipLaunchKernelGGL(pack_tiled_kernel<float>, tg, dim3(256), plds, (hipStream_t)s, w, (float*)packed_fwd, (float*)packed_dgrad, pa,
                         c->Cout, c->Cin, c->CinS, c->CoutS, c->kh, c->kw, c->stride);
    return (int)hipGetLastError();
  }
  const unsigned grid = grid_for((size_t)(pa.fwd_total + off), 256);
  if (c->dtype == DT_BF16)
    hipLaunchKernelGGL(pack_all_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, w, (bf16_t*)packed_fwd, (bf16_t*)packed_dgrad,
                       pa, c->Cout, c->Cin, c->CinS, c->CoutS, c->kh, c->kw, c->stride);
  else
    hipLaunchKernelGGL(pack_all_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, w, (float*)packed_fwd, (float*)packed_dgrad,
                       pa, c->Cout, c->Cin, c->CinS, c->CoutS, c->kh, c->kw, c->stride);
  return (int)hipGetLastError();
}

int dei2i_unpack_wgrad(const dei2i_conv* c, const float* dw_packed, float* dw_oihw, float beta, dei2i_stream s) {
  if (!valid_conv(c) || !dw_packed || !dw_oihw) return DEI2I_ERR_BAD_ARG;
  const size_t total = (size_t)c->Cout * c->Cin * c->kh * c->kw;
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s, dw_packed, dw_oihw,
                     c->Cout, c->Cin, c->CinS, c->kh * c->kw, beta);
  return (int)hipGetLastError();
}

void dei2i_conv2d_out_shape(const dei2i_conv* c, int* Ho, int* Wo) {
  *Ho = conv_out_dim(c->H << c->up, c->kh, c->stride, c->pad);
  *Wo = conv_out_dim(c->W << c->up, c->kw, c->stride, c->pad);
}

void dei2i_conv2d_dgrad_shape(const dei2i_conv* c, int* OH, int* OW) {
  ConvShape s = to_shape(c);
  *OH = dgrad_out_h(s);
  *OW = dgrad_out_w(s);
}

size_t dei2i_conv2d_workspace_bytes(const dei2i_conv* c) {
  int Ho, Wo, OH, OW;
  dei2i_conv2d_out_shape(c, &Ho, &Wo);
  dei2i_conv2d_dgrad_shape(c, &OH, &OW);
  const size_t f = (size_t)c->N * Ho * Wo * c->CoutS, d = (size_t)c->N * OH * OW * c->CinS;
  // split-K slices write one fp32 slab each: room for 16 slabs, but no more than 64 MB beyond a single slab (layers
  // with large outputs have enough tiles to fill the chip without split-K)
  const size_t one = (f > d ? f : d) * sizeof(float);
  const size_t many = one * 16, cap = one > ((size_t)64 << 20) ? one : ((size_t)64 << 20);
  return many < cap ? many : cap;
}

int dei2i_conv2d_fwd(const dei2i_conv* c, const void* x, const void* w_packed, const float* bias, int act, void* y, float* ws,
                     size_t ws_bytes, dei2i_stream s) {
  if (!valid_conv(c) || !x || !w_packed || !y) return DEI2I_ERR_BAD_ARG;
  GatherDesc g = make_fwd_desc(to_shape(c), c->CinS);
  if (g.Ho <= 0 || g.Wo <= 0) return DEI2I_ERR_BAD_ARG;
  return (int)gather_gemm(c->dtype, g, x, w_packed, c->Cout, bias, y, ws, ws_bytes, c->CoutS, act, (hipStream_t)s);
}

int dei2i_conv2d_dgrad(const dei2i_conv* c, const void* dy, const void* wd_packed, void* dx_ext, float* ws, size_t ws_bytes,
                       dei2i_stream s) {
  if (!valid_conv(c) || !dy || !wd_packed || !dx_ext) return DEI2I_ERR_BAD_ARG;
  const ConvShape sh = to_shape(c);
  const int ncls = c->stride * c->stride;
  if (ncls > 4) return DEI2I_ERR_BAD_ARG;
  GatherDesc descs[4];
  long long woffs[4];
  long long off = 0;
  int n = 0;
  for (int ay = 0; ay < c->stride; ++ay)
    for (int ax = 0; ax < c->stride; ++ax) {
      descs[n] = make_dgrad_desc(sh, c->CoutS, ay, ax);
      woffs[n] = off;
      off += (long long)c->Cin * dgrad_taps(c->kh, c->stride, ay) * dgrad_taps(c->kw, c->stride, ax) * c->CoutS;
      ++n;
    }
  // all parity classes in ONE launch (blockIdx.y); split-K accumulates every class into the shared fp32 workspace
  return (int)gather_gemm_multi(c->dtype, descs, woffs, n, dy, wd_packed, c->Cin, nullptr, dx_ext, ws, ws_bytes, c->CinS,
                                ACT_NONE, (hipStream_t)s);
}

/* Gradient w.r.t. the conv's physical input, folded: dx (N,H,W,CinS).  ext_scratch: (N,OH,OW,CinS) elements
 * (dei2i_conv2d_dgrad_shape), used whenever the frame differs from the input (reflect padding and/or fused upsample).
 * Stride-1 reflect convs without upsample take the decomposed path: the interior of the padded frame IS the zero-
 * boundary dgrad on the input grid (one GEMM over N*H*W rows straight into dx -- tile-aligned, no full-size fold
 * pass), the reflect ring (4 thin rectangles of the frame) is a second small multi-descriptor GEMM into ext_scratch,
 * and fold_border_kernel adds the ring's images to the O(perimeter) border pixels of dx. */
int dei2i_conv2d_dgrad_input(const dei2i_conv* c, const void* dy, const void* wd_packed, void* ext_scratch, void* dx, float* ws,
                             size_t ws_bytes, dei2i_stream s) {
  if (!valid_conv(c) || !dy || !wd_packed || !dx) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const bool reflect = c->pad_mode == PAD_REFLECT && c->pad > 0;
  if (!reflect && !c->up) return dei2i_conv2d_dgrad(c, dy, wd_packed, dx, ws, ws_bytes, s);     // frame == input
  if (!ext_scratch) return DEI2I_ERR_BAD_ARG;
  const int p = c->pad;
  if (reflect && c->stride == 1 && !c->up && c->H >= 2 * p + 2 && c->W >= 2 * p + 2) {
    ConvShape sh = to_shape(c);
    const GatherDesc ring_frame = make_dgrad_desc(sh, c->CoutS, 0, 0);        // reflect: the padded frame
    sh.pad_mode = PAD_ZERO;
    const GatherDesc interior = make_dgrad_desc(sh, c->CoutS, 0, 0);          // zero: the input grid itself
    if (c->dtype == DT_BF16 && c->kh == 3 && c->kw == 3 && p == 1) {
      // large grids: the 16 x 32 tile kernel folds the ring AND the four corners itself (conv_halo16.hip FOLD): one launch
      hipError_t ef = halo16_conv(interior, dy, wd_packed, c->Cin, nullptr, dx, c->CinS, ACT_NONE, num_cu(), st, nullptr, nullptr, true);
      if (ef != hipErrorNotSupported) return (int)ef;
    }
    hipError_t e = gather_gemm(c->dtype, interior, dy, wd_packed, c->Cin, nullptr, dx, ws, ws_bytes, c->CinS, ACT_NONE, st);
    if (e != hipSuccess) return (int)e;
    const int OH = c->H + 2 * p, OW = c->W + 2 * p;
    GatherDesc ring[4] = {sub_rect_desc(ring_frame, 0, p, 0, OW), sub_rect_desc(ring_frame, c->H + p, p, 0, OW),
                          sub_rect_desc(ring_frame, p, c->H, 0, p), sub_rect_desc(ring_frame, p, c->H, c->W + p, p)};
    if (c->kh == 2 * p + 1 && c->kw == 2 * p + 1) {
      // "same" convs: the top p frame rows only see tap rows 0..p-1 (the others read above dY), the bottom rows only
      // tap rows kh-p..kh-1, likewise the side columns -- run each rectangle on its live taps (a third of K for 3x3)
      ring[0] = sub_taps_desc(ring[0], 0, p, 0, c->kw);
      ring[1] = sub_taps_desc(ring[1], c->kh - p, p, 0, c->kw);
      ring[2] = sub_taps_desc(ring[2], 0, c->kh, 0, p);
      ring[3] = sub_taps_desc(ring[3], 0, c->kh, c->kw - p, p);
    }
    const long long woffs[4] = {0, 0, 0, 0};
    // ring launch: dead taps are skipped (2 of 3 tap rows / columns fall outside dY for a whole tile); split-K with a
    // compact workspace -- the partials are indexed by ring row, so memset / finalize touch ring rows only
    // (measured at 256 channels, 64x64: split 24 + 6 + 5 us vs 38 us for ~70 unsplit workgroups)
    e = gather_gemm_multi(c->dtype, ring, woffs, 4, dy, wd_packed, c->Cin, nullptr, ext_scratch, ws, ws_bytes, c->CinS, ACT_NONE, st,
                          true);
    if (e != hipSuccess) return (int)e;
    const int vec = c->dtype == DT_BF16 ? 8 : 4;
    const size_t total = (size_t)c->N * (2 * p * c->W + 2 * p * (c->H - 2 * p)) * (c->CinS / vec);
    const unsigned grid = grid_for(total, 256);
    if (c->dtype == DT_BF16)
      hipLaunchKernelGGL(fold_border_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)ext_scratch, (bf16_t*)dx, c->N,
                         c->H, c->W, c->CinS, p);
    else
      hipLaunchKernelGGL(fold_border_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)ext_scratch, (float*)dx, c->N,
                         c->H, c->W, c->CinS, p);
    return (int)hipGetLastError();
  }
  if (reflect && g_dgrad_s2_ring && c->stride == 2 && c->kh == 4 && c->kw == 4 && p == 1 && !c->up && c->H % 2 == 0 && c->W % 2 == 0 &&
      c->H >= 8 && c->W >= 8 && (size_t)c->N * c->H * c->W * c->CinS >= ((size_t)48 << 20)) {     // (the fold pass costs ~0.5 us per M elements, the ring ~30 us)
    // The stride-2 4x4 convs of the encoder / discriminator (generator.py:107-116, discriminator.py:60-77), large frames: the same
    // decomposition as above, per parity class.  The frame's interior is the zero-boundary dgrad on the input grid (four classes,
    // one launch, straight into dx: no (H+2) x (W+2) frame tensor and no full-size fold pass over it -- up to 135 us per conv at
    // 256 x 256 x 64 x 32 images); frame row 0 / H+1 and column 0 / W+1 are one-pixel rectangles of the two classes of their
    // parity, two small launches into ext_scratch (dead taps skipped), folded onto rows 1 / H-2 and columns 1 / W-2 of dx.
    ConvShape sh = to_shape(c);
    GatherDesc fr[2][2], in4[4];
    long long woff[2][2], woffs[4], off = 0;
    for (int ay = 0; ay < 2; ++ay)
      for (int ax = 0; ax < 2; ++ax) {
        fr[ay][ax] = make_dgrad_desc(sh, c->CoutS, ay, ax);
        woff[ay][ax] = off;
        off += (long long)c->Cin * dgrad_taps(c->kh, 2, ay) * dgrad_taps(c->kw, 2, ax) * c->CoutS;
      }
    sh.pad_mode = PAD_ZERO;
    for (int ay = 0; ay < 2; ++ay)
      for (int ax = 0; ax < 2; ++ax) { in4[2 * ay + ax] = make_dgrad_desc(sh, c->CoutS, ay, ax); woffs[2 * ay + ax] = woff[ay][ax]; }
    hipError_t e = gather_gemm_multi(c->dtype, in4, woffs, 4, dy, wd_packed, c->Cin, nullptr, dx, ws, ws_bytes, c->CinS, ACT_NONE, st);
    if (e != hipSuccess) return (int)e;
    // frame rows: hp = 0 is row 0 of the classes ay = 0, hp = H + 1 (odd) the last row of the classes ay = 1; all their columns
    const GatherDesc rows[4] = {sub_rect_desc(fr[0][0], 0, 1, 0, fr[0][0].Wo), sub_rect_desc(fr[0][1], 0, 1, 0, fr[0][1].Wo),
                                sub_rect_desc(fr[1][0], fr[1][0].Ho - 1, 1, 0, fr[1][0].Wo),
                                sub_rect_desc(fr[1][1], fr[1][1].Ho - 1, 1, 0, fr[1][1].Wo)};
    const long long wrows4[4] = {woff[0][0], woff[0][1], woff[1][0], woff[1][1]};
    e = gather_gemm_multi(c->dtype, rows, wrows4, 4, dy, wd_packed, c->Cin, nullptr, ext_scratch, ws, ws_bytes, c->CinS, ACT_NONE, st, true);
    if (e != hipSuccess) return (int)e;
    // frame columns without the corners (they are in the rows): wp = 0 is column 0 of the classes ax = 0 (class ay = 0 without its
    // row 0, class ay = 1 without its last row), wp = W + 1 the last column of the classes ax = 1
    const GatherDesc cols[4] = {sub_rect_desc(fr[0][0], 1, fr[0][0].Ho - 1, 0, 1), sub_rect_desc(fr[1][0], 0, fr[1][0].Ho - 1, 0, 1),
                                sub_rect_desc(fr[0][1], 1, fr[0][1].Ho - 1, fr[0][1].Wo - 1, 1),
                                sub_rect_desc(fr[1][1], 0, fr[1][1].Ho - 1, fr[1][1].Wo - 1, 1)};
    const long long wcols4[4] = {woff[0][0], woff[1][0], woff[0][1], woff[1][1]};
    e = gather_gemm_multi(c->dtype, cols, wcols4, 4, dy, wd_packed, c->Cin, nullptr, ext_scratch, ws, ws_bytes, c->CinS, ACT_NONE, st, true);
    if (e != hipSuccess) return (int)e;
    const int vec = c->dtype == DT_BF16 ? 8 : 4;
    const size_t total = (size_t)c->N * (2 * c->W + 2 * (c->H - 2)) * (c->CinS / vec);
    if (c->dtype == DT_BF16)
      hipLaunchKernelGGL(fold_border_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const bf16_t*)ext_scratch, (bf16_t*)dx,
                         c->N, c->H, c->W, c->CinS, 1);
    else
      hipLaunchKernelGGL(fold_border_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const float*)ext_scratch, (float*)dx, c->N,
                         c->H, c->W, c->CinS, 1);
    return (int)hipGetLastError();
  }
  const int rc = dei2i_conv2d_dgrad(c, dy, wd_packed, ext_scratch, ws, ws_bytes, s);
  if (rc != 0) return rc;
  return dei2i_fold_pad(c->dtype, c->N, c->H, c->W, c->CinS, c->pad, c->pad_mode, c->up, ext_scratch, nullptr, dx, s);
}

/* ---- input gradient + the backward reductions of the norm layer in front of the conv, one launch (conv_halo16.hip EPIN) ---- */
static bool dgrad_norm_shape_ok(const dei2i_conv* c) {
  if (!g_halo16 || !g_halo16_fold || g_halo16_stages != 8 || g_halo_bn != 0 || g_halo_stages != 0) return false;   // A/B options
  if (!valid_conv(c) || c->dtype != DT_BF16 || c->up) return false;
  if (c->kh != 3 || c->kw != 3 || c->stride != 1 || c->pad != 1 || c->pad_mode != PAD_REFLECT) return false;
  if (c->H % 16 != 0 || c->W % 32 != 0 || c->H < 32 || c->W < 64 || c->CinS < 64 || c->CinS % 8 != 0 || c->CoutS % 32 != 0) return false;
  if ((long long)c->N * c->H * c->W * c->CoutS >= (1ll << 31)) return false;
  const int tiles_m = c->N * (c->H / 16) * (c->W / 32);
  const int tn = c->CinS >= 128 ? (c->CinS + 127) / 128 : 1;
  return tiles_m * tn >= (num_cu() * 7) / 8;
}

int dei2i_conv2d_dgrad_norm_supported(const dei2i_conv* c) { return dgrad_norm_shape_ok(c) ? 1 : 0; }

int dei2i_conv2d_dgrad_norm_chunks(const dei2i_conv* c) { return (c->H / 16) * (c->W / 32); }    // one record per 16 x 32 tile

int dei2i_conv2d_dgrad_input_norm(const dei2i_conv* c, const void* dy, const void* wd_packed, void* dx, const dei2i_epi_norm* en,
                                  dei2i_stream s) {
  if (!dy || !wd_packed || !dx || !en || !dgrad_norm_shape_ok(c)) return DEI2I_ERR_BAD_ARG;
  if (!en->x || !en->mean || !en->rstd || !en->partial || en->up < 0 || en->up > 1) return DEI2I_ERR_BAD_ARG;
  const int kind = en->kind & 0xff;            // (bits 8+: timing-only switches of tools/diag_epin.py)
  if (kind == 1 ? !en->gb : (kind != 2 || !en->a || !en->b || en->up)) return DEI2I_ERR_BAD_ARG;
  if (en->group_images < 0 || (en->group_images > 0 && c->N % en->group_images != 0)) return DEI2I_ERR_BAD_ARG;
  ConvShape sh = to_shape(c);
  sh.pad_mode = PAD_ZERO;
  const GatherDesc interior = make_dgrad_desc(sh, c->CoutS, 0, 0);
  EpiNorm e;
  e.x = (const uint16_t*)en->x; e.mean = en->mean; e.rstd = en->rstd; e.gb = (const uint16_t*)en->gb; e.a = en->a; e.b = en->b;
  e.partial = en->partial; e.kind = en->kind; e.up = en->up; e.act = en->act;
  e.group_images = kind == 2 ? en->group_images : 0;
  hipError_t ef = halo16_conv(interior, dy, wd_packed, c->Cin, nullptr, dx, c->CinS, ACT_NONE, num_cu(), (hipStream_t)s, nullptr, nullptr,
                              true, &e);
  return ef == hipErrorNotSupported ? DEI2I_ERR_BAD_ARG : (int)ef;
}

int dei2i_conv2d_wgrad(const dei2i_conv* c, const void* x, const void* dy, float* dw_packed, dei2i_stream s) {
  if (!valid_conv(c) || !x || !dy || !dw_packed) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  GatherDesc g = make_fwd_desc(to_shape(c), c->CinS);
  return (int)wgrad_gemm(c->dtype, g, x, dy, c->Cout, c->CoutS, dw_packed, 0, nullptr, st);
}

/* wgrad straight to the OIHW fp32 gradient: the bf16 hot shapes take the LDS-DMA slab kernel + fused reduce/un-pack
 * (scratch >= dei2i_wgrad_slab_elems floats; more lets it split the pixel range further); other shapes take the v1 kernel
 * into scratch[0 : packed elems] and the un-pack kernel. */
int dei2i_conv2d_wgrad_oihw(const dei2i_conv* c, const void* x, const void* dy, float* scratch, size_t scratch_elems,
                            float* dw_oihw, int accumulate, dei2i_stream s) {
  if (!valid_conv(c) || !x || !dy || !scratch || !dw_oihw) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  GatherDesc g = make_fwd_desc(to_shape(c), c->CinS);
  const size_t packed = (size_t)wgrad_slab_elems(c->Cout, g.K);      // slab stride: Cout rounded up to 8 rows
  if (scratch_elems < packed) return DEI2I_ERR_WORKSPACE;
  if (c->dtype == DT_BF16 && g_use_wgrad_thin) {       // 8-channel input, 7x7 (the stem): halo-resident, windowed B operand
    int nsplit = 0;
    hipError_t e = wgrad_thin(g, x, dy, c->Cout, c->CoutS, scratch, scratch_elems, num_cu(), &nsplit, st);
    if (e == hipSuccess)
      return (int)wgrad_reduce_unpack(scratch, nsplit, (long long)packed, dw_oihw, c->Cout, c->Cin, c->CinS, c->kh * c->kw, accumulate, st);
    if (e != hipErrorNotSupported) return (int)e;
  }
  if (c->dtype == DT_BF16 && g_use_wgrad_halo) {       // stride-1 3x3: (co, ci, 9 taps) block resident in registers
    int nsplit = 0;
    hipError_t e = wgrad_halo(g, x, dy, c->Cout, c->CoutS, scratch, scratch_elems, num_cu(), &nsplit, g_use_wgrad_halo == 2, st);
    if (e == hipSuccess)
      return (int)wgrad_reduce_unpack(scratch, nsplit, (long long)packed, dw_oihw, c->Cout, c->Cin, c->CinS, c->kh * c->kw, accumulate, st);
    if (e != hipErrorNotSupported) return (int)e;
  }
  if (c->dtype == DT_BF16 && g_use_wgrad_v2) {
    int nsplit = 0;
    hipError_t e = wgrad_v2(g, x, dy, c->Cout, c->CoutS, scratch, scratch_elems, num_cu(), &nsplit, st);
    if (e == hipSuccess)
      return (int)wgrad_reduce_unpack(scratch, nsplit, (long long)packed, dw_oihw, c->Cout, c->Cin, c->CinS, c->kh * c->kw, accumulate, st);
    if (e != hipErrorNotSupported) return (int)e;
  }
  int nsplit = 0;
  hipError_t e = wgrad_gemm(c->dtype, g, x, dy, c->Cout, c->CoutS, scratch, scratch_elems, &nsplit, st);
  if (e != hipSuccess) return (int)e;
  return (int)wgrad_reduce_unpack(scratch, nsplit, (long long)packed, dw_oihw, c->Cout, c->Cin, c->CinS, c->kh * c->kw, accumulate, st);
}

/* ---- fused conv + norm + act (halo-resident kernels only; ask *_supported first -- there is no fallback inside) ---- */
static bool to_pro(const dei2i_conv* c, const dei2i_pro* p, ConvPro& out) {
  if (p == nullptr) return false;
  out.A = p->A; out.B = p->B; out.n_stride = p->n_stride; out.slope = p->slope;
  out.ring = (const uint16_t*)p->ring;
  out.ring_pix = p->ring != nullptr ? ring_pixels(c->H << c->up, c->W << c->up) : 0;
  return true;
}

static bool halo_fwd_shape_ok(const dei2i_conv* c, int want_pro) {
  if (!valid_conv(c) || c->dtype != DT_BF16) return false;
  if (c->kh == 4 && c->kw == 4 && c->stride == 2 && c->pad == 1 && !want_pro)          // the stride-2 form (conv_halo16.hip S2)
    return halo16_s2_shape_ok(make_fwd_desc(to_shape(c), c->CinS), c->CoutS, num_cu());
  if (c->kh != 3 || c->kw != 3 || c->stride != 1 || c->pad != 1) return false;
  const int Ho = c->H << c->up, Wo = c->W << c->up;
  if (c->CinS % 64 != 0 || Ho % 8 != 0 || Wo % 32 != 0 || c->CoutS < 64) return false;
  if ((long long)c->N * c->H * c->W * c->CinS >= (1ll << 31)) return false;
  if (want_pro && (c->CinS > 512 || Ho < 4 || Wo < 4)) return false;
  const int tiles_m = c->N * (Ho / 8) * (Wo / 32);
  const int tn = c->CoutS >= 128 ? (c->CoutS + 127) / 128 : 1;
  return tiles_m * tn >= num_cu() / 2;
}

int dei2i_conv2d_fused_supported(const dei2i_conv* c, int want_pro) { return halo_fwd_shape_ok(c, want_pro) ? 1 : 0; }

int dei2i_conv2d_stats_chunks(const dei2i_conv* c) {
  int Ho, Wo;
  dei2i_conv2d_out_shape(c, &Ho, &Wo);
  return (Ho / 8) * (Wo / 32);
}

int dei2i_conv2d_fwd_fused(const dei2i_conv* c, const void* x, const void* w_packed, const float* bias, int act, void* y,
                           const dei2i_pro* pro, float* stats, dei2i_stream s) {
  if (!x || !w_packed || !y || !halo_fwd_shape_ok(c, pro != nullptr)) return DEI2I_ERR_BAD_ARG;
  if (pro != nullptr && (!pro->A || !pro->B)) return DEI2I_ERR_BAD_ARG;
  GatherDesc g = make_fwd_desc(to_shape(c), c->CinS);
  ConvPro cp;
  const bool has = to_pro(c, pro, cp);
  if (!has) {
    hipError_t e = halo16_conv(g, x, w_packed, c->Cout, bias, y, c->CoutS, act, num_cu(), (hipStream_t)s, stats);
    if (e != hipErrorNotSupported) return (int)e;
  }
  return (int)halo_conv(g, x, w_packed, c->Cout, bias, y, c->CoutS, act, num_cu(), (hipStream_t)s, nullptr, has ? &cp : nullptr, stats);
}

/* conv of the upsampled z of a SPADE block, z kept at the SOURCE resolution + the logical frame's ring tensor: the 16 x 32
 * tile kernel only (no fallback: ask first) */
int dei2i_conv2d_ring_supported(const dei2i_conv* c) {
  if (!halo_fwd_shape_ok(c, 0) || !c->up) return 0;
  const int Ho = c->H << c->up, Wo = c->W << c->up;
  if (Ho % 16 != 0 || Wo % 32 != 0 || c->CinS % 32 != 0) return 0;
  const int tiles_m = c->N * (Ho / 16) * (Wo / 32);
  const int tn = c->CoutS >= 128 ? (c->CoutS + 127) / 128 : 1;
  return tiles_m * tn >= (num_cu() * 7) / 8 ? 1 : 0;
}

int dei2i_conv2d_fwd_ring(const dei2i_conv* c, const void* z_src, const void* ring, const void* w_packed, const float* bias, int act,
                          void* y, float* stats, dei2i_stream s) {
  if (!z_src || !ring || !w_packed || !y || !dei2i_conv2d_ring_supported(c)) return DEI2I_ERR_BAD_ARG;
  GatherDesc g = make_fwd_desc(to_shape(c), c->CinS);
  return (int)halo16_conv(g, z_src, w_packed, c->Cout, bias, y, c->CoutS, act, num_cu(), (hipStream_t)s, stats, ring);
}

int dei2i_conv2d_wgrad_pro_supported(const dei2i_conv* c) {
  if (!valid_conv(c) || c->dtype != DT_BF16) return 0;
  if (c->kh != 3 || c->kw != 3 || c->stride != 1 || c->pad != 1) return 0;
  const int Ho = c->H << c->up, Wo = c->W << c->up;
  if (Ho % 4 != 0 || Wo % 32 != 0 || Ho < 4 || Wo < 4) return 0;
  if ((long long)c->N * c->H * c->W * c->CinS >= (1ll << 31)) return 0;
  const int co = c->Cout;
  if (co <= 64) return ((c->CinS % 128 == 0 && co >= 48) || (c->CinS % 64 == 0 && co <= 32)) ? 1 : 0;
  return (c->CinS % 64 == 0 && co >= 96) ? 1 : 0;
}

int dei2i_conv2d_wgrad_oihw_pro(const dei2i_conv* c, const void* x, const void* dy, float* scratch, size_t scratch_elems,
                                float* dw_oihw, int accumulate, const dei2i_pro* pro, dei2i_stream s) {
  if (!x || !dy || !scratch || !dw_oihw || !pro || !dei2i_conv2d_wgrad_pro_supported(c)) return DEI2I_ERR_BAD_ARG;
  if ((pro->A == nullptr) != (pro->B == nullptr) || (pro->A == nullptr && pro->ring == nullptr)) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  GatherDesc g = make_fwd_desc(to_shape(c), c->CinS);
  const size_t packed = (size_t)wgrad_slab_elems(c->Cout, g.K);
  if (scratch_elems < packed) return DEI2I_ERR_WORKSPACE;
  ConvPro cp;
  to_pro(c, pro, cp);
  int nsplit = 0;
  hipError_t e = wgrad_halo(g, x, dy, c->Cout, c->CoutS, scratch, scratch_elems, num_cu(), &nsplit, true, st, &cp);
  if (e != hipSuccess) return (int)e;
  return (int)wgrad_reduce_unpack(scratch, nsplit, (long long)packed, dw_oihw, c->Cout, c->Cin, c->CinS, c->kh * c->kw, accumulate, st);
}

int dei2i_fold_pad(int dtype, int N, int H, int W, int C, int pad, int pad_mode, int up, const void* dx_ext,
                   const void* addend, void* dx, dei2i_stream s) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % vec || pad < 0 || pad > 3 || up < 0 || up > 1 || !dx_ext || !dx)
    return DEI2I_ERR_BAD_ARG;
  const size_t total = (size_t)N * H * W * (C / vec);
  const unsigned grid = grid_for(total, 256, 256u * 16u);
  const int reflect = (pad_mode == PAD_REFLECT && pad > 0) ? 1 : 0;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(fold_pad_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const bf16_t*)dx_ext,
                       (const bf16_t*)addend, (bf16_t*)dx, N, H, W, C, pad, reflect, up);
  else
    hipLaunchKernelGGL(fold_pad_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const float*)dx_ext,
                       (const float*)addend, (float*)dx, N, H, W, C, pad, reflect, up);
  return (int)hipGetLastError();
}

}  // extern "C"
