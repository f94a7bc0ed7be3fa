// HBM-bound elementwise kernels (NHWC, 16-byte vectors): layout conversion at the module boundary, BatchNorm-apply
// + activation, SPADE modulate + ReLU, activation backward, head compose, NaN guard, casts.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <math.h>

#include "../../include/dei2i_hip.h"
#include "launch.h"

namespace dei2i {

// ---- NCHW fp32 (N,C,hs,ws) -> NHWC T (N,H,W,Cs), nearest resize when (hs,ws) != (H,W), zero channel padding ----
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int hs, int ws, int H,
                                    int W, int Cs) {
  constexpr int VEC = Elem<T>::VEC;
  const int cv = Cs / VEC;
  const size_t total = (size_t)N * H * W * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // pixel-fastest mapping so the NCHW plane reads stay coalesced
    const size_t hw = (size_t)H * W;
    const size_t p = i % hw;
    size_t r = i / hw;
    const int v = (int)(r % cv);
    const int n = (int)(r / cv);
    const int h = (int)(p / W), w = (int)(p % W);
    const int sy = (int)(((long long)h * hs) / H), sx = (int)(((long long)w * ws) / W);   // F.interpolate 'nearest'
    float f[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const int c = v * VEC + e;
      f[e] = c < C ? src[(((size_t)n * C + c) * hs + sy) * ws + sx] : 0.f;
    }
    *reinterpret_cast<u32x4*>(dst + (((size_t)n * hw + p) * Cs + (size_t)v * VEC)) = Elem<T>::pack(f);
  }
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W, int Cs) {
  const size_t hw = (size_t)H * W;
  const size_t total = (size_t)N * C * hw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % hw;
    size_t r = i / hw;
    const int c = (int)(r % C);
    const int n = (int)(r / C);
    dst[i] = Elem<T>::load(src + ((size_t)n * hw + p) * Cs + c);
  }
}

// ---- out = act(a[c]*x + b[c]) (+ res) ----
// When the launch's thread count is a multiple of cv, a thread's channel vector is the same in every grid-stride
// iteration and its coefficients live in registers (INVARIANT); otherwise they are re-read per iteration.
template <typename T, bool INVARIANT>
__global__ void affine_act_kernel(const T* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b,
                                  const T* __restrict__ res, T* __restrict__ out, size_t nvec, int cv, int act,
                                  unsigned char* __restrict__ out8, float scale8) {
  constexpr int VEC = Elem<T>::VEC;
  {                                                    // blockIdx.y: a group of the batch (nvec vectors, its own row of coefficients)
    const size_t grp = blockIdx.y;
    x += grp * nvec * VEC; out += grp * nvec * VEC;
    if (res != nullptr) res += grp * nvec * VEC;
    a += grp * (size_t)cv * VEC; b += grp * (size_t)cv * VEC;
    if (out8 != nullptr) out8 += grp * nvec * 8;
  }
  float av[VEC], bv[VEC];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (INVARIANT) {
    const int c = (int)(i % cv) * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { av[e] = a[c + e]; bv[e] = b[c + e]; }
  }
  for (; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    if (!INVARIANT) {
      const int c = (int)(i % cv) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) { av[e] = a[c + e]; bv[e] = b[c + e]; }
    }
    float f[VEC], r[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(x + i * VEC), f);
    if (res != nullptr) Elem<T>::unpack(*reinterpret_cast<const u32x4*>(res + i * VEC), r);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float v = apply_act(fmaf(av[e], f[e], bv[e]), act);
      if (res != nullptr) v += r[e];
      f[e] = v;
    }
    const u32x4 pk = Elem<T>::pack(f);
    *reinterpret_cast<u32x4*>(out + i * VEC) = pk;
    if constexpr (sizeof(T) == 2) {              // fp8 forward mode: the e4m3 copy the next convolution reads
      if (out8 != nullptr) store_e4m3_of_bf16x8(out8 + i * 8, pk, scale8);
    }
  }
}

// ---- out = act(A[n,c]*x + B[n,c]) with per-image coefficients (grid.y = image) ----
template <typename T>
__global__ void affine_act_img_kernel(const T* __restrict__ x, const float* __restrict__ A, const float* __restrict__ B,
                                      T* __restrict__ out, size_t nvec_img, int cv, float slope) {
  constexpr int VEC = Elem<T>::VEC;
  const int n = blockIdx.y;
  const size_t base = (size_t)n * nvec_img;
  float av[VEC], bv[VEC];
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // launch: gridDim.x * blockDim.x is a multiple of cv
  const int c = (int)(i % cv) * VEC;
  ldcoef<VEC>(A + (size_t)n * cv * VEC + c, av);
  ldcoef<VEC>(B + (size_t)n * cv * VEC + c, bv);
  for (; i < nvec_img; i += (size_t)gridDim.x * blockDim.x) {
    float f[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(x + (base + i) * VEC), f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float v = fmaf(av[e], f[e], bv[e]);
      f[e] = fmaf(slope, fminf(v, 0.f), fmaxf(v, 0.f));
    }
    *reinterpret_cast<u32x4*>(out + (base + i) * VEC) = Elem<T>::pack(f);
  }
}


// ---- SPADE modulate + ReLU ----
template <typename T>
__global__ void spade_act_kernel(const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                 const T* __restrict__ gb, T* __restrict__ out, int N, int H, int W, int C, int up,
                                 int gb_mode, unsigned char* __restrict__ out8, float scale8) {
  constexpr int VEC = Elem<T>::VEC;
  const int cv = C / VEC;
  const int Hs = H >> up, Ws = W >> up;
  const size_t total = (size_t)N * H * W * cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    size_t r = i / cv;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float xv[VEC], gm[VEC], bt[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(x + (((size_t)n * Hs + (h >> up)) * Ws + (w >> up)) * C + c), xv);
    size_t gpix;
    if (gb_mode == 0) gpix = ((size_t)n * H + h) * W + w;
    else gpix = ((size_t)n * 5 + border_class(h, H)) * 5 + border_class(w, W);
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(gb + gpix * 2 * C + c), gm);
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(gb + gpix * 2 * C + C + c), bt);
    float o[VEC], mv[VEC], rv[VEC];
    ldcoef<VEC>(mean + (size_t)n * C + c, mv);
    ldcoef<VEC>(rstd + (size_t)n * C + c, rv);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float xh = (xv[e] - mv[e]) * rv[e];
      const float v = fmaf(xh, 1.f + gm[e], bt[e]);
      o[e] = v > 0.f ? v : 0.f;
    }
    const u32x4 pk = Elem<T>::pack(o);
    *reinterpret_cast<u32x4*>(out + i * VEC) = pk;
    if constexpr (sizeof(T) == 2) {
      if (out8 != nullptr) store_e4m3_of_bf16x8(out8 + i * 8, pk, scale8);
    }
  }
}

// ---- g = dz * act'(z) ----
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dz, const T* __restrict__ z, T* __restrict__ g, size_t nvec, int act) {
  constexpr int VEC = Elem<T>::VEC;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    float d[VEC], zz[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(dz + i * VEC), d);
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(z + i * VEC), zz);
#pragma unroll
    for (int e = 0; e < VEC; ++e) d[e] *= act_grad_from_out(zz[e], act);
    *reinterpret_cast<u32x4*>(g + i * VEC) = Elem<T>::pack(d);
  }
}

// ---- heads: tanh / sigmoid / compose, NHWC raw -> NCHW fp32 outputs ----
template <typename T>
__global__ void compose_fwd_kernel(const T* __restrict__ raw, const float* __restrict__ x_in, float* __restrict__ out,
                                   float* __restrict__ prob, int N, int HW, int Cs) {
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, p = i % HW;
    const T* rp = raw + i * Cs;
    const float pr = 1.f / (1.f + expf(-Elem<T>::load(rp + 3)));
    prob[i] = pr;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float fg = tanhf(Elem<T>::load(rp + c));
      const size_t o = (n * 3 + c) * HW + p;
      out[o] = x_in[o] * (1.f - pr) + fg * pr;
    }
  }
}

template <typename T>
__global__ void compose_bwd_kernel(const T* __restrict__ raw, const float* __restrict__ x_in, const float* __restrict__ d_out,
                                   const float* __restrict__ d_prob, T* __restrict__ d_raw, float* __restrict__ d_x, int N,
                                   int HW, int Cs) {
  constexpr int VEC = Elem<T>::VEC;
  const size_t total = (size_t)N * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / HW, p = i % HW;
    const T* rp = raw + i * Cs;
    const float pr = 1.f / (1.f + expf(-Elem<T>::load(rp + 3)));
    float dp = d_prob != nullptr ? d_prob[i] : 0.f;
    float dr[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float fg = tanhf(Elem<T>::load(rp + c));
      const size_t o = (n * 3 + c) * HW + p;
      const float go = d_out != nullptr ? d_out[o] : 0.f;
      dp += go * (fg - x_in[o]);
      dr[c] = go * pr * (1.f - fg * fg);
      if (d_x != nullptr) d_x[o] = go * (1.f - pr);
    }
    dr[3] = dp * pr * (1.f - pr);
    T* dst = d_raw + i * Cs;
    for (int v = 0; v < Cs; v += VEC) {
      float f[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) f[e] = (v + e) < 4 ? dr[(v + e) & 3] : 0.f;
      *reinterpret_cast<u32x4*>(dst + v) = Elem<T>::pack(f);
    }
  }
}

// ---- NaN guard (generator.py:266-267) ----
template <typename T>
__global__ void nan_flag_kernel(const T* __restrict__ x, size_t nvec, int* __restrict__ flag) {
  constexpr int VEC = Elem<T>::VEC;
  int found = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    float f[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(x + i * VEC), f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) found |= (f[e] != f[e]) ? 1 : 0;
  }
  if (__any(found)) {
    if ((threadIdx.x & 63) == 0) atomicOr(flag, 1);
  }
}

template <typename T>
__global__ void nan_to_num_kernel(T* __restrict__ x, size_t nvec, const int* __restrict__ flag) {
  constexpr int VEC = Elem<T>::VEC;
  if (*flag == 0) return;
  const float big = sizeof(T) == 2 ? 3.3895313892515355e38f : 3.4028234663852886e38f;   // dtype max (bf16 / f32)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    float f[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(x + i * VEC), f);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      if (f[e] != f[e]) f[e] = 0.f;
      else if (isinf(f[e])) f[e] = f[e] > 0.f ? big : -big;
    }
    *reinterpret_cast<u32x4*>(x + i * VEC) = Elem<T>::pack(f);
  }
}

template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    Elem<T>::store(dst + i, src[i]);
}

// nn.AvgPool2d(2, 2) on NHWC (the down-scaling ResBlocks of the conv StyleExtractor, architecture.py:157-168): fp32 sum of the 2x2
// cell, one rounding; backward spreads dout / 4 over the cell
template <typename T>
__global__ void avgpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int Ho, int Wo, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int cv = C / VEC;
  const size_t total = (size_t)N * Ho * Wo * cv;
  const size_t row = (size_t)2 * Wo * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    size_t r = i / cv;
    const int w = (int)(r % Wo); r /= Wo;
    const int h = (int)(r % Ho);
    const int n = (int)(r / Ho);
    const T* p = x + ((size_t)n * 2 * Ho + 2 * h) * row + (size_t)2 * w * C + c;
    float a[VEC], b[VEC], cc[VEC], d[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(p), a);
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(p + C), b);
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(p + row), cc);
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(p + row + C), d);
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[e] = ((a[e] + b[e]) + (cc[e] + d[e])) * 0.25f;
    *reinterpret_cast<u32x4*>(out + i * VEC) = Elem<T>::pack(a);
  }
}

template <typename T>
__global__ void avgpool2_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, int N, int Ho, int Wo, int C) {
  constexpr int VEC = Elem<T>::VEC;
  const int cv = C / VEC;
  const size_t total = (size_t)N * Ho * Wo * cv;
  const size_t row = (size_t)2 * Wo * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    size_t r = i / cv;
    const int w = (int)(r % Wo); r /= Wo;
    const int h = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float g[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(dout + i * VEC), g);
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] *= 0.25f;
    const u32x4 q = Elem<T>::pack(g);
    T* p = dx + ((size_t)n * 2 * Ho + 2 * h) * row + (size_t)2 * w * C + c;
    *reinterpret_cast<u32x4*>(p) = q;
    *reinterpret_cast<u32x4*>(p + C) = q;
    *reinterpret_cast<u32x4*>(p + row) = q;
    *reinterpret_cast<u32x4*>(p + row + C) = q;
  }
}

}  // namespace dei2i

using namespace dei2i;

static inline int vec_of(int dtype) { return dtype == DT_BF16 ? 8 : 4; }
static const unsigned EW_CAP = 256u * 16u;

extern "C" {

int dei2i_nchw_to_nhwc_resize(int dtype, int N, int C, int hs, int ws, int H, int W, int Cs, const float* src, void* dst,
                              dei2i_stream s) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || hs <= 0 || ws <= 0 || Cs < C || Cs % vec_of(dtype) || !src || !dst)
    return DEI2I_ERR_BAD_ARG;
  const size_t total = (size_t)N * H * W * (Cs / vec_of(dtype));
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(grid_for(total, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s, src,
                       (bf16_t*)dst, N, C, hs, ws, H, W, Cs);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s, src,
                       (float*)dst, N, C, hs, ws, H, W, Cs);
  return (int)hipGetLastError();
}

int dei2i_nchw_to_nhwc(int dtype, int N, int C, int H, int W, int Cs, const float* src, void* dst, dei2i_stream s) {
  return dei2i_nchw_to_nhwc_resize(dtype, N, C, H, W, H, W, Cs, src, dst, s);
}

int dei2i_nhwc_to_nchw(int dtype, int N, int C, int H, int W, int Cs, const void* src, float* dst, dei2i_stream s) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cs < C || !src || !dst) return DEI2I_ERR_BAD_ARG;
  const size_t total = (size_t)N * C * H * W;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(grid_for(total, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s,
                       (const bf16_t*)src, dst, N, C, H, W, Cs);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(total, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s,
                       (const float*)src, dst, N, C, H, W, Cs);
  return (int)hipGetLastError();
}

int dei2i_affine_act_fwd(int dtype, size_t pixels, int C, const void* x, const float* a, const float* b, const void* res,
                         int act, void* out, void* out_e4m3, float e4m3_scale, dei2i_stream s) {
  const int vec = vec_of(dtype);
  if (pixels == 0 || C <= 0 || C % vec || !x || !a || !b || !out) return DEI2I_ERR_BAD_ARG;
  if (out_e4m3 != nullptr && (dtype != DT_BF16 || !(e4m3_scale > 0.f))) return DEI2I_ERR_BAD_ARG;
  unsigned char* o8 = (unsigned char*)out_e4m3;
  const size_t nvec = pixels * (size_t)(C / vec);
  const int cv = C / vec;
  const bool inv = (256 % cv) == 0;
  const unsigned grid = grid_for(nvec, 256, EW_CAP);
  hipStream_t st = (hipStream_t)s;
  if (dtype == DT_BF16) {
    if (inv) hipLaunchKernelGGL((affine_act_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, a, b, (const bf16_t*)res, (bf16_t*)out, nvec, cv, act, o8, e4m3_scale);
    else hipLaunchKernelGGL((affine_act_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, a, b, (const bf16_t*)res, (bf16_t*)out, nvec, cv, act, o8, e4m3_scale);
  } else {
    if (inv) hipLaunchKernelGGL((affine_act_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float*)x, a, b, (const float*)res, (float*)out, nvec, cv, act, o8, e4m3_scale);
    else hipLaunchKernelGGL((affine_act_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float*)x, a, b, (const float*)res, (float*)out, nvec, cv, act, o8, e4m3_scale);
  }
  return (int)hipGetLastError();
}

/* `groups` groups of `pixels` pixels each, coefficient rows a, b: (groups, C) -- one launch (grid.y = group) */
int dei2i_affine_act_groups_fwd(int dtype, int groups, size_t pixels, int C, const void* x, const float* a, const float* b, const void* res,
                                int act, void* out, void* out_e4m3, float e4m3_scale, dei2i_stream s) {
  const int vec = vec_of(dtype);
  if (groups <= 0 || pixels == 0 || C <= 0 || C % vec || !x || !a || !b || !out) return DEI2I_ERR_BAD_ARG;
  if (out_e4m3 != nullptr && (dtype != DT_BF16 || !(e4m3_scale > 0.f))) return DEI2I_ERR_BAD_ARG;
  unsigned char* o8 = (unsigned char*)out_e4m3;
  const size_t nvec = pixels * (size_t)(C / vec);
  const int cv = C / vec;
  const bool inv = (256 % cv) == 0;
  const unsigned grid = grid_for(nvec, 256, EW_CAP);
  hipStream_t st = (hipStream_t)s;
  if (dtype == DT_BF16) {
    if (inv) hipLaunchKernelGGL((affine_act_kernel<bf16_t, true>), dim3(grid, groups), dim3(256), 0, st, (const bf16_t*)x, a, b, (const bf16_t*)res, (bf16_t*)out, nvec, cv, act, o8, e4m3_scale);
    else hipLaunchKernelGGL((affine_act_kernel<bf16_t, false>), dim3(grid, groups), dim3(256), 0, st, (const bf16_t*)x, a, b, (const bf16_t*)res, (bf16_t*)out, nvec, cv, act, o8, e4m3_scale);
  } else {
    if (inv) hipLaunchKernelGGL((affine_act_kernel<float, true>), dim3(grid, groups), dim3(256), 0, st, (const float*)x, a, b, (const float*)res, (float*)out, nvec, cv, act, o8, e4m3_scale);
    else hipLaunchKernelGGL((affine_act_kernel<float, false>), dim3(grid, groups), dim3(256), 0, st, (const float*)x, a, b, (const float*)res, (float*)out, nvec, cv, act, o8, e4m3_scale);
  }
  return (int)hipGetLastError();
}

int dei2i_affine_act_img_fwd(int dtype, int N, int HW, int C, const void* x, const float* A, const float* B, float slope, void* out,
                             dei2i_stream s) {
  const int vec = vec_of(dtype);
  if (N <= 0 || HW <= 0 || C <= 0 || C % vec || !x || !A || !B || !out) return DEI2I_ERR_BAD_ARG;
  const int cv = C / vec;
  if (256 % cv != 0 && cv % 256 != 0) return DEI2I_ERR_BAD_ARG;      // a thread keeps one channel vector
  const size_t nvec_img = (size_t)HW * cv;
  unsigned blocks = (unsigned)std::min<size_t>((nvec_img + 255) / 256, 512);
  if (cv > 256) blocks = (blocks / (cv / 256)) * (cv / 256);
  if (blocks < 1) blocks = cv > 256 ? cv / 256 : 1;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(affine_act_img_kernel<bf16_t>, dim3(blocks, N), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, A, B, (bf16_t*)out,
                       nvec_img, cv, slope);
  else
    hipLaunchKernelGGL(affine_act_img_kernel<float>, dim3(blocks, N), dim3(256), 0, (hipStream_t)s, (const float*)x, A, B, (float*)out,
                       nvec_img, cv, slope);
  return (int)hipGetLastError();
}

int dei2i_spade_act_fwd(int dtype, int N, int H, int W, int C, int up, const void* x, const float* mean, const float* rstd,
                        const void* gb, int gb_mode, void* out, void* out_e4m3, float e4m3_scale, dei2i_stream s) {
  const int vec = vec_of(dtype);
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || C % vec || up < 0 || up > 1 || !x || !mean || !rstd || !gb || !out)
    return DEI2I_ERR_BAD_ARG;
  if (out_e4m3 != nullptr && (dtype != DT_BF16 || !(e4m3_scale > 0.f))) return DEI2I_ERR_BAD_ARG;
  unsigned char* o8 = (unsigned char*)out_e4m3;
  if (up && ((H | W) & 1)) return DEI2I_ERR_BAD_ARG;
  if (gb_mode == 1 && (H < 4 || W < 4)) return DEI2I_ERR_BAD_ARG;
  const size_t total = (size_t)N * H * W * (C / vec);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(spade_act_kernel<bf16_t>, dim3(grid_for(total, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s,
                       (const bf16_t*)x, mean, rstd, (const bf16_t*)gb, (bf16_t*)out, N, H, W, C, up, gb_mode, o8, e4m3_scale);
  else
    hipLaunchKernelGGL(spade_act_kernel<float>, dim3(grid_for(total, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s,
                       (const float*)x, mean, rstd, (const float*)gb, (float*)out, N, H, W, C, up, gb_mode, o8, e4m3_scale);
  return (int)hipGetLastError();
}

int dei2i_act_bwd(int dtype, size_t n, const void* dz, const void* z, int act, void* g, dei2i_stream s) {
  const int vec = vec_of(dtype);
  if (n == 0 || n % vec || !dz || !z || !g) return DEI2I_ERR_BAD_ARG;
  const size_t nvec = n / vec;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(grid_for(nvec, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s,
                       (const bf16_t*)dz, (const bf16_t*)z, (bf16_t*)g, nvec, act);
  else
    hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(grid_for(nvec, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s,
                       (const float*)dz, (const float*)z, (float*)g, nvec, act);
  return (int)hipGetLastError();
}

int dei2i_compose_fwd(int dtype, int N, int H, int W, int Cs, const void* raw, const float* x_in, float* out, float* prob,
                      dei2i_stream s) {
  if (N <= 0 || H <= 0 || W <= 0 || Cs < 4 || Cs % vec_of(dtype) || !raw || !x_in || !out || !prob) return DEI2I_ERR_BAD_ARG;
  const size_t total = (size_t)N * H * W;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(compose_fwd_kernel<bf16_t>, dim3(grid_for(total, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s,
                       (const bf16_t*)raw, x_in, out, prob, N, H * W, Cs);
  else
    hipLaunchKernelGGL(compose_fwd_kernel<float>, dim3(grid_for(total, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s,
                       (const float*)raw, x_in, out, prob, N, H * W, Cs);
  return (int)hipGetLastError();
}

int dei2i_compose_bwd(int dtype, int N, int H, int W, int Cs, const void* raw, const float* x_in, const float* d_out,
                      const float* d_prob, void* d_raw, float* d_x, dei2i_stream s) {
  if (N <= 0 || H <= 0 || W <= 0 || Cs < 4 || Cs % vec_of(dtype) || !raw || !x_in || !d_raw) return DEI2I_ERR_BAD_ARG;
  const size_t total = (size_t)N * H * W;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(compose_bwd_kernel<bf16_t>, dim3(grid_for(total, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s,
                       (const bf16_t*)raw, x_in, d_out, d_prob, (bf16_t*)d_raw, d_x, N, H * W, Cs);
  else
    hipLaunchKernelGGL(compose_bwd_kernel<float>, dim3(grid_for(total, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s,
                       (const float*)raw, x_in, d_out, d_prob, (float*)d_raw, d_x, N, H * W, Cs);
  return (int)hipGetLastError();
}

int dei2i_nan_guard(int dtype, size_t n, void* x, int* flag, dei2i_stream s) {
  const int vec = vec_of(dtype);
  if (n == 0 || n % vec || !x || !flag) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), st);
  if (e != hipSuccess) return (int)e;
  const size_t nvec = n / vec;
  const unsigned grid = grid_for(nvec, 256, EW_CAP);
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL(nan_flag_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, nvec, flag);
    hipLaunchKernelGGL(nan_to_num_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (bf16_t*)x, nvec, (const int*)flag);
  } else {
    hipLaunchKernelGGL(nan_flag_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, nvec, flag);
    hipLaunchKernelGGL(nan_to_num_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)x, nvec, (const int*)flag);
  }
  return (int)hipGetLastError();
}

int dei2i_cast_from_f32(int dtype, size_t n, const float* src, void* dst, dei2i_stream s) {
  if (n == 0 || !src || !dst) return DEI2I_ERR_BAD_ARG;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, dim3(grid_for(n, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s, src,
                       (bf16_t*)dst, n);
  else
    hipLaunchKernelGGL(cast_from_f32_kernel<float>, dim3(grid_for(n, 256, EW_CAP)), dim3(256), 0, (hipStream_t)s, src,
                       (float*)dst, n);
  return (int)hipGetLastError();
}

/* nn.AvgPool2d(2, 2) on an NHWC tensor of (N, H, W, C), H and W even; out / dout: (N, H/2, W/2, C) */
int dei2i_avgpool2_fwd(int dtype, int N, int H, int W, int C, const void* x, void* out, dei2i_stream s) {
  const int vec = vec_of(dtype);
  if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || C % vec || !x || !out) return DEI2I_ERR_BAD_ARG;
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / vec);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(avgpool2_fwd_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)out,
                       N, H / 2, W / 2, C);
  else
    hipLaunchKernelGGL(avgpool2_fwd_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s, (const float*)x, (float*)out, N,
                       H / 2, W / 2, C);
  return (int)hipGetLastError();
}

int dei2i_avgpool2_bwd(int dtype, int N, int H, int W, int C, const void* dout, void* dx, dei2i_stream s) {
  const int vec = vec_of(dtype);
  if (N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || C % vec || !dout || !dx) return DEI2I_ERR_BAD_ARG;
  const size_t total = (size_t)N * (H / 2) * (W / 2) * (C / vec);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(avgpool2_bwd_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)dout, (bf16_t*)dx,
                       N, H / 2, W / 2, C);
  else
    hipLaunchKernelGGL(avgpool2_bwd_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)s, (const float*)dout, (float*)dx, N,
                       H / 2, W / 2, C);
  return (int)hipGetLastError();
}

}  // extern "C"
