// Spectral normalisation of a conv weight (torch.nn.utils.spectral_norm as the reference applies it with --use_spectral,
// architecture.py:68-72,109-112,238-239,338-341), W = weight_orig viewed as a (Cout, K) fp32 matrix, K = Cin*kh*kw:
//   training forward:  t = W^T u;  v = t / max(|t|, eps);  s = W v;  u = s / max(|s|, eps);  sigma = u . s
//   eval forward    :  s = W v;  sigma = u . s            (stored u, v as they are)
//   w_eff = W / sigma
//   backward (u, v constants):  dW = G / sigma - (sum(G . W) / sigma^2) * u v^T
// As plain torch ops this is ~14 small launches per conv forward and ~11 per backward (3 000 launches per step with the
// README recipe options); here it is 3 / 2 launches forward and 2 backward, all reductions in a fixed order
// (deterministic).  The in-place (u, v) buffers are updated by the kernels; the vectors sigma was computed with are also
// written to u_used / v_used, which the backward pass reads (later forwards iterate the buffers again).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dei2i_hip.h"
#include "common.h"
#include "launch.h"

namespace dei2i {

constexpr float SN_EPS = 1e-12f;
constexpr int SN_COLS = 64;             // columns per workgroup of the W^T u pass

DEI2I_D float sn_block_sum(float v, float* red) {       // 256 threads, fixed order
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// t[k] = sum_r W[r][k] u[r]; part[blockIdx.x] = sum over this block's columns of t[k]^2.
// 64 columns per workgroup, the rows dealt to its sixteen waves (wave g takes rows g, g+16, ...): K = 2304 is only 36
// such workgroups, so the launch is bound by each thread's chain of dependent loads -- 16 row groups keep that chain at
// Cout/16 loads (four independent accumulators), and every wave still reads 256 contiguous bytes per row.
constexpr int SN_RG = 16;
__global__ __launch_bounds__(SN_COLS * SN_RG) void sn_wt_u_kernel(const float* __restrict__ W, const float* __restrict__ u, int Cout,
                                                                  int K, float* __restrict__ t, float* __restrict__ part) {
  __shared__ float accg[SN_RG][SN_COLS];
  const int col = threadIdx.x & (SN_COLS - 1), g = threadIdx.x / SN_COLS;
  const int k = blockIdx.x * SN_COLS + col;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (k < K) {
    int r = g;
    for (; r + 3 * SN_RG < Cout; r += 4 * SN_RG) {
      a0 = fmaf(W[(size_t)r * K + k], u[r], a0);
      a1 = fmaf(W[(size_t)(r + SN_RG) * K + k], u[r + SN_RG], a1);
      a2 = fmaf(W[(size_t)(r + 2 * SN_RG) * K + k], u[r + 2 * SN_RG], a2);
      a3 = fmaf(W[(size_t)(r + 3 * SN_RG) * K + k], u[r + 3 * SN_RG], a3);
    }
    for (; r < Cout; r += SN_RG) a0 = fmaf(W[(size_t)r * K + k], u[r], a0);
  }
  accg[g][col] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (g == 0) {                       // wave 0 holds the block's 64 columns: its wave sum is the block's partial
    float tk = 0.f;
#pragma unroll
    for (int i = 0; i < SN_RG; ++i) tk += accg[i][col];
    if (k < K) t[k] = tk; else tk = 0.f;
    const float s = wave_sum(tk * tk);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
  }
}

// The two normalisations of the power iteration are scalings, and s = W v is linear in v: with t = W^T u un-normalised,
//   v = t / max(|t|, eps),   s = W v = (W t) / max(|t|, eps),   u' = s / max(|s|, eps),   sigma = u' . s = |s|^2 / max(|s|, eps)
// so no pass has to wait for a normalised vector to exist: the W t pass applies 1 / max(|t|, eps) to its row sums (every
// workgroup sums the |t|^2 partials itself, same fixed pattern, bit-identical across workgroups) and writes its slice of v; the
// scale pass sums the |s|^2 partials the same way, writes its slice of u', and divides W by sigma.  3 launches instead of 5.
DEI2I_D float sn_sum_parts(const float* __restrict__ part, int nparts, float* red) {
  float p = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) p += part[i];
  return sn_block_sum(p, red);
}

// s[r] = inv * sum_k W[r][k] x[k]  (one workgroup per row), inv = 1 / max(sqrt(sum tpart), eps) when tpart != nullptr (training:
// x = t un-normalised; the workgroup also writes its slice of v = x * inv to v and v_used, and workgroup 0 scal[0] = |t|), else 1
// (eval: x = the stored v).  part[r] = s[r]^2 (training) or u[r] * s[r] (eval: sigma partials).
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* __restrict__ W, const float* __restrict__ x, const float* __restrict__ u,
                                                     int Cout, int K, const float* __restrict__ tpart, int ntparts,
                                                     float* __restrict__ v, float* __restrict__ v_used, float* __restrict__ scal,
                                                     float* __restrict__ s, float* __restrict__ part) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  float inv = 1.f;
  if (tpart != nullptr) {
    const float norm = sqrtf(sn_sum_parts(tpart, ntparts, red));
    inv = 1.f / fmaxf(norm, SN_EPS);
    const int per = (K + Cout - 1) / Cout;                               // this workgroup's slice of v
    for (int k = r * per + threadIdx.x; k < min(K, (r + 1) * per); k += 256) {
      const float vk = x[k] * inv;
      v[k] = vk;
      v_used[k] = vk;
    }
    if (r == 0 && threadIdx.x == 0) scal[0] = norm;
    __syncthreads();
  }
  const float* row = W + (size_t)r * K;
  float a0 = 0.f, a1 = 0.f;
  int k = threadIdx.x;
  for (; k + 256 < K; k += 512) {
    a0 = fmaf(row[k], x[k], a0);
    a1 = fmaf(row[k + 256], x[k + 256], a1);
  }
  if (k < K) a0 = fmaf(row[k], x[k], a0);
  const float sr = sn_block_sum(a0 + a1, red) * inv;
  if (threadIdx.x == 0) {
    s[r] = sr;
    part[r] = tpart != nullptr ? sr * sr : u[r] * sr;
  }
}

// out = W / sigma, sigma from the Cout partials of the pass before (every workgroup sums them itself):
//   training: tot = sum s^2, sigma = tot / max(sqrt(tot), eps); the workgroup writes its slice of u = s / max(|s|, eps) to u and u_used
//   eval    : sigma = sum u[r] s[r]; the workgroup copies its slices of the stored u, v to u_used, v_used
// scal[1] = |s| (training), scal[2] = sigma.  n4 = n / 4 float4 groups, then the (< 4) tail elements.
__global__ __launch_bounds__(256) void sn_scale_kernel(const float* __restrict__ W, const float* __restrict__ spart, const float* __restrict__ sv,
                                                       int Cout, int K, int training, float* __restrict__ u, const float* __restrict__ v,
                                                       float* __restrict__ u_used, float* __restrict__ v_used, float* __restrict__ scal,
                                                       size_t n, float* __restrict__ out) {
  __shared__ float red[4];
  const float tot = sn_sum_parts(spart, Cout, red);
  float sigma = tot;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (training) {
    const float norm = sqrtf(tot);
    const float inv_s = 1.f / fmaxf(norm, SN_EPS);
    sigma = tot * inv_s;                                                 // u . s with u = s / max(|s|, eps)
    for (size_t i = tid; i < (size_t)Cout; i += stride) {
      const float ur = sv[i] * inv_s;
      u[i] = ur;
      u_used[i] = ur;
    }
    if (tid == 0) scal[1] = norm;
  } else {
    for (size_t i = tid; i < (size_t)(Cout + K); i += stride) {
      if (i < (size_t)Cout) u_used[i] = u[i];
      else v_used[i - Cout] = v[i - Cout];
    }
  }
  if (tid == 0) scal[2] = sigma;
  const float inv = 1.f / sigma;
  const size_t n4 = n >> 2;
  for (size_t i = tid; i < n4; i += stride) {
    float4 w = reinterpret_cast<const float4*>(W)[i];
    w.x *= inv; w.y *= inv; w.z *= inv; w.w *= inv;
    reinterpret_cast<float4*>(out)[i] = w;
  }
  for (size_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = W[i] * inv;
}

// part[block] = sum over the block's elements of G . W_eff   (W_eff = W / sigma, so sum(G.W) = sigma * sum(G.W_eff))
__global__ __launch_bounds__(256) void sn_bwd_dot_kernel(const float* __restrict__ G, const float* __restrict__ Weff, size_t n,
                                                         float* __restrict__ part) {
  __shared__ float red[4];
  const size_t n4 = n >> 2, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (size_t i = tid; i < n4; i += stride) {
    const float4 g = reinterpret_cast<const float4*>(G)[i], w = reinterpret_cast<const float4*>(Weff)[i];
    a0 = fmaf(g.x, w.x, a0); a1 = fmaf(g.y, w.y, a1); a2 = fmaf(g.z, w.z, a2); a3 = fmaf(g.w, w.w, a3);
  }
  for (size_t i = (n4 << 2) + tid; i < n; i += stride) a0 = fmaf(G[i], Weff[i], a0);
  const float s = sn_block_sum((a0 + a1) + (a2 + a3), red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// dW[r][k] (+)= (G[r][k] - c * u[r] v[k]) / sigma,  c = sum(G . W_eff).  K4: K is a multiple of 4 -> float4 groups that
// never straddle a row.
template <bool K4>
__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(const float* __restrict__ G, const float* __restrict__ part, int nparts,
                                                           const float* __restrict__ scal, const float* __restrict__ u,
                                                           const float* __restrict__ v, int Cout, int K, float* __restrict__ dW,
                                                           int accumulate) {
  __shared__ float red[4];
  float p = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) p += part[i];
  const float c = sn_block_sum(p, red);
  const float inv = 1.f / scal[2];
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  if (K4) {
    const int k4n = K >> 2;
    const size_t n4 = (size_t)Cout * k4n;
    for (size_t i = tid; i < n4; i += stride) {
      const int r = (int)(i / k4n), k4 = (int)(i - (size_t)r * k4n);
      const float cu = c * u[r];
      const float4 g = reinterpret_cast<const float4*>(G)[i], vv = reinterpret_cast<const float4*>(v)[k4];
      float4 o;
      o.x = (g.x - cu * vv.x) * inv; o.y = (g.y - cu * vv.y) * inv; o.z = (g.z - cu * vv.z) * inv; o.w = (g.w - cu * vv.w) * inv;
      if (accumulate) {
        const float4 old = reinterpret_cast<const float4*>(dW)[i];
        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
      }
      reinterpret_cast<float4*>(dW)[i] = o;
    }
  } else {
    const size_t n = (size_t)Cout * K;
    for (size_t i = tid; i < n; i += stride) {
      const int r = (int)(i / K), k = (int)(i - (size_t)r * K);
      const float o = (G[i] - c * u[r] * v[k]) * inv;
      dW[i] = accumulate ? dW[i] + o : o;
    }
  }
}

static inline int sn_dot_blocks(size_t n) {            // >= 2048 elements (two float4 pairs per thread) per workgroup
  size_t b = n / 2048;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace dei2i

using namespace dei2i;

// Eval-mode BatchNorm folded into the conv in front of it (architecture.py:116-118 with running statistics: z = act(a * conv(x, W)
// + b), a = weight * rsqrt(running_var + eps), b = bias - running_mean * a, a fixed per-channel affine): W_eff[co] = a[co] * W[co],
// b_eff = b -- the conv's own epilogue (bias + activation) then writes z, and the BatchNorm-apply pass over the conv's output is
// not run.  One launch per conv: row co of W (K floats) scaled by a[co]; block 0 also writes b_eff.
__global__ __launch_bounds__(256) void fold_bn_weight_kernel(const float* __restrict__ W, const float* __restrict__ bn_w,
                                                             const float* __restrict__ bn_b, const float* __restrict__ rm,
                                                             const float* __restrict__ rv, const float eps, const int Cout, const int K,
                                                             float* __restrict__ w_eff, float* __restrict__ b_eff) {
  const size_t n = (size_t)Cout * K;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int co = (int)(i / K);
    w_eff[i] = W[i] * (bn_w[co] * rsqrtf(rv[co] + eps));
  }
  if (blockIdx.x == 0)
    for (int co = threadIdx.x; co < Cout; co += 256) {
      const float a = bn_w[co] * rsqrtf(rv[co] + eps);
      b_eff[co] = bn_b[co] - rm[co] * a;
    }
}

extern "C" {

size_t dei2i_spectral_scratch_floats(int Cout, int K) { return (size_t)K + Cout + (size_t)((K + SN_COLS - 1) / SN_COLS) + Cout + 1024; }

int dei2i_spectral_fwd(int Cout, int K, const float* W, float* u, float* v, int iterate, float* scratch, float* u_used,
                       float* v_used, float* scal, float* w_eff, dei2i_stream s) {
  if (Cout <= 0 || K <= 0 || !W || !u || !v || !scratch || !u_used || !v_used || !scal || !w_eff) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const int cb = (K + SN_COLS - 1) / SN_COLS;
  float* t = scratch;                    // K
  float* sv = scratch + K;               // Cout
  float* part_t = sv + Cout;             // cb
  float* part_s = part_t + cb;           // Cout
  const size_t n = (size_t)Cout * K;
  if (iterate) {
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3(cb), dim3(SN_COLS * SN_RG), 0, st, W, (const float*)u, Cout, K, t, part_t);
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(Cout), dim3(256), 0, st, W, (const float*)t, (const float*)u, Cout, K, (const float*)part_t, cb,
                       v, v_used, scal, sv, part_s);
  } else {
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(Cout), dim3(256), 0, st, W, (const float*)v, (const float*)u, Cout, K, (const float*)nullptr, 0,
                       (float*)nullptr, (float*)nullptr, scal, sv, part_s);
  }
  hipLaunchKernelGGL(sn_scale_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, st, W, (const float*)part_s, (const float*)sv, Cout, K,
                     iterate ? 1 : 0, u, (const float*)v, u_used, v_used, scal, n, w_eff);
  return (int)hipGetLastError();
}

int dei2i_spectral_bwd(int Cout, int K, const float* G, const float* w_eff, const float* u_used, const float* v_used,
                       const float* scal, float* scratch, float* dW, int accumulate, dei2i_stream s) {
  if (Cout <= 0 || K <= 0 || !G || !w_eff || !u_used || !v_used || !scal || !scratch || !dW) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const size_t n = (size_t)Cout * K;
  const int nb = sn_dot_blocks(n);                 // <= 1024 partials at the head of `scratch`
  hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(nb), dim3(256), 0, st, G, w_eff, n, scratch);
  if (K % 4 == 0)
    hipLaunchKernelGGL(sn_bwd_apply_kernel<true>, dim3(grid_for(n / 4, 256)), dim3(256), 0, st, G, (const float*)scratch, nb, scal,
                       u_used, v_used, Cout, K, dW, accumulate);
  else
    hipLaunchKernelGGL(sn_bwd_apply_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, st, G, (const float*)scratch, nb, scal,
                       u_used, v_used, Cout, K, dW, accumulate);
  return (int)hipGetLastError();
}


int dei2i_fold_bn_weight(int Cout, int K, const float* W, const float* bn_weight, const float* bn_bias, const float* running_mean,
                         const float* running_var, float eps, float* w_eff, float* b_eff, dei2i_stream s) {
  if (Cout <= 0 || K <= 0 || !W || !bn_weight || !bn_bias || !running_mean || !running_var || !w_eff || !b_eff) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(fold_bn_weight_kernel, dim3(grid_for((size_t)Cout * K, 256)), dim3(256), 0, (hipStream_t)s, W, bn_weight, bn_bias,
                     running_mean, running_var, eps, Cout, K, w_eff, b_eff);
  return (int)hipGetLastError();
}

}  // extern "C"
