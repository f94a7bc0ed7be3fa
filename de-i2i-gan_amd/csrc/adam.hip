// Fused multi-tensor Adam / AdamW: one launch updates every parameter tensor of a network.
// torch.optim.Adam single-tensor semantics (no amsgrad), reference call site trainers/base_trainer.py:75-89
// (Adam betas (0.5, 0.999); AdamW betas (0.9, 0.95) with torch's default decoupled weight decay 1e-2 for the MAE stage):
//   p *= 1 - lr*wd  (AdamW only: keep = 1 - lr*wd, 1 for Adam)
//   m += (1-b1)(g-m); v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#include <hip/hip_runtime.h>

#include "../../include/dei2i_hip.h"
#include "launch.h"

namespace dei2i {

__global__ __launch_bounds__(256) void adam_kernel(const dei2i_adam_rec* __restrict__ table, float lr, float beta1,
                                                   float beta2, float eps, float bias_c1, float bias_c2_sqrt,
                                                   float grad_scale, float keep) {
  const dei2i_adam_rec rec = table[blockIdx.y];
  const float step_size = lr / bias_c1;
  const float inv_c2 = 1.f / bias_c2_sqrt;
  const int64_t n = rec.n;
  const int64_t n4 = ((reinterpret_cast<uintptr_t>(rec.p) | reinterpret_cast<uintptr_t>(rec.g) |
                       reinterpret_cast<uintptr_t>(rec.m) | reinterpret_cast<uintptr_t>(rec.v)) & 15) == 0 ? n / 4 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 p = reinterpret_cast<float4*>(rec.p)[i];
    const float4 g = reinterpret_cast<const float4*>(rec.g)[i];
    float4 m = reinterpret_cast<float4*>(rec.m)[i];
    float4 v = reinterpret_cast<float4*>(rec.v)[i];
    float* pp = &p.x; const float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gv = gp[e] * grad_scale;
      mp[e] = mp[e] + (1.f - beta1) * (gv - mp[e]);
      vp[e] = beta2 * vp[e] + (1.f - beta2) * gv * gv;
      const float denom = sqrtf(vp[e]) * inv_c2 + eps;
      pp[e] = pp[e] * keep - step_size * (mp[e] / denom);
    }
    reinterpret_cast<float4*>(rec.p)[i] = p;
    reinterpret_cast<float4*>(rec.m)[i] = m;
    reinterpret_cast<float4*>(rec.v)[i] = v;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gv = rec.g[i] * grad_scale;
    const float m = rec.m[i] + (1.f - beta1) * (gv - rec.m[i]);
    const float v = beta2 * rec.v[i] + (1.f - beta2) * gv * gv;
    rec.m[i] = m;
    rec.v[i] = v;
    rec.p[i] = rec.p[i] * keep - step_size * (m / (sqrtf(v) * inv_c2 + eps));
  }
}

// torch.optim.SGD / torch.optim.RMSprop with the defaults the reference constructs them with (trainers/base_trainer.py:71-74: only
// lr is passed -- no momentum, no weight decay, RMSprop alpha 0.99, eps 1e-8, not centered), same pointer table (RMSprop's
// square average lives in rec.m; rec.v is not touched):
//   kind 0  p -= lr * g                       kind 1  sq = alpha sq + (1 - alpha) g^2;  p -= lr * g / (sqrt(sq) + eps)
__global__ __launch_bounds__(256) void sgd_rmsprop_kernel(const dei2i_adam_rec* __restrict__ table, int kind, float lr, float alpha,
                                                          float eps, float grad_scale) {
  const dei2i_adam_rec rec = table[blockIdx.y];
  const int64_t n = rec.n, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gv = rec.g[i] * grad_scale;
    if (kind == 0) {
      rec.p[i] -= lr * gv;
    } else {
      const float sq = alpha * rec.m[i] + (1.f - alpha) * gv * gv;
      rec.m[i] = sq;
      rec.p[i] -= lr * (gv / (sqrtf(sq) + eps));
    }
  }
}

}  // namespace dei2i

using namespace dei2i;

extern "C" int dei2i_sgd_rmsprop_step(const dei2i_adam_rec* table_dev, int count, int64_t max_n, int kind, float lr, float alpha,
                                      float eps, float grad_scale, dei2i_stream s) {
  if (!table_dev || count <= 0 || max_n <= 0 || (kind != 0 && kind != 1)) return DEI2I_ERR_BAD_ARG;
  int64_t bx = (max_n + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(sgd_rmsprop_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, (hipStream_t)s, table_dev, kind, lr, alpha, eps,
                     grad_scale);
  return (int)hipGetLastError();
}

extern "C" int dei2i_adam_step(const dei2i_adam_rec* table_dev, int count, int64_t max_n, float lr, float beta1, float beta2,
                               float eps, float bias_c1, float bias_c2_sqrt, float grad_scale, float decoupled_decay,
                               dei2i_stream s) {
  if (!table_dev || count <= 0 || max_n <= 0) return DEI2I_ERR_BAD_ARG;
  int64_t bx = (max_n / 4 + 255) / 256;
  if (bx < 1) bx = 1;
  if (bx > 512) bx = 512;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, (hipStream_t)s, table_dev, lr, beta1, beta2,
                     eps, bias_c1, bias_c2_sqrt, grad_scale, 1.f - lr * decoupled_decay);
  return (int)hipGetLastError();
}
