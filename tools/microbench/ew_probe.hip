// Diagnostic (not part of the library): how should a 2-read + 1-write bf16 elementwise kernel with BatchNorm-backward-like
// arithmetic be shaped to reach the copy rate?  Variants over the three activation shapes of the 256x256 batch-16 step:
//   0  grid-stride loop, 2048 workgroups, two vectors (4 loads) in flight per thread   (reduce.hip: bn_bwd_apply_kernel today)
//   1  one 16-byte vector per thread, one workgroup per 256 vectors (the shape of at::native::vectorized_elementwise_kernel)
//   2  two vectors per thread, no loop
//   3  four vectors per thread, no loop
//   4  variant 1 with plain a + b (no per-channel coefficients, no activation): the traffic alone
// Build: hipcc --offload-arch=gfx950 -O3 -o ew_probe ew_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__device__ inline float lo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ inline float hi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
__device__ inline uint32_t pk(float a, float b) {
  uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
  x += 0x7fffu + ((x >> 16) & 1u); y += 0x7fffu + ((y >> 16) & 1u);
  return (x >> 16) | (y & 0xffff0000u);
}
struct Coef { float a[8], b[8], m[8], c2[8], c3[8]; };
__device__ inline void load_coef(const float* __restrict__ co, int C, int c, Coef& k) {
#pragma unroll
  for (int e = 0; e < 8; ++e) { k.a[e] = co[c + e]; k.b[e] = co[C + c + e]; k.m[e] = co[2 * C + c + e]; k.c2[e] = co[3 * C + c + e]; k.c3[e] = co[4 * C + c + e]; }
}
__device__ inline u32x4 apply(const u32x4 d, const u32x4 y, const Coef& k) {
  float dv[8] = {lo(d.x), hi(d.x), lo(d.y), hi(d.y), lo(d.z), hi(d.z), lo(d.w), hi(d.w)};
  float yv[8] = {lo(y.x), hi(y.x), lo(y.y), hi(y.y), lo(y.z), hi(y.z), lo(y.w), hi(y.w)};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float z = fmaf(k.a[e], yv[e], k.b[e]);
    const float g = dv[e] * (z >= 0.f ? 1.f : 0.2f);
    dv[e] = fmaf(k.a[e], g, fmaf(k.c2[e], yv[e] - k.m[e], k.c3[e]));
  }
  u32x4 o; o.x = pk(dv[0], dv[1]); o.y = pk(dv[2], dv[3]); o.z = pk(dv[4], dv[5]); o.w = pk(dv[6], dv[7]);
  return o;
}
__global__ __launch_bounds__(256) void v0(const u32x4* __restrict__ dz, const u32x4* __restrict__ y, u32x4* __restrict__ out,
                                          const float* __restrict__ co, int C, size_t nvec) {
  const int cv = C / 8;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  Coef k; load_coef(co, C, (int)(i % cv) * 8, k);
  for (; i + stride < nvec; i += 2 * stride) {
    const u32x4 d0 = dz[i], y0 = y[i], d1 = dz[i + stride], y1 = y[i + stride];
    out[i] = apply(d0, y0, k); out[i + stride] = apply(d1, y1, k);
  }
  if (i < nvec) out[i] = apply(dz[i], y[i], k);
}
template <int V, bool PLAIN>
__global__ __launch_bounds__(256) void v1(const u32x4* __restrict__ dz, const u32x4* __restrict__ y, u32x4* __restrict__ out,
                                          const float* __restrict__ co, int C, size_t nvec) {
  const int cv = C / 8;
  const size_t base = (size_t)blockIdx.x * 256 * V + threadIdx.x;       // 256 % cv == 0: the channel vector is the thread's
  Coef k;
  if (!PLAIN) load_coef(co, C, (int)(base % cv) * 8, k);
  u32x4 d[V], yy[V];
#pragma unroll
  for (int q = 0; q < V; ++q) { const size_t i = base + q * 256; if (i < nvec) { d[q] = dz[i]; yy[q] = y[i]; } }
#pragma unroll
  for (int q = 0; q < V; ++q) {
    const size_t i = base + q * 256;
    if (i < nvec) {
      if (PLAIN) {
        u32x4 o;
        o.x = pk(lo(d[q].x) + lo(yy[q].x), hi(d[q].x) + hi(yy[q].x)); o.y = pk(lo(d[q].y) + lo(yy[q].y), hi(d[q].y) + hi(yy[q].y));
        o.z = pk(lo(d[q].z) + lo(yy[q].z), hi(d[q].z) + hi(yy[q].z)); o.w = pk(lo(d[q].w) + lo(yy[q].w), hi(d[q].w) + hi(yy[q].w));
        out[i] = o;
      } else out[i] = apply(d[q], yy[q], k);
    }
  }
}
int main() {
  hipStream_t st; hipStreamCreate(&st);
  const int shapes[3][3] = {{64, 64, 256}, {128, 128, 128}, {256, 256, 64}};
  for (auto& s : shapes) {
    const int C = s[2];
    const size_t n = (size_t)16 * s[0] * s[1] * C, nvec = n / 8;
    u32x4 *dz, *y, *out; float* co;
    hipMalloc(&dz, n * 2); hipMalloc(&y, n * 2); hipMalloc(&out, n * 2); hipMalloc(&co, 5 * C * 4);
    hipMemset(dz, 0x3c, n * 2); hipMemset(y, 0x3d, n * 2); hipMemset(co, 0, 5 * C * 4);
    for (int var = 0; var < 5; ++var) {
      auto go = [&]() {
        if (var == 0) hipLaunchKernelGGL(v0, dim3(2048), dim3(256), 0, st, dz, y, out, co, C, nvec);
        if (var == 1) hipLaunchKernelGGL((v1<1, false>), dim3((nvec + 255) / 256), dim3(256), 0, st, dz, y, out, co, C, nvec);
        if (var == 2) hipLaunchKernelGGL((v1<2, false>), dim3((nvec + 511) / 512), dim3(256), 0, st, dz, y, out, co, C, nvec);
        if (var == 3) hipLaunchKernelGGL((v1<4, false>), dim3((nvec + 1023) / 1024), dim3(256), 0, st, dz, y, out, co, C, nvec);
        if (var == 4) hipLaunchKernelGGL((v1<1, true>), dim3((nvec + 255) / 256), dim3(256), 0, st, dz, y, out, co, C, nvec);
      };
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) go();
      hipStreamSynchronize(st);
      hipEventRecord(e0, st);
      for (int i = 0; i < 20; ++i) go();
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / 20;
      printf("%dx%dx%d variant %d: %7.1f us  %5.2f TB/s\n", s[0], s[1], C, var, us, 3.0 * n * 2 / us / 1e6);
    }
    hipFree(dz); hipFree(y); hipFree(out); hipFree(co);
  }
  return 0;
}
