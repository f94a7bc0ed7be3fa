// Diagnostic (not part of the library): what a bare bf16 MFMA loop sustains on this GPU, on random operands, and the
// in-kernel clock it holds (delta s_memtime / delta s_memrealtime x 100 MHz).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(const unsigned* __restrict__ seed, float* __restrict__ sink, int iters,
                                                 unsigned long long* __restrict__ stamps) {
  const int tid = threadIdx.x;
  unsigned s = seed[(blockIdx.x * 512 + tid) & 4095];
  bf16x8 a[2], b[2];
  for (int i = 0; i < 2; ++i) {
    unsigned v[4], w[4];
    for (int q = 0; q < 4; ++q) { s = s * 1664525u + 1013904223u; v[q] = (s & 0x3fff3fffu) | 0x3c003c00u; s = s * 1664525u + 1013904223u; w[q] = (s & 0x3fff3fffu) | 0x3c003c00u; }
    a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(v));
    b[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<uint4*>(w));
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (SHAPE == 32) {
    f32x16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j & 1], b[(j >> 1) & 1], acc[j], 0, 0, 0);
    }
    float r = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) r += acc[j][e];
    sink[blockIdx.x * 512 + tid] = r;
  } else {
    f32x4 acc[16] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 1], b[(j >> 1) & 1], acc[j], 0, 0, 0);
    }
    float r = 0.f;
    for (int j = 0; j < 16; ++j) for (int e = 0; e < 4; ++e) r += acc[j][e];
    sink[blockIdx.x * 512 + tid] = r;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int blocks = 256;
  unsigned* seed; float* sink; unsigned long long* stamps;
  hipMalloc(&seed, 4096 * 4); hipMalloc(&sink, blocks * 512 * 4); hipMalloc(&stamps, blocks * 16);
  std::vector<unsigned> h(4096); for (auto& x : h) x = rand();
  hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  for (int shape : {32, 16}) for (int wpb : {512, 256}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (shape == 32) hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(wpb), 0, 0, seed, sink, iters, stamps);
      else hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(wpb), 0, 0, seed, sink, iters, stamps);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> st(blocks * 2); hipMemcpy(st.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
    const double flop_per_mfma = shape == 32 ? 2.0 * 32 * 32 * 16 : 2.0 * 16 * 16 * 32;
    const double flops = (double)blocks * (wpb / 64) * iters * 16 * flop_per_mfma;
    printf("mfma %dx%d  %d waves/CU  %.3f ms  %.1f TF/s   in-kernel clock %.2f GHz (block 0) cycles/MFMA/SIMD %.1f\n", shape, shape, wpb / 64, ms,
           flops / ms / 1e9, (double)st[0] / (double)st[1] * 0.1, (double)st[0] / ((double)iters * 16 * (wpb / 256)));
  }
  return 0;
}
