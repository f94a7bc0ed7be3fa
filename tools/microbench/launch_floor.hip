// Diagnostic (not part of the library): what does one more small launch cost on a busy stream?
//   (a) back-to-back empty kernels (1 workgroup)          -> dispatch throughput of the queue
//   (b) back-to-back tiny kernels that touch 256 CUs       -> same with a full-width grid
//   (c) a 100 us kernel followed by k tiny kernels, per k  -> does a tiny launch hide behind its predecessor?
// Build: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }
__global__ void tiny_kernel(float* p) { p[blockIdx.x * 256 + threadIdx.x] += 1.f; }
__global__ void busy_kernel(float* p, int iters) {
  float a = p[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  p[blockIdx.x * 256 + threadIdx.x] = a;
}
static float timed(hipStream_t st, int reps, void (*body)(hipStream_t, void*), void* arg) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  body(st, arg); hipStreamSynchronize(st);
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) body(st, arg);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / reps;
}
static float* g_buf; static int g_k;
int main() {
  hipStream_t st; hipStreamCreate(&st);
  hipMalloc(&g_buf, 1024 * 256 * 4); hipMemset(g_buf, 0, 1024 * 256 * 4);
  float a = timed(st, 5000, [](hipStream_t s, void*) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr); }, nullptr);
  float b = timed(st, 5000, [](hipStream_t s, void*) { hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, s, g_buf); }, nullptr);
  printf("back-to-back empty kernels      : %.2f us per launch\n", a);
  printf("back-to-back 256-workgroup tiny : %.2f us per launch\n", b);
  for (int k = 0; k <= 8; k += 2) {
    g_k = k;
    float c = timed(st, 300, [](hipStream_t s, void*) {
      hipLaunchKernelGGL(busy_kernel, dim3(1024), dim3(256), 0, s, g_buf, 60000);
      for (int i = 0; i < g_k; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, s, g_buf);
    }, nullptr);
    printf("busy kernel + %d tiny launches   : %.2f us per group\n", k, c);
  }
  return 0;
}
