// Diagnostic (not part of the library): how fast can a CU move an L2-resident 16 KB tile into LDS, per iteration,
//   (a) with LDS-DMA  (global_load_lds_dwordx4: the halo conv's weight ring today),
//   (b) with plain global_load_dwordx4 into VGPRs followed by ds_write_b128 (register staging)?
// 8 waves per workgroup, one workgroup per CU, every workgroup walks the same 2.4 MB "weight" buffer (L2 / MALL hits),
// each iteration = one 16 KB stage (2 KB per wave = two 16-byte loads per lane).  Prints shader cycles (s_memtime) and ns
// (s_memrealtime, 100 MHz) per stage.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_feed lds_feed.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

template <int MODE>
__global__ __launch_bounds__(512) void feed(const unsigned char* __restrict__ w, size_t wbytes, int iters, unsigned* __restrict__ sink,
                                            unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // 4 stages x 16 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t nstage = wbytes / 16384;
  unsigned acc = 0;
  u32x4 r0 = {0, 0, 0, 0}, r1 = r0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    const unsigned char* src = w + ((size_t)it % nstage) * 16384;
    unsigned char* dst = smem + (it & 3) * 16384;
    if (MODE == 0) {
      __builtin_amdgcn_global_load_lds((gbl_void*)(src + (wave * 2 + 0) * 1024 + lane * 16), (lds_void*)(dst + (wave * 2 + 0) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void*)(src + (wave * 2 + 1) * 1024 + lane * 16), (lds_void*)(dst + (wave * 2 + 1) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");        // the previous stage has landed
    } else {
      // write the PREVIOUS iteration's registers to LDS, then issue this iteration's loads (one stage in flight)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned char* pdst = smem + ((it + 3) & 3) * 16384;
      *reinterpret_cast<u32x4*>(pdst + (wave * 2 + 0) * 1024 + lane * 16) = r0;
      *reinterpret_cast<u32x4*>(pdst + (wave * 2 + 1) * 1024 + lane * 16) = r1;
      r0 = *reinterpret_cast<const u32x4*>(src + (wave * 2 + 0) * 1024 + lane * 16);
      r1 = *reinterpret_cast<const u32x4*>(src + (wave * 2 + 1) * 1024 + lane * 16);
    }
    __builtin_amdgcn_s_barrier();
    // touch the stage that landed two iterations ago so nothing is optimised away
    acc += *reinterpret_cast<const unsigned*>(smem + ((it + 2) & 3) * 16384 + ((tid * 16 + it * 4) & 16383));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
  sink[blockIdx.x * 512 + tid] = acc + r0.x + r1.y;
  if (tid == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = q1 - q0; }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 4000;
  const int blocks = 256;
  const size_t wbytes = 2359296;          // 256 x 9 x 256 x 2 B + ... : a res-block weight tensor, ~2.4 MB
  unsigned char* w; unsigned* sink; unsigned long long* stamps;
  hipMalloc(&w, wbytes); hipMalloc(&sink, blocks * 512 * 4); hipMalloc(&stamps, blocks * 16);
  hipMemset(w, 1, wbytes);
  hipFuncSetAttribute(reinterpret_cast<const void*>(feed<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(feed<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(feed<0>, dim3(blocks), dim3(512), 65536, 0, w, wbytes, iters, sink, stamps);
      else hipLaunchKernelGGL(feed<1>, dim3(blocks), dim3(512), 65536, 0, w, wbytes, iters, sink, stamps);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(blocks * 2);
      hipMemcpy(h.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
      double c = 0, q = 0;
      for (int b = 0; b < blocks; ++b) { c += (double)h[2 * b]; q += (double)h[2 * b + 1]; }
      const double cyc = c / blocks / iters, ns = q / blocks / iters * 10.0;      // s_memrealtime ticks at 100 MHz
      printf("%s rep %d: %.0f shader cycles = %.0f ns per 16 KB stage per CU -> %.1f B/clk, %.2f TB/s over 256 CUs\n",
             mode == 0 ? "lds-dma        " : "load + ds_write", rep, cyc, ns, 16384.0 / cyc, 16384.0 / ns * 256 / 1e3);
    }
  return 0;
}
