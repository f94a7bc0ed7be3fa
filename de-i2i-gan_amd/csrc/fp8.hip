// fp8 (OCP e4m3) forward path of the stride-1 3x3 convolutions (BASELINE.json configs[4]).
//
// Scope: the FORWARD GEMM of the halo-resident conv runs on v_mfma_f32_16x16x32_fp8_fp8 with e4m3 operands and fp32
// accumulation; activations stay bf16 in HBM and are quantised by one elementwise pass in front of the conv (per-tensor
// scale), weights are quantised when they are packed (per-tensor scale from their amax, computed on the device);
// dgrad / wgrad keep the bf16 kernels on the bf16 tensors (straight-through, the usual fp8-forward recipe).
// There is NO fallback in here: dei2i_conv2d_fwd_fp8 on a shape the fp8 kernel does not take returns an error, the caller
// asks dei2i_conv2d_fp8_supported first.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dei2i_hip.h"
#include "common.h"
#include "geom.h"
#include "launch.h"

namespace dei2i {

// 8 bf16 -> 8 e4m3 per thread (16 bytes in, 8 bytes out)
__global__ void quantize_fp8_kernel(const bf16_t* __restrict__ x, const float scale, unsigned char* __restrict__ out, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    store_e4m3_of_bf16x8(out + i * 8, *reinterpret_cast<const u32x4*>(x + i * 8), scale);
  }
}

// packed forward layout [Cout][kh*kw][CinS] in e4m3, element = w_oihw * s with s = 448 / amax[0]; 4 consecutive ci per
// thread.  Thread 0 also writes the conv's dequantisation factor 1 / (act_scale * s).
__global__ void pack_fwd_fp8_kernel(const float* __restrict__ w, const float* __restrict__ amax, unsigned char* __restrict__ dst,
                                    int Cout, int Cin, int CinS, int taps, float act_scale, float* __restrict__ dequant) {
  const size_t total4 = (size_t)Cout * taps * CinS / 4;
  const float s = 448.f / fmaxf(amax[0], 1e-30f);
  if (blockIdx.x == 0 && threadIdx.x == 0) dequant[0] = 1.f / (act_scale * s);
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (size_t)gridDim.x * blockDim.x) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long si = packed_fwd_src((long long)(q * 4 + e), Cin, CinS, taps);
      v[e] = si >= 0 ? clamp_e4m3(w[si] * s) : 0.f;
    }
    int p = 0;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], p, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], p, true);
    *reinterpret_cast<uint32_t*>(dst + q * 4) = (uint32_t)p;
  }
}

// the descriptor of the SAME convolution seen as a bf16 conv over byte pairs (half the input channels)
static bool fp8_desc(const dei2i_conv* c, GatherDesc& g) {
  if (!c || c->dtype != DT_BF16 || c->kh != 3 || c->kw != 3 || c->stride != 1 || c->pad != 1) return false;
  if (c->Cin <= 0 || c->CinS % 128 != 0 || c->CinS < c->Cin || c->Cout <= 0 || c->CoutS < c->Cout || c->CoutS % 8 != 0) return false;
  if (c->N <= 0 || c->H <= 0 || c->W <= 0 || c->up < 0 || c->up > 1) return false;
  ConvShape s;
  s.N = c->N; s.H = c->H; s.W = c->W; s.Cin = c->CinS / 2; s.Cout = c->Cout;
  s.kh = 3; s.kw = 3; s.stride = 1; s.pad = 1; s.pad_mode = c->pad_mode; s.up = c->up;
  g = make_fwd_desc(s, c->CinS / 2);
  return true;
}

}  // namespace dei2i

using namespace dei2i;

extern "C" {

int dei2i_quantize_fp8(size_t n, const void* x_bf16, float scale, void* out_e4m3, dei2i_stream s) {
  if (n == 0 || n % 8 != 0 || !x_bf16 || !out_e4m3 || !(scale > 0.f)) return DEI2I_ERR_BAD_ARG;
  const size_t nvec = n / 8;
  hipLaunchKernelGGL(quantize_fp8_kernel, dim3(grid_for(nvec, 256, 256u * 16u)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x_bf16,
                     scale, (unsigned char*)out_e4m3, nvec);
  return (int)hipGetLastError();
}

int dei2i_pack_weight_fwd_fp8(const dei2i_conv* c, const float* w_oihw, const float* amax, float act_scale, void* packed_e4m3,
                              float* dequant, dei2i_stream s) {
  GatherDesc g;
  if (!fp8_desc(c, g) || !w_oihw || !amax || !packed_e4m3 || !dequant || !(act_scale > 0.f)) return DEI2I_ERR_BAD_ARG;
  const size_t total4 = (size_t)c->Cout * 9 * c->CinS / 4;
  hipLaunchKernelGGL(pack_fwd_fp8_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, (hipStream_t)s, w_oihw, amax,
                     (unsigned char*)packed_e4m3, c->Cout, c->Cin, c->CinS, 9, act_scale, dequant);
  return (int)hipGetLastError();
}

int dei2i_conv2d_fp8_supported(const dei2i_conv* c) {
  GatherDesc g;
  if (!fp8_desc(c, g)) return 0;
  if (g.Ho % 8 != 0 || g.Wo % 32 != 0 || c->CoutS < 64) return 0;
  if ((long long)g.N * g.Hs * g.Ws * g.Cs >= (1ll << 31)) return 0;
  const int tiles_m = g.N * (g.Ho / 8) * (g.Wo / 32);
  const int tiles = c->CoutS >= 128 ? tiles_m * ((c->CoutS + 127) / 128) : tiles_m;
  return tiles >= num_cu() / 2 ? 1 : 0;
}

int dei2i_conv2d_fwd_fp8(const dei2i_conv* c, const void* x_e4m3, const void* w_e4m3, const float* bias, const float* dequant,
                         int act, void* y_bf16, dei2i_stream s) {
  GatherDesc g;
  if (!fp8_desc(c, g) || !x_e4m3 || !w_e4m3 || !dequant || !y_bf16) return DEI2I_ERR_BAD_ARG;
  const hipError_t e = halo_conv(g, x_e4m3, w_e4m3, c->Cout, bias, y_bf16, c->CoutS, act, num_cu(), (hipStream_t)s, dequant);
  if (e == hipErrorNotSupported) return DEI2I_ERR_BAD_ARG;      // no fallback: the caller checks dei2i_conv2d_fp8_supported
  return (int)e;
}

}  // extern "C"
