// Internal declarations shared by the .hip translation units of libdei2i_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "common.h"
#include "geom.h"

namespace dei2i {

enum ProfFamily : int { PROF_GATHER_GEMM = 0, PROF_WGRAD = 1, PROF_HALO_CONV = 2, PROF_HALO_FOLD = 3, PROF_FAMILIES = 4 };
// host-side launch counters, one per MFMA kernel family (dei2i_launch_counts: the tests assert WHICH kernel served a shape,
// so that a silent fall-through to the generic GEMM cannot pass a parity test meant for a tuned kernel)
enum KernelId : int { K_GATHER_V1 = 0, K_GATHER_V2, K_HALO_CONV, K_HALO_CONV_FP8, K_THIN_CIN, K_THIN_COUT, K_WGRAD_V1, K_WGRAD_V2,
                      K_WGRAD_HALO, K_WGRAD_THIN, K_HALO16_CONV, K_SPLITK_FINALIZE, K_HALO16_S2, K_COUNT };
void count_launch(int kid);
void prof_begin(int family, double flops, hipStream_t st);
void prof_end(int family, hipStream_t st);

struct DescPack {
  GatherDesc d[4];
  long long woff[4];      // element offset of each class's packed weights
  FastDiv fd_taps[4];     // divide by th*tw (v2 k-step order)
  int n;
  int ws_compact;         // split-K partials indexed by (m_base[class] + m) instead of the output pixel
  int ws_atomic;          // split-K slices accumulate into ONE slab with fp32 atomics (A/B option) instead of one slab each
  int skip_dead_taps;     // v1: skip the k-steps of a tap that contributes zero to every row of the tile (reflect ring)
  int m_base[4];
};

void set_num_cu(int n);
void set_use_v2(int on);
hipError_t gather_gemm_v2(const DescPack& pack, const void* src, const void* wgt, int wrows, const float* bias, void* out,
                          float* ws, size_t ws_bytes, int ldc, int act, int num_cu, hipStream_t st);
int num_cu();
void set_use_halo(int on);
void set_use_thin(int on);
void set_splitk_atomic(int on);
hipError_t thin_cout_conv(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out, int ldc,
                          int act, int num_cu, hipStream_t st);
hipError_t thin_cin_conv(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out, int ldc,
                         int act, int num_cu, hipStream_t st);
hipError_t halo_conv(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out, int ldc,
                     int act, int num_cu, hipStream_t st, const float* dequant = nullptr, const ConvPro* pro = nullptr,
                     float* stats = nullptr);

// 16 x 32 pixel tiles, 32-channel slices (conv_halo16.hip): the large-grid 3x3 stride-1 layers
hipError_t halo16_conv(const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias, void* out, int ldc,
                       int act, int num_cu, hipStream_t st, float* stats = nullptr, const void* ring = nullptr, bool fold = false,
                       const EpiNorm* en = nullptr);
bool halo16_s2_shape_ok(const GatherDesc& g, int ldc, int num_cu);     // the 4x4 stride-2 form of the 16 x 32 tile kernel takes this conv
hipError_t gather_gemm(int dtype, const GatherDesc& g, const void* src, const void* wgt, int wrows, const float* bias,
                       void* out, float* ws, size_t ws_bytes, int ldc, int act, hipStream_t st);
hipError_t gather_gemm_multi(int dtype, const GatherDesc* descs, const long long* woffs, int n, const void* src,
                             const void* wgt, int wrows, const float* bias, void* out, float* ws, size_t ws_bytes, int ldc,
                             int act, hipStream_t st, bool compact_ws = false);
// One wgrad slab: the packed [Cout][K] gradient with the row count rounded up to 8, so the kernels' 8-row store
// groups need no per-row guard (the rows beyond Cout are never read).
static inline long long wgrad_slab_elems(int co_rows, int K) { return (long long)((co_rows + 7) & ~7) * K; }
// nsplit_out != nullptr: dw holds up to capacity_elems floats; split z writes slab z, *nsplit_out slabs to be summed
// by wgrad_reduce_unpack (deterministic).  nsplit_out == nullptr: dw is one packed buffer, written by a single split.
hipError_t wgrad_gemm(int dtype, const GatherDesc& g, const void* src, const void* dy, int co_rows, int ldy, float* dw,
                      size_t capacity_elems, int* nsplit_out, hipStream_t st);

hipError_t wgrad_v2(const GatherDesc& g, const void* src, const void* dy, int co_rows, int ldy, float* slabs,
                    size_t slab_capacity_elems, int num_cu, int* nsplit_out, hipStream_t st);
hipError_t wgrad_halo(const GatherDesc& g, const void* src, const void* dy, int co_rows, int ldy, float* slabs,
                      size_t slab_capacity_elems, int num_cu, int* nsplit_out, bool force, hipStream_t st,
                      const ConvPro* pro = nullptr);
hipError_t wgrad_thin(const GatherDesc& g, const void* src, const void* dy, int co_rows, int ldy, float* slabs,
                      size_t slab_capacity_elems, int num_cu, int* nsplit_out, hipStream_t st);
hipError_t wgrad_reduce_unpack(float* slabs, int nsplit, long long slab_elems, float* dw, int Cout, int Cin, int CinS,
                               int taps, int accumulate, hipStream_t st);

inline unsigned grid_for(size_t work_items, int threads, unsigned cap = 256u * 8u) {
  size_t b = (work_items + threads - 1) / threads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

}  // namespace dei2i
