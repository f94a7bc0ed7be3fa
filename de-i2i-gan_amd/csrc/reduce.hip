// Reduction kernels: per-channel moments (BatchNorm / InstanceNorm statistics), the two-pass backward of
// BatchNorm and SPADE(InstanceNorm), bias-gradient column sums, and the scalar losses (BCE-with-logits, L1).
//
// Common shape: an NHWC tensor is viewed as (rows, C); a 256-thread workgroup owns a contiguous row range,
// thread t owns the 16-byte channel vector (t % cv) and walks rows (t / cv), (t / cv) + 256/cv, ...  Per-thread
// fp32 partials are combined across the row-threads through LDS; the (small) cross-workgroup combine runs in
// fp64 in a finalize kernel, so the statistics do not depend on atomics ordering.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/dei2i_hip.h"
#include "launch.h"

namespace dei2i {

static inline int vec_of(int dtype) { return dtype == DT_BF16 ? 8 : 4; }

// Combine per-thread partials vals[NV][VEC] of the threads sharing a channel vector and store the sums:
//   dst[q * C + vcol * VEC + e] = sum over prow (in order) of vals[q][e]      (dst: this workgroup's record)
// The sum runs as a ROLLED loop over (quantity, element, channel-vector) outputs on all 256 threads -- a few hundred
// bytes of code instead of NV*VEC unrolled serial LDS sums on cv threads (a dispatch walks its code cold: code size
// is launch latency for these short kernels).
template <int NV, int VEC, typename OT = float>
DEI2I_D void block_combine_store(const float (&vals)[NV][VEC], int cv, int rpp, float* smem, OT* __restrict__ dst, int C) {
  const int tid = threadIdx.x;
  const int vcol = tid % cv, prow = tid / cv;
  __syncthreads();
  if (prow < rpp) {
#pragma unroll
    for (int q = 0; q < NV; ++q)
#pragma unroll
      for (int e = 0; e < VEC; ++e) smem[((q * VEC + e) * rpp + prow) * cv + vcol] = vals[q][e];
  }
  __syncthreads();
#pragma unroll 1
  for (int o = tid; o < NV * VEC * cv; o += 256) {
    const int qe = o / cv, vc = o - qe * cv;
    const float* src = smem + (size_t)qe * rpp * cv + vc;
    float sum = 0.f;
    for (int r = 0; r < rpp; ++r) sum += src[r * cv];
    const int q = qe / VEC, e = qe - q * VEC;
    Elem<OT>::store(dst + (size_t)q * C + vc * VEC + e, sum);
  }
}

// ---- moments: partial[(n*chunks + chunk)*2*C + {0,C} + c] = sum x, sum x^2 over the chunk's rows ----
template <typename T>
__global__ __launch_bounds__(256) void moments_partial_kernel(const T* __restrict__ x, float* __restrict__ partial, int HW,
                                                              int C, int chunks) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ float smem[];
  const int cv = C / VEC, rpp = 256 / cv;
  const int tid = threadIdx.x, vcol = tid % cv, prow = tid / cv;
  const int chunk = blockIdx.x, n = blockIdx.y;
  const int rows_per_chunk = (HW + chunks - 1) / chunks;
  const int rbeg = chunk * rows_per_chunk, rend = min(HW, rbeg + rows_per_chunk);
  float v[2][VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[0][e] = v[1][e] = 0.f;
  if (prow < rpp) {
    const T* __restrict__ base = x + (size_t)n * HW * C + (size_t)vcol * VEC;
    auto acc = [&](const u32x4& q) {
      float f[VEC];
      Elem<T>::unpack(q, f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) { v[0][e] += f[e]; v[1][e] = fmaf(f[e], f[e], v[1][e]); }
    };
    int r = rbeg + prow;
    for (; r + 3 * rpp < rend; r += 4 * rpp) {       // four 16-byte loads in flight per thread
      const u32x4 q0 = *reinterpret_cast<const u32x4*>(base + (size_t)r * C);
      const u32x4 q1 = *reinterpret_cast<const u32x4*>(base + (size_t)(r + rpp) * C);
      const u32x4 q2 = *reinterpret_cast<const u32x4*>(base + (size_t)(r + 2 * rpp) * C);
      const u32x4 q3 = *reinterpret_cast<const u32x4*>(base + (size_t)(r + 3 * rpp) * C);
      acc(q0); acc(q1); acc(q2); acc(q3);
    }
    for (; r < rend; r += rpp) acc(*reinterpret_cast<const u32x4*>(base + (size_t)r * C));
  }
  block_combine_store<2, VEC>(v, cv, rpp, smem, partial + ((size_t)n * chunks + chunk) * 2 * C, C);
}

// Cross-workgroup combine of the partial records, in fp64 (deterministic, no atomics).  One workgroup per channel
// (64 or 256 threads, by record count): the threads stride over the records -- C * blockDim independent 4-byte loads
// keep the (L2-resident, <= 2 MB) partial buffer's latency covered even for narrow layers -- then meet through
// wave shuffles and LDS.  record r, quantity q, channel c lives at partial[(r * NQ + q) * C + c].
// Result valid in thread 0.
DEI2I_D double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NQ>
DEI2I_D void combine_records(const float* __restrict__ partial, size_t rec0, int nrec, int C, int c, double (&s)[NQ]) {
  __shared__ double red[NQ][4];
#pragma unroll
  for (int q = 0; q < NQ; ++q) s[q] = 0.0;
  for (int r = threadIdx.x; r < nrec; r += blockDim.x)
#pragma unroll
    for (int q = 0; q < NQ; ++q) s[q] += (double)partial[((rec0 + r) * NQ + q) * C + c];
#pragma unroll
  for (int q = 0; q < NQ; ++q) s[q] = wave_sum_f64(s[q]);
  if (blockDim.x > 64) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0)
#pragma unroll
      for (int q = 0; q < NQ; ++q) red[q][wave] = s[q];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
      for (int q = 0; q < NQ; ++q) s[q] = red[q][0] + red[q][1] + red[q][2] + red[q][3];
  }
}
static inline int combine_threads(int nrec) { return nrec > 256 ? 256 : 64; }

__global__ __launch_bounds__(256) void bn_finalize_train_kernel(const float* __restrict__ partial, int N, int chunks, int C, double count,
                                         const float* __restrict__ weight, const float* __restrict__ bias,
                                         float* __restrict__ rmean, float* __restrict__ rvar, float momentum, float eps,
                                         float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ a,
                                         float* __restrict__ b, long long* __restrict__ num_batches_tracked,
                                         int running_stride) {
  // blockIdx.y: a GROUP of the batch with statistics of its own (ops.bn_batch_groups: several passes in one batch) -- N images and
  // N * chunks records per group, one row of C per group in mean / rstd / a / b, the running buffers `running_stride` floats apart
  const int c = blockIdx.x, grp = blockIdx.y;
  partial += (size_t)grp * N * chunks * 2 * C;
  mean += (size_t)grp * C; rstd += (size_t)grp * C; a += (size_t)grp * C; b += (size_t)grp * C;
  if (rmean != nullptr) { rmean += (size_t)grp * running_stride; rvar += (size_t)grp * running_stride; }
  if (c == 0 && grp == 0 && threadIdx.x == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;   // nn.BatchNorm2d's counter
  double sq[2];
  combine_records<2>(partial, 0, N * chunks, C, c, sq);
  if (threadIdx.x != 0) return;
  const double mu = sq[0] / count;
  double var = sq[1] / count - mu * mu;
  if (var < 0.0) var = 0.0;
  const float rs = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)mu;
  rstd[c] = rs;
  const float av = weight[c] * rs;
  a[c] = av;
  b[c] = bias[c] - (float)mu * av;
  if (rmean != nullptr) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
  }
}

__global__ void bn_finalize_eval_kernel(int C, const float* __restrict__ weight, const float* __restrict__ bias,
                                        const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                        float* __restrict__ a, float* __restrict__ b) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float av = weight[c] / sqrtf(rvar[c] + eps);
  a[c] = av;
  b[c] = bias[c] - rmean[c] * av;
}

__global__ __launch_bounds__(256) void in_finalize_kernel(const float* __restrict__ partial, int N, int chunks, int C, double count,
                                                          float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x, n = blockIdx.y;
  double sq[2];
  combine_records<2>(partial, (size_t)n * chunks, chunks, C, c, sq);
  if (threadIdx.x != 0) return;
  const double mu = sq[0] / count;
  double var = sq[1] / count - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[(size_t)n * C + c] = (float)mu;
  rstd[(size_t)n * C + c] = (float)(1.0 / sqrt(var + (double)eps));
}

// ---- BatchNorm backward ----
// partial[(chunk*2 + {0,1})*C + c] = sum g, sum g*xhat ; g = dz*act'(a*y+b), xhat = (y-mean)*rstd
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                             const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             int act, float* __restrict__ partial, size_t pixels, int C,
                                                             int chunks) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ float smem[];
  const int cv = C / VEC, rpp = 256 / cv;
  const int tid = threadIdx.x, vcol = tid % cv, prow = tid / cv;
  const int chunk = blockIdx.x, grp = blockIdx.y;       // blockIdx.y: a group of the batch (`pixels` pixels, own coefficients, own records)
  dz += (size_t)grp * pixels * C; y += (size_t)grp * pixels * C;
  a += (size_t)grp * C; b += (size_t)grp * C; mean += (size_t)grp * C; rstd += (size_t)grp * C;
  partial += (size_t)grp * chunks * 2 * C;
  const size_t rows_per_chunk = (pixels + chunks - 1) / chunks;
  const size_t rbeg = (size_t)chunk * rows_per_chunk;
  const size_t rend = rbeg + rows_per_chunk < pixels ? rbeg + rows_per_chunk : pixels;
  float v[2][VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[0][e] = v[1][e] = 0.f;
  if (prow < rpp) {
    float av[VEC], bv[VEC], mv[VEC], rv[VEC];
    const int c = vcol * VEC;
    ldcoef<VEC>(a + c, av); ldcoef<VEC>(b + c, bv); ldcoef<VEC>(mean + c, mv); ldcoef<VEC>(rstd + c, rv);
    auto acc = [&](const u32x4& dq, const u32x4& yq) {
      float d[VEC], yy[VEC];
      Elem<T>::unpack(dq, d);
      Elem<T>::unpack(yq, yy);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float z = fmaf(av[e], yy[e], bv[e]);
        const float g = d[e] * act_grad_from_out(z, act);
        v[0][e] += g;
        v[1][e] = fmaf(g, (yy[e] - mv[e]) * rv[e], v[1][e]);
      }
    };
    const T* __restrict__ dzb = dz + c;
    const T* __restrict__ yb = y + c;
    size_t r = rbeg + prow;
    for (; r + rpp < rend; r += 2 * (size_t)rpp) {      // four 16-byte loads in flight per thread
      const u32x4 d0 = *reinterpret_cast<const u32x4*>(dzb + r * C);
      const u32x4 y0 = *reinterpret_cast<const u32x4*>(yb + r * C);
      const u32x4 d1 = *reinterpret_cast<const u32x4*>(dzb + (r + rpp) * C);
      const u32x4 y1 = *reinterpret_cast<const u32x4*>(yb + (r + rpp) * C);
      acc(d0, y0); acc(d1, y1);
    }
    if (r < rend) acc(*reinterpret_cast<const u32x4*>(dzb + r * C), *reinterpret_cast<const u32x4*>(yb + r * C));
  }
  block_combine_store<2, VEC>(v, cv, rpp, smem, partial + (size_t)chunk * 2 * C, C);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int chunks, int C,
                                                              float* __restrict__ dweight, float* __restrict__ dbias,
                                                              float* __restrict__ acc_dweight, float* __restrict__ acc_dbias) {
  const int c = blockIdx.x;
  double sq[2];
  combine_records<2>(partial, 0, chunks, C, c, sq);
  if (threadIdx.x != 0) return;
  dbias[c] = (float)sq[0];
  dweight[c] = (float)sq[1];
  if (acc_dweight != nullptr) {      // a further use of the same parameters in this backward pass: add into its gradient
    acc_dbias[c] += (float)sq[0];
    acc_dweight[c] += (float)sq[1];
  }
}

// The same for `groups` groups of the batch: per group its own sums (gsum[g][0][c] = dweight, gsum[g][1][c] = dbias: what the apply kernel
// of the group needs), the parameters' gradient = their total over the groups (written, or added when `accumulate`).
__global__ __launch_bounds__(256) void bn_bwd_finalize_groups_kernel(const float* __restrict__ partial, int chunks, int C, int groups,
                                                                     float* __restrict__ gsum, float* __restrict__ dweight,
                                                                     float* __restrict__ dbias, int accumulate) {
  const int c = blockIdx.x;
  double tw = 0.0, tb = 0.0;
  for (int g = 0; g < groups; ++g) {
    double sq[2];
    __syncthreads();                                   // (combine_records' scratch is reused)
    combine_records<2>(partial + (size_t)g * chunks * 2 * C, 0, chunks, C, c, sq);
    if (threadIdx.x == 0) {
      gsum[((size_t)g * 2 + 0) * C + c] = (float)sq[1];
      gsum[((size_t)g * 2 + 1) * C + c] = (float)sq[0];
      tw += (double)(float)sq[1]; tb += (double)(float)sq[0];
    }
  }
  if (threadIdx.x != 0) return;
  if (accumulate) { dweight[c] += (float)tw; dbias[c] += (float)tb; }
  else { dweight[c] = (float)tw; dbias[c] = (float)tb; }
}

// dy = a*(g - sum_g/M - xhat*sum_gx/M) with g = dz*act'(a*y+b), xhat = (y-mean)*rstd, folded per channel into
//   dy = a*g + c2*(y - mean) + c3,   c2 = -a*rstd*sum_gx/M,  c3 = -a*sum_g/M        (eval mode: c2 = c3 = 0)
template <typename T, bool INVARIANT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                           const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           int act, int train, const float* __restrict__ dweight,
                                                           const float* __restrict__ dbias, float inv_count,
                                                           T* __restrict__ dy, size_t nvec, int cv, int sum_stride) {
  constexpr int VEC = Elem<T>::VEC;
  {                                                    // blockIdx.y: a group of the batch (nvec vectors, own coefficients and sums)
    const size_t grp = blockIdx.y;
    dz += grp * nvec * VEC; y += grp * nvec * VEC; dy += grp * nvec * VEC;
    const size_t co = grp * (size_t)cv * VEC;
    a += co; b += co; mean += co; rstd += co;
    dweight += grp * sum_stride; dbias += grp * sum_stride;
  }
  float av[VEC], bv[VEC], mv[VEC], c2[VEC], c3[VEC];
  auto load_coef = [&](int c) {
    float rv[VEC], dw[VEC], db[VEC];
    ldcoef<VEC>(a + c, av); ldcoef<VEC>(b + c, bv); ldcoef<VEC>(mean + c, mv);
    if (train) {
      ldcoef<VEC>(rstd + c, rv); ldcoef<VEC>(dweight + c, dw); ldcoef<VEC>(dbias + c, db);
#pragma unroll
      for (int e = 0; e < VEC; ++e) { c2[e] = -av[e] * rv[e] * (dw[e] * inv_count); c3[e] = -av[e] * (db[e] * inv_count); }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) c2[e] = c3[e] = 0.f;
    }
  };
  auto apply = [&](const u32x4& dq, const u32x4& yq) {
    float d[VEC], yy[VEC];
    Elem<T>::unpack(dq, d);
    Elem<T>::unpack(yq, yy);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float z = fmaf(av[e], yy[e], bv[e]);
      const float g = d[e] * act_grad_from_out(z, act);
      d[e] = fmaf(av[e], g, fmaf(c2[e], yy[e] - mv[e], c3[e]));
    }
    return Elem<T>::pack(d);
  };
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (INVARIANT) {                       // stride % cv == 0: the thread keeps its channel vector
    load_coef((int)(i % cv) * VEC);
    for (; i + stride < nvec; i += 2 * stride) {       // four 16-byte loads in flight per thread
      const u32x4 d0 = *reinterpret_cast<const u32x4*>(dz + i * VEC);
      const u32x4 y0 = *reinterpret_cast<const u32x4*>(y + i * VEC);
      const u32x4 d1 = *reinterpret_cast<const u32x4*>(dz + (i + stride) * VEC);
      const u32x4 y1 = *reinterpret_cast<const u32x4*>(y + (i + stride) * VEC);
      *reinterpret_cast<u32x4*>(dy + i * VEC) = apply(d0, y0);
      *reinterpret_cast<u32x4*>(dy + (i + stride) * VEC) = apply(d1, y1);
    }
    if (i < nvec)
      *reinterpret_cast<u32x4*>(dy + i * VEC) = apply(*reinterpret_cast<const u32x4*>(dz + i * VEC),
                                                      *reinterpret_cast<const u32x4*>(y + i * VEC));
  } else {
    for (; i < nvec; i += stride) {
      load_coef((int)(i % cv) * VEC);
      *reinterpret_cast<u32x4*>(dy + i * VEC) = apply(*reinterpret_cast<const u32x4*>(dz + i * VEC),
                                                      *reinterpret_cast<const u32x4*>(y + i * VEC));
    }
  }
}


// ---- SPADE backward ----
// z = relu(v), v = xhat*(1+gamma) + beta, xhat = (x - mean)*rstd.  With g = dz*[v > 0]:
//   dgamma = g*xhat, dbeta = g, dxhat = g*(1+gamma), dx = rstd*(sum_cell dxhat - cnt*mean(dxhat) - cnt*xhat*mean(dxhat*xhat)).
// Both passes RECOMPUTE v from x and the gamma/beta table instead of reading the saved output z and a stored dxhat
// tensor: pass 1 reads dz + x (2 tensors, no activation-sized write), pass 2 reads dz + x and writes dx -- 5 tensor
// transfers instead of 7, and the op no longer keeps its output alive for backward.
// partial[((n*chunks + chunk)*4 + q)*C + c], q: 0 sum dxhat, 1 sum dxhat*xhat, 2 sum dgamma (interior class), 3 sum dbeta (interior)
template <typename T>
__global__ __launch_bounds__(256) void spade_bwd_partial_kernel(const T* __restrict__ dz, const T* __restrict__ x,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const T* __restrict__ gb, int gb_mode,
                                                                T* __restrict__ dgb_dense, float* __restrict__ partial, int H,
                                                                int W, int C, int up, int chunks, float slope = 0.f) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ float smem[];
  const int cv = C / VEC, rpp = 256 / cv;
  const int tid = threadIdx.x, vcol = tid % cv, prow = tid / cv;
  const int chunk = blockIdx.x, n = blockIdx.y;
  const int HW = H * W, Hs = H >> up, Ws = W >> up;
  const int rows_per_chunk = (HW + chunks - 1) / chunks;
  const int rbeg = chunk * rows_per_chunk, rend = min(HW, rbeg + rows_per_chunk);
  float v[4][VEC];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[q][e] = 0.f;
  if (prow < rpp) {
    const int c = vcol * VEC;
    float mv[VEC], rv[VEC];
    ldcoef<VEC>(mean + (size_t)n * C + c, mv);
    ldcoef<VEC>(rstd + (size_t)n * C + c, rv);
    struct Row { u32x4 d, x, gm, bt; size_t opix; bool interior; };
    auto load = [&](int r) {
      Row q;
      const int h = r / W, w = r - h * W;
      q.opix = (size_t)n * HW + r;
      q.d = *reinterpret_cast<const u32x4*>(dz + q.opix * C + c);
      q.x = *reinterpret_cast<const u32x4*>(x + (((size_t)n * Hs + (h >> up)) * Ws + (w >> up)) * C + c);
      size_t gpix = q.opix;
      q.interior = true;
      if (gb_mode != 0) {
        const int cy = border_class(h, H), cx = border_class(w, W);
        gpix = ((size_t)n * 5 + cy) * 5 + cx;
        q.interior = cy == 2 && cx == 2;
      }
      q.gm = *reinterpret_cast<const u32x4*>(gb + gpix * 2 * C + c);
      q.bt = *reinterpret_cast<const u32x4*>(gb + gpix * 2 * C + C + c);
      return q;
    };
    auto use = [&](const Row& q) {
      float d[VEC], xv[VEC], gm[VEC], bt[VEC], dg[VEC], db[VEC];
      Elem<T>::unpack(q.d, d); Elem<T>::unpack(q.x, xv); Elem<T>::unpack(q.gm, gm); Elem<T>::unpack(q.bt, bt);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float xh = (xv[e] - mv[e]) * rv[e];
        const float g = fmaf(xh, 1.f + gm[e], bt[e]) > 0.f ? d[e] : slope * d[e];      // (slope: 0 = ReLU -- SPADE; dei2i_in_act_bwd: any)
        const float dxh = g * (1.f + gm[e]);
        dg[e] = g * xh;
        db[e] = g;
        v[0][e] += dxh;
        v[1][e] = fmaf(dxh, xh, v[1][e]);
      }
      if (gb_mode == 0) {
        *reinterpret_cast<u32x4*>(dgb_dense + q.opix * 2 * C + c) = Elem<T>::pack(dg);
        *reinterpret_cast<u32x4*>(dgb_dense + q.opix * 2 * C + C + c) = Elem<T>::pack(db);
      } else if (q.interior) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { v[2][e] += dg[e]; v[3][e] += db[e]; }
      }             // border classes are reduced by spade_bwd_border_kernel (no atomics)
    };
    int r = rbeg + prow;
    for (; r + rpp < rend; r += 2 * rpp) {           // two rows (eight 16-byte loads) in flight per thread
      const Row q0 = load(r), q1 = load(r + rpp);
      use(q0); use(q1);
    }
    if (r < rend) use(load(r));
  }
  block_combine_store<4, VEC>(v, cv, rpp, smem, partial + ((size_t)n * chunks + chunk) * 4 * C, C);
}

// class-mode gamma/beta gradients of the 24 border classes: one workgroup per (class, image) walks that class's
// O(perimeter) pixel list and reduces dgamma = g*xhat, dbeta = g with plain stores (deterministic, no atomics).
template <typename T>
__global__ __launch_bounds__(256) void spade_bwd_border_kernel(const T* __restrict__ dz, const T* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const T* __restrict__ gb, T* __restrict__ dgb_cls, int H,
                                                               int W, int C, int up, float slope = 0.f) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ float smem[];
  const int cls = blockIdx.x, n = blockIdx.y;
  const int cy = cls / 5, cx = cls - cy * 5;
  if (cy == 2 && cx == 2) return;                       // interior class: handled by the streaming pass
  const int cv = C / VEC, rpp = 256 / cv;
  const int tid = threadIdx.x, vcol = tid % cv, prow = tid / cv;
  const int Hs = H >> up, Ws = W >> up;
  // rows / cols belonging to the class
  const int y0 = cy < 2 ? cy : (cy == 2 ? 2 : H - 5 + cy), ny = cy == 2 ? H - 4 : 1;
  const int x0 = cx < 2 ? cx : (cx == 2 ? 2 : W - 5 + cx), nx = cx == 2 ? W - 4 : 1;
  const int count = ny * nx;
  float v[2][VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[0][e] = v[1][e] = 0.f;
  if (prow < rpp) {
    const int c = vcol * VEC;
    float mv[VEC], rv[VEC], gm[VEC], bt[VEC];
    ldcoef<VEC>(mean + (size_t)n * C + c, mv);
    ldcoef<VEC>(rstd + (size_t)n * C + c, rv);
    const size_t gpix = ((size_t)n * 5 + cy) * 5 + cx;
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(gb + gpix * 2 * C + c), gm);
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(gb + gpix * 2 * C + C + c), bt);
    for (int i = prow; i < count; i += rpp) {
      const int h = y0 + i / nx, w = x0 + i % nx;
      const size_t opix = ((size_t)n * H + h) * W + w;
      float d[VEC], xv[VEC];
      Elem<T>::unpack(*reinterpret_cast<const u32x4*>(dz + opix * C + c), d);
      Elem<T>::unpack(*reinterpret_cast<const u32x4*>(x + (((size_t)n * Hs + (h >> up)) * Ws + (w >> up)) * C + c), xv);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float xh = (xv[e] - mv[e]) * rv[e];
        const float g = fmaf(xh, 1.f + gm[e], bt[e]) > 0.f ? d[e] : slope * d[e];
        v[0][e] = fmaf(g, xh, v[0][e]);
        v[1][e] += g;
      }
    }
  }
  block_combine_store<2, VEC, T>(v, cv, rpp, smem, dgb_cls + (((size_t)n * 5 + cy) * 5 + cx) * 2 * C, C);
}

// coef[(n*2 + {0,1})*C + c] = s1/M, s2/M ; interior-class gamma/beta sums added into dgb_cls[n,2,2,:]
__global__ __launch_bounds__(256) void spade_bwd_finalize_kernel(const float* __restrict__ partial, int N, int chunks, int C,
                                                                 double count, float* __restrict__ coef,
                                                                 void* __restrict__ dgb_cls, int dtype) {
  const int c = blockIdx.x, n = blockIdx.y;
  double sq[4];
  combine_records<4>(partial, (size_t)n * chunks, chunks, C, c, sq);
  if (threadIdx.x != 0) return;
  coef[((size_t)n * 2 + 0) * C + c] = (float)(sq[0] / count);
  coef[((size_t)n * 2 + 1) * C + c] = (float)(sq[1] / count);
  if (dgb_cls != nullptr) {
    const size_t gpix = ((size_t)n * 5 + 2) * 5 + 2;
    if (dtype == DT_BF16) {          // the border kernel wrote the other 24 classes; this is the only writer of (2,2)
      bf16_t* o = (bf16_t*)dgb_cls;
      o[gpix * 2 * C + c] = f32_to_bf16((float)sq[2]);
      o[gpix * 2 * C + C + c] = f32_to_bf16((float)sq[3]);
    } else {
      float* o = (float*)dgb_cls;
      o[gpix * 2 * C + c] = (float)sq[2];
      o[gpix * 2 * C + C + c] = (float)sq[3];
    }
  }
}

// dx (source resolution) = rstd * ( sum_cell dxhat - cnt*c1 - cnt*xhat*c2 ) (+ addend); dxhat recomputed from dz, x, gamma/beta
template <typename T>
__global__ void spade_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ x, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, const T* __restrict__ gb, int gb_mode,
                                       const float* __restrict__ coef, const T* __restrict__ addend, T* __restrict__ dx, int N,
                                       int H, int W, int C, int up, float slope = 0.f) {
  constexpr int VEC = Elem<T>::VEC;
  const int cv = C / VEC;
  const int Hs = H >> up, Ws = W >> up;
  const size_t total = (size_t)N * Hs * Ws * cv;
  const float cnt = (float)(1 << (2 * up));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cv) * VEC;
    size_t r = i / cv;
    const int ws = (int)(r % Ws); r /= Ws;
    const int hs = (int)(r % Hs);
    const int n = (int)(r / Hs);
    float acc[VEC], xv[VEC], xh[VEC], mv[VEC], rv[VEC], k1[VEC], k2[VEC];
    ldcoef<VEC>(mean + (size_t)n * C + c, mv);
    ldcoef<VEC>(rstd + (size_t)n * C + c, rv);
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(x + i * VEC), xv);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { acc[e] = 0.f; xh[e] = (xv[e] - mv[e]) * rv[e]; }
    for (int a = 0; a < (1 << up); ++a)
      for (int b = 0; b < (1 << up); ++b) {
        const int h = (hs << up) + a, w = (ws << up) + b;
        const size_t opix = ((size_t)n * H + h) * W + w;
        const size_t gpix = gb_mode == 0 ? opix : ((size_t)n * 5 + border_class(h, H)) * 5 + border_class(w, W);
        float d[VEC], gm[VEC], bt[VEC];
        Elem<T>::unpack(*reinterpret_cast<const u32x4*>(dz + opix * C + c), d);
        Elem<T>::unpack(*reinterpret_cast<const u32x4*>(gb + gpix * 2 * C + c), gm);
        Elem<T>::unpack(*reinterpret_cast<const u32x4*>(gb + gpix * 2 * C + C + c), bt);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float g = fmaf(xh[e], 1.f + gm[e], bt[e]) > 0.f ? d[e] : slope * d[e];
          acc[e] = fmaf(g, 1.f + gm[e], acc[e]);
        }
      }
    float ad[VEC];
    if (addend != nullptr) Elem<T>::unpack(*reinterpret_cast<const u32x4*>(addend + i * VEC), ad);
    ldcoef<VEC>(coef + ((size_t)n * 2) * C + c, k1);
    ldcoef<VEC>(coef + ((size_t)n * 2 + 1) * C + c, k2);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float v = rv[e] * (acc[e] - cnt * k1[e] - cnt * xh[e] * k2[e]);
      if (addend != nullptr) v += ad[e];
      acc[e] = v;
    }
    *reinterpret_cast<u32x4*>(dx + i * VEC) = Elem<T>::pack(acc);
  }
}

// ---- column sums (bias gradient) ----
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ g, float* __restrict__ partial, size_t rows, int C) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ float smem[];
  const int cv = C / VEC, rpp = 256 / cv;
  const int tid = threadIdx.x, vcol = tid % cv, prow = tid / cv;
  const size_t rows_per_block = (rows + gridDim.x - 1) / gridDim.x;
  const size_t rbeg = (size_t)blockIdx.x * rows_per_block;
  const size_t rend = rbeg + rows_per_block < rows ? rbeg + rows_per_block : rows;
  float v[1][VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[0][e] = 0.f;
  if (prow < rpp) {
    const T* __restrict__ base = g + (size_t)vcol * VEC;
    auto acc = [&](const u32x4& q) {
      float f[VEC];
      Elem<T>::unpack(q, f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[0][e] += f[e];
    };
    size_t r = rbeg + prow;
    for (; r + 3 * (size_t)rpp < rend; r += 4 * (size_t)rpp) {
      const u32x4 q0 = *reinterpret_cast<const u32x4*>(base + r * C);
      const u32x4 q1 = *reinterpret_cast<const u32x4*>(base + (r + rpp) * C);
      const u32x4 q2 = *reinterpret_cast<const u32x4*>(base + (r + 2 * (size_t)rpp) * C);
      const u32x4 q3 = *reinterpret_cast<const u32x4*>(base + (r + 3 * (size_t)rpp) * C);
      acc(q0); acc(q1); acc(q2); acc(q3);
    }
    for (; r < rend; r += rpp) acc(*reinterpret_cast<const u32x4*>(base + r * C));
  }
  block_combine_store<1, VEC>(v, cv, rpp, smem, partial + (size_t)blockIdx.x * C, C);
}

// out[c] = sum over blocks of partial[block][c] in a fixed order (deterministic; replaces a memset + fp32 atomics).
// One workgroup per 16 channels: thread (slice, c) sums blocks slice, slice+16, ...; the 16 slices are then added in order.
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partial, int blocks, int C,
                                                              float* __restrict__ out) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float s = 0.f;
  if (c < C)
    for (int b = slice; b < blocks; b += 16) s += partial[(size_t)b * C + c];
  red[slice][cl] = s;
  __syncthreads();
  if (slice == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][cl];
    out[c] = t;
  }
}

// ---- scalar losses ----
DEI2I_D float block_sum_256(float v, float* smem) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  return smem[0] + smem[1] + smem[2] + smem[3];
}

__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ t, float tconst,
                                                      size_t n, float inv_n, float* __restrict__ out) {
  __shared__ float smem[4];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float xv = x[i], tv = t != nullptr ? t[i] : tconst;
    acc += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
  }
  const float s = block_sum_256(acc, smem);
  if (threadIdx.x == 0) out[blockIdx.x] = s;          // partial; loss_finalize_kernel sums them in order
}

// out[0] = inv_n * sum of the block partials, in index order (deterministic: no atomics, and `out` needs no zero fill)
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ part, int nparts, float inv_n,
                                                            float* __restrict__ out, int accumulate) {
  __shared__ float smem[4];
  float p = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) p += part[i];
  const float s = block_sum_256(p, smem);
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s * inv_n : s * inv_n;
}

// ---- noise injection (architecture.py:374-389 of the reference): out[r][c] = x[r][c] + w * noise[r], one N(0,1) value per
//      pixel shared by the channels, one scalar weight; dx = dy passes through, dw = sum_r noise[r] * sum_c dy[r][c] ----
template <typename T>
__global__ void noise_fwd_kernel(const T* __restrict__ x, const float* __restrict__ noise, const float* __restrict__ w,
                                 T* __restrict__ out, size_t nvec, int vec_per_row) {
  constexpr int VEC = Elem<T>::VEC;
  const float wt = w[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    float v[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(x + i * VEC), v);
    const float add = wt * noise[i / (size_t)vec_per_row];
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] += add;
    *reinterpret_cast<u32x4*>(out + i * VEC) = Elem<T>::pack(v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void noise_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ noise, size_t nvec,
                                                        int vec_per_row, float* __restrict__ part) {
  constexpr int VEC = Elem<T>::VEC;
  __shared__ float smem[4];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    float v[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(dy + i * VEC), v);
    float t = 0.f;
#pragma unroll
    for (int e = 0; e < VEC; ++e) t += v[e];
    acc = fmaf(noise[i / (size_t)vec_per_row], t, acc);
  }
  const float s = block_sum_256(acc, smem);
  if (threadIdx.x == 0) part[blockIdx.x] = s;          // loss_finalize_kernel sums the partials in index order
}

__global__ void bce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ t, float tconst, size_t n, float inv_n,
                               const float* __restrict__ gout, float* __restrict__ dx) {
  const float go = gout[0] * inv_n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float xv = x[i], tv = t != nullptr ? t[i] : tconst;
    dx[i] = (1.f / (1.f + expf(-xv)) - tv) * go;
  }
}

__global__ __launch_bounds__(256) void l1_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                     float inv_n, float* __restrict__ out) {
  __shared__ float smem[4];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += fabsf(a[i] - (b != nullptr ? b[i] : 0.f));
  const float s = block_sum_256(acc, smem);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}

__global__ void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n, float inv_n,
                              const float* __restrict__ gout, float* __restrict__ da, float* __restrict__ db) {
  const float go = gout[0] * inv_n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float d = a[i] - (b != nullptr ? b[i] : 0.f);
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (da != nullptr) da[i] = sgn * go;
    if (db != nullptr) db[i] = -sgn * go;
  }
}

// ---- fused conv + norm + act support (BASELINE.json configs[1]) ------------------------------
// out = act(a[c]*x + b[c]) (+ res) like affine_act_kernel (elementwise.hip), AND the per-channel sum / sum of squares of
// the stored (rounded) output in the moments_partial record layout: the InstanceNorm / BatchNorm that follows needs no
// statistics pass of its own.  grid (chunks, N), the row split of moments_partial_kernel.
template <typename T>
__global__ __launch_bounds__(256) void affine_act_stats_kernel(const T* __restrict__ x, const float* __restrict__ a,
                                                               const float* __restrict__ b, const T* __restrict__ res,
                                                               T* __restrict__ out, float* __restrict__ partial, int HW, int C,
                                                               int chunks, int act, int group_images) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ float smem[];
  const int cv = C / VEC, rpp = 256 / cv;
  const int tid = threadIdx.x, vcol = tid % cv, prow = tid / cv;
  const int chunk = blockIdx.x, n = blockIdx.y;
  const int rows_per_chunk = (HW + chunks - 1) / chunks;
  const int rbeg = chunk * rows_per_chunk, rend = min(HW, rbeg + rows_per_chunk);
  float v[2][VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[0][e] = v[1][e] = 0.f;
  if (prow < rpp) {
    float av[VEC], bv[VEC];
    const size_t crow = group_images > 0 ? (size_t)(n / group_images) * C : 0;      // coefficients per group of the batch
    ldcoef<VEC>(a + crow + vcol * VEC, av);
    ldcoef<VEC>(b + crow + vcol * VEC, bv);
    const size_t base = (size_t)n * HW * C + (size_t)vcol * VEC;
    auto one = [&](int r, const u32x4& xq, const u32x4& rq) {
      float f[VEC], rr[VEC], o[VEC];
      Elem<T>::unpack(xq, f);
      if (res != nullptr) Elem<T>::unpack(rq, rr);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float t = apply_act(fmaf(av[e], f[e], bv[e]), act);
        if (res != nullptr) t += rr[e];
        o[e] = t;
      }
      const u32x4 pk = Elem<T>::pack(o);
      *reinterpret_cast<u32x4*>(out + base + (size_t)r * C) = pk;
      Elem<T>::unpack(pk, o);                    // statistics of what the consumer will read
#pragma unroll
      for (int e = 0; e < VEC; ++e) { v[0][e] += o[e]; v[1][e] = fmaf(o[e], o[e], v[1][e]); }
    };
    int r = rbeg + prow;
    for (; r + rpp < rend; r += 2 * rpp) {        // four 16-byte loads in flight per thread
      const u32x4 x0 = *reinterpret_cast<const u32x4*>(x + base + (size_t)r * C);
      const u32x4 x1 = *reinterpret_cast<const u32x4*>(x + base + (size_t)(r + rpp) * C);
      u32x4 r0 = x0, r1 = x1;
      if (res != nullptr) {
        r0 = *reinterpret_cast<const u32x4*>(res + base + (size_t)r * C);
        r1 = *reinterpret_cast<const u32x4*>(res + base + (size_t)(r + rpp) * C);
      }
      one(r, x0, r0);
      one(r + rpp, x1, r1);
    }
    if (r < rend) {
      const u32x4 x0 = *reinterpret_cast<const u32x4*>(x + base + (size_t)r * C);
      const u32x4 r0 = res != nullptr ? *reinterpret_cast<const u32x4*>(res + base + (size_t)r * C) : x0;
      one(r, x0, r0);
    }
  }
  block_combine_store<2, VEC>(v, cv, rpp, smem, partial + ((size_t)n * chunks + chunk) * 2 * C, C);
}

// SPADE -> conv fusion, the one small kernel in front of the conv (replaces in_finalize + the full-size modulate pass):
//   * InstanceNorm statistics of image n from the partial records (fp64 combine in record order: deterministic) ->
//     mean / rstd (kept for the backward pass) and the INTERIOR-class coefficients of the conv's operand-path transform,
//     A = rstd * (1 + gamma), B = beta - mean * A   (z = relu(A*x + B) == relu(IN(x) * (1 + gamma) + beta));
//   * z of the logical image's 2-pixel frame, whose gamma / beta classes differ pixel by pixel, into the compact ring
//     tensor [N][ring_pixels][C] (geom.h) the conv kernels read instead of x there.
// gb: the (N, 5, 5, 2C) border-class table (normalization.py:24-37 on a constant label map).  grid (blocks, N): every
// block of an image recomputes the image's statistics (a few KB of L2-resident records), block 0 stores them.
DEI2I_D int prep_border_class(int i, int extent) { return i < 2 ? i : (i >= extent - 2 ? 4 - (extent - 1 - i) : 2); }

template <typename T>
__global__ __launch_bounds__(256) void spade_prep_kernel(const T* __restrict__ x, const float* __restrict__ partial,
                                                         const T* __restrict__ gb, float* __restrict__ mean,
                                                         float* __restrict__ rstd, float* __restrict__ A, float* __restrict__ B,
                                                         T* __restrict__ ring, int Hs, int Ws, int C, int up, int chunks,
                                                         double count, float eps) {
  constexpr int VEC = Elem<T>::VEC;
  extern __shared__ float smem[];                 // mean[C] | rstd[C]
  const int n = blockIdx.y, tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) {
    double s0 = 0.0, s1 = 0.0;
    const float* p = partial + (size_t)n * chunks * 2 * C + c;
    for (int r = 0; r < chunks; ++r) { s0 += (double)p[(size_t)r * 2 * C]; s1 += (double)p[(size_t)r * 2 * C + C]; }
    const double mu = s0 / count;
    double var = s1 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float m = (float)mu, rs = (float)(1.0 / sqrt(var + (double)eps));
    smem[c] = m;
    smem[C + c] = rs;
    if (blockIdx.x == 0) {
      mean[(size_t)n * C + c] = m;
      rstd[(size_t)n * C + c] = rs;
      const T* gi = gb + ((size_t)n * 25 + 12) * 2 * C;       // interior class (2, 2)
      const float av = rs * (1.f + Elem<T>::load(gi + c));
      A[(size_t)n * C + c] = av;
      B[(size_t)n * C + c] = Elem<T>::load(gi + C + c) - m * av;
    }
  }
  __syncthreads();
  if (ring == nullptr) return;
  const int H = Hs << up, W = Ws << up, cv = C / VEC;
  const int rp = ring_pixels(H, W);
  const int total = rp * cv;
  for (int i = blockIdx.x * 256 + tid; i < total; i += gridDim.x * 256) {
    const int r = i / cv, c = (i - r * cv) * VEC;
    int y, xx;
    ring_coord(r, H, W, y, xx);
    float xv[VEC], gm[VEC], bt[VEC], o[VEC];
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(x + (((size_t)n * Hs + (y >> up)) * Ws + (xx >> up)) * C + c), xv);
    const size_t gpix = ((size_t)n * 5 + prep_border_class(y, H)) * 5 + prep_border_class(xx, W);
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(gb + gpix * 2 * C + c), gm);
    Elem<T>::unpack(*reinterpret_cast<const u32x4*>(gb + gpix * 2 * C + C + c), bt);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float xh = (xv[e] - smem[c + e]) * smem[C + c + e];
      const float v = fmaf(xh, 1.f + gm[e], bt[e]);
      o[e] = v > 0.f ? v : 0.f;
    }
    *reinterpret_cast<u32x4*>(ring + ((size_t)n * rp + r) * C + c) = Elem<T>::pack(o);
  }
}

}  // namespace dei2i

using namespace dei2i;

static inline bool cv_ok(int dtype, int C) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  return C > 0 && C % vec == 0 && C / vec <= 256;
}
static inline size_t combine_lds(int dtype, int nv) { return (size_t)nv * (dtype == DT_BF16 ? 8 : 4) * 256 * sizeof(float); }

extern "C" {

int dei2i_moments_chunks(int HW) {            // per image: >= 64 rows per workgroup, up to 64 workgroups (x N images)
  int c = HW / 64;
  if (c < 1) c = 1;
  if (c > 64) c = 64;
  return c;
}

int dei2i_moments_partial(int dtype, int N, int HW, int C, const void* x, float* partial, dei2i_stream s) {
  if (N <= 0 || HW <= 0 || !cv_ok(dtype, C) || !x || !partial) return DEI2I_ERR_BAD_ARG;
  const int chunks = dei2i_moments_chunks(HW);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(moments_partial_kernel<bf16_t>, dim3(chunks, N), dim3(256), combine_lds(dtype, 2), (hipStream_t)s,
                       (const bf16_t*)x, partial, HW, C, chunks);
  else
    hipLaunchKernelGGL(moments_partial_kernel<float>, dim3(chunks, N), dim3(256), combine_lds(dtype, 2), (hipStream_t)s,
                       (const float*)x, partial, HW, C, chunks);
  return (int)hipGetLastError();
}

int dei2i_bn_finalize_train(int N, int HW, int C, const float* partial, const float* weight, const float* bias,
                            float* running_mean, float* running_var, float momentum, float eps, float* mean, float* rstd,
                            float* a, float* b, long long* num_batches_tracked, dei2i_stream s) {
  if (N <= 0 || HW <= 0 || C <= 0 || !partial || !weight || !bias || !mean || !rstd || !a || !b) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(C), dim3(combine_threads(N * dei2i_moments_chunks(HW))), 0, (hipStream_t)s, partial, N,
                     dei2i_moments_chunks(HW), C, (double)N * (double)HW, weight, bias, running_mean, running_var, momentum,
                     eps, mean, rstd, a, b, num_batches_tracked, 0);
  return (int)hipGetLastError();
}

int dei2i_bn_finalize_train_chunks(int N, int HW, int C, int chunks, const float* partial, const float* weight, const float* bias,
                                   float* running_mean, float* running_var, float momentum, float eps, float* mean, float* rstd,
                                   float* a, float* b, long long* num_batches_tracked, dei2i_stream s) {
  if (N <= 0 || HW <= 0 || C <= 0 || chunks <= 0 || !partial || !weight || !bias || !mean || !rstd || !a || !b) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(C), dim3(combine_threads(N * chunks)), 0, (hipStream_t)s, partial, N, chunks, C,
                     (double)N * (double)HW, weight, bias, running_mean, running_var, momentum, eps, mean, rstd, a, b,
                     num_batches_tracked, 0);
  return (int)hipGetLastError();
}

int dei2i_bn_finalize_train_groups(int groups, int N, int HW, int C, int chunks, const float* partial, const float* weight, const float* bias,
                                   float* running_mean, float* running_var, int running_stride, float momentum, float eps, float* mean,
                                   float* rstd, float* a, float* b, dei2i_stream s) {
  if (groups <= 0 || N <= 0 || HW <= 0 || C <= 0 || chunks <= 0 || !partial || !weight || !bias || !mean || !rstd || !a || !b ||
      (running_mean == nullptr) != (running_var == nullptr) || (running_mean != nullptr && groups > 1 && running_stride < C))
    return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(bn_finalize_train_kernel, dim3(C, groups), dim3(combine_threads(N * chunks)), 0, (hipStream_t)s, partial, N, chunks, C,
                     (double)N * (double)HW, weight, bias, running_mean, running_var, momentum, eps, mean, rstd, a, b,
                     (long long*)nullptr, running_stride);
  return (int)hipGetLastError();
}

int dei2i_affine_act_stats_fwd(int dtype, int N, int HW, int C, const void* x, const float* a, const float* b, const void* res,
                               int act, void* out, float* partial, dei2i_stream s) {
  if (N <= 0 || HW <= 0 || !cv_ok(dtype, C) || !x || !a || !b || !out || !partial) return DEI2I_ERR_BAD_ARG;
  const int chunks = dei2i_moments_chunks(HW);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(affine_act_stats_kernel<bf16_t>, dim3(chunks, N), dim3(256), combine_lds(dtype, 2), (hipStream_t)s,
                       (const bf16_t*)x, a, b, (const bf16_t*)res, (bf16_t*)out, partial, HW, C, chunks, act, 0);
  else
    hipLaunchKernelGGL(affine_act_stats_kernel<float>, dim3(chunks, N), dim3(256), combine_lds(dtype, 2), (hipStream_t)s,
                       (const float*)x, a, b, (const float*)res, (float*)out, partial, HW, C, chunks, act, 0);
  return (int)hipGetLastError();
}

int dei2i_affine_act_stats_groups_fwd(int dtype, int groups, int N, int HW, int C, const void* x, const float* a, const float* b,
                                      const void* res, int act, void* out, float* partial, dei2i_stream s) {
  if (groups <= 0 || N <= 0 || N % groups != 0 || HW <= 0 || !cv_ok(dtype, C) || !x || !a || !b || !out || !partial) return DEI2I_ERR_BAD_ARG;
  const int chunks = dei2i_moments_chunks(HW);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(affine_act_stats_kernel<bf16_t>, dim3(chunks, N), dim3(256), combine_lds(dtype, 2), (hipStream_t)s,
                       (const bf16_t*)x, a, b, (const bf16_t*)res, (bf16_t*)out, partial, HW, C, chunks, act, N / groups);
  else
    hipLaunchKernelGGL(affine_act_stats_kernel<float>, dim3(chunks, N), dim3(256), combine_lds(dtype, 2), (hipStream_t)s,
                       (const float*)x, a, b, (const float*)res, (float*)out, partial, HW, C, chunks, act, N / groups);
  return (int)hipGetLastError();
}

size_t dei2i_ring_pixels(int H, int W) { return H >= 4 && W >= 4 ? (size_t)ring_pixels(H, W) : 0; }

int dei2i_spade_prep(int dtype, int N, int Hs, int Ws, int C, int up, const void* x, const float* partial, int chunks, float eps,
                     const void* gb_table, float* mean, float* rstd, float* A, float* B, void* ring, dei2i_stream s) {
  if (N <= 0 || Hs <= 0 || Ws <= 0 || !cv_ok(dtype, C) || up < 0 || up > 1 || chunks <= 0 || !x || !partial || !gb_table || !mean ||
      !rstd || !A || !B)
    return DEI2I_ERR_BAD_ARG;
  if (ring != nullptr && ((Hs << up) < 4 || (Ws << up) < 4)) return DEI2I_ERR_BAD_ARG;
  const int vec = dtype == DT_BF16 ? 8 : 4;
  const int work = ring != nullptr ? ring_pixels(Hs << up, Ws << up) * (C / vec) : 0;
  int blocks = (work + 256 * 8 - 1) / (256 * 8);
  if (blocks < 1) blocks = 1;
  if (blocks > 64) blocks = 64;
  const size_t lds = 2 * (size_t)C * sizeof(float);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(spade_prep_kernel<bf16_t>, dim3(blocks, N), dim3(256), lds, (hipStream_t)s, (const bf16_t*)x, partial,
                       (const bf16_t*)gb_table, mean, rstd, A, B, (bf16_t*)ring, Hs, Ws, C, up, chunks, (double)Hs * (double)Ws, eps);
  else
    hipLaunchKernelGGL(spade_prep_kernel<float>, dim3(blocks, N), dim3(256), lds, (hipStream_t)s, (const float*)x, partial,
                       (const float*)gb_table, mean, rstd, A, B, (float*)ring, Hs, Ws, C, up, chunks, (double)Hs * (double)Ws, eps);
  return (int)hipGetLastError();
}

int dei2i_bn_finalize_eval(int C, const float* weight, const float* bias, const float* running_mean, const float* running_var,
                           float eps, float* a, float* b, dei2i_stream s) {
  if (C <= 0 || !weight || !bias || !running_mean || !running_var || !a || !b) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(bn_finalize_eval_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)s, C, weight, bias, running_mean,
                     running_var, eps, a, b);
  return (int)hipGetLastError();
}

int dei2i_in_finalize_chunks(int N, int HW, int C, int chunks, const float* partial, float eps, float* mean, float* rstd,
                             dei2i_stream s) {
  if (N <= 0 || HW <= 0 || C <= 0 || chunks <= 0 || !partial || !mean || !rstd) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(in_finalize_kernel, dim3(C, N), dim3(combine_threads(chunks)), 0, (hipStream_t)s, partial, N, chunks, C,
                     (double)HW, eps, mean, rstd);
  return (int)hipGetLastError();
}

int dei2i_in_finalize(int N, int HW, int C, const float* partial, float eps, float* mean, float* rstd, dei2i_stream s) {
  if (N <= 0 || HW <= 0 || C <= 0 || !partial || !mean || !rstd) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(in_finalize_kernel, dim3(C, N), dim3(combine_threads(dei2i_moments_chunks(HW))), 0, (hipStream_t)s, partial, N,
                     dei2i_moments_chunks(HW), C, (double)HW, eps, mean, rstd);
  return (int)hipGetLastError();
}

int dei2i_colsum_blocks(size_t rows) {      // >= 256 rows per workgroup, one workgroup per CU at most
  size_t blocks = rows / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;
  return (int)blocks;
}

int dei2i_colsum(int dtype, size_t rows, int C, const void* g, float* partial, float* out, dei2i_stream s) {
  if (rows == 0 || !cv_ok(dtype, C) || !g || !partial || !out) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const int blocks = dei2i_colsum_blocks(rows);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(colsum_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), combine_lds(dtype, 1), st, (const bf16_t*)g, partial, rows, C);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, dim3((unsigned)blocks), dim3(256), combine_lds(dtype, 1), st, (const float*)g, partial, rows, C);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, st, (const float*)partial, blocks, C, out);
  return (int)hipGetLastError();
}

int dei2i_bn_bwd_chunks(size_t pixels) {      // one workgroup per chunk: >= 64 rows each, up to 8 workgroups per CU
  size_t c = pixels / 64;
  if (c < 1) c = 1;
  if (c > 2048) c = 2048;
  return (int)c;
}

int dei2i_bn_bwd_partial(int dtype, size_t pixels, int C, const void* dz, const void* y, const float* a, const float* b,
                         const float* mean, const float* rstd, int act, float* partial, dei2i_stream s) {
  if (pixels == 0 || !cv_ok(dtype, C) || !dz || !y || !a || !b || !mean || !rstd || !partial) return DEI2I_ERR_BAD_ARG;
  const int chunks = dei2i_bn_bwd_chunks(pixels);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(bn_bwd_partial_kernel<bf16_t>, dim3(chunks), dim3(256), combine_lds(dtype, 2), (hipStream_t)s,
                       (const bf16_t*)dz, (const bf16_t*)y, a, b, mean, rstd, act, partial, pixels, C, chunks);
  else
    hipLaunchKernelGGL(bn_bwd_partial_kernel<float>, dim3(chunks), dim3(256), combine_lds(dtype, 2), (hipStream_t)s,
                       (const float*)dz, (const float*)y, a, b, mean, rstd, act, partial, pixels, C, chunks);
  return (int)hipGetLastError();
}

int dei2i_bn_bwd_partial_groups(int dtype, int groups, size_t pixels, int C, const void* dz, const void* y, const float* a, const float* b,
                                const float* mean, const float* rstd, int act, float* partial, dei2i_stream s) {
  if (groups <= 0 || pixels == 0 || !cv_ok(dtype, C) || !dz || !y || !a || !b || !mean || !rstd || !partial) return DEI2I_ERR_BAD_ARG;
  const int chunks = dei2i_bn_bwd_chunks(pixels);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(bn_bwd_partial_kernel<bf16_t>, dim3(chunks, groups), dim3(256), combine_lds(dtype, 2), (hipStream_t)s,
                       (const bf16_t*)dz, (const bf16_t*)y, a, b, mean, rstd, act, partial, pixels, C, chunks);
  else
    hipLaunchKernelGGL(bn_bwd_partial_kernel<float>, dim3(chunks, groups), dim3(256), combine_lds(dtype, 2), (hipStream_t)s,
                       (const float*)dz, (const float*)y, a, b, mean, rstd, act, partial, pixels, C, chunks);
  return (int)hipGetLastError();
}

/* `groups` groups of `pixels` pixels each, coefficient rows (groups, C), records (groups, chunks, 2, C): two launches for all groups.
 * group_sums: (groups, 2, C) floats of scratch (each group's own sums, read by its share of the apply launch); dweight / dbias (C): the
 * total over the groups, written -- or added to when `accumulate` (a further use of the same parameters in this backward pass). */
int dei2i_bn_bwd_apply_groups(int dtype, int groups, size_t pixels, int C, const void* dz, const void* y, const float* a, const float* b,
                              const float* mean, const float* rstd, int act, int train, const float* partial, int chunks,
                              float* group_sums, float* dweight, float* dbias, int accumulate, void* dy, dei2i_stream s) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  if (groups <= 0 || pixels == 0 || !cv_ok(dtype, C) || !dz || !y || !a || !b || !mean || !rstd || !partial || !group_sums || !dweight ||
      !dbias || !dy || chunks <= 0)
    return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(bn_bwd_finalize_groups_kernel, dim3(C), dim3(combine_threads(chunks)), 0, st, partial, chunks, C, groups, group_sums,
                     dweight, dbias, accumulate);
  const size_t nvec = pixels * (size_t)(C / vec);
  const unsigned grid = grid_for((nvec + 1) / 2, 256, 256u * 8u);
  const float inv = 1.f / (float)pixels;
  const int cv = C / vec;
  const bool invc = (256 % cv) == 0;
  const float* gw = group_sums;
  const float* gb = group_sums + C;
  if (dtype == DT_BF16) {
    if (invc) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, true>), dim3(grid, groups), dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)y, a, b, mean, rstd, act, train, gw, gb, inv, (bf16_t*)dy, nvec, cv, 2 * C);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, false>), dim3(grid, groups), dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)y, a, b, mean, rstd, act, train, gw, gb, inv, (bf16_t*)dy, nvec, cv, 2 * C);
  } else {
    if (invc) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, true>), dim3(grid, groups), dim3(256), 0, st, (const float*)dz, (const float*)y, a, b, mean, rstd, act, train, gw, gb, inv, (float*)dy, nvec, cv, 2 * C);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<float, false>), dim3(grid, groups), dim3(256), 0, st, (const float*)dz, (const float*)y, a, b, mean, rstd, act, train, gw, gb, inv, (float*)dy, nvec, cv, 2 * C);
  }
  return (int)hipGetLastError();
}

int dei2i_bn_bwd_apply(int dtype, size_t pixels, int C, const void* dz, const void* y, const float* a, const float* b,
                       const float* mean, const float* rstd, int act, int train, const float* partial, int chunks,
                       float* dweight, float* dbias, float* acc_dweight, float* acc_dbias, void* dy, dei2i_stream s) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  if (pixels == 0 || !cv_ok(dtype, C) || !dz || !y || !a || !b || !mean || !rstd || !partial || !dweight || !dbias || !dy ||
      (acc_dweight == nullptr) != (acc_dbias == nullptr))
    return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(combine_threads(chunks)), 0, st, partial, chunks, C, dweight, dbias,
                     acc_dweight, acc_dbias);
  const size_t nvec = pixels * (size_t)(C / vec);
  const unsigned grid = grid_for((nvec + 1) / 2, 256, 256u * 8u);
  const float inv = 1.f / (float)pixels;
  const int cv = C / vec;
  const bool invc = (256 % cv) == 0;
  if (dtype == DT_BF16) {
    if (invc) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)y, a, b, mean, rstd, act, train, (const float*)dweight, (const float*)dbias, inv, (bf16_t*)dy, nvec, cv, 0);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)y, a, b, mean, rstd, act, train, (const float*)dweight, (const float*)dbias, inv, (bf16_t*)dy, nvec, cv, 0);
  } else {
    if (invc) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float*)dz, (const float*)y, a, b, mean, rstd, act, train, (const float*)dweight, (const float*)dbias, inv, (float*)dy, nvec, cv, 0);
    else hipLaunchKernelGGL((bn_bwd_apply_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float*)dz, (const float*)y, a, b, mean, rstd, act, train, (const float*)dweight, (const float*)dbias, inv, (float*)dy, nvec, cv, 0);
  }
  return (int)hipGetLastError();
}

int dei2i_spade_bwd_partial(int dtype, int N, int H, int W, int C, int up, const void* dz, const void* x, const float* mean,
                            const float* rstd, const void* gb, int gb_mode, void* dgb, float* partial, dei2i_stream s) {
  if (N <= 0 || H <= 0 || W <= 0 || !cv_ok(dtype, C) || up < 0 || up > 1 || !dz || !x || !mean || !rstd || !gb || !dgb || !partial)
    return DEI2I_ERR_BAD_ARG;
  if (gb_mode == 1 && (H < 4 || W < 4)) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const int chunks = dei2i_moments_chunks(H * W);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(spade_bwd_partial_kernel<bf16_t>, dim3(chunks, N), dim3(256), combine_lds(dtype, 4), st,
                       (const bf16_t*)dz, (const bf16_t*)x, mean, rstd, (const bf16_t*)gb, gb_mode,
                       gb_mode == 0 ? (bf16_t*)dgb : (bf16_t*)nullptr, partial, H, W, C, up, chunks);
  else
    hipLaunchKernelGGL(spade_bwd_partial_kernel<float>, dim3(chunks, N), dim3(256), combine_lds(dtype, 4), st,
                       (const float*)dz, (const float*)x, mean, rstd, (const float*)gb, gb_mode,
                       gb_mode == 0 ? (float*)dgb : (float*)nullptr, partial, H, W, C, up, chunks);
  if (gb_mode == 1) {
    if (dtype == DT_BF16)
      hipLaunchKernelGGL(spade_bwd_border_kernel<bf16_t>, dim3(25, N), dim3(256), combine_lds(dtype, 2), st, (const bf16_t*)dz,
                         (const bf16_t*)x, mean, rstd, (const bf16_t*)gb, (bf16_t*)dgb, H, W, C, up);
    else
      hipLaunchKernelGGL(spade_bwd_border_kernel<float>, dim3(25, N), dim3(256), combine_lds(dtype, 2), st, (const float*)dz,
                         (const float*)x, mean, rstd, (const float*)gb, (float*)dgb, H, W, C, up);
  }
  return (int)hipGetLastError();
}

int dei2i_spade_bwd_border(int dtype, int N, int H, int W, int C, int up, const void* dz, const void* x, const float* mean,
                           const float* rstd, const void* gb, void* dgb_cls, dei2i_stream s) {
  if (N <= 0 || H < 4 || W < 4 || !cv_ok(dtype, C) || up < 0 || up > 1 || !dz || !x || !mean || !rstd || !gb || !dgb_cls)
    return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(spade_bwd_border_kernel<bf16_t>, dim3(25, N), dim3(256), combine_lds(dtype, 2), st, (const bf16_t*)dz,
                       (const bf16_t*)x, mean, rstd, (const bf16_t*)gb, (bf16_t*)dgb_cls, H, W, C, up);
  else
    hipLaunchKernelGGL(spade_bwd_border_kernel<float>, dim3(25, N), dim3(256), combine_lds(dtype, 2), st, (const float*)dz,
                       (const float*)x, mean, rstd, (const float*)gb, (float*)dgb_cls, H, W, C, up);
  return (int)hipGetLastError();
}

int dei2i_spade_bwd_apply(int dtype, int N, int H, int W, int C, int up, const void* dz, const void* x, const float* mean,
                          const float* rstd, const void* gb, int gb_mode, const float* partial, int chunks, void* dgb_cls,
                          float* coef, const void* addend, void* dx, dei2i_stream s) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  if (N <= 0 || H <= 0 || W <= 0 || !cv_ok(dtype, C) || up < 0 || up > 1 || !dz || !x || !mean || !rstd || !gb || !partial ||
      !coef || !dx)
    return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(spade_bwd_finalize_kernel, dim3(C, N), dim3(combine_threads(chunks)), 0, st, partial, N, chunks, C,
                     (double)H * (double)W, coef, dgb_cls, dtype);
  const size_t total = (size_t)N * (H >> up) * (W >> up) * (C / vec);
  const unsigned grid = grid_for(total, 256, 256u * 16u);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(spade_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)x, mean,
                       rstd, (const bf16_t*)gb, gb_mode, (const float*)coef, (const bf16_t*)addend, (bf16_t*)dx, N, H, W, C, up);
  else
    hipLaunchKernelGGL(spade_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dz, (const float*)x, mean,
                       rstd, (const float*)gb, gb_mode, (const float*)coef, (const float*)addend, (float*)dx, N, H, W, C, up);
  return (int)hipGetLastError();
}

/* Backward of z = act(IN(x)), IN = InstanceNorm2d(affine=False), act of the ReLU family with negative slope `slope` (0.2:
 * LeakyReLU, 1: no activation) -- the SPADE backward kernels with gamma = beta = 0: `zero_table` is an all-zero (N,5,5,2C) table
 * in the compute dtype.  partial: (N, dei2i_moments_chunks(H*W), 4, C) floats, coef: (N, 2, C) floats (scratch). */
int dei2i_in_act_bwd(int dtype, int N, int H, int W, int C, const void* dz, const void* x, const float* mean, const float* rstd,
                     float slope, const void* zero_table, float* partial, float* coef, const void* addend, void* dx, dei2i_stream s) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  if (N <= 0 || H < 4 || W < 4 || !cv_ok(dtype, C) || !dz || !x || !mean || !rstd || !zero_table || !partial || !coef || !dx)
    return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const int chunks = dei2i_moments_chunks(H * W);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(spade_bwd_partial_kernel<bf16_t>, dim3(chunks, N), dim3(256), combine_lds(dtype, 4), st, (const bf16_t*)dz,
                       (const bf16_t*)x, mean, rstd, (const bf16_t*)zero_table, 1, (bf16_t*)nullptr, partial, H, W, C, 0, chunks, slope);
  else
    hipLaunchKernelGGL(spade_bwd_partial_kernel<float>, dim3(chunks, N), dim3(256), combine_lds(dtype, 4), st, (const float*)dz,
                       (const float*)x, mean, rstd, (const float*)zero_table, 1, (float*)nullptr, partial, H, W, C, 0, chunks, slope);
  hipLaunchKernelGGL(spade_bwd_finalize_kernel, dim3(C, N), dim3(combine_threads(chunks)), 0, st, (const float*)partial, N, chunks, C,
                     (double)H * (double)W, coef, (void*)nullptr, dtype);
  const size_t total = (size_t)N * H * W * (C / vec);
  const unsigned grid = grid_for(total, 256, 256u * 16u);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(spade_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)x, mean, rstd,
                       (const bf16_t*)zero_table, 1, (const float*)coef, (const bf16_t*)addend, (bf16_t*)dx, N, H, W, C, 0, slope);
  else
    hipLaunchKernelGGL(spade_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dz, (const float*)x, mean, rstd,
                       (const float*)zero_table, 1, (const float*)coef, (const float*)addend, (float*)dx, N, H, W, C, 0, slope);
  return (int)hipGetLastError();
}

/* Backward of z = act(IN(x) * (1 + gamma) + beta) with per-(n, c) gamma / beta -- AdaIN (stargan-v2/core/model.py:69-80) and
 * InstanceNorm2d(affine=True) (model.py:39-40, 333: gamma = weight - 1, beta = bias for every image) followed by LeakyReLU(0.2)
 * (or any activation of the ReLU family with negative slope `slope`; 1: none): the SPADE backward kernels in class mode on a
 * (N,5,5,2C) table that holds the same (gamma | beta) in all 25 classes, with the activation's slope.  dgb_table: the table's
 * gradient (the caller sums its 25 classes); partial (N, dei2i_moments_chunks(H*W), 4, C) and coef (N, 2, C) floats are scratch. */
int dei2i_in_affine_act_bwd(int dtype, int N, int H, int W, int C, const void* dz, const void* x, const float* mean,
                            const float* rstd, float slope, const void* gb_table, void* dgb_table, float* partial, float* coef,
                            const void* addend, void* dx, dei2i_stream s) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  if (N <= 0 || H < 4 || W < 4 || !cv_ok(dtype, C) || !dz || !x || !mean || !rstd || !gb_table || !dgb_table || !partial || !coef || !dx)
    return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  const int chunks = dei2i_moments_chunks(H * W);
  const size_t total = (size_t)N * H * W * (C / vec);
  const unsigned grid = grid_for(total, 256, 256u * 16u);
  if (dtype == DT_BF16) {
    hipLaunchKernelGGL(spade_bwd_partial_kernel<bf16_t>, dim3(chunks, N), dim3(256), combine_lds(dtype, 4), st, (const bf16_t*)dz,
                       (const bf16_t*)x, mean, rstd, (const bf16_t*)gb_table, 1, (bf16_t*)nullptr, partial, H, W, C, 0, chunks, slope);
    hipLaunchKernelGGL(spade_bwd_border_kernel<bf16_t>, dim3(25, N), dim3(256), combine_lds(dtype, 2), st, (const bf16_t*)dz,
                       (const bf16_t*)x, mean, rstd, (const bf16_t*)gb_table, (bf16_t*)dgb_table, H, W, C, 0, slope);
  } else {
    hipLaunchKernelGGL(spade_bwd_partial_kernel<float>, dim3(chunks, N), dim3(256), combine_lds(dtype, 4), st, (const float*)dz,
                       (const float*)x, mean, rstd, (const float*)gb_table, 1, (float*)nullptr, partial, H, W, C, 0, chunks, slope);
    hipLaunchKernelGGL(spade_bwd_border_kernel<float>, dim3(25, N), dim3(256), combine_lds(dtype, 2), st, (const float*)dz,
                       (const float*)x, mean, rstd, (const float*)gb_table, (float*)dgb_table, H, W, C, 0, slope);
  }
  hipLaunchKernelGGL(spade_bwd_finalize_kernel, dim3(C, N), dim3(combine_threads(chunks)), 0, st, (const float*)partial, N, chunks, C,
                     (double)H * (double)W, coef, dgb_table, dtype);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(spade_bwd_apply_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)dz, (const bf16_t*)x, mean, rstd,
                       (const bf16_t*)gb_table, 1, (const float*)coef, (const bf16_t*)addend, (bf16_t*)dx, N, H, W, C, 0, slope);
  else
    hipLaunchKernelGGL(spade_bwd_apply_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)dz, (const float*)x, mean, rstd,
                       (const float*)gb_table, 1, (const float*)coef, (const float*)addend, (float*)dx, N, H, W, C, 0, slope);
  return (int)hipGetLastError();
}

// block partials of the scalar losses: one stream-ordered scratch per process (the losses of a step run on one stream)
__device__ float g_loss_partials[1024];

int dei2i_bce_logits_fwd(size_t n, const float* x, const float* target, float tconst, float* out, dei2i_stream s) {
  if (n == 0 || !x || !out) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  float* part = nullptr;
  if (hipGetSymbolAddress((void**)&part, HIP_SYMBOL(g_loss_partials)) != hipSuccess) return DEI2I_ERR_BAD_ARG;
  const unsigned nb = grid_for(n, 256, 1024);
  hipLaunchKernelGGL(bce_fwd_kernel, dim3(nb), dim3(256), 0, st, x, target, tconst, n, 1.f / (float)n, part);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)part, (int)nb, 1.f / (float)n, out, 0);
  return (int)hipGetLastError();
}

int dei2i_bce_logits_bwd(size_t n, const float* x, const float* target, float tconst, const float* gout, float* dx,
                         dei2i_stream s) {
  if (n == 0 || !x || !gout || !dx) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, (hipStream_t)s, x, target, tconst, n,
                     1.f / (float)n, gout, dx);
  return (int)hipGetLastError();
}

int dei2i_l1_fwd(size_t n, const float* a, const float* b, float* out, dei2i_stream s) {
  if (n == 0 || !a || !out) return DEI2I_ERR_BAD_ARG;
  hipStream_t st = (hipStream_t)s;
  float* part = nullptr;
  if (hipGetSymbolAddress((void**)&part, HIP_SYMBOL(g_loss_partials)) != hipSuccess) return DEI2I_ERR_BAD_ARG;
  const unsigned nb = grid_for(n, 256, 1024);
  hipLaunchKernelGGL(l1_fwd_kernel, dim3(nb), dim3(256), 0, st, a, b, n, 1.f / (float)n, part);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)part, (int)nb, 1.f / (float)n, out, 0);
  return (int)hipGetLastError();
}

int dei2i_l1_bwd(size_t n, const float* a, const float* b, const float* gout, float* da, float* db, dei2i_stream s) {
  if (n == 0 || !a || !gout) return DEI2I_ERR_BAD_ARG;
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, (hipStream_t)s, a, b, n, 1.f / (float)n, gout,
                     da, db);
  return (int)hipGetLastError();
}

int dei2i_noise_fwd(int dtype, size_t rows, int C, const void* x, const float* noise, const float* weight, void* out,
                    dei2i_stream s) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  if (rows == 0 || C <= 0 || C % vec || !x || !noise || !weight || !out) return DEI2I_ERR_BAD_ARG;
  const size_t nvec = rows * (size_t)(C / vec);
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(noise_fwd_kernel<bf16_t>, dim3(grid_for(nvec, 256, 4096)), dim3(256), 0, (hipStream_t)s,
                       (const bf16_t*)x, noise, weight, (bf16_t*)out, nvec, C / vec);
  else
    hipLaunchKernelGGL(noise_fwd_kernel<float>, dim3(grid_for(nvec, 256, 4096)), dim3(256), 0, (hipStream_t)s,
                       (const float*)x, noise, weight, (float*)out, nvec, C / vec);
  return (int)hipGetLastError();
}

int dei2i_noise_bwd(int dtype, size_t rows, int C, const void* dy, const float* noise, float* partials, float* dweight,
                    int accumulate, dei2i_stream s) {
  const int vec = dtype == DT_BF16 ? 8 : 4;
  if (rows == 0 || C <= 0 || C % vec || !dy || !noise || !partials || !dweight) return DEI2I_ERR_BAD_ARG;
  const size_t nvec = rows * (size_t)(C / vec);
  hipStream_t st = (hipStream_t)s;
  const unsigned nb = grid_for(nvec, 256, 1024);          // `partials`: at least 1024 floats
  if (dtype == DT_BF16)
    hipLaunchKernelGGL(noise_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)dy, noise, nvec, C / vec, partials);
  else
    hipLaunchKernelGGL(noise_bwd_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)dy, noise, nvec, C / vec, partials);
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, st, (const float*)partials, (int)nb, 1.f, dweight, accumulate);
  return (int)hipGetLastError();
}

}  // extern "C"
