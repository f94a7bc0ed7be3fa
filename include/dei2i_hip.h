/*
 * dei2i_hip.h -- C ABI of libdei2i_hip.so, the MI355X (gfx950) kernel library under the de-i2i-gan_amd
 * Python host code.
 *
 * The reference (jason2714/de-i2i-gan) has no FFI of its own: its hot path is Python calling ATen ops
 * implicitly.  Each entry point below therefore replaces the ATen op(s) issued at the cited reference
 * call sites (paths relative to /root/reference/defectGAN).  All functions:
 *   - take raw device pointers + plain ints, launch on the caller's hipStream_t, allocate nothing,
 *   - return 0 on success or a hipError_t / negative dei2i error code (the Python side raises RuntimeError),
 *   - activations are NHWC with the channel stride padded to a 16-byte vector (8 bf16 / 4 f32); `dtype`
 *     0 = bf16 storage + bf16 MFMA (fp32 accumulate), 1 = f32 storage + exact f32 MFMA (parity mode).
 */
#ifndef DEI2I_HIP_H
#define DEI2I_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dei2i_stream;   /* hipStream_t */

#define DEI2I_BF16 0
#define DEI2I_F32 1
#define DEI2I_ACT_NONE 0
#define DEI2I_ACT_RELU 1
#define DEI2I_ACT_LRELU 2      /* LeakyReLU(0.2), architecture.py:15 */
#define DEI2I_PAD_ZERO 0       /* also 'valid' when pad == 0 */
#define DEI2I_PAD_REFLECT 1

#define DEI2I_ERR_BAD_ARG (-2)
#define DEI2I_ERR_WORKSPACE (-3)

/* nn.Conv2d geometry (architecture.py:51-56,95-100,228-233,320-337; normalization.py:17-22;
 * discriminator.py:60-90).  `up` = 1 fuses the preceding nn.Upsample(scale_factor=2) (architecture.py:203). */
typedef struct dei2i_conv {
  int dtype;
  int N, H, W;          /* physical input extent (before the fused upsample) */
  int Cin, Cout;        /* logical channels */
  int CinS, CoutS;      /* channel strides of the x / y tensors (padded to the vector width) */
  int kh, kw, stride, pad, pad_mode, up;
} dei2i_conv;

/* ---- library / device ---- */
int dei2i_version(void);
int dei2i_init(int device);                         /* queries the CU count; optional */
const char* dei2i_error_string(int code);
/* tuning / A-B switches: "gather_gemm_v2" / "wgrad_v2" = 0|1 (LDS-DMA kernels for the bf16 hot shapes) */
int dei2i_set_option(const char* name, int value);
/* diagnostic builds only: device buffer receiving in-kernel s_memtime stamps (option "v2_ablate" = 5) */
int dei2i_set_debug_buffer(void* device_ptr);

/* ---- packed weight layouts (replaces nothing in the reference: layout prep for the kernels) ---- */
size_t dei2i_packed_fwd_elems(const dei2i_conv* c);      /* Cout * kh*kw * CinS */
size_t dei2i_packed_dgrad_elems(const dei2i_conv* c);    /* sum over stride^2 parity classes of Cin * taps * CoutS */
int dei2i_pack_weight_fwd(const dei2i_conv* c, const float* w_oihw, void* packed, dei2i_stream s);
int dei2i_pack_weight_dgrad(const dei2i_conv* c, const float* w_oihw, void* packed, dei2i_stream s);
/* both layouts in one launch (the training path: a weight is re-packed once per optimizer step) */
int dei2i_pack_weight_both(const dei2i_conv* c, const float* w_oihw, void* packed_fwd, void* packed_dgrad, dei2i_stream s);
/* packed fp32 wgrad [Cout][kh*kw][CinS] -> OIHW fp32 (beta = 0: overwrite, 1: accumulate) */
int dei2i_unpack_wgrad(const dei2i_conv* c, const float* dw_packed, float* dw_oihw, float beta, dei2i_stream s);

/* ---- convolution (ATen convolution / convolution_backward) ---- */
void dei2i_conv2d_out_shape(const dei2i_conv* c, int* Ho, int* Wo);
/* dgrad output extent: reflect -> padded logical input (H<<up)+2*pad; zero -> logical input */
void dei2i_conv2d_dgrad_shape(const dei2i_conv* c, int* OH, int* OW);
size_t dei2i_conv2d_workspace_bytes(const dei2i_conv* c);   /* fp32 split-K workspace upper bound (fwd and dgrad) */
int dei2i_conv2d_fwd(const dei2i_conv* c, const void* x, const void* w_packed, const float* bias, int act, void* y,
                     float* ws, size_t ws_bytes, dei2i_stream s);
int dei2i_conv2d_dgrad(const dei2i_conv* c, const void* dy, const void* wd_packed, void* dx_ext, float* ws,
                       size_t ws_bytes, dei2i_stream s);
/* convolution_backward's grad_input in one call: dx (N,H,W,CinS) = the dgrad frame folded back onto the physical
 * input (reflection_pad2d_backward + upsample_nearest2d_backward).  ext_scratch: N*OH*OW*CinS elements of the compute
 * dtype (dei2i_conv2d_dgrad_shape), may be NULL for zero-padded convs without upsample.  Stride-1 reflect convs run
 * as interior GEMM (straight into dx) + reflect-ring GEMM + border fold; everything else as dgrad + dei2i_fold_pad. */
int dei2i_conv2d_dgrad_input(const dei2i_conv* c, const void* dy, const void* wd_packed, void* ext_scratch, void* dx,
                             float* ws, size_t ws_bytes, dei2i_stream s);
/* one wgrad slab: the packed [Cout][kh*kw][CinS] fp32 gradient with Cout rounded up to 8 rows (the kernels store
 * 8-row groups unguarded; the extra rows are never read) */
size_t dei2i_wgrad_slab_elems(const dei2i_conv* c);
/* dw_packed (fp32, dei2i_wgrad_slab_elems), written by a single pass over the pixels (plain stores, deterministic);
 * the production path is dei2i_conv2d_wgrad_oihw, which splits the pixel range over slabs */
int dei2i_conv2d_wgrad(const dei2i_conv* c, const void* x, const void* dy, float* dw_packed, dei2i_stream s);
/* wgrad straight to the OIHW fp32 gradient (what autograd hands to the optimizer).  scratch: fp32 device buffer of at
 * least dei2i_wgrad_slab_elems(c) floats; extra capacity lets the kernels split the pixel range over more
 * workgroups (one partial slab per split, summed and un-packed by a second kernel: no float atomics). */
/* accumulate != 0: dw_oihw += (a further use of the same weight in one backward pass adds into its gradient in place) */
int dei2i_conv2d_wgrad_oihw(const dei2i_conv* c, const void* x, const void* dy, float* scratch, size_t scratch_elems,
                            float* dw_oihw, int accumulate, dei2i_stream s);
/* ---- fp8 (OCP e4m3) forward of the stride-1 3x3 convolutions (BASELINE.json configs[4]) ----
 * c describes the convolution as usual (dtype = DEI2I_BF16: y, bias and the backward path are bf16 / fp32); x and the
 * packed weights are e4m3 bytes with the same NHWC / [Cout][9][CinS] layouts (CinS % 128 == 0).  fp32 accumulation;
 * y = act(dequant[0] * sum(xq * wq) + bias).  No fallback: unsupported shapes return an error, ask _supported first. */
int dei2i_conv2d_fp8_supported(const dei2i_conv* c);                       /* 1 / 0 */
/* out[i] = e4m3(clamp(x[i] * scale, +-448)); n % 8 == 0 */
int dei2i_quantize_fp8(size_t n, const void* x_bf16, float scale, void* out_e4m3, dei2i_stream s);
/* packed forward weights in e4m3: w_oihw * (448 / amax[0]) (amax: device scalar max|w|); also writes the conv's
 * dequant[0] = 1 / (act_scale * 448 / amax[0]), act_scale being the scale the activations are quantised with */
int dei2i_pack_weight_fwd_fp8(const dei2i_conv* c, const float* w_oihw, const float* amax, float act_scale, void* packed_e4m3,
                              float* dequant, dei2i_stream s);
int dei2i_conv2d_fwd_fp8(const dei2i_conv* c, const void* x_e4m3, const void* w_e4m3, const float* bias, const float* dequant,
                         int act, void* y_bf16, dei2i_stream s);
/* reflection_pad2d_backward + upsample_nearest2d_backward: fold the dgrad output (N,OH,OW,C) back onto the
 * physical input (N,H,W,C); optional addend (residual-branch gradient) is summed in the same pass. */
int dei2i_fold_pad(int dtype, int N, int H, int W, int C, int pad, int pad_mode, int up, const void* dx_ext,
                   const void* addend, void* dx, dei2i_stream s);

/* ---- layout at the module boundary (NCHW fp32 <-> NHWC compute dtype) ---- */
int dei2i_nchw_to_nhwc(int dtype, int N, int C, int H, int W, int Cs, const float* src, void* dst, dei2i_stream s);
int dei2i_nhwc_to_nchw(int dtype, int N, int C, int H, int W, int Cs, const void* src, float* dst, dei2i_stream s);
/* same as nchw_to_nhwc with F.interpolate(mode='nearest') from (hs,ws) to (H,W) fused (SPADE label map,
 * normalization.py:29) */
int dei2i_nchw_to_nhwc_resize(int dtype, int N, int C, int hs, int ws, int H, int W, int Cs, const float* src, void* dst,
                              dei2i_stream s);
/* fp32 -> compute dtype cast of a flat buffer */
int dei2i_cast_from_f32(int dtype, size_t n, const float* src, void* dst, dei2i_stream s);

/* ---- per-channel moments (native_batch_norm statistics; generator.py:71,113,124; normalization.py:14) ----
 * partial: (N, chunks, 2, C) fp32 sums / sums of squares.  chunks = dei2i_moments_chunks(HW). */
int dei2i_moments_chunks(int HW);
int dei2i_moments_partial(int dtype, int N, int HW, int C, const void* x, float* partial, dei2i_stream s);
/* BatchNorm2d train: batch stats over (N,HW) -> scale/shift a[c], b[c]; mean/rstd saved for backward; running
 * stats updated (momentum 0.1, unbiased var).  All vectors fp32[C]. */
/* num_batches_tracked (int64 device scalar, may be NULL) is incremented by the same launch */
int dei2i_bn_finalize_train(int N, int HW, int C, const float* partial, const float* weight, const float* bias,
                            float* running_mean, float* running_var, float momentum, float eps, float* mean,
                            float* rstd, float* a, float* b, long long* num_batches_tracked, dei2i_stream s);
int dei2i_bn_finalize_eval(int C, const float* weight, const float* bias, const float* running_mean,
                           const float* running_var, float eps, float* a, float* b, dei2i_stream s);
/* InstanceNorm2d(affine=False): per (n,c) mean / rstd, fp32 [N][C] */
int dei2i_in_finalize(int N, int HW, int C, const float* partial, float eps, float* mean, float* rstd, dei2i_stream s);

/* ---- fused normalisation + activation, forward ---- */
/* out = act(a[c]*x + b[c]) (+ res)   -- BatchNorm apply + LeakyReLU (+ ResBlock identity), architecture.py:116-118,174-176 */
/* out_e4m3 (bf16 only, may be NULL): also write e4m3(out * e4m3_scale) -- the operand copy of the fp8 forward mode */
int dei2i_affine_act_fwd(int dtype, size_t pixels, int C, const void* x, const float* a, const float* b, const void* res,
                         int act, void* out, void* out_e4m3, float e4m3_scale, dei2i_stream s);
/* SPADE + ReLU (normalization.py:24-37, architecture.py:241-245,343-350):
 *   out[n,h,w,c] = relu( (x[n,h>>up,w>>up,c] - mean[n,c]) * rstd[n,c] * (1 + gamma) + beta ) (+ nothing)
 * gb is (N, Hg, Wg, 2*C): gamma = [..., :C], beta = [..., C:].  gb_mode 0: Hg x Wg == output extent;
 * gb_mode 1: Hg = Wg = 5 "border class" table (labels constant over space; class = min(i,2) / 4-(H-1-i)). */
int dei2i_spade_act_fwd(int dtype, int N, int H, int W, int C, int up, const void* x, const float* mean,
                        const float* rstd, const void* gb, int gb_mode, void* out, void* out_e4m3, float e4m3_scale, dei2i_stream s);

/* ---- backward of the above ---- */
/* g = dz * act'(z) (LeakyReLU / ReLU expressed through the saved output z) */
int dei2i_act_bwd(int dtype, size_t n, const void* dz, const void* z, int act, void* g, dei2i_stream s);
/* column sums: out[c] = sum over rows of g[row][c]   (conv bias gradient).  Two deterministic stages (no atomics):
 * partial is fp32 scratch of dei2i_colsum_blocks(rows) * C floats. */
int dei2i_colsum_blocks(size_t rows);
int dei2i_colsum(int dtype, size_t rows, int C, const void* g, float* partial, float* out, dei2i_stream s);
/* BatchNorm backward (train): z = act(a*y+b); g = dz*act'(z); needs sum(g), sum(g*xhat) per channel.
 * bn_bwd_partial writes (chunks, 2, C) partial sums over `pixels` rows; bn_bwd_apply finishes:
 *   dy = a * (g - sum_g/M - xhat * sum_gx/M),  dweight = sum_gx, dbias = sum_g.   train = 0 -> dy = a*g.
 *   acc_dweight / acc_dbias (both or neither, may be NULL): this call's sums are ALSO added into them -- the gradient
 *   buffers of parameters that an earlier node of the same backward pass already wrote. */
int dei2i_bn_bwd_chunks(size_t pixels);
int dei2i_bn_bwd_partial(int dtype, size_t pixels, int C, const void* dz, const void* y, const float* a, const float* b,
                         const float* mean, const float* rstd, int act, float* partial, dei2i_stream s);
int dei2i_bn_bwd_apply(int dtype, size_t pixels, int C, const void* dz, const void* y, const float* a, const float* b,
                       const float* mean, const float* rstd, int act, int train, const float* partial, int chunks,
                       float* dweight, float* dbias, float* acc_dweight, float* acc_dbias, void* dy, dei2i_stream s);
/* SPADE backward.  z = relu(v), v = xhat*(1+gamma)+beta is RECOMPUTED from x and the gamma/beta table in both passes (the
 * op keeps neither its output nor a dxhat tensor): g = dz*[v>0]; dgamma = g*xhat, dbeta = g -> dgb (T: dense tensor, or
 * the (N,5,5,2C) border-class table -- its 24 border classes are written by pass 1, the interior class by pass 2); partial (N, chunks, 4, C) fp32 sums of dxhat = g*(1+gamma), dxhat*xhat and the
 * interior-class dgamma / dbeta.  chunks = dei2i_moments_chunks(H*W of the OUTPUT). */
int dei2i_spade_bwd_partial(int dtype, int N, int H, int W, int C, int up, const void* dz, const void* x, const float* mean,
                            const float* rstd, const void* gb, int gb_mode, void* dgb, float* partial, dei2i_stream s);
/* pass 2 (finalize + apply): dx = rstd*(sum_cell dxhat - cnt*mean(dxhat) - cnt*xhat*mean(dxhat*xhat)) (+ addend).
 * coef: fp32 scratch (N,2,C).  dgb_cls: the class-mode (N,5,5,2C) table of pass 1 (its interior class is written here), or NULL. */
int dei2i_spade_bwd_apply(int dtype, int N, int H, int W, int C, int up, const void* dz, const void* x, const float* mean,
                          const float* rstd, const void* gb, int gb_mode, const float* partial, int chunks, void* dgb_cls,
                          float* coef, const void* addend, void* dx, dei2i_stream s);

/* ---- generator heads + compose (generator.py:266-275) ----
 * raw: (N,H,W,Cs>=4) = [fg_pre(3), prob_pre(1)]; x_in, out: NCHW fp32 (N,3,H,W); prob: NCHW fp32 (N,1,H,W)
 *   out = x*(1-p) + tanh(fg_pre)*p,  p = sigmoid(prob_pre) */
int dei2i_compose_fwd(int dtype, int N, int H, int W, int Cs, const void* raw, const float* x_in, float* out, float* prob,
                      dei2i_stream s);
int dei2i_compose_bwd(int dtype, int N, int H, int W, int Cs, const void* raw, const float* x_in, const float* d_out,
                      const float* d_prob, void* d_raw, float* d_x, dei2i_stream s);
/* NaN guard (generator.py:266-267): flag = any(isnan(x)); then, only if flag: nan->0, +-inf->+-max. No host sync. */
int dei2i_nan_guard(int dtype, size_t n, void* x, int* flag, dei2i_stream s);

/* ---- losses (models/base_model.py:68-80), fp32 tensors ---- */
/* mean( max(x,0) - x*t + log1p(exp(-|x|)) ); target == NULL -> constant `tconst`.  out[0] = loss, summed through ordered
 * block partials (no atomics: bit-reproducible; `out` needs no zero fill). */
int dei2i_bce_logits_fwd(size_t n, const float* x, const float* target, float tconst, float* out, dei2i_stream s);
int dei2i_bce_logits_bwd(size_t n, const float* x, const float* target, float tconst, const float* gout, float* dx,
                         dei2i_stream s);
/* mean |a - b| ; b == NULL -> 0; out[0] = loss.  backward: da = sign(a-b)*gout/n, db = -da (either may be NULL) */
int dei2i_l1_fwd(size_t n, const float* a, const float* b, float* out, dei2i_stream s);
int dei2i_l1_bwd(size_t n, const float* a, const float* b, const float* gout, float* da, float* db, dei2i_stream s);

/* ---- NoiseInjection (models/networks/architecture.py:374-389, the 'constant' weight the blocks use) ----
 * x, out, dy: NHWC activations viewed as [rows = N*H*W][C] in the compute dtype (C a multiple of 8 for bf16, 4 for fp32);
 * noise: one fp32 N(0,1) value per row; weight: the module's single fp32 scalar (device pointer).
 * fwd: out[r][c] = x[r][c] + weight * noise[r].  bwd: dweight[0] = sum_r noise[r] * sum_c dy[r][c] (ordered block partials:
 * `partials` holds >= 1024 floats of scratch; accumulate != 0 adds into dweight); the activation gradient is dy itself. */
int dei2i_noise_fwd(int dtype, size_t rows, int C, const void* x, const float* noise, const float* weight, void* out,
                    dei2i_stream s);
int dei2i_noise_bwd(int dtype, size_t rows, int C, const void* dy, const float* noise, float* partials, float* dweight,
                    int accumulate, dei2i_stream s);

/* ---- fused multi-tensor Adam (torch.optim.Adam semantics; trainers/base_trainer.py:75-89) ----
 * table: device array of `count` records {p, g, m, v (fp32*), n (int64)}; one launch updates all. */
typedef struct dei2i_adam_rec {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} dei2i_adam_rec;
int dei2i_adam_step(const dei2i_adam_rec* table_dev, int count, int64_t max_n, float lr, float beta1, float beta2,
                    float eps, float bias_c1, float bias_c2_sqrt, float grad_scale, float decoupled_decay, dei2i_stream s);
/* decoupled_decay: AdamW's weight decay (p *= 1 - lr*decay before the Adam update); 0 = torch.optim.Adam */
/* torch.optim.SGD (kind 0: p -= lr*g) / torch.optim.RMSprop (kind 1: sq = alpha*sq + (1-alpha)*g^2 in rec.m; p -= lr*g/(sqrt(sq)+eps))
 * as trainers/base_trainer.py:71-74 constructs them (lr only: torch's defaults otherwise); same pointer table, rec.v unused */
int dei2i_sgd_rmsprop_step(const dei2i_adam_rec* table_dev, int count, int64_t max_n, int kind, float lr, float alpha, float eps,
                           float grad_scale, dei2i_stream s);

/* ---- spectral normalisation of a conv weight (--use_spectral; torch.nn.utils.spectral_norm semantics, one power
 * iteration per training-mode forward) ----  W = weight_orig as a (Cout, K) fp32 matrix, u (Cout) / v (K) the module's
 * buffers (updated in place when iterate != 0); u_used / v_used receive the vectors sigma was computed with (the backward
 * pass needs them: later forwards iterate the buffers again); scal: 4 floats (|W^T u|, |W v|, sigma, -); w_eff = W / sigma.
 * backward: dW (+)= (G - sum(G . w_eff) * u v^T) / sigma (accumulate != 0 adds into dW).  scratch:
 * dei2i_spectral_scratch_floats(Cout, K) floats forward, 1024 floats backward. */
size_t dei2i_spectral_scratch_floats(int Cout, int K);
int dei2i_spectral_fwd(int Cout, int K, const float* W, float* u, float* v, int iterate, float* scratch, float* u_used,
                       float* v_used, float* scal, float* w_eff, dei2i_stream s);
int dei2i_spectral_bwd(int Cout, int K, const float* G, const float* w_eff, const float* u_used, const float* v_used,
                       const float* scal, float* scratch, float* dW, int accumulate, dei2i_stream s);

/* Eval-mode BatchNorm folded into the weights of the conv in front of it (the ConvBlock of architecture.py:79-118 with running
 * statistics -- the generator passes of the D step, defectgan_model.py:251-262, and inference): w_eff[co] = a[co] * W[co],
 * b_eff[co] = bias[co] - running_mean[co] * a[co], a = weight * rsqrt(running_var + eps).  W / w_eff: Cout x K fp32 (OIHW). */
int dei2i_fold_bn_weight(int Cout, int K, const float* W, const float* bn_weight, const float* bn_bias, const float* running_mean,
                         const float* running_var, float eps, float* w_eff, float* b_eff, dei2i_stream s);

/* ---- BatchNorm over a batch that carries several passes: statistics, apply and backward per GROUP of the batch, one launch for all
 * groups (the paired generator passes of the G loss: defectgan_model.py:185-190 as two passes over 2 x batch).  Same kernels and
 * argument meaning as the one-group entry points above (finalize_train_chunks, affine_act / affine_act_stats, bn_bwd_partial / apply); N /
 * pixels are PER GROUP, the groups are consecutive in the activation tensors, mean / rstd / a / b hold one row of C per group, the
 * statistics records one block per group; running_mean / running_var of group g at + g * running_stride floats. ---- */
int dei2i_bn_finalize_train_groups(int groups, int N, int HW, int C, int chunks, const float* partial, const float* weight, const float* bias,
                                   float* running_mean, float* running_var, int running_stride, float momentum, float eps, float* mean,
                                   float* rstd, float* a, float* b, dei2i_stream s);
int dei2i_affine_act_groups_fwd(int dtype, int groups, size_t pixels, int C, const void* x, const float* a, const float* b, const void* res,
                                int act, void* out, void* out_e4m3, float e4m3_scale, dei2i_stream s);
int dei2i_affine_act_stats_groups_fwd(int dtype, int groups, int N, int HW, int C, const void* x, const float* a, const float* b,
                                      const void* res, int act, void* out, float* partial, dei2i_stream s);   /* N: the whole batch */
int dei2i_bn_bwd_partial_groups(int dtype, int groups, size_t pixels, int C, const void* dz, const void* y, const float* a, const float* b,
                                const float* mean, const float* rstd, int act, float* partial, dei2i_stream s);
int dei2i_bn_bwd_apply_groups(int dtype, int groups, size_t pixels, int C, const void* dz, const void* y, const float* a, const float* b,
                              const float* mean, const float* rstd, int act, int train, const float* partial, int chunks,
                              float* group_sums, float* dweight, float* dbias, int accumulate, void* dy, dei2i_stream s);

/* ---- SPADE's label path, second stage, for all modules of a generator at once (csrc/label_path.hip) ----
 * normalization.py:17-37 with a constant label map (the 5 x 5 class image of networks/architecture.py SPADE): gamma | beta =
 * conv3x3(actv; mlp_gamma | mlp_beta) + bias per module, every module reading `hidden` channels at its offset `in_off` of ONE activation
 * tensor actv (N, 5, 5, ctot) bf16 (the modules' mlp_shared convs run as one conv over their concatenated filters).  bf16 only.
 * One table entry per module (<= 16); C = norm_nc (multiple of 16), hidden in {32, 64, 96, 128}.
 *   pack : gamma_weight / beta_weight (C, hidden, 3, 3 fp32) -> packed_fwd, packed_dgrad (dei2i_label_gb_packed_elems(C, hidden) bf16 each)
 *   fwd  : gb (N, 5, 5, 2C) bf16 = [gamma | beta]
 *   dgrad: dactv (N, 5, 5, ctot) bf16, the module's channel slice written from gb = dL/d(gb) (zeros when live == 0)
 *   wgrad: d_gamma_weight / d_beta_weight (C, hidden, 3, 3 fp32), d_gamma_bias / d_beta_bias (C) written in full, from gb = dL/d(gb) */
typedef struct dei2i_label_mod {
  const float* gamma_weight;
  const float* beta_weight;
  const float* gamma_bias;
  const float* beta_bias;
  void* packed_fwd;
  void* packed_dgrad;
  void* gb;
  float* d_gamma_weight;
  float* d_beta_weight;
  float* d_gamma_bias;
  float* d_beta_bias;
  int C;
  int in_off;
  int live;
  int reserved;
} dei2i_label_mod;
size_t dei2i_label_gb_packed_elems(int C, int hidden);
int dei2i_label_gb_pack(const dei2i_label_mod* mods, int n, int hidden, dei2i_stream s);
int dei2i_label_gb_fwd(const dei2i_label_mod* mods, int n, int hidden, int ctot, int N, const void* actv, dei2i_stream s);
int dei2i_label_gb_dgrad(const dei2i_label_mod* mods, int n, int hidden, int ctot, int N, void* dactv, dei2i_stream s);
int dei2i_label_gb_wgrad(const dei2i_label_mod* mods, int n, int hidden, int ctot, int N, const void* actv, dei2i_stream s);

/* ---- in-library kernel timing (bench.py roofline leg): HIP events around every launch of one kernel family ---- */
#define DEI2I_PROF_GATHER_GEMM 0  /* the other conv forward / dgrad kernels: gather GEMM v1 / v2, thin convs */
#define DEI2I_PROF_WGRAD 1
#define DEI2I_PROF_HALO_CONV 2   /* the halo-resident 3x3 conv kernels, forward / zero-boundary launches (not part of family 0) */
#define DEI2I_PROF_HALO_FOLD 3   /* their FOLD launches (input gradients of the reflect-padded convs: run during backward passes,
                                    i.e. beside the weight-gradient kernels of ops' side stream) */
/* on = 1: HIP events around every launch of the family + FLOP count; on = 2: launch and FLOP count only (no events: the
 * stream sees nothing extra, total_ms reads 0); on = k >= 3: events around every k-th launch only (a sample -- two event
 * records per launch break the stream's back-to-back dispatch); on = 0: off */
int dei2i_prof_enable(int family, int on);
/* (call BEFORE dei2i_prof_collect) how many launches were bracketed with events since enable, and their FLOPs: total_ms of
 * dei2i_prof_collect belongs to these */
int dei2i_prof_collect_timed(int family, int64_t* timed_launches, double* timed_flops);
/* synchronises the recorded events; returns launches, total ms and total algorithmic FLOPs since enable */
int dei2i_prof_collect(int family, int64_t* launches, double* total_ms, double* total_flops);

/* ---- fused conv + norm + act (BASELINE.json configs[1]; SURVEY.md Appendix B groups G3..G16) -------------------------
 * The normalisation + activation that the reference applies BETWEEN two convs (architecture.py:116-118: conv -> BatchNorm
 * -> LeakyReLU; architecture.py:241-245,343-350 with normalization.py:24-37: SPADE's InstanceNorm * (1+gamma) + beta ->
 * ReLU -> conv) runs inside the halo-resident conv kernels:
 *   - `stats` (epilogue): per-channel sum / sum of squares of the conv's stored output, one record per 8x32-pixel tile,
 *     (N, dei2i_conv2d_stats_chunks, 2, CoutS) fp32 -- the dei2i_moments_partial layout, consumed by the finalize kernels;
 *   - `pro` (operand path): the conv's INPUT is z = act(A[n*n_stride + c] * x + B[...]), act(v) = max(v,0) + slope*min(v,0),
 *     applied to the input halo in LDS; z is never written to memory.  BatchNorm: A = a, B = b of dei2i_bn_finalize_*,
 *     n_stride = 0, slope = 0.2.  SPADE on a constant label map: A, B, ring from dei2i_spade_prep, n_stride = CinS,
 *     slope = 0; `ring` holds z of the logical input's 2-pixel frame (whose gamma / beta differ per pixel),
 *     [N][dei2i_ring_pixels(H<<up, W<<up)][CinS] in the compute dtype.
 * bf16, 3x3, stride 1, pad 1 only; no fallback inside: ask dei2i_conv2d_fused_supported / _wgrad_pro_supported first. */
typedef struct dei2i_pro {
  const float* A;
  const float* B;
  int n_stride;
  float slope;
  const void* ring;
} dei2i_pro;
int dei2i_conv2d_fused_supported(const dei2i_conv* c, int want_pro);          /* 1 / 0 */
int dei2i_conv2d_stats_chunks(const dei2i_conv* c);                           /* records per image written to `stats` */
int dei2i_conv2d_fwd_fused(const dei2i_conv* c, const void* x, const void* w_packed, const float* bias, int act, void* y,
                           const dei2i_pro* pro, float* stats, dei2i_stream s);   /* pro, stats: either may be NULL */
int dei2i_conv2d_wgrad_pro_supported(const dei2i_conv* c);
/* dei2i_conv2d_wgrad_oihw for a conv whose input was normalised on the operand path: x is the UN-normalised tensor */
int dei2i_conv2d_wgrad_oihw_pro(const dei2i_conv* c, const void* x, const void* dy, float* scratch, size_t scratch_elems,
                                float* dw_oihw, int accumulate, const dei2i_pro* pro, dei2i_stream s);
/* SPADE -> nearest x2 upsample -> conv (architecture.py:241-245) with z kept at the SOURCE resolution: gamma / beta of a constant
 * label map only differ on the 2-pixel frame of the upsampled image, so z[y][x] = z_src[y >> 1][x >> 1] everywhere else; the
 * conv (c->up = 1) reads z_src through its fused upsample and the frame's pixels from `ring` (dei2i_spade_prep) -- the
 * upsampled, normalised tensor (4x the bytes) is never written or read.  dei2i_conv2d_wgrad_oihw_pro with pro->A = pro->B =
 * NULL and pro->ring set is the matching weight gradient (x = z_src). */
int dei2i_conv2d_ring_supported(const dei2i_conv* c);
int dei2i_conv2d_fwd_ring(const dei2i_conv* c, const void* z_src, const void* ring, const void* w_packed, const float* bias, int act,
                          void* y, float* stats, dei2i_stream s);
/* Backward reductions of the normalisation layer in FRONT of a conv, taken from the epilogue of that conv's input-gradient
 * launch (the dz tile is still on chip) instead of by dei2i_spade_bwd_partial / dei2i_bn_bwd_partial re-reading dz and x:
 *   kind 1  z = relu(IN(x)*(1+gamma)+beta) with the (N,5,5,2C) class table `gb` (normalization.py:24-37); mean / rstd (N,C);
 *           partial (N, chunks, 4, C) -- the record layout of dei2i_spade_bwd_partial, for dei2i_spade_bwd_apply; the 24 border
 *           classes of the table's gradient still come from dei2i_spade_bwd_border
 *   kind 2  z = act(a*y + b), BatchNorm + activation (architecture.py:116-118), `x` = y; a / b / mean / rstd (C);
 *           partial (N*chunks, 2, C) -- the record layout of dei2i_bn_bwd_partial, for dei2i_bn_bwd_apply
 * chunks = dei2i_conv2d_dgrad_norm_chunks(c) records per image.  `x` has the conv input's channel stride (c->CinS) and, with
 * `up`, half the resolution of the conv's input (SPADE behind a nearest x2 upsample; c describes the conv at the upsampled
 * size with c->up = 0).  bf16, stride-1 reflect-padded 3x3 convs on grids the 16 x 32 tile kernel takes: ask _supported first --
 * dei2i_conv2d_dgrad_input_norm has no fallback inside. */
typedef struct dei2i_epi_norm {
  int kind;
  int up;
  int act;
  int group_images;   /* kind 2: a / b / mean / rstd hold one row of C per group of this many images (BatchNorm statistics taken
                         per group of a batch that carries several passes); 0: one row for the whole batch */
  const void* x;
  const float* mean;
  const float* rstd;
  const void* gb;
  const float* a;
  const float* b;
  float* partial;
} dei2i_epi_norm;
int dei2i_conv2d_dgrad_norm_supported(const dei2i_conv* c);                   /* 1 / 0 */
int dei2i_conv2d_dgrad_norm_chunks(const dei2i_conv* c);
int dei2i_conv2d_dgrad_input_norm(const dei2i_conv* c, const void* dy, const void* wd_packed, void* dx, const dei2i_epi_norm* en,
                                  dei2i_stream s);
/* Backward of z = act(InstanceNorm2d(x)) (affine=False) with an activation of the ReLU family of negative slope `slope` (0.2:
 * LeakyReLU, 1: none) -- the norm of the conv StyleExtractor's blocks (models/networks/extractor.py:50-80 with
 * architecture.py:79-176): the SPADE backward kernels with gamma = beta = 0.  zero_table: an all-zero (N,5,5,2C) table in the
 * compute dtype; partial (N, dei2i_moments_chunks(H*W), 4, C) and coef (N, 2, C) floats are scratch; addend (optional) is added to dx. */
int dei2i_in_act_bwd(int dtype, int N, int H, int W, int C, const void* dz, const void* x, const float* mean, const float* rstd,
                     float slope, const void* zero_table, float* partial, float* coef, const void* addend, void* dx, dei2i_stream s);
/* The same with a real (gamma | beta): z = act(IN(x) * (1 + gamma) + beta), gamma / beta per (n, c) -- AdaIN (stargan-v2/core/model.py:
 * 69-80) and InstanceNorm2d(affine=True) + LeakyReLU (model.py:39-40,56-61,333-334: gamma = weight - 1, beta = bias).  gb_table: the
 * (N,5,5,2C) class table holding the same (gamma | beta) in all 25 classes (compute dtype); dgb_table: its gradient, same shape (the
 * caller sums the 25 classes).  Forward of the op: dei2i_in_finalize + dei2i_affine_act_img_fwd with A = rstd (1 + gamma),
 * B = beta - mean A. */
int dei2i_in_affine_act_bwd(int dtype, int N, int H, int W, int C, const void* dz, const void* x, const float* mean,
                            const float* rstd, float slope, const void* gb_table, void* dgb_table, float* partial, float* coef,
                            const void* addend, void* dx, dei2i_stream s);
/* nn.AvgPool2d(2, 2) on NHWC (N, H, W, C), H and W even (architecture.py:157-168); out / dout: (N, H/2, W/2, C) */
int dei2i_avgpool2_fwd(int dtype, int N, int H, int W, int C, const void* x, void* out, dei2i_stream s);
int dei2i_avgpool2_bwd(int dtype, int N, int H, int W, int C, const void* dout, void* dx, dei2i_stream s);
/* the border-class half of dei2i_spade_bwd_partial alone (class mode): dgb_cls[n, cy, cx, :] for the 24 classes (cy, cx) != (2, 2) */
int dei2i_spade_bwd_border(int dtype, int N, int H, int W, int C, int up, const void* dz, const void* x, const float* mean,
                           const float* rstd, const void* gb, void* dgb_cls, dei2i_stream s);
/* out[n, p, c] = max(A[n, c] * x[n, p, c] + B[n, c], 0) + slope * min(..., 0): an affine + activation with per-IMAGE
 * coefficients (SPADE's interior class at the source resolution; HW pixels per image) */
int dei2i_affine_act_img_fwd(int dtype, int N, int HW, int C, const void* x, const float* A, const float* B, float slope, void* out,
                             dei2i_stream s);
size_t dei2i_ring_pixels(int H, int W);                                        /* 4*W + 4*(H-4) */
/* InstanceNorm finalize + SPADE coefficient preparation in one launch: mean / rstd (N,C) from the partial records (`chunks`
 * per image, HW = Hs*Ws pixels each image), A = rstd*(1+gamma_int), B = beta_int - mean*A for the interior class of the
 * (N,5,5,2C) table, and (ring != NULL) relu(IN(x)*(1+gamma)+beta) of the 2-pixel frame of the (Hs<<up, Ws<<up) image */
int dei2i_spade_prep(int dtype, int N, int Hs, int Ws, int C, int up, const void* x, const float* partial, int chunks, float eps,
                     const void* gb_table, float* mean, float* rstd, float* A, float* B, void* ring, dei2i_stream s);
/* dei2i_in_finalize / dei2i_bn_finalize_train with an explicit record count per image (records written by a conv epilogue) */
int dei2i_in_finalize_chunks(int N, int HW, int C, int chunks, const float* partial, float eps, float* mean, float* rstd,
                             dei2i_stream s);
int dei2i_bn_finalize_train_chunks(int N, int HW, int C, int chunks, const float* partial, const float* weight, const float* bias,
                                   float* running_mean, float* running_var, float momentum, float eps, float* mean, float* rstd,
                                   float* a, float* b, long long* num_batches_tracked, dei2i_stream s);
/* dei2i_affine_act_fwd that also writes the statistics records of its output: partial (N, dei2i_moments_chunks(HW), 2, C) */
int dei2i_affine_act_stats_fwd(int dtype, int N, int HW, int C, const void* x, const float* a, const float* b, const void* res,
                               int act, void* out, float* partial, dei2i_stream s);

/* ---- which kernel served a call: host-side launch counters per MFMA kernel family (tests assert the family, so a
 * fall-through from a tuned kernel to the generic GEMM cannot pass unnoticed).  dei2i_launch_counts copies up to n
 * counters and returns how many families exist; dei2i_kernel_name(i) names family i ("halo_conv", "gather_v2",
 * "gather_v1", "thin_cin", "thin_cout", "wgrad_halo", "wgrad_v2", "wgrad_v1", "wgrad_thin", "halo_conv_fp8",
 * "halo16_conv"). */
int dei2i_launch_counts(int64_t* out, int n);
void dei2i_launch_counts_reset(void);
const char* dei2i_kernel_name(int kid);

#ifdef __cplusplus
}
#endif
#endif /* DEI2I_HIP_H */
