// Gather geometry shared by the implicit-GEMM conv kernels (forward, dgrad, wgrad) and by the CPU
// geometry check in tests/hostcheck.  One descriptor describes "output pixel m, tap t -> source pixel":
//
//   m -> (n, oy, ox)                       over an Ho x Wo output grid, M = N*Ho*Wo
//   t -> (ty, tx) on a th x tw tap grid
//   y = oy*sh + by0 + ty*ys ; x likewise   logical source coordinate (post-upsample extent Hl x Wl)
//   boundary: reflect (index -1 -> 1, no repeat) or zero (out of range contributes 0)
//   nearest upsample: physical row = y >> up
//   output pixel lands at (n, oy*oys + oy0, ox*oxs + ox0) of an (N, OH, OW, ldc) tensor
//
// forward conv : sh = stride, by0 = -pad, ys = +1                      (reference: nn.Conv2d call sites,
//                architecture.py:51-56,95-100,228-233; normalization.py:17-22)
// dgrad        : source = dY, ys = -1 (the tap sign flip IS the kernel flip), zero boundary on dY's extent,
//                one launch per output parity class for stride 2 (oys = 2, oy0 = parity)
// wgrad        : the forward descriptor; the reduction runs over m.
#pragma once
#include "common.h"

namespace dei2i {

struct GatherDesc {
  int N, Hs, Ws, Cs;      // physical source tensor (NHWC), Cs = channel count padded to the vector width
  int Clog;               // logical (unpadded) source channels -- FLOP accounting only
  int Hl, Wl, up;         // logical extent (Hs << up), upsample shift
  int Ho, Wo, M;          // output grid, M = N*Ho*Wo
  int sh, sw, by0, bx0;   // y = oy*sh + by0 + ty*ys
  int th, tw, ys, xs;     // tap grid and per-tap step
  int pad_mode;           // PAD_ZERO / PAD_REFLECT
  int K;                  // th*tw*Cs
  // weight addressing: row stride wK, and tap (ty, tx) of this descriptor is weight tap (wty0+ty)*wtw + (wtx0+tx) --
  // a descriptor may cover a sub-rectangle of the packed tap grid (reflect-ring launches)
  int wK, wtw, wty0, wtx0;
  int OH, OW, oys, oxs, oy0, ox0;   // output placement
  int out_identity;       // 1 when the output pixel index == m
  FastDiv fd_cs, fd_tw, fd_wo, fd_howo;
};

inline void finish_desc(GatherDesc& g) {
  g.Hl = g.Hs << g.up;
  g.Wl = g.Ws << g.up;
  g.M = g.N * g.Ho * g.Wo;
  g.K = g.th * g.tw * g.Cs;
  g.fd_cs = make_fastdiv((uint32_t)g.Cs);
  g.fd_tw = make_fastdiv((uint32_t)g.tw);
  g.fd_wo = make_fastdiv((uint32_t)g.Wo);
  g.fd_howo = make_fastdiv((uint32_t)(g.Ho * g.Wo));
  g.out_identity = (g.OH == g.Ho && g.OW == g.Wo && g.oys == 1 && g.oxs == 1 && g.oy0 == 0 && g.ox0 == 0) ? 1 : 0;
}

DEI2I_HD void decode_m(const GatherDesc& g, int m, int& n, int& oy, int& ox) {
  uint32_t um = (uint32_t)m;
  uint32_t un = fd_div(um, g.fd_howo);
  uint32_t rem = um - un * (uint32_t)(g.Ho * g.Wo);
  uint32_t uy = fd_div(rem, g.fd_wo);
  n = (int)un;
  oy = (int)uy;
  ox = (int)(rem - uy * (uint32_t)g.Wo);
}

// element offset, within a packed weight row, of this descriptor's tap (ty, tx)
DEI2I_HD int weight_tap_offset(const GatherDesc& g, int ty, int tx) { return ((g.wty0 + ty) * g.wtw + g.wtx0 + tx) * g.Cs; }

DEI2I_HD void decode_k(const GatherDesc& g, int k, int& tap, int& ci) {
  uint32_t t = fd_div((uint32_t)k, g.fd_cs);
  tap = (int)t;
  ci = k - (int)t * g.Cs;
}

// 1-D boundary handling; returns -1 when the coordinate contributes zero.
DEI2I_HD int bound_coord(int v, int extent, int pad_mode) {
  if (pad_mode == PAD_REFLECT) {
    v = v < 0 ? -v : v;
    v = v >= extent ? 2 * extent - 2 - v : v;
    return v;
  }
  return (v < 0 || v >= extent) ? -1 : v;
}

// source pixel index (n*Hs + y)*Ws + x, or -1 for a zero contribution
DEI2I_HD int src_pixel(const GatherDesc& g, int n, int oy, int ox, int tap) {
  int ty = (int)fd_div((uint32_t)tap, g.fd_tw);
  int tx = tap - ty * g.tw;
  int y = bound_coord(oy * g.sh + g.by0 + ty * g.ys, g.Hl, g.pad_mode);
  int x = bound_coord(ox * g.sw + g.bx0 + tx * g.xs, g.Wl, g.pad_mode);
  if ((y | x) < 0) return -1;
  return (n * g.Hs + (y >> g.up)) * g.Ws + (x >> g.up);
}

DEI2I_HD int out_pixel(const GatherDesc& g, int n, int oy, int ox) {
  return (n * g.OH + oy * g.oys + g.oy0) * g.OW + ox * g.oxs + g.ox0;
}

// ---------------------------------------------------------------------------------------------
// Descriptor builders (host).  conv parameters follow nn.Conv2d; `Cs` is the padded channel count
// of whatever tensor is gathered.
// ---------------------------------------------------------------------------------------------
struct ConvShape {
  int N, H, W;        // physical input (before the fused nearest upsample)
  int Cin, Cout;      // logical channel counts
  int kh, kw, stride, pad, pad_mode, up;
};

inline int conv_out_dim(int in_logical, int k, int stride, int pad) { return (in_logical + 2 * pad - k) / stride + 1; }

inline GatherDesc make_fwd_desc(const ConvShape& c, int Cs) {
  GatherDesc g{};
  g.Clog = c.Cin;
  g.N = c.N; g.Hs = c.H; g.Ws = c.W; g.Cs = Cs; g.up = c.up;
  g.Ho = conv_out_dim(c.H << c.up, c.kh, c.stride, c.pad);
  g.Wo = conv_out_dim(c.W << c.up, c.kw, c.stride, c.pad);
  g.sh = g.sw = c.stride; g.by0 = g.bx0 = -c.pad;
  g.th = c.kh; g.tw = c.kw; g.ys = g.xs = 1;
  g.pad_mode = c.pad_mode;
  g.OH = g.Ho; g.OW = g.Wo; g.oys = g.oxs = 1; g.oy0 = g.ox0 = 0;
  finish_desc(g);
  g.wK = g.K; g.wtw = g.tw; g.wty0 = g.wtx0 = 0;
  return g;
}

// Gradient w.r.t. the conv's (logical, upsampled) input.
//  reflect : the output is the gradient of the PADDED logical input, (N, Hl+2p, Wl+2p, Cin) -- fold_pad folds
//            the halo (and the 2x2 upsample cells) back afterwards;
//  zero    : the output is the gradient of the logical input itself (the halo's gradient is dropped).
// One descriptor per output parity class (ay, ax) in [0,stride)^2; class (ay,ax) uses taps ky = ay + stride*j.
inline int dgrad_num_classes(const ConvShape& c) { return c.stride * c.stride; }
inline int dgrad_taps(int k, int stride, int a) { return a < k ? (k - a + stride - 1) / stride : 0; }
inline int dgrad_out_h(const ConvShape& c) { return (c.H << c.up) + (c.pad_mode == PAD_REFLECT ? 2 * c.pad : 0); }
inline int dgrad_out_w(const ConvShape& c) { return (c.W << c.up) + (c.pad_mode == PAD_REFLECT ? 2 * c.pad : 0); }

inline GatherDesc make_dgrad_desc(const ConvShape& c, int CoutS, int ay, int ax) {
  GatherDesc g{};
  const int Ho = conv_out_dim(c.H << c.up, c.kh, c.stride, c.pad);
  const int Wo = conv_out_dim(c.W << c.up, c.kw, c.stride, c.pad);
  const int OHt = dgrad_out_h(c), OWt = dgrad_out_w(c);
  // coordinate in the padded frame: hp = (h_out_tensor) + off, off = 0 (reflect) or pad (zero mode)
  const int off = (c.pad_mode == PAD_REFLECT) ? 0 : c.pad;
  // class a covers padded coords hp == a (mod stride): hp = stride*u + a ; tensor row = hp - off
  // first u such that stride*u + a - off >= 0
  const int s = c.stride;
  auto first_u = [&](int a) { int v = off - a; return v <= 0 ? 0 : (v + s - 1) / s; };
  auto count_u = [&](int a, int extent) {   // rows r = s*u + a - off in [0, extent)
    int u0 = first_u(a);
    int r0 = s * u0 + a - off;
    return r0 >= extent ? 0 : (extent - 1 - r0) / s + 1;
  };
  const int uy0 = first_u(ay), ux0 = first_u(ax);
  g.N = c.N; g.Hs = Ho; g.Ws = Wo; g.Cs = CoutS; g.up = 0; g.Clog = c.Cout;
  g.Ho = count_u(ay, OHt); g.Wo = count_u(ax, OWt);
  // source row ho = (hp - ky)/s = u - j with hp = s*u + a, ky = a + s*j ; u = oy + u0
  g.sh = g.sw = 1; g.by0 = uy0; g.bx0 = ux0;
  g.th = dgrad_taps(c.kh, s, ay); g.tw = dgrad_taps(c.kw, s, ax); g.ys = g.xs = -1;
  g.pad_mode = PAD_ZERO;
  g.OH = OHt; g.OW = OWt; g.oys = g.oxs = s;
  g.oy0 = s * uy0 + ay - off; g.ox0 = s * ux0 + ax - off;
  if (g.Ho <= 0 || g.Wo <= 0 || g.th <= 0 || g.tw <= 0) { g.Ho = g.Wo = 0; g.th = g.tw = 1; }
  if (g.Ho == 0) { g.Ho = 0; }
  // finish_desc needs non-zero divisors
  int Ho_keep = g.Ho, Wo_keep = g.Wo;
  if (g.Ho == 0) g.Ho = 1;
  if (g.Wo == 0) g.Wo = 1;
  finish_desc(g);
  g.wK = g.K; g.wtw = g.tw; g.wty0 = g.wtx0 = 0;
  if (Ho_keep == 0 || Wo_keep == 0) g.M = 0;
  return g;
}

// The same gather restricted to the output sub-rectangle rows [ry0, ry0+rh) x cols [rx0, rx0+rw) of g's output grid
// (output placement follows).  Used to compute only the reflect ring of a dgrad frame.
inline GatherDesc sub_rect_desc(const GatherDesc& g, int ry0, int rh, int rx0, int rw) {
  GatherDesc r = g;
  r.Ho = rh; r.Wo = rw;
  r.by0 = g.by0 + ry0 * g.sh; r.bx0 = g.bx0 + rx0 * g.sw;
  r.oy0 = g.oy0 + ry0 * g.oys; r.ox0 = g.ox0 + rx0 * g.oxs;
  finish_desc(r);
  return r;
}

// The same gather restricted to taps [ty0, ty0+nty) x [tx0, tx0+ntx) of g's tap grid (weights keep their packed layout:
// the descriptor remembers where its taps sit in the full grid).  Reflect-ring rectangles read outside dY for every
// other tap row / column, so their GEMMs run on the live third of K only.
inline GatherDesc sub_taps_desc(const GatherDesc& g, int ty0, int nty, int tx0, int ntx) {
  GatherDesc r = g;
  r.th = nty; r.tw = ntx;
  r.by0 = g.by0 + ty0 * g.ys; r.bx0 = g.bx0 + tx0 * g.xs;
  r.wty0 = g.wty0 + ty0; r.wtx0 = g.wtx0 + tx0;
  finish_desc(r);
  return r;
}

// ---------------------------------------------------------------------------------------------
// The 2-pixel frame ("ring") of an H x W image: the pixels whose SPADE gamma/beta class is not the interior one when
// the label map is constant (normalization.py:24-37: two zero-padded 3x3 convs reach 2 pixels in).  The fused
// SPADE -> conv path keeps their normalised values in a compact side tensor [N][ring_pixels][C]:
//   rows 0, 1 (W pixels each) | rows H-2, H-1 | for rows 2 .. H-3: columns 0, 1, W-2, W-1
// ---------------------------------------------------------------------------------------------
DEI2I_HD int ring_pixels(int H, int W) { return 4 * W + 4 * (H - 4); }
DEI2I_HD bool ring_interior(int i, int extent) { return i >= 2 && i < extent - 2; }
DEI2I_HD int ring_index(int y, int x, int H, int W) {
  if (y < 2) return y * W + x;
  if (y >= H - 2) return (2 + y - (H - 2)) * W + x;
  return 4 * W + (y - 2) * 4 + (x < 2 ? x : 2 + x - (W - 2));
}
// inverse: ring pixel r -> (y, x)
DEI2I_HD void ring_coord(int r, int H, int W, int& y, int& x) {
  if (r < 2 * W) { y = r / W; x = r - y * W; return; }
  if (r < 4 * W) { const int q = r - 2 * W; y = H - 2 + q / W; x = q - (q / W) * W; return; }
  const int q = r - 4 * W;
  y = 2 + (q >> 2);
  const int k = q & 3;
  x = k < 2 ? k : W - 2 + (k - 2);
}

// Operand-path normalisation of a conv's INPUT (fused conv + norm + act, SURVEY.md Appendix B groups G3..G16):
//   z[n, y, x, c] = act(A[n*n_stride + c] * x[n, y>>up, x>>up, c] + B[...]),  act(v) = max(v,0) + slope*min(v,0)
// applied by the halo-resident kernels to the input halo in LDS (BatchNorm apply + LeakyReLU: one coefficient set for
// the batch, n_stride = 0; SPADE's InstanceNorm * (1+gamma) + beta + ReLU on its interior class: per image).  `ring`
// (optional) holds z of the logical image's 2-pixel frame, already normalised with its own class coefficients: halo
// pixels of the frame are fetched from it instead of from x and are not transformed.
struct ConvPro {
  const float* A;
  const float* B;
  int n_stride;
  float slope;
  const uint16_t* ring;
  int ring_pix;
};

// Backward reductions of the normalisation layer that PRODUCED a conv's input, taken in the epilogue of that conv's input-
// gradient launch (conv_halo16.hip, FOLD build) while the dz tile is still on chip -- instead of a streaming pass that reads
// dz and x again (reduce.hip: spade_bwd_partial_kernel / bn_bwd_partial_kernel, whose record layouts these are).
//   kind 1: z = relu(IN(x)*(1+gamma)+beta), gamma / beta from the (N,5,5,2C) class table (normalization.py:24-37):
//           partial[((n*chunks + chunk)*4 + q)*C + c], q = sum dxhat | sum dxhat*xhat | interior sum dgamma | interior sum dbeta
//   kind 2: z = act(a*y + b) (BatchNorm + activation, architecture.py:116-118): partial[(chunk*2 + q)*C + c], q = sum g | sum g*xhat
// chunk = the 16 x 32 tile (dei2i_conv2d_dgrad_norm_chunks records per image); x / y has the conv input's channel stride; `up`: it
// lives at half the resolution (SPADE behind a nearest x2 upsample).
struct EpiNorm {
  const uint16_t* x;
  const float* mean;
  const float* rstd;
  const uint16_t* gb;
  const float* a;
  const float* b;
  float* partial;
  int kind;
  int up;
  int act;
  int group_images;      // kind 2: a / b / mean / rstd are (groups, C), image n takes row n / group_images (0: one row for the batch)
};

// ---------------------------------------------------------------------------------------------
// Index maps shared by the device kernels and the CPU geometry check
// ---------------------------------------------------------------------------------------------
// packed forward weight element i of [Cout][kh*kw][CinS] -> OIHW source index, or -1 for channel padding
DEI2I_HD long long packed_fwd_src(long long i, int Cin, int CinS, int taps) {
  const int ci = (int)(i % CinS);
  const long long r = i / CinS;
  const int t = (int)(r % taps);
  const long long co = r / taps;
  return ci < Cin ? (co * Cin + ci) * taps + t : -1;
}

// packed dgrad weight element i of class (ay,ax): [Cin][th*tw][CoutS], entry = w[co][ci][ay+s*jy][ax+s*jx]
DEI2I_HD long long packed_dgrad_src(long long i, int Cout, int Cin, int CoutS, int kh, int kw, int s, int ay, int ax,
                                    int th, int tw) {
  const int co = (int)(i % CoutS);
  const long long r = i / CoutS;
  const int t = (int)(r % (th * tw));
  const long long ci = r / (th * tw);
  const int jy = t / tw, jx = t - jy * tw;
  return co < Cout ? (((long long)co * Cin + ci) * kh + (ay + s * jy)) * kw + (ax + s * jx) : -1;
}

// rows (or columns) of the dgrad output frame that fold onto physical row hs: every logical row of the 2^up cell,
// plus its reflected halo images.  Writes at most 6 entries, returns the count.
DEI2I_HD int fold_sources(int hs, int up, int Hl, int pad, int reflect, int* out) {
  const int off = reflect ? pad : 0;
  int n = 0;
  for (int a = 0; a < (1 << up); ++a) {
    const int hl = (hs << up) + a;
    out[n++] = hl + off;
    if (reflect) {
      if (hl >= 1 && hl <= pad) out[n++] = pad - hl;
      if (hl >= Hl - 1 - pad && hl <= Hl - 2) out[n++] = 2 * (Hl - 1) - hl + pad;
    }
  }
  return n;
}

}  // namespace dei2i
