// TEST INFRASTRUCTURE: executes the product's gather descriptors (de-i2i-gan_amd/csrc/geom.h) with plain CPU loops so
// the conv / dgrad / fold / weight-packing GEOMETRY can be checked against torch on a machine without a GPU.
// Built with g++ by tests/test_geometry_cpu.py; never linked into libdei2i_hip.so.
#include <cstring>
#include <vector>

#include "../../de-i2i-gan_amd/csrc/geom.h"

using namespace dei2i;

static ConvShape shape_of(const int* p) {
  ConvShape s;
  s.N = p[0]; s.H = p[1]; s.W = p[2]; s.Cin = p[3]; s.Cout = p[4];
  s.kh = p[7]; s.kw = p[8]; s.stride = p[9]; s.pad = p[10]; s.pad_mode = p[11]; s.up = p[12];
  return s;
}

static void run_gather_gemm(const GatherDesc& g, const float* src, const float* wgt, int wrows, float* out, int ldc) {
  for (int m = 0; m < g.M; ++m) {
    int n, oy, ox;
    decode_m(g, m, n, oy, ox);
    const long long op = out_pixel(g, n, oy, ox);
    for (int r = 0; r < wrows; ++r) {
      double acc = 0.0;
      for (int k = 0; k < g.K; ++k) {
        int tap, ci;
        decode_k(g, k, tap, ci);
        const int pix = src_pixel(g, n, oy, ox, tap);
        const int ty = tap / g.tw, tx = tap - ty * g.tw;
        if (pix >= 0) acc += (double)src[(long long)pix * g.Cs + ci] * (double)wgt[(long long)r * g.wK + weight_tap_offset(g, ty, tx) + ci];
      }
      out[op * ldc + r] = (float)acc;
    }
  }
}

extern "C" {

// p = {N,H,W,Cin,Cout,CinS,CoutS,kh,kw,stride,pad,pad_mode,up}
void hc_out_shape(const int* p, int* dims) {
  ConvShape s = shape_of(p);
  dims[0] = conv_out_dim(s.H << s.up, s.kh, s.stride, s.pad);
  dims[1] = conv_out_dim(s.W << s.up, s.kw, s.stride, s.pad);
  dims[2] = dgrad_out_h(s);
  dims[3] = dgrad_out_w(s);
}

void hc_pack_fwd(const int* p, const float* w, float* packed) {
  const int Cin = p[3], Cout = p[4], CinS = p[5], taps = p[7] * p[8];
  const long long total = (long long)Cout * taps * CinS;
  for (long long i = 0; i < total; ++i) {
    const long long si = packed_fwd_src(i, Cin, CinS, taps);
    packed[i] = si >= 0 ? w[si] : 0.f;
  }
}

long long hc_pack_dgrad(const int* p, const float* w, float* packed) {
  const int Cin = p[3], Cout = p[4], CoutS = p[6], kh = p[7], kw = p[8], s = p[9];
  long long off = 0;
  for (int ay = 0; ay < s; ++ay)
    for (int ax = 0; ax < s; ++ax) {
      const int th = dgrad_taps(kh, s, ay), tw = dgrad_taps(kw, s, ax);
      const long long total = (long long)Cin * th * tw * CoutS;
      for (long long i = 0; i < total; ++i) {
        const long long si = packed_dgrad_src(i, Cout, Cin, CoutS, kh, kw, s, ay, ax, th, tw);
        packed[off + i] = si >= 0 ? w[si] : 0.f;
      }
      off += total;
    }
  return off;
}

void hc_conv_fwd(const int* p, const float* x, const float* wpacked, float* y) {
  GatherDesc g = make_fwd_desc(shape_of(p), p[5]);
  run_gather_gemm(g, x, wpacked, p[4], y, p[6]);
}

void hc_conv_dgrad(const int* p, const float* dy, const float* wd_packed, float* dx_ext) {
  ConvShape s = shape_of(p);
  long long off = 0;
  for (int ay = 0; ay < s.stride; ++ay)
    for (int ax = 0; ax < s.stride; ++ax) {
      GatherDesc g = make_dgrad_desc(s, p[6], ay, ax);
      const long long welems = (long long)s.Cin * g.th * g.tw * p[6];
      if (g.M > 0) run_gather_gemm(g, dy, wd_packed + off, s.Cin, dx_ext, p[5]);
      off += welems;
    }
}

// wgrad through the forward descriptor: dw[co][k] = sum_m dy[m][co] * gather(x)[m][k]
void hc_conv_wgrad(const int* p, const float* x, const float* dy, float* dw_packed) {
  GatherDesc g = make_fwd_desc(shape_of(p), p[5]);
  const int Cout = p[4], CoutS = p[6];
  std::vector<double> acc((size_t)Cout * g.K, 0.0);
  for (int m = 0; m < g.M; ++m) {
    int n, oy, ox;
    decode_m(g, m, n, oy, ox);
    for (int k = 0; k < g.K; ++k) {
      int tap, ci;
      decode_k(g, k, tap, ci);
      const int pix = src_pixel(g, n, oy, ox, tap);
      if (pix < 0) continue;
      const double xv = x[(long long)pix * g.Cs + ci];
      for (int co = 0; co < Cout; ++co) acc[(size_t)co * g.K + k] += xv * (double)dy[(long long)m * CoutS + co];
    }
  }
  for (size_t i = 0; i < acc.size(); ++i) dw_packed[i] = (float)acc[i];
}

void hc_fold(int N, int H, int W, int C, int pad, int reflect, int up, const float* ext, float* dx) {
  const int Hl = H << up, Wl = W << up, off = reflect ? pad : 0;
  const int OH = Hl + 2 * off, OW = Wl + 2 * off;
  for (int n = 0; n < N; ++n)
    for (int h = 0; h < H; ++h)
      for (int w = 0; w < W; ++w) {
        int ys[6], xs[6];
        const int ny = fold_sources(h, up, Hl, pad, reflect, ys), nx = fold_sources(w, up, Wl, pad, reflect, xs);
        for (int c = 0; c < C; ++c) {
          double acc = 0.0;
          for (int a = 0; a < ny; ++a)
            for (int b = 0; b < nx; ++b) acc += ext[(((long long)n * OH + ys[a]) * OW + xs[b]) * C + c];
          dx[(((long long)n * H + h) * W + w) * C + c] = (float)acc;
        }
      }
}

// Decomposed grad_input of a stride-1 reflect conv (dei2i_conv2d_dgrad_input, conv_api.hip): interior = the
// zero-boundary dgrad on the input grid written straight into dx; reflect ring = four sub-rectangle descriptors of the
// padded frame written into ext (every other ext element stays untouched); border pixels add the ring's images.
// Returns the number of ext elements the ring descriptors wrote (must be the ring, nothing else).
long long hc_conv_dgrad_decomposed(const int* p, const float* dy, const float* wd, float* ext, float* dx) {
  ConvShape s = shape_of(p);
  const int CinS = p[5], CoutS = p[6], pad = s.pad;
  const GatherDesc frame = make_dgrad_desc(s, CoutS, 0, 0);
  ConvShape z = s;
  z.pad_mode = PAD_ZERO;
  const GatherDesc interior = make_dgrad_desc(z, CoutS, 0, 0);
  run_gather_gemm(interior, dy, wd, CinS, dx, CinS);
  const int OH = s.H + 2 * pad, OW = s.W + 2 * pad;
  GatherDesc ring[4] = {sub_rect_desc(frame, 0, pad, 0, OW), sub_rect_desc(frame, s.H + pad, pad, 0, OW),
                        sub_rect_desc(frame, pad, s.H, 0, pad), sub_rect_desc(frame, pad, s.H, s.W + pad, pad)};
  if (s.kh == 2 * pad + 1 && s.kw == 2 * pad + 1) {           // live taps only, as dei2i_conv2d_dgrad_input does
    ring[0] = sub_taps_desc(ring[0], 0, pad, 0, s.kw);
    ring[1] = sub_taps_desc(ring[1], s.kh - pad, pad, 0, s.kw);
    ring[2] = sub_taps_desc(ring[2], 0, s.kh, 0, pad);
    ring[3] = sub_taps_desc(ring[3], 0, s.kh, s.kw - pad, pad);
  }
  long long written = 0;
  for (int k = 0; k < 4; ++k) {
    run_gather_gemm(ring[k], dy, wd, CinS, ext, CinS);
    written += (long long)ring[k].M * CinS;
  }
  // border fold, same pixel enumeration as fold_border_kernel
  const int per_img = 2 * pad * s.W + 2 * pad * (s.H - 2 * pad);
  for (int n = 0; n < s.N; ++n)
    for (int q = 0; q < per_img; ++q) {
      int h, w;
      if (q < 2 * pad * s.W) {
        const int k = q / s.W;
        w = q - k * s.W;
        h = k < pad ? 1 + k : s.H - 1 - pad + (k - pad);
      } else {
        const int q2 = q - 2 * pad * s.W;
        const int k = q2 % (2 * pad);
        const int hr = q2 / (2 * pad);
        w = k < pad ? 1 + k : s.W - 1 - pad + (k - pad);
        h = hr == 0 ? 0 : (hr <= s.H - 2 - 2 * pad ? pad + hr : s.H - 1);
      }
      int ys[6], xs[6];
      const int ny = fold_sources(h, 0, s.H, pad, 1, ys), nx = fold_sources(w, 0, s.W, pad, 1, xs);
      for (int c = 0; c < CinS; ++c) {
        double acc = dx[(((long long)n * s.H + h) * s.W + w) * CinS + c];
        for (int a = 0; a < ny; ++a)
          for (int b = 0; b < nx; ++b)
            if (a != 0 || b != 0) acc += ext[(((long long)n * OH + ys[a]) * OW + xs[b]) * CinS + c];
        dx[(((long long)n * s.H + h) * s.W + w) * CinS + c] = (float)acc;
      }
    }
  return written;
}

// The same decomposition for the 4x4 stride-2 pad-1 reflect convs (dei2i_conv2d_dgrad_input, conv_api.hip): per parity class the
// zero-boundary dgrad straight into dx; frame rows 0 / H+1 and columns 0 / W+1 as one-pixel rectangles of the classes of their
// parity into ext; border fold.  Returns the number of ext elements written (must be the ring, nothing else, each once).
long long hc_conv_dgrad_decomposed_s2(const int* p, const float* dy, const float* wd, float* ext, float* dx) {
  ConvShape s = shape_of(p);
  const int CinS = p[5], CoutS = p[6];
  GatherDesc fr[2][2];
  long long woff[2][2], off = 0;
  for (int ay = 0; ay < 2; ++ay)
    for (int ax = 0; ax < 2; ++ax) {
      fr[ay][ax] = make_dgrad_desc(s, CoutS, ay, ax);
      woff[ay][ax] = off;
      off += (long long)s.Cin * dgrad_taps(s.kh, 2, ay) * dgrad_taps(s.kw, 2, ax) * CoutS;
    }
  ConvShape z = s;
  z.pad_mode = PAD_ZERO;
  for (int ay = 0; ay < 2; ++ay)
    for (int ax = 0; ax < 2; ++ax) {
      const GatherDesc g = make_dgrad_desc(z, CoutS, ay, ax);
      if (g.M > 0) run_gather_gemm(g, dy, wd + woff[ay][ax], s.Cin, dx, CinS);
    }
  const GatherDesc rects[8] = {sub_rect_desc(fr[0][0], 0, 1, 0, fr[0][0].Wo), sub_rect_desc(fr[0][1], 0, 1, 0, fr[0][1].Wo),
                               sub_rect_desc(fr[1][0], fr[1][0].Ho - 1, 1, 0, fr[1][0].Wo),
                               sub_rect_desc(fr[1][1], fr[1][1].Ho - 1, 1, 0, fr[1][1].Wo),
                               sub_rect_desc(fr[0][0], 1, fr[0][0].Ho - 1, 0, 1), sub_rect_desc(fr[1][0], 0, fr[1][0].Ho - 1, 0, 1),
                               sub_rect_desc(fr[0][1], 1, fr[0][1].Ho - 1, fr[0][1].Wo - 1, 1),
                               sub_rect_desc(fr[1][1], 0, fr[1][1].Ho - 1, fr[1][1].Wo - 1, 1)};
  const long long woffs[8] = {woff[0][0], woff[0][1], woff[1][0], woff[1][1], woff[0][0], woff[1][0], woff[0][1], woff[1][1]};
  long long written = 0;
  for (int k = 0; k < 8; ++k) {
    run_gather_gemm(rects[k], dy, wd + woffs[k], s.Cin, ext, CinS);
    written += (long long)rects[k].M * CinS;
  }
  const int OH = s.H + 2, OW = s.W + 2, per_img = 2 * s.W + 2 * (s.H - 2);
  for (int n = 0; n < s.N; ++n)
    for (int q = 0; q < per_img; ++q) {
      int h, w;
      if (q < 2 * s.W) {
        const int k = q / s.W;
        w = q - k * s.W;
        h = k < 1 ? 1 + k : s.H - 2 + (k - 1);
      } else {
        const int q2 = q - 2 * s.W;
        const int k = q2 % 2;
        const int hr = q2 / 2;
        w = k < 1 ? 1 + k : s.W - 2 + (k - 1);
        h = hr == 0 ? 0 : (hr <= s.H - 4 ? 1 + hr : s.H - 1);
      }
      int ys[6], xs[6];
      const int ny = fold_sources(h, 0, s.H, 1, 1, ys), nx = fold_sources(w, 0, s.W, 1, 1, xs);
      for (int c = 0; c < CinS; ++c) {
        double acc = dx[(((long long)n * s.H + h) * s.W + w) * CinS + c];
        for (int a = 0; a < ny; ++a)
          for (int b = 0; b < nx; ++b)
            if (a != 0 || b != 0) acc += ext[(((long long)n * OH + ys[a]) * OW + xs[b]) * CinS + c];
        dx[(((long long)n * s.H + h) * s.W + w) * CinS + c] = (float)acc;
      }
    }
  return written;
}

int hc_fastdiv_selftest(void) {
  const unsigned ds[] = {1, 2, 3, 4, 5, 7, 8, 9, 16, 25, 49, 64, 100, 147, 255, 256, 1000, 4096, 65536, 1048576, 16777215};
  const unsigned ns[] = {0, 1, 2, 3, 7, 8, 63, 64, 65, 999, 1000, 1001, 65535, 65536, 1048575, 16777216, 2147483647u};
  for (unsigned d : ds) {
    FastDiv f = make_fastdiv(d);
    for (unsigned n : ns)
      if (fd_div(n, f) != n / d) return 1;
    for (unsigned n = 0; n < 5000; ++n)
      if (fd_div(n * 7919u % 2147483647u, f) != (n * 7919u % 2147483647u) / d) return 2;
  }
  return 0;
}

}  // extern "C"
