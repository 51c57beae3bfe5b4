/*
 * conv_ref.c — plain-C restatement of the convolution arithmetic of the hot path (oracle, TEST INFRASTRUCTURE
 * ONLY; never linked into or called by the product).  It is the second, independent CPU implementation that
 * pins oracle/tfops.py: direct loops, double accumulation, no im2col, no library.
 *
 * Semantics restated (tf.keras, Keras-2; the reference's call sites are predict_model/v3plus.py:173-345,
 * scse.py:52-95, res34.py:33-156, hrnet.py:21):
 *   - layout NHWC activations, HWIO kernels, depthwise [kh][kw][C], transpose kernels [kh][kw][Cout][Cin];
 *   - padding='same': out = ceil(in/stride); total = max((out-1)*stride + (k-1)*dil + 1 - in, 0);
 *     pad_before = total/2 (the smaller half first), so 3x3 stride 2 on an even size pads (0,1);
 *   - Conv2DTranspose(stride 2, 'same') = input-gradient of the SAME conv mapping the 2x grid back:
 *     out[2*i + a - pad_before] += x[i] * w[a].
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC oracle/conv_ref.c -o oracle/libconvref.so   (see oracle/ref_c.py)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static void same_pad(int in, int k, int stride, int dil, int* out, int* before) {
  int o = (in + stride - 1) / stride;
  int total = (o - 1) * stride + (k - 1) * dil + 1 - in;
  if (total < 0) total = 0;
  *out = o;
  *before = total / 2;
}

/* y[N,Ho,Wo,Cout] = conv2d(x[N,H,W,Cin], w[KH,KW,Cin,Cout]) + bias, padding 'same' */
void ref_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin,
                    int Cout, int KH, int KW, int stride, int dil) {
  int Ho, Wo, pt, pl;
  same_pad(H, KH, stride, dil, &Ho, &pt);
  same_pad(W, KW, stride, dil, &Wo, &pl);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int oh = 0; oh < Ho; ++oh) {
      double* acc = (double*)malloc(sizeof(double) * Cout);
      for (int ow = 0; ow < Wo; ++ow) {
        for (int co = 0; co < Cout; ++co) acc[co] = bias ? bias[co] : 0.0;
        for (int kh = 0; kh < KH; ++kh) {
          int ih = oh * stride - pt + kh * dil;
          if (ih < 0 || ih >= H) continue;
          for (int kw = 0; kw < KW; ++kw) {
            int iw = ow * stride - pl + kw * dil;
            if (iw < 0 || iw >= W) continue;
            const float* xp = x + (((int64_t)n * H + ih) * W + iw) * Cin;
            const float* wp = w + ((int64_t)(kh * KW + kw) * Cin) * Cout;
            for (int ci = 0; ci < Cin; ++ci) {
              double xv = xp[ci];
              const float* wr = wp + (int64_t)ci * Cout;
              for (int co = 0; co < Cout; ++co) acc[co] += xv * wr[co];
            }
          }
        }
        float* yp = y + (((int64_t)n * Ho + oh) * Wo + ow) * Cout;
        for (int co = 0; co < Cout; ++co) yp[co] = (float)acc[co];
      }
      free(acc);
    }
}

/* dx[N,H,W,Cin] = input gradient of the conv above given dy[N,Ho,Wo,Cout] (scatter form, per image) */
void ref_conv2d_dgrad(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, int KH,
                      int KW, int stride, int dil) {
  int Ho, Wo, pt, pl;
  same_pad(H, KH, stride, dil, &Ho, &pt);
  same_pad(W, KW, stride, dil, &Wo, &pl);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    double* acc = (double*)calloc((size_t)H * W * Cin, sizeof(double));
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow) {
        const float* gp = dy + (((int64_t)n * Ho + oh) * Wo + ow) * Cout;
        for (int kh = 0; kh < KH; ++kh) {
          int ih = oh * stride - pt + kh * dil;
          if (ih < 0 || ih >= H) continue;
          for (int kw = 0; kw < KW; ++kw) {
            int iw = ow * stride - pl + kw * dil;
            if (iw < 0 || iw >= W) continue;
            double* ap = acc + ((int64_t)ih * W + iw) * Cin;
            const float* wp = w + ((int64_t)(kh * KW + kw) * Cin) * Cout;
            for (int ci = 0; ci < Cin; ++ci) {
              const float* wr = wp + (int64_t)ci * Cout;
              double s = 0.0;
              for (int co = 0; co < Cout; ++co) s += (double)gp[co] * wr[co];
              ap[ci] += s;
            }
          }
        }
      }
    float* dp = dx + (int64_t)n * H * W * Cin;
    for (int64_t i = 0; i < (int64_t)H * W * Cin; ++i) dp[i] = (float)acc[i];
    free(acc);
  }
}

/* dw[KH,KW,Cin,Cout] and db[Cout] (db may be NULL) */
void ref_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, int N, int H, int W, int Cin, int Cout,
                      int KH, int KW, int stride, int dil) {
  int Ho, Wo, pt, pl;
  same_pad(H, KH, stride, dil, &Ho, &pt);
  same_pad(W, KW, stride, dil, &Wo, &pl);
#pragma omp parallel for collapse(2) schedule(static)
  for (int kh = 0; kh < KH; ++kh)
    for (int kw = 0; kw < KW; ++kw) {
      double* acc = (double*)calloc((size_t)Cin * Cout, sizeof(double));
      for (int n = 0; n < N; ++n)
        for (int oh = 0; oh < Ho; ++oh) {
          int ih = oh * stride - pt + kh * dil;
          if (ih < 0 || ih >= H) continue;
          for (int ow = 0; ow < Wo; ++ow) {
            int iw = ow * stride - pl + kw * dil;
            if (iw < 0 || iw >= W) continue;
            const float* xp = x + (((int64_t)n * H + ih) * W + iw) * Cin;
            const float* gp = dy + (((int64_t)n * Ho + oh) * Wo + ow) * Cout;
            for (int ci = 0; ci < Cin; ++ci) {
              double xv = xp[ci];
              double* ar = acc + (int64_t)ci * Cout;
              for (int co = 0; co < Cout; ++co) ar[co] += xv * gp[co];
            }
          }
        }
      float* wp = dw + ((int64_t)(kh * KW + kw) * Cin) * Cout;
      for (int64_t i = 0; i < (int64_t)Cin * Cout; ++i) wp[i] = (float)acc[i];
      free(acc);
    }
  if (db) {
    for (int co = 0; co < Cout; ++co) {
      double s = 0.0;
      for (int64_t p = 0; p < (int64_t)N * Ho * Wo; ++p) s += dy[p * Cout + co];
      db[co] = (float)s;
    }
  }
}

/* depthwise 3x3-style conv, w[KH][KW][C], padding 'same' */
void ref_dwconv2d_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int KH, int KW,
                      int stride) {
  int Ho, Wo, pt, pl;
  same_pad(H, KH, stride, 1, &Ho, &pt);
  same_pad(W, KW, stride, 1, &Wo, &pl);
#pragma omp parallel for collapse(2) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow)
        for (int c = 0; c < C; ++c) {
          double s = 0.0;
          for (int kh = 0; kh < KH; ++kh) {
            int ih = oh * stride - pt + kh;
            if (ih < 0 || ih >= H) continue;
            for (int kw = 0; kw < KW; ++kw) {
              int iw = ow * stride - pl + kw;
              if (iw < 0 || iw >= W) continue;
              s += (double)x[(((int64_t)n * H + ih) * W + iw) * C + c] * w[(kh * KW + kw) * C + c];
            }
          }
          y[(((int64_t)n * Ho + oh) * Wo + ow) * C + c] = (float)s;
        }
}

/* Conv2DTranspose(stride, 'same'): y[N,H*s,W*s,Cout] from x[N,H,W,Cin], w[KH,KW,Cout,Cin] */
void ref_conv2d_transpose(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin,
                          int Cout, int KH, int KW, int stride) {
  int OH = H * stride, OW = W * stride, tmp, pt, pl;
  same_pad(OH, KH, stride, 1, &tmp, &pt);
  same_pad(OW, KW, stride, 1, &tmp, &pl);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < N; ++n) {
    double* acc = (double*)calloc((size_t)OH * OW * Cout, sizeof(double));
    for (int i = 0; i < H; ++i)
      for (int j = 0; j < W; ++j) {
        const float* xp = x + (((int64_t)n * H + i) * W + j) * Cin;
        for (int a = 0; a < KH; ++a) {
          int oh = i * stride + a - pt;
          if (oh < 0 || oh >= OH) continue;
          for (int b = 0; b < KW; ++b) {
            int ow = j * stride + b - pl;
            if (ow < 0 || ow >= OW) continue;
            double* ap = acc + ((int64_t)oh * OW + ow) * Cout;
            const float* wp = w + ((int64_t)(a * KW + b) * Cout) * Cin;
            for (int co = 0; co < Cout; ++co) {
              const float* wr = wp + (int64_t)co * Cin;
              double s = 0.0;
              for (int ci = 0; ci < Cin; ++ci) s += (double)xp[ci] * wr[ci];
              ap[co] += s;
            }
          }
        }
      }
    float* yp = y + (int64_t)n * OH * OW * Cout;
    for (int64_t p = 0; p < (int64_t)OH * OW; ++p)
      for (int co = 0; co < Cout; ++co) yp[p * Cout + co] = (float)(acc[p * Cout + co] + (bias ? bias[co] : 0.0));
    free(acc);
  }
}

/* MaxPooling2D(pool, stride, same|valid): padded cells are ignored (-inf) */
void ref_maxpool_fwd(const float* x, float* y, int N, int H, int W, int C, int k, int stride, int same) {
  int Ho, Wo, pt = 0, pl = 0;
  if (same) {
    same_pad(H, k, stride, 1, &Ho, &pt);
    same_pad(W, k, stride, 1, &Wo, &pl);
  } else {
    Ho = (H - k) / stride + 1;
    Wo = (W - k) / stride + 1;
  }
  for (int n = 0; n < N; ++n)
    for (int oh = 0; oh < Ho; ++oh)
      for (int ow = 0; ow < Wo; ++ow)
        for (int c = 0; c < C; ++c) {
          float m = -3.402823466e+38f;
          for (int a = 0; a < k; ++a) {
            int ih = oh * stride - pt + a;
            if (ih < 0 || ih >= H) continue;
            for (int b = 0; b < k; ++b) {
              int iw = ow * stride - pl + b;
              if (iw < 0 || iw >= W) continue;
              float v = x[(((int64_t)n * H + ih) * W + iw) * C + c];
              if (v > m) m = v;
            }
          }
          y[(((int64_t)n * Ho + oh) * Wo + ow) * C + c] = m;
        }
}
