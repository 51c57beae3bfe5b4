// Loss (binary_crossentropy / focal_loss / edge_focal_loss of train_model/DeepLabv3plus.py:490-527), the
// shared confusion count behind PA / IoU / MIoU / F1_score (:530-623), Keras-2 Adam (:835), and the inference
// tail of predict.py:110-114 and model_fuse.py:315,323.  All single-pass HBM-bound kernels; reductions are
// block-tree + fixed-order second stage (no float atomics => reproducible loss values).
#include "sg_reduce.h"

namespace {

constexpr float K_EPS = 1e-7f;  // tf.keras.backend.epsilon()

__device__ __forceinline__ void loss_coeffs(int kind, int y_cols, const float* __restrict__ yt, float& a0, float& a1) {
  const float y0 = yt[0], y1 = yt[1];
  if (kind == SG_LOSS_CE2) {
    a0 = y0; a1 = y1;
  } else if (kind == SG_LOSS_FOCAL) {
    a0 = 0.5f * y0; a1 = 0.5f * y1;
  } else {
    const float w0 = y_cols >= 4 ? yt[2] : 1.f, w1 = y_cols >= 4 ? yt[3] : 1.f;
    a0 = 0.35f * w0 * y0; a1 = 0.65f * w1 * y1;
  }
}

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) t = sm[0] + sm[1] + sm[2] + sm[3];
  return t;  // valid in thread 0
}

__global__ __launch_bounds__(256) void loss_fwd_kernel(int kind, int64_t rows, int y_cols, const float* __restrict__ p,
                                                       const float* __restrict__ yt, float* __restrict__ part) {
  __shared__ float sm[4];
  float acc = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
    const float2 pv = *reinterpret_cast<const float2*>(p + 2 * i);
    float a0, a1;
    loss_coeffs(kind, y_cols, yt + i * y_cols, a0, a1);
    const float f0 = kind == SG_LOSS_CE2 ? 1.f : (1.f - pv.x) * (1.f - pv.x);
    const float f1 = kind == SG_LOSS_CE2 ? 1.f : (1.f - pv.y) * (1.f - pv.y);
    acc += a0 * f0 * logf(pv.x + K_EPS) + a1 * f1 * logf(pv.y + K_EPS);
  }
  const float t = block_sum_256(acc, sm);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ void loss_final_kernel(const float* __restrict__ part, int nparts, int64_t rows, float* __restrict__ out) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += (double)part[i];
    out[0] = (float)(-s / (double)rows);
  }
}

// dL/dp_c = -(scale/rows) * a_c * d/dp [ f(p) log(p+eps) ];  f = (1-p)^2 -> -2(1-p) log(p+eps) + (1-p)^2/(p+eps)
__global__ void loss_bwd_kernel(int kind, int64_t rows, int y_cols, const float* __restrict__ p,
                                const float* __restrict__ yt, float* __restrict__ dp, float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float k = -scale / (float)rows;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
    const float2 pv = *reinterpret_cast<const float2*>(p + 2 * i);
    float a0, a1;
    loss_coeffs(kind, y_cols, yt + i * y_cols, a0, a1);
    float g0, g1;
    if (kind == SG_LOSS_CE2) {
      g0 = 1.f / (pv.x + K_EPS);
      g1 = 1.f / (pv.y + K_EPS);
    } else {
      const float q0 = 1.f - pv.x, q1 = 1.f - pv.y;
      g0 = -2.f * q0 * logf(pv.x + K_EPS) + q0 * q0 / (pv.x + K_EPS);
      g1 = -2.f * q1 * logf(pv.y + K_EPS) + q1 * q1 / (pv.y + K_EPS);
    }
    float2 o;
    o.x = k * a0 * g0;
    o.y = k * a1 * g1;
    *reinterpret_cast<float2*>(dp + 2 * i) = o;
  }
}

// argmax ties -> class 0 (tf.argmax returns the lowest index): pred = p1 > p0 ; truth = y1 > y0
__global__ __launch_bounds__(256) void confusion_kernel(int64_t rows, int y_cols, const float* __restrict__ p,
                                                        const float* __restrict__ yt,
                                                        unsigned long long* __restrict__ out) {
  __shared__ unsigned int sm[4][4];
  unsigned int tp = 0, tn = 0, fp = 0, fn = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
    const float2 pv = *reinterpret_cast<const float2*>(p + 2 * i);
    const int pred = pv.y > pv.x, truth = yt[i * y_cols + 1] > yt[i * y_cols];
    tp += pred & truth;
    tn += (1 - pred) & (1 - truth);
    fp += pred & (1 - truth);
    fn += (1 - pred) & truth;
  }
  for (int off = 32; off > 0; off >>= 1) {
    tp += __shfl_xor(tp, off, 64);
    tn += __shfl_xor(tn, off, 64);
    fp += __shfl_xor(fp, off, 64);
    fn += __shfl_xor(fn, off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sm[wave][0] = tp; sm[wave][1] = tn; sm[wave][2] = fp; sm[wave][3] = fn;
  }
  __syncthreads();
  if (threadIdx.x < 4) {  // integer atomics: order-independent, exact
    const unsigned int t = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
    atomicAdd(out + threadIdx.x, (unsigned long long)t);
  }
}

// lr_dev != nullptr: the step's learning rate is read from device memory (a hipGraph-captured training step: the
// bias-corrected rate changes every step, the captured kernel arguments do not)
__global__ void adam_kernel(int64_t n, float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ g, float lr_arg, const float* __restrict__ lr_dev, float b1, float b2,
                            float eps, float gs) {
  const float lr_t = lr_dev ? lr_dev[0] : lr_arg;
  const int64_t nv = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    f32x4 wv = reinterpret_cast<f32x4*>(w)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
    const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gv[k] * gs;
      mv[k] = b1 * mv[k] + (1.f - b1) * gk;
      vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
      wv[k] -= lr_t * mv[k] / (sqrtf(vv[k]) + eps);
    }
    reinterpret_cast<f32x4*>(w)[i] = wv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
  }
  // tail
  const int64_t i = (nv << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && i < n) {
    const float gk = g[i] * gs;
    const float mk = b1 * m[i] + (1.f - b1) * gk, vk = b2 * v[i] + (1.f - b2) * gk * gk;
    m[i] = mk;
    v[i] = vk;
    w[i] -= lr_t * mk / (sqrtf(vk) + eps);
  }
}

__global__ void argmax_acc_kernel(const float* __restrict__ p, int TH, int TW, signed char* __restrict__ canvas, int CH,
                                  int CW, int y0, int x0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= TH * TW) return;
  const int r = i / TW, c = i - r * TW;
  const int yy = y0 + r, xx = x0 + c;
  if (yy < 0 || yy >= CH || xx < 0 || xx >= CW) return;
  const float2 pv = *reinterpret_cast<const float2*>(p + 2 * (int64_t)i);
  canvas[(int64_t)yy * CW + xx] += (signed char)(pv.y > pv.x);
}

struct VoteArgs {
  const unsigned char* m[8];
  int n;
};

__global__ void vote_kernel(const VoteArgs a, int64_t n, int k, unsigned char* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int s = 0;
    for (int j = 0; j < a.n; ++j) s += a.m[j][i] / 255;
    out[i] = s >= k ? 255 : 0;
  }
}

// train_data_gen label channels (train_model/DeepLabv3plus.py:70-100): one-hot by integer truncation (only
// label == 1.0 is class 1), 5 x erode / dilate with a 3x3 kernel == (2R+1)x(2R+1) min / max box with R = 5 (out-of-image
// cells are ignored, as cv.erode / cv.dilate's default border does), p_edge = 2 where label - erode == 1, f_edge = 2
// where dilate - label == 1; channel order (bg, fg, f_edge, p_edge).
//
// Separable form: a workgroup owns an EL_TH x EL_TW output tile of one image.  (1) the (EL_TH+2R) x (EL_TW+2R) label
// window goes to LDS once; (2) row pass: per window row and output column the min / max over the 2R+1 columns that lie
// inside the image (rows outside the image become +inf / -inf, the identities); (3) column pass over 2R+1 window rows.
// 2 (2R+1) = 22 LDS reads for the two passes and ~1.6 global loads per pixel, where the plain window walk
// (edge_labels_window_kernel below, kept for R > EL_RMAX) issues (2R+1)^2 = 121 global loads.
constexpr int EL_TH = 16, EL_TW = 64, EL_RMAX = 8;

__global__ __launch_bounds__(256) void edge_labels_sep_kernel(const float* __restrict__ lab, float* __restrict__ y, int H, int W,
                                                             int R, int tiles_x, int tiles_y) {
  __shared__ float win[(EL_TH + 2 * EL_RMAX) * (EL_TW + 2 * EL_RMAX)];
  __shared__ float rmin[(EL_TH + 2 * EL_RMAX) * EL_TW];
  __shared__ float rmax[(EL_TH + 2 * EL_RMAX) * EL_TW];
  const int t = threadIdx.x;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, n = blockIdx.x / (tiles_x * tiles_y);
  const int x0 = tx * EL_TW, y0 = ty * EL_TH;
  const int WW = EL_TW + 2 * R, WH = EL_TH + 2 * R;
  const float* img = lab + (int64_t)n * H * W;
  for (int i = t; i < WH * WW; i += 256) {
    const int r = i / WW, c = i - r * WW;
    const int gy = y0 - R + r, gx = x0 - R + c;
    win[i] = ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? img[(int64_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = t; i < WH * EL_TW; i += 256) {
    const int r = i / EL_TW, c = i - r * EL_TW;
    const int gy = y0 - R + r;
    float mn = INFINITY, mx = -INFINITY;
    if ((unsigned)gy < (unsigned)H) {
      for (int b = 0; b <= 2 * R; ++b) {
        const int gx = x0 + c - R + b;
        const float v = win[r * WW + c + b];
        const bool in = (unsigned)gx < (unsigned)W;
        mn = in ? fminf(mn, v) : mn;
        mx = in ? fmaxf(mx, v) : mx;
      }
    }
    rmin[i] = mn;
    rmax[i] = mx;
  }
  __syncthreads();
  for (int i = t; i < EL_TH * EL_TW; i += 256) {
    const int r = i / EL_TW, c = i - r * EL_TW;
    const int gy = y0 + r, gx = x0 + c;
    if (gy >= H || gx >= W) continue;
    float er = INFINITY, di = -INFINITY;
    for (int a = 0; a <= 2 * R; ++a) {
      er = fminf(er, rmin[(r + a) * EL_TW + c]);
      di = fmaxf(di, rmax[(r + a) * EL_TW + c]);
    }
    const float m = win[(r + R) * WW + c + R];
    const float fg = ((int)m == 1) ? 1.f : 0.f;
    float4 o;
    o.x = 1.f - fg;
    o.y = fg;
    o.z = (di - m == 1.f) ? 2.f : 1.f;
    o.w = (m - er == 1.f) ? 2.f : 1.f;
    *reinterpret_cast<float4*>(y + 4 * ((int64_t)n * H * W + (int64_t)gy * W + gx)) = o;
  }
}

__global__ void edge_labels_window_kernel(const float* __restrict__ lab, float* __restrict__ y, int N, int H, int W, int radius) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)N * H * W) return;
  const int w = (int)(i % W), h = (int)((i / W) % H);
  const int64_t base = i - (int64_t)h * W - w;
  const float m = lab[i];
  float er = m, di = m;
  for (int a = -radius; a <= radius; ++a) {
    const int hh = h + a;
    if ((unsigned)hh >= (unsigned)H) continue;
    for (int b = -radius; b <= radius; ++b) {
      const int ww = w + b;
      if ((unsigned)ww >= (unsigned)W) continue;
      const float v = lab[base + (int64_t)hh * W + ww];
      er = fminf(er, v);
      di = fmaxf(di, v);
    }
  }
  const float fg = ((int)m == 1) ? 1.f : 0.f;
  float4 o;
  o.x = 1.f - fg;
  o.y = fg;
  o.z = (di - m == 1.f) ? 2.f : 1.f;
  o.w = (m - er == 1.f) ? 2.f : 1.f;
  *reinterpret_cast<float4*>(y + 4 * i) = o;
}

// cv.resize(img, (OW, OH)) of 8-bit pixels, default INTER_LINEAR (decode_img / decode_lbel, DeepLabv3plus.py:35,45), in
// OpenCV's own fixed-point arithmetic (resize.cpp; restated independently in oracle/input_pipeline.py, bit-exact):
// 11-bit coefficients from fx = float((d + 0.5) * scale - 0.5), horizontal pass in int32, vertical pass
// ((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2.  The double-precision products are taken with explicit
// round-to-nearest intrinsics so that no fused multiply-add changes the last bit of fx.  area2 = 1: exact 2x downscale,
// which resize() routes to the fast INTER_AREA ((a + b + c + d + 2) >> 2).
__device__ __forceinline__ void cv_linear_coeff(int d, double scale, int n_in, int& s, int& c0, int& c1, bool clamp_index) {
  float f = __double2float_rn(__dadd_rn(__dmul_rn((double)d + 0.5, scale), -0.5));
  s = (int)floorf(f);
  f = f - (float)s;
  if (clamp_index) {  // columns: the index is clamped AND the weight reset; rows keep their weights (clamped at the read)
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n_in - 1) { s = n_in - 1; f = 0.f; }
  }
  c0 = __float2int_rn((1.f - f) * 2048.f);
  c1 = __float2int_rn(f * 2048.f);
}

__global__ void resize_linear_u8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int N, int H, int W,
                                        int C, int OH, int OW, double sx_scale, double sy_scale, int area2) {
  const int64_t total = (int64_t)N * OH * OW * C;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C);
    const int dx = (int)((i / C) % OW);
    const int dy = (int)((i / ((int64_t)C * OW)) % OH);
    const int n = (int)(i / ((int64_t)C * OW * OH));
    const unsigned char* img = src + (int64_t)n * H * W * C;
    if (area2) {
      const unsigned char* q = img + ((int64_t)(2 * dy) * W + 2 * dx) * C + c;
      dst[i] = (unsigned char)((q[0] + q[C] + q[(int64_t)W * C] + q[(int64_t)W * C + C] + 2) >> 2);
      continue;
    }
    int sx, a0, a1, sy, b0, b1;
    cv_linear_coeff(dx, sx_scale, W, sx, a0, a1, true);
    cv_linear_coeff(dy, sy_scale, H, sy, b0, b1, false);
    const int sx1 = min(sx + 1, W - 1);
    const int y0 = min(max(sy, 0), H - 1), y1 = min(max(sy + 1, 0), H - 1);
    const int d0 = img[((int64_t)y0 * W + sx) * C + c] * a0 + img[((int64_t)y0 * W + sx1) * C + c] * a1;
    const int d1 = img[((int64_t)y1 * W + sx) * C + c] * a0 + img[((int64_t)y1 * W + sx1) * C + c] * a1;
    int v = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    dst[i] = (unsigned char)v;
  }
}

inline int loss_parts(int64_t rows) {
  int64_t b = sg_cdiv(rows, 256 * 4);
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {

size_t sg_loss_ws_bytes(const sg_ctx*, int64_t rows) { return (size_t)loss_parts(rows) * sizeof(float) + 256; }

int sg_loss_fwd(sg_ctx* ctx, void* stream, int kind, int64_t rows, int y_cols, const void* p, const void* y_true,
                void* loss_out, void* ws, size_t ws_bytes) {
  SG_CHECK_ARG(ctx && p && y_true && loss_out && rows > 0, "sg_loss_fwd: bad argument");
  SG_CHECK_ARG(kind >= SG_LOSS_CE2 && kind <= SG_LOSS_EDGE_FOCAL, "sg_loss_fwd: unknown loss %d", kind);
  SG_CHECK_ARG(y_cols == 2 || y_cols == 4, "sg_loss_fwd: y_true must have 2 or 4 columns");
  SG_CHECK_ARG(kind != SG_LOSS_EDGE_FOCAL || y_cols == 4, "sg_loss_fwd: edge_focal_loss needs y_true[...,4]");
  const int parts = loss_parts(rows);
  if (!ws || ws_bytes < (size_t)parts * sizeof(float)) {
    sg_set_error("sg_loss_fwd: workspace %zu < %zu", ws_bytes, (size_t)parts * sizeof(float));
    return SG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(parts), dim3(256), 0, st, kind, rows, y_cols, (const float*)p,
                     (const float*)y_true, (float*)ws);
  SG_LAUNCH_CHECK("loss_fwd_kernel");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, parts, rows, (float*)loss_out);
  SG_LAUNCH_CHECK("loss_final_kernel");
  return 0;
}

int sg_loss_bwd(sg_ctx* ctx, void* stream, int kind, int64_t rows, int y_cols, const void* p, const void* y_true,
                void* dp, float grad_scale) {
  SG_CHECK_ARG(ctx && p && y_true && dp && rows > 0, "sg_loss_bwd: bad argument");
  SG_CHECK_ARG(kind >= SG_LOSS_CE2 && kind <= SG_LOSS_EDGE_FOCAL, "sg_loss_bwd: unknown loss %d", kind);
  SG_CHECK_ARG(y_cols == 2 || y_cols == 4, "sg_loss_bwd: y_true must have 2 or 4 columns");
  SG_CHECK_ARG(kind != SG_LOSS_EDGE_FOCAL || y_cols == 4, "sg_loss_bwd: edge_focal_loss needs y_true[...,4]");
  int64_t blocks = sg_cdiv(rows, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, kind, rows, y_cols,
                     (const float*)p, (const float*)y_true, (float*)dp, grad_scale);
  SG_LAUNCH_CHECK("loss_bwd_kernel");
  return 0;
}

int sg_confusion_counts(sg_ctx* ctx, void* stream, int64_t rows, int y_cols, const void* p, const void* y_true,
                        void* out_i64x4) {
  SG_CHECK_ARG(ctx && p && y_true && out_i64x4 && rows > 0, "sg_confusion_counts: bad argument");
  SG_CHECK_ARG(y_cols == 2 || y_cols == 4, "sg_confusion_counts: y_true must have 2 or 4 columns");
  int64_t blocks = sg_cdiv(rows, 256 * 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, y_cols,
                     (const float*)p, (const float*)y_true, (unsigned long long*)out_i64x4);
  SG_LAUNCH_CHECK("confusion_kernel");
  return 0;
}

int sg_adam_step(sg_ctx* ctx, void* stream, int64_t n, void* w, void* m, void* v, const void* g, float lr_t,
                 float beta1, float beta2, float eps, float grad_scale) {
  SG_CHECK_ARG(ctx && w && m && v && g && n >= 0, "sg_adam_step: bad argument");
  SG_CHECK_ARG(sg_aligned16(w) && sg_aligned16(m) && sg_aligned16(v) && sg_aligned16(g),
               "sg_adam_step: arenas must be 16-byte aligned");
  if (n == 0) return 0;
  int64_t blocks = sg_cdiv(n / 4 + 1, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, (float*)w, (float*)m,
                     (float*)v, (const float*)g, lr_t, (const float*)nullptr, beta1, beta2, eps, grad_scale);
  SG_LAUNCH_CHECK("adam_kernel");
  return 0;
}

int sg_adam_step_lr(sg_ctx* ctx, void* stream, int64_t n, void* w, void* m, void* v, const void* g, const void* lr_t_dev,
                    float beta1, float beta2, float eps, float grad_scale) {
  SG_CHECK_ARG(ctx && w && m && v && g && lr_t_dev && n >= 0, "sg_adam_step_lr: bad argument");
  SG_CHECK_ARG(sg_aligned16(w) && sg_aligned16(m) && sg_aligned16(v) && sg_aligned16(g),
               "sg_adam_step_lr: arenas must be 16-byte aligned");
  if (n == 0) return 0;
  int64_t blocks = sg_cdiv(n / 4 + 1, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n, (float*)w, (float*)m,
                     (float*)v, (const float*)g, 0.f, (const float*)lr_t_dev, beta1, beta2, eps, grad_scale);
  SG_LAUNCH_CHECK("adam_kernel");
  return 0;
}

int sg_edge_labels(sg_ctx* ctx, void* stream, int N, int H, int W, int iterations, const void* label, void* y_true4) {
  SG_CHECK_ARG(ctx && label && y_true4 && N > 0 && H > 0 && W > 0 && iterations >= 0, "sg_edge_labels: bad argument");
  const int64_t n = (int64_t)N * H * W;
  if (iterations <= EL_RMAX) {  // separable row / column min-max through LDS
    const int tiles_x = (int)sg_cdiv(W, EL_TW), tiles_y = (int)sg_cdiv(H, EL_TH);
    const int64_t blocks = (int64_t)N * tiles_x * tiles_y;
    SG_CHECK_ARG(blocks < (1ll << 31), "sg_edge_labels: %lld tiles", (long long)blocks);
    hipLaunchKernelGGL(edge_labels_sep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)label,
                       (float*)y_true4, H, W, iterations, tiles_x, tiles_y);
    SG_LAUNCH_CHECK("edge_labels_sep_kernel");
    return 0;
  }
  hipLaunchKernelGGL(edge_labels_window_kernel, dim3((unsigned)sg_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)label, (float*)y_true4, N, H, W, iterations);
  SG_LAUNCH_CHECK("edge_labels_window_kernel");
  return 0;
}

int sg_resize_linear_u8(sg_ctx* ctx, void* stream, int N, int H, int W, int C, const void* src_u8, int OH, int OW, void* dst_u8) {
  SG_CHECK_ARG(ctx && src_u8 && dst_u8 && N > 0 && H > 0 && W > 0 && C > 0 && OH > 0 && OW > 0, "sg_resize_linear_u8: bad argument");
  SG_CHECK_ARG((int64_t)H * W * C < (1ll << 31), "sg_resize_linear_u8: one image of %d x %d x %d exceeds 2 GiB", H, W, C);
  const int64_t total = (int64_t)N * OH * OW * C;
  int64_t blocks = sg_cdiv(total, 256);
  if (blocks > 65536) blocks = 65536;
  const int area2 = (H == 2 * OH && W == 2 * OW) ? 1 : 0;
  hipLaunchKernelGGL(resize_linear_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned char*)src_u8,
                     (unsigned char*)dst_u8, N, H, W, C, OH, OW, (double)W / (double)OW, (double)H / (double)OH, area2);
  SG_LAUNCH_CHECK("resize_linear_u8_kernel");
  return 0;
}

int sg_argmax_accumulate_i8(sg_ctx* ctx, void* stream, const void* p, int TH, int TW, void* canvas, int CH, int CW,
                            int y0, int x0) {
  SG_CHECK_ARG(ctx && p && canvas && TH > 0 && TW > 0 && CH > 0 && CW > 0, "sg_argmax_accumulate_i8: bad argument");
  hipLaunchKernelGGL(argmax_acc_kernel, dim3((unsigned)sg_cdiv((int64_t)TH * TW, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const float*)p, TH, TW, (signed char*)canvas, CH, CW, y0, x0);
  SG_LAUNCH_CHECK("argmax_acc_kernel");
  return 0;
}

int sg_vote_ge(sg_ctx* ctx, void* stream, int nmasks, const void* const* masks, int64_t n, int k, void* out_u8) {
  SG_CHECK_ARG(ctx && masks && out_u8 && n >= 0, "sg_vote_ge: bad argument");
  SG_CHECK_ARG(nmasks >= 1 && nmasks <= 8, "sg_vote_ge: nmasks=%d outside [1,8]", nmasks);
  if (n == 0) return 0;
  VoteArgs a;
  for (int i = 0; i < 8; ++i) a.m[i] = i < nmasks ? (const unsigned char*)masks[i] : nullptr;
  for (int i = 0; i < nmasks; ++i) SG_CHECK_ARG(masks[i] != nullptr, "sg_vote_ge: null mask %d", i);
  a.n = nmasks;
  int64_t blocks = sg_cdiv(n, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(vote_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, n, k, (unsigned char*)out_u8);
  SG_LAUNCH_CHECK("vote_kernel");
  return 0;
}

}  // extern "C"
