// HBM-bound spatial kernels of the hot path (NHWC, channels innermost so every access is a coalesced run of
// 16-byte channel chunks): depthwise 3x3 (SeparableConv2D's first half) forward / dgrad / wgrad, max / average
// pooling, nearest up-sampling.
#include "sg_reduce.h"

namespace {

inline unsigned ew_blocks(int64_t total) {
  int64_t b = sg_cdiv(total, 256);
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename T>
struct DwParams {
  const T* __restrict__ x;       // forward input (or mask source for dgrad)
  const float* __restrict__ w;   // [KH][KW][C] (fp32 master weights whatever the activation storage)
  const T* __restrict__ dy;
  T* __restrict__ out;
  int N, H, W, C, Ho, Wo, KH, KW, stride, dil, pad_t, pad_l, x_ld, y_ld, pre_relu;
  FastDiv fd_cv, fd_w, fd_h;  // decomposition of the flat index: channel chunk, then width, then height
};

// y[n,oh,ow,c] = sum_taps relu?(x[n, oh*s - pt + kh*d, ow*s - pl + kw*d, c]) * w[kh,kw,c]
template <int V, typename T>
__global__ void dw_fwd_kernel(const DwParams<T> p) {
  const uint32_t cv = p.C / V;
  const uint32_t total = (uint32_t)((int64_t)p.N * p.Ho * p.Wo * cv), stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t pix, cc, row, ow, n, oh;
    fd_divmod(i, p.fd_cv, pix, cc);
    fd_divmod(pix, p.fd_w, row, ow);
    fd_divmod(row, p.fd_h, n, oh);
    const int c = (int)cc * V;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int kh = 0; kh < p.KH; ++kh) {
      const int ih = (int)oh * p.stride - p.pad_t + kh * p.dil;
      if ((unsigned)ih >= (unsigned)p.H) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        const int iw = (int)ow * p.stride - p.pad_l + kw * p.dil;
        if ((unsigned)iw >= (unsigned)p.W) continue;
        float xv[V], wv[V];
        ldv<V>(p.x + ((int64_t)(n * p.H + ih) * p.W + iw) * p.x_ld + c, xv);
        ldv<V>(p.w + (kh * p.KW + kw) * p.C + c, wv);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = fmaf(p.pre_relu ? fmaxf(xv[k], 0.f) : xv[k], wv[k], acc[k]);
      }
    }
    stv<V>(p.out + (int64_t)pix * p.y_ld + c, acc);
  }
}

// dx[n,ih,iw,c] = sum_taps dy[n,(ih+pt-kh*d)/s,(iw+pl-kw*d)/s,c] * w[kh,kw,c]   (* [x>0] if pre_relu)
template <int V, typename T>
__global__ void dw_dgrad_kernel(const DwParams<T> p) {
  const uint32_t cv = p.C / V;
  const uint32_t total = (uint32_t)((int64_t)p.N * p.H * p.W * cv), stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t pix, cc, row, iw, n, ih;
    fd_divmod(i, p.fd_cv, pix, cc);
    fd_divmod(pix, p.fd_w, row, iw);
    fd_divmod(row, p.fd_h, n, ih);
    const int c = (int)cc * V;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int kh = 0; kh < p.KH; ++kh) {
      int oh = (int)ih + p.pad_t - kh * p.dil;
      if (oh < 0 || oh % p.stride) continue;
      oh /= p.stride;
      if (oh >= p.Ho) continue;
      for (int kw = 0; kw < p.KW; ++kw) {
        int ow = (int)iw + p.pad_l - kw * p.dil;
        if (ow < 0 || ow % p.stride) continue;
        ow /= p.stride;
        if (ow >= p.Wo) continue;
        float gv[V], wv[V];
        ldv<V>(p.dy + ((int64_t)(n * p.Ho + oh) * p.Wo + ow) * p.y_ld + c, gv);
        ldv<V>(p.w + (kh * p.KW + kw) * p.C + c, wv);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = fmaf(gv[k], wv[k], acc[k]);
      }
    }
    if (p.pre_relu) {
      float xv[V];
      ldv<V>(p.x + (int64_t)pix * p.x_ld + c, xv);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] = xv[k] > 0.f ? acc[k] : 0.f;
    }
    stv<V>(p.out + (int64_t)pix * p.x_ld + c, acc);
  }
}

// dw[kh,kw,c] = sum_{n,oh,ow} relu?(x[n,ih,iw,c]) * dy[n,oh,ow,c]  — 9 outputs per channel
template <typename T>
struct DwWgradOp {
  static constexpr int NOUT = 9;
  const T* __restrict__ x;
  const T* __restrict__ dy;
  float* dw;
  int H, W, C, Ho, Wo, stride, dil, pad_t, pad_l, x_ld, y_ld, pre_relu;
  FastDiv fd_w, fd_h;
  template <int V>
  __device__ __forceinline__ void accum(int, int64_t r, int c, float (&acc)[9][V]) const {
    uint32_t row, ow, n, oh;
    fd_divmod((uint32_t)r, fd_w, row, ow);
    fd_divmod(row, fd_h, n, oh);
    float gv[V];
    ldv<V>(dy + r * y_ld + c, gv);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = (int)oh * stride - pad_t + kh * dil;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = (int)ow * stride - pad_l + kw * dil;
        // branch-free: a tap outside the image loads pixel (0, 0) of the image and is dropped by a select (nine loads in flight)
        const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
        float xv[V];
        ldv<V>(x + ((int64_t)(n * H + (ok ? ih : 0)) * W + (ok ? iw : 0)) * x_ld + c, xv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const float xr = pre_relu ? fmaxf(xv[k], 0.f) : xv[k];
          const float a = fmaf(xr, gv[k], acc[kh * 3 + kw][k]);
          acc[kh * 3 + kw][k] = ok ? a : acc[kh * 3 + kw][k];
        }
      }
    }
  }
  __device__ __forceinline__ void finalize(int, int c, const double (&s)[9]) const {
#pragma unroll
    for (int t = 0; t < 9; ++t) dw[t * C + c] = (float)s[t];
  }
};

// ---- stride-1 3x3 'same' fast paths: each thread owns one 16-byte channel chunk and a run of 4 output columns,
// so the 3 x 6 input window is loaded once (18 loads for 4 outputs instead of 36) and the 9 kernel taps live in
// registers.  Forward and dgrad are the same stencil (dgrad: kernel flipped, dy as input, optional [x > 0] mask).
template <typename T>
struct DwRunParams {
  const T* __restrict__ in;        // x (fwd) or dy (dgrad)
  const float* __restrict__ w;     // [3][3][C]
  const T* __restrict__ mask;      // dgrad with pre_relu: forward input, else null
  const T* res;                    // dgrad: a gradient already collected for the same tensor, added to the result (may be `out`)
  T* out;
  int N, H, W, C, in_ld, out_ld, mask_ld, relu_in, flip;
  // BN = true kernels: the input is the RAW output of the producing convolution and the preceding training-mode
  // BatchNormalization (+ ReLU) is applied as the window is loaded - fmaf((x - mean) * invstd, gamma, beta), the very
  // expression of bn_apply_kernel - so that normalised tensor is never written or read (sg_dwconv2d_fwd_bn)
  const float* __restrict__ bn_gamma;
  const float* __restrict__ bn_beta;
  const float* __restrict__ bn_mean;
  const float* __restrict__ bn_invstd;
  // SUMS = true kernels (dgrad only): `out` is the gradient of a training-mode BatchNormalization's OUTPUT, whose raw input is
  // bs_x.  Besides writing it the kernel sums, per channel, what that layer's backward needs - sum g and sum g * xhat with
  // g = out [masked by the layer's fused ReLU, recomputed from bs_x as sg_bn_train_bwd does] and xhat = (bs_x - mean) * invstd
  // - into bs_part[blockIdx.y][2][C]: the reduction pass of BatchNormalization's backward (two tensor reads) disappears for
  // one more read here (sg_dwconv2d_dgrad_bnsums)
  const T* __restrict__ bs_x = nullptr;
  const float* __restrict__ bs_mean = nullptr;
  const float* __restrict__ bs_invstd = nullptr;
  const float* __restrict__ bs_gamma = nullptr;
  const float* __restrict__ bs_beta = nullptr;
  float* bs_part = nullptr;
  int bs_ld = 0, bs_relu = 0;
  int lc;                          // lanes per run along the channels (set by launch_dw_run)
  int runs_per_row;                // W / 4
  int64_t nruns;                   // N * H * runs_per_row
  FastDiv fd_rpr, fd_h;
};

// RR output rows per run: the (RR + 2) x 6 input window is loaded once for RR x 4 outputs - 4.5 loads per output
// at RR = 1, 2.25 at RR = 4 (fd_h divides by H / RR then).
// RELU / MASK are compile-time and the window is branch-free (rows and columns outside the image load a valid address
// and are zeroed by a select): with the run-time `if (relu_in)` and the `continue` on the row test every load sat in its
// own basic block behind an s_waitcnt vmcnt(0) - 18..36 serialised memory latencies per run.
// (SUMS: two workgroups per SIMD-row instead of three - its 40 more live registers spilled 150 of them under the 168-register
// cap, 590 bytes of scratch per lane, and the kernel lost more than the reduction pass it replaces costs)
template <int RR, typename T, bool RELU, bool MASK, bool BN = false, bool SUMS = false>
__global__ __launch_bounds__(256, SUMS ? 2 : 3) void dw_s1_run_kernel(const DwRunParams<T> p) {
  // p.lc lanes (a power of two <= 64) cover the channel chunks of one run; with few channels (C = 64: 16 chunks)
  // a wave takes several runs instead of idling three quarters of its lanes
  const int lc = p.lc, rpb = 256 / lc;
  const int c_raw = (blockIdx.x * lc + (threadIdx.x & (lc - 1))) * 4;
  if constexpr (!SUMS) {
    if (c_raw >= p.C) return;
  }
  // SUMS: every thread reaches the workgroup's reduction below; a lane past the last channel chunk walks chunk 0 and drops
  // its results
  const bool live = c_raw < p.C;
  const int c = live ? c_raw : 0;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  f32x4 bmv, biv, bgm, bbt;
  if constexpr (SUMS) {
    bmv = *reinterpret_cast<const f32x4*>(p.bs_mean + c);
    biv = *reinterpret_cast<const f32x4*>(p.bs_invstd + c);
    bgm = *reinterpret_cast<const f32x4*>(p.bs_gamma + c);
    bbt = *reinterpret_cast<const f32x4*>(p.bs_beta + c);
  }
  f32x4 wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const f32x4*>(p.w + (p.flip ? 8 - t : t) * p.C + c);
  f32x4 gm, bt, mv, iv;
  if constexpr (BN) {
    gm = *reinterpret_cast<const f32x4*>(p.bn_gamma + c);
    bt = *reinterpret_cast<const f32x4*>(p.bn_beta + c);
    mv = *reinterpret_cast<const f32x4*>(p.bn_mean + c);
    iv = *reinterpret_cast<const f32x4*>(p.bn_invstd + c);
  }
  const int64_t stride = (int64_t)gridDim.y * rpb;
  for (int64_t run = (int64_t)blockIdx.y * rpb + threadIdx.x / lc; run < p.nruns; run += stride) {
    uint32_t rowi, q, n, ohb;
    fd_divmod((uint32_t)run, p.fd_rpr, rowi, q);
    fd_divmod(rowi, p.fd_h, n, ohb);
    const int ow0 = (int)q * 4, oh0 = (int)ohb * RR;
    const bool lok = ow0 > 0, rok = ow0 + 4 < p.W;   // the window's first / last column lies inside the image
    f32x4 acc[RR][4];
#pragma unroll
    for (int rr = 0; rr < RR; ++rr)
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[rr][k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < RR + 2; ++a) {
      const int ih = oh0 - 1 + a;
      const bool rowok = (a >= 1 && a <= RR) || (unsigned)ih < (unsigned)p.H;  // the RR middle rows always are
      const T* rowp = p.in + ((int64_t)(n * p.H + (rowok ? ih : oh0)) * p.W + ow0) * p.in_ld + c;
      f32x4 v[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const bool ok = rowok && (b == 0 ? lok : (b == 5 ? rok : true));
        const int db = (b == 0 && !lok) ? 0 : ((b == 5 && !rok) ? 3 : b - 1);   // a valid column when masked
        f32x4 t = ld4<T>(rowp + (int64_t)db * p.in_ld);
        if constexpr (BN) {
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = fmaf((t[e] - mv[e]) * iv[e], gm[e], bt[e]);
        }
        v[b] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};   // zero padding of the NORMALISED tensor
        if constexpr (RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[b][e] = fmaxf(v[b][e], 0.f);
        }
      }
      // input row a feeds output row rr = a - ta through kernel row ta (0..2)
#pragma unroll
      for (int ta = 0; ta < 3; ++ta) {
        const int rr = a - ta;
        if (rr < 0 || rr >= RR) continue;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc[rr][k] += v[k + b] * wt[ta * 3 + b];
      }
    }
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      const int64_t opix = ((int64_t)(n * p.H + oh0 + rr) * p.W + ow0);
      f32x4 m[4], bx[4];
      if constexpr (MASK) {
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = ld4<T>(p.mask + (opix + k) * p.mask_ld + c);
      }
      if constexpr (SUMS) {
#pragma unroll
        for (int k = 0; k < 4; ++k) bx[k] = ld4<T>(p.bs_x + (opix + k) * p.bs_ld + c);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f32x4 o = acc[rr][k];
        if constexpr (MASK) {
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = m[k][e] > 0.f ? o[e] : 0.f;
        }
        if (p.res) {   // uniform: the other consumer's gradient of this tensor rides along (sg_dwconv2d_dgrad_acc)
          const f32x4 rv = ld4<T>(p.res + (opix + k) * p.out_ld + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += rv[e];
        }
        if constexpr (SUMS) {
          // BnBwdOp's accumulation (norm.hip) on the complete gradient: the fused ReLU's mask from the forward's own fmaf
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xh = (bx[k][e] - bmv[e]) * biv[e];
            const bool on = !p.bs_relu || fmaf(xh, bgm[e], bbt[e]) > 0.f;
            const float g = on ? o[e] : 0.f;
            s1[e] += g;
            s2[e] = fmaf(g, xh, s2[e]);
          }
        }
        if (SUMS && !live) continue;
        st4<T>(p.out + (opix + k) * p.out_ld + c, o);
      }
    }
  }
  if constexpr (SUMS) {
    // the run slots of this workgroup (threads that share a channel chunk) are added in slot order; the partial row of this
    // workgroup goes to bs_part[blockIdx.y], the rows are added in fp64 by the finalize launch (fixed order: deterministic)
    __shared__ float red[256 * 8];
    float* mine = red + threadIdx.x * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) { mine[e] = s1[e]; mine[4 + e] = s2[e]; }
    __syncthreads();
    if ((int)threadIdx.x < lc && live) {
      f32x4 t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < rpb; ++j) {
        const float* q = red + (j * lc + threadIdx.x) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) { t1[e] += q[e]; t2[e] += q[4 + e]; }
      }
      float* row = p.bs_part + (int64_t)blockIdx.y * 2 * p.C + c;
      *reinterpret_cast<f32x4*>(row) = t1;
      *reinterpret_cast<f32x4*>(row + p.C) = t2;
    }
  }
}

// wgrad, same window: a "row" of the segmented reducer is a run of RR rows x 4 output pixels.  Branch-free like the
// stencil above (PRE compile-time, rows / columns outside the image masked by a select): 22 loads of a run in flight
// together instead of one at a time.
// BN: x is the raw output of the producing convolution; the training-mode BatchNormalization (+ ReLU, PRE) in front of this
// depthwise convolution is applied on load with bn_apply_kernel's expression (sg_dwconv2d_wgrad_bn), its four per-channel
// parameters loaded once per thread (the reducer's context hook).
template <int V>
struct DwBnCtx {
  float gm[V], bt[V], mv[V], iv[V];
};

template <int RR, typename T, bool PRE, bool BN = false>
struct DwWgradRunOp {
  static constexpr int NOUT = 9;
  const T* __restrict__ x;
  const T* __restrict__ dy;
  float* dw;
  int H, W, C, x_ld, y_ld;
  FastDiv fd_rpr, fd_h;
  const float* __restrict__ bn_gamma;
  const float* __restrict__ bn_beta;
  const float* __restrict__ bn_mean;
  const float* __restrict__ bn_invstd;
  template <int V>
  __device__ __forceinline__ DwBnCtx<V> begin(int, int c) const {
    DwBnCtx<V> k;
    if constexpr (BN) {
      ldv<V>(bn_gamma + c, k.gm);
      ldv<V>(bn_beta + c, k.bt);
      ldv<V>(bn_mean + c, k.mv);
      ldv<V>(bn_invstd + c, k.iv);
    }
    return k;
  }
  template <int V>
  __device__ __forceinline__ void accum(int, int64_t r, int c, float (&acc)[9][V], const DwBnCtx<V>& bn) const {
    if constexpr (V != 4) {
      return;  // the run path is only planned with 16-byte chunks (seg_plan vec_ok = true, C % 4 == 0)
    } else {
    uint32_t rowi, q, n, ohb;
    fd_divmod((uint32_t)r, fd_rpr, rowi, q);
    fd_divmod(rowi, fd_h, n, ohb);
    const int ow0 = (int)q * 4, oh0 = (int)ohb * RR;
    const bool lok = ow0 > 0, rok = ow0 + 4 < W;
    f32x4 g[RR][4];
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      const T* gp = dy + ((int64_t)(n * H + oh0 + rr) * W + ow0) * y_ld + c;
#pragma unroll
      for (int k = 0; k < 4; ++k) g[rr][k] = ld4<T>(gp + (int64_t)k * y_ld);
    }
#pragma unroll
    for (int a = 0; a < RR + 2; ++a) {
      const int ih = oh0 - 1 + a;
      const bool rowok = (a >= 1 && a <= RR) || (unsigned)ih < (unsigned)H;
      const T* rowp = x + ((int64_t)(n * H + (rowok ? ih : oh0)) * W + ow0) * x_ld + c;
      f32x4 v[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const bool ok = rowok && (b == 0 ? lok : (b == 5 ? rok : true));
        const int db = (b == 0 && !lok) ? 0 : ((b == 5 && !rok) ? 3 : b - 1);
        f32x4 t = ld4<T>(rowp + (int64_t)db * x_ld);
        if constexpr (BN) {
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = fmaf((t[e] - bn.mv[e]) * bn.iv[e], bn.gm[e], bn.bt[e]);
        }
        v[b] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PRE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[b][e] = fmaxf(v[b][e], 0.f);
        }
      }
#pragma unroll
      for (int ta = 0; ta < 3; ++ta) {
        const int rr = a - ta;  // input row a meets output row rr under kernel row ta
        if (rr < 0 || rr >= RR) continue;
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ta * 3 + b][e] = fmaf(v[k + b][e], g[rr][k][e], acc[ta * 3 + b][e]);
      }
    }
    }
  }
  __device__ __forceinline__ void finalize(int, int c, const double (&s)[9]) const {
#pragma unroll
    for (int t = 0; t < 9; ++t) dw[t * C + c] = (float)s[t];
  }
};

// wgrad as column strips (round 3; SG_DW_STRIP=0 restores the run reducer above).  The run form loads every x element 4.5
// times and every dy element once per run of 4 pixels (22 loads of 16 bytes for 144 FMAs per lane) and was bound by the
// load issue, not by HBM: 40 us + 8 us finalize for the 95 MB of the 32x32x728 layers (floor 20 us).  Here a lane owns 4
// channels of a strip of 4 columns x HS rows and walks DOWN it with the three x rows of the window in registers: one new
// x row (6 loads) and one dy row (4 loads) per 4 output pixels, both requested one row ahead of their use.  A workgroup is
// 16 channel chunks (256 contiguous bytes per pixel) x 16 strips; the 16 strip partials are added in fixed order through
// LDS, the S workgroup partials by seg_finalize_kernel in fp64 as before (part layout of seg_reduce_kernel, segment 0).
// BN: x is the raw output of the producing convolution and the training-mode BatchNormalization (+ ReLU = PRE) in front of
// this depthwise convolution is applied where a row is consumed, with bn_apply_kernel's expression (SG_BN_DEFER, as
// DwWgradRunOp<.., BN = true>); the zero padding then has to be put back by masks, since BN(0) != 0.
// (BN: held to two waves per SIMD - left to itself the variant takes a few registers more than 256, i.e. ONE wave per SIMD,
// and ran 110 us where the plain kernel takes 50 (profiles/r04_bench_kernel_stats_final.csv))
// OCC2 (SG_DW_STRIP_OCC2, A/B switch for the BN variants): two waves per SIMD with 12 - 62 spilled registers
template <typename T, bool PRE, bool BN, bool OCC2 = false>
__global__ __launch_bounds__(256, OCC2 ? 2 : 1) __attribute__((amdgpu_waves_per_eu(1, 2))) void dw_wgrad_strip_kernel(
    const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, const int H, const int W, const int C,
    const int x_ld, const int y_ld, const int HS, const int nstrips, const int S, const unsigned x_bytes, const unsigned y_bytes,
    const FastDiv fd_q, const FastDiv fd_hs, const float* __restrict__ bn_gamma, const float* __restrict__ bn_beta,
    const float* __restrict__ bn_mean, const float* __restrict__ bn_invstd) {
  constexpr int TX = 16, TY = 16, EB = (int)sizeof(T);
  constexpr unsigned OOB = 0x80000000u;  // beyond num_records: the hardware returns 0 (image border, no select, no branch)
  typedef typename std::conditional<EB == 4, u32x4_c, u32x2_c>::type raw_t;
  __shared__ float red[TY][TX][36];
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> 4;
  const int c = (blockIdx.x * TX + tx) * 4;
  const int z = blockIdx.y;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = zero;
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, (int)x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dy), 0, (int)y_bytes, 0x00020000);
  auto ldraw = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> raw_t {
    if constexpr (EB == 4) return __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    else return __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
  };
  // the registers a load fills are touched only where the row is consumed (widening, ReLU): anything done to them at the
  // load would wait for it
  auto widen = [&](const raw_t r, const bool relu) -> f32x4 {
    f32x4 o;
    if constexpr (EB == 4) {
      o = __builtin_bit_cast(f32x4, r);
    } else {
      o = (f32x4){__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                  __uint_as_float(r[1] & 0xffff0000u)};
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    return o;
  };
  f32x4 gm = zero, bt = zero, mv = zero, iv = zero;
  if constexpr (BN) {
    if (c < C) { gm = ld4<float>(bn_gamma + c); bt = ld4<float>(bn_beta + c); mv = ld4<float>(bn_mean + c); iv = ld4<float>(bn_invstd + c); }
  }
  if (c < C) {
    for (int s = z * TY + ty; s < nstrips; s += S * TY) {
      uint32_t rowi, q, n, hs;
      fd_divmod((uint32_t)s, fd_q, rowi, q);     // strips of one row band lie side by side: neighbours share their halo columns
      fd_divmod(rowi, fd_hs, n, hs);
      const int ow0 = (int)q * 4, h0 = (int)hs * HS;
      const int h1 = h0 + HS < H ? h0 + HS : H;
      const bool lok = ow0 > 0, rok = ow0 + 4 < W;
      const unsigned xpix0 = ((unsigned)n * H * W + ow0), xrow = (unsigned)W * x_ld * EB, grow = (unsigned)W * y_ld * EB;
      const unsigned xoff0 = (xpix0 * x_ld + c) * EB, goff0 = (xpix0 * y_ld + c) * EB;
      // Offsets are sums, never selects around a load (a select feeding a load became control flow, and with more than one
      // basic block per step the FMAs were sunk out of the steps altogether): a column outside the image adds 2^30, a row
      // outside it sets bit 31 - either way the offset is beyond num_records (< 2^30, dw_strip_ok) and the load returns 0.
      unsigned coff[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) coff[b] = (unsigned)((b - 1) * x_ld * EB);
      coff[0] = lok ? coff[0] : 0x40000000u;
      coff[5] = rok ? coff[5] : 0x40000000u;
      auto load_x = [&](int ih, raw_t (&v)[6]) {
        const unsigned flag = (unsigned)ih < (unsigned)H ? 0u : OOB;
        const unsigned ro = xoff0 + (unsigned)ih * xrow;
#pragma unroll
        for (int b = 0; b < 6; ++b) v[b] = ldraw(rsrc_x, (ro + coff[b]) | flag);
      };
      auto load_g = [&](int oh, raw_t (&g)[4]) {
        const unsigned flag = oh < h1 ? 0u : OOB;   // the row behind the strip is never used
        const unsigned ro = goff0 + (unsigned)oh * grow;
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = ldraw(rsrc_g, (ro + (unsigned)(k * y_ld * EB)) | flag);
      };
      auto rowmac = [&](const int ta, const raw_t (&vr)[6], const f32x4 (&g)[4], const int ih) {
        f32x4 v[6];
        if constexpr (BN) {
          const bool rowok = (unsigned)ih < (unsigned)H;
#pragma unroll
          for (int b = 0; b < 6; ++b) {
            f32x4 t = widen(vr[b], false);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              t[e] = fmaf((t[e] - mv[e]) * iv[e], gm[e], bt[e]);
              if (PRE) t[e] = fmaxf(t[e], 0.f);
            }
            v[b] = (rowok && (b == 0 ? lok : (b == 5 ? rok : true))) ? t : zero;
          }
        } else {
#pragma unroll
          for (int b = 0; b < 6; ++b) v[b] = widen(vr[b], PRE);
        }
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[ta * 3 + b][e] = fmaf(v[k + b][e], g[k][e], acc[ta * 3 + b][e]);
      };
      // four x-row buffers and two dy-row buffers whose roles rotate with the (unrolled) row step: no register copies - a
      // copy of a row that is still in flight would be a use and wait for it (HS and H are multiples of 4: dw_strip_ok)
      raw_t X[4][6], G[2][4];
      load_x(h0 - 1, X[0]);
      load_x(h0, X[1]);
      load_x(h0 + 1, X[2]);
      load_g(h0, G[0]);
      auto step = [&](auto I_, int oh) {
        constexpr int I = decltype(I_)::value;
        load_x(oh + 2, X[(I + 3) & 3]);  // one row ahead: consumed by the next step
        load_g(oh + 1, G[(I + 1) & 1]);
        // Two pins per step, or the FMAs of all four steps end up behind the last step's loads (sched_barrier holds the
        // machine scheduler only) and the loads are sunk to their uses: the rows consumed first in this step pass through an
        // asm statement placed behind this step's loads (memory clobber: the loads stay in front of it), the accumulators
        // through one at the end of the step.
        raw_t (&xc)[6] = X[(I + 2) & 3];
        raw_t (&gc)[4] = G[I & 1];
        asm volatile("" : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]), "+v"(xc[4]), "+v"(xc[5]), "+v"(gc[0]), "+v"(gc[1]),
                          "+v"(gc[2]), "+v"(gc[3]) : : "memory");
        __builtin_amdgcn_sched_barrier(0);
        f32x4 g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = widen(gc[k], false);
        rowmac(0, X[I & 3], g, oh - 1);
        rowmac(1, X[(I + 1) & 3], g, oh);
        rowmac(2, xc, g, oh + 1);
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                          "+v"(acc[7]), "+v"(acc[8]));
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int oh = h0; oh < h1; oh += 4) {
        step(std::integral_constant<int, 0>{}, oh);
        step(std::integral_constant<int, 1>{}, oh + 1);
        step(std::integral_constant<int, 2>{}, oh + 2);
        step(std::integral_constant<int, 3>{}, oh + 3);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ty][tx][t * 4 + e] = acc[t][e];
  __syncthreads();
  // 16 x 36 sums of 16 terms: thread (ty, tx) finishes outputs ty, ty + 16, .. of chunk tx, strips added in order 0..15
  if (c < C) {
    for (int o = ty; o < 36; o += TY) {
      float sum = 0.f;
#pragma unroll
      for (int y = 0; y < TY; ++y) sum += red[y][tx][o];
      part[((int64_t)z * 9 + (o >> 2)) * C + c + (o & 3)] = sum;
    }
  }
}

struct DwStripPlan {
  int HS, nhs, nstrips, gx, S;
  size_t part_bytes;
};

inline DwStripPlan dw_strip_plan(int num_cus, const sg_conv_desc* d) {
  DwStripPlan pl;
  pl.HS = d->H >= 128 ? 16 : (d->H >= 16 ? 8 : d->H);   // multiples of 4 (dw_strip_ok: H % 4 == 0); the last band may be 4 short
  pl.nhs = (int)sg_cdiv(d->H, pl.HS);
  pl.nstrips = d->N * pl.nhs * (d->W / 4);
  pl.gx = (int)sg_cdiv(d->Cin / 4, 16);
  int64_t S = sg_cdiv(pl.nstrips, 16);
  const int64_t cap = sg_cdiv((int64_t)4 * num_cus, pl.gx);
  if (S > cap) S = cap;
  if (S > 256) S = 256;  // few-channel maps: the finalize adds the S partial rows with 4 lanes per channel (1024 rows of a
                         // 64-channel map took longer than the strips themselves); a lane walks several strips instead
  if (S < 1) S = 1;
  pl.S = (int)S;
  pl.part_bytes = (size_t)pl.S * 9 * d->Cin * sizeof(float);
  return pl;
}

inline bool dw_run_ok(const sg_conv_desc* d) {
  return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->dilation == 1 && d->pad_t == 1 && d->pad_l == 1 &&
         d->Ho == d->H && d->Wo == d->W && (d->W % 4 == 0) && (d->Cin % 4 == 0);
}

inline bool dw_strip_ok(const sg_conv_desc* d) {
  static const int on = getenv("SG_DW_STRIP") ? atoi(getenv("SG_DW_STRIP")) : 1;
  const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
  const int64_t pix = (int64_t)d->N * d->H * d->W;
  // byte offsets are 32-bit buffer offsets in which bit 30 / bit 31 mark a column / row outside the image: tensors below 1 GiB
  return on && dw_run_ok(d) && d->H % 4 == 0 && pix * (xl > yl ? xl : yl) * 4 < (1ll << 30);
}

// Rows per run, measured (profiles/r01_bw_census.txt, SG_DW_RR = 1 / 2 / 4): the stencil kernels want tall strips on
// small maps (32x32x728 bs16: 35 -> 24 us at 4 rows) and 2 rows on large ones; the kernel-gradient reduction loses
// more from the shrinking number of runs than it gains on small maps (51 -> 65 us at 4 rows) and takes 2 rows
// only from 64x64 up (64x64x728: 248 -> 157 us).
inline int dw_rows_per_run(int H, int64_t pixels, bool wgrad) {
  static const int force = getenv("SG_DW_RR") ? atoi(getenv("SG_DW_RR")) : 0;  // A/B switch: 1, 2 or 4
  const bool small = pixels <= 32768;
  // (round 4: two rows per stencil run on the small maps too - the four-row window runs at the 168-register cap with 47 - 73
  // spilled registers; three alternating repetitions on one box: 75.90 -> 75.04 ms per step, profiles/r04_ab_runs.txt)
  const int want = force ? force : (wgrad ? (small ? 1 : 2) : 2);
  return (want >= 4 && H % 4 == 0) ? 4 : ((want >= 2 && H % 2 == 0) ? 2 : 1);
}

constexpr int DW_SUMS_MAX_ROWS = 1024;

// ---- the stencil as column strips (round 4; maps of 64 rows and more - dw_fstrip_ok; SG_DW_FSTRIP=0 restores the run kernel) ------
// dw_s1_run_kernel fetches its window once per run of RR x 4 outputs: 3 loads of 16 bytes per output chunk at RR = 2 (the
// four-row window needs more registers than three workgroups per CU leave), and was measured at 38 us (forward with the
// BatchNormalization in the gather) / 47 us (dgrad with the BatchNormalization sums) on the 47.7 MB middle-flow tensors that a
// copy moves in 20.6 us - load-issue bound like the filter gradient before dw_wgrad_strip_kernel.  Here, as there, a lane owns 4
// channels of a strip of 4 columns x HS rows and walks DOWN it with the window's three input rows in registers (transformed once:
// widening, BatchNormalization, ReLU, the zero padding put back): one new input row (6 loads) per 4 outputs, requested one row
// ahead; the epilogue's operands (ReLU mask, collected gradient, the BatchNormalization's raw input) are requested at the top of
// the step.  Rows and columns outside the image are offsets beyond the buffer descriptor (bit 31 / bit 30): no select in front of
// a load.  Same products in the same order as the run kernel (kernel rows 0..2, columns 0..2, one fp32 accumulator per output):
// bit-identical outputs; the BatchNormalization sums are added in another (fixed) order.
template <typename T>
struct DwStripGeom {
  int HS, nstrips;
  unsigned in_bytes;
  FastDiv fd_q, fd_hs;
};

template <typename T, bool RELU, bool MASK, bool BN, bool SUMS>
__global__ __launch_bounds__(256, 2) void dw_strip_kernel(const DwRunParams<T> p, const DwStripGeom<T> gm_) {
  constexpr int TX = 16, TY = 16, EB = (int)sizeof(T);
  constexpr unsigned OOB = 0x80000000u;
  typedef typename std::conditional<EB == 4, u32x4_c, u32x2_c>::type raw_t;
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> 4;
  const int c_raw = (blockIdx.x * TX + tx) * 4;
  if constexpr (!SUMS) {
    if (c_raw >= p.C) return;
  }
  const bool live = c_raw < p.C;   // SUMS: every thread reaches the reduction; a lane past the last chunk walks chunk 0 and drops its results
  const int c = live ? c_raw : 0;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p.in), 0, (int)gm_.in_bytes, 0x00020000);
  auto ldraw = [&](unsigned off) -> raw_t {
    if constexpr (EB == 4) return __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, (int)off, 0, 0);
    else return __builtin_amdgcn_raw_buffer_load_b64(rsrc_in, (int)off, 0, 0);
  };
  auto widen = [&](const raw_t r) -> f32x4 {
    if constexpr (EB == 4) {
      return __builtin_bit_cast(f32x4, r);
    } else {
      return (f32x4){__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
                     __uint_as_float(r[1] & 0xffff0000u)};
    }
  };
  f32x4 wt[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wt[t] = *reinterpret_cast<const f32x4*>(p.w + (p.flip ? 8 - t : t) * p.C + c);
  f32x4 gm = zero, bt = zero, mv = zero, iv = zero;
  if constexpr (BN) {
    gm = *reinterpret_cast<const f32x4*>(p.bn_gamma + c);
    bt = *reinterpret_cast<const f32x4*>(p.bn_beta + c);
    mv = *reinterpret_cast<const f32x4*>(p.bn_mean + c);
    iv = *reinterpret_cast<const f32x4*>(p.bn_invstd + c);
  }
  f32x4 s1 = zero, s2 = zero, bmv = zero, biv = zero, bgm = zero, bbt = zero;
  if constexpr (SUMS) {
    bmv = *reinterpret_cast<const f32x4*>(p.bs_mean + c);
    biv = *reinterpret_cast<const f32x4*>(p.bs_invstd + c);
    bgm = *reinterpret_cast<const f32x4*>(p.bs_gamma + c);
    bbt = *reinterpret_cast<const f32x4*>(p.bs_beta + c);
  }
  const int HS = gm_.HS;
  for (int s = blockIdx.y * TY + ty; s < gm_.nstrips; s += gridDim.y * TY) {
    uint32_t rowi, q, n, hs;
    fd_divmod((uint32_t)s, gm_.fd_q, rowi, q);       // strips of one row band lie side by side: neighbours share their halo columns
    fd_divmod(rowi, gm_.fd_hs, n, hs);
    const int ow0 = (int)q * 4, h0 = (int)hs * HS;
    const int h1 = h0 + HS < p.H ? h0 + HS : p.H;
    const bool lok = ow0 > 0, rok = ow0 + 4 < p.W;
    const unsigned pix0 = (unsigned)n * p.H * p.W + ow0, inrow = (unsigned)p.W * p.in_ld * EB;
    const unsigned inoff0 = (pix0 * p.in_ld + c) * EB;
    unsigned coff[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) coff[b] = (unsigned)((b - 1) * p.in_ld * EB);
    coff[0] = lok ? coff[0] : 0x40000000u;
    coff[5] = rok ? coff[5] : 0x40000000u;
    auto load_row = [&](int ih, raw_t (&v)[6]) {
      const unsigned flag = ((unsigned)ih < (unsigned)p.H && ih <= h1) ? 0u : OOB;   // (the row behind the halo is never used)
      const unsigned ro = inoff0 + (unsigned)ih * inrow;
#pragma unroll
      for (int b = 0; b < 6; ++b) v[b] = ldraw((ro + coff[b]) | flag);
    };
    // a raw row -> the six window values the products use (the run kernel's expressions, in its order)
    auto transform = [&](const raw_t (&r)[6], f32x4 (&v)[6], int ih) {
      const bool rowok = (unsigned)ih < (unsigned)p.H;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        f32x4 t = widen(r[b]);
        if constexpr (BN) {
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = fmaf((t[e] - mv[e]) * iv[e], gm[e], bt[e]);
          t = (rowok && (b == 0 ? lok : (b == 5 ? rok : true))) ? t : zero;   // zero padding of the NORMALISED tensor
        }
        if constexpr (RELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) t[e] = fmaxf(t[e], 0.f);
        }
        v[b] = t;
      }
    };
    // V[j]: the three transformed rows of the window, roles rotating with the (unrolled by three) row step; R: the raw row in flight
    f32x4 V[3][6];
    raw_t R[6];
    {
      raw_t r0[6], r1[6];
      load_row(h0 - 1, r0);
      load_row(h0, r1);
      load_row(h0 + 1, R);
      transform(r0, V[0], h0 - 1);
      transform(r1, V[1], h0);
    }
    auto step = [&](auto I_, int oh) {
      constexpr int I = decltype(I_)::value;
      // row oh + 1 has arrived (requested one step ago): transform it into the buffer of row oh - 2, then request row oh + 2 into
      // the same raw registers
      transform(R, V[(I + 2) % 3], oh + 1);
      load_row(oh + 2, R);
      const int64_t opix = (int64_t)(n * p.H + oh) * p.W + ow0;
      f32x4 acc[4] = {zero, zero, zero, zero};
#pragma unroll
      for (int ta = 0; ta < 3; ++ta) {
        const f32x4 (&v)[6] = V[(I + ta) % 3];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc[k] += v[k + b] * wt[ta * 3 + b];
      }
      // the epilogue in two halves of two output pixels (its operands - ReLU mask, collected gradient, the BatchNormalization's raw
      // input - would hold 48 registers if all four were requested together)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        f32x4 m[2], rv[2], bx[2];
        if constexpr (MASK) {
#pragma unroll
          for (int k = 0; k < 2; ++k) m[k] = ld4<T>(p.mask + (opix + 2 * hf + k) * p.mask_ld + c);
        }
        if (p.res) {   // uniform
#pragma unroll
          for (int k = 0; k < 2; ++k) rv[k] = ld4<T>(p.res + (opix + 2 * hf + k) * p.out_ld + c);
        }
        if constexpr (SUMS) {
#pragma unroll
          for (int k = 0; k < 2; ++k) bx[k] = ld4<T>(p.bs_x + (opix + 2 * hf + k) * p.bs_ld + c);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          f32x4 o = acc[2 * hf + k];
          if constexpr (MASK) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = m[k][e] > 0.f ? o[e] : 0.f;
          }
          if (p.res) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += rv[k][e];
          }
          if constexpr (SUMS) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float xh = (bx[k][e] - bmv[e]) * biv[e];
              const bool on = !p.bs_relu || fmaf(xh, bgm[e], bbt[e]) > 0.f;
              const float g = on ? o[e] : 0.f;
              s1[e] += g;
              s2[e] = fmaf(g, xh, s2[e]);
            }
          }
          if (SUMS && !live) continue;
          st4<T>(p.out + (opix + 2 * hf + k) * p.out_ld + c, o);
        }
      }
    };
    for (int oh = h0; oh < h1; oh += 3) {
      step(std::integral_constant<int, 0>{}, oh);
      if (oh + 1 >= h1) break;
      step(std::integral_constant<int, 1>{}, oh + 1);
      if (oh + 2 >= h1) break;
      step(std::integral_constant<int, 2>{}, oh + 2);
    }
  }
  if constexpr (SUMS) {
    // the 16 strip slots of this workgroup are added in slot order; the partial row of this workgroup goes to
    // bs_part[blockIdx.y], the rows are added in fp64 by the finalize launch (fixed order: deterministic)
    __shared__ float red[256 * 8];
    float* mine = red + threadIdx.x * 8;
#pragma unroll
    for (int e = 0; e < 4; ++e) { mine[e] = s1[e]; mine[4 + e] = s2[e]; }
    __syncthreads();
    if (ty == 0 && live) {
      f32x4 t1 = zero, t2 = zero;
      for (int j = 0; j < TY; ++j) {
        const float* qq = red + (j * TX + tx) * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) { t1[e] += qq[e]; t2[e] += qq[4 + e]; }
      }
      float* row = p.bs_part + (int64_t)blockIdx.y * 2 * p.C + c;
      *reinterpret_cast<f32x4*>(row) = t1;
      *reinterpret_cast<f32x4*>(row + p.C) = t2;
    }
  }
}

template <typename T>
inline bool dw_fstrip_ok(const DwRunParams<T>& p) {
  // SG_DW_FSTRIP: 0 never, 1 (default) maps of 64 rows and more, 2 every map.  Measured (profiles/r04_dw_fstrip_ab.txt): 64x64x256
  // 34.0 -> 30.7 us, 128x128x128 62.0 -> 54.9, 256x256x64 115.0 -> 106.1 (strips of 16 rows); the 32x32 maps of the middle flow run
  // no faster (27.2 -> 31.8 us alone, +-0.1 ms in the step at any strip height): with 1.4 waves per SIMD the strips are
  // latency-bound there, and the run kernel was not load-issue bound to begin with (76 % of copy speed).
  static const int on = getenv("SG_DW_FSTRIP") ? atoi(getenv("SG_DW_FSTRIP")) : 1;
  const int64_t pix = (int64_t)p.N * p.H * p.W;
  if (!on || (on == 1 && p.H < 64)) return false;
  // byte offsets are 32-bit buffer offsets in which bit 30 / bit 31 mark a column / row outside the image: inputs below 1 GiB
  return p.H % 4 == 0 && p.W % 4 == 0 && pix * p.in_ld * (int64_t)sizeof(T) < (1ll << 30);
}

template <typename T>
int launch_dw_strip(const DwRunParams<T>& p, hipStream_t st, int* sums_rows) {
  static const int hs_force = getenv("SG_DW_FSTRIP_HS") ? atoi(getenv("SG_DW_FSTRIP_HS")) : 0;
  DwStripGeom<T> g;
  g.HS = hs_force > 0 ? hs_force : (p.H >= 64 ? 16 : (p.H >= 16 ? 8 : p.H));
  if (g.HS % 4 != 0 || g.HS > p.H) g.HS = 4;
  const int nhs = (int)sg_cdiv(p.H, g.HS);
  g.nstrips = p.N * nhs * (p.W / 4);
  g.in_bytes = (unsigned)((int64_t)p.N * p.H * p.W * p.in_ld * (int64_t)sizeof(T));
  g.fd_q = make_fastdiv((uint32_t)(p.W / 4));
  g.fd_hs = make_fastdiv((uint32_t)nhs);
  const unsigned gx = (unsigned)sg_cdiv(p.C / 4, 16);
  int64_t gy = sg_cdiv(g.nstrips, 16);
  const int64_t cap = sg_cdiv(16384, gx);
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  if (p.bs_part && gy > DW_SUMS_MAX_ROWS) gy = DW_SUMS_MAX_ROWS;
  const dim3 grid(gx, (unsigned)gy);
  if (p.bs_part) {
    if (p.mask) hipLaunchKernelGGL((dw_strip_kernel<T, false, true, false, true>), grid, dim3(256), 0, st, p, g);
    else hipLaunchKernelGGL((dw_strip_kernel<T, false, false, false, true>), grid, dim3(256), 0, st, p, g);
    SG_LAUNCH_CHECK("dw_strip_kernel<SUMS>");
    if (sums_rows) *sums_rows = (int)gy;
    return 0;
  }
  if (p.bn_gamma) {
    if (p.relu_in) hipLaunchKernelGGL((dw_strip_kernel<T, true, false, true, false>), grid, dim3(256), 0, st, p, g);
    else hipLaunchKernelGGL((dw_strip_kernel<T, false, false, true, false>), grid, dim3(256), 0, st, p, g);
  } else if (p.mask) {
    hipLaunchKernelGGL((dw_strip_kernel<T, false, true, false, false>), grid, dim3(256), 0, st, p, g);
  } else if (p.relu_in) {
    hipLaunchKernelGGL((dw_strip_kernel<T, true, false, false, false>), grid, dim3(256), 0, st, p, g);
  } else {
    hipLaunchKernelGGL((dw_strip_kernel<T, false, false, false, false>), grid, dim3(256), 0, st, p, g);
  }
  SG_LAUNCH_CHECK("dw_strip_kernel");
  return 0;
}

template <typename T>
int launch_dw_run(const DwRunParams<T>& p_in, hipStream_t st, int* sums_rows = nullptr) {
  DwRunParams<T> p = p_in;
  if (dw_fstrip_ok(p)) {
    if (p.bs_part && (p.relu_in || p.bn_gamma)) {
      sg_set_error("dw_s1_run: BatchNormalization sums together with relu_in / a fused BatchNormalization");
      return SG_EINVAL;
    }
    if (p.mask && (p.relu_in || p.bn_gamma)) {
      sg_set_error("dw_s1_run: mask together with relu_in / a fused BatchNormalization");
      return SG_EINVAL;
    }
    return launch_dw_strip(p, st, sums_rows);
  }
  int rr = dw_rows_per_run(p.H, (int64_t)p.N * p.H * p.W, false);
  if (p.bs_part && rr == 4) rr = 2;   // the four-row window plus the BatchNormalization sums does not fit the register file
  p.nruns = (int64_t)p.N * (p.H / rr) * p.runs_per_row;
  p.fd_h = make_fastdiv((uint32_t)(p.H / rr));
  int lc = 1;
  while (lc < p.C / 4 && lc < 64) lc <<= 1;
  p.lc = lc;
  const unsigned gx = (unsigned)sg_cdiv(p.C / 4, lc);
  int64_t gy = sg_cdiv(p.nruns, 256 / lc);
  const int64_t cap = sg_cdiv(16384, gx);
  if (gy > cap) gy = cap;
  if (gy < 1) gy = 1;
  if (p.bs_part && gy > DW_SUMS_MAX_ROWS) gy = DW_SUMS_MAX_ROWS;   // partial rows the finalize launch adds per channel
  const dim3 grid(gx, (unsigned)gy);
  if (p.bs_part) {
    if (p.relu_in || p.bn_gamma) {
      sg_set_error("dw_s1_run: BatchNormalization sums together with relu_in / a fused BatchNormalization");
      return SG_EINVAL;
    }
#define SG_DW_RUN_SUMS(RR_)                                                                                             \
  do {                                                                                                                  \
    if (p.mask) hipLaunchKernelGGL((dw_s1_run_kernel<RR_, T, false, true, false, true>), grid, dim3(256), 0, st, p);    \
    else hipLaunchKernelGGL((dw_s1_run_kernel<RR_, T, false, false, false, true>), grid, dim3(256), 0, st, p);          \
  } while (0)
    if (rr == 4) SG_DW_RUN_SUMS(4);
    else if (rr == 2) SG_DW_RUN_SUMS(2);
    else SG_DW_RUN_SUMS(1);
#undef SG_DW_RUN_SUMS
    SG_LAUNCH_CHECK("dw_s1_run_kernel<SUMS>");
    if (sums_rows) *sums_rows = (int)gy;   // the number of partial rows written
    return 0;
  }
#define SG_DW_RUN(RR_)                                                                                                  \
  do {                                                                                                                  \
    if (p.mask) hipLaunchKernelGGL((dw_s1_run_kernel<RR_, T, false, true>), grid, dim3(256), 0, st, p);                 \
    else if (p.relu_in) hipLaunchKernelGGL((dw_s1_run_kernel<RR_, T, true, false>), grid, dim3(256), 0, st, p);         \
    else hipLaunchKernelGGL((dw_s1_run_kernel<RR_, T, false, false>), grid, dim3(256), 0, st, p);                       \
  } while (0)
  if (p.mask && (p.relu_in || p.bn_gamma)) {
    sg_set_error("dw_s1_run: mask together with relu_in / a fused BatchNormalization");
    return SG_EINVAL;
  }
#define SG_DW_RUN_BN(RR_)                                                                                               \
  do {                                                                                                                  \
    if (p.relu_in) hipLaunchKernelGGL((dw_s1_run_kernel<RR_, T, true, false, true>), grid, dim3(256), 0, st, p);        \
    else hipLaunchKernelGGL((dw_s1_run_kernel<RR_, T, false, false, true>), grid, dim3(256), 0, st, p);                 \
  } while (0)
  if (p.bn_gamma) {
    if (rr == 4) SG_DW_RUN_BN(4);
    else if (rr == 2) SG_DW_RUN_BN(2);
    else SG_DW_RUN_BN(1);
  } else if (rr == 4) SG_DW_RUN(4);
  else if (rr == 2) SG_DW_RUN(2);
  else SG_DW_RUN(1);
#undef SG_DW_RUN_BN
#undef SG_DW_RUN
  SG_LAUNCH_CHECK("dw_s1_run_kernel");
  return 0;
}

// ---------------------------------------------------------------------------------------------- pooling
template <typename T>
struct PoolParams {
  const T* __restrict__ x;
  const T* __restrict__ y;
  const T* __restrict__ dy;
  T* __restrict__ out;
  int N, H, W, C, Ho, Wo, k, stride, pad_t, pad_l;
  FastDiv fd_cv, fd_w, fd_h;
};

template <int V, typename T>
__global__ void maxpool_fwd_kernel(const PoolParams<T> p) {
  const uint32_t cv = p.C / V;
  const uint32_t total = (uint32_t)((int64_t)p.N * p.Ho * p.Wo * cv), stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t pix, cc, row, ow, n, oh;
    fd_divmod(i, p.fd_cv, pix, cc);
    fd_divmod(pix, p.fd_w, row, ow);
    fd_divmod(row, p.fd_h, n, oh);
    const int c = (int)cc * V;
    float m[V];
#pragma unroll
    for (int k = 0; k < V; ++k) m[k] = -INFINITY;
    for (int a = 0; a < p.k; ++a) {
      const int ih = (int)oh * p.stride - p.pad_t + a;
      if ((unsigned)ih >= (unsigned)p.H) continue;
      for (int b = 0; b < p.k; ++b) {
        const int iw = (int)ow * p.stride - p.pad_l + b;
        if ((unsigned)iw >= (unsigned)p.W) continue;
        float xv[V];
        ldv<V>(p.x + ((int64_t)(n * p.H + ih) * p.W + iw) * p.C + c, xv);
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = fmaxf(m[k], xv[k]);
      }
    }
    stv<V>(p.out + (int64_t)pix * p.C + c, m);
  }
}

// Gather form (deterministic, no atomics): input element (ih,iw) receives dy of every window in which it is the
// FIRST maximum in window scan order (row-major), which is where TF / Eigen route the gradient.
template <int V, typename T>
__global__ void maxpool_bwd_kernel(const PoolParams<T> p) {
  const uint32_t cv = p.C / V;
  const uint32_t total = (uint32_t)((int64_t)p.N * p.H * p.W * cv), stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t pix, cc, row, iw, n, ih;
    fd_divmod(i, p.fd_cv, pix, cc);
    fd_divmod(pix, p.fd_w, row, iw);
    fd_divmod(row, p.fd_h, n, ih);
    const int c = (int)cc * V;
    float xv[V], acc[V];
    ldv<V>(p.x + (int64_t)pix * p.C + c, xv);
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    // windows (oh, ow) that contain (ih, iw): oh*s - pt <= ih < oh*s - pt + k
    const int th = (int)ih + p.pad_t, tw = (int)iw + p.pad_l;
    int oh_hi = th / p.stride, ow_hi = tw / p.stride;
    int oh_lo = (th - p.k + p.stride) / p.stride, ow_lo = (tw - p.k + p.stride) / p.stride;  // ceil((t-k+1)/s)
    if (th - p.k + 1 <= 0) oh_lo = 0;
    if (tw - p.k + 1 <= 0) ow_lo = 0;
    if (oh_hi >= p.Ho) oh_hi = p.Ho - 1;
    if (ow_hi >= p.Wo) ow_hi = p.Wo - 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int64_t opix = ((int64_t)n * p.Ho + oh) * p.Wo + ow;
        float yv[V], gv[V];
        ldv<V>(p.y + opix * p.C + c, yv);
        ldv<V>(p.dy + opix * p.C + c, gv);
        bool first[V];
#pragma unroll
        for (int k = 0; k < V; ++k) first[k] = (xv[k] == yv[k]);
        // any earlier element of this window (scan order) equal to the max takes precedence
        const int a0 = (int)ih - (oh * p.stride - p.pad_t), b0 = (int)iw - (ow * p.stride - p.pad_l);
        for (int a = 0; a <= a0; ++a) {
          const int jh = oh * p.stride - p.pad_t + a;
          if ((unsigned)jh >= (unsigned)p.H) continue;
          const int bend = (a == a0) ? b0 : p.k;
          for (int b = 0; b < bend; ++b) {
            const int jw = ow * p.stride - p.pad_l + b;
            if ((unsigned)jw >= (unsigned)p.W) continue;
            float ev[V];
            ldv<V>(p.x + ((int64_t)(n * p.H + jh) * p.W + jw) * p.C + c, ev);
#pragma unroll
            for (int k = 0; k < V; ++k) first[k] = first[k] && !(ev[k] == yv[k]);
          }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += first[k] ? gv[k] : 0.f;
      }
    }
    stv<V>(p.out + (int64_t)pix * p.C + c, acc);
  }
}

// Training form (round 3): the forward also writes WHICH cell of the window held the (first) maximum, one byte per output
// element (cell = a * k + b in scan order), and the backward reads that byte and dy instead of x, y and up to eight more x
// values per window: 0.7 GB instead of 1.3 GB of traffic for the 256x256x128 pool of the Xception entry flow and a fraction
// of the loads (maxpool_bwd_kernel: 711 us for a 280 us floor).  Same routing as maxpool_bwd_kernel: strict '>' keeps the
// first maximum in scan order; a window of -inf only (or NaN only) routes to its first valid cell.
template <int V, typename T>
__global__ void maxpool_fwd_idx_kernel(const PoolParams<T> p, unsigned char* __restrict__ idx) {
  const uint32_t cv = p.C / V;
  const uint32_t total = (uint32_t)((int64_t)p.N * p.Ho * p.Wo * cv), stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t pix, cc, row, ow, n, oh;
    fd_divmod(i, p.fd_cv, pix, cc);
    fd_divmod(pix, p.fd_w, row, ow);
    fd_divmod(row, p.fd_h, n, oh);
    const int c = (int)cc * V;
    float m[V];
    unsigned id[V];
#pragma unroll
    for (int k = 0; k < V; ++k) { m[k] = -INFINITY; id[k] = 255u; }
    for (int a = 0; a < p.k; ++a) {
      const int ih = (int)oh * p.stride - p.pad_t + a;
      if ((unsigned)ih >= (unsigned)p.H) continue;
      for (int b = 0; b < p.k; ++b) {
        const int iw = (int)ow * p.stride - p.pad_l + b;
        if ((unsigned)iw >= (unsigned)p.W) continue;
        float xv[V];
        ldv<V>(p.x + ((int64_t)(n * p.H + ih) * p.W + iw) * p.C + c, xv);
        const unsigned cell = (unsigned)(a * p.k + b);
#pragma unroll
        for (int k = 0; k < V; ++k) {
          const bool take = xv[k] > m[k] || id[k] == 255u;
          m[k] = take ? xv[k] : m[k];
          id[k] = take ? cell : id[k];
        }
      }
    }
    stv<V>(p.out + (int64_t)pix * p.C + c, m);
    if constexpr (V == 4) {
      *reinterpret_cast<unsigned*>(idx + (int64_t)pix * p.C + c) = id[0] | (id[1] << 8) | (id[2] << 16) | (id[3] << 24);
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) idx[(int64_t)pix * p.C + c + k] = (unsigned char)id[k];
    }
  }
}

// gather, deterministic: input element (ih, iw) adds dy of every window whose recorded cell is (ih, iw)
template <int V, typename T>
__global__ void maxpool_bwd_idx_kernel(const PoolParams<T> p, const unsigned char* __restrict__ idx) {
  const uint32_t cv = p.C / V;
  const uint32_t total = (uint32_t)((int64_t)p.N * p.H * p.W * cv), stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t pix, cc, row, iw, n, ih;
    fd_divmod(i, p.fd_cv, pix, cc);
    fd_divmod(pix, p.fd_w, row, iw);
    fd_divmod(row, p.fd_h, n, ih);
    const int c = (int)cc * V;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    const int th = (int)ih + p.pad_t, tw = (int)iw + p.pad_l;
    int oh_hi = th / p.stride, ow_hi = tw / p.stride;
    int oh_lo = (th - p.k + p.stride) / p.stride, ow_lo = (tw - p.k + p.stride) / p.stride;  // ceil((t-k+1)/s)
    if (th - p.k + 1 <= 0) oh_lo = 0;
    if (tw - p.k + 1 <= 0) ow_lo = 0;
    if (oh_hi >= p.Ho) oh_hi = p.Ho - 1;
    if (ow_hi >= p.Wo) ow_hi = p.Wo - 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const unsigned ca = (unsigned)(th - oh * p.stride) * (unsigned)p.k;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const unsigned cell = ca + (unsigned)(tw - ow * p.stride);
        const int64_t o = (((int64_t)n * p.Ho + oh) * p.Wo + ow) * p.C + c;
        float gv[V];
        ldv<V>(p.dy + o, gv);
        if constexpr (V == 4) {
          const unsigned w = *reinterpret_cast<const unsigned*>(idx + o);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] += ((w >> (8 * k)) & 255u) == cell ? gv[k] : 0.f;
        } else {
#pragma unroll
          for (int k = 0; k < V; ++k) acc[k] += (unsigned)idx[o + k] == cell ? gv[k] : 0.f;
        }
      }
    }
    stv<V>(p.out + (int64_t)pix * p.C + c, acc);
  }
}

// AveragePooling2D(k) / GlobalAveragePooling2D as a segmented reduction: segment = output pixel (n,oh,ow),
// rows = kh*kw window cells.
template <typename T>
struct AvgPoolOp {
  static constexpr int NOUT = 1;
  const T* __restrict__ x;
  T* y;
  int H, W, C, Ho, Wo, kh, kw;
  FastDiv fd_howo, fd_wo, fd_kw;
  template <int V>
  __device__ __forceinline__ void accum(int seg, int64_t r, int c, float (&acc)[1][V]) const {
    uint32_t n, rem, oh, ow, a, b;
    fd_divmod((uint32_t)seg, fd_howo, n, rem);
    fd_divmod(rem, fd_wo, oh, ow);
    fd_divmod((uint32_t)r, fd_kw, a, b);
    float xv[V];
    ldv<V>(x + ((int64_t)(n * H + oh * kh + a) * W + ow * kw + b) * C + c, xv);
#pragma unroll
    for (int k = 0; k < V; ++k) acc[0][k] += xv[k];
  }
  __device__ __forceinline__ void finalize(int seg, int c, const double (&s)[1]) const {
    st1<T>(y + (int64_t)seg * C + c, (float)(s[0] / (double)(kh * kw)));
  }
};

template <int V, typename T>
__global__ void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C,
                                   int kh, int kw, int accumulate, FastDiv fd_cv, FastDiv fd_w, FastDiv fd_h) {
  const uint32_t cv = C / V;
  const uint32_t total = (uint32_t)((int64_t)N * H * W * cv), stride = gridDim.x * blockDim.x;
  const int Ho = H / kh, Wo = W / kw;
  const float sc = 1.0f / (float)(kh * kw);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t pix, cc, row, iw, n, ih;
    fd_divmod(i, fd_cv, pix, cc);
    fd_divmod(pix, fd_w, row, iw);
    fd_divmod(row, fd_h, n, ih);
    const int c = (int)cc * V;
    const int oh = (int)ih / kh, ow = (int)iw / kw;
    float o[V];
    if (oh < Ho && ow < Wo) {
      ldv<V>(dy + (((int64_t)n * Ho + oh) * Wo + ow) * C + c, o);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] *= sc;
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = 0.f;
    }
    if (accumulate) {
      float t[V];
      ldv<V>(dx + (int64_t)pix * C + c, t);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] += t[k];
    }
    stv<V>(dx + (int64_t)pix * C + c, o);
  }
}

template <int V, typename T>
__global__ void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                    int sh, int sw, int y_ld, FastDiv fd_cv, FastDiv fd_w, FastDiv fd_h) {
  const uint32_t cv = C / V;
  const int OH = H * sh, OW = W * sw;
  const uint32_t total = (uint32_t)((int64_t)N * OH * OW * cv), stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t pix, cc, row, ow, n, oh;
    fd_divmod(i, fd_cv, pix, cc);
    fd_divmod(pix, fd_w, row, ow);
    fd_divmod(row, fd_h, n, oh);
    const int c = (int)cc * V;
    float v[V];
    ldv<V>(x + (((int64_t)n * H + oh / sh) * W + ow / sw) * C + c, v);
    stv<V>(y + (int64_t)pix * y_ld + c, v);
  }
}

template <int V, typename T>
__global__ void upsample_bwd_kernel(const T* __restrict__ dy, int dy_ld, T* __restrict__ dx, int N, int H,
                                    int W, int C, int sh, int sw, int accumulate, FastDiv fd_cv, FastDiv fd_w,
                                    FastDiv fd_h) {
  const uint32_t cv = C / V;
  const uint32_t total = (uint32_t)((int64_t)N * H * W * cv), stride = gridDim.x * blockDim.x;
  const int OW = W * sw, OH = H * sh;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    uint32_t pix, cc, row, iw, n, ih;
    fd_divmod(i, fd_cv, pix, cc);
    fd_divmod(pix, fd_w, row, iw);
    fd_divmod(row, fd_h, n, ih);
    const int c = (int)cc * V;
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    for (int a = 0; a < sh; ++a)
      for (int b = 0; b < sw; ++b) {
        float g[V];
        ldv<V>(dy + (((int64_t)n * OH + ih * sh + a) * OW + iw * sw + b) * dy_ld + c, g);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += g[k];
      }
    if (accumulate) {
      float t[V];
      ldv<V>(dx + (int64_t)pix * C + c, t);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += t[k];
    }
    stv<V>(dx + (int64_t)pix * C + c, acc);
  }
}

// Large windows (the ASPP image-pooling branch: 1x1 -> 32x32, 1024 cells per output element): the kernel above would give
// each of N*C/4 = 1024 lanes a serial sum of 1024 loads (285 us for 17 MB).  Here a workgroup owns one output pixel and 8
// channel chunks, 32 lanes share the window cells, the 32 partial sums are added in fixed order through LDS.
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_window_kernel(const T* __restrict__ dy, int dy_ld, T* __restrict__ dx, int H,
                                                                  int W, int C, int sh, int sw, int accumulate, FastDiv fd_w,
                                                                  FastDiv fd_h, FastDiv fd_sw) {
  constexpr int TX = 8, TY = 32;
  __shared__ float red[TY][TX][4];
  const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> 3;
  const int c = (blockIdx.x * TX + tx) * 4;
  uint32_t row, iw, n, ih;
  fd_divmod(blockIdx.y, fd_w, row, iw);
  fd_divmod(row, fd_h, n, ih);
  const int OW = W * sw, OH = H * sh;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const T* base = dy + (((int64_t)n * OH + ih * sh) * OW + iw * sw) * dy_ld + c;
#pragma unroll 4
    for (int r = ty; r < sh * sw; r += TY) {
      uint32_t a, b;
      fd_divmod((uint32_t)r, fd_sw, a, b);
      const f32x4 g = ld4<T>(base + ((int64_t)a * OW + b) * dy_ld);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] += g[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[ty][tx][e] = acc[e];
  __syncthreads();
  if (ty == 0 && c < C) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int y = 0; y < TY; ++y)
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += red[y][tx][e];
    T* o = dx + (int64_t)blockIdx.y * C + c;
    if (accumulate) {
      const f32x4 t = ld4<T>(o);
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] += t[e];
    }
    st4<T>(o, s);
  }
}

int dw_check(const sg_ctx* ctx, int dtype, const sg_conv_desc* d, const char* who) {
  SG_CHECK_ARG(ctx && (dtype == SG_F32 || dtype == SG_BF16) && d, "%s: bad ctx/dtype/desc", who);
  SG_CHECK_ARG(d->Cin == d->Cout, "%s: depthwise needs Cin == Cout (depth_multiplier 1)", who);
  SG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Ho > 0 && d->Wo > 0 && d->KH > 0 && d->KW > 0 &&
                   d->stride > 0 && d->dilation > 0,
               "%s: bad geometry", who);
  const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
  SG_CHECK_ARG((int64_t)d->N * d->H * d->W * xl < (1ll << 31) && (int64_t)d->N * d->Ho * d->Wo * yl < (1ll << 31),
               "%s: tensor exceeds 2^31 elements", who);
  return 0;
}

template <typename T>
void dw_fill(DwParams<T>& p, const sg_conv_desc* d) {
  p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.KH = d->KH; p.KW = d->KW;
  p.stride = d->stride; p.dil = d->dilation; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  p.x_ld = d->x_ld ? d->x_ld : d->Cin;
  p.y_ld = d->y_ld ? d->y_ld : d->Cout;
}

}  // namespace

extern "C" {

static int dwconv2d_fwd_impl(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w, void* y,
                             int pre_relu, const void* bn_gamma, const void* bn_beta, const void* bn_mean, const void* bn_invstd) {
  int rc = dw_check(ctx, dtype, d, "sg_dwconv2d_fwd");
  if (rc) return rc;
  SG_CHECK_ARG(x && w && y, "sg_dwconv2d_fwd: null tensor");
  if (bn_gamma) {
    SG_CHECK_ARG(bn_beta && bn_mean && bn_invstd, "sg_dwconv2d_fwd_bn: null BatchNormalization parameter");
    const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
    if (!((d->Cin % 4 == 0) && (xl % 4 == 0) && (yl % 4 == 0) && sg_aligned16(x) && sg_aligned16(w) && sg_aligned16(y) &&
          sg_aligned16(bn_gamma) && sg_aligned16(bn_beta) && sg_aligned16(bn_mean) && sg_aligned16(bn_invstd) && dw_run_ok(d))) {
      sg_set_error("sg_dwconv2d_fwd_bn: only the stride-1 3x3 run kernels (W %% 4 == 0, C %% 4 == 0, 16-byte aligned) fuse the "
                   "BatchNormalization; materialise it instead");
      return SG_EUNSUPPORTED;
    }
  }
  SG_DTYPE_SWITCH(dtype, "sg_dwconv2d_fwd", {
    DwParams<T> p;
    dw_fill(p, d);
    p.x = (const T*)x; p.w = (const float*)w; p.dy = nullptr; p.out = (T*)y; p.pre_relu = pre_relu;
    const bool vec = (p.C % 4 == 0) && (p.x_ld % 4 == 0) && (p.y_ld % 4 == 0) && sg_aligned16(x) && sg_aligned16(w) && sg_aligned16(y);
    if (vec && dw_run_ok(d)) {
      DwRunParams<T> r;
      r.in = (const T*)x; r.w = (const float*)w; r.mask = nullptr; r.res = nullptr; r.out = (T*)y;
      r.bn_gamma = (const float*)bn_gamma; r.bn_beta = (const float*)bn_beta; r.bn_mean = (const float*)bn_mean; r.bn_invstd = (const float*)bn_invstd;
      r.N = d->N; r.H = d->H; r.W = d->W; r.C = p.C; r.in_ld = p.x_ld; r.out_ld = p.y_ld; r.mask_ld = 0;
      r.relu_in = pre_relu; r.flip = 0; r.runs_per_row = d->W / 4; r.nruns = (int64_t)d->N * d->H * r.runs_per_row;
      r.fd_rpr = make_fastdiv((uint32_t)r.runs_per_row); r.fd_h = make_fastdiv((uint32_t)d->H);
      return launch_dw_run(r, (hipStream_t)stream);
    }
    const int V = vec ? 4 : 1;
    p.fd_cv = make_fastdiv((uint32_t)(p.C / V)); p.fd_w = make_fastdiv((uint32_t)p.Wo); p.fd_h = make_fastdiv((uint32_t)p.Ho);
    const unsigned blocks = ew_blocks((int64_t)p.N * p.Ho * p.Wo * (p.C / V));
    if (vec) hipLaunchKernelGGL((dw_fwd_kernel<4, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((dw_fwd_kernel<1, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  });
  SG_LAUNCH_CHECK("dw_fwd_kernel");
  return 0;
}

int sg_dwconv2d_fwd(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w, void* y,
                    int pre_relu) {
  return dwconv2d_fwd_impl(ctx, stream, dtype, d, x, w, y, pre_relu, nullptr, nullptr, nullptr, nullptr);
}

int sg_dwconv2d_fwd_bn(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w, void* y,
                       const void* gamma, const void* beta, const void* mean, const void* invstd, int relu) {
  SG_CHECK_ARG(gamma != nullptr, "sg_dwconv2d_fwd_bn: null gamma");
  return dwconv2d_fwd_impl(ctx, stream, dtype, d, x, w, y, relu, gamma, beta, mean, invstd);
}

static int dwconv2d_dgrad_impl(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                              const void* x_for_mask, void* dx, int pre_relu, const void* res) {
  int rc = dw_check(ctx, dtype, d, "sg_dwconv2d_dgrad");
  if (rc) return rc;
  SG_CHECK_ARG(dy && w && dx, "sg_dwconv2d_dgrad: null tensor");
  SG_CHECK_ARG(!pre_relu || x_for_mask, "sg_dwconv2d_dgrad: pre_relu needs the forward input");
  if (res) {
    const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
    if (!((d->Cin % 4 == 0) && (xl % 4 == 0) && (yl % 4 == 0) && sg_aligned16(dy) && sg_aligned16(w) && sg_aligned16(dx) &&
          sg_aligned16(res) && (!pre_relu || sg_aligned16(x_for_mask)) && dw_run_ok(d))) {
      sg_set_error("sg_dwconv2d_dgrad_acc: only the stride-1 3x3 run kernels (W %% 4 == 0, C %% 4 == 0, 16-byte aligned) add a "
                   "collected gradient; add it afterwards instead");
      return SG_EUNSUPPORTED;
    }
  }
  SG_DTYPE_SWITCH(dtype, "sg_dwconv2d_dgrad", {
    DwParams<T> p;
    dw_fill(p, d);
    p.x = (const T*)x_for_mask; p.w = (const float*)w; p.dy = (const T*)dy; p.out = (T*)dx; p.pre_relu = pre_relu;
    const bool vec = (p.C % 4 == 0) && (p.x_ld % 4 == 0) && (p.y_ld % 4 == 0) && sg_aligned16(dy) && sg_aligned16(w) &&
                     sg_aligned16(dx) && (!pre_relu || sg_aligned16(x_for_mask));
    if (vec && dw_run_ok(d)) {  // stride-1 dgrad = the same stencil with the kernel flipped
      DwRunParams<T> r;
      r.in = (const T*)dy; r.w = (const float*)w; r.mask = pre_relu ? (const T*)x_for_mask : nullptr; r.out = (T*)dx;
      r.res = (const T*)res;
      r.bn_gamma = r.bn_beta = r.bn_mean = r.bn_invstd = nullptr;
      r.N = d->N; r.H = d->H; r.W = d->W; r.C = p.C; r.in_ld = p.y_ld; r.out_ld = p.x_ld; r.mask_ld = p.x_ld;
      r.relu_in = 0; r.flip = 1; r.runs_per_row = d->W / 4; r.nruns = (int64_t)d->N * d->H * r.runs_per_row;
      r.fd_rpr = make_fastdiv((uint32_t)r.runs_per_row); r.fd_h = make_fastdiv((uint32_t)d->H);
      return launch_dw_run(r, (hipStream_t)stream);
    }
    const int V = vec ? 4 : 1;
    p.fd_cv = make_fastdiv((uint32_t)(p.C / V)); p.fd_w = make_fastdiv((uint32_t)p.W); p.fd_h = make_fastdiv((uint32_t)p.H);
    const unsigned blocks = ew_blocks((int64_t)p.N * p.H * p.W * (p.C / V));
    if (vec) hipLaunchKernelGGL((dw_dgrad_kernel<4, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((dw_dgrad_kernel<1, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  });
  SG_LAUNCH_CHECK("dw_dgrad_kernel");
  return 0;
}

int sg_dwconv2d_dgrad(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                      const void* x_for_mask, void* dx, int pre_relu) {
  return dwconv2d_dgrad_impl(ctx, stream, dtype, d, dy, w, x_for_mask, dx, pre_relu, nullptr);
}

int sg_dwconv2d_dgrad_acc(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                          const void* x_for_mask, void* dx, int pre_relu, const void* res) {
  SG_CHECK_ARG(res != nullptr, "sg_dwconv2d_dgrad_acc: null res");
  return dwconv2d_dgrad_impl(ctx, stream, dtype, d, dy, w, x_for_mask, dx, pre_relu, res);
}

// second stage of the BatchNormalization sums written by dw_s1_run_kernel<SUMS>: seg_finalize_kernel adds the partial rows in
// fp64 (fixed order) and hands the two totals of a channel to this op
struct BnSumsFinalOp {
  static constexpr int NOUT = 2;
  float* dgamma;
  float* dbeta;
  __device__ __forceinline__ void finalize(int, int c, const double (&s)[2]) const {
    dbeta[c] = (float)s[0];
    dgamma[c] = (float)s[1];
  }
};

size_t sg_dwconv2d_dgrad_bnsums_ws_bytes(const sg_ctx* ctx, const sg_conv_desc* d) {
  if (!ctx || !d) return 0;
  return (size_t)DW_SUMS_MAX_ROWS * 2 * (size_t)d->Cin * sizeof(float) + 256;
}

int sg_dwconv2d_dgrad_bnsums(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                             const void* x_for_mask, void* dx, int pre_relu, const void* res, const void* bn_x,
                             const void* bn_mean, const void* bn_invstd, const void* bn_gamma, const void* bn_beta, int bn_relu,
                             void* dgamma, void* dbeta, void* ws, size_t ws_bytes) {
  int rc = dw_check(ctx, dtype, d, "sg_dwconv2d_dgrad_bnsums");
  if (rc) return rc;
  SG_CHECK_ARG(dy && w && dx && bn_x && bn_mean && bn_invstd && bn_gamma && bn_beta && dgamma && dbeta,
               "sg_dwconv2d_dgrad_bnsums: null tensor");
  SG_CHECK_ARG(!pre_relu || x_for_mask, "sg_dwconv2d_dgrad_bnsums: pre_relu needs the forward input");
  const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
  if (!((d->Cin % 4 == 0) && (xl % 4 == 0) && (yl % 4 == 0) && sg_aligned16(dy) && sg_aligned16(w) && sg_aligned16(dx) &&
        sg_aligned16(bn_x) && sg_aligned16(bn_mean) && sg_aligned16(bn_invstd) && sg_aligned16(bn_gamma) && sg_aligned16(bn_beta) &&
        (!res || sg_aligned16(res)) && (!pre_relu || sg_aligned16(x_for_mask)) && dw_run_ok(d))) {
    sg_set_error("sg_dwconv2d_dgrad_bnsums: only the stride-1 3x3 run kernels (W %% 4 == 0, C %% 4 == 0, 16-byte aligned); "
                 "use sg_dwconv2d_dgrad and sg_bn_train_bwd instead");
    return SG_EUNSUPPORTED;
  }
  if (!ws || ws_bytes < sg_dwconv2d_dgrad_bnsums_ws_bytes(ctx, d) - 256) {
    sg_set_error("sg_dwconv2d_dgrad_bnsums: workspace %zu < %zu", ws_bytes, sg_dwconv2d_dgrad_bnsums_ws_bytes(ctx, d) - 256);
    return SG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  SG_DTYPE_SWITCH(dtype, "sg_dwconv2d_dgrad_bnsums", {
    DwRunParams<T> r;
    r.in = (const T*)dy; r.w = (const float*)w; r.mask = pre_relu ? (const T*)x_for_mask : nullptr; r.out = (T*)dx;
    r.res = (const T*)res;
    r.bn_gamma = r.bn_beta = r.bn_mean = r.bn_invstd = nullptr;
    r.N = d->N; r.H = d->H; r.W = d->W; r.C = d->Cin; r.in_ld = yl; r.out_ld = xl; r.mask_ld = xl;
    r.relu_in = 0; r.flip = 1; r.runs_per_row = d->W / 4; r.nruns = (int64_t)d->N * d->H * r.runs_per_row;
    r.fd_rpr = make_fastdiv((uint32_t)r.runs_per_row); r.fd_h = make_fastdiv((uint32_t)d->H);
    r.bs_x = (const T*)bn_x; r.bs_ld = d->Cin;   /* the BatchNormalization's input is a dense tensor */ r.bs_mean = (const float*)bn_mean; r.bs_invstd = (const float*)bn_invstd;
    r.bs_gamma = (const float*)bn_gamma; r.bs_beta = (const float*)bn_beta; r.bs_relu = bn_relu ? 1 : 0;
    r.bs_part = (float*)ws;
    int rows = 0;
    rc = launch_dw_run(r, st, &rows);
    if (rc) return rc;
    BnSumsFinalOp op;
    op.dgamma = (float*)dgamma; op.dbeta = (float*)dbeta;
    seg_finalize_launch(op, 1, d->Cin, rows, (const float*)ws, st);
  });
  SG_LAUNCH_CHECK("sg_dwconv2d_dgrad_bnsums");
  return 0;
}

size_t sg_dwconv2d_wgrad_ws_bytes(const sg_ctx* ctx, const sg_conv_desc* d) {
  if (!ctx || !d) return 0;
  const int64_t rows = (int64_t)d->N * d->Ho * d->Wo;
  const SegPlan a = seg_plan<9>(ctx->num_cus, 1, rows, d->Cin, true), b = seg_plan<9>(ctx->num_cus, 1, rows, d->Cin, false);
  const SegPlan r = seg_plan<9>(ctx->num_cus, 1, sg_cdiv(rows, 4), d->Cin, true);
  size_t m = a.part_bytes > b.part_bytes ? a.part_bytes : b.part_bytes;
  if (r.part_bytes > m) m = r.part_bytes;
  if (dw_run_ok(d)) {
    const DwStripPlan sp = dw_strip_plan(ctx->num_cus, d);
    if (sp.part_bytes > m) m = sp.part_bytes;
  }
  return m + 256;
}

static int dwconv2d_wgrad_impl(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* dy,
                               void* dw, int pre_relu, void* ws, size_t ws_bytes, const void* bn_gamma, const void* bn_beta,
                               const void* bn_mean, const void* bn_invstd) {
  int rc = dw_check(ctx, dtype, d, "sg_dwconv2d_wgrad");
  if (rc) return rc;
  SG_CHECK_ARG(x && dy && dw, "sg_dwconv2d_wgrad: null tensor");
  if (bn_gamma) {
    SG_CHECK_ARG(bn_beta && bn_mean && bn_invstd, "sg_dwconv2d_wgrad_bn: null BatchNormalization parameter");
    const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
    if (!((d->Cin % 4 == 0) && (xl % 4 == 0) && (yl % 4 == 0) && sg_aligned16(x) && sg_aligned16(dy) && sg_aligned16(bn_gamma) &&
          sg_aligned16(bn_beta) && sg_aligned16(bn_mean) && sg_aligned16(bn_invstd) && dw_run_ok(d))) {
      sg_set_error("sg_dwconv2d_wgrad_bn: only the stride-1 3x3 run kernels fuse the BatchNormalization; materialise it instead");
      return SG_EUNSUPPORTED;
    }
  }
  SG_CHECK_ARG(d->KH == 3 && d->KW == 3, "sg_dwconv2d_wgrad: only 3x3 depthwise kernels occur on this path");
  SG_DTYPE_SWITCH(dtype, "sg_dwconv2d_wgrad", {
    DwWgradOp<T> op;
    op.x = (const T*)x; op.dy = (const T*)dy; op.dw = (float*)dw;
    op.H = d->H; op.W = d->W; op.C = d->Cin; op.Ho = d->Ho; op.Wo = d->Wo; op.stride = d->stride; op.dil = d->dilation;
    op.pad_t = d->pad_t; op.pad_l = d->pad_l; op.x_ld = d->x_ld ? d->x_ld : d->Cin; op.y_ld = d->y_ld ? d->y_ld : d->Cout;
    op.pre_relu = pre_relu;
    op.fd_w = make_fastdiv((uint32_t)d->Wo); op.fd_h = make_fastdiv((uint32_t)d->Ho);
    const int64_t rows = (int64_t)d->N * d->Ho * d->Wo;
    const bool vec = (op.C % 4 == 0) && (op.x_ld % 4 == 0) && (op.y_ld % 4 == 0) && sg_aligned16(x) && sg_aligned16(dy);
    if (vec && dw_strip_ok(d)) {
      const DwStripPlan sp = dw_strip_plan(ctx->num_cus, d);
      if (!ws || ws_bytes < sp.part_bytes) {
        sg_set_error("sg_dwconv2d_wgrad: workspace %zu < %zu", ws_bytes, sp.part_bytes);
        return SG_EWORKSPACE;
      }
      const dim3 grid((unsigned)sp.gx, (unsigned)sp.S);
      const FastDiv fq = make_fastdiv((uint32_t)(d->W / 4)), fh = make_fastdiv((uint32_t)sp.nhs);
      const unsigned xb_ = (unsigned)((int64_t)d->N * d->H * d->W * op.x_ld * (int64_t)sizeof(T));
      const unsigned yb_ = (unsigned)((int64_t)d->N * d->H * d->W * op.y_ld * (int64_t)sizeof(T));
      auto strip = [&](auto pre_, auto bn_) {
        constexpr bool PRE_ = decltype(pre_)::value, BN_ = decltype(bn_)::value;
        static const int occ2 = getenv("SG_DW_STRIP_OCC2") ? atoi(getenv("SG_DW_STRIP_OCC2")) : 0;
        if (BN_ && occ2)
          hipLaunchKernelGGL((dw_wgrad_strip_kernel<T, PRE_, BN_, BN_>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy,
                             (float*)ws, d->H, d->W, op.C, op.x_ld, op.y_ld, sp.HS, sp.nstrips, sp.S, xb_, yb_, fq, fh,
                             (const float*)bn_gamma, (const float*)bn_beta, (const float*)bn_mean, (const float*)bn_invstd);
        else
        hipLaunchKernelGGL((dw_wgrad_strip_kernel<T, PRE_, BN_>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)dy,
                           (float*)ws, d->H, d->W, op.C, op.x_ld, op.y_ld, sp.HS, sp.nstrips, sp.S, xb_, yb_, fq, fh,
                           (const float*)bn_gamma, (const float*)bn_beta, (const float*)bn_mean, (const float*)bn_invstd);
      };
      if (bn_gamma) {
        if (pre_relu) strip(std::true_type{}, std::true_type{});
        else strip(std::false_type{}, std::true_type{});
      } else if (pre_relu) strip(std::true_type{}, std::false_type{});
      else strip(std::false_type{}, std::false_type{});
      SG_LAUNCH_CHECK("dw_wgrad_strip_kernel");
      DwWgradRunOp<1, T, false> fin;   // finalize() only: dw[t][c] = the fp64 sum of the S partial rows
      fin.dw = (float*)dw; fin.C = op.C;
      seg_finalize_launch(fin, 1, op.C, sp.S, (const float*)ws, (hipStream_t)stream);
      SG_LAUNCH_CHECK("dw_wgrad_strip finalize");
      return 0;
    }
    if (vec && dw_run_ok(d)) {
      const int rr = dw_rows_per_run(d->H, rows, true);
      const int64_t nruns = rows / (4 * rr);
      const SegPlan rp = seg_plan<9>(ctx->num_cus, 1, nruns, op.C, true);
      if (!ws || ws_bytes < rp.part_bytes) {
        sg_set_error("sg_dwconv2d_wgrad: workspace %zu < %zu", ws_bytes, rp.part_bytes);
        return SG_EWORKSPACE;
      }
      auto run = [&](auto ro) -> int {
        ro.x = (const T*)x; ro.dy = (const T*)dy; ro.dw = (float*)dw; ro.H = d->H; ro.W = d->W; ro.C = op.C;
        ro.x_ld = op.x_ld; ro.y_ld = op.y_ld;
        ro.bn_gamma = (const float*)bn_gamma; ro.bn_beta = (const float*)bn_beta; ro.bn_mean = (const float*)bn_mean;
        ro.bn_invstd = (const float*)bn_invstd;
        ro.fd_rpr = make_fastdiv((uint32_t)(d->W / 4)); ro.fd_h = make_fastdiv((uint32_t)(d->H / rr));
        return seg_reduce_launch(ro, rp, 1, nruns, op.C, (float*)ws, (hipStream_t)stream, "dw_wgrad_run");
      };
      if (bn_gamma) {
        if (pre_relu) {
          if (rr == 4) return run(DwWgradRunOp<4, T, true, true>{});
          if (rr == 2) return run(DwWgradRunOp<2, T, true, true>{});
          return run(DwWgradRunOp<1, T, true, true>{});
        }
        if (rr == 4) return run(DwWgradRunOp<4, T, false, true>{});
        if (rr == 2) return run(DwWgradRunOp<2, T, false, true>{});
        return run(DwWgradRunOp<1, T, false, true>{});
      }
      if (pre_relu) {
        if (rr == 4) return run(DwWgradRunOp<4, T, true>{});
        if (rr == 2) return run(DwWgradRunOp<2, T, true>{});
        return run(DwWgradRunOp<1, T, true>{});
      }
      if (rr == 4) return run(DwWgradRunOp<4, T, false>{});
      if (rr == 2) return run(DwWgradRunOp<2, T, false>{});
      return run(DwWgradRunOp<1, T, false>{});
    }
    const SegPlan pl = seg_plan<9>(ctx->num_cus, 1, rows, op.C, vec);
    if (!ws || ws_bytes < pl.part_bytes) {
      sg_set_error("sg_dwconv2d_wgrad: workspace %zu < %zu", ws_bytes, pl.part_bytes);
      return SG_EWORKSPACE;
    }
    return seg_reduce_launch(op, pl, 1, rows, op.C, (float*)ws, (hipStream_t)stream, "dw_wgrad");
  });
  return 0;
}

int sg_dwconv2d_wgrad(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* dy,
                      void* dw, int pre_relu, void* ws, size_t ws_bytes) {
  return dwconv2d_wgrad_impl(ctx, stream, dtype, d, x, dy, dw, pre_relu, ws, ws_bytes, nullptr, nullptr, nullptr, nullptr);
}

int sg_dwconv2d_wgrad_bn(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* dy, void* dw,
                         const void* gamma, const void* beta, const void* mean, const void* invstd, int relu, void* ws,
                         size_t ws_bytes) {
  SG_CHECK_ARG(gamma != nullptr, "sg_dwconv2d_wgrad_bn: null gamma");
  return dwconv2d_wgrad_impl(ctx, stream, dtype, d, x, dy, dw, relu, ws, ws_bytes, gamma, beta, mean, invstd);
}

int sg_maxpool_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int k, int stride, int pad_t,
                   int pad_l, int Ho, int Wo, const void* x, void* y) {
  SG_CHECK_ARG(ctx && x && y, "sg_maxpool_fwd: bad argument");
  SG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && Ho > 0 && Wo > 0 && pad_t >= 0 && pad_l >= 0,
               "sg_maxpool_fwd: bad geometry");
  SG_CHECK_ARG((int64_t)N * H * W * C < (1ll << 31), "sg_maxpool_fwd: tensor exceeds 2^31 elements");
  SG_DTYPE_SWITCH(dtype, "sg_maxpool_fwd", {
    PoolParams<T> p;
    p.x = (const T*)x; p.y = nullptr; p.dy = nullptr; p.out = (T*)y;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
    const bool vec = (C % 4 == 0) && sg_aligned16(x) && sg_aligned16(y);
    const int V = vec ? 4 : 1;
    p.fd_cv = make_fastdiv((uint32_t)(C / V)); p.fd_w = make_fastdiv((uint32_t)Wo); p.fd_h = make_fastdiv((uint32_t)Ho);
    const unsigned blocks = ew_blocks((int64_t)N * Ho * Wo * (C / V));
    if (vec) hipLaunchKernelGGL((maxpool_fwd_kernel<4, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((maxpool_fwd_kernel<1, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  });
  SG_LAUNCH_CHECK("maxpool_fwd_kernel");
  return 0;
}

int sg_maxpool_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int k, int stride, int pad_t,
                   int pad_l, int Ho, int Wo, const void* x, const void* y, const void* dy, void* dx) {
  SG_CHECK_ARG(ctx && x && y && dy && dx, "sg_maxpool_bwd: bad argument");
  SG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && Ho > 0 && Wo > 0 && pad_t >= 0 && pad_l >= 0,
               "sg_maxpool_bwd: bad geometry");
  SG_CHECK_ARG((int64_t)N * H * W * C < (1ll << 31), "sg_maxpool_bwd: tensor exceeds 2^31 elements");
  SG_DTYPE_SWITCH(dtype, "sg_maxpool_bwd", {
    PoolParams<T> p;
    p.x = (const T*)x; p.y = (const T*)y; p.dy = (const T*)dy; p.out = (T*)dx;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
    const bool vec = (C % 4 == 0) && sg_aligned16(x) && sg_aligned16(y) && sg_aligned16(dy) && sg_aligned16(dx);
    const int V = vec ? 4 : 1;
    p.fd_cv = make_fastdiv((uint32_t)(C / V)); p.fd_w = make_fastdiv((uint32_t)W); p.fd_h = make_fastdiv((uint32_t)H);
    const unsigned blocks = ew_blocks((int64_t)N * H * W * (C / V));
    if (vec) hipLaunchKernelGGL((maxpool_bwd_kernel<4, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((maxpool_bwd_kernel<1, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  });
  SG_LAUNCH_CHECK("maxpool_bwd_kernel");
  return 0;
}

int sg_maxpool_fwd_idx(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int k, int stride, int pad_t,
                       int pad_l, int Ho, int Wo, const void* x, void* y, void* idx) {
  SG_CHECK_ARG(ctx && x && y && idx, "sg_maxpool_fwd_idx: bad argument");
  SG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && k <= 15 && stride > 0 && Ho > 0 && Wo > 0 && pad_t >= 0 && pad_l >= 0,
               "sg_maxpool_fwd_idx: bad geometry (windows up to 15 x 15: the cell index is one byte)");
  SG_CHECK_ARG((int64_t)N * H * W * C < (1ll << 31), "sg_maxpool_fwd_idx: tensor exceeds 2^31 elements");
  SG_DTYPE_SWITCH(dtype, "sg_maxpool_fwd_idx", {
    PoolParams<T> p;
    p.x = (const T*)x; p.y = nullptr; p.dy = nullptr; p.out = (T*)y;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
    const bool vec = (C % 4 == 0) && sg_aligned16(x) && sg_aligned16(y) && sg_aligned16(idx);
    const int V = vec ? 4 : 1;
    p.fd_cv = make_fastdiv((uint32_t)(C / V)); p.fd_w = make_fastdiv((uint32_t)Wo); p.fd_h = make_fastdiv((uint32_t)Ho);
    const unsigned blocks = ew_blocks((int64_t)N * Ho * Wo * (C / V));
    if (vec) hipLaunchKernelGGL((maxpool_fwd_idx_kernel<4, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, (unsigned char*)idx);
    else hipLaunchKernelGGL((maxpool_fwd_idx_kernel<1, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, (unsigned char*)idx);
  });
  SG_LAUNCH_CHECK("maxpool_fwd_idx_kernel");
  return 0;
}

int sg_maxpool_bwd_idx(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int k, int stride, int pad_t,
                       int pad_l, int Ho, int Wo, const void* dy, const void* idx, void* dx) {
  SG_CHECK_ARG(ctx && dy && idx && dx, "sg_maxpool_bwd_idx: bad argument");
  SG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && k <= 15 && stride > 0 && Ho > 0 && Wo > 0 && pad_t >= 0 && pad_l >= 0,
               "sg_maxpool_bwd_idx: bad geometry");
  SG_CHECK_ARG((int64_t)N * H * W * C < (1ll << 31), "sg_maxpool_bwd_idx: tensor exceeds 2^31 elements");
  SG_DTYPE_SWITCH(dtype, "sg_maxpool_bwd_idx", {
    PoolParams<T> p;
    p.x = nullptr; p.y = nullptr; p.dy = (const T*)dy; p.out = (T*)dx;
    p.N = N; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.k = k; p.stride = stride; p.pad_t = pad_t; p.pad_l = pad_l;
    const bool vec = (C % 4 == 0) && sg_aligned16(dy) && sg_aligned16(dx) && sg_aligned16(idx);
    const int V = vec ? 4 : 1;
    p.fd_cv = make_fastdiv((uint32_t)(C / V)); p.fd_w = make_fastdiv((uint32_t)W); p.fd_h = make_fastdiv((uint32_t)H);
    const unsigned blocks = ew_blocks((int64_t)N * H * W * (C / V));
    if (vec) hipLaunchKernelGGL((maxpool_bwd_idx_kernel<4, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, (const unsigned char*)idx);
    else hipLaunchKernelGGL((maxpool_bwd_idx_kernel<1, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, (const unsigned char*)idx);
  });
  SG_LAUNCH_CHECK("maxpool_bwd_idx_kernel");
  return 0;
}

size_t sg_avgpool_ws_bytes(const sg_ctx* ctx, int N, int H, int W, int C, int kh, int kw) {
  if (!ctx || kh <= 0 || kw <= 0) return 0;
  const int nseg = N * (H / kh) * (W / kw);
  const SegPlan a = seg_plan<1>(ctx->num_cus, nseg, (int64_t)kh * kw, C, true),
                b = seg_plan<1>(ctx->num_cus, nseg, (int64_t)kh * kw, C, false);
  return (a.part_bytes > b.part_bytes ? a.part_bytes : b.part_bytes) + 256;
}

int sg_avgpool_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int kh, int kw, const void* x,
                   void* y, void* ws, size_t ws_bytes) {
  SG_CHECK_ARG(ctx && x && y, "sg_avgpool_fwd: bad argument");
  SG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && H >= kh && W >= kw, "sg_avgpool_fwd: bad geometry");
  SG_CHECK_ARG((int64_t)N * H * W * C < (1ll << 31), "sg_avgpool_fwd: tensor exceeds 2^31 elements");
  SG_DTYPE_SWITCH(dtype, "sg_avgpool_fwd", {
  AvgPoolOp<T> op;
  op.x = (const T*)x; op.y = (T*)y; op.H = H; op.W = W; op.C = C; op.Ho = H / kh; op.Wo = W / kw; op.kh = kh; op.kw = kw;
  op.fd_howo = make_fastdiv((uint32_t)(op.Ho * op.Wo)); op.fd_wo = make_fastdiv((uint32_t)op.Wo); op.fd_kw = make_fastdiv((uint32_t)kw);
  const int nseg = N * op.Ho * op.Wo;
  const bool vec = (C % 4 == 0) && sg_aligned16(x);
  const SegPlan pl = seg_plan<1>(ctx->num_cus, nseg, (int64_t)kh * kw, C, vec);
  if (!ws || ws_bytes < pl.part_bytes) {
    sg_set_error("sg_avgpool_fwd: workspace %zu < %zu", ws_bytes, pl.part_bytes);
    return SG_EWORKSPACE;
  }
  return seg_reduce_launch(op, pl, nseg, (int64_t)kh * kw, C, (float*)ws, (hipStream_t)stream, "avgpool_fwd");
  });
  return 0;
}

int sg_avgpool_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int kh, int kw, const void* dy,
                   void* dx, int accumulate) {
  SG_CHECK_ARG(ctx && dy && dx, "sg_avgpool_bwd: bad argument");
  SG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && H >= kh && W >= kw, "sg_avgpool_bwd: bad geometry");
  SG_CHECK_ARG((int64_t)N * H * W * C < (1ll << 31), "sg_avgpool_bwd: tensor exceeds 2^31 elements");
  const bool vec = (C % 4 == 0) && sg_aligned16(dy) && sg_aligned16(dx);
  const int V = vec ? 4 : 1;
  const unsigned blocks = ew_blocks((int64_t)N * H * W * (C / V));
  const FastDiv a = make_fastdiv((uint32_t)(C / V)), b = make_fastdiv((uint32_t)W), c = make_fastdiv((uint32_t)H);
  SG_DTYPE_SWITCH(dtype, "sg_avgpool_bwd", {
    if (vec)
      hipLaunchKernelGGL((avgpool_bwd_kernel<4, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, N, H,
                         W, C, kh, kw, accumulate, a, b, c);
    else
      hipLaunchKernelGGL((avgpool_bwd_kernel<1, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, N, H,
                         W, C, kh, kw, accumulate, a, b, c);
  });
  SG_LAUNCH_CHECK("avgpool_bwd_kernel");
  return 0;
}

int sg_upsample_nearest_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int sh, int sw,
                            const void* x, void* y, int y_ld) {
  SG_CHECK_ARG(ctx && x && y, "sg_upsample_nearest_fwd: bad argument");
  SG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && sh > 0 && sw > 0, "sg_upsample_nearest_fwd: bad geometry");
  if (y_ld == 0) y_ld = C;
  SG_CHECK_ARG(y_ld >= C, "sg_upsample_nearest_fwd: y_ld < C");
  SG_CHECK_ARG((int64_t)N * H * sh * W * sw * y_ld < (1ll << 31), "sg_upsample_nearest_fwd: tensor exceeds 2^31 elements");
  const bool vec = (C % 4 == 0) && (y_ld % 4 == 0) && sg_aligned16(x) && sg_aligned16(y);
  const int V = vec ? 4 : 1;
  const unsigned blocks = ew_blocks((int64_t)N * H * sh * W * sw * (C / V));
  const FastDiv a = make_fastdiv((uint32_t)(C / V)), b = make_fastdiv((uint32_t)(W * sw)), c = make_fastdiv((uint32_t)(H * sh));
  SG_DTYPE_SWITCH(dtype, "sg_upsample_nearest_fwd", {
    if (vec)
      hipLaunchKernelGGL((upsample_fwd_kernel<4, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, N, H, W,
                         C, sh, sw, y_ld, a, b, c);
    else
      hipLaunchKernelGGL((upsample_fwd_kernel<1, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, N, H, W,
                         C, sh, sw, y_ld, a, b, c);
  });
  SG_LAUNCH_CHECK("upsample_fwd_kernel");
  return 0;
}

int sg_upsample_nearest_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int H, int W, int C, int sh, int sw,
                            const void* dy, int dy_ld, void* dx, int accumulate) {
  SG_CHECK_ARG(ctx && dy && dx, "sg_upsample_nearest_bwd: bad argument");
  SG_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && sh > 0 && sw > 0, "sg_upsample_nearest_bwd: bad geometry");
  if (dy_ld == 0) dy_ld = C;
  SG_CHECK_ARG(dy_ld >= C, "sg_upsample_nearest_bwd: dy_ld < C");
  SG_CHECK_ARG((int64_t)N * H * sh * W * sw * dy_ld < (1ll << 31), "sg_upsample_nearest_bwd: tensor exceeds 2^31 elements");
  const bool vec = (C % 4 == 0) && (dy_ld % 4 == 0) && sg_aligned16(dy) && sg_aligned16(dx);
  const int V = vec ? 4 : 1;
  const unsigned blocks = ew_blocks((int64_t)N * H * W * (C / V));
  const FastDiv a = make_fastdiv((uint32_t)(C / V)), b = make_fastdiv((uint32_t)W), c = make_fastdiv((uint32_t)H);
  SG_DTYPE_SWITCH(dtype, "sg_upsample_nearest_bwd", {
    if (vec && sh * sw >= 64 && (int64_t)N * H * W < 65536) {
      hipLaunchKernelGGL((upsample_bwd_window_kernel<T>), dim3((unsigned)sg_cdiv(C / 4, 8), (unsigned)(N * H * W)), dim3(256), 0,
                         (hipStream_t)stream, (const T*)dy, dy_ld, (T*)dx, H, W, C, sh, sw, accumulate, b, c,
                         make_fastdiv((uint32_t)sw));
    } else if (vec)
      hipLaunchKernelGGL((upsample_bwd_kernel<4, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)dy, dy_ld, (T*)dx,
                         N, H, W, C, sh, sw, accumulate, a, b, c);
    else
      hipLaunchKernelGGL((upsample_bwd_kernel<1, T>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const T*)dy, dy_ld, (T*)dx,
                         N, H, W, C, sh, sw, accumulate, a, b, c);
  });
  SG_LAUNCH_CHECK("upsample_bwd_kernel");
  return 0;
}

}  // extern "C"
