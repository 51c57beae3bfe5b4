// BatchNormalization (Keras defaults) for NHWC [rows][C]: training forward/backward and inference.
// HBM-bound: forward = one statistics pass (read x) + one apply pass (read x, write y); backward = one
// reduction pass (read x, dy[, y]) + one apply pass.  Statistics use a per-channel pivot (the first row) so
// the single-pass variance  E[(x-K)^2] - E[x-K]^2  does not cancel, and are combined in fp64.
#include "sg_reduce.h"

namespace {

template <typename T>
struct BnStatsOp {
  static constexpr int NOUT = 2;
  const T* __restrict__ x;
  int C;
  int64_t rows;
  float* moving_mean;
  float* moving_var;
  float* save_mean;
  float* save_invstd;
  float momentum, eps;
  int unbiased;

  template <int V>
  __device__ __forceinline__ void accum(int, int64_t r, int c, float (&acc)[2][V]) const {
    float k[V], v[V];
    ldv<V>(x + c, k);  // pivot = row 0
    ldv<V>(x + r * C + c, v);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float d = v[i] - k[i];
      acc[0][i] += d;
      acc[1][i] = fmaf(d, d, acc[1][i]);
    }
  }
  __device__ __forceinline__ void finalize(int, int c, const double (&s)[2]) const {
    const double n = (double)rows;
    const double m1 = s[0] / n;
    const double mean = (double)ld1<T>(x + c) + m1;
    double var = s[1] / n - m1 * m1;
    if (var < 0.0) var = 0.0;
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    const double var_u = (unbiased && rows > 1) ? var * (n / (n - 1.0)) : var;
    moving_mean[c] = (float)((double)moving_mean[c] * momentum + mean * (1.0 - (double)momentum));
    moving_var[c] = (float)((double)moving_var[c] * momentum + var_u * (1.0 - (double)momentum));
  }
};

// MODE: 0 = no fused ReLU, 1 = ReLU mask read from y, 2 = ReLU mask recomputed from x (gamma, beta given).  Compile-time,
// and the per-channel parameters are loaded once per thread (BnCtx): the row loop is two loads (three in mode 1) and
// arithmetic, four rows in flight.
template <int V>
struct BnCtx {
  float mv[V], iv[V], gm[V], bt[V];
};

template <typename T, int MODE>
struct BnBwdOp {
  static constexpr int NOUT = 2;  // sum dy, sum dy * xhat
  const T* __restrict__ x;
  const T* __restrict__ y;
  const T* __restrict__ dy;
  const float* __restrict__ mean;
  const float* __restrict__ invstd;
  const float* __restrict__ gamma;  // with beta: the ReLU mask is recomputed from x instead of read from y
  const float* __restrict__ beta;
  float* dgamma;
  float* dbeta;
  int C;

  template <int V>
  __device__ __forceinline__ BnCtx<V> begin(int, int c) const {
    BnCtx<V> k;
    ldv<V>(mean + c, k.mv);
    ldv<V>(invstd + c, k.iv);
    if constexpr (MODE == 2) {
      ldv<V>(gamma + c, k.gm);
      ldv<V>(beta + c, k.bt);
    }
    return k;
  }
  template <int V>
  __device__ __forceinline__ void accum(int, int64_t r, int c, float (&acc)[2][V], const BnCtx<V>& k) const {
    float xv[V], gv[V];
    ldv<V>(x + r * C + c, xv);
    ldv<V>(dy + r * C + c, gv);
    if constexpr (MODE == 2) {
#pragma unroll
      for (int i = 0; i < V; ++i) gv[i] = fmaf((xv[i] - k.mv[i]) * k.iv[i], k.gm[i], k.bt[i]) > 0.f ? gv[i] : 0.f;
    } else if constexpr (MODE == 1) {
      float yv[V];
      ldv<V>(y + r * C + c, yv);
#pragma unroll
      for (int i = 0; i < V; ++i) gv[i] = yv[i] > 0.f ? gv[i] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      acc[0][i] += gv[i];
      acc[1][i] = fmaf(gv[i], (xv[i] - k.mv[i]) * k.iv[i], acc[1][i]);
    }
  }
  __device__ __forceinline__ void finalize(int, int c, const double (&s)[2]) const {
    dbeta[c] = (float)s[0];
    dgamma[c] = (float)s[1];
  }
};

template <int V, typename T>
__global__ void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, T* __restrict__ y, int64_t rows, int C, int relu,
                                float eps, int infer, FastDiv fd_cv) {
  const uint32_t cv = C / V;
  const uint32_t total = (uint32_t)(rows * cv);
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = fd_div(i, fd_cv);
    const int c = (int)(i - (uint32_t)r * cv) * V;
    float xv[V], mv[V], iv[V], gv[V], bv[V], o[V];
    ldv<V>(x + r * C + c, xv);
    ldv<V>(mean + c, mv);
    ldv<V>(invstd + c, iv);  // inference: this is the moving variance
    ldv<V>(gamma + c, gv);
    ldv<V>(beta + c, bv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float is = infer ? rsqrtf(iv[k] + eps) : iv[k];
      float t = fmaf((xv[k] - mv[k]) * is, gv[k], bv[k]);  // the backward re-evaluates exactly this for the ReLU mask
      if (relu) t = fmaxf(t, 0.f);
      o[k] = t;
    }
    stv<V>(y + r * C + c, o);
  }
}

// MODE as in BnBwdOp (compile-time: no load behind a branch)
template <int V, typename T, int MODE>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                    const T* __restrict__ dy, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ dgamma,
                                    const float* __restrict__ dbeta, T* __restrict__ dx, int64_t rows, int C,
                                    FastDiv fd_cv) {
  const uint32_t cv = C / V;
  const uint32_t total = (uint32_t)(rows * cv);
  const uint32_t stride = gridDim.x * blockDim.x;
  const float inv_n = 1.0f / (float)rows;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = fd_div(i, fd_cv);
    const int c = (int)(i - (uint32_t)r * cv) * V;
    float xv[V], gv[V], mv[V], iv[V], gam[V], dg[V], db[V], o[V];
    ldv<V>(x + r * C + c, xv);
    ldv<V>(dy + r * C + c, gv);
    ldv<V>(mean + c, mv);
    ldv<V>(invstd + c, iv);
    ldv<V>(gamma + c, gam);
    ldv<V>(dgamma + c, dg);
    ldv<V>(dbeta + c, db);
    if constexpr (MODE == 2) {
      float bt[V];
      ldv<V>(beta + c, bt);
#pragma unroll
      for (int k = 0; k < V; ++k) gv[k] = fmaf((xv[k] - mv[k]) * iv[k], gam[k], bt[k]) > 0.f ? gv[k] : 0.f;
    } else if constexpr (MODE == 1) {
      float yv[V];
      ldv<V>(y + r * C + c, yv);
#pragma unroll
      for (int k = 0; k < V; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xh = (xv[k] - mv[k]) * iv[k];
      o[k] = gam[k] * iv[k] * (gv[k] - db[k] * inv_n - xh * dg[k] * inv_n);
    }
    stv<V>(dx + r * C + c, o);
  }
}

// Column-block forms of the two kernels above (round 3): a thread keeps ONE chunk of V channels and walks down the rows, so
// the per-channel parameters are loaded once per thread instead of once per element.  With bf16 storage a 16-byte access is
// 8 channels and the flat kernels issued 10 (apply) / 14 (backward) 16-byte parameter loads per 16 bytes of tensor: 20.8 /
// 29.8 us for the 24 MB tensors of the middle flow against a copy's 11.8 us (profiles/r03_bw_bench.txt) - bound by the
// load issue, not by bytes.  The flat index space is kept (a wave touches 1 KB of consecutive addresses: 16 chunk lanes x 16
// row lanes with 256-byte runs measured 3.4 instead of 4.6 TB/s in fp32), but the number of threads is a multiple of the
// chunks per row, so a thread's grid stride is a whole number of rows and its channel chunk never changes.  Four (backward:
// two) rows in flight per thread, taken from consecutive periods so that the workgroups in flight sweep one window of the tensor.  Same per-element expressions, so results are bit-identical to the flat kernels.
template <int V, typename T>
__global__ __launch_bounds__(256) void bn_apply_cols_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y, int64_t rows,
                                                            int C, int relu, float eps, int infer, int prow, FastDiv fd_cv) {
  // blockIdx.x: block within a period of prow whole rows; blockIdx.y: group of four consecutive periods
  const uint32_t i0 = blockIdx.x * 256u + threadIdx.x, cv = (uint32_t)(C / V);
  const uint32_t r0 = fd_div(i0, fd_cv);
  const int c = (int)(i0 - r0 * cv) * V;
  float mv[V], is[V], gv[V], bv[V];
  ldv<V>(mean + c, mv);
  ldv<V>(invstd + c, is);  // inference: this is the moving variance
  ldv<V>(gamma + c, gv);
  ldv<V>(beta + c, bv);
  if (infer) {
#pragma unroll
    for (int k = 0; k < V; ++k) is[k] = rsqrtf(is[k] + eps);
  }
  auto one = [&](const float (&xv)[V], int64_t r) {
    float o[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float t = fmaf((xv[k] - mv[k]) * is[k], gv[k], bv[k]);  // the backward re-evaluates exactly this for the ReLU mask
      if (relu) t = fmaxf(t, 0.f);
      o[k] = t;
    }
    stv<V>(y + r * C + c, o);
  };
  const int64_t stride = prow, gstep = (int64_t)gridDim.y * 4 * prow;
  for (int64_t r = (int64_t)blockIdx.y * 4 * prow + r0; r < rows; r += gstep) {
    if (r + 3 * stride < rows) {
      float x0[V], x1[V], x2[V], x3[V];
      ldv<V>(x + r * C + c, x0);
      ldv<V>(x + (r + stride) * C + c, x1);
      ldv<V>(x + (r + 2 * stride) * C + c, x2);
      ldv<V>(x + (r + 3 * stride) * C + c, x3);
      one(x0, r); one(x1, r + stride); one(x2, r + 2 * stride); one(x3, r + 3 * stride);
    } else {
      for (int64_t q = r; q < rows; q += stride) {
        float x0[V];
        ldv<V>(x + q * C + c, x0);
        one(x0, q);
      }
    }
  }
}

template <int V, typename T, int MODE>
__global__ __launch_bounds__(256) void bn_bwd_apply_cols_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                                const T* __restrict__ dy, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ dgamma,
                                                                const float* __restrict__ dbeta, T* __restrict__ dx, int64_t rows,
                                                                int C, int prow, FastDiv fd_cv) {
  const uint32_t i0 = blockIdx.x * 256u + threadIdx.x, cv = (uint32_t)(C / V);
  const uint32_t r0 = fd_div(i0, fd_cv);
  const int c = (int)(i0 - r0 * cv) * V;
  float mv[V], iv[V], gam[V], dg[V], db[V], bt[V];
  ldv<V>(mean + c, mv);
  ldv<V>(invstd + c, iv);
  ldv<V>(gamma + c, gam);
  ldv<V>(dgamma + c, dg);
  ldv<V>(dbeta + c, db);
  if constexpr (MODE == 2) ldv<V>(beta + c, bt);
  const float inv_n = 1.0f / (float)rows;
  auto one = [&](const float (&xv)[V], float (&gv)[V], const float (&yv)[V], int64_t r) {
    float o[V];
    if constexpr (MODE == 2) {
#pragma unroll
      for (int k = 0; k < V; ++k) gv[k] = fmaf((xv[k] - mv[k]) * iv[k], gam[k], bt[k]) > 0.f ? gv[k] : 0.f;
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int k = 0; k < V; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xh = (xv[k] - mv[k]) * iv[k];
      o[k] = gam[k] * iv[k] * (gv[k] - db[k] * inv_n - xh * dg[k] * inv_n);
    }
    stv<V>(dx + r * C + c, o);
  };
  // (two rows a whole grid apart: measured 2 - 5 % faster here than two adjacent periods, the opposite of the forward kernel)
  const int64_t stride = (int64_t)gridDim.y * prow, gstep = 2 * stride;
  for (int64_t r = (int64_t)blockIdx.y * prow + r0; r < rows; r += gstep) {
    if (r + stride < rows) {
      float x0[V], x1[V], g0[V], g1[V], y0[V], y1[V];
      ldv<V>(x + r * C + c, x0);
      ldv<V>(x + (r + stride) * C + c, x1);
      ldv<V>(dy + r * C + c, g0);
      ldv<V>(dy + (r + stride) * C + c, g1);
      if constexpr (MODE == 1) {
        ldv<V>(y + r * C + c, y0);
        ldv<V>(y + (r + stride) * C + c, y1);
      }
      one(x0, g0, y0, r);
      one(x1, g1, y1, r + stride);
    } else {
      float x0[V], g0[V], y0[V];
      ldv<V>(x + r * C + c, x0);
      ldv<V>(dy + r * C + c, g0);
      if constexpr (MODE == 1) ldv<V>(y + r * C + c, y0);
      one(x0, g0, y0, r);
    }
  }
}

// y = [relu]( f_a(a) + f_b(b) ), f = BatchNormalization's apply (training: saved mean / invstd; inference: moving mean /
// variance) for an operand whose parameter pointers are given, the identity otherwise: the residual add of an Xception block
// applies the normalisation of the branch (and of the 1x1 shortcut) it sums, so that tensor is never written and read again
// (sg_add2_bn; one of BatchNormalization's two forward passes for those layers).  Column-stationary like bn_apply_cols_kernel;
// the same expression, so with fp32 storage the result has the bits of bn_apply + add_n.
struct Add2BnArgs {
  const float* mean[2];
  const float* invstd[2];
  const float* gamma[2];
  const float* beta[2];
  int relu_op[2];   // the operand's BatchNormalization has a fused ReLU (Res34's blocks activate before they add)
};

template <int V, typename T>
__global__ __launch_bounds__(256) void add2_bn_kernel(const T* __restrict__ a, const T* __restrict__ b, const Add2BnArgs q,
                                                      T* __restrict__ y, int64_t rows, int C, int relu, float eps, int infer,
                                                      int prow, FastDiv fd_cv) {
  const uint32_t i0 = blockIdx.x * 256u + threadIdx.x, cv = (uint32_t)(C / V);
  const uint32_t r0 = fd_div(i0, fd_cv);
  const int c = (int)(i0 - r0 * cv) * V;
  float mv[2][V], is[2][V], gv[2][V], bv[2][V];
  const bool on[2] = {q.mean[0] != nullptr, q.mean[1] != nullptr};
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    if (on[o]) {
      ldv<V>(q.mean[o] + c, mv[o]);
      ldv<V>(q.invstd[o] + c, is[o]);
      ldv<V>(q.gamma[o] + c, gv[o]);
      ldv<V>(q.beta[o] + c, bv[o]);
      if (infer) {
#pragma unroll
        for (int k = 0; k < V; ++k) is[o][k] = rsqrtf(is[o][k] + eps);
      }
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) { mv[o][k] = 0.f; is[o][k] = 1.f; gv[o][k] = 1.f; bv[o][k] = 0.f; }
    }
  }
  auto one = [&](const float (&xa)[V], const float (&xb)[V], int64_t r) {
    float o[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float ta = on[0] ? fmaf((xa[k] - mv[0][k]) * is[0][k], gv[0][k], bv[0][k]) : xa[k];
      float tb = on[1] ? fmaf((xb[k] - mv[1][k]) * is[1][k], gv[1][k], bv[1][k]) : xb[k];
      if (q.relu_op[0]) ta = fmaxf(ta, 0.f);
      if (q.relu_op[1]) tb = fmaxf(tb, 0.f);
      float t = ta + tb;
      if (relu) t = fmaxf(t, 0.f);
      o[k] = t;
    }
    stv<V>(y + r * C + c, o);
  };
  const int64_t stride = prow, gstep = (int64_t)gridDim.y * 2 * prow;
  for (int64_t r = (int64_t)blockIdx.y * 2 * prow + r0; r < rows; r += gstep) {
    if (r + stride < rows) {
      float a0[V], a1[V], b0[V], b1[V];
      ldv<V>(a + r * C + c, a0);
      ldv<V>(a + (r + stride) * C + c, a1);
      ldv<V>(b + r * C + c, b0);
      ldv<V>(b + (r + stride) * C + c, b1);
      one(a0, b0, r);
      one(a1, b1, r + stride);
    } else {
      float a0[V], b0[V];
      ldv<V>(a + r * C + c, a0);
      ldv<V>(b + r * C + c, b0);
      one(a0, b0, r);
    }
  }
}

// grid of the column-stationary kernels: x = the blocks of one period (b0 x 256 threads = prow whole rows), y = groups of
// `unroll` consecutive periods; a thread walks groups gridDim.y apart.  Returns false when no such grid of a sensible size
// exists (the flat kernels take the launch).
inline bool bn_cols_grid(int num_cus, int64_t rows, int cv, int unroll, int& prow, dim3& grid) {
  int g = 256, a = cv;
  while (a) { const int t_ = g % a; g = a; a = t_; }   // g = gcd(256, cv)
  const int64_t b0 = cv / g;
  if (b0 > 16384) return false;
  prow = 256 / g;
  int64_t k = sg_cdiv((int64_t)8 * num_cus, b0);
  const int64_t maxk = sg_cdiv(rows, (int64_t)unroll * prow);
  if (k > maxk) k = maxk;
  if (k > 65535) k = 65535;
  if (k < 1) k = 1;
  grid = dim3((unsigned)b0, (unsigned)k);
  return true;
}

inline unsigned ew_blocks(int64_t total) {
  int64_t b = sg_cdiv(total, 256);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename T>
int launch_bn_apply(hipStream_t st, bool vec, const T* x, const float* mean, const float* invstd, const float* gamma,
                    const float* beta, T* y, int64_t rows, int C, int relu, float eps, int infer, int num_cus) {
  const bool wide = vec && sizeof(T) == 2 && C % 8 == 0;  // bf16: 8 channels = one 16-byte access
  const int V = wide ? 8 : (vec ? 4 : 1);
  static const int cols_on = getenv("SG_BN_COLS") ? atoi(getenv("SG_BN_COLS")) : 1;
  int prow = 0;
  dim3 cgrid;
  if (vec && cols_on && bn_cols_grid(num_cus, rows, C / V, 4, prow, cgrid)) {
    const FastDiv fd = make_fastdiv((uint32_t)(C / V));
    if (wide)
      hipLaunchKernelGGL((bn_apply_cols_kernel<8, T>), cgrid, dim3(256), 0, st, x, mean, invstd, gamma, beta, y, rows, C, relu, eps,
                         infer, prow, fd);
    else
      hipLaunchKernelGGL((bn_apply_cols_kernel<4, T>), cgrid, dim3(256), 0, st, x, mean, invstd, gamma, beta, y, rows, C, relu, eps,
                         infer, prow, fd);
    SG_LAUNCH_CHECK("bn_apply_cols_kernel");
    return 0;
  }
  const unsigned blocks = ew_blocks(rows * (C / V));
  if (wide)
    hipLaunchKernelGGL((bn_apply_kernel<8, T>), dim3(blocks), dim3(256), 0, st, x, mean, invstd, gamma, beta, y, rows, C, relu, eps,
                       infer, make_fastdiv((uint32_t)(C / V)));
  else if (vec)
    hipLaunchKernelGGL((bn_apply_kernel<4, T>), dim3(blocks), dim3(256), 0, st, x, mean, invstd, gamma, beta, y, rows, C, relu, eps,
                       infer, make_fastdiv((uint32_t)(C / V)));
  else
    hipLaunchKernelGGL((bn_apply_kernel<1, T>), dim3(blocks), dim3(256), 0, st, x, mean, invstd, gamma, beta, y, rows, C, relu, eps,
                       infer, make_fastdiv((uint32_t)(C / V)));
  SG_LAUNCH_CHECK("bn_apply_kernel");
  return 0;
}

}  // namespace

// dx of a training-mode BatchNormalization from its finished column sums (dbeta = sum g, dgamma = sum g * xhat): the apply
// pass of sg_bn_train_bwd, also entered on its own by sg_bn_train_bwd_apply
template <typename T>
static void bn_bwd_apply_launch(sg_ctx* ctx, hipStream_t st, bool vec, int64_t rows, int C, const void* x, const void* y,
                                const void* dy, const void* gamma, const void* beta, const void* save_mean,
                                const void* save_invstd, void* dx, const void* dgamma, const void* dbeta, int relu) {
    const bool wide = vec && sizeof(T) == 2 && C % 8 == 0;
    const int V = wide ? 8 : (vec ? 4 : 1);
    const unsigned blocks = ew_blocks(rows * (C / V));
    const int mode = !relu ? 0 : (beta ? 2 : 1);
    static const int cols_on = getenv("SG_BN_COLS") ? atoi(getenv("SG_BN_COLS")) : 1;
    auto apply = [&](auto vt, auto mt) {
      constexpr int V_ = decltype(vt)::value, M_ = decltype(mt)::value;
      if constexpr (V_ > 1) {
        int prow = 0;
        dim3 cgrid;
        if (cols_on && bn_cols_grid(ctx->num_cus, rows, C / V_, 1, prow, cgrid)) {
          hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<V_, T, M_>), cgrid, dim3(256), 0, st, (const T*)x, (const T*)y,
                             (const T*)dy, (const float*)save_mean, (const float*)save_invstd, (const float*)gamma,
                             (const float*)beta, (const float*)dgamma, (const float*)dbeta, (T*)dx, rows, C, prow,
                             make_fastdiv((uint32_t)(C / V_)));
          return;
        }
      }
      hipLaunchKernelGGL((bn_bwd_apply_kernel<V_, T, M_>), dim3(blocks), dim3(256), 0, st, (const T*)x, (const T*)y, (const T*)dy,
                         (const float*)save_mean, (const float*)save_invstd, (const float*)gamma, (const float*)beta,
                         (const float*)dgamma, (const float*)dbeta, (T*)dx, rows, C, make_fastdiv((uint32_t)(C / V)));
    };
    auto apply_v = [&](auto vt) {
      if (mode == 0) apply(vt, std::integral_constant<int, 0>{});
      else if (mode == 1) apply(vt, std::integral_constant<int, 1>{});
      else apply(vt, std::integral_constant<int, 2>{});
    };
    if (wide) apply_v(std::integral_constant<int, 8>{});
    else if (vec) apply_v(std::integral_constant<int, 4>{});
    else apply_v(std::integral_constant<int, 1>{});
}

extern "C" {

size_t sg_bn_ws_bytes(const sg_ctx* ctx, int64_t rows, int C) {
  if (!ctx) return 0;
  SegPlan pl = seg_plan<2>(ctx->num_cus, 1, rows, C, true);
  // scalar plan can only be smaller or equal in part_bytes (same formula, S differs): take the max of both
  SegPlan pls = seg_plan<2>(ctx->num_cus, 1, rows, C, false);
  SegPlan plw = seg_plan<2>(ctx->num_cus, 1, rows, C, true, true);
  size_t m = pl.part_bytes > pls.part_bytes ? pl.part_bytes : pls.part_bytes;
  if (plw.part_bytes > m) m = plw.part_bytes;
  return m + 256;
}

int sg_bn_train_fwd(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* x, const void* gamma,
                    const void* beta, void* moving_mean, void* moving_var, void* y, void* save_mean,
                    void* save_invstd, float momentum, float eps, int relu, int unbiased_update, void* ws,
                    size_t ws_bytes) {
  SG_CHECK_ARG(ctx && (dtype == SG_F32 || dtype == SG_BF16), "sg_bn_train_fwd: bad ctx/dtype");
  SG_CHECK_ARG(rows > 0 && C > 0 && x && gamma && beta && moving_mean && moving_var && y && save_mean && save_invstd,
               "sg_bn_train_fwd: bad argument");
  SG_CHECK_ARG(rows * C < (1ll << 31), "sg_bn_train_fwd: tensor exceeds 2^31 elements");
  const bool vec = (C % 4 == 0) && sg_aligned16(x) && sg_aligned16(y);
  const SegPlan pl = seg_plan<2>(ctx->num_cus, 1, rows, C, vec, dtype == SG_BF16);
  if (!ws || ws_bytes < pl.part_bytes) {
    sg_set_error("sg_bn_train_fwd: workspace %zu < %zu", ws_bytes, pl.part_bytes);
    return SG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  SG_DTYPE_SWITCH(dtype, "sg_bn_train_fwd", {
    BnStatsOp<T> op;
    op.x = (const T*)x; op.C = C; op.rows = rows;
    op.moving_mean = (float*)moving_mean; op.moving_var = (float*)moving_var;
    op.save_mean = (float*)save_mean; op.save_invstd = (float*)save_invstd;
    op.momentum = momentum; op.eps = eps; op.unbiased = unbiased_update;
    int rc = seg_reduce_launch(op, pl, 1, rows, C, (float*)ws, st, "bn_stats");
    if (rc) return rc;
    return launch_bn_apply<T>(st, vec, (const T*)x, (const float*)save_mean, (const float*)save_invstd, (const float*)gamma,
                              (const float*)beta, (T*)y, rows, C, relu, eps, 0, ctx->num_cus);
  });
  return 0;
}

int sg_bn_train_bwd(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* x, const void* y,
                    const void* dy, const void* gamma, const void* beta, const void* save_mean, const void* save_invstd,
                    void* dx, void* dgamma, void* dbeta, int relu, void* ws, size_t ws_bytes) {
  SG_CHECK_ARG(ctx && (dtype == SG_F32 || dtype == SG_BF16), "sg_bn_train_bwd: bad ctx/dtype");
  SG_CHECK_ARG(rows > 0 && C > 0 && x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta,
               "sg_bn_train_bwd: bad argument");
  SG_CHECK_ARG(!relu || y || beta, "sg_bn_train_bwd: relu set but neither y nor beta given");
  SG_CHECK_ARG(rows * C < (1ll << 31), "sg_bn_train_bwd: tensor exceeds 2^31 elements");
  const bool vec = (C % 4 == 0) && sg_aligned16(x) && sg_aligned16(dy) && sg_aligned16(dx) && (!relu || sg_aligned16(y));
  const SegPlan pl = seg_plan<2>(ctx->num_cus, 1, rows, C, vec, dtype == SG_BF16);
  if (!ws || ws_bytes < pl.part_bytes) {
    sg_set_error("sg_bn_train_bwd: workspace %zu < %zu", ws_bytes, pl.part_bytes);
    return SG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  SG_DTYPE_SWITCH(dtype, "sg_bn_train_bwd", {
    auto reduce = [&](auto op) -> int {
      op.x = (const T*)x; op.y = (const T*)y; op.dy = (const T*)dy;
      op.mean = (const float*)save_mean; op.invstd = (const float*)save_invstd;
      op.gamma = (const float*)gamma; op.beta = (const float*)beta;
      op.dgamma = (float*)dgamma; op.dbeta = (float*)dbeta; op.C = C;
      return seg_reduce_launch(op, pl, 1, rows, C, (float*)ws, st, "bn_bwd_reduce");
    };
    int rc = !relu ? reduce(BnBwdOp<T, 0>{}) : (beta ? reduce(BnBwdOp<T, 2>{}) : reduce(BnBwdOp<T, 1>{}));
    if (rc) return rc;
    bn_bwd_apply_launch<T>(ctx, st, vec, rows, C, x, y, dy, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, relu);
    SG_LAUNCH_CHECK("bn_bwd_apply_kernel");
  });
  return 0;
}

int sg_bn_train_bwd_apply(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* x, const void* dy,
                          const void* gamma, const void* beta, const void* save_mean, const void* save_invstd,
                          const void* dgamma, const void* dbeta, void* dx, int relu) {
  SG_CHECK_ARG(ctx && (dtype == SG_F32 || dtype == SG_BF16), "sg_bn_train_bwd_apply: bad ctx/dtype");
  SG_CHECK_ARG(rows > 0 && C > 0 && x && dy && gamma && save_mean && save_invstd && dx && dgamma && dbeta,
               "sg_bn_train_bwd_apply: bad argument");
  SG_CHECK_ARG(!relu || beta, "sg_bn_train_bwd_apply: relu needs beta (the mask is recomputed from x)");
  SG_CHECK_ARG(rows * C < (1ll << 31), "sg_bn_train_bwd_apply: tensor exceeds 2^31 elements");
  const bool vec = (C % 4 == 0) && sg_aligned16(x) && sg_aligned16(dy) && sg_aligned16(dx);
  SG_DTYPE_SWITCH(dtype, "sg_bn_train_bwd_apply", {
    bn_bwd_apply_launch<T>(ctx, (hipStream_t)stream, vec, rows, C, x, nullptr, dy, gamma, beta, save_mean, save_invstd, dx, dgamma,
                           dbeta, relu);
  });
  SG_LAUNCH_CHECK("bn_bwd_apply_kernel");
  return 0;
}

int sg_bn_apply(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* x, const void* gamma,
                const void* beta, const void* mean, const void* invstd, void* y, int relu) {
  SG_CHECK_ARG(ctx && (dtype == SG_F32 || dtype == SG_BF16), "sg_bn_apply: bad ctx/dtype");
  SG_CHECK_ARG(rows > 0 && C > 0 && x && gamma && beta && mean && invstd && y, "sg_bn_apply: bad argument");
  SG_CHECK_ARG(rows * C < (1ll << 31), "sg_bn_apply: tensor exceeds 2^31 elements");
  const bool vec = (C % 4 == 0) && sg_aligned16(x) && sg_aligned16(y);
  SG_DTYPE_SWITCH(dtype, "sg_bn_apply", {
    return launch_bn_apply<T>((hipStream_t)stream, vec, (const T*)x, (const float*)mean, (const float*)invstd, (const float*)gamma,
                              (const float*)beta, (T*)y, rows, C, relu, 0.f, 0, ctx->num_cus);
  });
  return 0;
}

int sg_add2_bn(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* a, const void* b, const void* a_mean,
               const void* a_invstd, const void* a_gamma, const void* a_beta, const void* b_mean, const void* b_invstd,
               const void* b_gamma, const void* b_beta, void* y, int relu, int infer, float eps, int a_relu, int b_relu) {
  SG_CHECK_ARG(ctx && (dtype == SG_F32 || dtype == SG_BF16), "sg_add2_bn: bad ctx/dtype");
  SG_CHECK_ARG(rows > 0 && C > 0 && a && b && y, "sg_add2_bn: bad argument");
  SG_CHECK_ARG((a_mean != nullptr) == (a_invstd != nullptr && a_gamma != nullptr && a_beta != nullptr) &&
                   (b_mean != nullptr) == (b_invstd != nullptr && b_gamma != nullptr && b_beta != nullptr),
               "sg_add2_bn: an operand's four BatchNormalization parameters come together or not at all");
  if (!((C % 4 == 0) && sg_aligned16(a) && sg_aligned16(b) && sg_aligned16(y))) {
    sg_set_error("sg_add2_bn: needs C %% 4 == 0 and 16-byte aligned tensors; apply the BatchNormalization and add instead");
    return SG_EUNSUPPORTED;
  }
  Add2BnArgs q;
  q.mean[0] = (const float*)a_mean; q.invstd[0] = (const float*)a_invstd; q.gamma[0] = (const float*)a_gamma; q.beta[0] = (const float*)a_beta;
  q.mean[1] = (const float*)b_mean; q.invstd[1] = (const float*)b_invstd; q.gamma[1] = (const float*)b_gamma; q.beta[1] = (const float*)b_beta;
  q.relu_op[0] = (a_mean && a_relu) ? 1 : 0;
  q.relu_op[1] = (b_mean && b_relu) ? 1 : 0;
  // The kernel indexes with 32 bits: a tensor of 2^31 elements or more (the BatchNormalization in front of this add has
  // already handed its RAW input on, so there is no unfused form to fall back to) is walked in row chunks below that, each
  // an even number of rows so that a chunk starts 16-byte aligned with bf16 storage too.  Element-wise: same bits.
  const int64_t chunk_rows = rows * C < (1ll << 31) ? rows : (((1ll << 31) - 1) / C) & ~1ll;
  SG_CHECK_ARG(chunk_rows > 0, "sg_add2_bn: a single row exceeds 2^31 elements");
  SG_DTYPE_SWITCH(dtype, "sg_add2_bn", {
    const bool wide = sizeof(T) == 2 && C % 8 == 0;
    const int V = wide ? 8 : 4;
    for (int64_t r0 = 0; r0 < rows; r0 += chunk_rows) {
      const int64_t nr = rows - r0 < chunk_rows ? rows - r0 : chunk_rows;
      const T* ca = (const T*)a + r0 * C;
      const T* cb = (const T*)b + r0 * C;
      T* cy = (T*)y + r0 * C;
      int prow = 0;
      dim3 grid;
      if (!bn_cols_grid(ctx->num_cus, nr, C / V, 2, prow, grid)) {
        sg_set_error("sg_add2_bn: no column-stationary grid for C = %d", C);
        return SG_EUNSUPPORTED;
      }
      const FastDiv fd = make_fastdiv((uint32_t)(C / V));
      if (wide)
        hipLaunchKernelGGL((add2_bn_kernel<8, T>), grid, dim3(256), 0, (hipStream_t)stream, ca, cb, q, cy, nr, C, relu, eps, infer, prow, fd);
      else
        hipLaunchKernelGGL((add2_bn_kernel<4, T>), grid, dim3(256), 0, (hipStream_t)stream, ca, cb, q, cy, nr, C, relu, eps, infer, prow, fd);
    }
  });
  SG_LAUNCH_CHECK("add2_bn_kernel");
  return 0;
}

int sg_bn_infer(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* x, const void* gamma,
                const void* beta, const void* moving_mean, const void* moving_var, void* y, float eps, int relu) {
  SG_CHECK_ARG(ctx && (dtype == SG_F32 || dtype == SG_BF16), "sg_bn_infer: bad ctx/dtype");
  SG_CHECK_ARG(rows > 0 && C > 0 && x && gamma && beta && moving_mean && moving_var && y, "sg_bn_infer: bad argument");
  SG_CHECK_ARG(rows * C < (1ll << 31), "sg_bn_infer: tensor exceeds 2^31 elements");
  const bool vec = (C % 4 == 0) && sg_aligned16(x) && sg_aligned16(y);
  SG_DTYPE_SWITCH(dtype, "sg_bn_infer", {
    return launch_bn_apply<T>((hipStream_t)stream, vec, (const T*)x, (const float*)moving_mean, (const float*)moving_var,
                              (const float*)gamma, (const float*)beta, (T*)y, rows, C, relu, eps, 1, ctx->num_cus);
  });
  return 0;
}

}  // extern "C"
