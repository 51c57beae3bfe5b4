// HBM-bound element-wise and gate kernels of the hot path: activations, n-ary add, channel-slice copies
// (concat), 2-class / branch softmax, broadcast multiplies, and the fused scSE / BAM combines with their
// backward "squeeze" reductions (per-pixel reductions over C run as sub-wave shuffles; per-channel reductions
// over H*W use the segmented reducer in sg_reduce.h).
#include "sg_reduce.h"

namespace {

inline unsigned ew_blocks(int64_t total) {
  int64_t b = sg_cdiv(total, 256);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

// ------------------------------------------------------------------------------------------- activations
template <int V, typename T>
__global__ void act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n, int act) {
  const int64_t nv = n / V, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float v[V], o[V];
    ldv<V>(x + i * V, v);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = act == SG_ACT_RELU ? fmaxf(v[k], 0.f) : sigmoidf_(v[k]);
    stv<V>(y + i * V, o);
  }
}

template <int V, typename T>
__global__ void act_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx,
                               int64_t n, int act, int accumulate) {
  const int64_t nv = n / V, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float yv[V], g[V], o[V];
    ldv<V>(y + i * V, yv);
    ldv<V>(dy + i * V, g);
    if (accumulate) ldv<V>(dx + i * V, o);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float d = act == SG_ACT_RELU ? (yv[k] > 0.f ? g[k] : 0.f) : g[k] * yv[k] * (1.f - yv[k]);
      o[k] = accumulate ? o[k] + d : d;
    }
    stv<V>(dx + i * V, o);
  }
}

struct AddNArgs {
  const void* xs[8];
  int k;
};

template <int V, typename T>
__global__ void add_n_kernel(const AddNArgs a, T* __restrict__ y, int64_t n, int relu) {
  const int64_t nv = n / V, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += stride) {
    float s[V];
    ldv<V>((const T*)a.xs[0] + i * V, s);
    for (int j = 1; j < a.k; ++j) {
      float t[V];
      ldv<V>((const T*)a.xs[j] + i * V, t);
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] += t[k];
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] = fmaxf(s[k], 0.f);
    }
    stv<V>(y + i * V, s);
  }
}

template <int V, typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, int src_ld, int src_off, T* __restrict__ dst,
                                     int dst_ld, int dst_off, int64_t rows, int C, int accumulate, FastDiv fd_cv) {
  const uint32_t cv = C / V;
  const uint32_t total = (uint32_t)(rows * cv), stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = fd_div(i, fd_cv);
    const int c = (int)(i - (uint32_t)r * cv) * V;
    float v[V];
    ldv<V>(src + r * src_ld + src_off + c, v);
    T* d = dst + r * dst_ld + dst_off + c;
    if (accumulate) {
      float o[V];
      ldv<V>(d, o);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] += o[k];
    }
    stv<V>(d, v);
  }
}

// ---------------------------------------------------------------------------------------------- softmax
__global__ void softmax2_fwd_kernel(const float* __restrict__ z, float* __restrict__ p, int64_t rows) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
    const float2 v = *reinterpret_cast<const float2*>(z + 2 * i);
    const float m = fmaxf(v.x, v.y);
    const float e0 = expf(v.x - m), e1 = expf(v.y - m);
    const float inv = 1.0f / (e0 + e1);
    float2 o;
    o.x = e0 * inv;
    o.y = e1 * inv;
    *reinterpret_cast<float2*>(p + 2 * i) = o;
  }
}

__global__ void softmax2_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp, float* __restrict__ dz,
                                    int64_t rows) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += stride) {
    const float2 pv = *reinterpret_cast<const float2*>(p + 2 * i);
    const float2 g = *reinterpret_cast<const float2*>(dp + 2 * i);
    const float dot = g.x * pv.x + g.y * pv.y;
    float2 o;
    o.x = pv.x * (g.x - dot);
    o.y = pv.y * (g.y - dot);
    *reinterpret_cast<float2*>(dz + 2 * i) = o;
  }
}

// z[N][B][C]: softmax over B for every (n, c)
template <typename T>
__global__ void softmax_branch_fwd_kernel(const T* __restrict__ z, T* __restrict__ p, int N, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  const T* zp = z + (int64_t)n * B * C + c;
  float m = ld1<T>(zp);
  for (int b = 1; b < B; ++b) m = fmaxf(m, ld1<T>(zp + (int64_t)b * C));
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += expf(ld1<T>(zp + (int64_t)b * C) - m);
  const float inv = 1.0f / s;
  for (int b = 0; b < B; ++b) st1<T>(p + (int64_t)n * B * C + (int64_t)b * C + c, expf(ld1<T>(zp + (int64_t)b * C) - m) * inv);
}

template <typename T>
__global__ void softmax_branch_bwd_kernel(const T* __restrict__ p, const T* __restrict__ dp,
                                          T* __restrict__ dz, int N, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i - n * C;
  const int64_t base = (int64_t)n * B * C + c;
  float dot = 0.f;
  for (int b = 0; b < B; ++b) dot += ld1<T>(p + base + (int64_t)b * C) * ld1<T>(dp + base + (int64_t)b * C);
  for (int b = 0; b < B; ++b)
    st1<T>(dz + base + (int64_t)b * C, ld1<T>(p + base + (int64_t)b * C) * (ld1<T>(dp + base + (int64_t)b * C) - dot));
}

// -------------------------------------------------------------------------- generic (row, c) element-wise
// f(row, n, c, V) -> writes; rows = N*HW
template <class F, int V>
__global__ void rowcol_kernel(const F f, int64_t rows, int C, FastDiv fd_cv, FastDiv fd_hw) {
  const uint32_t cv = C / V;
  const uint32_t total = (uint32_t)(rows * cv), stride = gridDim.x * blockDim.x;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const uint32_t r = fd_div(i, fd_cv);
    const int c = (int)(i - r * cv) * V;
    const uint32_t n = fd_div(r, fd_hw);
    f.template apply<V>((int64_t)r, (int)n, c);
  }
}

template <class F>
int launch_rowcol(const F& f, int64_t rows, int64_t HW, int C, bool vec, hipStream_t st, const char* name) {
  const int V = vec ? 4 : 1;
  const unsigned blocks = ew_blocks(rows * (C / V));
  if (vec)
    hipLaunchKernelGGL((rowcol_kernel<F, 4>), dim3(blocks), dim3(256), 0, st, f, rows, C, make_fastdiv((uint32_t)(C / 4)),
                       make_fastdiv((uint32_t)HW));
  else
    hipLaunchKernelGGL((rowcol_kernel<F, 1>), dim3(blocks), dim3(256), 0, st, f, rows, C, make_fastdiv((uint32_t)C),
                       make_fastdiv((uint32_t)HW));
  SG_LAUNCH_CHECK(name);
  return 0;
}

// ---------------------------------------------------------------- per-pixel reductions over C (sub-wave)
// Each row is handled by G lanes (G = power of two <= 64) that stride over V-wide channel chunks; the op may
// also write per-element outputs (dx) while it has the row in registers.
template <class Op, int V>
__global__ __launch_bounds__(256) void row_reduce_kernel(const Op op, int64_t rows, int C, int G, FastDiv fd_hw) {
  const int lane = threadIdx.x & 63;
  const int rpw = 64 / G;
  const int sub = lane / G, gl = lane - sub * G;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t base = wave * rpw; base < rows; base += nwaves * rpw) {
    const int64_t row = base + sub;
    const bool valid = row < rows;
    float acc = 0.f;
    if (valid) {
      const int n = (int)fd_div((uint32_t)row, fd_hw);
      const auto rc = op.row_ctx(row, n);
      for (int c = gl * V; c < C; c += G * V) acc += op.template term<V>(row, n, c, rc);
    }
    for (int off = G >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (valid && gl == 0) op.store(row, acc);
  }
}

template <class Op>
int launch_row_reduce(const Op& op, int64_t rows, int64_t HW, int C, bool vec, hipStream_t st, const char* name) {
  const int V = vec ? 4 : 1;
  int G = 1;
  while (G < C / V && G < 64) G <<= 1;
  const int rpw = 64 / G;
  int64_t blocks = sg_cdiv(sg_cdiv(rows, rpw), 4);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  if (vec)
    hipLaunchKernelGGL((row_reduce_kernel<Op, 4>), dim3((unsigned)blocks), dim3(256), 0, st, op, rows, C, G,
                       make_fastdiv((uint32_t)HW));
  else
    hipLaunchKernelGGL((row_reduce_kernel<Op, 1>), dim3((unsigned)blocks), dim3(256), 0, st, op, rows, C, G,
                       make_fastdiv((uint32_t)HW));
  SG_LAUNCH_CHECK(name);
  return 0;
}

// ---------------------------------------------------------------------------------------- broadcast mul
template <typename T>
struct BcastMulFwd {
  const T* __restrict__ x;
  const T* __restrict__ g;
  T* __restrict__ y;
  int C, mode, accumulate;
  template <int V>
  __device__ __forceinline__ void apply(int64_t r, int n, int c) const {
    float xv[V], o[V];
    ldv<V>(x + r * C + c, xv);
    if (accumulate) ldv<V>(y + r * C + c, o);
    if (mode == 0) {
      float gv[V];
      ldv<V>(g + (int64_t)n * C + c, gv);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = accumulate ? fmaf(xv[k], gv[k], o[k]) : xv[k] * gv[k];
    } else {
      const float gs = ld1<T>(g + r);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = accumulate ? fmaf(xv[k], gs, o[k]) : xv[k] * gs;
    }
    stv<V>(y + r * C + c, o);
  }
};

// mode 0 backward: dx (+)= dy * g[n,c]   (element-wise) ; dg[n,c] = sum_hw dy * x  (segmented reduce)
template <typename T>
struct BcastMulBwdDx0 {
  const T* __restrict__ g;
  const T* __restrict__ dy;
  T* __restrict__ dx;
  int C, accumulate;
  template <int V>
  __device__ __forceinline__ void apply(int64_t r, int n, int c) const {
    float gv[V], d[V], o[V];
    ldv<V>(g + (int64_t)n * C + c, gv);
    ldv<V>(dy + r * C + c, d);
    if (accumulate) ldv<V>(dx + r * C + c, o);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = accumulate ? fmaf(d[k], gv[k], o[k]) : d[k] * gv[k];
    stv<V>(dx + r * C + c, o);
  }
};

template <typename T>
struct ChanDotOp {  // out[n,c] = scale(n,c) * sum_hw dy*x ; kind 0: plain, 1: * sig'(logit[n,c])
  static constexpr int NOUT = 1;
  const T* __restrict__ x;
  const T* __restrict__ dy;
  const T* __restrict__ logit;  // [N,C] (kind 1)
  T* out;
  int64_t HW;
  int C, kind;
  template <int V>
  __device__ __forceinline__ void accum(int seg, int64_t r, int c, float (&acc)[1][V]) const {
    const int64_t row = (int64_t)seg * HW + r;
    float xv[V], d[V];
    ldv<V>(x + row * C + c, xv);
    ldv<V>(dy + row * C + c, d);
#pragma unroll
    for (int k = 0; k < V; ++k) acc[0][k] = fmaf(xv[k], d[k], acc[0][k]);
  }
  __device__ __forceinline__ void finalize(int seg, int c, const double (&s)[1]) const {
    float v = (float)s[0];
    if (kind == 1) {
      const float sg = sigmoidf_(ld1<T>(logit + (int64_t)seg * C + c));
      v *= sg * (1.f - sg);
    }
    st1<T>(out + (int64_t)seg * C + c, v);
  }
};

// mode 1 backward (spatial gate g[row]): dx (+)= dy*g[row]; dg[row] = sum_c dy*x
template <typename T>
struct BcastMulBwdRow1 {
  const T* __restrict__ x;
  const T* __restrict__ g;
  const T* __restrict__ dy;
  T* __restrict__ dx;
  T* __restrict__ dg;
  int C, accumulate;
  __device__ __forceinline__ float row_ctx(int64_t row, int) const { return ld1<T>(g + row); }
  template <int V>
  __device__ __forceinline__ float term(int64_t row, int, int c, float gs) const {
    float xv[V], d[V], o[V];
    ldv<V>(x + row * C + c, xv);
    ldv<V>(dy + row * C + c, d);
    if (accumulate) ldv<V>(dx + row * C + c, o);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      s = fmaf(xv[k], d[k], s);
      o[k] = accumulate ? fmaf(d[k], gs, o[k]) : d[k] * gs;
    }
    stv<V>(dx + row * C + c, o);
    return s;
  }
  __device__ __forceinline__ void store(int64_t row, float acc) const { st1<T>(dg + row, acc); }
};

// ------------------------------------------------------------------------------------------------- scSE
template <typename T>
struct ScseFwd {
  const T* __restrict__ x;
  const T* __restrict__ s;  // [N*HW] logits
  const T* __restrict__ cl;  // [N,C] logits
  T* __restrict__ y;
  int C;
  template <int V>
  __device__ __forceinline__ void apply(int64_t r, int n, int c) const {
    float xv[V], cv[V], o[V];
    ldv<V>(x + r * C + c, xv);
    ldv<V>(cl + (int64_t)n * C + c, cv);
    const float ss = sigmoidf_(ld1<T>(s + r));
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = xv[k] * (ss + sigmoidf_(cv[k]));
    stv<V>(y + r * C + c, o);
  }
};

template <typename T>
struct ScseBwdRow {  // dx = dy*(sig s + sig c) ; ds[row] = sig'(s) * sum_c dy*x
  const T* __restrict__ x;
  const T* __restrict__ s;
  const T* __restrict__ cl;
  const T* __restrict__ dy;
  T* __restrict__ dx;
  T* __restrict__ ds;
  int C;
  __device__ __forceinline__ float row_ctx(int64_t row, int) const { return sigmoidf_(ld1<T>(s + row)); }
  template <int V>
  __device__ __forceinline__ float term(int64_t row, int n, int c, float ss) const {
    float xv[V], d[V], cv[V], o[V];
    ldv<V>(x + row * C + c, xv);
    ldv<V>(dy + row * C + c, d);
    ldv<V>(cl + (int64_t)n * C + c, cv);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      acc = fmaf(xv[k], d[k], acc);
      o[k] = d[k] * (ss + sigmoidf_(cv[k]));
    }
    stv<V>(dx + row * C + c, o);
    return acc;
  }
  __device__ __forceinline__ void store(int64_t row, float acc) const {
    const float ss = sigmoidf_(ld1<T>(s + row));
    st1<T>(ds + row, acc * ss * (1.f - ss));
  }
};

// -------------------------------------------------------------------------------------------------- BAM
template <typename T>
struct BamFwd {  // y = x + x * sigmoid(mc[n,c] + ms[row])
  const T* __restrict__ x;
  const T* __restrict__ mc;
  const T* __restrict__ ms;
  T* __restrict__ y;
  int C;
  template <int V>
  __device__ __forceinline__ void apply(int64_t r, int n, int c) const {
    float xv[V], mv[V], o[V];
    ldv<V>(x + r * C + c, xv);
    ldv<V>(mc + (int64_t)n * C + c, mv);
    const float sp = ld1<T>(ms + r);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = xv[k] * (1.f + sigmoidf_(mv[k] + sp));
    stv<V>(y + r * C + c, o);
  }
};

template <typename T>
struct BamBwdRow {  // dx = dy*(1+g) ; dms[row] = sum_c dy*x*g*(1-g)
  const T* __restrict__ x;
  const T* __restrict__ mc;
  const T* __restrict__ ms;
  const T* __restrict__ dy;
  T* __restrict__ dx;
  T* __restrict__ dms;
  int C;
  __device__ __forceinline__ float row_ctx(int64_t row, int) const { return ld1<T>(ms + row); }
  template <int V>
  __device__ __forceinline__ float term(int64_t row, int n, int c, float sp) const {
    float xv[V], d[V], mv[V], o[V];
    ldv<V>(x + row * C + c, xv);
    ldv<V>(dy + row * C + c, d);
    ldv<V>(mc + (int64_t)n * C + c, mv);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float g = sigmoidf_(mv[k] + sp);
      acc = fmaf(xv[k] * d[k], g * (1.f - g), acc);
      o[k] = d[k] * (1.f + g);
    }
    stv<V>(dx + row * C + c, o);
    return acc;
  }
  __device__ __forceinline__ void store(int64_t row, float acc) const { st1<T>(dms + row, acc); }
};

template <typename T>
struct BamChanOp {  // dmc[n,c] = sum_hw dy*x*g*(1-g)
  static constexpr int NOUT = 1;
  const T* __restrict__ x;
  const T* __restrict__ dy;
  const T* __restrict__ mc;
  const T* __restrict__ ms;
  T* out;
  int64_t HW;
  int C;
  template <int V>
  __device__ __forceinline__ void accum(int seg, int64_t r, int c, float (&acc)[1][V]) const {
    const int64_t row = (int64_t)seg * HW + r;
    float xv[V], d[V], mv[V];
    ldv<V>(x + row * C + c, xv);
    ldv<V>(dy + row * C + c, d);
    ldv<V>(mc + (int64_t)seg * C + c, mv);
    const float sp = ld1<T>(ms + row);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float g = sigmoidf_(mv[k] + sp);
      acc[0][k] = fmaf(xv[k] * d[k], g * (1.f - g), acc[0][k]);
    }
  }
  __device__ __forceinline__ void finalize(int seg, int c, const double (&s)[1]) const {
    st1<T>(out + (int64_t)seg * C + c, (float)s[0]);
  }
};

inline bool vec_ok(int C, std::initializer_list<const void*> ptrs) {
  if (C % 4) return false;
  for (const void* p : ptrs)
    if (p && !sg_aligned16(p)) return false;
  return true;
}

#define SG_GATE_ARGS_CHECK(name)                                                                 \
  SG_CHECK_ARG(ctx && (dtype == SG_F32 || dtype == SG_BF16), name ": bad ctx/dtype");                                 \
  SG_CHECK_ARG(N > 0 && HW > 0 && C > 0, name ": non-positive dims");                            \
  SG_CHECK_ARG((int64_t)N * HW * C < (1ll << 31), name ": tensor exceeds 2^31 elements")

}  // namespace

extern "C" {

int sg_act_fwd(sg_ctx* ctx, void* stream, int dtype, int act, int64_t n, const void* x, void* y) {
  SG_CHECK_ARG(ctx && x && y && n >= 0, "sg_act_fwd: bad argument");
  SG_CHECK_ARG(act == SG_ACT_RELU || act == SG_ACT_SIGMOID, "sg_act_fwd: unknown activation %d", act);
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  SG_DTYPE_SWITCH(dtype, "sg_act_fwd", {
    if (n % 4 == 0 && sg_aligned16(x) && sg_aligned16(y))
      hipLaunchKernelGGL((act_fwd_kernel<4, T>), dim3(ew_blocks(n / 4)), dim3(256), 0, st, (const T*)x, (T*)y, n, act);
    else
      hipLaunchKernelGGL((act_fwd_kernel<1, T>), dim3(ew_blocks(n)), dim3(256), 0, st, (const T*)x, (T*)y, n, act);
  });
  SG_LAUNCH_CHECK("act_fwd_kernel");
  return 0;
}

int sg_act_bwd(sg_ctx* ctx, void* stream, int dtype, int act, int64_t n, const void* y, const void* dy, void* dx,
               int accumulate) {
  SG_CHECK_ARG(ctx && y && dy && dx && n >= 0, "sg_act_bwd: bad argument");
  SG_CHECK_ARG(act == SG_ACT_RELU || act == SG_ACT_SIGMOID, "sg_act_bwd: unknown activation %d", act);
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  SG_DTYPE_SWITCH(dtype, "sg_act_bwd", {
    if (sizeof(T) == 2 && n % 8 == 0 && sg_aligned16(y) && sg_aligned16(dy) && sg_aligned16(dx))
      hipLaunchKernelGGL((act_bwd_kernel<8, T>), dim3(ew_blocks(n / 8)), dim3(256), 0, st, (const T*)y, (const T*)dy, (T*)dx, n,
                         act, accumulate);
    else if (n % 4 == 0 && sg_aligned16(y) && sg_aligned16(dy) && sg_aligned16(dx))
      hipLaunchKernelGGL((act_bwd_kernel<4, T>), dim3(ew_blocks(n / 4)), dim3(256), 0, st, (const T*)y, (const T*)dy, (T*)dx, n,
                         act, accumulate);
    else
      hipLaunchKernelGGL((act_bwd_kernel<1, T>), dim3(ew_blocks(n)), dim3(256), 0, st, (const T*)y, (const T*)dy, (T*)dx, n, act,
                         accumulate);
  });
  SG_LAUNCH_CHECK("act_bwd_kernel");
  return 0;
}

int sg_add_n(sg_ctx* ctx, void* stream, int dtype, int k, const void* const* xs, int64_t n, void* y, int relu) {
  SG_CHECK_ARG(ctx && xs && y && n >= 0, "sg_add_n: bad argument");
  SG_CHECK_ARG(k >= 1 && k <= 8, "sg_add_n: k=%d outside [1,8]", k);
  if (n == 0) return 0;
  AddNArgs a;
  bool vec = (n % 4 == 0) && sg_aligned16(y);
  for (int i = 0; i < 8; ++i) {
    a.xs[i] = i < k ? xs[i] : nullptr;
    if (i < k) {
      SG_CHECK_ARG(xs[i] != nullptr, "sg_add_n: null operand %d", i);
      vec = vec && sg_aligned16(xs[i]);
    }
  }
  a.k = k;
  hipStream_t st = (hipStream_t)stream;
  SG_DTYPE_SWITCH(dtype, "sg_add_n", {
    if (vec && sizeof(T) == 2 && n % 8 == 0)
      hipLaunchKernelGGL((add_n_kernel<8, T>), dim3(ew_blocks(n / 8)), dim3(256), 0, st, a, (T*)y, n, relu);
    else if (vec)
      hipLaunchKernelGGL((add_n_kernel<4, T>), dim3(ew_blocks(n / 4)), dim3(256), 0, st, a, (T*)y, n, relu);
    else
      hipLaunchKernelGGL((add_n_kernel<1, T>), dim3(ew_blocks(n)), dim3(256), 0, st, a, (T*)y, n, relu);
  });
  SG_LAUNCH_CHECK("add_n_kernel");
  return 0;
}

int sg_copy_channels(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* src, int src_ld,
                     int src_off, void* dst, int dst_ld, int dst_off, int accumulate) {
  SG_CHECK_ARG(ctx && src && dst, "sg_copy_channels: bad argument");
  SG_CHECK_ARG(rows >= 0 && C > 0 && src_ld >= src_off + C && dst_ld >= dst_off + C && src_off >= 0 && dst_off >= 0,
               "sg_copy_channels: slice out of range (C=%d src %d+%d dst %d+%d)", C, src_off, src_ld, dst_off, dst_ld);
  SG_CHECK_ARG(rows * (int64_t)(src_ld > dst_ld ? src_ld : dst_ld) < (1ll << 31), "sg_copy_channels: tensor too large");
  if (rows == 0) return 0;
  const bool vec = (C % 4 == 0) && (src_ld % 4 == 0) && (dst_ld % 4 == 0) && (src_off % 4 == 0) && (dst_off % 4 == 0) &&
                   sg_aligned16(src) && sg_aligned16(dst);
  hipStream_t st = (hipStream_t)stream;
  SG_DTYPE_SWITCH(dtype, "sg_copy_channels", {
    if (vec && sizeof(T) == 2 && (C % 8 == 0) && (src_ld % 8 == 0) && (dst_ld % 8 == 0) && (src_off % 8 == 0) && (dst_off % 8 == 0))
      hipLaunchKernelGGL((copy_channels_kernel<8, T>), dim3(ew_blocks(rows * (C / 8))), dim3(256), 0, st, (const T*)src, src_ld,
                         src_off, (T*)dst, dst_ld, dst_off, rows, C, accumulate, make_fastdiv((uint32_t)(C / 8)));
    else if (vec)
      hipLaunchKernelGGL((copy_channels_kernel<4, T>), dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, st, (const T*)src, src_ld,
                         src_off, (T*)dst, dst_ld, dst_off, rows, C, accumulate, make_fastdiv((uint32_t)(C / 4)));
    else
      hipLaunchKernelGGL((copy_channels_kernel<1, T>), dim3(ew_blocks(rows * C)), dim3(256), 0, st, (const T*)src, src_ld,
                         src_off, (T*)dst, dst_ld, dst_off, rows, C, accumulate, make_fastdiv((uint32_t)C));
  });
  SG_LAUNCH_CHECK("copy_channels_kernel");
  return 0;
}

int sg_softmax2_fwd(sg_ctx* ctx, void* stream, int dtype, int64_t rows, const void* z, void* p) {
  SG_CHECK_ARG(ctx && dtype == SG_F32 && z && p && rows >= 0, "sg_softmax2_fwd: bad argument");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(softmax2_fwd_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, (const float*)z,
                     (float*)p, rows);
  SG_LAUNCH_CHECK("softmax2_fwd_kernel");
  return 0;
}

int sg_softmax2_bwd(sg_ctx* ctx, void* stream, int dtype, int64_t rows, const void* p, const void* dp, void* dz) {
  SG_CHECK_ARG(ctx && dtype == SG_F32 && p && dp && dz && rows >= 0, "sg_softmax2_bwd: bad argument");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(softmax2_bwd_kernel, dim3(ew_blocks(rows)), dim3(256), 0, (hipStream_t)stream, (const float*)p,
                     (const float*)dp, (float*)dz, rows);
  SG_LAUNCH_CHECK("softmax2_bwd_kernel");
  return 0;
}

int sg_softmax_branch_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int B, int C, const void* z, void* p) {
  SG_CHECK_ARG(ctx && z && p && N > 0 && B > 0 && C > 0, "sg_softmax_branch_fwd: bad argument");
  SG_DTYPE_SWITCH(dtype, "sg_softmax_branch_fwd", {
    hipLaunchKernelGGL(softmax_branch_fwd_kernel<T>, dim3((unsigned)sg_cdiv((int64_t)N * C, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)z, (T*)p, N, B, C);
  });
  SG_LAUNCH_CHECK("softmax_branch_fwd_kernel");
  return 0;
}

int sg_softmax_branch_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int B, int C, const void* p, const void* dp,
                          void* dz) {
  SG_CHECK_ARG(ctx && p && dp && dz && N > 0 && B > 0 && C > 0, "sg_softmax_branch_bwd: bad argument");
  SG_DTYPE_SWITCH(dtype, "sg_softmax_branch_bwd", {
    hipLaunchKernelGGL(softmax_branch_bwd_kernel<T>, dim3((unsigned)sg_cdiv((int64_t)N * C, 256)), dim3(256), 0,
                       (hipStream_t)stream, (const T*)p, (const T*)dp, (T*)dz, N, B, C);
  });
  SG_LAUNCH_CHECK("softmax_branch_bwd_kernel");
  return 0;
}

int sg_bcast_mul_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, int mode, const void* x,
                     const void* g, void* y, int accumulate) {
  SG_GATE_ARGS_CHECK("sg_bcast_mul_fwd");
  SG_CHECK_ARG(x && g && y && (mode == 0 || mode == 1), "sg_bcast_mul_fwd: bad argument");
  const bool vec = vec_ok(C, {x, y, mode == 0 ? g : nullptr});
  SG_DTYPE_SWITCH(dtype, "sg_bcast_mul_fwd", {
    BcastMulFwd<T> f;
    f.x = (const T*)x; f.g = (const T*)g; f.y = (T*)y; f.C = C; f.mode = mode; f.accumulate = accumulate;
    return launch_rowcol(f, (int64_t)N * HW, HW, C, vec, (hipStream_t)stream, "bcast_mul_fwd");
  });
  return 0;
}

size_t sg_bcast_mul_bwd_ws_bytes(const sg_ctx* ctx, int N, int64_t HW, int C, int mode) {
  if (!ctx || mode != 0) return 256;
  const SegPlan a = seg_plan<1>(ctx->num_cus, N, HW, C, true), b = seg_plan<1>(ctx->num_cus, N, HW, C, false);
  return (a.part_bytes > b.part_bytes ? a.part_bytes : b.part_bytes) + 256;
}

int sg_bcast_mul_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, int mode, const void* x,
                     const void* g, const void* dy, void* dx, void* dg, int accumulate_dx, void* ws, size_t ws_bytes) {
  SG_GATE_ARGS_CHECK("sg_bcast_mul_bwd");
  SG_CHECK_ARG(x && g && dy && dx && dg && (mode == 0 || mode == 1), "sg_bcast_mul_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int64_t rows = (int64_t)N * HW;
  SG_DTYPE_SWITCH(dtype, "sg_bcast_mul_bwd", {
    if (mode == 0) {
      const bool vec = vec_ok(C, {x, g, dy, dx});
      BcastMulBwdDx0<T> f;
      f.g = (const T*)g; f.dy = (const T*)dy; f.dx = (T*)dx; f.C = C; f.accumulate = accumulate_dx;
      int rc = launch_rowcol(f, rows, HW, C, vec, st, "bcast_mul_bwd_dx");
      if (rc) return rc;
      const SegPlan pl = seg_plan<1>(ctx->num_cus, N, HW, C, vec);
      if (!ws || ws_bytes < pl.part_bytes) {
        sg_set_error("sg_bcast_mul_bwd: workspace %zu < %zu", ws_bytes, pl.part_bytes);
        return SG_EWORKSPACE;
      }
      ChanDotOp<T> op;
      op.x = (const T*)x; op.dy = (const T*)dy; op.logit = nullptr; op.out = (T*)dg; op.HW = HW; op.C = C;
      op.kind = 0;
      return seg_reduce_launch(op, pl, N, HW, C, (float*)ws, st, "bcast_mul_bwd_dg");
    }
    const bool vec = vec_ok(C, {x, dy, dx});
    BcastMulBwdRow1<T> op;
    op.x = (const T*)x; op.g = (const T*)g; op.dy = (const T*)dy; op.dx = (T*)dx; op.dg = (T*)dg;
    op.C = C; op.accumulate = accumulate_dx;
    return launch_row_reduce(op, rows, HW, C, vec, st, "bcast_mul_bwd_row");
  });
  return 0;
}

int sg_scse_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, const void* x, const void* s,
                const void* c, void* y) {
  SG_GATE_ARGS_CHECK("sg_scse_fwd");
  SG_CHECK_ARG(x && s && c && y, "sg_scse_fwd: null tensor");
  SG_DTYPE_SWITCH(dtype, "sg_scse_fwd", {
    ScseFwd<T> f;
    f.x = (const T*)x; f.s = (const T*)s; f.cl = (const T*)c; f.y = (T*)y; f.C = C;
    return launch_rowcol(f, (int64_t)N * HW, HW, C, vec_ok(C, {x, c, y}), (hipStream_t)stream, "scse_fwd");
  });
  return 0;
}

size_t sg_scse_bwd_ws_bytes(const sg_ctx* ctx, int N, int64_t HW, int C) {
  return sg_bcast_mul_bwd_ws_bytes(ctx, N, HW, C, 0);
}

int sg_scse_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, const void* x, const void* s,
                const void* c, const void* dy, void* dx, void* ds, void* dc, void* ws, size_t ws_bytes) {
  SG_GATE_ARGS_CHECK("sg_scse_bwd");
  SG_CHECK_ARG(x && s && c && dy && dx && ds && dc, "sg_scse_bwd: null tensor");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = vec_ok(C, {x, c, dy, dx});
  SG_DTYPE_SWITCH(dtype, "sg_scse_bwd", {
    ScseBwdRow<T> r;
    r.x = (const T*)x; r.s = (const T*)s; r.cl = (const T*)c; r.dy = (const T*)dy; r.dx = (T*)dx;
    r.ds = (T*)ds; r.C = C;
    int rc = launch_row_reduce(r, (int64_t)N * HW, HW, C, vec, st, "scse_bwd_row");
    if (rc) return rc;
    const SegPlan pl = seg_plan<1>(ctx->num_cus, N, HW, C, vec);
    if (!ws || ws_bytes < pl.part_bytes) {
      sg_set_error("sg_scse_bwd: workspace %zu < %zu", ws_bytes, pl.part_bytes);
      return SG_EWORKSPACE;
    }
    ChanDotOp<T> op;
    op.x = (const T*)x; op.dy = (const T*)dy; op.logit = (const T*)c; op.out = (T*)dc; op.HW = HW; op.C = C;
    op.kind = 1;
    return seg_reduce_launch(op, pl, N, HW, C, (float*)ws, st, "scse_bwd_dc");
  });
  return 0;
}

int sg_bam_fwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, const void* x, const void* mc,
               const void* ms, void* y) {
  SG_GATE_ARGS_CHECK("sg_bam_fwd");
  SG_CHECK_ARG(x && mc && ms && y, "sg_bam_fwd: null tensor");
  SG_DTYPE_SWITCH(dtype, "sg_bam_fwd", {
    BamFwd<T> f;
    f.x = (const T*)x; f.mc = (const T*)mc; f.ms = (const T*)ms; f.y = (T*)y; f.C = C;
    return launch_rowcol(f, (int64_t)N * HW, HW, C, vec_ok(C, {x, mc, y}), (hipStream_t)stream, "bam_fwd");
  });
  return 0;
}

size_t sg_bam_bwd_ws_bytes(const sg_ctx* ctx, int N, int64_t HW, int C) {
  return sg_bcast_mul_bwd_ws_bytes(ctx, N, HW, C, 0);
}

int sg_bam_bwd(sg_ctx* ctx, void* stream, int dtype, int N, int64_t HW, int C, const void* x, const void* mc,
               const void* ms, const void* dy, void* dx, void* dmc, void* dms, void* ws, size_t ws_bytes) {
  SG_GATE_ARGS_CHECK("sg_bam_bwd");
  SG_CHECK_ARG(x && mc && ms && dy && dx && dmc && dms, "sg_bam_bwd: null tensor");
  hipStream_t st = (hipStream_t)stream;
  const bool vec = vec_ok(C, {x, mc, dy, dx});
  SG_DTYPE_SWITCH(dtype, "sg_bam_bwd", {
    BamBwdRow<T> r;
    r.x = (const T*)x; r.mc = (const T*)mc; r.ms = (const T*)ms; r.dy = (const T*)dy; r.dx = (T*)dx;
    r.dms = (T*)dms; r.C = C;
    int rc = launch_row_reduce(r, (int64_t)N * HW, HW, C, vec, st, "bam_bwd_row");
    if (rc) return rc;
    const SegPlan pl = seg_plan<1>(ctx->num_cus, N, HW, C, vec);
    if (!ws || ws_bytes < pl.part_bytes) {
      sg_set_error("sg_bam_bwd: workspace %zu < %zu", ws_bytes, pl.part_bytes);
      return SG_EWORKSPACE;
    }
    BamChanOp<T> op;
    op.x = (const T*)x; op.dy = (const T*)dy; op.mc = (const T*)mc; op.ms = (const T*)ms;
    op.out = (T*)dmc; op.HW = HW; op.C = C;
    return seg_reduce_launch(op, pl, N, HW, C, (float*)ws, st, "bam_bwd_dmc");
  });
  return 0;
}

}  // extern "C"
