// The RGB stems: Conv2D(Cout, 3) on a three-channel image (stride 1: the U-Nets' first convolution, predict_model/res34.py:50-52,
// scse.py; stride 2: DeepLabv3+ / HRNet, v3plus.py:173).  Included by conv_igemm.hip.  Round 4 (VERDICT r2 / r3 "Cin = 3 stem").
//
// K = 27: as an implicit GEMM on the matrix pipe every tap is padded to 32 reduction channels (ten times the useful work,
// tried in round 2), and the any-shape scalar kernel these layers ran on took 3.2 ms forward + 3.2 ms filter gradient at
// 512 x 512 x 16 -> 64 channels (Res34-UNet) for 1.1 GB of output: 0.33 TB/s.  These two kernels are what the shape is - a
// bandwidth-bound stencil with 27 multiplies per output:
//   stem3_fwd_kernel    a thread owns 4 output pixels of a row x 8 output channels; the 3 x (3 + 3 stride) x 3 input window
//                       comes row by row through registers, the 27 x Cout kernel sits in LDS; a pixel's Cout outputs leave
//                       as contiguous 32-byte pieces of 8 neighbouring lanes.  The 27 products of an output are added with
//                       fmaf in the order (kh, kw, ci) onto 0, the bias last - the very chain the fp32 MFMA kernel it replaces
//                       evaluates (v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain): fp32 results are BIT-IDENTICAL to it.
//   stem3_wgrad_kernel  dw[27][Cout] = sum over pixels x[window] * dy: a workgroup walks a range of output pixels, thread (co,
//                       group of 7 taps) reads dy[p][co] once and multiplies it with its 7 window values (staged per row
//                       segment in LDS); per-workgroup partial slabs are added by stem3_reduce_kernel in fixed order.
// There is no input gradient (the image has none).
#pragma once

template <typename T>
struct Stem3Params {
  const T* __restrict__ x;      // [N][H][W][3] dense
  const float* __restrict__ w;  // [3][3][3][Cout]
  const float* __restrict__ bias;
  T* __restrict__ y;            // [N][OH][OW][y_ld]
  int N, H, W, OH, OW, Cout, y_ld, stride, pad_t, pad_l, relu;
  FastDiv fd_q, fd_oh;          // quads per output row, output rows per image
};

template <typename T, int STRIDE>
__global__ __launch_bounds__(256) void stem3_fwd_kernel(const Stem3Params<T> p) {
  extern __shared__ float sw[];   // [27][Cout]
  for (int i = threadIdx.x; i < 27 * p.Cout; i += 256) sw[i] = p.w[i];
  __syncthreads();
  const int oct = p.Cout / 8;                      // channel octets per pixel: 4 (Cout 32) or 8 (Cout 64) ... a power of two
  const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
  const uint32_t co8 = gid % (uint32_t)oct, quad = gid / (uint32_t)oct;
  uint32_t rowi, q, n, oh;
  fd_divmod(quad, p.fd_q, rowi, q);
  fd_divmod(rowi, p.fd_oh, n, oh);
  if ((int)n >= p.N) return;
  const int ow0 = (int)q * 4, c0 = (int)co8 * 8;
  constexpr int NC = 3 + 3 * STRIDE;               // input columns of the window of 4 outputs
  float acc[4][8];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
  const int iw0 = ow0 * STRIDE - p.pad_l;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = (int)oh * STRIDE - p.pad_t + kh;
    const bool rok = (unsigned)ih < (unsigned)p.H;
    const T* rowp = p.x + ((int64_t)((int)n * p.H + (rok ? ih : 0)) * p.W) * 3;
    float xv[NC][3];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int iw = iw0 + c;
      const bool ok = rok && (unsigned)iw < (unsigned)p.W;
      const T* px = rowp + (int64_t)(ok ? iw : 0) * 3;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float v = ld1<T>(px + ci);
        xv[c][ci] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float* wr = sw + ((kh * 3 + kw) * 3 + ci) * p.Cout + c0;
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xs = xv[k * STRIDE + kw][ci];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[k][e] = fmaf(xs, w0[e], acc[k][e]);
            acc[k][4 + e] = fmaf(xs, w1[e], acc[k][4 + e]);
          }
        }
      }
  }
  f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    b0 = *reinterpret_cast<const f32x4*>(p.bias + c0);
    b1 = *reinterpret_cast<const f32x4*>(p.bias + c0 + 4);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (ow0 + k >= p.OW) break;
    f32x4 o0, o1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o0[e] = acc[k][e] + b0[e];
      o1[e] = acc[k][4 + e] + b1[e];
      if (p.relu) { o0[e] = fmaxf(o0[e], 0.f); o1[e] = fmaxf(o1[e], 0.f); }
    }
    T* dst = p.y + ((int64_t)((int)n * p.OH + (int)oh) * p.OW + ow0 + k) * p.y_ld + c0;
    st4<T>(dst, o0);
    st4<T>(dst + 4, o1);
  }
}

template <typename T>
struct Stem3WParams {
  const T* __restrict__ x;      // [N][H][W][3]
  const T* __restrict__ dy;     // [N][OH][OW][y_ld]
  float* __restrict__ part;     // [blocks][27][Cout]
  int N, H, W, OH, OW, Cout, y_ld, stride, pad_t, pad_l;
  int rows_per_block;           // output rows (of N * OH) per workgroup
};

// One workgroup = rows_per_block output rows; thread t -> output channel co = t % Cout (Cout = 64: 4 tap groups of 7 taps;
// Cout = 32: 8 tap groups of 4 taps, the last ones ragged).  Per output row the three input rows (W x 3 values, zero padded
// by one pixel on either side) are staged in LDS; per output pixel a thread loads dy[p][co] and multiplies it with its taps'
// window values (LDS broadcast reads).
template <typename T, int COUT>
__global__ __launch_bounds__(256) void stem3_wgrad_kernel(const Stem3WParams<T> p) {
  constexpr int NG = 256 / COUT;                    // tap groups
  constexpr int TPG = (27 + NG - 1) / NG;           // taps per group: 7 (Cout 64) / 4 (Cout 32)
  extern __shared__ float srow[];                   // [3][(W + 2) * 3]
  const int t = threadIdx.x, co = t % COUT, g = t / COUT;
  const int rowlen = (p.W + 2) * 3;
  float acc[TPG];
#pragma unroll
  for (int i = 0; i < TPG; ++i) acc[i] = 0.f;
  // tap i of this group: k = g * TPG + i -> (kh, kw, ci); LDS offset of its value for output column 0
  int toff[TPG];
  bool tok[TPG];
#pragma unroll
  for (int i = 0; i < TPG; ++i) {
    const int k = g * TPG + i;
    tok[i] = k < 27;
    const int kk = tok[i] ? k : 0;
    const int kh = kk / 9, kw = (kk / 3) % 3, ci = kk % 3;
    toff[i] = kh * rowlen + (kw - p.pad_l + 1) * 3 + ci;   // input column ow * stride + kw - pad_l, shifted by the left pad pixel
  }
  const int64_t row_begin = (int64_t)blockIdx.x * p.rows_per_block;
  int64_t row_end = row_begin + p.rows_per_block;
  const int64_t rows_total = (int64_t)p.N * p.OH;
  if (row_end > rows_total) row_end = rows_total;
  for (int64_t r = row_begin; r < row_end; ++r) {
    const int n = (int)(r / p.OH), oh = (int)(r - (int64_t)n * p.OH);
    __syncthreads();   // the previous row's readers are done
    for (int i = t; i < 3 * rowlen; i += 256) {
      const int kh = i / rowlen, rem = i - kh * rowlen;
      const int col = rem / 3 - 1, ci = rem - (rem / 3) * 3;
      const int ih = oh * p.stride - p.pad_t + kh;
      float v = 0.f;
      if ((unsigned)ih < (unsigned)p.H && (unsigned)col < (unsigned)p.W)
        v = ld1<T>(p.x + ((int64_t)(n * p.H + ih) * p.W + col) * 3 + ci);
      srow[i] = v;
    }
    __syncthreads();
    const T* dyr = p.dy + ((int64_t)r * p.OW) * p.y_ld + co;
    for (int ow = 0; ow < p.OW; ow += 4) {
      float gv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) gv[u] = (ow + u < p.OW) ? ld1<T>(dyr + (int64_t)(ow + u) * p.y_ld) : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int base = (ow + u) * p.stride * 3;
#pragma unroll
        for (int i = 0; i < TPG; ++i) {
          // (a column past the staged row only occurs for ow + u >= OW, where gv is 0: the index is clamped)
          int idx = toff[i] + base;
          idx = idx < 3 * rowlen ? idx : 0;
          acc[i] = fmaf(srow[idx], gv[u], acc[i]);
        }
      }
    }
  }
  float* out = p.part + (int64_t)blockIdx.x * 27 * COUT;
#pragma unroll
  for (int i = 0; i < TPG; ++i)
    if (tok[i]) out[(g * TPG + i) * COUT + co] = acc[i];
}

// 3x3 on three dense input channels, no dilation, stride 1 or 2, 32 or 64 output channels, W small enough for the staged rows
inline bool stem3_ok(const sg_conv_desc* d) {
  static const int on = getenv("SG_STEM3") ? atoi(getenv("SG_STEM3")) : 1;
  if (!on) return false;
  const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
  return d->Cin == 3 && xl == 3 && d->KH == 3 && d->KW == 3 && d->dilation == 1 && (d->stride == 1 || d->stride == 2) &&
         (d->Cout == 32 || d->Cout == 64) && yl % 4 == 0 && d->W <= 4096 && d->pad_t <= 1 && d->pad_l <= 1;
}

inline int stem3_wgrad_blocks(int num_cus, const sg_conv_desc* d, int& rows_per_block) {
  const int64_t rows = (int64_t)d->N * d->Ho;
  int64_t blocks = (int64_t)num_cus * 4;   // partial slabs the second stage adds per output (stem3_reduce_kernel)
  if (blocks > rows) blocks = rows;
  rows_per_block = (int)sg_cdiv(rows, blocks);
  return (int)sg_cdiv(rows, rows_per_block);
}
inline size_t stem3_wgrad_ws_bytes(int num_cus, const sg_conv_desc* d) {
  int rpb;
  return (size_t)stem3_wgrad_blocks(num_cus, d, rpb) * 27 * d->Cout * sizeof(float);
}

// dw[i] = sum over the S partial slabs, 16 lanes per output (lane l adds slabs l, l + 16, ... in order, the 16 lane sums are
// added in lane order): deterministic
__global__ __launch_bounds__(256) void stem3_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int S) {
  __shared__ float red[16][17];
  const int t = threadIdx.x, o = t & 15, l = t >> 4;
  const int i = blockIdx.x * 16 + o;
  float s = 0.f;
  if (i < n) {
#pragma unroll 8
    for (int z = l; z < S; z += 16) s += part[(int64_t)z * n + i];
  }
  red[l][o] = s;
  __syncthreads();
  if (l == 0 && i < n) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += red[q][o];
    out[i] = tot;
  }
}

template <typename T>
int launch_stem3_fwd(const sg_conv_desc* d, const void* x, const void* w, const void* bias, void* y, int flags, hipStream_t st) {
  Stem3Params<T> p;
  p.x = (const T*)x; p.w = (const float*)w; p.bias = (flags & SG_EPI_BIAS) ? (const float*)bias : nullptr; p.y = (T*)y;
  p.N = d->N; p.H = d->H; p.W = d->W; p.OH = d->Ho; p.OW = d->Wo; p.Cout = d->Cout; p.y_ld = d->y_ld ? d->y_ld : d->Cout;
  p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.relu = (flags & SG_EPI_RELU) ? 1 : 0;
  const int quads = (int)sg_cdiv(d->Wo, 4);
  p.fd_q = make_fastdiv((uint32_t)quads);
  p.fd_oh = make_fastdiv((uint32_t)d->Ho);
  const int64_t threads = (int64_t)d->N * d->Ho * quads * (d->Cout / 8);
  if (threads > 0x7fffffffll * 128) {
    sg_set_error("stem3: too many outputs");
    return SG_EINVAL;
  }
  const size_t lds = (size_t)27 * d->Cout * sizeof(float);
  const dim3 grid((unsigned)sg_cdiv(threads, 256));
  if (d->stride == 1) hipLaunchKernelGGL((stem3_fwd_kernel<T, 1>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((stem3_fwd_kernel<T, 2>), grid, dim3(256), lds, st, p);
  SG_LAUNCH_CHECK("stem3_fwd_kernel");
  return 0;
}

template <typename T>
int launch_stem3_wgrad(int num_cus, const sg_conv_desc* d, const void* x, const void* dy, float* part, int& blocks_out, hipStream_t st) {
  Stem3WParams<T> p;
  p.x = (const T*)x; p.dy = (const T*)dy; p.part = part;
  p.N = d->N; p.H = d->H; p.W = d->W; p.OH = d->Ho; p.OW = d->Wo; p.Cout = d->Cout; p.y_ld = d->y_ld ? d->y_ld : d->Cout;
  p.stride = d->stride; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  const int blocks = stem3_wgrad_blocks(num_cus, d, p.rows_per_block);
  blocks_out = blocks;
  const size_t lds = (size_t)3 * (d->W + 2) * 3 * sizeof(float);
  if (d->Cout == 64) hipLaunchKernelGGL((stem3_wgrad_kernel<T, 64>), dim3((unsigned)blocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((stem3_wgrad_kernel<T, 32>), dim3((unsigned)blocks), dim3(256), lds, st, p);
  SG_LAUNCH_CHECK("stem3_wgrad_kernel");
  return 0;
}

inline int launch_stem3_reduce(const float* part, float* dw, int n, int S, hipStream_t st) {
  hipLaunchKernelGGL(stem3_reduce_kernel, dim3((unsigned)sg_cdiv(n, 16)), dim3(256), 0, st, part, dw, n, S);
  SG_LAUNCH_CHECK("stem3_reduce_kernel");
  return 0;
}
