// Implicit-GEMM NHWC convolution on the gfx950 matrix cores (fp32-in / fp32-acc MFMA 32x32x2).
//
//   forward  y[M = N*Ho*Wo][Cout]  = A[M][K = KH*KW*Cin] * W[K][Cout]      A gathered from x (im2col on the fly)
//   dgrad    dx[M = N*H*W][Cin]    = A'[M][K' = KH*KW*Cout] * Wt[K'][Cin]  A' gathered from dy, Wt = per-tap
//                                                                         transpose of W (built in workspace)
//   wgrad    dw[K][Cout]           = sum_p A[p][K]^T * dy[p][Cout]         reduction over p = N*Ho*Wo, split
//                                                                         over workgroups, fixed-order reduce
//
// One gather formula serves forward and dgrad (and therefore Conv2DTranspose forward):
//     ih = (oh * a_mul + kh * k_mul + off_h);  valid iff ih % div == 0 and 0 <= ih/div < H
//   forward: a_mul = stride, k_mul = dilation,  off = -pad_before, div = 1
//   dgrad  : a_mul = 1,      k_mul = -dilation, off = +pad_before, div = stride (1 or 2)
//
// Tiling (wave64; BM = 128 rows, BK = 32):
//   128x128 and 128x64 tiles with 8 waves (2 waves per SIMD from ONE workgroup, so one wave's gather /
//   LDS traffic overlaps its partner's MFMAs even at one workgroup per CU) or 4 waves; 128x32 with 4 waves.
//   A slab [128][32] is staged k-contiguous in LDS with a 36-float row stride (ds_read_b128 conflict-free:
//   16-B slot = 9*row mod 16), the B slab [32][BN] n-contiguous (ds_read_b32, lanes = consecutive dwords).
//   The k index inside a group of 8 is split across the two lane halves (lanes 0-31: k..k+3, lanes 32-63:
//   k+4..k+7) so one ds_read_b128 feeds four MFMAs; A and B use the same split, so the GEMM is unchanged.
//   Global -> register -> LDS staging with the loads of slab s+PF issued before the MFMAs of slab s (PF = 1 or 2
//   register sets; fp32 MFMA makes a slab only ~1.7 us of work, so two slabs in flight cover HBM/MALL latency
//   under load), LDS double-buffered, ONE barrier per slab.  All gathers are branch-free (out-of-range taps
//   read element 0 and are zeroed by a select) so a slab is a single basic block for the scheduler.
//   Block ids are remapped so each XCD owns a contiguous run of M-tiles (shared halo / weights hit one L2).
//   Tile width and the wgrad split are chosen per launch to fill whole "waves" of 2 workgroups per CU.
#include "sg_reduce.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDA = BK + 4;  // floats; 144-byte rows

template <int V>
using IC = std::integral_constant<int, V>;

// A BatchNormalization (+ReLU) applied to the convolution's INPUT while it is loaded (round 5: sg_conv2d_fwd_stats_bn /
// sg_conv2d_wgrad_bn; the patch kernels and the thin 1x1 kernels): y = [relu](fmaf((x - mean) * invstd, gamma, beta)), bn_apply's own
// expression, on every pixel inside the image (the zero padding stays zero).  mean == nullptr: none.
struct BnIn {
  const float* mean;
  const float* invstd;   // infer: the moving variance, invstd = rsqrtf(var + eps)
  const float* gamma;
  const float* beta;
  int relu, infer;
  float eps;
};
__device__ __forceinline__ f32x4 bn_in_inv(const BnIn& b, int c) {
  f32x4 is = *reinterpret_cast<const f32x4*>(b.invstd + c);
  if (b.infer) {
#pragma unroll
    for (int k = 0; k < 4; ++k) is[k] = rsqrtf(is[k] + b.eps);
  }
  return is;
}
__device__ __forceinline__ float bn_in_one(float x, float m, float is, float g, float bt, int relu) {
  const float t = fmaf((x - m) * is, g, bt);
  return relu ? fmaxf(t, 0.f) : t;
}

// pw_wide_kernel<3, float, BN, true> (conv_pw.h): the BatchNormalization backward apply evaluated in the dgrad's A path
struct BnBwdIn {
  const float* x;        // the BatchNormalization's raw input (= the pointwise convolution's forward output), same layout as dy
  const float* mean;
  const float* invstd;
  const float* gamma;
  const float* beta;     // null: no fused ReLU
  const float* dgamma;   // finished column sums
  const float* dbeta;
  float* dz;             // out: the applied gradient (for the filter gradient)
  int relu;
  float inv_n;
};

struct IgemmParams {
  const float* __restrict__ x;
  const float* __restrict__ w;
  const float* __restrict__ bias;
  float* __restrict__ y;
  int H, W, C, x_ld;
  int OH, OW;
  int Nout, y_ld;
  int a_mul, k_mul, off_h, off_w, div;
  int K, M;
  int flags;
  int stagger;  // waves in the upper half of the workgroup issue their gathers AFTER their MFMAs
  int ablate;   // timing-only diagnostics (results wrong): 1 = no global loads in the loop, 2 = no LDS store/barrier
  uint32_t x_bytes, w_bytes;  // extents for the buffer descriptors of the UT path (both < 2^31)
  int skip_taps;              // drop taps that are pure padding for the whole tile (dilated convs)
  int group_m;                // > 1: tiles are walked in groups of group_m row tiles, row tile fastest (L2 reuse of B)
  int cb;                     // > 0: K order = (channel block of cb slabs) x tap x slab, so the 9 taps of a
                              //      channel block re-read x from L2 instead of the fabric; 0: tap-major
  FastDiv fd_ohow, fd_ow, fd_c, fd_kw;
  // x6 path (conv_x6.h): the weights as three bf16 planes [3][Npad][Kpad]; w_bytes then is their extent
  const unsigned short* __restrict__ wq;
  int Kpad, Npad;
  int kd;  // k-block depth of the plane layout [npl][Kpad / kd][Npad][kd] (= the kernel's slab depth); 0: rows [npl][Npad][Kpad]
  float* __restrict__ stats;  // SG_EPI_BN_STATS: [tiles_m][2][Nout] per-tile (sum, centred sum of squares), else null
  // Stride-2 dgrad (and Conv2DTranspose forward) with rows in PARITY-CLASS order (conv_x6_kernel / conv_b16_kernel only):
  // GEMM row m = ((cls * N + n) * OH/2 + i) * OW/2 + j is output pixel (n, 2i + (cls >> 1), 2j + (cls & 1)).  Of the KH x KW
  // taps only those of matching parity reach a pixel of class cls; in raster order a 128-row tile mixes the column
  // parities, so every tap had a valid row and the tile multiplied all of them (3/4 of the products masked to zero).
  // Class-pure tiles let the existing padding-tap elimination drop the taps of the other parities: a 3x3 kernel runs 1 / 2 /
  // 2 / 4 taps instead of 9 four times, a 1x1 kernel one tap for a quarter of the tiles and none for the rest.
  int perm2;
  FastDiv fd_mc, fd_hcwc, fd_wc;
  // dgrad: a gradient already collected for the same tensor (y's layout, may be y itself), added in the epilogue
  // (sg_conv2d_dgrad_acc; conv_x6_kernel / conv_b16_kernel only)
  const float* res;
  // the A operand already split into three bf16 planes [3][pixels][C] (sg_split_planes; sg_conv2d_fwd_stats_ap / _dgrad_ap): the
  // planes-in kernel (conv_x6w.h) reads them instead of splitting x itself; ignored by every other kernel; null: none
  const unsigned short* a_planes;
  BnIn bn;   // conv_x6p_kernel only: the BatchNormalization applied in the patch loader (mean == nullptr: none)
  BnBwdIn bnb;   // pw_wide_kernel<.., BNB> only (x == nullptr: none)
  // host side only: bytes of workspace that start at the weight planes (the planes, then scratch of a split-K launch:
  // conv_b16w.h); SIZE_MAX = prepared planes, whose arena slot sg_conv2d_planes_job sized for both
  size_t ws_room;
};

__device__ __forceinline__ int spt_of(const IgemmParams& p) { return p.C / BK; }  // slabs per tap (UT)

// GEMM row -> output pixel coordinates (image, row, column); see IgemmParams::perm2
__device__ __forceinline__ void row_to_pixel(const IgemmParams& p, uint32_t m, uint32_t& n, uint32_t& oh, uint32_t& ow) {
  if (p.perm2) {
    uint32_t cls, r, rem, i, j;
    fd_divmod(m, p.fd_mc, cls, r);
    fd_divmod(r, p.fd_hcwc, n, rem);
    fd_divmod(rem, p.fd_wc, i, j);
    oh = 2 * i + (cls >> 1);
    ow = 2 * j + (cls & 1);
  } else {
    uint32_t rem;
    fd_divmod(m, p.fd_ohow, n, rem);
    fd_divmod(rem, p.fd_ow, oh, ow);
  }
}
// element offset of GEMM row `row` in y
__device__ __forceinline__ int64_t row_to_yoff(const IgemmParams& p, int row) {
  if (!p.perm2) return (int64_t)row * p.y_ld;
  uint32_t n, oh, ow;
  row_to_pixel(p, (uint32_t)row, n, oh, ow);
  return ((int64_t)(n * p.OH + oh) * p.OW + ow) * p.y_ld;
}

// UT ("uniform tap"): Cin % 32 == 0 or a 1x1 kernel, so every 32-deep slab lies inside ONE filter tap.  The
// per-row source offsets and bounds flags then change only when the slab stream crosses a tap boundary (every
// Cin/32 slabs) and a slab's gather costs a handful of adds instead of ~170 VALU/SALU instructions of
// div/mod, bounds and 64-bit address arithmetic ahead of the first MFMA.
template <int BN, int WGM, int WGN, int PF, bool VEC, bool UT, typename TA = float, typename TY = TA>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN / 2) void igemm_conv_kernel(const IgemmParams p) {
  // TA = storage type of the activations x and y (bf16: widened to fp32 on load, rounded on store; the arithmetic is
  // the fp32 MFMA either way).  The buffer-addressed UT path exists for fp32 storage only: bf16 tensors that qualify
  // for it run on the bf16 matrix-pipe kernels instead (conv_x6.h), this kernel is their any-shape fallback.
  static_assert(!UT || std::is_same<TA, float>::value, "the UT gather is fp32-only");
  const TA* __restrict__ px = reinterpret_cast<const TA*>(p.x);
  TY* __restrict__ py = reinterpret_cast<TY*>(p.y);  // TY != TA: the fp32 softmax head of a bf16 model and its backward
  constexpr int NT = 64 * WGM * WGN;        // 4 or 8 waves; two workgroups per CU => 2 or 4 waves per SIMD
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDB = BN;
  constexpr int NA = (BM * BK / 4) / NT;    // float4 A chunks per thread (rows r0 + RS*j)
  constexpr int RS = NT / 8;
  constexpr int NB = (BK * BN / 4) / NT;    // float4 B chunks per thread
  static_assert(NA >= 1 && NB >= 1 && TM >= 1 && TN >= 1, "tile too small for the wave layout");
  static_assert(PF == 1 || PF == 2, "prefetch depth");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);  // [2][BM*LDA]
  float* Bs = As + 2 * BM * LDA;                // [2][BK*LDB]

  const int t = threadIdx.x;
  const uint32_t ntn = (p.Nout + BN - 1) / BN;
  // Tile order.  xcd_remap hands every XCD one contiguous range of the ordered tile ids; inside it either the
  // column tile runs fastest (the few column tiles of a row tile run together and share its A rows), or - wide
  // outputs, ntn >= 4 - tiles are walked in groups of group_m row tiles with the ROW tile fastest, so that the
  // ~64 tiles an XCD has in flight form a block that shares both operands through its L2 (ASPP dgrad: the
  // 18 MB transposed kernel is fetched once per XCD instead of once per four row tiles).
  const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  uint32_t tile_m, tile_n;
  if (p.group_m > 1) {
    const uint32_t ntm = gridDim.x / ntn, gm = (uint32_t)p.group_m;
    const uint32_t per = gm * ntn, g = bid / per, r = bid - g * per;
    const uint32_t left = ntm - g * gm, gsz = left < gm ? left : gm;
    tile_n = r / gsz;
    tile_m = g * gm + (r - tile_n * gsz);
  } else {
    tile_m = bid / ntn;
    tile_n = bid - tile_m * ntn;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread A rows: NA rows (r0 + RS j), one 4-wide k chunk (kc) ------------------------------
  const int kc = t & 7, r0 = t >> 3;
  // (oh, ow) of a row, scaled and offset, are packed as two signed 16-bit halves of one register (maps are at
  // most a few thousand pixels wide): the kernel runs at the 128-VGPR cap of 4 waves per SIMD and every live
  // register counts (a spill inside the MFMA loop costs a vmcnt(0) that drains the prefetch).
  // row_lin = n*H*W + oh'*W + ow' (the source pixel of tap (0,0) when div == 1) is needed only when the tap
  // changes: it lives in LDS (one private slot per thread and row), not in a register - and not in scratch,
  // whose reload would wait on vmcnt(0) and drain the prefetched slabs.
  int row_hw[NA];
  int* row_lin_lds = reinterpret_cast<int*>(Bs + 2 * BK * LDB) + 64;  // [NA][NT], after tapinfo[64]
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int m = m0 + r0 + RS * j;
    if (m < p.M) {
      uint32_t n, rem, oh, ow;
      fd_divmod((uint32_t)m, p.fd_ohow, n, rem);
      fd_divmod(rem, p.fd_ow, oh, ow);
      const int ohs = (int)oh * p.a_mul + p.off_h, ows = (int)ow * p.a_mul + p.off_w;
      row_lin_lds[j * NT + t] = (int)n * p.H * p.W + ohs * p.W + ows;
      row_hw[j] = (ohs << 16) | (ows & 0xffff);
    } else {
      row_lin_lds[j * NT + t] = 0;
      row_hw[j] = (int)0x80008000u;  // (-32768, -32768): out of bounds for every tap
    }
  }

  f32x4 ra[PF][NA];
  f32x4 rb[PF][NB];

  auto gather_elem_addr = [&](int j, int dh, int dw, bool kvalid, int64_t& off) -> bool {
    const int ohs = row_hw[j] >> 16, ows = (int)(short)(row_hw[j] & 0xffff);
    int ih = ohs + dh, iw = ows + dw;
    bool v = kvalid;
    const int rl = row_lin_lds[j * NT + t];
    int pix = rl + dh * p.W + dw;
    if (p.div == 2) {  // only strides 1 and 2 occur on this path (host rejects others for dgrad)
      v = v && (((ih | iw) & 1) == 0);
      ih >>= 1;
      iw >>= 1;
      pix = rl - ohs * p.W - ows + ih * p.W + iw;
    }
    v = v && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
    off = (int64_t)pix * p.x_ld;
    return v;
  };

  // ---- UT path: hardware buffer addressing ------------------------------------------------------------
  // Per row, the byte offset of the current tap's source pixel (plus this lane's 16-byte chunk) lives in a
  // VGPR for the whole tap; the slab's channel advance is the scalar soffset of the buffer load; rows whose tap
  // is out of bounds carry an offset beyond num_records, for which the hardware returns 0 (no select, no
  // branch).
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned OOB = 0x80000000u;
  int cur_tap = -1;
  unsigned tap_voff[NA];
  unsigned b_voff[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) tap_voff[j] = OOB;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = t + NT * i;
    const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
    b_voff[i] = (n0 + 4 * c4) < p.Nout ? (unsigned)(kr * p.Nout + n0 + 4 * c4) * 4u : OOB;
  }
  const int ntaps = p.K / p.C;
  const bool ktail = (p.K % BK) != 0;
  __amdgpu_buffer_rsrc_t rsrc_x, rsrc_w;
  if constexpr (VEC && UT) {
    rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  }

  // loads of slab k0 into register set S (compile-time): branch-free, invalid lanes read element 0
  auto load_AB = [&](int k0, auto SET) {
    constexpr int S = decltype(SET)::value;
    if constexpr (VEC && UT) {
      const int tap = (int)fd_div((uint32_t)k0, p.fd_c);  // uniform: scalar unit
      if (tap != cur_tap) {                                // uniform branch, taken once per Cin/32 slabs
        cur_tap = tap;
        uint32_t kh, kw;
        fd_divmod((uint32_t)tap, p.fd_kw, kh, kw);
        const int dh = (int)kh * p.k_mul, dw = (int)kw * p.k_mul;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          int64_t off;
          const bool ok = gather_elem_addr(j, dh, dw, tap < ntaps, off);
          tap_voff[j] = ok ? (unsigned)off * 4u + 16u * kc : OOB;
        }
      }
      const int soff_a = (k0 - tap * p.C) * 4;             // channel offset inside the tap, bytes (scalar)
      const int soff_b = k0 * p.Nout * 4;
      const bool kvalid = !ktail || (k0 + 4 * kc < p.K);   // only a ragged last slab (e.g. K = 728) masks lanes
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(kvalid ? tap_voff[j] : OOB), soff_a, 0);
        ra[S][j] = __builtin_bit_cast(f32x4, v);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        // the scalar offset takes no part in the hardware range check, so a ragged last slab masks its rows
        const bool bv = !ktail || (k0 + (t + NT * i) / (BN / 4) < p.K);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)(bv ? b_voff[i] : OOB), soff_b, 0);
        rb[S][i] = __builtin_bit_cast(f32x4, v);
      }
      return;
    }
    if constexpr (VEC) {
      const int k = k0 + 4 * kc;
      const bool kvalid = k < p.K;
      uint32_t tap, ci, kh, kw;
      fd_divmod((uint32_t)k, p.fd_c, tap, ci);
      fd_divmod(tap, p.fd_kw, kh, kw);
      const int dh = (int)kh * p.k_mul, dw = (int)kw * p.k_mul;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int64_t off;
        const bool v = gather_elem_addr(j, dh, dw, kvalid, off);
        const f32x4 val = ld4<TA>(px + (v ? off + ci : 0));
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ra[S][j] = v ? val : z;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = k0 + 4 * kc + e;
        const bool kvalid = k < p.K;
        uint32_t tap, ci, kh, kw;
        fd_divmod((uint32_t)k, p.fd_c, tap, ci);
        fd_divmod(tap, p.fd_kw, kh, kw);
        const int dh = (int)kh * p.k_mul, dw = (int)kw * p.k_mul;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          int64_t off;
          const bool v = gather_elem_addr(j, dh, dw, kvalid, off);
          const float val = ld1<TA>(px + (v ? off + ci : 0));
          ra[S][j][e] = v ? val : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + NT * i;
      const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
      const int k = k0 + kr, n = n0 + 4 * c4;
      if constexpr (VEC) {
        const bool v = (k < p.K) && (n < p.Nout);
        const f32x4 val = *reinterpret_cast<const f32x4*>(p.w + (v ? (int64_t)k * p.Nout + n : 0));
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        rb[S][i] = v ? val : z;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool v = (k < p.K) && (n + e < p.Nout);
          const float val = p.w[v ? (int64_t)k * p.Nout + n + e : 0];
          rb[S][i][e] = v ? val : 0.f;
        }
      }
    }
  };

  auto store_AB = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    float* a = As + buf * BM * LDA;
    float* b = Bs + buf * BK * LDB;
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<f32x4*>(a + (r0 + RS * j) * LDA + 4 * kc) = ra[S][j];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + NT * i;
      const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
      *reinterpret_cast<f32x4*>(b + kr * LDB + 4 * c4) = rb[S][i];
    }
  };

  // ---- MFMA side -------------------------------------------------------------------------------------
  const int wave = t >> 6, lane = t & 63;
  const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // The B fragment reads are 4-byte ds_read2st64 pairs whose immediate reaches 255 * 256 bytes from the address
  // register.  Measured from LDS offset 0 the upper rows of B buffer 1 lie beyond that (A takes the first
  // 36 KB), which cost three extra address registers - spilled, and reloaded behind a vmcnt(0) in the middle of
  // the MFMA block.  So the lane's B base carries the region offset itself and is hidden from constant folding.
  int b_lane_off = 2 * BM * LDA + (4 * lh) * LDB + wn + lr;  // floats from the start of LDS (= As)
  asm volatile("" : "+v"(b_lane_off));
  const float* b_lane = As + b_lane_off;
  auto compute = [&](int buf) {
    const float* a = As + buf * BM * LDA;
    const float* b = b_lane + buf * BK * LDB;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 af[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        af[i] = *reinterpret_cast<const f32x4*>(a + (wm + 32 * i + lr) * LDA + kk * 8 + 4 * lh);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float bf[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = b[(kk * 8 + e) * LDB + 32 * j];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- slab stream ------------------------------------------------------------------------------------
  // Dilated convolutions on small maps gather mostly zero padding (rate 18 on 32x32: only 39 % of the taps are in
  // bounds).  A tap for which ALL 128 rows of this tile fall outside the input contributes exactly nothing, so
  // its Cin/32 slabs are dropped from the stream: one block-wide vote per tap up front, then the slab index
  // runs over the surviving taps only (tapinfo[] in LDS maps it back to k0).
  int nslab = (p.K + BK - 1) / BK;
  bool use_map = false;
  int* tapinfo = reinterpret_cast<int*>(Bs + 2 * BK * LDB);
  if constexpr (VEC && UT) {
    if (p.skip_taps && ntaps > 1) {  // uniform
      int nact = 0;
      for (int tap = 0; tap < ntaps; ++tap) {
        uint32_t kh, kw;
        fd_divmod((uint32_t)tap, p.fd_kw, kh, kw);
        const int dh = (int)kh * p.k_mul, dw = (int)kw * p.k_mul;
        bool any = false;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          int64_t off;
          any = any || gather_elem_addr(j, dh, dw, true, off);
        }
        if (__syncthreads_or(any ? 1 : 0)) {
          if (t == 0) tapinfo[nact] = tap;
          ++nact;
        }
      }
      __syncthreads();
      nslab = __builtin_amdgcn_readfirstlane(nact) * spt_of(p);
      use_map = true;
    }
  }
  // Slab iterator: the stream only ever asks for the next slab (the tail repeats the last one), so the
  // (channel block, tap, slab-in-run) counters advance incrementally on the scalar unit.  Order: for every
  // channel block of `run` slabs, for every surviving tap, the block's slabs.  run = Cin/32 gives the plain
  // tap-major order; a short run (p.cb) makes the taps of one channel block consecutive in time, so that x is
  // fetched over the fabric once and re-read by the other taps from L2.
  const int last = nslab - 1;
  bool it_lin = true;
  int it_run = 1, it_ntap = 1, it_ci = 0, it_ti = 0, it_cb = 0, it_s = 0;
  if constexpr (VEC && UT) {
    if (ntaps > 1) {  // uniform
      it_lin = false;
      it_ntap = use_map ? nslab / spt_of(p) : ntaps;
      it_run = p.cb > 0 ? p.cb : spt_of(p);
    }
  }
  auto next_k0 = [&]() -> int {
    int k0;
    if (it_lin) {
      k0 = it_s * BK;
    } else {
      const int tap = use_map ? __builtin_amdgcn_readfirstlane(tapinfo[it_ti]) : it_ti;
      k0 = tap * p.C + (it_cb * it_run + it_ci) * BK;
    }
    if (it_s < last) {
      ++it_s;
      if (!it_lin && ++it_ci == it_run) {
        it_ci = 0;
        if (++it_ti == it_ntap) { it_ti = 0; ++it_cb; }
      }
    }
    return k0;
  };
  // Stagger (MI355X_MICROARCH "two waves per SIMD", item 9): waves i and i + nwaves/2 share a SIMD and run the
  // same program between the same barriers; the upper half defers its gather (address math + global loads)
  // until after its MFMAs, so on every SIMD one wave's VALU phase overlaps its partner's matrix phase.
  const bool late = (p.stagger != 0) && (__builtin_amdgcn_readfirstlane(t >> 6) >= (NT / 128));
  if (nslab > 0) {
  if constexpr (PF == 1) {
    load_AB(next_k0(), IC<0>{});
    store_AB(0, IC<0>{});
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
      const int buf = s & 1;
      load_AB(next_k0(), IC<0>{});  // tail reloads the last slab (unused): no branch
      compute(buf);
      store_AB(buf ^ 1, IC<0>{});
      __syncthreads();
    }
  } else {
    // two register sets: while slab s is computed from LDS, slab s+1 sits in one set (landing) and slab s+2
    // is being fetched into the other; the store of a set waits only for that set's (older) loads
    load_AB(next_k0(), IC<0>{});
    load_AB(next_k0(), IC<1>{});
    store_AB(0, IC<0>{});
    __syncthreads();
    const bool do_ld = !(p.ablate & 1), do_st = !(p.ablate & 2);
    for (int s = 0; s < nslab; s += 2) {
      const int ka = next_k0();
      const int kb = next_k0();
      if (!late && do_ld) load_AB(ka, IC<0>{});
      compute(0);
      if (late && do_ld) load_AB(ka, IC<0>{});
      if (do_st) {
        store_AB(1, IC<1>{});
        __syncthreads();
      }
      if (s + 1 >= nslab) break;
      if (!late && do_ld) load_AB(kb, IC<1>{});
      compute(1);
      if (late && do_ld) load_AB(kb, IC<1>{});
      if (do_st) {
        store_AB(0, IC<0>{});
        __syncthreads();
      }
    }
  }
  }  // nslab > 0

  // ---- epilogue: bias, relu, store (each store: 2 x 128 contiguous bytes per wave) -------------------
  const bool has_bias = (p.flags & SG_EPI_BIAS) != 0;
  const bool do_relu = (p.flags & SG_EPI_RELU) != 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn + 32 * j + lr;
    const bool cv = col < p.Nout;
    const float bv = (has_bias && cv) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (cv && row < p.M) {
          float v = acc[i][j][r] + bv;
          if (do_relu) v = fmaxf(v, 0.f);
          st1<TY>(py + (int64_t)row * p.y_ld + col, v);
        }
      }
    }
  }
}

// ---- per-tap transpose of the kernel: w[tap][ci][co] -> wt[tap][co][ci] ------------------------------
__global__ void transpose_taps_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cin, int Cout) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const float* src = w + (int64_t)tap * Cin * Cout;
  float* dst = wt + (int64_t)tap * Cin * Cout;
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows at a time
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    tile[r][tx] = (ci < Cin && co < Cout) ? src[(int64_t)ci * Cout + co] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    if (co < Cout && ci < Cin) dst[(int64_t)co * Cin + ci] = tile[tx][r];
  }
}

// ---- wgrad --------------------------------------------------------------------------------------------
struct WgradParams {
  const float* __restrict__ x;
  const float* __restrict__ dy;
  float* __restrict__ out;  // [S][K][Cout] partials (or dw itself when S == 1)
  int H, W, Cin, x_ld;
  int OH, OW, Cout, y_ld;
  int stride, dil, pad_t, pad_l;
  int K, P;
  int slabs_per_split;
  int stagger;
  uint32_t x_bytes, dy_bytes;  // extents for the buffer descriptors of the FAST path (both < 2^31)
  int skip_slabs;              // drop 32-pixel slabs that are pure padding for every tap of the tile (dilated convs)
  int KH_KW;
  int tap_inner;               // tile order (channel block, tap, column tile) instead of (tap, channel block, column tile)
  uint32_t x_plane_bytes, dy_plane_bytes;   // planes-in filter gradient (wgrad_x6_kernel<.., PIN>): bytes from one bf16 plane of
                               //    x / dy to the next; x_bytes / dy_bytes then cover all three
  BnIn bn;                     // wgrad_x6wp_kernel only: the BatchNormalization applied to x in the patch loader (mean == nullptr: none)
  int up;                      // 1: x is the source of a nearest 2x up-sampling (SG_X_UP2; wgrad_x6wp_kernel only): pixel (h, w)
                               //    of the H x W tensor is x[n, h >> 1, w >> 1, :] of the (H / 2) x (W / 2) source
  FastDiv fd_ohow, fd_ow, fd_c, fd_kw, fd_oh;
};

// FAST: stride-1 "same" convolution (H == Ho, W == Wo) with 16-byte-aligned channel runs: the source pixel of
// output pixel p under tap (dh,dw) is p + dh*W + dw, i.e. LINEAR in p, so the gather offset advances by a constant
// per slab; only the tap's bounds flags depend on (oh, ow), which are carried incrementally (no div/mod), and
// invalid taps / rows are sent to an out-of-range buffer offset (hardware returns 0).
//
// FAST == 2 ("aligned slabs"): additionally OW % 32 == 0, so a 32-pixel slab lies in ONE image row, and the
// tile's 128 r-rows lie in ONE tap (1x1 kernel, or Cin % 128 == 0).  Then everything that changes from slab to
// slab is wave-uniform: the image row (bounds of ih), the first column ow0 and the byte offset of the slab's
// source pixels, all computed on the scalar unit and passed as the buffer load's soffset; a thread keeps only
// two CONSTANT voffsets (pixel-in-slab x channel) and tests its column against W.  No per-thread running
// state, no div/mod, ~10 fewer VGPRs: the kernel runs at the 128-VGPR cap and the state used to spill, and a
// spill reload inside the slab loop waits on vmcnt(0), i.e. drains the prefetch.
template <int BN, int WGM, int WGN, int PF, bool VEC, int FAST, typename TA = float, typename TD = TA>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN / 2) void igemm_wgrad_kernel(const WgradParams p) {
  static_assert(FAST == 0 || std::is_same<TA, float>::value, "the buffer-addressed wgrad paths are fp32-only");
  const TA* __restrict__ px = reinterpret_cast<const TA*>(p.x);
  const TD* __restrict__ pdy = reinterpret_cast<const TD*>(p.dy);  // TD != TA: fp32 dy of the softmax head of a bf16 model
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LDAW = BM;  // A' slab is [32 pixels][128 r], r contiguous
  constexpr int LDB = BN;
  constexpr int NA = (BK * BM / 4) / NT;   // float4 A' chunks per thread (pixel rows pr0 + PS*j)
  constexpr int PS = NT / 32;
  constexpr int NB = (BK * BN / 4) / NT;
  static_assert(NA >= 1 && NB >= 1 && TM >= 1 && TN >= 1, "tile too small for the wave layout");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);  // [2][BK*LDAW]
  float* Bs = As + 2 * BK * LDAW;               // [2][BK*LDB]

  const int t = threadIdx.x;
  const uint32_t ntn = (p.Cout + BN - 1) / BN;
  // Workgroup order.  The hardware deals workgroups round-robin to the XCDs in linear (z, x) order; the remap
  // gives each XCD one contiguous range of (split, tile) pairs, i.e. mostly ONE pixel split, whose dy stays in
  // that XCD's L2 for the whole range.  Inside a split, with tap_inner the tiles of one 128-channel block come
  // together: all taps x all column tiles read the same x columns (shifted by the tap) and share them through
  // L2, and every XCD gets the same mix of cheap and expensive taps when padding slabs are skipped.
  uint32_t bid, split;
  {
    const uint32_t lin = blockIdx.z * gridDim.x + blockIdx.x;
    const uint32_t o = xcd_remap(lin, gridDim.x * gridDim.z);
    split = o / gridDim.x;
    bid = o - split * gridDim.x;
  }
  uint32_t tile_r, tile_n;
  if (p.tap_inner) {
    const uint32_t per = (uint32_t)p.KH_KW * ntn, ncib = (uint32_t)p.Cin / BM;
    const uint32_t cib = bid / per, rem = bid - cib * per, tap = rem / ntn;
    tile_n = rem - tap * ntn;
    tile_r = tap * ncib + cib;
  } else {
    tile_r = bid / ntn;
    tile_n = bid - tile_r * ntn;
  }
  const int rbase = tile_r * BM, n0 = tile_n * BN;

  // A' chunk owned by this thread: r = rbase + 4*rc (fixed tap / channel), pixel rows (t>>5) + PS j
  const int rc = t & 31, pr0 = t >> 5;
  constexpr int NV = VEC ? 1 : 4;
  int dh[NV], dw[NV], ci_e[NV];
  bool rvalid[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    const int r = rbase + 4 * rc + e;
    rvalid[e] = r < p.K;
    uint32_t tap, ci, kh, kw;
    fd_divmod((uint32_t)r, p.fd_c, tap, ci);
    fd_divmod(tap, p.fd_kw, kh, kw);
    dh[e] = (int)kh * p.dil - p.pad_t;
    dw[e] = (int)kw * p.dil - p.pad_l;
    ci_e[e] = (int)ci;
  }

  const int slab_begin = (int)split * p.slabs_per_split;
  const int nslab_total = (p.P + BK - 1) / BK;
  int slab_end = slab_begin + p.slabs_per_split;
  if (slab_end > nslab_total) slab_end = nslab_total;

  f32x4 ra[PF][NA];
  f32x4 rb[PF][NB];

  // ---- FAST path state --------------------------------------------------------------------------------
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  constexpr unsigned OOB = 0x80000000u;
  int f_oh[NA], f_ow[NA], f_p[NA];   // (oh, ow) and pixel index of this thread's rows in the NEXT slab to load
  unsigned f_voff[NA];               // byte offset of x[p + dh*W + dw][ci] for that slab
  unsigned b_voff[NB];
  int f_next = slab_begin * BK;      // pixel index the running state corresponds to
  __amdgpu_buffer_rsrc_t rsrc_x, rsrc_dy;
  const int adv_oh = BK / p.OW, adv_ow = BK % p.OW;
  // FAST == 2 state: constant voffsets; the descriptor's base sits SH pixels BEFORE x so that the scalar offset
  // (p0 + dh*W + dw + SH) * x_ld * 4 is non-negative whenever a lane of the slab is in bounds
  constexpr int SH = 32;
  unsigned a_voffc[NA];
  int s_dh = 0, s_dw = 0;
  if constexpr (FAST == 2) {
    // num_records spans the whole (shifted) tensor: whether or not the hardware adds soffset into its range
    // check, in-bounds lanes pass and the OOB voffset fails
    rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) - (int64_t)SH * p.x_ld, 0,
                                               (int)(p.x_bytes + (uint32_t)(SH * p.x_ld * 4)), 0x00020000);
    rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
    {
      const int tap = rbase / p.Cin;  // uniform: the whole tile lies in this tap
      uint32_t kh, kw;
      fd_divmod((uint32_t)tap, p.fd_kw, kh, kw);
      s_dh = (int)kh * p.dil - p.pad_t;
      s_dw = (int)kw * p.dil - p.pad_l;
    }
#pragma unroll
    for (int j = 0; j < NA; ++j)
      a_voffc[j] = rvalid[0] ? (unsigned)((pr0 + PS * j) * p.x_ld + ci_e[0]) * 4u : OOB;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + NT * i;
      const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
      b_voff[i] = (n0 + 4 * c4) < p.Cout ? (unsigned)(kr * p.y_ld + n0 + 4 * c4) * 4u : OOB;
    }
  }
  if constexpr (FAST == 1) {
    rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int pp = f_next + pr0 + PS * j;
      uint32_t n, rem, oh, ow;
      fd_divmod((uint32_t)pp, p.fd_ohow, n, rem);
      fd_divmod(rem, p.fd_ow, oh, ow);
      f_oh[j] = (int)oh; f_ow[j] = (int)ow; f_p[j] = pp;
      f_voff[j] = (unsigned)((pp + dh[0] * p.W + dw[0]) * p.x_ld + ci_e[0]) * 4u;  // may wrap when the tap is invalid
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + NT * i;
      const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
      b_voff[i] = (n0 + 4 * c4) < p.Cout ? (unsigned)(kr * p.y_ld + n0 + 4 * c4) * 4u : OOB;
    }
  }

  auto load_AB = [&](int p0, auto SET) {
    constexpr int S = decltype(SET)::value;
    if constexpr (FAST == 2) {
      uint32_t q, ow0, n_, oh;
      fd_divmod((uint32_t)p0, p.fd_ow, q, ow0);   // all scalar: p0 is wave-uniform
      fd_divmod(q, p.fd_oh, n_, oh);
      const int ih = (int)oh + s_dh;
      const bool row_ok = (unsigned)ih < (unsigned)p.H;
      const int soff_a = row_ok ? (p0 + s_dh * p.W + s_dw + SH) * p.x_ld * 4 : 0;
      const int col0 = (int)ow0 + s_dw + pr0;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const bool v = row_ok && ((unsigned)(col0 + PS * j) < (unsigned)p.W);
        const u32x4 val = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(v ? a_voffc[j] : OOB), soff_a, 0);
        ra[S][j] = __builtin_bit_cast(f32x4, val);
      }
      const int soff_b = p0 * p.y_ld * 4;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const u32x4 val = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)b_voff[i], soff_b, 0);  // P % 32 == 0
        rb[S][i] = __builtin_bit_cast(f32x4, val);
      }
      return;
    }
    if constexpr (FAST == 1) {
      // the slab stream is sequential except for the repeated (unused) tail slab: advance only when it moves on
      if (p.skip_slabs) {  // uniform: slabs are visited with gaps, so derive the state from p0 directly
        if (p0 != f_next) {
          f_next = p0;
#pragma unroll
          for (int j = 0; j < NA; ++j) {
            const int pp = p0 + pr0 + PS * j;
            uint32_t n, rem, oh, ow;
            fd_divmod((uint32_t)pp, p.fd_ohow, n, rem);
            fd_divmod(rem, p.fd_ow, oh, ow);
            f_oh[j] = (int)oh; f_ow[j] = (int)ow; f_p[j] = pp;
            f_voff[j] = (unsigned)((pp + dh[0] * p.W + dw[0]) * p.x_ld + ci_e[0]) * 4u;
          }
        }
      } else if (p0 != f_next) {  // uniform
        f_next = p0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          f_p[j] += BK;
          f_voff[j] += (unsigned)(BK * p.x_ld) * 4u;
          f_ow[j] += adv_ow;
          f_oh[j] += adv_oh;
          if (f_ow[j] >= p.OW) { f_ow[j] -= p.OW; f_oh[j] += 1; }
          while (f_oh[j] >= p.OH) f_oh[j] -= p.OH;
        }
      }
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int ih = f_oh[j] + dh[0], iw = f_ow[j] + dw[0];
        const bool v = rvalid[0] && (f_p[j] < p.P) && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
        const u32x4 val = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(v ? f_voff[j] : OOB), 0, 0);
        ra[S][j] = __builtin_bit_cast(f32x4, val);
      }
      const int soff_b = p0 * p.y_ld * 4;
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const bool bv = (p0 + (t + NT * i) / (BN / 4)) < p.P;  // soffset is not range-checked
        const u32x4 val = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)(bv ? b_voff[i] : OOB), soff_b, 0);
        rb[S][i] = __builtin_bit_cast(f32x4, val);
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int pp = p0 + pr0 + PS * j;
      const bool pv = pp < p.P;
      uint32_t n, rem, oh, ow;
      fd_divmod((uint32_t)(pv ? pp : 0), p.fd_ohow, n, rem);
      fd_divmod(rem, p.fd_ow, oh, ow);
      const int pixbase = n * p.H * p.W;
      if constexpr (VEC) {
        const int ih = (int)oh * p.stride + dh[0], iw = (int)ow * p.stride + dw[0];
        const bool v = pv && rvalid[0] && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
        const f32x4 val = ld4<TA>(px + (v ? (int64_t)(pixbase + ih * p.W + iw) * p.x_ld + ci_e[0] : 0));
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ra[S][j] = v ? val : z;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ih = (int)oh * p.stride + dh[e], iw = (int)ow * p.stride + dw[e];
          const bool v = pv && rvalid[e] && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
          const float val = ld1<TA>(px + (v ? (int64_t)(pixbase + ih * p.W + iw) * p.x_ld + ci_e[e] : 0));
          ra[S][j][e] = v ? val : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + NT * i;
      const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
      const int pp = p0 + kr, n = n0 + 4 * c4;
      if constexpr (VEC) {
        const bool v = (pp < p.P) && (n < p.Cout);
        const f32x4 val = ld4<TD>(pdy + (v ? (int64_t)pp * p.y_ld + n : 0));
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        rb[S][i] = v ? val : z;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool v = (pp < p.P) && (n + e < p.Cout);
          const float val = ld1<TD>(pdy + (v ? (int64_t)pp * p.y_ld + n + e : 0));
          rb[S][i][e] = v ? val : 0.f;
        }
      }
    }
  };

  auto store_AB = [&](int buf, auto SET) {
    constexpr int S = decltype(SET)::value;
    float* a = As + buf * BK * LDAW;
    float* b = Bs + buf * BK * LDB;
#pragma unroll
    for (int j = 0; j < NA; ++j) *reinterpret_cast<f32x4*>(a + (pr0 + PS * j) * LDAW + 4 * rc) = ra[S][j];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + NT * i;
      const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
      *reinterpret_cast<f32x4*>(b + kr * LDB + 4 * c4) = rb[S][i];
    }
  };

  const int wave = t >> 6, lane = t & 63;
  const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf) {
    const float* a = As + buf * BK * LDAW;
    const float* b = Bs + buf * BK * LDB;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const int pk = 2 * kk + lh;
      float af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = a[pk * LDAW + wm + 32 * i + lr];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = b[pk * LDB + wn + 32 * j + lr];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- slab stream: with dilation, whole 32-pixel slabs can be padding for every tap this tile covers
  // (rate 18 on a 32x32 map: 18 of 32 image rows per vertical tap); those are dropped.  Each thread classifies
  // slabs in parallel, thread 0 compacts the survivors into an LDS list that the stream then walks.
  int nslab = slab_end - slab_begin;  // >= 1 by construction of the split plan
  int* slist = reinterpret_cast<int*>(Bs + 2 * BK * LDB);  // [1 + 1024]
  bool use_list = false;
  if constexpr (FAST) {
    if (p.skip_slabs && nslab <= 1024) {  // uniform
      use_list = true;
      const int tap_lo = rbase / p.Cin;
      int tap_hi = (rbase + BM - 1 < p.K ? rbase + BM - 1 : p.K - 1) / p.Cin;
      if (tap_hi >= p.KH_KW) tap_hi = p.KH_KW - 1;
      int* flags = slist + 1 + 1024;  // [1024]
      for (int i = t; i < nslab; i += NT) {
        const int pa = (slab_begin + i) * BK;
        int pb = pa + BK - 1;
        if (pb > p.P - 1) pb = p.P - 1;
        uint32_t na, ra_, nb, rb_, oha, ohb, tmp;
        fd_divmod((uint32_t)pa, p.fd_ohow, na, ra_);
        fd_divmod((uint32_t)pb, p.fd_ohow, nb, rb_);
        fd_divmod(ra_, p.fd_ow, oha, tmp);
        fd_divmod(rb_, p.fd_ow, ohb, tmp);
        bool act = na != nb;
        for (int tap = tap_lo; tap <= tap_hi && !act; ++tap) {
          uint32_t kh, kw;
          fd_divmod((uint32_t)tap, p.fd_kw, kh, kw);
          const int ddh = (int)kh * p.dil - p.pad_t, ddw = (int)kw * p.dil - p.pad_l;
          act = ((int)ohb + ddh >= 0) && ((int)oha + ddh < p.H) && (ddw > -p.W) && (ddw < p.W);
        }
        flags[i] = act ? 1 : 0;
      }
      __syncthreads();
      if (t == 0) {
        int n = 0;
        for (int i = 0; i < nslab; ++i)
          if (flags[i]) slist[1 + n++] = slab_begin + i;
        slist[0] = n;
      }
      __syncthreads();
      nslab = slist[0];
    }
  }
  auto slab_of = [&](int i) -> int { return use_list ? slist[1 + i] : slab_begin + i; };
  const int lasti = nslab - 1;
  if (nslab > 0) {
    if constexpr (PF == 1) {
      load_AB(slab_of(0) * BK, IC<0>{});
      store_AB(0, IC<0>{});
      __syncthreads();
      for (int s = 0; s < nslab; ++s) {
        const int buf = s & 1;
        load_AB(slab_of(s < lasti ? s + 1 : lasti) * BK, IC<0>{});
        compute(buf);
        store_AB(buf ^ 1, IC<0>{});
        __syncthreads();
      }
    } else {
      load_AB(slab_of(0) * BK, IC<0>{});
      load_AB(slab_of(1 < lasti ? 1 : lasti) * BK, IC<1>{});
      store_AB(0, IC<0>{});
      __syncthreads();
      const bool late = (p.stagger != 0) && (__builtin_amdgcn_readfirstlane(t >> 6) >= (NT / 128));
      for (int s = 0; s < nslab; s += 2) {
        const int pa = slab_of(s + 2 < lasti ? s + 2 : lasti) * BK, pb = slab_of(s + 3 < lasti ? s + 3 : lasti) * BK;
        if (!late) load_AB(pa, IC<0>{});
        compute(0);
        if (late) load_AB(pa, IC<0>{});
        store_AB(1, IC<1>{});
        __syncthreads();
        if (s + 1 >= nslab) break;
        if (!late) load_AB(pb, IC<1>{});
        compute(1);
        if (late) load_AB(pb, IC<1>{});
        store_AB(0, IC<0>{});
        __syncthreads();
      }
    }
  }

  float* out = p.out + (int64_t)split * p.K * p.Cout;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn + 32 * j + lr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (col < p.Cout && row < p.K) out[(int64_t)row * p.Cout + col] = acc[i][j][r];
      }
    }
  }
}

// out[i] = sum_z part[z][i]  in fixed z order (V = 4: 16-byte accesses, n % 4 == 0 and aligned pointers; the z loop is
// unrolled so that the partial rows' loads are in flight together - the adds stay in z order)
template <int V>
__global__ __launch_bounds__(256) void reduce_splits_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n, int S) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * V;
  for (; i < n; i += stride) {
    float s[V];
    ldv<V>(part + i, s);
#pragma unroll 8
    for (int z = 1; z < S; ++z) {
      float v[V];
      ldv<V>(part + (int64_t)z * n + i, v);
#pragma unroll
      for (int k = 0; k < V; ++k) s[k] += v[k];
    }
    stv<V>(out + i, s);
  }
}

// The same with FOUR lanes per output quad (round 4; for S >= 8): lane zl adds slabs zl, zl + 4, ... in order, the four lane sums
// are added ((0 + 1) + 2) + 3 through LDS - deterministic.  A 21-slab reduction of a 728 x 728 gradient (44 MB) is 518
// workgroups of the kernel above whose threads each walk 21 loads, eight in flight: 20 - 29 us, 2.2 TB/s
// (profiles/r04_bench_kernel_stats_final.csv: 90 launches per step); four times the threads walk five loads each.
__global__ __launch_bounds__(256) void reduce_splits_z4_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n, int S) {
  __shared__ f32x4 red[4][64];
  const int q = threadIdx.x & 63, zl = threadIdx.x >> 6;
  const int64_t stride = (int64_t)gridDim.x * 64 * 4;
  for (int64_t i0 = (int64_t)blockIdx.x * 64 * 4; i0 < n; i0 += stride) {   // (uniform trip count per workgroup: barriers inside)
    const int64_t i = i0 + (int64_t)q * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
#pragma unroll 4
      for (int z = zl; z < S; z += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(part + (int64_t)z * n + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += v[k];
      }
    }
    red[zl][q] = s;
    __syncthreads();
    if (zl == 0 && i < n) {
      f32x4 t = red[0][q];
#pragma unroll
      for (int l = 1; l < 4; ++l)
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] += red[l][q][k];
      *reinterpret_cast<f32x4*>(out + i) = t;
    }
    __syncthreads();
  }
}

// bias gradient = column sums of dy[rows][C] (pixel stride ld), through the fixed-order segmented reducer
template <typename T>
struct ColSumOp {
  static constexpr int NOUT = 1;
  const T* __restrict__ a;
  float* out;
  int ld;
  template <int V>
  __device__ __forceinline__ void accum(int, int64_t r, int c, float (&acc)[1][V]) const {
    float v[V];
    ldv<V>(a + r * ld + c, v);
#pragma unroll
    for (int k = 0; k < V; ++k) acc[0][k] += v[k];
  }
  __device__ __forceinline__ void finalize(int, int c, const double (&s)[1]) const { out[c] = (float)s[0]; }
};

size_t colsum_ws_bytes(int num_cus, int64_t rows, int C) {
  const SegPlan a = seg_plan<1>(num_cus, 1, rows, C, true), b = seg_plan<1>(num_cus, 1, rows, C, false);
  return a.part_bytes > b.part_bytes ? a.part_bytes : b.part_bytes;
}

template <typename T>
int launch_colsum(int num_cus, const T* dy, int64_t rows, int C, int ld, float* out, float* part, hipStream_t st) {
  const bool vec = (C % 4 == 0) && (ld % 4 == 0) && sg_aligned16(dy);
  const SegPlan pl = seg_plan<1>(num_cus, 1, rows, C, vec);
  ColSumOp<T> op;
  op.a = dy; op.out = out; op.ld = ld;
  return seg_reduce_launch(op, pl, 1, rows, C, part, st, "colsum");
}

template <typename KernelT>
int set_dyn_lds(KernelT k, size_t bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    sg_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

// Experiment switch (A/B runs of the micro-benchmark): SG_CONV_VARIANT bit 0 = 8-wave workgroups,
// bit 1 = two-slab prefetch, bit 2 = staggered wave halves (needs bit 1).  Default = all (7).
int conv_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SG_CONV_VARIANT");
    v = e ? atoi(e) & 7 : 7;
  }
  return v;
}

// x6 == true: the switches as the x6 kernels use them.  Re-measured with those kernels (profiles/r01_ab_l2_orders.txt,
// second part): the grouped order takes the ASPP dgrad fetch from 1998 to 485 MiB and the tap-inner wgrad order from
// 795 to 279 MiB at 0-3 % of kernel time (the fp32 kernels paid 6-13 %), so both are on there: default 7.
int conv_l2(bool x6 = false) {
  static int v = -1, vx = -1;
  if (vx < 0) {
    const char* e = getenv("SG_CONV_L2");
    vx = e ? atoi(e) & 7 : 7;
  }
  if (x6) return vx;
  if (v < 0) {
    const char* e = getenv("SG_CONV_L2");
    // default 2: measured on the ASPP convs (scripts/ab_l2.sh, 30 iterations, two interleaved rounds):
    //   bit 1 (channel-block K order): forward fetch 1039 -> 451 MiB per launch, time unchanged      -> on
    //   bit 0 (grouped tile order):    dgrad fetch 1808 -> 461 MiB, but 6-13 % SLOWER               -> off
    //   bit 2 (wgrad tap-inner order): wgrad fetch 1018 -> 314 MiB, but 2-8 % SLOWER                 -> off
    // The re-reads are served by the 256 MiB Infinity Cache (operands total < 150 MiB), not by HBM, and the
    // orders that remove them make many CUs hit the same L2 lines at the same time.
    v = e ? atoi(e) & 7 : 2;
  }
  return v;
}

template <int BN, int WGM, int WGN, int PF, bool VEC, bool UT, typename TA = float, typename TY = TA>
int launch_igemm_ut(const IgemmParams& p, hipStream_t st) {
  // + tapinfo[64] + row_lin[NA][NT] (NA * NT = BM * BK / 4 ints)
  constexpr size_t lds = (size_t)(2 * BM * LDA + 2 * BK * BN) * sizeof(float) + 256 + (size_t)(BM * BK / 4) * sizeof(int);
  static bool attr_done = false;  // idempotent; racing threads set the same value
  if (!attr_done) {
    int rc = set_dyn_lds(igemm_conv_kernel<BN, WGM, WGN, PF, VEC, UT, TA, TY>, lds);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.M, BM) * sg_cdiv(p.Nout, BN);
  if (tiles <= 0 || tiles > 0x7fffffff) {
    sg_set_error("igemm: bad tile count %lld", (long long)tiles);
    return SG_EINVAL;
  }
  hipLaunchKernelGGL((igemm_conv_kernel<BN, WGM, WGN, PF, VEC, UT, TA, TY>), dim3((unsigned)tiles), dim3(64 * WGM * WGN), lds, st, p);
  SG_LAUNCH_CHECK("igemm_conv_kernel");
  return 0;
}

template <int BN, int WGM, int WGN, int PF, bool VEC>
int launch_igemm(const IgemmParams& p, hipStream_t st) {
  if constexpr (VEC) {
    // slab never straddles a tap, and both operands fit a 2 GiB buffer descriptor
    const bool ut = ((p.C % BK == 0) || (p.K == p.C)) && p.x_bytes != 0 && p.w_bytes != 0;
    if (ut) return launch_igemm_ut<BN, WGM, WGN, PF, true, true>(p, st);
  }
  return launch_igemm_ut<BN, WGM, WGN, PF, VEC, false>(p, st);
}

// Tile width by wave quantisation: two workgroups fit a CU (LDS), so one "wave" of the grid is 2*CUs tiles; pick
// the BN in {128, 64} whose tile count wastes the least of its last wave and of its last column tile (e.g.
// M = 16384, N = 728: 128-wide = 768 tiles = 1.5 waves (75 %), 64-wide = 1536 tiles = 3.0 waves (100 %)).
int pick_bn(int64_t M, int N, int num_cus) {
  if (N <= 32) return 32;
  if (N <= 64) return 64;
  // quantum = one workgroup per CU: an 8-wave workgroup keeps a CU's matrix pipes fed on its own, and tiles
  // beyond the resident ones are dispatched as CUs free up, so the makespan is ceil(tiles / CUs) tile-times
  const int64_t slots = (int64_t)num_cus;
  double best = -1.0;
  int best_bn = 128;
  for (int bn : {128, 64}) {
    const int64_t tiles = sg_cdiv(M, BM) * sg_cdiv(N, bn);
    const double eff = (double)tiles / (double)(sg_cdiv(tiles, slots) * slots) * (double)N / (double)(sg_cdiv(N, bn) * bn) *
                       (bn == 128 ? 1.0 : 0.93);  // the wider tile re-reads A half as often
    if (eff > best) { best = eff; best_bn = bn; }
  }
  return best_bn;
}

// fields shared by the fp32 and the x6 kernel: padding-tap elimination, tile order, K order
void plan_common(IgemmParams& p, bool vec, int bn, bool x6 = false, int eb = 4) {
  static int noskip = -1;
  if (noskip < 0) noskip = getenv("SG_CONV_NOSKIP") ? 1 : 0;  // A/B switch for the padding-tap elimination
  const int ntaps = p.K / p.C, spt = p.C / BK;
  p.skip_taps = (!noskip && (p.k_mul > 1 || p.k_mul < -1) && ntaps > 1 && ntaps <= 64) ? 1 : 0;
  if (p.perm2) p.skip_taps = 1;  // parity-pure tiles: taps of the other parities have no valid row (also a lone 1x1 tap)
  // L2 locality (A/B switch SG_CONV_L2: bit 0 grouped tile order, bit 1 channel-block K order, bit 2 wgrad order).
  // An XCD owns 1/8 of the tiles and with them about 1/8 of the A operand's pixels.
  const int64_t ntn = sg_cdiv(p.Nout, bn);
  const int64_t a_per_xcd = p.x_bytes ? (int64_t)p.x_bytes / 8 : (1ll << 40);
  p.group_m = ((conv_l2(x6) & 1) && ntn >= 4) ? (a_per_xcd <= (2ll << 20) ? 16 : 8) : 1;
  static const int gm_force = getenv("SG_CONV_GM") ? atoi(getenv("SG_CONV_GM")) : 0;   // experiment: row tiles per group
  if (gm_force > 0 && ntn >= 2) p.group_m = gm_force;
  const bool ut = vec && (p.C % BK == 0) && p.x_bytes != 0 && p.w_bytes != 0;
  // decided from ONE image's footprint, never from the batch: the K order fixes the rounding order, and
  // inference must not depend on how many tiles travel together (tests/test_fullsize_gpu.py)
  const int64_t img_bytes = (int64_t)p.H * p.W * p.x_ld * eb;
  static int cbv = -1;  // experiment switch SG_CONV_CB: slabs per channel block of the K order (default 4)
  if (cbv < 0) cbv = getenv("SG_CONV_CB") ? atoi(getenv("SG_CONV_CB")) : 4;
  p.cb = ((conv_l2(x6) & 2) && ut && ntaps > 1 && spt > cbv && spt % cbv == 0 && img_bytes > (2ll << 20)) ? cbv : 0;
}

// set while a launch beyond 2 GiB runs as sub-batches of whole images (sg_conv2d_fwd_ws / sg_conv2d_dgrad recursion)
thread_local bool g_sub_batch = false;

#include "conv_x6.h"
#include "conv_x6p.h"
#include "conv_b16.h"
#include "conv_x6wp.h"
#include "conv_pw.h"
#include "conv_b16w.h"
#include "conv_x6w.h"

template <int NPL, typename TA>
int dispatch_x6(const IgemmParams& p_in, int num_cus, hipStream_t st) {
  IgemmParams p = p_in;
  int bn = pick_bn(p.M, p.Nout, num_cus);
  {
    // experiment switch: 128-wide tiles that give at most one tile per CU are halved (two workgroups per CU)
    static const bool bn64 = getenv("SG_X6_BN64") != nullptr;
    if (bn64 && bn == 128 && sg_cdiv(p.M, BM) * sg_cdiv(p.Nout, 128) <= (int64_t)num_cus) bn = 64;
  }
  plan_common(p, true, bn, true, EL<TA>::BYTES);
  {
    static int il = -1;  // SG_X6_INTERLEAVE=0: staggered halves instead of the hand-interleaved step (A/B switch)
    if (il < 0) il = getenv("SG_X6_INTERLEAVE") ? atoi(getenv("SG_X6_INTERLEAVE")) : 1;
    p.stagger = il ? 2 : 0;
  }
  {
    static int abl = -1;
    if (abl < 0) {
      const char* e = getenv("SG_X6_ABLATE");
      abl = e ? atoi(e) & 15 : 0;
    }
    p.ablate = abl;
  }
  // Structure by tile count (measured, profiles/r01_ab_x6.txt, r01_ab_x6_interleave.txt, r01_ab_tiles1024.txt): up to
  // 3 tiles per CU the double-buffered one-workgroup-per-CU form with the hand-interleaved step wins (ASPP forward
  // 1.02 -> 0.89 ms, pointwise 728 forward 0.150 -> 0.137 ms); from 4 tiles per CU - i.e. two full rounds of the
  // two-workgroups-per-CU single-buffer form - that one does (1024 tiles: 64x64 512->256 forward 887 vs 930 us,
  // 1024->1024 pointwise 233 vs 252 us; ASPP dgrad 0.83 vs 0.94 ms, decoder 128->64 dgrad 0.92 vs 1.03 ms).
  // SG_X6_VARIANT overrides.
  const int64_t tiles = sg_cdiv(p.M, BM) * sg_cdiv(p.Nout, bn);
  int var = x6_variant();
  if (var < 0) var = (tiles <= 3 * (int64_t)num_cus) ? 1 : 0;
  if constexpr (NPL == 1) {  // the one-plane (bf16 product) kernels come in the two 8-wave structures only
    if (bn == 128) return (var & 1) ? launch_x6<128, 2, 4, 2, 1, TA>(p, st) : launch_x6<128, 2, 4, 1, 1, TA>(p, st);
    if (bn == 64) return (var & 1) ? launch_x6<64, 4, 2, 2, 1, TA>(p, st) : launch_x6<64, 4, 2, 1, 1, TA>(p, st);
    return (var & 1) ? launch_x6<32, 4, 1, 2, 1, TA>(p, st) : launch_x6<32, 4, 1, 1, 1, TA>(p, st);
  } else {
  // 16x16x32 products for the multi-tap convolutions (conv_x6_kernel, MF): a property of the layer, never of the batch.
  // SG_X6_MF16=0 keeps every launch on 32x32x16 (A/B runs).
  static const bool mf16_on = !(getenv("SG_X6_MF16") && atoi(getenv("SG_X6_MF16")) == 0);
  const bool mf = mf16_on && p.K != p.C;
  if (bn == 128) {
    if (mf && var == 0) return launch_x6<128, 2, 4, 1, 3, float, 1>(p, st);
    if (mf && var == 1) return launch_x6<128, 2, 4, 2, 3, float, 1>(p, st);
    switch (var) {
      case 0: return launch_x6<128, 2, 4, 1>(p, st);
      case 1: return launch_x6<128, 2, 4, 2>(p, st);
      case 2: return launch_x6<128, 2, 2, 1>(p, st);
      default: return launch_x6<128, 2, 2, 2>(p, st);
    }
  }
  if (bn == 64) {
    if (mf && var == 0) return launch_x6<64, 4, 2, 1, 3, float, 1>(p, st);
    if (mf && var == 1) return launch_x6<64, 4, 2, 2, 3, float, 1>(p, st);
    switch (var) {
      case 0: return launch_x6<64, 4, 2, 1>(p, st);
      case 1: return launch_x6<64, 4, 2, 2>(p, st);
      case 2: return launch_x6<64, 2, 2, 1>(p, st);
      default: return launch_x6<64, 2, 2, 2>(p, st);
    }
  }
  if (mf) return (var & 1) ? launch_x6<32, 4, 1, 2, 3, float, 1>(p, st) : launch_x6<32, 4, 1, 1, 3, float, 1>(p, st);
  return (var & 1) ? launch_x6<32, 4, 1, 2>(p, st) : launch_x6<32, 4, 1, 1>(p, st);
  }
}

// split the weights into the x6 planes (in `ws`) and run the x6 kernel
// NPL = 3: the exact fp32 emulation; NPL = 1: bf16 products (TA = float: fp32 storage rounded on the way into LDS,
// TA = bf16_t: SG_BF16 storage).  p.C is the depth of one tap of the reduction (Cin forward, Cout dgrad).
template <int NPL, typename TA>
int run_x6(IgemmParams& p, const float* w, bool dgrad, int Cin, int Cout, int KH, int KW, void* ws, int num_cus,
           hipStream_t st, bool prepared = false) {
  // prepared: `ws` already holds this launch's weight planes (sg_prepare_planes, once per optimiser step), no split here
  if (p.res) {  // only the slab kernels below add a collected gradient (callers ask sg_conv2d_planes_job: kind 1)
    bool patch = false, wide = false;
    if constexpr (NPL == 3) patch = x6p_ok(p, KH, KW);
    if constexpr ((NPL == 3 && std::is_same<TA, float>::value) || (NPL == 1 && !std::is_same<TA, float>::value))
      wide = pw_wide_ok(p, EL<TA>::BYTES);
    if (patch || wide) {
      IgemmParams q = p;
      q.res = nullptr;
      sg_set_error("sg_conv2d_dgrad_acc: this launch takes the %s kernel, which does not add a collected gradient "
                   "(reduction channels per tap %d, K %d, output columns %d, %d x %d outputs per image; planes-in plan without res: %d)",
                   patch ? "patch" : "wide pointwise", p.C, p.K, p.Nout, p.OH, p.OW, x6w_plan(q));
      return SG_EUNSUPPORTED;
    }
  }
  if constexpr (NPL == 3) {
    if (x6p_ok(p, KH, KW)) return run_x6p(p, w, dgrad, Cin, Cout, ws, num_cus, st, prepared);
  }
  // 1x1 / stride 1 with enough columns for 384-wide tiles (the 728-wide middle flow and the exit flow): conv_pw.h, with its
  // own k-block-major plane layout
  if constexpr ((NPL == 3 && std::is_same<TA, float>::value) || (NPL == 1 && !std::is_same<TA, float>::value)) {
    if (const int wbn = pw_wide_bn(p, EL<TA>::BYTES)) {
      const int K = p.K, N = p.Nout;
      p.Kpad = pw_kpad(K, NPL);
      p.Npad = pw_npad(N, wbn);
      p.wq = (const unsigned short*)ws;
      p.w_bytes = (uint32_t)pw_planes_bytes(K, N, NPL, wbn);
      if (!prepared) {
        dim3 grid((unsigned)sg_cdiv(p.Kpad, 32), (unsigned)sg_cdiv(p.Npad, 32));
        hipLaunchKernelGGL(split3_weights_kernel, grid, dim3(256), 0, st, w, (unsigned short*)ws, K, N, p.Kpad, p.Npad, K,
                           Cin * Cout, dgrad ? 1 : Cout, dgrad ? Cout : 1, NPL, K, pw_kd(NPL));
        SG_LAUNCH_CHECK("split3_weights_kernel");
      }
      if constexpr (NPL == 3) {
        if (p.bnb.x) {   // the BatchNormalization backward in the A path (sg_conv2d_dgrad_bnb)
          if (wbn == 512) {
            sg_set_error("pw_wide: no BatchNormalization-backward form of the 512-wide tile");
            return SG_EUNSUPPORTED;
          }
          return wbn == 256 ? launch_pw_wide<NPL, TA, 256, true>(p, st) : launch_pw_wide<NPL, TA, 384, true>(p, st);
        }
        if (wbn == 512) return launch_pw_wide<NPL, TA, 512>(p, st);
      }
      return wbn == 256 ? launch_pw_wide<NPL, TA, 256>(p, st) : launch_pw_wide<NPL, TA, 384>(p, st);
    }
  }
  if (p.bnb.x) {   // only the wide pointwise kernel above evaluates the BatchNormalization backward in its A path
    sg_set_error("sg_conv2d_dgrad_bnb: this launch does not take the wide pointwise kernel");
    return SG_EUNSUPPORTED;
  }
  const int Ck = p.C;
  int Ckp = Ck;
  if (Ck % BK != 0 && p.K != Ck) {  // virtual channel padding (x6_ok admitted the shape): whole slabs inside one tap
    Ckp = x6_vpad_c(Ck);
    p.K = (p.K / Ck) * Ckp;
    p.C = Ckp;
    p.fd_c = make_fastdiv((uint32_t)Ckp);
  }
  const int K = p.K, N = p.Nout;
  p.kd = x6_plane_kd(NPL == 1 && !std::is_same<TA, float>::value, p.C, p.K == p.C);
  p.Kpad = x6_kpad(K, p.kd);
  p.Npad = x6_npad(N);
  p.wq = (const unsigned short*)ws;
  p.w_bytes = (uint32_t)x6_planes_bytes(K, N, NPL, p.kd);
  if (!prepared) {
    dim3 grid((unsigned)(p.Kpad / 32), (unsigned)(p.Npad / 32));
    if (!dgrad)
      hipLaunchKernelGGL(split3_weights_kernel, grid, dim3(256), 0, st, w, (unsigned short*)ws, K, N, p.Kpad, p.Npad, Ck,
                         Cin * Cout, Cout, 1, NPL, Ckp, p.kd);
    else
      hipLaunchKernelGGL(split3_weights_kernel, grid, dim3(256), 0, st, w, (unsigned short*)ws, K, N, p.Kpad, p.Npad, Ck,
                         Cin * Cout, 1, Cout, NPL, Ckp, p.kd);
    SG_LAUNCH_CHECK("split3_weights_kernel");
  }
  if constexpr (NPL == 3 && std::is_same<TA, float>::value) {
    // the dilated long-K convolutions: the activation split once into bf16 planes, then 128 x 256 tiles with both operands by
    // LDS-DMA (conv_x6w.h); planes of A and split-K partial slabs lie behind the weight planes
    const int S = p.kd == XW_KD ? x6w_plan(p) : 0;
    if (S > 0) {
      const size_t planes_end = ((size_t)p.w_bytes + 255) & ~(size_t)255;
      const size_t need = planes_end + x6w_scratch_bytes(p, S);
      if (p.ws_room != SIZE_MAX && p.ws_room < need) {
        sg_set_error("conv_x6w: workspace %zu < %zu (weight planes + activation planes + %d partial slabs)", p.ws_room, need, S);
        return SG_EWORKSPACE;
      }
      IgemmParams q = p;
      plan_common(q, true, 128, true, 4);
      {
        static const int abl = getenv("SG_X6W_ABLATE") ? atoi(getenv("SG_X6W_ABLATE")) : 0;
        q.ablate = abl;
      }
      return launch_x6w(q, S, (char*)ws + planes_end, st);
    }
  }
  if constexpr (NPL == 1 && !std::is_same<TA, float>::value) {
    static const bool deep = !(getenv("SG_B16_DEEP") && atoi(getenv("SG_B16_DEEP")) == 0);  // A/B switch
    if (deep) {
      // long reductions with 192+ output columns: 256 x 256 tiles, both operands by LDS-DMA, split-K where the tiles are few
      // (conv_b16w.h); its scratch lies behind the planes
      const int S = p.kd == BW_KD ? b16w_plan(p) : 0;
      if (S > 0) {
        const size_t planes_end = ((size_t)p.w_bytes + 255) & ~(size_t)255;
        const size_t need = planes_end + b16w_scratch_bytes(S, p.M, p.Nout);
        if (S > 1 && p.ws_room != SIZE_MAX && p.ws_room < need) {
          sg_set_error("conv_b16w: workspace %zu < %zu (weight planes + %d split-K partial slabs)", p.ws_room, need, S);
          return SG_EWORKSPACE;
        }
        IgemmParams q = p;
        plan_common(q, true, 128, true, 2);
        return launch_b16w(q, S, S > 1 ? reinterpret_cast<float*>((char*)ws + planes_end) : nullptr, st);
      }
      return dispatch_b16(p, num_cus, st);
    }
  }
  return dispatch_x6<NPL, TA>(p, num_cus, st);
}

// workspace of the weight planes for a launch with `taps` taps of depth C (virtual padding included), N columns
inline size_t x6_ws_bytes(int taps, int C, int N) {
  const int k = taps > 1 ? taps * x6_vpad_c(C) : C;
  const size_t rows = x6_planes_bytes(k, N, 3, 64);   // (the deepest k-block any kernel asks for)
  const size_t w384 = taps == 1 ? pw_planes_bytes(C, N, 3, 384) : 0, w256 = taps == 1 ? pw_planes_bytes(C, N, 3, 512) : 0;   // (512 >= 256)
  const size_t wide = w384 > w256 ? w384 : w256;   // the wide pointwise kernel pads N to its tile width
  return rows > wide ? rows : wide;
}

int dispatch_igemm(const IgemmParams& p_in, bool vec, int num_cus, hipStream_t st) {
  IgemmParams p = p_in;
  const int bn = pick_bn(p.M, p.Nout, num_cus);
  plan_common(p, vec, bn);
  const int var = conv_variant() & 3;
  p.stagger = (conv_variant() >> 2) & 1;
  {
    static int abl = -1;
    if (abl < 0) {
      const char* e = getenv("SG_CONV_ABLATE");
      abl = e ? atoi(e) & 3 : 0;
    }
    p.ablate = abl;
  }
  if (!vec) {
    if (bn == 128) return launch_igemm<128, 2, 4, 1, false>(p, st);
    if (bn == 64) return launch_igemm<64, 4, 2, 1, false>(p, st);
    return launch_igemm<32, 4, 1, 1, false>(p, st);
  }
  if (bn == 128) {
    switch (var) {
      case 0: return launch_igemm<128, 2, 2, 1, true>(p, st);
      case 1: return launch_igemm<128, 2, 4, 1, true>(p, st);
      case 2: return launch_igemm<128, 2, 2, 2, true>(p, st);
      default: return launch_igemm<128, 2, 4, 2, true>(p, st);
    }
  }
  if (bn == 64) {
    switch (var) {
      case 0: return launch_igemm<64, 2, 2, 1, true>(p, st);
      case 1: return launch_igemm<64, 4, 2, 1, true>(p, st);
      case 2: return launch_igemm<64, 2, 2, 2, true>(p, st);
      default: return launch_igemm<64, 4, 2, 2, true>(p, st);
    }
  }
  return (var & 2) ? launch_igemm<32, 4, 1, 2, true>(p, st) : launch_igemm<32, 4, 1, 1, true>(p, st);
}

template <int BN, int WGM, int WGN, int PF, bool VEC, int FAST, typename TA = float, typename TD = TA>
int launch_wgrad_f(const WgradParams& p, int S, hipStream_t st) {
  constexpr size_t lds = (size_t)(2 * BK * BM + 2 * BK * BN) * sizeof(float) + (FAST ? (2 * 1024 + 4) * sizeof(int) : 0);
  static bool attr_done = false;
  if (!attr_done) {
    int rc = set_dyn_lds(igemm_wgrad_kernel<BN, WGM, WGN, PF, VEC, FAST, TA, TD>, lds);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.K, BM) * sg_cdiv(p.Cout, BN);
  hipLaunchKernelGGL((igemm_wgrad_kernel<BN, WGM, WGN, PF, VEC, FAST, TA, TD>), dim3((unsigned)tiles, 1, (unsigned)S), dim3(64 * WGM * WGN), lds, st, p);
  SG_LAUNCH_CHECK("igemm_wgrad_kernel");
  return 0;
}

template <int BN, int WGM, int WGN, int PF, bool VEC>
int launch_wgrad(const WgradParams& p, int S, hipStream_t st) {
  if constexpr (VEC && PF == 2) {
    const bool fast = p.stride == 1 && p.OH == p.H && p.OW == p.W && p.x_bytes != 0 && p.dy_bytes != 0;
    static const bool force1 = getenv("SG_WGRAD_FAST1") != nullptr;  // A/B switch
    const bool aligned = fast && !force1 && (p.OW % BK == 0) && (p.KH_KW == 1 || p.Cin % BM == 0) &&
                         ((int64_t)p.x_bytes + 2 * 32 * (int64_t)p.x_ld * 4 < (1ll << 31));
    if (aligned) return launch_wgrad_f<BN, WGM, WGN, PF, true, 2>(p, S, st);
    if (fast) return launch_wgrad_f<BN, WGM, WGN, PF, true, 1>(p, S, st);
  }
  return launch_wgrad_f<BN, WGM, WGN, PF, VEC, 0>(p, S, st);
}

inline int wgrad_bn(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

int dispatch_wgrad(const WgradParams& p_in, int S, bool vec, hipStream_t st) {
  WgradParams p = p_in;
  {
    static int noskip = -1;
    if (noskip < 0) noskip = getenv("SG_CONV_NOSKIP") ? 1 : 0;
    p.KH_KW = p.K / p.Cin;
    p.skip_slabs = (!noskip && p.dil > 1 && p.KH_KW > 1) ? 1 : 0;
    p.tap_inner = ((conv_l2() & 4) && p.KH_KW > 1 && p.Cin % BM == 0) ? 1 : 0;
  }
  const int bn = wgrad_bn(p.Cout);
  const int var = conv_variant() & 3;
  p.stagger = (conv_variant() >> 2) & 1;
  if (wgrad_x6_ok(p, vec)) {
    p.tap_inner = ((conv_l2(true) & 4) && p.KH_KW > 1 && p.Cin % BM == 0) ? 1 : 0;
    {
      static int abl = -1;
      if (abl < 0) {
        const char* e = getenv("SG_X6_ABLATE");
        abl = e ? atoi(e) & 7 : 0;
      }
      p.stagger = abl;  // the x6 wgrad kernel has no stagger; the field carries the ablation mask
    }
    static const int wpf2 = getenv("SG_X6_WGRAD_PF2") ? atoi(getenv("SG_X6_WGRAD_PF2")) : 0;  // A/B switch
    if (wpf2 && p.stride == 1) {  // (the double-buffered experiment addresses its slabs for stride 1 only)
      if (bn == 128) return launch_wgrad_x6<128, 2, 4, 2>(p, S, st);
      if (bn == 64) return launch_wgrad_x6<64, 4, 2, 2>(p, S, st);
      return launch_wgrad_x6<32, 4, 1, 2>(p, S, st);
    }
    if (x6_mode() == 2) {  // bf16 products in one pass on fp32 storage
      if (bn == 128) return launch_wgrad_x6<128, 2, 4, 1, 1, float>(p, S, st);
      if (bn == 64) return launch_wgrad_x6<64, 4, 2, 1, 1, float>(p, S, st);
      return launch_wgrad_x6<32, 4, 1, 1, 1, float>(p, S, st);
    }
    if (bn == 128) return launch_wgrad_x6<128, 2, 4, 1>(p, S, st);
    if (bn == 64) return launch_wgrad_x6<64, 4, 2, 1>(p, S, st);
    return launch_wgrad_x6<32, 4, 1, 1>(p, S, st);
  }
  if (!vec) {
    if (bn == 128) return launch_wgrad<128, 2, 4, 1, false>(p, S, st);
    if (bn == 64) return launch_wgrad<64, 4, 2, 1, false>(p, S, st);
    return launch_wgrad<32, 4, 1, 1, false>(p, S, st);
  }
  if (bn == 128) {
    switch (var) {
      case 0: return launch_wgrad<128, 2, 2, 1, true>(p, S, st);
      case 1: return launch_wgrad<128, 2, 4, 1, true>(p, S, st);
      case 2: return launch_wgrad<128, 2, 2, 2, true>(p, S, st);
      default: return launch_wgrad<128, 2, 4, 2, true>(p, S, st);
    }
  }
  if (bn == 64) {
    switch (var) {
      case 0: return launch_wgrad<64, 2, 2, 1, true>(p, S, st);
      case 1: return launch_wgrad<64, 4, 2, 1, true>(p, S, st);
      case 2: return launch_wgrad<64, 2, 2, 2, true>(p, S, st);
      default: return launch_wgrad<64, 4, 2, 2, true>(p, S, st);
    }
  }
  return (var & 2) ? launch_wgrad<32, 4, 1, 2, true>(p, S, st) : launch_wgrad<32, 4, 1, 1, true>(p, S, st);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_desc(const sg_conv_desc* d, const char* who) {
  SG_CHECK_ARG(d != nullptr, "%s: null desc", who);
  SG_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "%s: non-positive dims", who);
  SG_CHECK_ARG(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dilation > 0, "%s: bad kernel geometry", who);
  SG_CHECK_ARG(d->Ho > 0 && d->Wo > 0, "%s: bad output dims", who);
  SG_CHECK_ARG(d->pad_t >= 0 && d->pad_l >= 0, "%s: negative pad", who);
  // the gather packs a row's scaled (oh, ow) into signed 16-bit halves
  SG_CHECK_ARG(d->H < 8192 && d->W < 8192 && (int64_t)d->Ho * d->stride < 16384 && (int64_t)d->Wo * d->stride < 16384 &&
                   (int64_t)d->KH * d->dilation < 8192 && (int64_t)d->KW * d->dilation < 8192,
               "%s: map or kernel extent beyond the 16-bit coordinate range", who);
  const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
  SG_CHECK_ARG(xl >= d->Cin && yl >= d->Cout, "%s: pixel stride smaller than channel count", who);
  SG_CHECK_ARG((int64_t)d->N * d->H * d->W * xl < (1ll << 31) && (int64_t)d->N * d->Ho * d->Wo * yl < (1ll << 31),
               "%s: tensor exceeds 2^31 elements", who);
  return 0;
}


// ---- "thin" 1x1 convolutions: Cout <= 4 ----------------------------------------------------------------
// The sSE gate (C -> 1, sigmoid) and the softmax head (32 -> 2 at full resolution) are pixel-wise dot
// products: 2*Cout FLOP per 4 bytes read, bandwidth-bound by two orders of magnitude.  A 128-wide MFMA tile
// would spend 97 % of its columns on zero padding, so they get streaming kernels instead: forward = LP lanes
// per pixel, each lane one 16-byte channel chunk, sub-wave shuffle reduction; dgrad = one 16-byte store per
// thread; wgrad = the segmented column reducer with Cout outputs per channel.
template <int CO, typename TA, typename TY>
__global__ __launch_bounds__(256) void thin_fwd_kernel(const TA* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, TY* __restrict__ y, int64_t P,
                                                       int chunks, int x_ld, int y_ld, int LP, int flags, const BnIn bn) {
  constexpr int PIX = 4;  // pixels per lane group: four independent 16-byte loads in flight
  const int t = threadIdx.x;
  const int gpb = 256 / LP;  // lane groups per block
  const int li = t & (LP - 1), grp = t / LP;
  float acc[PIX][CO];
#pragma unroll
  for (int i = 0; i < PIX; ++i)
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[i][o] = 0.f;
  int64_t pix[PIX];
#pragma unroll
  for (int i = 0; i < PIX; ++i) pix[i] = ((int64_t)blockIdx.x * PIX + i) * gpb + grp;
  for (int c = li; c < chunks; c += LP) {
    float wv[4][CO];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int o = 0; o < CO; ++o) wv[k][o] = w[(4 * c + k) * CO + o];
    f32x4 xv[PIX];
#pragma unroll
    for (int i = 0; i < PIX; ++i) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      xv[i] = pix[i] < P ? ld4<TA>(x + pix[i] * x_ld + 4 * c) : z;
    }
    if (bn.mean) {   // uniform: the BatchNormalization (+ReLU) in front of this layer, applied here
      const f32x4 bm = *reinterpret_cast<const f32x4*>(bn.mean + 4 * c), bi = bn_in_inv(bn, 4 * c);
      const f32x4 bg = *reinterpret_cast<const f32x4*>(bn.gamma + 4 * c), bb = *reinterpret_cast<const f32x4*>(bn.beta + 4 * c);
#pragma unroll
      for (int i = 0; i < PIX; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) xv[i][k] = pix[i] < P ? bn_in_one(xv[i][k], bm[k], bi[k], bg[k], bb[k], bn.relu) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < PIX; ++i)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[i][o] = fmaf(xv[i][k], wv[k][o], acc[i][o]);
  }
#pragma unroll
  for (int i = 0; i < PIX; ++i)
#pragma unroll
    for (int o = 0; o < CO; ++o)
      for (int off = LP >> 1; off > 0; off >>= 1) acc[i][o] += __shfl_xor(acc[i][o], off, 64);
  if (li == 0) {
#pragma unroll
    for (int i = 0; i < PIX; ++i) {
      if (pix[i] >= P) continue;
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        float v = acc[i][o] + ((flags & SG_EPI_BIAS) ? bias[o] : 0.f);
        if (flags & SG_EPI_RELU) v = fmaxf(v, 0.f);
        st1<TY>(y + pix[i] * y_ld + o, v);
      }
    }
  }
}

template <int CO, typename TA, typename TY>
__global__ __launch_bounds__(256) void thin_dgrad_kernel(const TY* __restrict__ dy, const float* __restrict__ w,
                                                         TA* dx, int64_t total, int chunks, int y_ld, int x_ld,
                                                         const TA* res) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int64_t pix = gid / chunks;
  const int c = (int)(gid - pix * chunks);
  float g[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) g[o] = ld1<TY>(dy + pix * y_ld + o);
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float a = 0.f;
#pragma unroll
    for (int o = 0; o < CO; ++o) a = fmaf(g[o], w[(4 * c + k) * CO + o], a);
    r[k] = a;
  }
  if (res) {   // uniform: a gradient already collected for the same tensor (sg_conv2d_dgrad_acc; may be dx itself)
    const f32x4 t = ld4<TA>(res + pix * x_ld + 4 * c);
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] += t[k];
  }
  st4<TA>(dx + pix * x_ld + 4 * c, r);
}

template <int CO, typename TA, typename TY>
struct ThinWgradOp {
  static constexpr int NOUT = CO;
  const TA* __restrict__ x;
  const TY* __restrict__ dy;
  float* dw;
  int x_ld, y_ld;
  BnIn bn;   // the BatchNormalization (+ReLU) in front of the layer, applied to x here (mean == nullptr: none)
  template <int V>
  __device__ __forceinline__ void accum(int, int64_t r, int c, float (&acc)[CO][V]) const {
    float v[V];
    ldv<V>(x + r * x_ld + c, v);
    if (bn.mean) {   // uniform
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float is = bn.infer ? rsqrtf(bn.invstd[c + k] + bn.eps) : bn.invstd[c + k];
        v[k] = bn_in_one(v[k], bn.mean[c + k], is, bn.gamma[c + k], bn.beta[c + k], bn.relu);
      }
    }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      const float g = ld1<TY>(dy + r * y_ld + o);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[o][k] = fmaf(v[k], g, acc[o][k]);
    }
  }
  __device__ __forceinline__ void finalize(int, int c, const double (&s)[CO]) const {
#pragma unroll
    for (int o = 0; o < CO; ++o) dw[c * CO + o] = (float)s[o];
  }
};

inline bool thin_ok(const sg_conv_desc* d) {
  static const bool off = getenv("SG_CONV_NOTHIN") != nullptr;  // A/B switch
  const int xl = d->x_ld ? d->x_ld : d->Cin;
  return !off && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->Cout <= 4 && d->Ho == d->H && d->Wo == d->W &&
         d->Cin % 4 == 0 && xl % 4 == 0 && d->Cin >= 16;
}

template <int NOUT>
size_t thin_part_bytes_t(int num_cus, int64_t P, int Cin) { return seg_plan<NOUT>(num_cus, 1, P, Cin, true).part_bytes; }
size_t thin_part_bytes(int num_cus, const sg_conv_desc* d) {
  const int64_t P = (int64_t)d->N * d->H * d->W;
  switch (d->Cout) {
    case 1: return thin_part_bytes_t<1>(num_cus, P, d->Cin);
    case 2: return thin_part_bytes_t<2>(num_cus, P, d->Cin);
    case 3: return thin_part_bytes_t<3>(num_cus, P, d->Cin);
    default: return thin_part_bytes_t<4>(num_cus, P, d->Cin);
  }
}

inline BnIn bn_in_none() {
  BnIn b;
  b.mean = b.invstd = b.gamma = b.beta = nullptr;
  b.relu = b.infer = 0;
  b.eps = 0.f;
  return b;
}

template <int CO, typename TA, typename TY>
int thin_fwd_t(const sg_conv_desc* d, const TA* x, const float* w, const float* bias, TY* y, int flags, hipStream_t st,
               const BnIn bn = bn_in_none()) {
  const int64_t P = (int64_t)d->N * d->H * d->W;
  const int chunks = d->Cin / 4;
  int LP = 1;
  while (LP * 2 <= chunks && LP < 64) LP <<= 1;
  const int64_t groups = sg_cdiv(P, (int64_t)4 * (256 / LP));
  hipLaunchKernelGGL((thin_fwd_kernel<CO, TA, TY>), dim3((unsigned)groups), dim3(256), 0, st, x, w, bias, y, P, chunks,
                     d->x_ld ? d->x_ld : d->Cin, d->y_ld ? d->y_ld : d->Cout, LP, flags, bn);
  SG_LAUNCH_CHECK("thin_fwd_kernel");
  return 0;
}

template <int CO, typename TA, typename TY>
int thin_dgrad_t(const sg_conv_desc* d, const TY* dy, const float* w, TA* dx, hipStream_t st, const void* res = nullptr) {
  const int64_t P = (int64_t)d->N * d->H * d->W;
  const int chunks = d->Cin / 4;
  const int64_t total = P * chunks;
  hipLaunchKernelGGL((thin_dgrad_kernel<CO, TA, TY>), dim3((unsigned)sg_cdiv(total, 256)), dim3(256), 0, st, dy, w, dx, total, chunks,
                     d->y_ld ? d->y_ld : d->Cout, d->x_ld ? d->x_ld : d->Cin, (const TA*)res);
  SG_LAUNCH_CHECK("thin_dgrad_kernel");
  return 0;
}

template <int CO, typename TA, typename TY>
int thin_wgrad_t(int num_cus, const sg_conv_desc* d, const TA* x, const TY* dy, float* dw, float* part, hipStream_t st,
                 const BnIn bn = bn_in_none()) {
  const int64_t P = (int64_t)d->N * d->H * d->W;
  const SegPlan pl = seg_plan<CO>(num_cus, 1, P, d->Cin, true);
  ThinWgradOp<CO, TA, TY> op;
  op.x = x; op.dy = dy; op.dw = dw;
  op.bn = bn;
  op.x_ld = d->x_ld ? d->x_ld : d->Cin;
  op.y_ld = d->y_ld ? d->y_ld : d->Cout;
  return seg_reduce_launch(op, pl, 1, P, d->Cin, part, st, "thin_wgrad");
}

#define THIN_SWITCH(co, CALL)      \
  switch (co) {                    \
    case 1: return CALL(1);        \
    case 2: return CALL(2);        \
    case 3: return CALL(3);        \
    default: return CALL(4);       \
  }

// images per sub-batch such that both activation tensors stay under 2 GiB (>= N: no split needed); eb = bytes per element
inline int64_t images_limit_bytes() {
  // test hook: SG_CONV_MAX_BYTES lowers the 2 GiB limit so that small tensors exercise the sub-batch paths
  static int64_t lim = -1;
  if (lim < 0) {
    const char* e = getenv("SG_CONV_MAX_BYTES");
    lim = e ? atoll(e) : ((1ll << 31) - (1ll << 20));
    if (lim <= 0) lim = (1ll << 31) - (1ll << 20);
  }
  return lim;
}
inline int images_per_2gib(const sg_conv_desc* d, int eb = 4) {
  const int64_t xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
  const int64_t xi = (int64_t)d->H * d->W * xl * eb, yi = (int64_t)d->Ho * d->Wo * yl * eb;
  const int64_t per = xi > yi ? xi : yi;
  const int64_t lim = images_limit_bytes();
  if (per * d->N < lim) return d->N;
  const int64_t nb = lim / per;
  return (int)(nb < 1 ? 0 : nb);
}

// the same for a launch whose input and output differ in element size (x: ebx bytes, y: eby bytes)
inline int images_per_2gib_mixed(const sg_conv_desc* d, int ebx, int eby) {
  const int64_t xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
  const int64_t xi = (int64_t)d->H * d->W * xl * ebx, yi = (int64_t)d->Ho * d->Wo * yl * eby;
  const int64_t per = xi > yi ? xi : yi;
  const int64_t lim = images_limit_bytes();
  if (per * d->N < lim) return d->N;
  const int64_t nb = lim / per;
  return (int)(nb < 1 ? 0 : nb);
}

struct WgradPlan {
  int S;
  int slabs_per_split;
  size_t dw_part_bytes;
  size_t bias_part_bytes;
  int chunks = 1;  // > 1: activations beyond 2 GiB are reduced as sub-batches of nb whole images, each with S splits
  int nb = 0;
  int patch = 0;   // the patch form (conv_x6wp.h): S = its workgroup count, every workgroup writes one partial slab
  int wide = 0;    // the wide pointwise form (conv_pw.h): slabs_per_split counts 16-pixel k-steps
};

// Split of the pixel reduction over S workgroups per tile.  Modelled time = MFMA work / (fraction of the
// 2*CUs workgroup slots kept busy over whole waves) + the traffic of writing and re-adding S partial slabs;
// the S with the smallest modelled time wins (e.g. ASPP: 288 tiles -> S = 7: 2016 workgroups = 3.94 waves).
WgradPlan plan_wgrad(int num_cus, const sg_conv_desc* d, bool b16 = false) {
  WgradPlan pl;
  const int64_t K = (int64_t)d->KH * d->KW * d->Cin;
  const int64_t P = (int64_t)d->N * d->Ho * d->Wo;
  {
    const int nb = images_per_2gib(d);
    if (nb >= 1 && nb < d->N && !thin_ok(d)) {
      sg_conv_desc sub = *d;
      sub.N = nb;
      pl = plan_wgrad(num_cus, &sub, b16);
      pl.nb = nb;
      pl.chunks = (int)sg_cdiv(d->N, nb);
      pl.dw_part_bytes = (size_t)pl.chunks * pl.S * K * d->Cout * 4;  // every chunk writes partial slabs
      pl.bias_part_bytes = colsum_ws_bytes(num_cus, P, d->Cout);
      return pl;
    }
  }
  if (thin_ok(d)) {
    pl.S = 1;
    pl.slabs_per_split = 0;
    pl.dw_part_bytes = thin_part_bytes(num_cus, d);
    pl.bias_part_bytes = colsum_ws_bytes(num_cus, P, d->Cout);
    return pl;
  }
  if ((b16 || x6_mode() == 1) && wgrad_pw_wide_geom(d, b16 ? 2 : 4)) {   // 1x1, stride 1, wide enough: 128 x 384 tiles of dw (conv_pw.h)
    pl.wide = 1;
    wgrad_pw_wide_plan(num_cus, d, pl.S, pl.slabs_per_split, b16 ? WPB_KP : 16);
    pl.dw_part_bytes = pl.S > 1 ? (size_t)pl.S * K * d->Cout * 4 : 0;
    pl.bias_part_bytes = colsum_ws_bytes(num_cus, P, d->Cout);
    return pl;
  }
  if (x6wp_geom(d) && (b16 || x6_enabled()) && (d->x_ld ? d->x_ld : d->Cin) % (b16 ? 8 : 4) == 0 &&
      (d->y_ld ? d->y_ld : d->Cout) % (b16 ? 8 : 4) == 0) {
    pl.patch = 1;
    pl.S = x6wp_grid(num_cus, d);
    pl.slabs_per_split = (int)sg_cdiv(sg_cdiv(P, BK), pl.S);  // for the slab kernels, should the launch fall back to them
    pl.dw_part_bytes = pl.S > 1 ? (size_t)pl.S * K * d->Cout * 4 : 0;
    pl.bias_part_bytes = colsum_ws_bytes(num_cus, P, d->Cout);
    return pl;
  }
  const int bn = wgrad_bn(d->Cout);
  const int64_t tiles = sg_cdiv(K, BM) * sg_cdiv(d->Cout, bn);
  const int64_t nslab = sg_cdiv(P, BK);
  const int64_t slots = 2 * (int64_t)num_cus;
  const double flops = 2.0 * (double)tiles * BM * bn * (double)P;  // padded tile work
  // sustained rate of the kernel that will run: the x6 wgrad (geometry test as in wgrad_x6_ok) or the fp32 MFMA one
  static const double rate_x6 = getenv("SG_WGRAD_PLAN_RATE") ? atof(getenv("SG_WGRAD_PLAN_RATE")) * 1e12 : 110e12;
  const bool x6_geom = ((d->stride == 1 && d->Ho == d->H && d->Wo == d->W) || (d->stride == 2 && d->dilation == 1)) &&
                      d->Wo % 32 == 0 && d->Cout >= 16 && d->Cin % 4 == 0;
  // the one-pass bf16 kernel multiplies ~4x faster than the six-pass one: the split's partial-slab traffic weighs more
  static const double rate_b16 = getenv("SG_WGRAD_PLAN_RATE_B16") ? atof(getenv("SG_WGRAD_PLAN_RATE_B16")) * 1e12 : 110e12;  // measured 110 / 250 / 450 / 900: no gain from a higher rate (profiles/r02_b16_deep_ab.txt)
  const double rate = x6_geom ? (b16 ? rate_b16 : rate_x6) : 110e12;
  int64_t maxS = nslab / 8;  // at least 8 slabs per split
  if (maxS < 1) maxS = 1;
  if (maxS > 512) maxS = 512;
  double best_t = 1e300;
  int64_t best_S = 1;
  for (int64_t S = 1; S <= maxS; ++S) {
    const int64_t wgs = tiles * S;
    const double eff = (double)wgs / (double)(sg_cdiv(wgs, slots) * slots);
    const double part_bytes = S > 1 ? (double)S * (double)K * d->Cout * 4.0 : 0.0;
    if (part_bytes > (double)(384ll << 20)) break;
    const double t = flops / (eff * rate) + 2.0 * part_bytes / 3.0e12 + (S > 1 ? 3e-6 : 0.0);
    if (t < best_t * 0.999) { best_t = t; best_S = S; }
  }
  pl.slabs_per_split = (int)sg_cdiv(nslab, best_S);
  pl.S = (int)sg_cdiv(nslab, pl.slabs_per_split);
  pl.dw_part_bytes = pl.S > 1 ? (size_t)pl.S * K * d->Cout * 4 : 0;
  pl.bias_part_bytes = colsum_ws_bytes(num_cus, P, d->Cout);
  return pl;
}

// ---- storage-type plumbing of the entry points ---------------------------------------------------------------------
// dtype = SG_F32 or SG_BF16 (activation storage), optionally | SG_HEAD_F32: the few-channel side of a thin 1x1
// convolution (y of the forward, dy of dgrad / wgrad) is fp32 although the activations are bf16 - the softmax head,
// whose logits, probabilities and loss stay in fp32.
inline int dt_storage(int dtype) { return dtype & 0xff; }
inline bool dt_ok(int dtype) {
  const int st = dt_storage(dtype);
  return (st == SG_F32 || st == SG_BF16) && (dtype & ~(0xff | SG_HEAD_F32 | SG_X_UP2)) == 0;
}
inline int dt_bytes(int dtype) { return dt_storage(dtype) == SG_BF16 ? 2 : 4; }

void fill_fwd_params(IgemmParams& p, const sg_conv_desc* d, const void* x, const void* w, const void* bias, void* y,
                     int flags, int eb) {
  p.x = (const float*)x;
  p.w = (const float*)w;
  p.bias = (const float*)bias;
  p.y = (float*)y;
  p.H = d->H; p.W = d->W; p.C = d->Cin; p.x_ld = d->x_ld ? d->x_ld : d->Cin;
  p.OH = d->Ho; p.OW = d->Wo;
  p.Nout = d->Cout; p.y_ld = d->y_ld ? d->y_ld : d->Cout;
  p.a_mul = d->stride; p.k_mul = d->dilation; p.off_h = -d->pad_t; p.off_w = -d->pad_l; p.div = 1;
  p.K = d->KH * d->KW * d->Cin;
  p.M = d->N * d->Ho * d->Wo;
  p.flags = flags;
  p.fd_ohow = make_fastdiv((uint32_t)(d->Ho * d->Wo));
  p.fd_ow = make_fastdiv((uint32_t)d->Wo);
  p.fd_c = make_fastdiv((uint32_t)d->Cin);
  p.fd_kw = make_fastdiv((uint32_t)d->KW);
  const int64_t xb = (((int64_t)d->N * d->H * d->W - 1) * p.x_ld + d->Cin) * eb, wb = (int64_t)p.K * d->Cout * 4;
  p.x_bytes = xb < (1ll << 31) ? (uint32_t)xb : 0;
  p.w_bytes = wb < (1ll << 31) ? (uint32_t)wb : 0;
  p.stats = nullptr;
  p.perm2 = 0;
  p.res = nullptr;
  p.a_planes = nullptr;
  p.bn.mean = nullptr;
  p.bnb.x = nullptr;
  p.ws_room = 0;
}

void fill_dgrad_params(IgemmParams& p, const sg_conv_desc* d, const void* dy, const void* wt, const void* bias, void* dx,
                       int flags, int eb) {
  p.x = (const float*)dy;
  p.w = (const float*)wt;
  p.bias = (const float*)bias;
  p.y = (float*)dx;
  p.H = d->Ho; p.W = d->Wo; p.C = d->Cout; p.x_ld = d->y_ld ? d->y_ld : d->Cout;
  p.OH = d->H; p.OW = d->W;
  p.Nout = d->Cin; p.y_ld = d->x_ld ? d->x_ld : d->Cin;
  p.a_mul = 1; p.k_mul = -d->dilation; p.off_h = d->pad_t; p.off_w = d->pad_l; p.div = d->stride;
  p.K = d->KH * d->KW * d->Cout;
  p.M = d->N * d->H * d->W;
  p.flags = flags;
  p.fd_ohow = make_fastdiv((uint32_t)(d->H * d->W));
  p.fd_ow = make_fastdiv((uint32_t)d->W);
  p.fd_c = make_fastdiv((uint32_t)d->Cout);
  p.fd_kw = make_fastdiv((uint32_t)d->KW);
  const int64_t xb = (((int64_t)d->N * d->Ho * d->Wo - 1) * p.x_ld + d->Cout) * eb, wb = (int64_t)p.K * d->Cin * 4;
  p.x_bytes = xb < (1ll << 31) ? (uint32_t)xb : 0;
  p.w_bytes = wb < (1ll << 31) ? (uint32_t)wb : 0;
  p.stats = nullptr;
  p.perm2 = 0;
  p.res = nullptr;
  p.a_planes = nullptr;
  p.bn.mean = nullptr;
  p.bnb.x = nullptr;
  p.ws_room = 0;
}

// any-shape fallback for bf16 storage: the native fp32-MFMA kernel with widening loads (TA) and a rounding store (TY).
// <bf16, bf16> is the plain fallback; <bf16, float> the forward of a softmax head that is not a thin 1x1 convolution
// (Res34-UNet: Conv2D(2, 3, activation='softmax'), res34.py:156), <float, bf16> its dgrad (fp32 dy in, bf16 dx out).
template <typename TA, typename TY>
int dispatch_igemm_mixed(const IgemmParams& p_in, bool vec, int num_cus, hipStream_t st) {
  IgemmParams p = p_in;
  const int bn = pick_bn(p.M, p.Nout, num_cus);
  plan_common(p, false, bn);
  p.stagger = 0;
  p.ablate = 0;
  p.skip_taps = 0;
  p.cb = 0;
  if (vec) {
    if (bn == 128) return launch_igemm_ut<128, 2, 4, 1, true, false, TA, TY>(p, st);
    if (bn == 64) return launch_igemm_ut<64, 4, 2, 1, true, false, TA, TY>(p, st);
    return launch_igemm_ut<32, 4, 1, 1, true, false, TA, TY>(p, st);
  }
  if (bn == 128) return launch_igemm_ut<128, 2, 4, 1, false, false, TA, TY>(p, st);
  if (bn == 64) return launch_igemm_ut<64, 4, 2, 1, false, false, TA, TY>(p, st);
  return launch_igemm_ut<32, 4, 1, 1, false, false, TA, TY>(p, st);
}
int dispatch_igemm_b16(const IgemmParams& p, bool vec, int num_cus, hipStream_t st) {
  return dispatch_igemm_mixed<bf16_t, bf16_t>(p, vec, num_cus, st);
}

// x bf16, dy fp32: the kernel gradient of a (non-thin) softmax head of a bf16 model
int dispatch_wgrad_head32(const WgradParams& p_in, int S, bool vec4, hipStream_t st) {
  WgradParams p = p_in;
  p.KH_KW = p.K / p.Cin;
  p.skip_slabs = 0;
  p.tap_inner = 0;
  p.stagger = 0;
  const int bn = wgrad_bn(p.Cout);
  if (vec4) {
    if (bn == 128) return launch_wgrad_f<128, 2, 4, 1, true, 0, bf16_t, float>(p, S, st);
    if (bn == 64) return launch_wgrad_f<64, 4, 2, 1, true, 0, bf16_t, float>(p, S, st);
    return launch_wgrad_f<32, 4, 1, 1, true, 0, bf16_t, float>(p, S, st);
  }
  if (bn == 128) return launch_wgrad_f<128, 2, 4, 1, false, 0, bf16_t, float>(p, S, st);
  if (bn == 64) return launch_wgrad_f<64, 4, 2, 1, false, 0, bf16_t, float>(p, S, st);
  return launch_wgrad_f<32, 4, 1, 1, false, 0, bf16_t, float>(p, S, st);
}

int dispatch_wgrad_b16(const WgradParams& p_in, int S, bool vec8, bool vec4, hipStream_t st) {
  WgradParams p = p_in;
  p.KH_KW = p.K / p.Cin;
  static int noskip = -1;
  if (noskip < 0) noskip = getenv("SG_CONV_NOSKIP") ? 1 : 0;
  p.skip_slabs = (!noskip && p.dil > 1 && p.KH_KW > 1) ? 1 : 0;
  p.stagger = 0;
  const int bn = wgrad_bn(p.Cout);
  if (wgrad_x6_ok(p, vec8, true)) {
    p.tap_inner = ((conv_l2(true) & 4) && p.KH_KW > 1 && p.Cin % BM == 0) ? 1 : 0;
    if (bn == 128) return launch_wgrad_x6<128, 2, 4, 1, 1, bf16_t>(p, S, st);
    if (bn == 64) return launch_wgrad_x6<64, 4, 2, 1, 1, bf16_t>(p, S, st);
    return launch_wgrad_x6<32, 4, 1, 1, 1, bf16_t>(p, S, st);
  }
  p.tap_inner = 0;
  p.skip_slabs = 0;
  if (vec4) {
    if (bn == 128) return launch_wgrad_f<128, 2, 4, 1, true, 0, bf16_t>(p, S, st);
    if (bn == 64) return launch_wgrad_f<64, 4, 2, 1, true, 0, bf16_t>(p, S, st);
    return launch_wgrad_f<32, 4, 1, 1, true, 0, bf16_t>(p, S, st);
  }
  if (bn == 128) return launch_wgrad_f<128, 2, 4, 1, false, 0, bf16_t>(p, S, st);
  if (bn == 64) return launch_wgrad_f<64, 4, 2, 1, false, 0, bf16_t>(p, S, st);
  return launch_wgrad_f<32, 4, 1, 1, false, 0, bf16_t>(p, S, st);
}

}  // namespace

extern "C" {

int sg_set_conv_x6(int on) {
  const int prev = x6_mode();
  g_x6_enabled = (on < 0 || on > 2) ? 1 : on;
  return prev;
}

// scratch of a split-K launch of the 256-wide bf16 kernel for this geometry (0: none), behind the weight planes
static size_t b16w_ws_extra(const sg_conv_desc* d, bool dgrad) {
  if (check_desc(d, "b16w_ws_extra")) return 0;
  if (dgrad && d->stride != 1) return 0;
  static const float dummy[4] = {0.f, 0.f, 0.f, 0.f};
  IgemmParams p;
  if (!dgrad) fill_fwd_params(p, d, dummy, dummy, nullptr, nullptr, 0, 2);
  else fill_dgrad_params(p, d, dummy, dummy, nullptr, nullptr, 0, 2);
  p.x = (const float*)(uintptr_t)16;
  p.res = nullptr;
  if (p.C % BK != 0 && p.K != p.C) {
    const int Ckp = x6_vpad_c(p.C);
    p.K = (p.K / p.C) * Ckp;
    p.C = Ckp;
  }
  const int S = b16w_plan(p);
  size_t extra = S > 1 ? b16w_scratch_bytes(S, p.M, p.Nout) + 256 : 0;
  // the fp32 planes-in kernel (conv_x6w.h): same geometry seen as fp32 storage
  IgemmParams q;
  if (!dgrad) fill_fwd_params(q, d, dummy, dummy, nullptr, nullptr, 0, 4);
  else fill_dgrad_params(q, d, dummy, dummy, nullptr, nullptr, 0, 4);
  q.x = (const float*)(uintptr_t)16;
  q.res = nullptr;
  if (q.C % BK != 0 && q.K != q.C) {
    const int Ckp = x6_vpad_c(q.C);
    q.K = (q.K / q.C) * Ckp;
    q.C = Ckp;
  }
  const int S3 = x6w_plan(q);
  if (S3 > 0) {
    const size_t e3 = x6w_scratch_bytes(q, S3) + 256;
    if (e3 > extra) extra = e3;
  }
  return extra;
}

int sg_conv2d_up2_supported(int dtype, const sg_conv_desc* d) {
  if (!d || (dtype & 0xff) != SG_F32 || (dtype & SG_HEAD_F32)) return 0;
  if (x6_mode() != 1) return 0;   // the six-pass arithmetic of the patch kernels
  return (x6p_up2_geom(d) && x6wp_geom(d)) ? 1 : 0;
}

size_t sg_conv2d_fwd_ws_bytes(const sg_conv_desc* d) {
  if (!d) return 0;
  size_t n = x6_ws_bytes(d->KH * d->KW, d->Cin, d->Cout) + 256 + b16w_ws_extra(d, false);
  if (n < x6p_up2_ws_bytes()) n = x6p_up2_ws_bytes();
  return n;
}

int sg_conv2d_fwd(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w,
                  const void* bias, void* y, int flags) {
  return sg_conv2d_fwd_ws(ctx, stream, dtype, d, x, w, bias, y, flags, nullptr, 0);
}

int sg_conv2d_fwd_ws(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w,
                     const void* bias, void* y, int flags, void* ws, size_t ws_bytes) {
  return sg_conv2d_fwd_stats(ctx, stream, dtype, d, x, w, bias, y, flags, ws, ws_bytes, nullptr, nullptr);
}

size_t sg_conv2d_fwd_stats_bytes(const sg_conv_desc* d) {
  if (!d) return 0;
  return (size_t)sg_cdiv((int64_t)d->N * d->Ho * d->Wo, BM) * 2 * d->Cout * sizeof(float);
}

static int conv2d_fwd_impl(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w,
                           const void* bias, void* y, int flags, void* ws, size_t ws_bytes, void* stats, int* tiles_out,
                           const void* x_planes, const sg_bn_in* bn = nullptr);

static BnIn bn_in_of(const sg_bn_in* b) {
  BnIn o = bn_in_none();
  if (b && b->mean) {
    o.mean = (const float*)b->mean; o.invstd = (const float*)b->invstd; o.gamma = (const float*)b->gamma; o.beta = (const float*)b->beta;
    o.relu = b->relu; o.infer = b->infer; o.eps = b->eps;
  }
  return o;
}
// geometry of the launches that can apply a BatchNormalization to their input: the thin 1x1 kernels (Cout <= 4) and the patch
// kernels (3x3, stride 1, SAME, 32 / 64 input channels: conv_x6p.h forward, conv_x6wp.h filter gradient), fp32 storage
static bool bn_in_geom(const sg_ctx* ctx, int dtype, const sg_conv_desc* d, bool* thin_out) {
  if ((dtype & 0xff) != SG_F32 || (dtype & (SG_HEAD_F32 | SG_X_UP2))) return false;
  if ((d->x_ld && d->x_ld != d->Cin)) return false;
  const bool thin = thin_ok(d);
  if (thin_out) *thin_out = thin;
  if (thin) return true;
  if (x6_mode() != 1 || !x6p_enabled()) return false;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dilation != 1 || d->pad_t != 1 || d->pad_l != 1 || d->Ho != d->H || d->Wo != d->W) return false;
  if (!(d->Cin == 32 || d->Cin == 64) || (d->H % 8) || (d->W % 16)) return false;
  if (!(d->Cout == 32 || d->Cout == 64 || d->Cout % 128 == 0)) return false;
  return plan_wgrad(ctx->num_cus, d, false).patch != 0;   // forward AND filter gradient must both take the patch kernels
}

int sg_conv2d_bn_in_supported(const sg_ctx* ctx, int dtype, const sg_conv_desc* d) {
  return (ctx && d && check_desc(d, "sg_conv2d_bn_in_supported") == 0 && bn_in_geom(ctx, dtype, d, nullptr)) ? 1 : 0;
}

int sg_conv2d_fwd_stats_bn(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w,
                           const void* bias, void* y, int flags, void* ws, size_t ws_bytes, void* stats, int* tiles_out,
                           const sg_bn_in* bn) {
  SG_CHECK_ARG(bn && bn->mean && bn->invstd && bn->gamma && bn->beta, "sg_conv2d_fwd_stats_bn: null BatchNormalization parameters");
  SG_CHECK_ARG(ctx && d, "sg_conv2d_fwd_stats_bn: null argument");
  if (check_desc(d, "sg_conv2d_fwd_stats_bn") || !bn_in_geom(ctx, dtype, d, nullptr) || (flags & SG_PRO_UP2)) {
    sg_set_error("sg_conv2d_fwd_stats_bn: this launch takes neither a thin 1x1 nor a patch kernel (3x3 s1 SAME, 32 / 64 input channels, "
                 "fp32 storage): no kernel to apply the BatchNormalization in");
    return SG_EUNSUPPORTED;
  }
  return conv2d_fwd_impl(ctx, stream, dtype, d, x, w, bias, y, flags, ws, ws_bytes, stats, tiles_out, nullptr, bn);
}

int sg_conv2d_fwd_stats(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w,
                        const void* bias, void* y, int flags, void* ws, size_t ws_bytes, void* stats, int* tiles_out) {
  return conv2d_fwd_impl(ctx, stream, dtype, d, x, w, bias, y, flags, ws, ws_bytes, stats, tiles_out, nullptr);
}

int sg_conv2d_fwd_stats_ap(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w,
                           const void* bias, void* y, int flags, void* ws, size_t ws_bytes, void* stats, int* tiles_out,
                           const void* x_planes) {
  SG_CHECK_ARG(!x_planes || aligned16(x_planes), "sg_conv2d_fwd_stats_ap: planes must be 16-byte aligned");
  return conv2d_fwd_impl(ctx, stream, dtype, d, x, w, bias, y, flags, ws, ws_bytes, stats, tiles_out, x_planes);
}

// does the fp32 launch of this geometry read its A operand as bf16 planes (conv_x6w.h)?  Then handing it planes that exist
// already (sg_split_planes: shared by several consumers of one tensor, or kept for the filter gradient) saves its own split.
int sg_conv2d_planes_in(const sg_conv_desc* d, int dgrad) {
  if (!d || check_desc(d, "sg_conv2d_planes_in") || x6_mode() != 1) return 0;
  if (dgrad && d->stride != 1) return 0;
  if ((d->x_ld && d->x_ld != d->Cin) || (d->y_ld && d->y_ld != d->Cout)) return 0;   // planes are dense
  static const float dummy[4] = {0.f, 0.f, 0.f, 0.f};
  IgemmParams q;
  if (!dgrad) fill_fwd_params(q, d, dummy, dummy, nullptr, nullptr, 0, 4);
  else fill_dgrad_params(q, d, dummy, dummy, nullptr, nullptr, 0, 4);
  q.x = (const float*)(uintptr_t)16;
  if (q.C % BK != 0 && q.K != q.C) return 0;   // (virtually padded channels: the planes would need the padded depth)
  if (x6p_ok(q, d->KH, d->KW)) return 0;
  return x6w_plan(q) > 0 ? 1 : 0;
}

static int conv2d_fwd_impl(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w,
                           const void* bias, void* y, int flags, void* ws, size_t ws_bytes, void* stats, int* tiles_out,
                           const void* x_planes, const sg_bn_in* bn) {
  if (tiles_out) *tiles_out = 0;
  SG_CHECK_ARG(ctx != nullptr, "sg_conv2d_fwd: null ctx");
  SG_CHECK_ARG(dt_ok(dtype), "sg_conv2d_fwd: dtype %d", dtype);
  int rc = check_desc(d, "sg_conv2d_fwd");
  if (rc) return rc;
  SG_CHECK_ARG(x && w && y, "sg_conv2d_fwd: null tensor");
  SG_CHECK_ARG(!(flags & SG_EPI_BIAS) || bias, "sg_conv2d_fwd: SG_EPI_BIAS without bias");
  const bool b16 = dt_storage(dtype) == SG_BF16, head32 = (dtype & SG_HEAD_F32) != 0;
  const int eb = dt_bytes(dtype);
  SG_CHECK_ARG(!head32 || (b16 && d->Cout <= 4), "sg_conv2d_fwd: SG_HEAD_F32 needs Cout <= 4 (a softmax head) on bf16 storage");
  SG_CHECK_ARG(!(dtype & SG_X_UP2), "sg_conv2d_fwd: SG_X_UP2 is the filter gradient's flag (forward: SG_PRO_UP2)");
  hipStream_t st = (hipStream_t)stream;
  if (flags & SG_PRO_UP2) {
    // UpSampling2D(2) -> Conv2D 3x3, sub-pixel form (conv_x6p.h): x is the SOURCE [N, H/2, W/2, Cin]; the kernel sees a
    // convolution on the source grid with 4 phases x Cout columns.  (The patch kernel addresses with 64-bit pointers: no
    // 2 GiB sub-batches.)
    const int xl = d->x_ld ? d->x_ld : d->Cin;
    if (!sg_conv2d_up2_supported(dtype, d) || !aligned16(x) || (xl % 4)) {
      sg_set_error("sg_conv2d_fwd: SG_PRO_UP2 on a launch the fused up-sampling kernel does not cover (3x3 s1 SAME, 64 -> 32, "
                   "H %% 16 = 0, W %% 32 = 0, fp32 storage, x6 arithmetic)");
      return SG_EUNSUPPORTED;
    }
    if (!ws || !aligned16(ws) || ws_bytes == SG_WS_PREPARED || ws_bytes < x6p_up2_ws_bytes()) {
      sg_set_error("sg_conv2d_fwd: SG_PRO_UP2 needs a plain workspace of %zu bytes (its summed-tap planes are made per launch)", x6p_up2_ws_bytes());
      return SG_EWORKSPACE;
    }
    sg_conv_desc sd = *d;
    sd.H = d->H / 2; sd.W = d->W / 2; sd.Ho = sd.H; sd.Wo = sd.W; sd.Cout = 4 * d->Cout;
    sd.x_ld = xl; sd.y_ld = d->y_ld ? d->y_ld : d->Cout;
    IgemmParams p;
    fill_fwd_params(p, &sd, x, w, bias, y, flags & (SG_EPI_BIAS | SG_EPI_RELU), eb);
    p.K = 4 * d->Cin;
    p.y_ld = d->y_ld ? d->y_ld : d->Cout;
    if (stats && tiles_out && !(flags & SG_EPI_RELU)) {
      p.stats = (float*)stats;
      *tiles_out = (int)sg_cdiv((int64_t)d->N * d->Ho * d->Wo, BM);   // one statistics tile per phase and source tile
    }
    return run_x6p_up2(p, (const float*)w, ws, ctx->num_cus, st);
  }
  if (head32 && !(thin_ok(d) && aligned16(x))) {  // a head that is not a 1x1 convolution: the any-shape kernel, fp32 out
    // sub-batches of whole images when the bf16 input or the fp32 output passes 2 GiB (1024 x 1024 ensemble tiles)
    const int nb = images_per_2gib_mixed(d, 2, 4);
    SG_CHECK_ARG(nb >= 1, "sg_conv2d_fwd: one image of the softmax head beyond 2 GiB");
    const int64_t xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
    for (int n0 = 0; n0 < d->N; n0 += nb) {
      sg_conv_desc sub = *d;
      sub.N = (d->N - n0 < nb) ? d->N - n0 : nb;
      const char* xs = (const char*)x + (int64_t)n0 * d->H * d->W * xl * 2;
      char* ys = (char*)y + (int64_t)n0 * d->Ho * d->Wo * yl * 4;
      IgemmParams ph;
      fill_fwd_params(ph, &sub, xs, w, bias, ys, flags, eb);
      const bool vec4 = (d->Cin % 4 == 0) && (ph.x_ld % 4 == 0) && (((uintptr_t)xs & 7) == 0);
      int rch = dispatch_igemm_mixed<bf16_t, float>(ph, vec4, ctx->num_cus, st);
      if (rch) return rch;
    }
    return 0;
  }
  if (!head32) {
    // The fast kernels address their operands through 2 GiB buffer descriptors.  A larger batch is run as
    // sub-batches of whole images (independent in a forward conv), so every image takes the same kernel - and
    // therefore the same rounding - whatever batch it travels in.
    const int nb = images_per_2gib(d, eb);
    if (nb < d->N && nb >= 1) {
      const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
      for (int n0 = 0; n0 < d->N; n0 += nb) {
        sg_conv_desc sub = *d;
        sub.N = (d->N - n0 < nb) ? d->N - n0 : nb;
        const char* xs = (const char*)x + (int64_t)n0 * d->H * d->W * xl * eb;
        char* ys = (char*)y + (int64_t)n0 * d->Ho * d->Wo * yl * eb;
        g_sub_batch = true;   // the weight planes were laid out for the whole batch: no batch-size dependent kernel choice
        int rcs = conv2d_fwd_impl(ctx, stream, dtype, &sub, xs, w, bias, ys, flags, ws, ws_bytes, nullptr, nullptr, nullptr, bn);  // (no statistics)
        g_sub_batch = false;
        if (rcs) return rcs;
      }
      return 0;
    }
  }
  if (thin_ok(d) && aligned16(x)) {
    if (!b16) {
#define CALL(CO) thin_fwd_t<CO, float, float>(d, (const float*)x, (const float*)w, (const float*)bias, (float*)y, flags, st, bn_in_of(bn))
      THIN_SWITCH(d->Cout, CALL)
#undef CALL
    } else if (head32) {
#define CALL(CO) thin_fwd_t<CO, bf16_t, float>(d, (const bf16_t*)x, (const float*)w, (const float*)bias, (float*)y, flags, st)
      THIN_SWITCH(d->Cout, CALL)
#undef CALL
    } else {
#define CALL(CO) thin_fwd_t<CO, bf16_t, bf16_t>(d, (const bf16_t*)x, (const float*)w, (const float*)bias, (bf16_t*)y, flags, st)
      THIN_SWITCH(d->Cout, CALL)
#undef CALL
    }
  }
  IgemmParams p;
  fill_fwd_params(p, d, x, w, bias, y, flags, eb);
  if (!b16 && p.x_ld == d->Cin) p.a_planes = (const unsigned short*)x_planes;
  p.bn = bn_in_of(bn);
  const int ch = b16 ? 8 : 4;
  const bool vec = (d->Cin % ch == 0) && (p.x_ld % ch == 0) && (d->Cout % 4 == 0) && aligned16(x) && aligned16(w);
  if (p.bn.mean && !(x6_ok(p, vec, b16) && x6p_ok(p, d->KH, d->KW) && ws && aligned16(ws))) {   // only the patch kernel applies it
    sg_set_error("sg_conv2d_fwd_stats_bn: the launch does not take the patch kernel after all (alignment / workspace)");
    return SG_EUNSUPPORTED;
  }
  const bool vpad_safe = (p.C % BK == 0) || (p.K == p.C) || (p.x_ld == p.C);  // padded reads must stay inside this tensor
  const bool prepared = ws_bytes == SG_WS_PREPARED;
  const bool have_ws = ws && aligned16(ws) && (prepared || ws_bytes >= x6_ws_bytes(d->KH * d->KW, d->Cin, d->Cout));
  if (prepared && !(have_ws && vpad_safe && x6_ok(p, vec, b16))) {
    sg_set_error("sg_conv2d_fwd: SG_WS_PREPARED planes given, but this launch does not take a prepared-planes kernel");
    return SG_EINVAL;
  }
  if (have_ws && vpad_safe && x6_ok(p, vec, b16)) {
    if (stats && tiles_out && !(flags & SG_EPI_RELU)) {  // the statistics ride in the x6 kernel's epilogue only
      p.stats = (float*)stats;
      *tiles_out = (int)sg_cdiv(p.M, BM);
    }
    p.ws_room = prepared ? SIZE_MAX : ws_bytes;
    if (b16) return run_x6<1, bf16_t>(p, (const float*)w, false, d->Cin, d->Cout, d->KH, d->KW, ws, ctx->num_cus, st, prepared);
    if (x6_mode() == 2) return run_x6<1, float>(p, (const float*)w, false, d->Cin, d->Cout, d->KH, d->KW, ws, ctx->num_cus, st, prepared);
    return run_x6<3, float>(p, (const float*)w, false, d->Cin, d->Cout, d->KH, d->KW, ws, ctx->num_cus, st, prepared);
  }
  if (b16) {
    const bool vec4 = (d->Cin % 4 == 0) && (p.x_ld % 4 == 0) && (((uintptr_t)x & 7) == 0);
    return dispatch_igemm_b16(p, vec4, ctx->num_cus, st);
  }
  return dispatch_igemm(p, vec, ctx->num_cus, st);
}

int sg_get_conv_x6(void) { return x6_mode(); }

int sg_conv2d_planes_job(const sg_ctx* ctx, int dtype, const sg_conv_desc* d, int dgrad, sg_planes_job* out, size_t* bytes) {
  SG_CHECK_ARG(ctx && d && out && bytes && dt_ok(dtype), "sg_conv2d_planes_job: bad argument");
  memset(out, 0, sizeof(*out));
  *bytes = 0;
  int rc = check_desc(d, "sg_conv2d_planes_job");
  if (rc) return rc;
  if ((dtype & SG_HEAD_F32) || thin_ok(d)) return 0;  // kind 0: streaming kernels, no planes
  if (dgrad && d->stride != 1 && d->stride != 2) return 0;
  const bool b16 = dt_storage(dtype) == SG_BF16;
  const int eb = dt_bytes(dtype);
  IgemmParams p;
  // geometry only: the pointers are assumed 16-byte aligned and the operands dense, as the host's arenas and torch's
  // allocator make them; a launch for which that does not hold refuses the prepared planes (SG_EINVAL)
  static const float dummy[4] = {0.f, 0.f, 0.f, 0.f};
  if (!dgrad) fill_fwd_params(p, d, dummy, dummy, nullptr, nullptr, 0, eb);
  else fill_dgrad_params(p, d, dummy, dummy, nullptr, nullptr, 0, eb);
  const int ch = b16 ? 8 : 4;
  const bool vec = !dgrad ? ((d->Cin % ch == 0) && (p.x_ld % ch == 0) && (d->Cout % 4 == 0))
                          : ((d->Cout % ch == 0) && (p.x_ld % ch == 0) && (d->Cin % 4 == 0));
  const bool vpad_safe = (p.C % BK == 0) || (p.K == p.C) || (p.x_ld == p.C);
  const int nb = images_per_2gib(d, eb);
  if (nb < 1) return 0;
  if (nb < d->N) {  // the launch will run as sub-batches: describe one of them
    const int64_t pix = !dgrad ? (int64_t)nb * d->H * d->W : (int64_t)nb * d->Ho * d->Wo;
    const int64_t xb = ((pix - 1) * p.x_ld + p.C) * eb;
    p.x_bytes = xb < (1ll << 31) ? (uint32_t)xb : 0;
  }
  if (!(vpad_safe && x6_ok(p, vec, b16))) return 0;
  const int npl = (b16 || x6_mode() == 2) ? 1 : 3;
  const int Ck = p.C;
  out->s_tap = d->Cin * d->Cout;
  out->s_k = dgrad ? 1 : d->Cout;
  out->s_n = dgrad ? d->Cout : 1;
  out->N = p.Nout;
  out->Ck = Ck;
  out->kd = 0;
  if (npl == 3) {
    p.x = (const float*)(uintptr_t)16;  // x6p_ok tests the alignment of x
    if (x6p_ok(p, d->KH, d->KW)) {
      out->kind = 2;
      out->npl = 3;
      out->K = p.K;
      out->Ckp = Ck;
      out->Kpad = p.K;
      out->Npad = p.Nout;
      const int64_t threads = (int64_t)(p.K / 16) * (p.Nout / 32) * 64;
      out->nblocks = (int32_t)sg_cdiv(threads, 256);
      *bytes = (size_t)3 * p.K * p.Nout * 2;
      return 0;
    }
  }
  p.x = (const float*)(uintptr_t)16;
  const int wbn = nb >= d->N ? pw_wide_bn(p, eb) : 0;
  if (wbn) {   // the wide pointwise kernel's k-block-major planes (conv_pw.h); never for sub-batches
    out->kind = 3;
    out->npl = npl;
    out->K = p.K;
    out->Ckp = p.K;
    out->kd = pw_kd(npl);
    out->Kpad = pw_kpad(p.K, npl);
    out->Npad = pw_npad(p.Nout, wbn);
    out->nblocks = (int32_t)(sg_cdiv(out->Kpad, 32) * sg_cdiv(out->Npad, 32));
    *bytes = pw_planes_bytes(p.K, p.Nout, npl, wbn);
    return 0;
  }
  int Ckp = Ck, K = p.K;
  if (Ck % BK != 0 && K != Ck) {
    Ckp = x6_vpad_c(Ck);
    K = (K / Ck) * Ckp;
  }
  out->kind = 1;
  out->npl = npl;
  out->K = K;
  out->Ckp = Ckp;
  out->kd = x6_plane_kd(b16 && npl == 1, Ckp, K == Ckp);
  out->Kpad = x6_kpad(K, out->kd);
  out->Npad = x6_npad(p.Nout);
  out->nblocks = (out->Kpad / 32) * (out->Npad / 32);
  *bytes = x6_planes_bytes(K, p.Nout, npl, out->kd);
  if (!b16 && npl == 3 && out->kd == XW_KD) {   // conv_x6w.h: the activation's planes and split-K partial slabs behind the weight planes
    IgemmParams q = p;
    q.C = Ckp;
    q.K = K;
    q.res = nullptr;
    const int S = x6w_plan(q);
    if (S > 0) *bytes = ((*bytes + 255) & ~(size_t)255) + x6w_scratch_bytes(q, S);
  }
  if (b16 && npl == 1 && out->kd == BW_KD) {   // a split-K launch of the 256-wide kernel keeps its partial slabs behind the planes
    IgemmParams q = p;
    q.C = Ckp;
    q.K = K;
    q.res = nullptr;
    const int S = b16w_plan(q);
    if (S > 1) *bytes = ((*bytes + 255) & ~(size_t)255) + b16w_scratch_bytes(S, q.M, q.Nout);
  }
  return 0;
}

int sg_prepare_planes(sg_ctx* ctx, void* stream, const void* w_arena, void* planes_arena, const sg_planes_job* jobs_dev,
                      int njobs, int total_blocks) {
  SG_CHECK_ARG(ctx && w_arena && planes_arena && jobs_dev && njobs > 0 && total_blocks > 0, "sg_prepare_planes: bad argument");
  SG_CHECK_ARG(aligned16(planes_arena), "sg_prepare_planes: planes arena must be 16-byte aligned");
  hipLaunchKernelGGL(prepare_planes_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const float*)w_arena,
                     (char*)planes_arena, jobs_dev, njobs);
  SG_LAUNCH_CHECK("prepare_planes_kernel");
  return 0;
}

size_t sg_bn_tiles_ws_bytes(const sg_ctx* ctx, int tiles, int C) {
  if (!ctx) return 0;
  return seg_plan<2>(ctx->num_cus, 1, tiles, C, true).part_bytes + seg_plan<2>(ctx->num_cus, 1, tiles, C, false).part_bytes;
}

int sg_bn_train_fwd_tiles(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, const void* stats, int tiles,
                          void* moving_mean, void* moving_var, void* save_mean, void* save_invstd, float momentum,
                          float eps, int unbiased_update, void* ws, size_t ws_bytes) {
  // dtype names the storage of the activations the statistics belong to; the statistics themselves are fp32
  SG_CHECK_ARG(ctx && (dtype == SG_F32 || dtype == SG_BF16), "sg_bn_train_fwd_tiles: bad ctx/dtype");
  SG_CHECK_ARG(rows > 0 && C > 0 && stats && tiles == (int)sg_cdiv(rows, BM) && moving_mean && moving_var && save_mean && save_invstd,
               "sg_bn_train_fwd_tiles: bad argument");
  const bool vec = (C % 4 == 0) && sg_aligned16(stats);
  const SegPlan pl = seg_plan<2>(ctx->num_cus, 1, tiles, C, vec);
  if (!ws || ws_bytes < pl.part_bytes) {
    sg_set_error("sg_bn_train_fwd_tiles: workspace %zu < %zu", ws_bytes, pl.part_bytes);
    return SG_EWORKSPACE;
  }
  BnTilesOp op;
  op.stats = (const float*)stats; op.C = C; op.rows = rows;
  op.moving_mean = (float*)moving_mean; op.moving_var = (float*)moving_var;
  op.save_mean = (float*)save_mean; op.save_invstd = (float*)save_invstd;
  op.momentum = momentum; op.eps = eps; op.unbiased = unbiased_update;
  return seg_reduce_launch(op, pl, 1, tiles, C, (float*)ws, (hipStream_t)stream, "bn_tiles");
}

size_t sg_conv2d_dgrad_ws_bytes(const sg_conv_desc* d) {
  if (!d) return 0;
  const size_t native = (size_t)d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
  const size_t x6 = x6_ws_bytes(d->KH * d->KW, d->Cout, d->Cin) + 256 + b16w_ws_extra(d, true);
  return native > x6 ? native : x6;
}

static int conv2d_dgrad_impl(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                            const void* bias, void* dx, int flags, void* ws, size_t ws_bytes, const void* res,
                            const void* dy_planes = nullptr, const sg_bn_bwd_in* bnb = nullptr);

// the input gradient of a pointwise convolution with the BatchNormalization backward apply in its A path (conv_pw.h, BNB form)
static bool dgrad_bnb_geom(const sg_ctx* ctx, int dtype, const sg_conv_desc* d) {
  if ((dtype & 0xff) != SG_F32 || (dtype & (SG_HEAD_F32 | SG_X_UP2)) || x6_mode() != 1) return false;
  if (d->KH != 1 || d->KW != 1 || d->stride != 1 || (d->x_ld && d->x_ld != d->Cin) || (d->y_ld && d->y_ld != d->Cout)) return false;
  if (images_per_2gib(d, 4) < d->N || d->Cout + 16 > PW_BNB_MAXK) return false;
  static const int var = getenv("SG_PW_VAR") ? atoi(getenv("SG_PW_VAR")) : 1;
  if (var != 1 || getenv("SG_PW_ABLATE")) return false;
  static const float dummy[4] = {0.f, 0.f, 0.f, 0.f};
  IgemmParams q;
  fill_dgrad_params(q, d, dummy, dummy, nullptr, nullptr, 0, 4);
  q.x = (const float*)(uintptr_t)16;
  const int wbn = pw_wide_bn(q, 4);
  return wbn == 384 || wbn == 256;
}

int sg_conv2d_dgrad_bnb_supported(const sg_ctx* ctx, int dtype, const sg_conv_desc* d) {
  return (ctx && d && check_desc(d, "sg_conv2d_dgrad_bnb_supported") == 0 && dgrad_bnb_geom(ctx, dtype, d)) ? 1 : 0;
}

int sg_conv2d_dgrad_bnb(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w, void* dx,
                        void* ws, size_t ws_bytes, const sg_bn_bwd_in* bnb) {
  SG_CHECK_ARG(ctx && d && bnb, "sg_conv2d_dgrad_bnb: null argument");
  SG_CHECK_ARG(bnb->x && bnb->mean && bnb->invstd && bnb->gamma && bnb->dgamma && bnb->dbeta && bnb->dz && bnb->rows > 0,
               "sg_conv2d_dgrad_bnb: null BatchNormalization operand");
  SG_CHECK_ARG(!bnb->relu || bnb->beta, "sg_conv2d_dgrad_bnb: a fused ReLU needs beta (the mask is recomputed from x)");
  if (check_desc(d, "sg_conv2d_dgrad_bnb") || !dgrad_bnb_geom(ctx, dtype, d) || !aligned16(bnb->x) || !aligned16(bnb->dz)) {
    sg_set_error("sg_conv2d_dgrad_bnb: not a launch of the wide pointwise kernel (1x1, stride 1, dense fp32 operands, >= 6144 rows, "
                 "x6 arithmetic, default schedule)");
    return SG_EUNSUPPORTED;
  }
  return conv2d_dgrad_impl(ctx, stream, dtype, d, dy, w, nullptr, dx, 0, ws, ws_bytes, nullptr, nullptr, bnb);
}

int sg_conv2d_dgrad(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                    const void* bias, void* dx, int flags, void* ws, size_t ws_bytes) {
  return conv2d_dgrad_impl(ctx, stream, dtype, d, dy, w, bias, dx, flags, ws, ws_bytes, nullptr);
}

int sg_conv2d_dgrad_ap(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                       const void* bias, void* dx, int flags, void* ws, size_t ws_bytes, const void* dy_planes) {
  SG_CHECK_ARG(!dy_planes || aligned16(dy_planes), "sg_conv2d_dgrad_ap: planes must be 16-byte aligned");
  return conv2d_dgrad_impl(ctx, stream, dtype, d, dy, w, bias, dx, flags, ws, ws_bytes, nullptr, dy_planes);
}

int sg_conv2d_dgrad_acc(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                        const void* bias, void* dx, int flags, void* ws, size_t ws_bytes, const void* res) {
  SG_CHECK_ARG(res != nullptr, "sg_conv2d_dgrad_acc: null res");
  return conv2d_dgrad_impl(ctx, stream, dtype, d, dy, w, bias, dx, flags, ws, ws_bytes, res);
}

static int conv2d_dgrad_impl(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                            const void* bias, void* dx, int flags, void* ws, size_t ws_bytes, const void* res, const void* dy_planes,
                            const sg_bn_bwd_in* bnb) {
  SG_CHECK_ARG(ctx != nullptr, "sg_conv2d_dgrad: null ctx");
  SG_CHECK_ARG(dt_ok(dtype), "sg_conv2d_dgrad: dtype %d", dtype);
  int rc = check_desc(d, "sg_conv2d_dgrad");
  if (rc) return rc;
  SG_CHECK_ARG(dy && w && dx, "sg_conv2d_dgrad: null tensor");
  SG_CHECK_ARG(!(flags & SG_EPI_BIAS) || bias, "sg_conv2d_dgrad: SG_EPI_BIAS without bias");
  if (d->stride != 1 && d->stride != 2) {
    sg_set_error("sg_conv2d_dgrad: stride %d unsupported (the path uses strides 1 and 2 only)", d->stride);
    return SG_EUNSUPPORTED;
  }
  const size_t need = sg_conv2d_dgrad_ws_bytes(d);
  const bool prepared = ws_bytes == SG_WS_PREPARED;
  if (!ws || (!prepared && ws_bytes < need)) {
    sg_set_error("sg_conv2d_dgrad: workspace %zu < %zu", ws_bytes, need);
    return SG_EWORKSPACE;
  }
  SG_CHECK_ARG(aligned16(ws), "sg_conv2d_dgrad: workspace must be 16-byte aligned");
  const bool b16 = dt_storage(dtype) == SG_BF16, head32 = (dtype & SG_HEAD_F32) != 0;
  const int eb = dt_bytes(dtype);
  const bool thin = thin_ok(d) && aligned16(dx) && !(flags & (SG_EPI_BIAS | SG_EPI_RELU));
  SG_CHECK_ARG(!head32 || (b16 && d->Cout <= 4), "sg_conv2d_dgrad: SG_HEAD_F32 needs Cout <= 4 (a softmax head) on bf16 storage");
  hipStream_t st = (hipStream_t)stream;
  if (res && head32) {
    sg_set_error("sg_conv2d_dgrad_acc: softmax-head launches do not add a collected gradient");
    return SG_EUNSUPPORTED;
  }
  SG_CHECK_ARG(!(dtype & SG_X_UP2) && !(flags & SG_PRO_UP2), "sg_conv2d_dgrad: the up-sampling flag of the input gradient is SG_EPI_DOWN2");
  const bool down2 = (flags & SG_EPI_DOWN2) != 0;
  if (down2) {  // dx = the SOURCE's gradient [N, H/2, W/2, Cin]: the patch kernel's epilogue adds the 2 x 2 cells (conv_x6p.h)
    const int yl_ = d->y_ld ? d->y_ld : d->Cout;
    if (!sg_conv2d_up2_supported(dtype, d) || res || (flags & (SG_EPI_BIAS | SG_EPI_RELU)) || !aligned16(dy) || (yl_ % 4)) {
      sg_set_error("sg_conv2d_dgrad: SG_EPI_DOWN2 on a launch the fused up-sampling kernel does not cover (3x3 s1 SAME, 64 -> 32, "
                   "H %% 16 = 0, W %% 32 = 0, fp32 storage, x6 arithmetic, no bias / ReLU / collected gradient)");
      return SG_EUNSUPPORTED;
    }
  }
  if (head32 && !thin) {  // fp32 dy in, bf16 dx out, any shape
    const int nb = images_per_2gib_mixed(d, 2, 4);  // dx bf16 (the forward's x), dy fp32
    SG_CHECK_ARG(nb >= 1, "sg_conv2d_dgrad: one image of the softmax head beyond 2 GiB");
    float* wth = (float*)ws;
    dim3 grid((unsigned)sg_cdiv(d->Cout, 32), (unsigned)sg_cdiv(d->Cin, 32), (unsigned)(d->KH * d->KW));
    hipLaunchKernelGGL(transpose_taps_kernel, grid, dim3(256), 0, st, (const float*)w, wth, d->Cin, d->Cout);
    SG_LAUNCH_CHECK("transpose_taps_kernel");
    const int64_t xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
    for (int n0 = 0; n0 < d->N; n0 += nb) {
      sg_conv_desc sub = *d;
      sub.N = (d->N - n0 < nb) ? d->N - n0 : nb;
      const char* dys = (const char*)dy + (int64_t)n0 * d->Ho * d->Wo * yl * 4;
      char* dxs = (char*)dx + (int64_t)n0 * d->H * d->W * xl * 2;
      IgemmParams ph;
      fill_dgrad_params(ph, &sub, dys, wth, bias, dxs, flags, 4);
      const bool vec4 = (d->Cout % 4 == 0) && (ph.x_ld % 4 == 0) && aligned16(dys);
      int rch = dispatch_igemm_mixed<float, bf16_t>(ph, vec4, ctx->num_cus, st);
      if (rch) return rch;
    }
    return 0;
  }
  if (!head32) {
    const int nb = images_per_2gib(d, eb);  // see sg_conv2d_fwd_ws
    if (nb < d->N && nb >= 1) {
      const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
      for (int n0 = 0; n0 < d->N; n0 += nb) {
        sg_conv_desc sub = *d;
        sub.N = (d->N - n0 < nb) ? d->N - n0 : nb;
        const char* dys = (const char*)dy + (int64_t)n0 * d->Ho * d->Wo * yl * eb;
        char* dxs = (char*)dx + (int64_t)n0 * (down2 ? (d->H / 2) * (d->W / 2) : d->H * d->W) * xl * eb;
        const char* ress = res ? (const char*)res + (int64_t)n0 * d->H * d->W * xl * eb : nullptr;
        g_sub_batch = true;
        int rcs = conv2d_dgrad_impl(ctx, stream, dtype, &sub, dys, w, bias, dxs, flags, ws, ws_bytes, ress);
        g_sub_batch = false;
        if (rcs) return rcs;
      }
      return 0;
    }
  }
  if (thin) {
    if (!b16) {
#define CALL(CO) thin_dgrad_t<CO, float, float>(d, (const float*)dy, (const float*)w, (float*)dx, st, res)
      THIN_SWITCH(d->Cout, CALL)
#undef CALL
    } else if (head32) {
#define CALL(CO) thin_dgrad_t<CO, bf16_t, float>(d, (const float*)dy, (const float*)w, (bf16_t*)dx, st)
      THIN_SWITCH(d->Cout, CALL)
#undef CALL
    } else {
#define CALL(CO) thin_dgrad_t<CO, bf16_t, bf16_t>(d, (const bf16_t*)dy, (const float*)w, (bf16_t*)dx, st, res)
      THIN_SWITCH(d->Cout, CALL)
#undef CALL
    }
  }
  float* wt = (float*)ws;
  IgemmParams p;
  fill_dgrad_params(p, d, dy, wt, bias, dx, flags, eb);
  p.res = (const float*)res;
  if (!b16 && p.x_ld == d->Cout) p.a_planes = (const unsigned short*)dy_planes;
  if (bnb) {
    p.bnb.x = (const float*)bnb->x; p.bnb.mean = (const float*)bnb->mean; p.bnb.invstd = (const float*)bnb->invstd;
    p.bnb.gamma = (const float*)bnb->gamma; p.bnb.beta = (const float*)bnb->beta; p.bnb.dgamma = (const float*)bnb->dgamma;
    p.bnb.dbeta = (const float*)bnb->dbeta; p.bnb.dz = (float*)bnb->dz; p.bnb.relu = bnb->relu ? 1 : 0;
    {
      static const int abl = getenv("SG_BNB_ABLATE") ? atoi(getenv("SG_BNB_ABLATE")) : 0;   // timing only: 2 = no dz store, 4 = no x load
      p.bnb.relu |= abl & 6;
    }
    p.bnb.inv_n = 1.0f / (float)bnb->rows;
  }
  const int ch = b16 ? 8 : 4;
  const bool vec = (d->Cout % ch == 0) && (p.x_ld % ch == 0) && (d->Cin % 4 == 0) && aligned16(dy);
  const bool vpad_safe = (p.C % BK == 0) || (p.K == p.C) || (p.x_ld == p.C);
  if (down2 && !(vpad_safe && x6_ok(p, vec, b16) && x6p_ok(p, d->KH, d->KW))) {  // only the patch kernel knows the flag
    sg_set_error("sg_conv2d_dgrad: SG_EPI_DOWN2, but this launch does not take the patch kernel");
    return SG_EUNSUPPORTED;
  }
  if (vpad_safe && x6_ok(p, vec, b16)) {
    static const int perm_on = getenv("SG_DGRAD_PERM2") ? atoi(getenv("SG_DGRAD_PERM2")) : 1;
    if (perm_on && d->stride == 2 && d->dilation == 1 && d->H % 2 == 0 && d->W % 2 == 0 && d->KH * d->KW <= 64) {
      p.perm2 = 1;  // rows in parity-class order: see IgemmParams::perm2
      p.fd_mc = make_fastdiv((uint32_t)(p.M / 4));
      p.fd_hcwc = make_fastdiv((uint32_t)((d->H / 2) * (d->W / 2)));
      p.fd_wc = make_fastdiv((uint32_t)(d->W / 2));
    }
    p.ws_room = prepared ? SIZE_MAX : ws_bytes;
    if (b16) return run_x6<1, bf16_t>(p, (const float*)w, true, d->Cin, d->Cout, d->KH, d->KW, ws, ctx->num_cus, st, prepared);
    if (x6_mode() == 2) return run_x6<1, float>(p, (const float*)w, true, d->Cin, d->Cout, d->KH, d->KW, ws, ctx->num_cus, st, prepared);
    return run_x6<3, float>(p, (const float*)w, true, d->Cin, d->Cout, d->KH, d->KW, ws, ctx->num_cus, st, prepared);
  }
  if (prepared) {
    sg_set_error("sg_conv2d_dgrad: SG_WS_PREPARED planes given, but this launch does not take a prepared-planes kernel");
    return SG_EINVAL;
  }
  if (res) {
    sg_set_error("sg_conv2d_dgrad_acc: this launch takes the fp32-MFMA kernels, which do not add a collected gradient");
    return SG_EUNSUPPORTED;
  }
  {
    dim3 grid((unsigned)sg_cdiv(d->Cout, 32), (unsigned)sg_cdiv(d->Cin, 32), (unsigned)(d->KH * d->KW));
    hipLaunchKernelGGL(transpose_taps_kernel, grid, dim3(256), 0, st, (const float*)w, wt, d->Cin, d->Cout);
    SG_LAUNCH_CHECK("transpose_taps_kernel");
  }
  if (b16) {
    const bool vec4 = (d->Cout % 4 == 0) && (p.x_ld % 4 == 0) && (((uintptr_t)dy & 7) == 0);
    return dispatch_igemm_b16(p, vec4, ctx->num_cus, st);
  }
  return dispatch_igemm(p, (d->Cout % 4 == 0) && (p.x_ld % 4 == 0) && (d->Cin % 4 == 0) && aligned16(dy), ctx->num_cus, st);
}

size_t sg_conv2d_wgrad_ws_bytes(const sg_ctx* ctx, const sg_conv_desc* d) {
  if (!ctx || !d) return 0;
  // the query does not know the storage type: the larger of the fp32 and the bf16 plan
  const WgradPlan pl = plan_wgrad(ctx->num_cus, d, false), pb = plan_wgrad(ctx->num_cus, d, true);
  const size_t a = pl.dw_part_bytes + pl.bias_part_bytes, b = pb.dw_part_bytes + pb.bias_part_bytes;
  return (a > b ? a : b) + 512;
}

static int conv2d_wgrad_impl(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* dy,
                             void* dw, void* dbias, void* ws, size_t ws_bytes, const sg_bn_in* bn);

int sg_conv2d_wgrad(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* dy,
                    void* dw, void* dbias, void* ws, size_t ws_bytes) {
  return conv2d_wgrad_impl(ctx, stream, dtype, d, x, dy, dw, dbias, ws, ws_bytes, nullptr);
}

int sg_conv2d_wgrad_bn(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* dy,
                       void* dw, void* dbias, void* ws, size_t ws_bytes, const sg_bn_in* bn) {
  SG_CHECK_ARG(bn && bn->mean && bn->invstd && bn->gamma && bn->beta, "sg_conv2d_wgrad_bn: null BatchNormalization parameters");
  SG_CHECK_ARG(ctx && d, "sg_conv2d_wgrad_bn: null argument");
  if (check_desc(d, "sg_conv2d_wgrad_bn") || !bn_in_geom(ctx, dtype, d, nullptr)) {
    sg_set_error("sg_conv2d_wgrad_bn: this launch takes neither the thin 1x1 nor the patch filter-gradient kernel");
    return SG_EUNSUPPORTED;
  }
  return conv2d_wgrad_impl(ctx, stream, dtype, d, x, dy, dw, dbias, ws, ws_bytes, bn);
}

static int conv2d_wgrad_impl(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* dy,
                             void* dw, void* dbias, void* ws, size_t ws_bytes, const sg_bn_in* bn) {
  SG_CHECK_ARG(ctx != nullptr, "sg_conv2d_wgrad: null ctx");
  SG_CHECK_ARG(dt_ok(dtype), "sg_conv2d_wgrad: dtype %d", dtype);
  int rc = check_desc(d, "sg_conv2d_wgrad");
  if (rc) return rc;
  SG_CHECK_ARG(x && dy && dw, "sg_conv2d_wgrad: null tensor");
  const bool b16 = dt_storage(dtype) == SG_BF16, head32 = (dtype & SG_HEAD_F32) != 0, up2 = (dtype & SG_X_UP2) != 0;
  const int eb = dt_bytes(dtype);
  const WgradPlan pl = plan_wgrad(ctx->num_cus, d, b16);
  if (up2 && !(pl.patch && !head32 && x6p_up2_geom(d) && aligned16(x) && aligned16(dy))) {
    // x = the source of a 2x nearest up-sampling: only the patch kernel gathers that way (conv_x6wp.h)
    sg_set_error("sg_conv2d_wgrad: SG_X_UP2 on a launch the patch filter-gradient kernel does not cover");
    return SG_EUNSUPPORTED;
  }
  const size_t need = pl.dw_part_bytes + pl.bias_part_bytes + 512;
  if (!ws || ws_bytes < need) {
    sg_set_error("sg_conv2d_wgrad: workspace %zu < %zu", ws_bytes, need);
    return SG_EWORKSPACE;
  }
  SG_CHECK_ARG(aligned16(ws), "sg_conv2d_wgrad: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const bool thin = thin_ok(d) && aligned16(x);
  SG_CHECK_ARG(!head32 || (b16 && d->Cout <= 4), "sg_conv2d_wgrad: SG_HEAD_F32 needs Cout <= 4 (a softmax head) on bf16 storage");
  const int64_t rows_y = (int64_t)d->N * d->Ho * d->Wo;
  const int yl_ = d->y_ld ? d->y_ld : d->Cout;
  float* bias_part = (float*)((char*)ws + ((pl.dw_part_bytes + 255) & ~(size_t)255));  // 256-byte aligned, after the dw partials
  auto bias_grad = [&]() -> int {
    if (!dbias) return 0;
    if (!b16 || head32) return launch_colsum<float>(ctx->num_cus, (const float*)dy, rows_y, d->Cout, yl_, (float*)dbias, bias_part, st);
    return launch_colsum<bf16_t>(ctx->num_cus, (const bf16_t*)dy, rows_y, d->Cout, yl_, (float*)dbias, bias_part, st);
  };
  if (thin) {
    auto run = [&]() -> int {
      if (!b16) {
#define CALL(CO) thin_wgrad_t<CO, float, float>(ctx->num_cus, d, (const float*)x, (const float*)dy, (float*)dw, (float*)ws, st, bn_in_of(bn))
        THIN_SWITCH(d->Cout, CALL)
#undef CALL
      } else if (head32) {
#define CALL(CO) thin_wgrad_t<CO, bf16_t, float>(ctx->num_cus, d, (const bf16_t*)x, (const float*)dy, (float*)dw, (float*)ws, st)
        THIN_SWITCH(d->Cout, CALL)
#undef CALL
      } else {
#define CALL(CO) thin_wgrad_t<CO, bf16_t, bf16_t>(ctx->num_cus, d, (const bf16_t*)x, (const bf16_t*)dy, (float*)dw, (float*)ws, st)
        THIN_SWITCH(d->Cout, CALL)
#undef CALL
      }
    };
    rc = run();
    if (rc) return rc;
    return bias_grad();
  }
  const int K_all = d->KH * d->KW * d->Cin;
  // parts: in = the plan's S, out = the partial slabs actually written (a patch plan that has to fall back to the slab
  // kernels - unaligned pointers - writes cdiv(slabs, slabs_per_split) <= S of them)
  auto launch_part = [&](const sg_conv_desc& dd, const void* xs, const void* dys, float* out, int& parts, int sps) -> int {
    int S = parts;
    WgradParams p;
    p.x = (const float*)xs;
    p.dy = (const float*)dys;
    p.out = out;
    p.H = dd.H; p.W = dd.W; p.Cin = dd.Cin; p.x_ld = dd.x_ld ? dd.x_ld : dd.Cin;
    p.OH = dd.Ho; p.OW = dd.Wo; p.Cout = dd.Cout; p.y_ld = dd.y_ld ? dd.y_ld : dd.Cout;
    p.stride = dd.stride; p.dil = dd.dilation; p.pad_t = dd.pad_t; p.pad_l = dd.pad_l;
    p.K = K_all;
    p.P = dd.N * dd.Ho * dd.Wo;
    p.slabs_per_split = sps;
    p.up = up2 ? 1 : 0;
    p.bn = bn_in_of(bn);
    p.x_plane_bytes = p.dy_plane_bytes = 0;
    p.fd_ohow = make_fastdiv((uint32_t)(dd.Ho * dd.Wo));
    p.fd_ow = make_fastdiv((uint32_t)dd.Wo);
    p.fd_c = make_fastdiv((uint32_t)dd.Cin);
    p.fd_kw = make_fastdiv((uint32_t)dd.KW);
    p.fd_oh = make_fastdiv((uint32_t)dd.Ho);
    {
      const int64_t xb = (((int64_t)dd.N * (dd.H >> p.up) * (dd.W >> p.up) - 1) * p.x_ld + dd.Cin) * eb;
      const int64_t yb = (((int64_t)p.P - 1) * p.y_ld + dd.Cout) * eb;
      p.x_bytes = xb < (1ll << 31) ? (uint32_t)xb : 0;
      p.dy_bytes = yb < (1ll << 31) ? (uint32_t)yb : 0;
    }
    if (pl.wide) {
      const bool al = aligned16(xs) && aligned16(dys) && p.x_bytes != 0 && p.dy_bytes != 0 && !head32;
      if (al) {
        int Sw, kps;   // re-planned for THIS part: a last, smaller sub-batch has fewer k-steps
        const int kp = b16 ? WPB_KP : 16;
        wgrad_pw_wide_plan(ctx->num_cus, &dd, Sw, kps, kp);
        if (Sw > S) {  // never more partial slabs than the workspace was sized for
          kps = (int)sg_cdiv(sg_cdiv((int64_t)p.P, kp), S);
          Sw = (int)sg_cdiv(sg_cdiv((int64_t)p.P, kp), kps);
        }
        p.slabs_per_split = kps;
        parts = Sw;
        return b16 ? launch_wgrad_pw_wide_b16(p, Sw, st) : launch_wgrad_pw_wide(p, Sw, st);
      }
    }
    if (pl.patch) {
      const bool al = aligned16(xs) && aligned16(dys) && p.x_bytes != 0 && p.dy_bytes != 0 && !head32;
      if ((up2 || p.bn.mean) && !al) {
        sg_set_error("sg_conv2d_wgrad: SG_X_UP2 / a BatchNormalization on x need 16-byte aligned operands below 2 GiB (the patch kernel)");
        return SG_EUNSUPPORTED;
      }
      if (al) {
        const int grid = x6wp_grid(ctx->num_cus, &dd);  // a last, smaller sub-batch may have fewer tiles than slots
        parts = grid;
        p.out = out;
        if (b16) return launch_x6wp<1, bf16_t>(p, grid, st);
        return x6_mode() == 2 ? launch_x6wp<1, float>(p, grid, st) : launch_x6wp<3, float>(p, grid, st);
      }
      const int nslab = (int)sg_cdiv((int64_t)p.P, BK);
      sps = (int)sg_cdiv(nslab, S);
      p.slabs_per_split = sps;
      S = (int)sg_cdiv(nslab, sps);
      parts = S;
    }
    if (p.bn.mean) {   // (the patch branch above returned; nothing below applies a BatchNormalization to x)
      sg_set_error("sg_conv2d_wgrad_bn: the launch does not take the patch kernel after all");
      return SG_EUNSUPPORTED;
    }
    if (pl.wide) {  // unaligned operands: the slab kernels, with the plan's share count
      const int nslab = (int)sg_cdiv((int64_t)p.P, BK);
      sps = (int)sg_cdiv(nslab, S);
      p.slabs_per_split = sps;
      S = (int)sg_cdiv(nslab, sps);
      parts = S;
    }
    if (b16 && head32) {  // x bf16, dy fp32 (not a thin 1x1 convolution)
      const int64_t yb32 = (((int64_t)p.P - 1) * p.y_ld + dd.Cout) * 4;
      p.dy_bytes = yb32 < (1ll << 31) ? (uint32_t)yb32 : 0;
      const bool vec4 = (dd.Cin % 4 == 0) && (p.x_ld % 4 == 0) && (dd.Cout % 4 == 0) && (p.y_ld % 4 == 0) &&
                        (((uintptr_t)xs & 7) == 0) && aligned16(dys);
      return dispatch_wgrad_head32(p, S, vec4, st);
    }
    if (b16) {
      const bool vec8 = (dd.Cin % 8 == 0) && (p.x_ld % 8 == 0) && (dd.Cout % 8 == 0) && (p.y_ld % 8 == 0) && aligned16(xs) && aligned16(dys);
      const bool vec4 = (dd.Cin % 4 == 0) && (p.x_ld % 4 == 0) && (dd.Cout % 4 == 0) && (p.y_ld % 4 == 0) &&
                        (((uintptr_t)xs & 7) == 0) && (((uintptr_t)dys & 7) == 0);
      return dispatch_wgrad_b16(p, S, vec8, vec4, st);
    }
    const bool vec = (dd.Cin % 4 == 0) && (p.x_ld % 4 == 0) && (dd.Cout % 4 == 0) && (p.y_ld % 4 == 0) &&
                     aligned16(xs) && aligned16(dys);
    return dispatch_wgrad(p, S, vec, st);
  };
  SG_CHECK_ARG(!(head32 && pl.chunks > 1), "sg_conv2d_wgrad: softmax head beyond 2 GiB");
  int total_parts = pl.S;
  if (pl.chunks > 1) {
    // sub-batches of whole images, each writing its S partial slabs one after the other; one reduce over all of them
    const int64_t xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
    const int64_t slab = (int64_t)K_all * d->Cout;
    total_parts = 0;
    for (int c = 0; c < pl.chunks; ++c) {
      sg_conv_desc sub = *d;
      const int n0 = c * pl.nb;
      sub.N = (d->N - n0 < pl.nb) ? d->N - n0 : pl.nb;
      int parts = pl.S;
      rc = launch_part(sub, (const char*)x + (int64_t)n0 * (up2 ? (d->H / 2) * (d->W / 2) : d->H * d->W) * xl * eb,
                       (const char*)dy + (int64_t)n0 * d->Ho * d->Wo * yl * eb,
                       (float*)ws + (int64_t)total_parts * slab, parts, pl.slabs_per_split);
      if (rc) return rc;
      total_parts += parts;
    }
  } else {
    int parts = pl.S;
    rc = launch_part(*d, x, dy, pl.S > 1 ? (float*)ws : (float*)dw, parts, pl.slabs_per_split);
    if (rc) return rc;
    total_parts = parts;
  }
  if (total_parts > 1) {
    const int64_t n = (int64_t)K_all * d->Cout;
    const bool v4 = (n % 4 == 0) && aligned16(ws) && aligned16(dw);
    int64_t blocks = sg_cdiv(n, v4 ? 1024 : 256);
    if (blocks > 2048) blocks = 2048;
    // SG_REDUCE_Z4=1: the four-lane kernel.  Measured (three alternating repetitions, gpurun_out/r4j): 73.98 against 74.08 ms per
    // step - the reduce launches sit on the side stream behind their filter gradient and hide either way; the one-lane
    // kernel of rounds 1 - 3 stays the default
    static const int z4 = getenv("SG_REDUCE_Z4") ? atoi(getenv("SG_REDUCE_Z4")) : 0;
    if (v4 && z4 && total_parts >= 8) {
      int64_t b4 = sg_cdiv(n, 256);
      if (b4 > 4096) b4 = 4096;
      hipLaunchKernelGGL(reduce_splits_z4_kernel, dim3((unsigned)b4), dim3(256), 0, st, (const float*)ws, (float*)dw, n, total_parts);
    } else if (v4)
      hipLaunchKernelGGL(reduce_splits_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, (float*)dw, n, total_parts);
    else
      hipLaunchKernelGGL(reduce_splits_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, (float*)dw, n, total_parts);
    SG_LAUNCH_CHECK("reduce_splits_kernel");
  }
  return bias_grad();
}

// ---- planes-in filter gradient (round 5; conv_x6.h: wgrad_x6_kernel<.., PIN>) -------------------------------------------
int sg_split_planes(sg_ctx* ctx, void* stream, const void* x, int64_t rows, int C, int ld, void* planes) {
  SG_CHECK_ARG(ctx && x && planes && rows > 0 && C > 0, "sg_split_planes: bad argument");
  const int xl = ld ? ld : C;
  SG_CHECK_ARG((C % 4 == 0) && (xl % 4 == 0) && xl >= C && aligned16(x) && aligned16(planes), "sg_split_planes: C and ld must be multiples of 4, pointers 16-byte aligned");
  int64_t blocks = sg_cdiv(rows * (C / 4), 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(x6w_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (unsigned short*)planes, rows, C, xl);
  SG_LAUNCH_CHECK("x6w_split_kernel");
  return 0;
}

static bool wgrad_planes_geom(const sg_ctx* ctx, const sg_conv_desc* d, WgradPlan* out) {
  if (x6_mode() != 1) return false;
  if (d->stride != 1 || d->Ho != d->H || d->Wo != d->W || (d->Wo % 32) || (d->Cin % 8) || (d->Cout % 8) || d->Cout < 16) return false;
  if ((d->x_ld && d->x_ld != d->Cin) || (d->y_ld && d->y_ld != d->Cout)) return false;   // planes are dense
  const int64_t xp = (int64_t)d->N * d->H * d->W * d->Cin * 2, yp = (int64_t)d->N * d->Ho * d->Wo * d->Cout * 2;
  const int64_t sh = ((int64_t)d->pad_t * d->W + d->pad_l + 64) * d->Cin * 4;
  if (3 * xp + 2 * sh >= (1ll << 31) || 3 * yp >= (1ll << 31)) return false;
  WgradPlan pl = plan_wgrad(ctx->num_cus, d, false);
  if (pl.wide || pl.patch || pl.chunks > 1 || thin_ok(d)) return false;
  // SG_WGRAD_PIN_SMUL: more, shorter pixel shares than the fp32-operand plan (the planes are 1.5 x the bytes per pixel: the share
  // of x that the workgroups of an XCD walk together should still fit its L2)
  static const int smul = getenv("SG_WGRAD_PIN_SMUL") ? atoi(getenv("SG_WGRAD_PIN_SMUL")) : 1;
  if (smul > 1 && pl.S > 1) {
    const int64_t nslab = sg_cdiv((int64_t)d->N * d->Ho * d->Wo, BK);
    int64_t S = (int64_t)pl.S * smul;
    if (S > nslab / 4) S = nslab / 4 > 0 ? nslab / 4 : 1;
    pl.slabs_per_split = (int)sg_cdiv(nslab, S);
    pl.S = (int)sg_cdiv(nslab, pl.slabs_per_split);
    pl.dw_part_bytes = (size_t)pl.S * d->KH * d->KW * d->Cin * d->Cout * 4;
  }
  if (out) *out = pl;
  return true;
}

size_t sg_conv2d_wgrad_planes_ws_bytes(const sg_ctx* ctx, const sg_conv_desc* d) {
  WgradPlan pl;
  if (!ctx || !d || check_desc(d, "sg_conv2d_wgrad_planes_ws_bytes") || !wgrad_planes_geom(ctx, d, &pl)) return 0;
  return pl.dw_part_bytes + 512;
}

int sg_conv2d_wgrad_planes_supported(const sg_ctx* ctx, const sg_conv_desc* d) {
  return (ctx && d && check_desc(d, "sg_conv2d_wgrad_planes_supported") == 0 && wgrad_planes_geom(ctx, d, nullptr)) ? 1 : 0;
}

int sg_conv2d_wgrad_planes(sg_ctx* ctx, void* stream, const sg_conv_desc* d, const void* x_planes, const void* dy_planes, void* dw,
                           void* ws, size_t ws_bytes) {
  SG_CHECK_ARG(ctx != nullptr, "sg_conv2d_wgrad_planes: null ctx");
  int rc = check_desc(d, "sg_conv2d_wgrad_planes");
  if (rc) return rc;
  SG_CHECK_ARG(x_planes && dy_planes && dw && aligned16(x_planes) && aligned16(dy_planes), "sg_conv2d_wgrad_planes: null / unaligned tensor");
  WgradPlan pl;
  if (!wgrad_planes_geom(ctx, d, &pl)) {
    sg_set_error("sg_conv2d_wgrad_planes: geometry outside the planes-in kernel (stride 1 SAME, W %% 32 = 0, channels %% 8 = 0, dense "
                 "planes below 2 GiB, not a wide pointwise / patch-form layer, x6 arithmetic)");
    return SG_EUNSUPPORTED;
  }
  const size_t need = pl.dw_part_bytes + 512;
  if (!ws || ws_bytes < need || !aligned16(ws)) {
    sg_set_error("sg_conv2d_wgrad_planes: workspace %zu < %zu", ws_bytes, need);
    return SG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  const int K_all = d->KH * d->KW * d->Cin;
  WgradParams p;
  memset(&p, 0, sizeof(p));
  p.x = (const float*)x_planes;
  p.dy = (const float*)dy_planes;
  p.out = pl.S > 1 ? (float*)ws : (float*)dw;
  p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_ld = d->Cin;
  p.OH = d->Ho; p.OW = d->Wo; p.Cout = d->Cout; p.y_ld = d->Cout;
  p.stride = 1; p.dil = d->dilation; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  p.K = K_all;
  p.P = d->N * d->Ho * d->Wo;
  p.slabs_per_split = pl.slabs_per_split;
  p.fd_ohow = make_fastdiv((uint32_t)(d->Ho * d->Wo));
  p.fd_ow = make_fastdiv((uint32_t)d->Wo);
  p.fd_c = make_fastdiv((uint32_t)d->Cin);
  p.fd_kw = make_fastdiv((uint32_t)d->KW);
  p.fd_oh = make_fastdiv((uint32_t)d->Ho);
  p.x_plane_bytes = (uint32_t)((int64_t)d->N * d->H * d->W * d->Cin * 2);
  p.dy_plane_bytes = (uint32_t)((int64_t)p.P * d->Cout * 2);
  p.x_bytes = 3 * p.x_plane_bytes;
  p.dy_bytes = 3 * p.dy_plane_bytes;
  {
    static int noskip = -1;
    if (noskip < 0) noskip = getenv("SG_CONV_NOSKIP") ? 1 : 0;
    p.KH_KW = p.K / p.Cin;
    p.skip_slabs = (!noskip && p.dil > 1 && p.KH_KW > 1) ? 1 : 0;
    p.tap_inner = ((conv_l2(true) & 4) && p.KH_KW > 1 && p.Cin % BM == 0) ? 1 : 0;   // as dispatch_wgrad: the same tile order
  }
  {
    static const int abl = getenv("SG_X6_ABLATE") ? atoi(getenv("SG_X6_ABLATE")) & 15 : 0;   // timing-only diagnostics of the PF = 1 form
    p.stagger = abl;
  }
  const int bn = wgrad_bn(p.Cout);
  // SG_WGRAD_PIN_PF: 1 = single LDS buffer, two workgroups per CU (the fp32-operand kernel's default structure); 2 = one workgroup
  // per CU, two buffers, the stores of slab s + 1 and the loads of slab s + 3 woven between the MFMAs of slab s
  // Measured (scripts/wgrad_planes_bench.py, fourteen long-K shapes, gpurun_out/r5h): 7301 us with PF 1, 7577 with PF 2, 8485 for
  // the fp32-operand kernel - the read + MFMA loop itself takes 640 of the 810 us of the ASPP launch either way.
  static const int pf = getenv("SG_WGRAD_PIN_PF") ? atoi(getenv("SG_WGRAD_PIN_PF")) : 1;
  static const int w22 = getenv("SG_WGRAD_PIN_W22") ? atoi(getenv("SG_WGRAD_PIN_W22")) : 0;   // experiment: 4 waves of 64 x 64 per 128 x 128 tile
  if (w22 && bn == 128) {
    rc = (pf == 2) ? launch_wgrad_x6<128, 2, 2, 2, 3, bf16_t, true>(p, pl.S, st) : launch_wgrad_x6<128, 2, 2, 1, 3, bf16_t, true>(p, pl.S, st);
  } else if (pf == 2) {
    if (bn == 128) rc = launch_wgrad_x6<128, 2, 4, 2, 3, bf16_t, true>(p, pl.S, st);
    else if (bn == 64) rc = launch_wgrad_x6<64, 4, 2, 2, 3, bf16_t, true>(p, pl.S, st);
    else rc = launch_wgrad_x6<32, 4, 1, 2, 3, bf16_t, true>(p, pl.S, st);
  } else {
    if (bn == 128) rc = launch_wgrad_x6<128, 2, 4, 1, 3, bf16_t, true>(p, pl.S, st);
    else if (bn == 64) rc = launch_wgrad_x6<64, 4, 2, 1, 3, bf16_t, true>(p, pl.S, st);
    else rc = launch_wgrad_x6<32, 4, 1, 1, 3, bf16_t, true>(p, pl.S, st);
  }
  if (rc) return rc;
  if (pl.S > 1) {
    const int64_t n = (int64_t)K_all * d->Cout;
    const bool v4 = (n % 4 == 0) && aligned16(dw);
    int64_t blocks = sg_cdiv(n, v4 ? 1024 : 256);
    if (blocks > 2048) blocks = 2048;
    if (v4) hipLaunchKernelGGL(reduce_splits_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, (float*)dw, n, pl.S);
    else hipLaunchKernelGGL(reduce_splits_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, (float*)dw, n, pl.S);
    SG_LAUNCH_CHECK("reduce_splits_kernel");
  }
  return 0;
}

size_t sg_bias_grad_ws_bytes(const sg_ctx* ctx, int64_t rows, int C) {
  if (!ctx) return 0;
  return colsum_ws_bytes(ctx->num_cus, rows, C) + 256;
}

int sg_bias_grad(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, int ld, const void* dy, void* dbias,
                 void* ws, size_t ws_bytes) {
  SG_CHECK_ARG(ctx && (dtype == SG_F32 || dtype == SG_BF16) && dy && dbias && rows > 0 && C > 0, "sg_bias_grad: bad argument");
  if (ld == 0) ld = C;
  SG_CHECK_ARG(ld >= C, "sg_bias_grad: ld < C");
  const size_t need = colsum_ws_bytes(ctx->num_cus, rows, C);
  if (!ws || ws_bytes < need) {
    sg_set_error("sg_bias_grad: workspace %zu < %zu", ws_bytes, need);
    return SG_EWORKSPACE;
  }
  if (dtype == SG_BF16)
    return launch_colsum<bf16_t>(ctx->num_cus, (const bf16_t*)dy, rows, C, ld, (float*)dbias, (float*)ws, (hipStream_t)stream);
  return launch_colsum<float>(ctx->num_cus, (const float*)dy, rows, C, ld, (float*)dbias, (float*)ws, (hipStream_t)stream);
}

int sg_dense_fwd(sg_ctx* ctx, void* stream, int dtype, int rows, int in, int out, const void* x, const void* w,
                 const void* bias, void* y, int flags) {
  sg_conv_desc d = {};
  d.N = rows; d.H = 1; d.W = 1; d.Cin = in; d.Cout = out; d.KH = 1; d.KW = 1; d.stride = 1; d.dilation = 1;
  d.Ho = 1; d.Wo = 1;
  return sg_conv2d_fwd(ctx, stream, dtype, &d, x, w, bias, y, flags);
}

}  // extern "C"
