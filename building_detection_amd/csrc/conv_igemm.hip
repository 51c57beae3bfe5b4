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
#include <type_traits>

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDA = BK + 4;  // floats; 144-byte rows

template <int V>
using IC = std::integral_constant<int, V>;

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
  FastDiv fd_ohow, fd_ow, fd_c, fd_kw, fd_spt;
};

// UT ("uniform tap"): Cin % 32 == 0 or a 1x1 kernel, so every 32-deep slab lies inside ONE filter tap.  The
// per-row source offsets and bounds flags then change only when the slab stream crosses a tap boundary (every
// Cin/32 slabs) and a slab's gather costs a handful of adds instead of ~170 VALU/SALU instructions of
// div/mod, bounds and 64-bit address arithmetic ahead of the first MFMA.
template <int BN, int WGM, int WGN, int PF, bool VEC, bool UT>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN / 2) void igemm_conv_kernel(const IgemmParams p) {
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
  const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t tile_m = bid / ntn, tile_n = bid - tile_m * ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread A rows: NA rows (r0 + RS j), one 4-wide k chunk (kc) ------------------------------
  const int kc = t & 7, r0 = t >> 3;
  int row_base[NA], row_oh[NA], row_ow[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int m = m0 + r0 + RS * j;
    if (m < p.M) {
      uint32_t n, rem, oh, ow;
      fd_divmod((uint32_t)m, p.fd_ohow, n, rem);
      fd_divmod(rem, p.fd_ow, oh, ow);
      row_base[j] = n * p.H * p.W;
      row_oh[j] = (int)oh * p.a_mul + p.off_h;
      row_ow[j] = (int)ow * p.a_mul + p.off_w;
    } else {
      row_base[j] = 0;
      row_oh[j] = -(1 << 28);
      row_ow[j] = -(1 << 28);
    }
  }

  f32x4 ra[PF][NA];
  f32x4 rb[PF][NB];

  auto gather_elem_addr = [&](int j, int dh, int dw, bool kvalid, int64_t& off) -> bool {
    int ih = row_oh[j] + dh, iw = row_ow[j] + dw;
    bool v = kvalid;
    if (p.div == 2) {  // only strides 1 and 2 occur on this path (host rejects others for dgrad)
      v = v && (((ih | iw) & 1) == 0);
      ih >>= 1;
      iw >>= 1;
    }
    v = v && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
    off = (int64_t)(row_base[j] + ih * p.W + iw) * p.x_ld;
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
        const f32x4 val = *reinterpret_cast<const f32x4*>(p.x + (v ? off + ci : 0));
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
          const float val = p.x[v ? off + ci : 0];
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

  auto compute = [&](int buf) {
    const float* a = As + buf * BM * LDA;
    const float* b = Bs + buf * BK * LDB;
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
        for (int j = 0; j < TN; ++j) bf[j] = b[(kk * 8 + 4 * lh + e) * LDB + wn + 32 * j + lr];
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
  int spt = 1;
  bool use_map = false;
  int* tapinfo = reinterpret_cast<int*>(Bs + 2 * BK * LDB);
  if constexpr (VEC && UT) {
    if (p.skip_taps && ntaps > 1) {  // uniform
      spt = p.C / BK;
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
      nslab = nact * spt;
      use_map = true;
    }
  }
  auto k0_of = [&](int s) -> int {
    if (use_map) {
      const int ti = (int)fd_div((uint32_t)s, p.fd_spt);
      return tapinfo[ti] * p.C + (s - ti * spt) * BK;
    }
    return s * BK;
  };
  const int last = nslab - 1;
  // Stagger (MI355X_MICROARCH "two waves per SIMD", item 9): waves i and i + nwaves/2 share a SIMD and run the
  // same program between the same barriers; the upper half defers its gather (address math + global loads)
  // until after its MFMAs, so on every SIMD one wave's VALU phase overlaps its partner's matrix phase.
  const bool late = (p.stagger != 0) && (__builtin_amdgcn_readfirstlane(t >> 6) >= (NT / 128));
  if (nslab > 0) {
  if constexpr (PF == 1) {
    load_AB(k0_of(0), IC<0>{});
    store_AB(0, IC<0>{});
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
      const int buf = s & 1;
      load_AB(k0_of(s < last ? s + 1 : last), IC<0>{});  // tail reloads the last slab (unused): no branch
      compute(buf);
      store_AB(buf ^ 1, IC<0>{});
      __syncthreads();
    }
  } else {
    // two register sets: while slab s is computed from LDS, slab s+1 sits in one set (landing) and slab s+2
    // is being fetched into the other; the store of a set waits only for that set's (older) loads
    load_AB(k0_of(0), IC<0>{});
    load_AB(k0_of(1 < last ? 1 : last), IC<1>{});
    store_AB(0, IC<0>{});
    __syncthreads();
    const bool do_ld = !(p.ablate & 1), do_st = !(p.ablate & 2);
    for (int s = 0; s < nslab; s += 2) {
      const int ka = k0_of(s + 2 < last ? s + 2 : last), kb = k0_of(s + 3 < last ? s + 3 : last);
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
          p.y[(int64_t)row * p.y_ld + col] = v;
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
  FastDiv fd_ohow, fd_ow, fd_c, fd_kw;
};

// FAST: stride-1 "same" convolution (H == Ho, W == Wo) with 16-byte-aligned channel runs: the source pixel of
// output pixel p under tap (dh,dw) is p + dh*W + dw, i.e. LINEAR in p, so the gather offset advances by a constant
// per slab; only the tap's bounds flags depend on (oh, ow), which are carried incrementally (no div/mod), and
// invalid taps / rows are sent to an out-of-range buffer offset (hardware returns 0).
template <int BN, int WGM, int WGN, int PF, bool VEC, bool FAST>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN / 2) void igemm_wgrad_kernel(const WgradParams p) {
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
  // with slab elimination the work per tile depends on its tap (centre-row taps skip nothing), and an
  // XCD-contiguous order would park all the heavy taps on two XCDs: spread tiles round-robin instead
  const uint32_t bid = p.skip_slabs ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t tile_r = bid / ntn, tile_n = bid - tile_r * ntn;
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

  const int slab_begin = blockIdx.z * p.slabs_per_split;
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
  if constexpr (FAST) {
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
    if constexpr (FAST) {
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
        const f32x4 val = *reinterpret_cast<const f32x4*>(p.x + (v ? (int64_t)(pixbase + ih * p.W + iw) * p.x_ld + ci_e[0] : 0));
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        ra[S][j] = v ? val : z;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ih = (int)oh * p.stride + dh[e], iw = (int)ow * p.stride + dw[e];
          const bool v = pv && rvalid[e] && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
          const float val = p.x[v ? (int64_t)(pixbase + ih * p.W + iw) * p.x_ld + ci_e[e] : 0];
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
        const f32x4 val = *reinterpret_cast<const f32x4*>(p.dy + (v ? (int64_t)pp * p.y_ld + n : 0));
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        rb[S][i] = v ? val : z;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool v = (pp < p.P) && (n + e < p.Cout);
          const float val = p.dy[v ? (int64_t)pp * p.y_ld + n + e : 0];
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

  float* out = p.out + (int64_t)blockIdx.z * p.K * p.Cout;
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

// out[i] = sum_z part[z][i]  in fixed z order
__global__ void reduce_splits_kernel(const float* __restrict__ part, float* __restrict__ out, int64_t n, int S) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float s = part[i];
    for (int z = 1; z < S; ++z) s += part[(int64_t)z * n + i];
    out[i] = s;
  }
}

// bias gradient = column sums of dy[rows][C] (pixel stride ld), through the fixed-order segmented reducer
struct ColSumOp {
  static constexpr int NOUT = 1;
  const float* __restrict__ a;
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

int launch_colsum(int num_cus, const float* dy, int64_t rows, int C, int ld, float* out, float* part, hipStream_t st) {
  const bool vec = (C % 4 == 0) && (ld % 4 == 0) && sg_aligned16(dy);
  const SegPlan pl = seg_plan<1>(num_cus, 1, rows, C, vec);
  ColSumOp op;
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

template <int BN, int WGM, int WGN, int PF, bool VEC, bool UT>
int launch_igemm_ut(const IgemmParams& p, hipStream_t st) {
  constexpr size_t lds = (size_t)(2 * BM * LDA + 2 * BK * BN) * sizeof(float) + 256;  // + tapinfo[64]
  static bool attr_done = false;  // idempotent; racing threads set the same value
  if (!attr_done) {
    int rc = set_dyn_lds(igemm_conv_kernel<BN, WGM, WGN, PF, VEC, UT>, lds);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.M, BM) * sg_cdiv(p.Nout, BN);
  if (tiles <= 0 || tiles > 0x7fffffff) {
    sg_set_error("igemm: bad tile count %lld", (long long)tiles);
    return SG_EINVAL;
  }
  hipLaunchKernelGGL((igemm_conv_kernel<BN, WGM, WGN, PF, VEC, UT>), dim3((unsigned)tiles), dim3(64 * WGM * WGN), lds, st, p);
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

int dispatch_igemm(const IgemmParams& p_in, bool vec, int num_cus, hipStream_t st) {
  IgemmParams p = p_in;
  {
    static int noskip = -1;
    if (noskip < 0) noskip = getenv("SG_CONV_NOSKIP") ? 1 : 0;  // A/B switch for the padding-tap elimination
    const int ntaps = p.K / p.C;
    p.skip_taps = (!noskip && (p.k_mul > 1 || p.k_mul < -1) && ntaps > 1 && ntaps <= 64) ? 1 : 0;
    p.fd_spt = make_fastdiv((uint32_t)(p.C / BK > 0 ? p.C / BK : 1));
  }
  const int bn = pick_bn(p.M, p.Nout, num_cus);
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

template <int BN, int WGM, int WGN, int PF, bool VEC, bool FAST>
int launch_wgrad_f(const WgradParams& p, int S, hipStream_t st) {
  constexpr size_t lds = (size_t)(2 * BK * BM + 2 * BK * BN) * sizeof(float) + (FAST ? (2 * 1024 + 4) * sizeof(int) : 0);
  static bool attr_done = false;
  if (!attr_done) {
    int rc = set_dyn_lds(igemm_wgrad_kernel<BN, WGM, WGN, PF, VEC, FAST>, lds);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.K, BM) * sg_cdiv(p.Cout, BN);
  hipLaunchKernelGGL((igemm_wgrad_kernel<BN, WGM, WGN, PF, VEC, FAST>), dim3((unsigned)tiles, 1, (unsigned)S), dim3(64 * WGM * WGN), lds, st, p);
  SG_LAUNCH_CHECK("igemm_wgrad_kernel");
  return 0;
}

template <int BN, int WGM, int WGN, int PF, bool VEC>
int launch_wgrad(const WgradParams& p, int S, hipStream_t st) {
  if constexpr (VEC && PF == 2) {
    const bool fast = p.stride == 1 && p.OH == p.H && p.OW == p.W && p.x_bytes != 0 && p.dy_bytes != 0;
    if (fast) return launch_wgrad_f<BN, WGM, WGN, PF, true, true>(p, S, st);
  }
  return launch_wgrad_f<BN, WGM, WGN, PF, VEC, false>(p, S, st);
}

inline int wgrad_bn(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

int dispatch_wgrad(const WgradParams& p_in, int S, bool vec, hipStream_t st) {
  WgradParams p = p_in;
  {
    static int noskip = -1;
    if (noskip < 0) noskip = getenv("SG_CONV_NOSKIP") ? 1 : 0;
    p.KH_KW = p.K / p.Cin;
    p.skip_slabs = (!noskip && p.dil > 1 && p.KH_KW > 1) ? 1 : 0;
  }
  const int bn = wgrad_bn(p.Cout);
  const int var = conv_variant() & 3;
  p.stagger = (conv_variant() >> 2) & 1;
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
  const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
  SG_CHECK_ARG(xl >= d->Cin && yl >= d->Cout, "%s: pixel stride smaller than channel count", who);
  SG_CHECK_ARG((int64_t)d->N * d->H * d->W * xl < (1ll << 31) && (int64_t)d->N * d->Ho * d->Wo * yl < (1ll << 31),
               "%s: tensor exceeds 2^31 elements", who);
  return 0;
}

struct WgradPlan {
  int S;
  int slabs_per_split;
  size_t dw_part_bytes;
  size_t bias_part_bytes;
};

// Split of the pixel reduction over S workgroups per tile.  Modelled time = MFMA work / (fraction of the
// 2*CUs workgroup slots kept busy over whole waves) + the traffic of writing and re-adding S partial slabs;
// the S with the smallest modelled time wins (e.g. ASPP: 288 tiles -> S = 7: 2016 workgroups = 3.94 waves).
WgradPlan plan_wgrad(int num_cus, const sg_conv_desc* d) {
  WgradPlan pl;
  const int64_t K = (int64_t)d->KH * d->KW * d->Cin;
  const int64_t P = (int64_t)d->N * d->Ho * d->Wo;
  const int bn = wgrad_bn(d->Cout);
  const int64_t tiles = sg_cdiv(K, BM) * sg_cdiv(d->Cout, bn);
  const int64_t nslab = sg_cdiv(P, BK);
  const int64_t slots = 2 * (int64_t)num_cus;
  const double flops = 2.0 * (double)tiles * BM * bn * (double)P;  // padded tile work
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
    const double t = flops / (eff * 110e12) + 2.0 * part_bytes / 3.0e12 + (S > 1 ? 3e-6 : 0.0);
    if (t < best_t * 0.999) { best_t = t; best_S = S; }
  }
  pl.slabs_per_split = (int)sg_cdiv(nslab, best_S);
  pl.S = (int)sg_cdiv(nslab, pl.slabs_per_split);
  pl.dw_part_bytes = pl.S > 1 ? (size_t)pl.S * K * d->Cout * 4 : 0;
  pl.bias_part_bytes = colsum_ws_bytes(num_cus, P, d->Cout);
  return pl;
}

}  // namespace

extern "C" {

int sg_conv2d_fwd(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* w,
                  const void* bias, void* y, int flags) {
  SG_CHECK_ARG(ctx != nullptr, "sg_conv2d_fwd: null ctx");
  SG_CHECK_ARG(dtype == SG_F32, "sg_conv2d_fwd: only SG_F32 is implemented");
  int rc = check_desc(d, "sg_conv2d_fwd");
  if (rc) return rc;
  SG_CHECK_ARG(x && w && y, "sg_conv2d_fwd: null tensor");
  SG_CHECK_ARG(!(flags & SG_EPI_BIAS) || bias, "sg_conv2d_fwd: SG_EPI_BIAS without bias");
  IgemmParams p;
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
  {
    const int64_t xb = (((int64_t)d->N * d->H * d->W - 1) * p.x_ld + d->Cin) * 4, wb = (int64_t)p.K * d->Cout * 4;
    p.x_bytes = xb < (1ll << 31) ? (uint32_t)xb : 0;
    p.w_bytes = wb < (1ll << 31) ? (uint32_t)wb : 0;
  }
  const bool vec = (d->Cin % 4 == 0) && (p.x_ld % 4 == 0) && (d->Cout % 4 == 0) && aligned16(x) && aligned16(w);
  return dispatch_igemm(p, vec, ctx->num_cus, (hipStream_t)stream);
}

size_t sg_conv2d_dgrad_ws_bytes(const sg_conv_desc* d) {
  if (!d) return 0;
  return (size_t)d->KH * d->KW * d->Cin * d->Cout * sizeof(float);
}

int sg_conv2d_dgrad(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* dy, const void* w,
                    const void* bias, void* dx, int flags, void* ws, size_t ws_bytes) {
  SG_CHECK_ARG(ctx != nullptr, "sg_conv2d_dgrad: null ctx");
  SG_CHECK_ARG(dtype == SG_F32, "sg_conv2d_dgrad: only SG_F32 is implemented");
  int rc = check_desc(d, "sg_conv2d_dgrad");
  if (rc) return rc;
  SG_CHECK_ARG(dy && w && dx, "sg_conv2d_dgrad: null tensor");
  SG_CHECK_ARG(!(flags & SG_EPI_BIAS) || bias, "sg_conv2d_dgrad: SG_EPI_BIAS without bias");
  if (d->stride != 1 && d->stride != 2) {
    sg_set_error("sg_conv2d_dgrad: stride %d unsupported (the path uses strides 1 and 2 only)", d->stride);
    return SG_EUNSUPPORTED;
  }
  const size_t need = sg_conv2d_dgrad_ws_bytes(d);
  if (!ws || ws_bytes < need) {
    sg_set_error("sg_conv2d_dgrad: workspace %zu < %zu", ws_bytes, need);
    return SG_EWORKSPACE;
  }
  SG_CHECK_ARG(aligned16(ws), "sg_conv2d_dgrad: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  float* wt = (float*)ws;
  {
    dim3 grid((unsigned)sg_cdiv(d->Cout, 32), (unsigned)sg_cdiv(d->Cin, 32), (unsigned)(d->KH * d->KW));
    hipLaunchKernelGGL(transpose_taps_kernel, grid, dim3(256), 0, st, (const float*)w, wt, d->Cin, d->Cout);
    SG_LAUNCH_CHECK("transpose_taps_kernel");
  }
  IgemmParams p;
  p.x = (const float*)dy;
  p.w = wt;
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
  {
    const int64_t xb = (((int64_t)d->N * d->Ho * d->Wo - 1) * p.x_ld + d->Cout) * 4, wb = (int64_t)p.K * d->Cin * 4;
    p.x_bytes = xb < (1ll << 31) ? (uint32_t)xb : 0;
    p.w_bytes = wb < (1ll << 31) ? (uint32_t)wb : 0;
  }
  const bool vec = (d->Cout % 4 == 0) && (p.x_ld % 4 == 0) && (d->Cin % 4 == 0) && aligned16(dy);
  return dispatch_igemm(p, vec, ctx->num_cus, st);
}

size_t sg_conv2d_wgrad_ws_bytes(const sg_ctx* ctx, const sg_conv_desc* d) {
  if (!ctx || !d) return 0;
  WgradPlan pl = plan_wgrad(ctx->num_cus, d);
  return pl.dw_part_bytes + pl.bias_part_bytes + 512;
}

int sg_conv2d_wgrad(sg_ctx* ctx, void* stream, int dtype, const sg_conv_desc* d, const void* x, const void* dy,
                    void* dw, void* dbias, void* ws, size_t ws_bytes) {
  SG_CHECK_ARG(ctx != nullptr, "sg_conv2d_wgrad: null ctx");
  SG_CHECK_ARG(dtype == SG_F32, "sg_conv2d_wgrad: only SG_F32 is implemented");
  int rc = check_desc(d, "sg_conv2d_wgrad");
  if (rc) return rc;
  SG_CHECK_ARG(x && dy && dw, "sg_conv2d_wgrad: null tensor");
  const WgradPlan pl = plan_wgrad(ctx->num_cus, d);
  const size_t need = pl.dw_part_bytes + pl.bias_part_bytes + 512;
  if (!ws || ws_bytes < need) {
    sg_set_error("sg_conv2d_wgrad: workspace %zu < %zu", ws_bytes, need);
    return SG_EWORKSPACE;
  }
  SG_CHECK_ARG(aligned16(ws), "sg_conv2d_wgrad: workspace must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  WgradParams p;
  p.x = (const float*)x;
  p.dy = (const float*)dy;
  p.out = pl.S > 1 ? (float*)ws : (float*)dw;
  p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_ld = d->x_ld ? d->x_ld : d->Cin;
  p.OH = d->Ho; p.OW = d->Wo; p.Cout = d->Cout; p.y_ld = d->y_ld ? d->y_ld : d->Cout;
  p.stride = d->stride; p.dil = d->dilation; p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  p.K = d->KH * d->KW * d->Cin;
  p.P = d->N * d->Ho * d->Wo;
  p.slabs_per_split = pl.slabs_per_split;
  p.fd_ohow = make_fastdiv((uint32_t)(d->Ho * d->Wo));
  p.fd_ow = make_fastdiv((uint32_t)d->Wo);
  p.fd_c = make_fastdiv((uint32_t)d->Cin);
  p.fd_kw = make_fastdiv((uint32_t)d->KW);
  {
    const int64_t xb = (((int64_t)d->N * d->H * d->W - 1) * p.x_ld + d->Cin) * 4;
    const int64_t yb = (((int64_t)p.P - 1) * p.y_ld + d->Cout) * 4;
    p.x_bytes = xb < (1ll << 31) ? (uint32_t)xb : 0;
    p.dy_bytes = yb < (1ll << 31) ? (uint32_t)yb : 0;
  }
  const bool vec = (d->Cin % 4 == 0) && (p.x_ld % 4 == 0) && (d->Cout % 4 == 0) && (p.y_ld % 4 == 0) &&
                   aligned16(x) && aligned16(dy);
  rc = dispatch_wgrad(p, pl.S, vec, st);
  if (rc) return rc;
  if (pl.S > 1) {
    const int64_t n = (int64_t)p.K * p.Cout;
    int64_t blocks = sg_cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_splits_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, (float*)dw, n, pl.S);
    SG_LAUNCH_CHECK("reduce_splits_kernel");
  }
  if (dbias) {
    // 256-byte aligned region after the dw partials
    float* part = (float*)((char*)ws + ((pl.dw_part_bytes + 255) & ~(size_t)255));
    rc = launch_colsum(ctx->num_cus, (const float*)dy, (int64_t)p.P, d->Cout, p.y_ld, (float*)dbias, part, st);
    if (rc) return rc;
  }
  return 0;
}

size_t sg_bias_grad_ws_bytes(const sg_ctx* ctx, int64_t rows, int C) {
  if (!ctx) return 0;
  return colsum_ws_bytes(ctx->num_cus, rows, C) + 256;
}

int sg_bias_grad(sg_ctx* ctx, void* stream, int dtype, int64_t rows, int C, int ld, const void* dy, void* dbias,
                 void* ws, size_t ws_bytes) {
  SG_CHECK_ARG(ctx && dtype == SG_F32 && dy && dbias && rows > 0 && C > 0, "sg_bias_grad: bad argument");
  if (ld == 0) ld = C;
  SG_CHECK_ARG(ld >= C, "sg_bias_grad: ld < C");
  const size_t need = colsum_ws_bytes(ctx->num_cus, rows, C);
  if (!ws || ws_bytes < need) {
    sg_set_error("sg_bias_grad: workspace %zu < %zu", ws_bytes, need);
    return SG_EWORKSPACE;
  }
  return launch_colsum(ctx->num_cus, (const float*)dy, rows, C, ld, (float*)dbias, (float*)ws, (hipStream_t)stream);
}

int sg_dense_fwd(sg_ctx* ctx, void* stream, int dtype, int rows, int in, int out, const void* x, const void* w,
                 const void* bias, void* y, int flags) {
  sg_conv_desc d = {};
  d.N = rows; d.H = 1; d.W = 1; d.Cin = in; d.Cout = out; d.KH = 1; d.KW = 1; d.stride = 1; d.dilation = 1;
  d.Ho = 1; d.Wo = 1;
  return sg_conv2d_fwd(ctx, stream, dtype, &d, x, w, bias, y, flags);
}

}  // extern "C"
