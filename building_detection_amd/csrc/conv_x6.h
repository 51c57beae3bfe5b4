// fp32 convolution on the bf16 matrix pipe ("x6"): included by conv_igemm.hip inside its anonymous namespace.
//
// CDNA4 runs v_mfma_f32_32x32x16_bf16 at 16x the FLOP rate of v_mfma_f32_32x32x2_f32 (2.4 PFLOP/s against
// 155 TFLOP/s measured in bare loops on this part, profiles/r01_exp_bf16x6.txt).  An fp32 value is the exact sum of
// three bf16 values (8 + 8 + 8 significand bits, round-to-nearest splits), a bf16 x bf16 product is exact in fp32
// and the MFMA accumulates in fp32, so
//     a*b = (a1+a2+a3)(b1+b2+b3) ~= a3*b1 + a1*b3 + a2*b2 + a2*b1 + a1*b2 + a1*b1      (six MFMA passes)
// drops only terms below 2^-25 |a||b|.  Measured on the ASPP reduction length K = 18432 against fp64: max error
// 1.6e-7 of sum|a||b| for the six passes, 2.0e-7 for the native fp32 MFMA (which equals a sequential fmaf chain):
// the emulation is at least as accurate as the instruction it replaces, at 6/16 of its issue time.
//
// Data flow of one 128 x BN x 32 slab (512 threads, two workgroups per CU, single LDS buffer):
//   A: the im2col gather of the fp32 kernel (hardware buffer addressing, padding taps skipped), two float4 per
//      thread prefetched into registers one slab ahead; split into the three planes when written to LDS.
//   B: the weights are split ONCE per launch by split3_weights_kernel into three bf16 planes [n][k] (k
//      contiguous, zero-padded to whole tiles), so a thread stages one 16-byte chunk per plane, no arithmetic.
//   LDS: per plane [row][64 bytes], the 16-byte chunks of a row XOR-swizzled by the row (XPITCH / xswz below:
//      conflict-free reads and writes); fragments are plain row reads for A and B alike (lane (r, h) takes
//      k = 8h..8h+7 of row r).
//   Per 16-deep k-step a wave (64 x 32 sub-tile) reads 9 fragments and issues 12 MFMAs (0.75 ds_read_b128 per
//   MFMA; the LDS array saturates at 2).
#pragma once

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// LDS rows of a plane slab are the bare 64 bytes (32 k as bf16); the four 16-byte chunks of row r are stored at
// slot (chunk ^ xswz(r)).  Reads: the 16 lanes of a ds_read_b128 group (rows {0-3,12-15,20-27} or {4-11,16-19,28-31})
// then hit 16 distinct 16-byte bank groups; writes: two consecutive rows fill the 32 write banks exactly once
// (the padded 80-byte rows this replaces cost 28 % of the LDS-active cycles in bank conflicts, SQ_LDS_BANK_CONFLICT).
// Round 3: the row group (row >> 2) & 3 is sent through the bijection (0, 2, 3, 1) instead of the identity.  Any bijection keeps the
// 32-row pattern above conflict-free (a lane group's four 4-row groups are distinct and share one chunk); this one also serves
// the 16x16x32 fragments (lane -> row lane & 15, chunk lane >> 4: a lane group then holds rows 0-3 and 12-15 with chunk c and rows
// 4-11 with chunk c ^ 1, and {g0, g3, 1 ^ g1, 1 ^ g2} must be four different slots).
constexpr int XPITCH = 64;
__device__ __forceinline__ int xswz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

// (x0, x1) -> three packed bf16 pairs (element 0 in the low half), x = h + m + l up to 2^-25 |x|
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  f32x2_t v = {x0, x1};
  h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  f32x2_t hf = {__uint_as_float(h << 16), __uint_as_float(h & 0xffff0000u)};
  f32x2_t r = v - hf;
  m = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
  f32x2_t mf = {__uint_as_float(m << 16), __uint_as_float(m & 0xffff0000u)};
  f32x2_t s = r - mf;
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(s, bf16x2_t));
}

// B[k][n] = w[tap*s_tap + kk*s_k + n*s_n]  (k = tap*Ck + kk)  ->  planes[3][Npad][Kpad] bf16, zero padded.
//   forward: Ck = Cin,  s_tap = Cin*Cout, s_k = Cout, s_n = 1      (w is HWIO)
//   dgrad  : Ck = Cout, s_tap = Cin*Cout, s_k = 1,    s_n = Cout   (the per-tap transpose)
// npl = 1 writes the bf16 rounding alone (the bf16 paths: one MFMA pass).  Ckp >= Ck is the per-tap depth of the k
// index: with Ckp > Ck every tap is zero-padded to Ckp channels ("virtual channel padding": the kernel then walks
// whole 32-deep slabs inside one tap for any channel count and multiplies the surplus A columns by these zeros).
// kd > 0: k-block-major planes [npl][Kpad / kd][Npad][kd] (the wide pointwise kernel, conv_pw.h) instead of rows [Npad][Kpad].
__global__ __launch_bounds__(256) void split3_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ planes,
                                                             int K, int N, int Kpad, int Npad, int Ck, int s_tap, int s_k,
                                                             int s_n, int npl, int Ckp, int kd = 0) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    // read with the fast thread index along the contiguous source dimension
    const int ky = (s_n == 1) ? i : tx, nx = (s_n == 1) ? tx : i;
    const int k = k0 + ky, n = n0 + nx;
    float v = 0.f;
    if (k < K && n < N) {
      const int tap = k / Ckp, kk = k - tap * Ckp;
      if (kk < Ck) v = w[(int64_t)tap * s_tap + (int64_t)kk * s_k + (int64_t)n * s_n];
    }
    tile[ky][nx] = v;
  }
  __syncthreads();
  const int64_t plane = (int64_t)Npad * Kpad;
  for (int i = ty; i < 32; i += 8) {
    const int n = n0 + i, k = k0 + tx;  // inside the padded planes (Kpad, Npad multiples of 32) unless kd says otherwise
    if (kd > 0 && (k >= Kpad || n >= Npad)) continue;
    const float x = tile[tx][i];
    unsigned h, m, l;
    split3_pair(x, 0.f, h, m, l);
    const int64_t o = kd > 0 ? ((int64_t)(k / kd) * Npad + n) * kd + (k % kd) : (int64_t)n * Kpad + k;
    planes[o] = (unsigned short)(h & 0xffffu);
    if (npl == 3) {
      planes[plane + o] = (unsigned short)(m & 0xffffu);
      planes[2 * plane + o] = (unsigned short)(l & 0xffffu);
    }
  }
}

inline int x6_kpad(int K, int kd = 32) { return (int)(sg_cdiv(K, kd) * kd); }
inline int x6_npad(int N) { return (int)(sg_cdiv(N, 128) * 128); }
inline size_t x6_planes_bytes(int K, int N, int npl = 3, int kd = 32) { return (size_t)npl * x6_kpad(K, kd) * x6_npad(N) * 2; }
// K of a launch under virtual channel padding: ntaps * roundup(C, 32)
inline int x6_vpad_c(int C) { return (int)(sg_cdiv(C, 32) * 32); }

// PF == 1: single LDS buffer, two barriers per slab, two workgroups per CU (128 VGPRs per wave).
// PF == 2: ONE workgroup per CU with the whole register file (256 VGPRs per wave): LDS double-buffered, one
//   barrier per slab, two register sets (loads run two slabs ahead), and the wave halves staggered - waves
//   0..n/2-1 split + store slab s+1 BEFORE their MFMAs of slab s, waves n/2..n-1 AFTER - so on every SIMD one
//   wave's VALU / LDS-write phase runs under its partner's MFMA phase.  (With the single buffer both workgroups of
//   a CU run in lockstep - store, barrier, compute - and the phases simply add up: measured store 0.41 ms +
//   compute 0.66 ms = 1.02 ms on the ASPP forward; the bare read + MFMA loop reaches 80-88 % of the pipe,
//   profiles/r01_exp_x6_loop.txt.)
// NPL = 3, TA = float: the exact fp32 emulation above.  NPL = 1: ONE bf16 plane per operand and one MFMA per k-step
// and tile (bf16 products, fp32 accumulation) - with TA = float the fp32 activations are rounded to bf16 on their way
// into LDS, with TA = bf16_t (SG_BF16 storage) a thread's 16-byte chunk holds 8 k and goes to LDS as it is (no
// arithmetic at all between the global load and the MFMA); the output is stored as TA.
// MF = 1 (NPL = 3 only): the products on v_mfma_f32_16x16x32_bf16 - one 32-deep k-step per slab, 2 TM x 2 TN blocks of 16 x 16 per
// wave - instead of 32x32x16 (two 16-deep k-steps, TM x TN blocks).  Same cycles per FLOP, same LDS fragment traffic (18 reads per
// slab at TM = 2, TN = 1), but the chip holds a higher clock on this shape under load: the bare x6 loop runs 1.10-1.125x faster
// (scripts/exp_x6_shape.hip, profiles/r03_exp_x6_shape.txt).  A 32-deep instruction adds its products in another order than two
// 16-deep ones, so MF is a property of the LAYER, never of the batch: the multi-tap convolutions take it (dispatch_x6); 1x1
// convolutions, which the wide pointwise kernel (16-deep stages) may serve at another batch size, stay on 32x32x16.
template <int BN, int WGM, int WGN, int PF, int NPL = 3, typename TA = float, int MF = 0>
__global__ __launch_bounds__(64 * WGM * WGN, PF == 2 ? (WGM * WGN + 3) / 4 : WGM * WGN / 2) void conv_x6_kernel(const IgemmParams p) {
  static_assert(NPL == 3 || NPL == 1, "planes");
  static_assert(MF == 0 || NPL == 3, "the 16x16x32 form exists for the six-pass product");
  constexpr bool MF16 = MF != 0;
  static_assert(NPL == 1 || std::is_same<TA, float>::value, "the three-plane split is the fp32 path");
  constexpr bool A16 = !std::is_same<TA, float>::value;  // bf16 storage
  constexpr int EB = EL<TA>::BYTES, CH = EL<TA>::CH;
  constexpr int CPR = BK / CH;             // 16-byte chunks per A row of a slab: 8 (fp32) or 4 (bf16)
  constexpr int NT = 64 * WGM * WGN;
  constexpr int BUFSZ = NPL * (BM + BN) * XPITCH;  // one LDS buffer: A planes then B planes
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NA = (BM * CPR) / NT;     // 16-byte A chunks per thread (rows r0 + RS*j)
  constexpr int RS = NT / CPR;
  constexpr int NBC = BN * 4 * NPL;       // 16-byte chunks of one B slab (NPL planes x BN rows x 4)
  constexpr int NB = (NBC + NT - 1) / NT;
  static_assert(TM >= 1 && TN >= 1 && NA >= 1, "tile too small for the wave layout");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ap = smem;                          // [PF buffers] x { A [NPL][BM][XPITCH], B [NPL][BN][XPITCH] }
  char* Bp = Ap + NPL * BM * XPITCH;
  int* tapinfo = reinterpret_cast<int*>(smem + PF * BUFSZ);  // [64]
  int* row_lin_lds = tapinfo + 64;          // [NA][NT]

  const int t = threadIdx.x;
  const uint32_t ntn = (p.Nout + BN - 1) / BN;
  const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  uint32_t tile_m, tile_n;
  if (p.group_m > 1) {  // grouped order (see igemm_conv_kernel): row tile fastest inside groups of group_m row tiles
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

  // ---- A rows (same gather as igemm_conv_kernel's UT path) ----------------------------------------------
  const int kc = t % CPR, r0 = t / CPR;
  int row_hw[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int m = m0 + r0 + RS * j;
    if (m < p.M) {
      uint32_t n, oh, ow;
      row_to_pixel(p, (uint32_t)m, n, oh, ow);
      const int ohs = (int)oh * p.a_mul + p.off_h, ows = (int)ow * p.a_mul + p.off_w;
      row_lin_lds[j * NT + t] = (int)n * p.H * p.W + ohs * p.W + ows;
      row_hw[j] = (ohs << 16) | (ows & 0xffff);
    } else {
      row_lin_lds[j * NT + t] = 0;
      row_hw[j] = (int)0x80008000u;
    }
  }
  auto gather_elem_addr = [&](int j, int dh, int dw, int64_t& off) -> bool {
    const int ohs = row_hw[j] >> 16, ows = (int)(short)(row_hw[j] & 0xffff);
    int ih = ohs + dh, iw = ows + dw;
    bool v = true;
    const int rl = row_lin_lds[j * NT + t];
    int pix = rl + dh * p.W + dw;
    if (p.div == 2) {
      v = (((ih | iw) & 1) == 0);
      ih >>= 1;
      iw >>= 1;
      pix = rl - ohs * p.W - ows + ih * p.W + iw;
    }
    v = v && ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
    off = (int64_t)pix * p.x_ld;
    return v;
  };

  constexpr unsigned OOB = 0x80000000u;
  int cur_tap = -1;
  unsigned tap_voff[NA];
  unsigned b_voff[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j) tap_voff[j] = OOB;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = t + NT * i;
    const int pl = idx / (BN * 4), rem = idx - pl * (BN * 4);
    const int row = rem >> 2, c = rem & 3;
    // k-block-major planes [pl][Kpad / 32][Npad][32]: the 64 bytes of a row's slab and the rows of a tile are contiguous - a
    // slab of B is BN x 64 bytes of whole cache lines.  (Row planes [n][k] gave each row half a line per slab: the other half
    // was fetched again for the next slab unless it had survived in L1; round 3.)
    b_voff[i] = idx < NBC ? (unsigned)((((pl * (p.Kpad >> 5)) * p.Npad + n0 + row) * 32 + c * 8) * 2) : OOB;
  }
  const int ntaps = p.K / p.C;
  const bool ktail = (p.K % BK) != 0;
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wq), 0, (int)p.w_bytes, 0x00020000);

  u32x4_t ra[PF][NA];
  u32x4_t rb[PF][NB];
  auto load_AB = [&](int k0, auto SET) {
    constexpr int S = decltype(SET)::value;
    const int tap = (int)fd_div((uint32_t)k0, p.fd_c);
    if (tap != cur_tap) {  // uniform
      cur_tap = tap;
      uint32_t kh, kw;
      fd_divmod((uint32_t)tap, p.fd_kw, kh, kw);
      const int dh = (int)kh * p.k_mul, dw = (int)kw * p.k_mul;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int64_t off;
        const bool ok = gather_elem_addr(j, dh, dw, off) && tap < ntaps;
        tap_voff[j] = ok ? (unsigned)off * (unsigned)EB + 16u * kc : OOB;
      }
    }
    const int soff_a = (k0 - tap * p.C) * EB;
    const bool kvalid = !ktail || (k0 + CH * kc < p.K);
#pragma unroll
    for (int j = 0; j < NA; ++j)
      ra[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(kvalid ? tap_voff[j] : OOB), soff_a, 0);
    const int soff_b = (k0 >> 5) * p.Npad * 64;
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)b_voff[i], soff_b, 0);
  };

  // one thread's A chunk of register set S, row r0 + RS j, to buffer `buf`
  auto store_A1 = [&](const u32x4_t v, int j, int buf) {
    const int arow = r0 + RS * j;
    if constexpr (A16) {  // 8 k of bf16: the chunk goes to LDS as it is
      *reinterpret_cast<u32x4_t*>(Ap + buf * BUFSZ + arow * XPITCH + ((kc ^ xswz(arow)) << 4)) = v;
    } else {
      const f32x4 f = __builtin_bit_cast(f32x4, v);
      char* dst = Ap + buf * BUFSZ + arow * XPITCH + (((kc >> 1) ^ xswz(arow)) << 4) + (kc & 1) * 8;
      if constexpr (NPL == 3) {
        unsigned h0, m0_, l0, h1, m1, l1;
        split3_pair(f[0], f[1], h0, m0_, l0);
        split3_pair(f[2], f[3], h1, m1, l1);
        *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){h0, h1};
        *reinterpret_cast<u32x2_t*>(dst + BM * XPITCH) = (u32x2_t){m0_, m1};
        *reinterpret_cast<u32x2_t*>(dst + 2 * BM * XPITCH) = (u32x2_t){l0, l1};
      } else {
        *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3])};
      }
    }
  };
  auto store_AB = [&](auto SET, int buf) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int j = 0; j < NA; ++j) store_A1(ra[S][j], j, buf);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + NT * i;
      if (idx < NBC) {
        const int pl = idx / (BN * 4), rem = idx - pl * (BN * 4);
        const int row = rem >> 2, c = rem & 3;
        *reinterpret_cast<u32x4_t*>(Bp + buf * BUFSZ + (pl * BN + row) * XPITCH + ((c ^ xswz(row)) << 4)) = rb[S][i];
      }
    }
  };

  // ---- MFMA side --------------------------------------------------------------------------------------------
  const int wave = t >> 6, lane = t & 63;
  const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
  // 32x32x16: lane -> row lane & 31 of a 32-row block, k half lane >> 5 of a 16-deep k-step; 16 result registers.
  // 16x16x32: lane -> row lane & 15 of a 16-row block, 8-k chunk lane >> 4 of the 32-deep slab; 4 result registers.
  constexpr int TMB = MF16 ? 2 * TM : TM, TNB = MF16 ? 2 * TN : TN, RB_ = MF16 ? 16 : 32, NR = MF16 ? 4 : 16;
  const int lr = MF16 ? (lane & 15) : (lane & 31), lh = MF16 ? (lane >> 4) : (lane >> 5);
  typedef float accv_t __attribute__((ext_vector_type(NR)));
  accv_t acc[TMB][TNB];
#pragma unroll
  for (int i = 0; i < TMB; ++i)
#pragma unroll
    for (int j = 0; j < TNB; ++j)
#pragma unroll
      for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.f;
  // result element r of block (i, j) in this lane: local row / column inside the wave's sub-tile
  auto acc_row = [&](int i, int r) { return MF16 ? 16 * i + 4 * lh + r : 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh; };
  auto acc_col = [&](int j) { return RB_ * j + lr; };
  const char* a_lane = Ap + (wm + lr) * XPITCH;
  const char* b_lane = Bp + (wn + lr) * XPITCH;
  const int koff16 = (lh ^ xswz(lr)) << 4;   // MF16: this lane's 16-byte chunk (k = 8 lh ..) of a row, after the swizzle
  // byte offset of this lane's 16-byte chunk (k = 16 ks + 8 lh ..) inside its row, after the swizzle; wm, wn and the
  // 32-row sub-tile offsets are multiples of 32, so the swizzle depends on lr only
  const int koff[2] = {((0 + lh) ^ xswz(lr)) << 4, ((2 + lh) ^ xswz(lr)) << 4};

  // All 18 (TM = 2, TN = 1) fragment reads of the slab's two k-steps are issued back to back, then the 24 MFMAs:
  // the MFMAs of k-step 0 start when its nine fragments have landed and cover the flight of k-step 1's.  (Left to
  // itself the scheduler sinks every read to just before its first MFMA behind an lgkmcnt(0) - minimal
  // registers, but the LDS latency is then paid ~12 times per slab and the phase ran at 42 % of the MFMA rate.)
  // (4-wave workgroups have the registers for both k-steps' fragments; 8-wave ones hoist one k-step at a time.)
  constexpr int KH = (NT == 256 || PF == 2) ? 2 : 1;  // k-steps whose fragments are in flight together
  auto compute = [&](int buf) {
    const char* a_lane_b = a_lane + buf * BUFSZ;
    const char* b_lane_b = b_lane + buf * BUFSZ;
    if constexpr (MF16) {
      // the B fragments of the slab once, the A fragments in groups of IH row blocks (all of them where the registers allow:
      // KH == 2; half of them in the 8-wave two-workgroups-per-CU form, which runs at the 128-register cap)
      constexpr int IH = (KH == 2 || TMB == 1) ? TMB : TMB / 2;
      bf16x8_t bf[TNB][3];
#pragma unroll
      for (int j = 0; j < TNB; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8_t*>(b_lane_b + (pl * BN + 16 * j) * XPITCH + koff16);
#pragma unroll
      for (int i0 = 0; i0 < TMB; i0 += IH) {
        bf16x8_t af[IH][3];
#pragma unroll
        for (int i = 0; i < IH; ++i)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            af[i][pl] = *reinterpret_cast<const bf16x8_t*>(a_lane_b + (pl * BM + 16 * (i0 + i)) * XPITCH + koff16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < IH; ++i)
#pragma unroll
          for (int j = 0; j < TNB; ++j) {
            auto& c = acc[i0 + i][j];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][2], bf[j][0], c, 0, 0, 0);   // smallest terms first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int k0s = 0; k0s < 2; k0s += KH) {
      bf16x8_t af[KH][TM][NPL], bf[KH][TN][NPL];
#pragma unroll
      for (int kq = 0; kq < KH; ++kq) {
        const int ks = k0s + kq;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            af[kq][i][pl] = *reinterpret_cast<const bf16x8_t*>(a_lane_b + (pl * BM + 32 * i) * XPITCH + koff[ks]);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            bf[kq][j][pl] = *reinterpret_cast<const bf16x8_t*>(b_lane_b + (pl * BN + 32 * j) * XPITCH + koff[ks]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kq = 0; kq < KH; ++kq) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            if constexpr (NPL == 1) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kq][i][0], bf[kq][j][0], acc[i][j], 0, 0, 0);
              continue;
            }
            // smallest terms first
            constexpr int P2 = NPL == 3 ? 2 : 0, P1 = NPL == 3 ? 1 : 0;  // (NPL == 1 never reaches these)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kq][i][P2], bf[kq][j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kq][i][0], bf[kq][j][P2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kq][i][P1], bf[kq][j][P1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kq][i][P1], bf[kq][j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kq][i][0], bf[kq][j][P1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kq][i][0], bf[kq][j][0], acc[i][j], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  };

  // ---- hand-interleaved step (p.stagger == 2): the MFMAs of slab s with, woven into their gaps in a fixed order,
  // the split + LDS store of slab s+1 (other buffer) and the loads of slab s+3 (same register set).  Every
  // piece is fenced with sched_barrier, so the instruction stream is exactly: fragment reads, then per MFMA one
  // small piece of staging.  An MFMA blocks the vector issue for 8 of its 32 cycles; the rest takes the piece.
  auto prep_load = [&](int k0) {
    const int tap = (int)fd_div((uint32_t)k0, p.fd_c);
    if (tap != cur_tap) {  // uniform
      cur_tap = tap;
      uint32_t kh, kw;
      fd_divmod((uint32_t)tap, p.fd_kw, kh, kw);
      const int dh = (int)kh * p.k_mul, dw = (int)kw * p.k_mul;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int64_t off;
        const bool ok = gather_elem_addr(j, dh, dw, off) && tap < ntaps;
        tap_voff[j] = ok ? (unsigned)off * (unsigned)EB + 16u * kc : OOB;
      }
    }
  };
  auto fused_step = [&](int k0, auto SET, int sbuf, int cbuf) {
    constexpr int S = decltype(SET)::value;
    constexpr int NM = 12 * TM * TN;
    const char* a_lane_b = a_lane + cbuf * BUFSZ;
    const char* b_lane_b = b_lane + cbuf * BUFSZ;
    // fragment reads in the order the MFMAs consume them (a3,b1 | a1,b3 | a2,b2 of the first tile, ...): LDS
    // returns in order, so the first MFMA waits for two reads, not for all eighteen
    constexpr int FK = MF16 ? 1 : 2, FM = MF16 ? TMB : TM, FN = MF16 ? TNB : TN, FR = MF16 ? 16 : 32;
    bf16x8_t af[FK][FM][3], bf[FK][FN][3];
#pragma unroll
    for (int ks = 0; ks < FK; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          constexpr int APL[3] = {2, 0, 1}, BPL[3] = {0, 2, 1};
          const int ko = MF16 ? koff16 : koff[ks];
          af[ks][i][APL[u]] = *reinterpret_cast<const bf16x8_t*>(a_lane_b + (APL[u] * BM + FR * i) * XPITCH + ko);
          if (i == 0) {
#pragma unroll
            for (int j = 0; j < FN; ++j)
              bf[ks][j][BPL[u]] = *reinterpret_cast<const bf16x8_t*>(b_lane_b + (BPL[u] * BN + FR * j) * XPITCH + ko);
          }
          if (ks == 0 && i == 0) __builtin_amdgcn_sched_barrier(0);  // keep the first tile's operands first in the queue
        }
    __builtin_amdgcn_sched_barrier(0);
    const int soff_a = (k0 - cur_tap * p.C) * EB, soff_b = (k0 >> 5) * p.Npad * 64;
    const bool kvalid = !ktail || (k0 + CH * kc < p.K);
    unsigned hs[NA][2], ms[NA][2], ls[NA][2];
    constexpr int P_SPLIT = 2 * NA;            // pieces: 2*NA splits, NA A-writes, NB B-writes, NA + NB loads
    constexpr int P_AW = P_SPLIT + NA, P_BW = P_AW + NB, P_LA = P_BW + NA, P_LB = P_LA + NB;
    auto piece = [&](int w) {
      if (w < P_SPLIT) {
        const int j = w >> 1, hf = w & 1;
        const f32x4 f = __builtin_bit_cast(f32x4, ra[S][j]);
        split3_pair(f[2 * hf], f[2 * hf + 1], hs[j][hf], ms[j][hf], ls[j][hf]);
      } else if (w < P_AW) {
        const int j = w - P_SPLIT;
        const int arow = r0 + RS * j;
        char* dst = Ap + sbuf * BUFSZ + arow * XPITCH + (((kc >> 1) ^ xswz(arow)) << 4) + (kc & 1) * 8;
        *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){hs[j][0], hs[j][1]};
        *reinterpret_cast<u32x2_t*>(dst + BM * XPITCH) = (u32x2_t){ms[j][0], ms[j][1]};
        *reinterpret_cast<u32x2_t*>(dst + 2 * BM * XPITCH) = (u32x2_t){ls[j][0], ls[j][1]};
      } else if (w < P_BW) {
        const int i = w - P_AW, idx = t + NT * i;
        if ((NBC % NT == 0) || idx < NBC) {
          const int pl = idx / (BN * 4), rem = idx - pl * (BN * 4);
          const int row = rem >> 2, c = rem & 3;
          *reinterpret_cast<u32x4_t*>(Bp + sbuf * BUFSZ + (pl * BN + row) * XPITCH + ((c ^ xswz(row)) << 4)) = rb[S][i];
        }
      } else if (w < P_LA) {
        const int j = w - P_BW;
        ra[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(kvalid ? tap_voff[j] : OOB), soff_a, 0);
      } else if (w < P_LB) {
        const int i = w - P_LA;
        rb[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)b_voff[i], soff_b, 0);
      }
    };
    if constexpr (MF16) {
      // 24 TM TN instructions of 16 cycles: one piece behind every SECOND one keeps the piece density per matrix-pipe cycle
      constexpr int NM16 = 6 * TMB * TNB;
#pragma unroll
      for (int q = 0; q < NM16; ++q) {
        const int tile = q / 6, term = q % 6;
        const int i = tile / TNB, j = tile % TNB;
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][i][PA_[term]], bf[0][j][PB_[term]], acc[i][j], 0, 0, 0);
        if (q & 1) piece(q >> 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int w = NM16 / 2; w < P_LB; ++w) piece(w);
    } else {
#pragma unroll
    for (int q = 0; q < NM; ++q) {
      const int ks = q / (6 * TM * TN), r = q % (6 * TM * TN), tile = r / 6, term = r % 6;
      const int i = tile / TN, j = tile % TN;
      constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][PA_[term]], bf[ks][j][PB_[term]], acc[i][j], 0, 0, 0);
      piece(q);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int w = NM; w < P_LB; ++w) piece(w);  // narrow tiles have more pieces than MFMAs
    }
  };

  // The same step for ONE plane (NPL == 1): 2 * TM * TN MFMAs per slab; the pieces woven between them are the LDS
  // stores of slab s+1 (a conversion first when the storage is fp32) and the loads of slab s+3.
  auto fused_step1 = [&](int k0, auto SET, int sbuf, int cbuf) {
    constexpr int S = decltype(SET)::value;
    constexpr int NM = 2 * TM * TN;
    const char* a_lane_b = a_lane + cbuf * BUFSZ;
    const char* b_lane_b = b_lane + cbuf * BUFSZ;
    bf16x8_t af[2][TM], bf[2][TN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TM; ++i) af[ks][i] = *reinterpret_cast<const bf16x8_t*>(a_lane_b + (32 * i) * XPITCH + koff[ks]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[ks][j] = *reinterpret_cast<const bf16x8_t*>(b_lane_b + (32 * j) * XPITCH + koff[ks]);
      if (ks == 0) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int soff_a = (k0 - cur_tap * p.C) * EB, soff_b = (k0 >> 5) * p.Npad * 64;
    const bool kvalid = !ktail || (k0 + CH * kc < p.K);
    constexpr int P_AW = NA, P_BW = P_AW + NB, P_LA = P_BW + NA, P_LB = P_LA + NB;
    auto piece = [&](int w) {
      if (w < P_AW) {
        store_A1(ra[S][w], w, sbuf);
      } else if (w < P_BW) {
        const int i = w - P_AW, idx = t + NT * i;
        if ((NBC % NT == 0) || idx < NBC) {
          const int row = idx >> 2, c = idx & 3;
          *reinterpret_cast<u32x4_t*>(Bp + sbuf * BUFSZ + row * XPITCH + ((c ^ xswz(row)) << 4)) = rb[S][i];
        }
      } else if (w < P_LA) {
        const int j = w - P_BW;
        ra[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(kvalid ? tap_voff[j] : OOB), soff_a, 0);
      } else if (w < P_LB) {
        const int i = w - P_LA;
        rb[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)b_voff[i], soff_b, 0);
      }
    };
#pragma unroll
    for (int q = 0; q < NM; ++q) {
      const int ks = q / (TM * TN), tile = q % (TM * TN);
      const int i = tile / TN, j = tile % TN;
      if constexpr (!MF16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
      if (q < P_LB) piece(q);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int w = NM; w < P_LB; ++w) piece(w);
  };

  // ---- slab stream (padding-tap elimination and channel-block order as in igemm_conv_kernel) ---------------
  const int spt = p.C / BK;
  int nslab = (p.K + BK - 1) / BK;
  bool use_map = false;
  if (p.skip_taps && (ntaps > 1 || p.perm2)) {  // uniform
    int nact = 0;
    for (int tap = 0; tap < ntaps; ++tap) {
      uint32_t kh, kw;
      fd_divmod((uint32_t)tap, p.fd_kw, kh, kw);
      const int dh = (int)kh * p.k_mul, dw = (int)kw * p.k_mul;
      bool any = false;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        int64_t off;
        any = any || gather_elem_addr(j, dh, dw, off);
      }
      if (__syncthreads_or(any ? 1 : 0)) {
        if (t == 0) tapinfo[nact] = tap;
        ++nact;
      }
    }
    __syncthreads();
    if (ntaps > 1) {
      nslab = __builtin_amdgcn_readfirstlane(nact) * spt;
      use_map = true;
    } else if (__builtin_amdgcn_readfirstlane(nact) == 0) {
      nslab = 0;  // a 1x1 stride-2 dgrad tile of a parity class no tap reaches: bias / zeros only
    }
  }
  const int last = nslab - 1;
  const bool it_lin = ntaps <= 1;
  const int it_ntap = use_map ? nslab / (spt > 0 ? spt : 1) : ntaps;
  const int it_run = p.cb > 0 ? p.cb : spt;
  int it_ci = 0, it_ti = 0, it_cb = 0, it_s = 0;
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

  if (nslab > 0) {
    if constexpr (PF == 1) {
      // p.ablate (timing-only diagnostics, results wrong): 1 = no global loads in the loop, 2 = no LDS store and
      // no barriers, 4 = no MFMA phase
      const bool do_ld = !(p.ablate & 1), do_st = !(p.ablate & 2), do_mm = !(p.ablate & 4);
      load_AB(next_k0(), IC<0>{});
      for (int s = 0; s < nslab; ++s) {
        if (do_st) {
          __syncthreads();  // every wave has finished reading the previous slab
          store_AB(IC<0>{}, 0);
          __syncthreads();
        }
        const int kn = next_k0();
        if (do_ld) load_AB(kn, IC<0>{});  // the tail reloads the last slab (unused): no branch
        if (do_mm) compute(0);
      }
    } else {
      const bool late = __builtin_amdgcn_readfirstlane(t >> 6) >= (NT / 128);
      // p.ablate (timing-only diagnostics, results wrong): 1 = no global loads, 2 = no split + LDS store, 4 = no MFMA
      // phase, 8 = no stagger (every wave stores after its MFMAs)
      const bool do_ld = !(p.ablate & 1), do_st = !(p.ablate & 2), do_mm = !(p.ablate & 4);
      const bool early = !late && !(p.ablate & 8);
      load_AB(next_k0(), IC<0>{});
      load_AB(next_k0(), IC<1>{});
      store_AB(IC<0>{}, 0);
      load_AB(next_k0(), IC<0>{});  // slab 2 (the iterator repeats the last slab past the end: stored, never read)
      __syncthreads();
      if (p.stagger == 2) {
        for (int s = 0; s < nslab; s += 2) {
          const int ka = next_k0();
          prep_load(ka);
          if constexpr (NPL == 3) fused_step(ka, IC<1>{}, 1, 0);
          else fused_step1(ka, IC<1>{}, 1, 0);
          __syncthreads();
          if (s + 1 >= nslab) break;
          const int kb = next_k0();
          prep_load(kb);
          if constexpr (NPL == 3) fused_step(kb, IC<0>{}, 0, 1);
          else fused_step1(kb, IC<0>{}, 0, 1);
          __syncthreads();
        }
      } else
      for (int s = 0; s < nslab; s += 2) {
        // slab s from buffer 0; slab s+1 (set 1) goes to buffer 1, set 1 then fetches slab s+3
        const int ka = next_k0();
        if (early) { if (do_st) store_AB(IC<1>{}, 1); if (do_ld) load_AB(ka, IC<1>{}); }
        if (do_mm) compute(0);
        if (!early) { if (do_st) store_AB(IC<1>{}, 1); if (do_ld) load_AB(ka, IC<1>{}); }
        __syncthreads();
        if (s + 1 >= nslab) break;
        // slab s+1 from buffer 1; slab s+2 (set 0) goes to buffer 0, set 0 then fetches slab s+4
        const int kb = next_k0();
        if (early) { if (do_st) store_AB(IC<0>{}, 0); if (do_ld) load_AB(kb, IC<0>{}); }
        if (do_mm) compute(1);
        if (!early) { if (do_st) store_AB(IC<0>{}, 0); if (do_ld) load_AB(kb, IC<0>{}); }
        __syncthreads();
      }
    }
  }

  // ---- epilogue ----------------------------------------------------------------------------------------------
  const bool has_bias = (p.flags & SG_EPI_BIAS) != 0;
  const bool do_relu = (p.flags & SG_EPI_RELU) != 0;
#pragma unroll
  for (int j = 0; j < TNB; ++j) {
    const int col = n0 + wn + acc_col(j);
    const bool cv = col < p.Nout;
    const float bv = (has_bias && cv) ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TMB; ++i) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int row = m0 + wm + acc_row(i, r);
        if (cv && row < p.M) {
          float v = acc[i][j][r] + bv;
          if (do_relu) v = fmaxf(v, 0.f);
          const int64_t yo = row_to_yoff(p, row) + col;
          if (p.res) v += ld1<TA>(reinterpret_cast<const TA*>(p.res) + yo);   // uniform: a collected gradient rides along
          st1<TA>(reinterpret_cast<TA*>(p.y) + yo, v);
        }
      }
    }
  }

  // ---- BatchNormalization statistics of this tile (SG_EPI_BN_STATS): per output channel the sum and the centred sum of
  // squares over the tile's valid rows, taken from the accumulators before they leave the registers; two-pass within
  // the tile (mean first, then sum (v - mean)^2), combined over the tiles in fp64 by bn_tiles_finalize_kernel.
  // Layout: stats[tile_m][2][Nout].  Saves BatchNormalization its own statistics pass over y.
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem);  // [WGM][BN] partials, then [BN] tile means
    float* tmean = red + WGM * BN;
    const int wrow = wave / WGN;
    const int nvalid = (p.M - m0) < BM ? (p.M - m0) : BM;
    float vals[TNB][TMB][NR];
#pragma unroll
    for (int j = 0; j < TNB; ++j) {
      const int col = n0 + wn + acc_col(j);
      const float bv = (has_bias && col < p.Nout) ? p.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TMB; ++i)
#pragma unroll
        for (int r = 0; r < NR; ++r) vals[j][i][r] = acc[i][j][r] + bv;
    }
    __syncthreads();  // every wave is done with the slab buffers
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int j = 0; j < TNB; ++j) {
        const int cl = wn + acc_col(j);
        const float mu = pass ? tmean[cl] : 0.f;
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < TMB; ++i)
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const int row = m0 + wm + acc_row(i, r);
            const float dlt = vals[j][i][r] - mu;
            if (row < p.M) sacc += pass ? dlt * dlt : dlt;
          }
        sacc += __shfl_xor(sacc, 32, 64);  // lanes l and l+32 hold the same column (16x16 blocks: l, l+16, l+32, l+48)
        if constexpr (MF16) sacc += __shfl_xor(sacc, 16, 64);
        if (lh == 0) red[wrow * BN + cl] = sacc;
      }
      __syncthreads();
      for (int cl = t; cl < BN; cl += NT) {
        float tot = 0.f;
#pragma unroll
        for (int wq = 0; wq < WGM; ++wq) tot += red[wq * BN + cl];
        const int col = n0 + cl;
        if (pass == 0) tmean[cl] = tot / (float)nvalid;
        if (col < p.Nout) p.stats[((int64_t)tile_m * 2 + pass) * p.Nout + col] = tot;
      }
      __syncthreads();
    }
  }
}

// mean / variance over all rows from the per-tile (sum, centred sum of squares) pairs: a segmented column reduction
// over the T tiles (rows of the reducer = tiles), pivoted on tile 0's mean so nothing cancels, combined in fp64, then
// the BatchNormalization bookkeeping of BnStatsOp::finalize.
struct BnTilesOp {
  static constexpr int NOUT = 2;
  const float* __restrict__ stats;  // [T][2][C]
  int C;
  int64_t rows;                     // pixels (not tiles)
  float* moving_mean;
  float* moving_var;
  float* save_mean;
  float* save_invstd;
  float momentum, eps;
  int unbiased;

  template <int V>
  __device__ __forceinline__ void accum(int, int64_t r, int c, float (&acc)[2][V]) const {
    float s1[V], m2[V], s0[V];
    ldv<V>(stats + (2 * r) * C + c, s1);
    ldv<V>(stats + (2 * r + 1) * C + c, m2);
    ldv<V>(stats + c, s0);
    const int64_t left = rows - r * BM;
    const float nt = (float)(left < BM ? left : BM), n0 = (float)(rows < BM ? rows : BM);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float d = s1[i] / nt - s0[i] / n0;  // tile mean minus the pivot (tile 0's mean)
      acc[0][i] = fmaf(nt, d, acc[0][i]);
      acc[1][i] += fmaf(nt * d, d, m2[i]);
    }
  }
  __device__ __forceinline__ void finalize(int, int c, const double (&s)[2]) const {
    const double n = (double)rows, n0 = (double)(rows < BM ? rows : BM);
    const double m1 = s[0] / n;
    const double mean = (double)stats[c] / n0 + m1;
    double var = s[1] / n - m1 * m1;
    if (var < 0.0) var = 0.0;
    save_mean[c] = (float)mean;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    const double var_u = (unbiased && rows > 1) ? var * (n / (n - 1.0)) : var;
    moving_mean[c] = (float)((double)moving_mean[c] * momentum + mean * (1.0 - (double)momentum));
    moving_var[c] = (float)((double)moving_var[c] * momentum + var_u * (1.0 - (double)momentum));
  }
};

template <int BN, int WGM, int WGN, int PF, int NPL = 3, typename TA = float, int MF = 0>
int launch_x6(const IgemmParams& p, hipStream_t st) {
  constexpr int NT = 64 * WGM * WGN;
  // the statistics epilogue reuses the front of LDS for [WGM][BN] + [BN] floats: keep at least that much
  constexpr size_t slab_lds = (size_t)PF * NPL * (BM + BN) * XPITCH;
  constexpr size_t stat_lds = (size_t)(WGM + 1) * BN * sizeof(float);
  constexpr size_t lds = (slab_lds > stat_lds ? slab_lds : stat_lds) + 256 + (size_t)(BM * BK / 4) * sizeof(int);
  (void)NT;
  static bool attr_done = false;
  if (!attr_done) {
    int rc = set_dyn_lds(conv_x6_kernel<BN, WGM, WGN, PF, NPL, TA, MF>, lds);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.M, BM) * sg_cdiv(p.Nout, BN);
  if (tiles <= 0 || tiles > 0x7fffffff) {
    sg_set_error("conv_x6: bad tile count %lld", (long long)tiles);
    return SG_EINVAL;
  }
  hipLaunchKernelGGL((conv_x6_kernel<BN, WGM, WGN, PF, NPL, TA, MF>), dim3((unsigned)tiles), dim3(64 * WGM * WGN), lds, st, p);
  SG_LAUNCH_CHECK("conv_x6_kernel");
  return 0;
}

// Variant switch for A/B runs: SG_X6_VARIANT bit 0 = the one-workgroup-per-CU structure (PF == 2 above), bit 1 =
// 4-wave workgroups (64x64 sub-tile per wave: half the LDS fragment traffic per MFMA).
inline int x6_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("SG_X6_VARIANT");
    v = e ? atoi(e) & 3 : -1;  // -1: chosen per launch (dispatch_x6)
  }
  return v;
}

// Convolution arithmetic on fp32 storage (sg_set_conv_x6 / SG_CONV_X6): 0 = native fp32 MFMA, 1 = the exact
// six-pass emulation (default), 2 = bf16 products, one pass (fp32 tensors rounded to bf16 on the way into LDS, fp32
// accumulation: the arithmetic of the SG_BF16 path on fp32 tensors).
int g_x6_enabled = -1;  // -1: not yet read from the environment; set by sg_set_conv_x6()
inline int x6_mode() {
  if (g_x6_enabled < 0) {
    const char* e = getenv("SG_CONV_X6");
    g_x6_enabled = e ? atoi(e) : 1;
    if (g_x6_enabled < 0 || g_x6_enabled > 2) g_x6_enabled = 1;
  }
  return g_x6_enabled;
}
inline bool x6_enabled() { return x6_mode() != 0; }
inline bool x6_vpad_on() {
  static int v = -1;
  if (v < 0) v = getenv("SG_X6_VPAD") ? atoi(getenv("SG_X6_VPAD")) : 1;
  return v != 0;
}

// Can this launch take the bf16-pipe kernels?  UT gather: every 32-deep slab inside one tap - Cin % 32 == 0, a 1x1
// kernel, or (virtual channel padding, run_x6) any channel count that keeps the pixel rows 16-byte aligned; `vec` =
// 16-byte channel runs (4 fp32 / 8 bf16).  force: SG_BF16 storage has no other fast kernel, so the mode switch is not asked.
inline bool x6_ok(const IgemmParams& p, bool vec, bool force = false) {
  const bool ut = (p.C % BK == 0) || (p.K == p.C) || x6_vpad_on();
  const int64_t kmax = (int64_t)(p.K / p.C) * x6_vpad_c(p.C);
  return (force || x6_enabled()) && vec && ut && p.x_bytes != 0 && x6_planes_bytes((int)kmax, p.Nout) < (1ull << 31) &&
         p.Nout >= 16;
}

// ---- wgrad on the bf16 pipe --------------------------------------------------------------------------------------
// dw[r][co] = sum_p x[p + tap shift][ci] * dy[p][co]: the reduction index is the PIXEL, while both operands are
// channel-contiguous in memory.  The slabs are therefore staged as they lie - [32 pixels][channels] per plane,
// 8-byte writes - and the fragments (8 consecutive pixels of one channel per lane) are taken with the transposed
// LDS read ds_read_b64_tr_b16: per 16-lane group a 4-pixel x 16-channel block, lane i receiving channel i of
// the four pixels (mapping verified by scripts/exp_trread.hip).  Row pitch = channel bytes + 64 (an odd
// multiple of 64 B modulo 256), which spreads the four pixel rows of a block over all 64 banks.
// Loads follow the aligned-slab scheme of igemm_wgrad_kernel<FAST = 2>: constant per-thread voffsets (which here
// also carry the thread's tap shift, so a 128-row tile may span several taps: Cin = 32, 64), the slab's position
// and image row on the scalar unit.  Requires OW % 32 == 0, stride 1 and "same" geometry.
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_ptr;

__device__ __forceinline__ bf16x8_t tr_frag(const char* lds_addr_a, const char* lds_addr_b) {
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds_addr_a));
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds_addr_b));
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, v);
}

template <int BN>
struct X6WPitch {
  static constexpr int A = 2 * BM + 64;                       // 320
  static constexpr int B = (BN == 128) ? 2 * BN + 64 : 192;   // 320 / 192 / 192
};

// PF == 1: single LDS buffer, two barriers per slab, two workgroups per CU.  PF == 2: one workgroup per CU, LDS
// double-buffered, two register sets and the hand-interleaved step of conv_x6_kernel (the staging of slab s+1
// and the loads of slab s+3 woven into the MFMA gaps of slab s).
// NPL / TA as in conv_x6_kernel: NPL = 1 is the bf16 product (one plane, one MFMA per k-step and tile); with TA = bf16_t
// both operands are bf16 in HBM and a thread's 16-byte chunk (8 channels) goes to LDS unchanged.
constexpr int WG_NS = 1;  // register sets of the bf16 wgrad: 2 measured no faster (728 -> 728: 55 us either way,
                          // profiles/r02_b16_prefetch_ab.txt), 4 spills at BN = 128
// PIN ("planes in", round 5): the fp32 (six-pass) filter gradient whose operands are ALREADY split - x and dy arrive as three
// bf16 planes [3][pixels][channels] each (x6w_split_kernel's layout: the forward's activation planes, kept; dy's made once per
// layer) and a thread's 16-byte chunk of every plane goes from the buffer load to LDS untouched, as in the bf16-storage form.
// The fp32 form splits 16 chunks per thread and slab on the VALU (two split3_pair each, ~200 instructions) for 24 MFMAs per
// wave, between two barriers: as long as the MFMA phase itself.  Same LDS image, same fragments, same products in the same order
// as the fp32 form: bit-identical results.
template <int BN, int WGM, int WGN, int PF, int NPL = 3, typename TA = float, bool PIN = false>
__global__ __launch_bounds__(64 * WGM * WGN, PF == 2 ? (WGM * WGN + 3) / 4 : WGM * WGN / 2) void wgrad_x6_kernel(const WgradParams p) {
  static_assert(NPL == 3 || NPL == 1, "planes");
  static_assert(NPL == 1 || std::is_same<TA, float>::value || PIN, "the three-plane split is the fp32 path");
  static_assert(!PIN || (NPL == 3 && !std::is_same<TA, float>::value), "planes in: three bf16 planes per operand");
  constexpr bool A16 = !std::is_same<TA, float>::value;
  constexpr int NLD = PIN ? 3 : 1;        // 16-byte loads per chunk position (one per plane)
  constexpr int EB = EL<TA>::BYTES, CH = EL<TA>::CH;
  constexpr int CPA = BM / CH;            // 16-byte chunks per A' pixel row: 32 (fp32) or 16 (bf16)
  constexpr int CPB = BN / CH;
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int NA = (BK * CPA) / NT;     // 16-byte A' chunks per thread (pixel rows pr0 + PS*j)
  constexpr int PS = NT / CPA;
  constexpr int NBC = BK * CPB;
  constexpr int NB = (NBC + NT - 1) / NT;
  constexpr int PA = X6WPitch<BN>::A, PB = X6WPitch<BN>::B;
  static_assert(NA >= 1 && TM >= 1 && TN >= 1, "tile too small for the wave layout");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BUFSZ = NPL * BK * (PA + PB);  // one LDS buffer: A' planes then B planes
  char* Ap = smem;                     // [PF buffers] x { A' [NPL][BK][PA], B [NPL][BK][PB] }
  char* Bp = Ap + NPL * BK * PA;
  int* slist = reinterpret_cast<int*>(smem + PF * BUFSZ);  // [1 + 1024] + flags[1024]

  const int t = threadIdx.x;
  const uint32_t ntn = (p.Cout + BN - 1) / BN;
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

  // This thread's four r-rows (one float4 of channels) lie in ONE tap (Cin % 4 == 0); the tile's 128 rows may
  // span several taps (Cin < 128), so the tap offset (dh, dw) is per thread and rides in the voffset.
  const int rc = t % CPA, pr0 = t / CPA;
  const int r_first = rbase + CH * rc;
  const bool rvalid = r_first < p.K;
  const int tap = (rvalid ? r_first : rbase) / p.Cin;
  int s_dh, s_dw;
  {
    uint32_t kh, kw;
    fd_divmod((uint32_t)tap, p.fd_kw, kh, kw);
    s_dh = (int)kh * p.dil - p.pad_t;
    s_dw = (int)kw * p.dil - p.pad_l;
  }
  const int ci0 = r_first - tap * p.Cin;

  const int slab_begin = (int)split * p.slabs_per_split;
  const int nslab_total = (p.P + BK - 1) / BK;
  int slab_end = slab_begin + p.slabs_per_split;
  if (slab_end > nslab_total) slab_end = nslab_total;

  constexpr unsigned OOB = 0x80000000u;
  // the descriptor's base sits SH pixels before x (SH = the most negative tap shift), so that voffset =
  // (pixel-in-slab + tap shift + SH) * x_ld + channel is non-negative and the slab's position is the scalar offset
  const int SH = p.pad_t * p.W + p.pad_l;
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<char*>(const_cast<float*>(p.x)) - (int64_t)SH * p.x_ld * EB, 0,
      (int)(p.x_bytes + (uint32_t)(SH * p.x_ld * EB)), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
  unsigned a_voffc[NA], b_voff[NB];
#pragma unroll
  for (int j = 0; j < NA; ++j)  // output pixel pr0 + PS j of the slab reads input pixel stride * (pr0 + PS j) + tap shift
    a_voffc[j] = rvalid ? (unsigned)((p.stride * (pr0 + PS * j) + s_dh * p.W + s_dw + SH) * p.x_ld + ci0) * (unsigned)EB : OOB;
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = t + NT * i;
    const int kr = idx / CPB, c4 = idx % CPB;
    b_voff[i] = (idx < NBC && (n0 + CH * c4) < p.Cout) ? (unsigned)(kr * p.y_ld + n0 + CH * c4) * (unsigned)EB : OOB;
  }

  // NS register sets of prefetched slabs.  The bf16-storage form has 4 MFMAs per wave and slab against a memory latency of
  // several slab times: it keeps WG_NS slabs of loads in flight (a set is two 16-byte registers there); the six-pass form
  // (24 MFMAs per slab, 24 registers per set) keeps one, or two with the double-buffered structure.
  constexpr int NS = (PF == 1 && NPL == 1 && A16) ? WG_NS : PF;
  u32x4_t ra[NS][NA * NLD], rb[NS][NB * NLD];
  auto load_AB = [&](int p0, auto SET) {
    constexpr int S = decltype(SET)::value;
    uint32_t q, ow0, n_, oh;
    fd_divmod((uint32_t)p0, p.fd_ow, q, ow0);
    fd_divmod(q, p.fd_oh, n_, oh);
    // the slab's 32 output pixels (one output row, OW % 32 == 0) read input row oh * stride + s_dh, columns
    // (ow0 + k) * stride + s_dw: the slab's first input pixel is the scalar offset, the rest rides in the voffsets
    const int ih = (int)oh * p.stride + s_dh;
    const bool row_ok = (unsigned)ih < (unsigned)p.H;
    const int soff_a = (((int)n_ * p.H + (int)oh * p.stride) * p.W + (int)ow0 * p.stride) * p.x_ld * EB;
    const int col0 = ((int)ow0 + pr0) * p.stride + s_dw;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const bool v = row_ok && ((unsigned)(col0 + PS * j * p.stride) < (unsigned)p.W);
#pragma unroll
      for (int pl = 0; pl < NLD; ++pl)   // PIN: the same chunk of every plane, the plane in the scalar offset
        ra[S][j * NLD + pl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(v ? a_voffc[j] : OOB), soff_a + pl * (int)p.x_plane_bytes, 0);
    }
    const int soff_b = p0 * p.y_ld * EB;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int pl = 0; pl < NLD; ++pl)
        rb[S][i * NLD + pl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)b_voff[i], soff_b + pl * (int)p.dy_plane_bytes, 0);
  };

  // one 16-byte chunk (4 fp32 or 8 bf16 channels of one pixel) to its place in the [pixel][channel] LDS image
  auto store_chunk = [&](const u32x4_t v, char* plane0, int pixel, int chunk, int pitch) {
    if constexpr (A16) {
      *reinterpret_cast<u32x4_t*>(plane0 + pixel * pitch + chunk * 16) = v;
    } else {
      const f32x4 f = __builtin_bit_cast(f32x4, v);
      char* dst = plane0 + pixel * pitch + chunk * 8;
      if constexpr (NPL == 3) {
        unsigned h0, m0_, l0, h1, m1, l1;
        split3_pair(f[0], f[1], h0, m0_, l0);
        split3_pair(f[2], f[3], h1, m1, l1);
        *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){h0, h1};
        *reinterpret_cast<u32x2_t*>(dst + BK * pitch) = (u32x2_t){m0_, m1};
        *reinterpret_cast<u32x2_t*>(dst + 2 * BK * pitch) = (u32x2_t){l0, l1};
      } else {
        *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3])};
      }
    }
  };
  auto store_AB = [&](auto SET, int buf) {
    constexpr int S = decltype(SET)::value;
    if constexpr (PIN) {   // plane pl of a chunk -> plane pl of the LDS image, unchanged
#pragma unroll
      for (int j = 0; j < NA; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          *reinterpret_cast<u32x4_t*>(Ap + buf * BUFSZ + pl * BK * PA + (pr0 + PS * j) * PA + rc * 16) = ra[S][j * 3 + pl];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int idx = t + NT * i;
        if (idx < NBC) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<u32x4_t*>(Bp + buf * BUFSZ + pl * BK * PB + (idx / CPB) * PB + (idx % CPB) * 16) = rb[S][i * 3 + pl];
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) store_chunk(ra[S][j], Ap + buf * BUFSZ, pr0 + PS * j, rc, PA);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + NT * i;
      if (idx < NBC) store_chunk(rb[S][i], Bp + buf * BUFSZ, idx / CPB, idx % CPB, PB);
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
  // transposed-read addresses: lane = 16 g + i supplies pixel row 8 (g>>1) + (i>>2) [+4 for the second read],
  // channels 16 (g&1) + 4 (i&3) .. +3 of its 32-channel tile
  const int tg = lane >> 4, ti = lane & 15;
  const int tr_row = 8 * (tg >> 1) + (ti >> 2), tr_col = 16 * (tg & 1) + 4 * (ti & 3);
  const char* a_lane = Ap + tr_row * PA + (wm + tr_col) * 2;
  const char* b_lane = Bp + tr_row * PB + (wn + tr_col) * 2;

  // Round 5: a rolling fragment pipeline.  The slab's MFMAs go in 2 TM blocks (k-step, row sub-tile i: 6 TN MFMAs); while block q
  // multiplies, the A fragments of block q + 1 (and, at the k-step change, its B fragments) are already on their way from LDS.
  // Before, every k-step opened with ALL of its reads and a full wait: the read + MFMA loop alone took 646 of the 830 us of the
  // ASPP filter gradient (gpurun_out/r5g).  One A set and one B set more in registers (requesting the whole slab up front
  // needs 36 more and spills at this kernel's 128-register budget).  Same products in the same order: bit-identical.
  auto compute = [&](int buf) {
    bf16x8_t afr[2][NPL], bfr[2][TN][NPL];
    auto rd_a = [&](int ks, int i, bf16x8_t (&d)[NPL]) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const char* a = a_lane + buf * BUFSZ + (pl * BK + 16 * ks) * PA + 64 * i;
        d[pl] = tr_frag(a, a + 4 * PA);
      }
    };
    auto rd_b = [&](int ks, bf16x8_t (&d)[TN][NPL]) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          const char* b = b_lane + buf * BUFSZ + (pl * BK + 16 * ks) * PB + 64 * j;
          d[j][pl] = tr_frag(b, b + 4 * PB);
        }
    };
    const bool rd = !(PF == 1 && (p.stagger & 8));   // SG_X6_ABLATE bit 8 (timing only): no fragment reads, MFMAs on whatever the registers hold
    if (rd) {
      rd_b(0, bfr[0]);
      rd_a(0, 0, afr[0]);
    } else {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          asm volatile("" : "=v"(afr[u][pl]));
#pragma unroll
          for (int j = 0; j < TN; ++j) asm volatile("" : "=v"(bfr[u][j][pl]));
        }
      }
    }
    constexpr int NQ = 2 * TM;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ks = q / TM, i = q % TM;
      if (q + 1 < NQ && rd) {
        const int ks1 = (q + 1) / TM, i1 = (q + 1) % TM;
        rd_a(ks1, i1, afr[(q + 1) & 1]);
        if (ks1 != ks) rd_b(ks1, bfr[ks1 & 1]);
      }
      __builtin_amdgcn_sched_barrier(0);   // (left alone, the scheduler sinks the reads back to their first use)
      const bf16x8_t(&af)[NPL] = afr[q & 1];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const bf16x8_t(&bf)[NPL] = bfr[ks & 1][j];
        if constexpr (NPL == 1) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[i][j], 0, 0, 0);
          continue;
        }
        constexpr int P2 = NPL == 3 ? 2 : 0, P1 = NPL == 3 ? 1 : 0;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[P2], bf[0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[P2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[P1], bf[P1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[P1], bf[0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[P1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- slab stream with padding-slab elimination (as igemm_wgrad_kernel) ----------------------------------------
  int nslab = slab_end - slab_begin;
  bool use_list = false;
  if (p.skip_slabs && nslab <= 1024) {  // uniform
    use_list = true;
    int* flags = slist + 1 + 1024;
    for (int i = t; i < nslab; i += NT) {
      const int pa = (slab_begin + i) * BK;  // aligned slabs: one image row per slab
      uint32_t q, tmp, n_, oh;
      fd_divmod((uint32_t)pa, p.fd_ow, q, tmp);
      fd_divmod(q, p.fd_oh, n_, oh);
      const int tap_lo = rbase / p.Cin;
      int tap_hi = (rbase + BM - 1 < p.K ? rbase + BM - 1 : p.K - 1) / p.Cin;
      bool act = false;
      for (int tp = tap_lo; tp <= tap_hi && !act; ++tp) {
        uint32_t kh, kw;
        fd_divmod((uint32_t)tp, p.fd_kw, kh, kw);
        const int ddh = (int)kh * p.dil - p.pad_t, ddw = (int)kw * p.dil - p.pad_l;
        act = ((int)oh + ddh >= 0) && ((int)oh + ddh < p.H) && (ddw > -p.W) && (ddw < p.W);
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
  auto slab_of = [&](int i) -> int { return use_list ? slist[1 + i] : slab_begin + i; };
  const int lasti = nslab - 1;
  if (nslab > 0) {
    // p.stagger carries SG_X6_ABLATE here (timing-only diagnostics, results wrong): 1 = no global loads in the loop,
    // 2 = no split + LDS store + barriers, 4 = no fragment reads + MFMAs
    if constexpr (PF == 1 && NS > 1) {
      auto slab_at = [&](int i) -> int { return slab_of(i < lasti ? i : lasti) * BK; };
      auto step = [&](auto SET, int s) {
        __syncthreads();
        store_AB(SET, 0);
        __syncthreads();
        load_AB(slab_at(s + NS), SET);  // past the end: the last slab again (unused)
        compute(0);
      };
      load_AB(slab_at(0), IC<0>{});
      load_AB(slab_at(1), IC<1>{});
      if constexpr (NS == 4) {
        load_AB(slab_at(2), IC<2>{});
        load_AB(slab_at(3), IC<3>{});
      }
      for (int s = 0; s < nslab; s += NS) {
        step(IC<0>{}, s);
        if (s + 1 >= nslab) break;
        step(IC<1>{}, s + 1);
        if constexpr (NS == 4) {
          if (s + 2 >= nslab) break;
          step(IC<2>{}, s + 2);
          if (s + 3 >= nslab) break;
          step(IC<3>{}, s + 3);
        }
      }
    } else if constexpr (PF == 1) {
      const bool do_ld = !(p.stagger & 1), do_st = !(p.stagger & 2), do_mm = !(p.stagger & 4);
      load_AB(slab_of(0) * BK, IC<0>{});
      for (int s = 0; s < nslab; ++s) {
        if (do_st) {
          __syncthreads();
          store_AB(IC<0>{}, 0);
          __syncthreads();
        }
        const int pn = slab_of(s < lasti ? s + 1 : lasti) * BK;
        if (do_ld) load_AB(pn, IC<0>{});
        if (do_mm) compute(0);
      }
    } else if constexpr (NPL == 3) {
      // fused step: MFMAs of the slab in buffer cbuf with, one piece per MFMA, the split + store of register set S
      // into buffer sbuf and the loads of slab p0 into the same set
      auto fused_step = [&](int p0, auto SET, int sbuf, int cbuf) {
        constexpr int S = decltype(SET)::value;
        constexpr int NM = 12 * TM * TN;
        bf16x8_t af[2][TM][3], bf[2][TN][3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              constexpr int APL[3] = {2, 0, 1}, BPL[3] = {0, 2, 1};
              const char* a = a_lane + cbuf * BUFSZ + (APL[u] * BK + 16 * ks) * PA + 64 * i;
              af[ks][i][APL[u]] = tr_frag(a, a + 4 * PA);
              if (i == 0) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                  const char* b = b_lane + cbuf * BUFSZ + (BPL[u] * BK + 16 * ks) * PB + 64 * j;
                  bf[ks][j][BPL[u]] = tr_frag(b, b + 4 * PB);
                }
              }
              if (ks == 0 && i == 0) __builtin_amdgcn_sched_barrier(0);
            }
        __builtin_amdgcn_sched_barrier(0);
        uint32_t q_, ow0, n_, oh;
        fd_divmod((uint32_t)p0, p.fd_ow, q_, ow0);
        fd_divmod(q_, p.fd_oh, n_, oh);
        const bool row_ok = (unsigned)((int)oh + s_dh) < (unsigned)p.H;
        const int soff_a = p0 * p.x_ld * EB, soff_b = p0 * p.y_ld * EB;
        const int col0 = (int)ow0 + s_dw + pr0;
        unsigned ha[NA][2], ma[NA][2], la[NA][2], hb[NB][2], mb[NB][2], lb[NB][2];
        constexpr int P_SA = PIN ? 0 : 2 * NA, P_WA = P_SA + NA * NLD, P_SB = P_WA + (PIN ? 0 : 2 * NB), P_WB = P_SB + NB * NLD,
                      P_LA = P_WB + NA * NLD, P_LB = P_LA + NB * NLD;
        auto piece = [&](int w) {
          if constexpr (PIN) {
            // planes in: nothing to compute - 12 LDS stores of register set S (slab s + 1) into buffer sbuf, then the 12 loads of
            // slab p0 into the same set, one piece behind each of the slab's 24 MFMAs
            if (w < P_WA) {
              const int j = w / 3, pl = w % 3;
              *reinterpret_cast<u32x4_t*>(Ap + sbuf * BUFSZ + pl * BK * PA + (pr0 + PS * j) * PA + rc * 16) = ra[S][j * 3 + pl];
            } else if (w < P_WB) {
              const int i = (w - P_WA) / 3, pl = (w - P_WA) % 3, idx = t + NT * i;
              if ((NBC % NT == 0) || idx < NBC)
                *reinterpret_cast<u32x4_t*>(Bp + sbuf * BUFSZ + pl * BK * PB + (idx / CPB) * PB + (idx % CPB) * 16) = rb[S][i * 3 + pl];
            } else if (w < P_LA) {
              const int j = (w - P_WB) / 3, pl = (w - P_WB) % 3;
              const bool v = row_ok && ((unsigned)(col0 + PS * j) < (unsigned)p.W);
              ra[S][j * 3 + pl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(v ? a_voffc[j] : OOB), soff_a + pl * (int)p.x_plane_bytes, 0);
            } else if (w < P_LB) {
              const int i = (w - P_LA) / 3, pl = (w - P_LA) % 3;
              rb[S][i * 3 + pl] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)b_voff[i], soff_b + pl * (int)p.dy_plane_bytes, 0);
            }
            return;
          }
          if (w < P_SA) {
            const int j = w >> 1, hf = w & 1;
            const f32x4 f = __builtin_bit_cast(f32x4, ra[S][j]);
            split3_pair(f[2 * hf], f[2 * hf + 1], ha[j][hf], ma[j][hf], la[j][hf]);
          } else if (w < P_WA) {
            const int j = w - P_SA;
            char* dst = Ap + sbuf * BUFSZ + (pr0 + PS * j) * PA + rc * 8;
            *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){ha[j][0], ha[j][1]};
            *reinterpret_cast<u32x2_t*>(dst + BK * PA) = (u32x2_t){ma[j][0], ma[j][1]};
            *reinterpret_cast<u32x2_t*>(dst + 2 * BK * PA) = (u32x2_t){la[j][0], la[j][1]};
          } else if (w < P_SB) {
            const int i = (w - P_WA) >> 1, hf = (w - P_WA) & 1;
            const f32x4 f = __builtin_bit_cast(f32x4, rb[S][i]);
            split3_pair(f[2 * hf], f[2 * hf + 1], hb[i][hf], mb[i][hf], lb[i][hf]);
          } else if (w < P_WB) {
            const int i = w - P_SB, idx = t + NT * i;
            if ((NBC % NT == 0) || idx < NBC) {
              const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
              char* dst = Bp + sbuf * BUFSZ + kr * PB + c4 * 8;
              *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){hb[i][0], hb[i][1]};
              *reinterpret_cast<u32x2_t*>(dst + BK * PB) = (u32x2_t){mb[i][0], mb[i][1]};
              *reinterpret_cast<u32x2_t*>(dst + 2 * BK * PB) = (u32x2_t){lb[i][0], lb[i][1]};
            }
          } else if (w < P_LA) {
            const int j = w - P_WB;
            const bool v = row_ok && ((unsigned)(col0 + PS * j) < (unsigned)p.W);
            ra[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(v ? a_voffc[j] : OOB), soff_a, 0);
          } else if (w < P_LB) {
            const int i = w - P_LA;
            rb[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)b_voff[i], soff_b, 0);
          }
        };
#pragma unroll
        for (int q = 0; q < NM; ++q) {
          const int ks = q / (6 * TM * TN), r = q % (6 * TM * TN), tile = r / 6, term = r % 6;
          const int i = tile / TN, j = tile % TN;
          constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i][PA_[term]], bf[ks][j][PB_[term]], acc[i][j], 0, 0, 0);
          piece(q);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int w = NM; w < P_LB; ++w) piece(w);
      };
      auto slab_at = [&](int i) -> int { return slab_of(i < lasti ? i : lasti) * BK; };
      load_AB(slab_at(0), IC<0>{});
      load_AB(slab_at(1), IC<1>{});
      store_AB(IC<0>{}, 0);
      load_AB(slab_at(2), IC<0>{});
      __syncthreads();
      for (int s = 0; s < nslab; s += 2) {
        fused_step(slab_at(s + 3), IC<1>{}, 1, 0);
        __syncthreads();
        if (s + 1 >= nslab) break;
        fused_step(slab_at(s + 4), IC<0>{}, 0, 1);
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

template <int BN, int WGM, int WGN, int PF, int NPL = 3, typename TA = float, bool PIN = false>
int launch_wgrad_x6(const WgradParams& p, int S, hipStream_t st) {
  constexpr size_t lds = (size_t)PF * NPL * BK * (X6WPitch<BN>::A + X6WPitch<BN>::B) + (2 * 1024 + 4) * sizeof(int);
  static bool attr_done = false;
  if (!attr_done) {
    int rc = set_dyn_lds(wgrad_x6_kernel<BN, WGM, WGN, PF, NPL, TA, PIN>, lds);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.K, BM) * sg_cdiv(p.Cout, BN);
  hipLaunchKernelGGL((wgrad_x6_kernel<BN, WGM, WGN, PF, NPL, TA, PIN>), dim3((unsigned)tiles, 1, (unsigned)S), dim3(64 * WGM * WGN), lds, st, p);
  SG_LAUNCH_CHECK("wgrad_x6_kernel");
  return 0;
}

inline bool wgrad_x6_ok(const WgradParams& p, bool vec, bool force = false) {
  // stride 1: "same" geometry; stride 2 (round 2: the Xception shortcuts, strided stems, Conv2DTranspose gradients): the
  // output is the input subsampled, OH = ceil(H / 2) - the slab addressing above covers both; dilation 1 only at stride 2
  const bool geom = (p.stride == 1 && p.OH == p.H && p.OW == p.W) ||
                    (p.stride == 2 && p.dil == 1 && p.OH == (p.H + 1) / 2 && p.OW == (p.W + 1) / 2);
  const bool fast = geom && p.x_bytes != 0 && p.dy_bytes != 0;
  const int64_t sh_bytes = ((int64_t)p.pad_t * p.W + p.pad_l + 64 * p.stride) * p.x_ld * 4;
  return (force || x6_enabled()) && vec && fast && (p.OW % BK == 0) && p.Cout >= 16 &&
         ((int64_t)p.x_bytes + 2 * sh_bytes < (1ll << 31));
}
