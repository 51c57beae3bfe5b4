// "Wide" pointwise GEMM on the bf16 matrix pipe: included by conv_igemm.hip after conv_x6.h (same namespace, IgemmParams).
//
//   Y[M][N] = A[M][K] . B[N][K]^T (+ bias, ReLU, BatchNormalization statistics)      1x1 convolution, stride 1, and its dgrad
//
// for the 728-wide middle flow of the Xception encoder (train_model/DeepLabv3plus.py:323-416: 16 x 3 SeparableConv2D(728),
// M = 16 x 32 x 32 pixels): 50 layers x (forward + dgrad) = 100 of the ~330 GEMM launches of a training step and its
// largest single share of time.  conv_x6_kernel covers that shape with 768 tiles of 128 x 128 in three rounds of one
// workgroup per CU: every A element is gathered, split into its three bf16 planes and staged SIX times (once per column
// tile), a wave's 64 x 32 sub-tile reads 0.75 LDS fragments per MFMA, and each round pays its own prologue and epilogue -
// 140 TFLOP/s, a third of the bf16 pipe / 6.  Here ONE workgroup per CU owns a 128 x 384 tile (256 tiles = one round):
//   * 8 waves as 2 (rows) x 4 (columns), 64 x 96 per wave: 15 fragment reads for 36 MFMAs per 16-deep k-step (0.42 / MFMA);
//   * A is staged twice, not six times: a thread loads 4 fp32 per k-step, splits them (split3_pair) and writes 3 x 8 bytes;
//   * the weight planes - already bf16, k-contiguous, prepared once per step - go from L2 to LDS by LDS-DMA
//     (buffer_load ... lds, 16 bytes per lane), no registers, no ds_write; the XOR swizzle of the LDS image is applied to
//     the per-lane SOURCE address (the DMA writes lane-linearly);
//   * two 48 KB LDS stages, ONE barrier per k-step: while stage s is multiplied, the A planes of k-step s+1 are written
//     and the DMA of its B planes is in flight; the wait before the barrier is counted (vmcnt(1): the fp32 load of k-step
//     s+2 stays in flight across the barrier), raw s_barrier (a __syncthreads would drain the queue).
// NPL = 1, TA = bf16_t (SG_BF16 storage): the same tile with 64-deep stages, BOTH operands by LDS-DMA, one MFMA per
// fragment pair, the bf16 tile leaves through LDS as 16-byte row chunks.
// Rounding order: a pixel's dot product is accumulated k-step by k-step in one fp32 accumulator whatever the batch:
// batch-slice invariance and run-to-run determinism hold as for conv_x6_kernel (tests/test_fullsize_gpu.py).
#pragma once

constexpr int PW_BM = 128, PW_BN = 384;

template <int NPL, int BN = 384>
struct PwGeom {
  static constexpr int KS = NPL == 3 ? 1 : 4;      // 16-deep k-steps per stage
  static constexpr int BKW = 16 * KS;              // reduction depth of a stage
  static constexpr int RB = 2 * BKW;               // bytes of one LDS row (bf16)
  static constexpr int CPR = RB / 16;              // 16-byte chunks per row
  static constexpr int WSH = RB == 32 ? 3 : (RB == 64 ? 2 : 1);   // log2(rows per 256 bytes)
  static constexpr int A_PLANE = PW_BM * RB, B_PLANE = BN * RB;
  static constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
  // the 16 lanes of a ds_read_b128 group ({0-3,12-15,20-27} / {4-11,16-19,28-31} of 32 rows) hit 16 distinct 16-byte
  // bank groups when chunk c of row r sits at slot c ^ swz(r)
  __device__ static __forceinline__ int swz(int row) { return (row >> WSH) & (CPR - 1); }
};

__device__ __forceinline__ void pw_lds_dma16(const __amdgpu_buffer_rsrc_t rsrc, char* lds_wave_base, unsigned voff, int soff) {
  // 64 lanes x 16 bytes -> lds_wave_base + 16 * lane (the LDS address is wave-uniform: M0); out-of-range lanes write zeros
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (void __attribute__((address_space(3)))*)lds_wave_base, 16, (int)voff, soff, 0, 0);
}

// BN: columns of the tile - 384 (8 waves of 64 x 96) or 256 (64 x 64: the 1024- and 2048-wide layers of the exit flow and the
// 256-wide ones at many rows, whose last 384-wide tile would be two thirds / one third empty; round 4)
// BNB (round 5; fp32 x6 form, barrier in the middle of the k-step): the launch is the input gradient of a pointwise convolution
// whose OUTPUT feeds a training-mode BatchNormalization, and the A operand is that layer's backward apply evaluated on the fly:
//     a[m][c] = gamma invstd ((g - dbeta / n) - xhat dgamma / n),  xhat = (x - mean) invstd,  g = dy [where the fused ReLU passed]
// from dy (the gradient of the BatchNormalization's output, p.x) and x (its raw input = this convolution's forward output,
// p.bnb.x), with the finished column sums dgamma / dbeta.  The per-channel constants sit in a table in LDS behind the two stages
// (seven float4 per four channels, built in the prologue); a thread's chunk is transformed in the MFMA gaps right before it is
// split, and the workgroups of column tile 0 also store it (p.bnb.dz): the filter gradient reads the applied gradient from there.
// The 47-microsecond bn_bwd_apply launch in front of every such dgrad (three passes over the tensor) is gone; its bytes ride in
// a kernel that is bound by the matrix pipe.
template <int NPL, typename TA, int BN = 384, bool BNB = false>
__global__ __launch_bounds__(512, 2) void pw_wide_kernel(const IgemmParams p) {
  static_assert((NPL == 3 && std::is_same<TA, float>::value) || (NPL == 1 && !std::is_same<TA, float>::value),
                "x6 on fp32 storage, one plane on bf16 storage");
  static_assert(BN == 384 || BN == 256 || BN == 512, "tile width");
  static_assert(!BNB || (NPL == 3 && BN != 512), "the BatchNormalization backward rides in the fp32 form with the barrier in the middle");
  using G = PwGeom<NPL, BN>;
  constexpr int KS = G::KS, BKW = G::BKW, RB = G::RB, CPR = G::CPR;
  constexpr int WGM = 2, WGN = 4, WM = 64, WN = BN / WGN, TM = 2, TN = WN / 32;
  constexpr unsigned OOB = 0x80000000u;

  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages] x { A planes [NPL][128][RB], B planes [NPL][384][RB] }
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // in an SGPR: LDS-DMA bases and piece numbers are scalar arithmetic
  const uint32_t bid = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t ntn = (p.Nout + BN - 1) / BN;
  const uint32_t tile_m = bid / ntn, tile_n = bid - tile_m * ntn;   // the column tiles of a row tile are neighbours: A from L2
  const int m0 = tile_m * PW_BM, n0 = tile_n * BN;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wq), 0, (int)p.w_bytes, 0x00020000);
  const int nk = (p.K + BKW - 1) / BKW;

  // ---- operand delivery ------------------------------------------------------------------------------------------------
  // B: DMA pieces of 1 KB = (1024 / RB) rows of one plane; piece g -> plane g / PPP, rows (g % PPP) * RPP ...
  constexpr int RPP = 1024 / RB;                  // rows per piece: 32 (x6) / 8 (bf16)
  constexpr int PPP = BN / RPP;                // pieces per B plane: 12 / 48
  constexpr int NBP = NPL * PPP;                  // B pieces per stage: 36 / 48
  constexpr int NBW = (NBP + 7) / 8;              // per wave: 5 (waves 0-3; 4 for waves 4-7) / 6
  unsigned b_voff[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int g = wave + 8 * i;
    const int pl = g / PPP, rb = g - pl * PPP;
    const int row = rb * RPP + lane / CPR, slot = lane % CPR;
    const int c = slot ^ G::swz(row);
    const int n = n0 + row;
    // k-block-major planes [pl][k block][Npad][BKW] (pw_planes_*): the 1 KB of a piece are contiguous in memory - eight full
    // 128-byte lines.  (From row planes [n][k] a piece of 32 rows x 32 bytes touches 32 lines and uses a quarter of each.)
    b_voff[i] = (g < NBP && n < p.Npad) ? (unsigned)((((pl * nk) * p.Npad + n) * BKW + 8 * c) * 2) : OOB;
  }
  const int b_step = p.Npad * BKW * 2;   // bytes from one k block to the next
  auto issue_B = [&](int ks, int stage) {
    char* base = smem + stage * G::STAGE + NPL * G::A_PLANE;
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      const int g = wave + 8 * i;
      if (NBP % 8 == 0 || g < NBP) {   // wave-uniform
        const int pl = g / PPP, rb = g - pl * PPP;
        pw_lds_dma16(rsrc_w, base + pl * G::B_PLANE + rb * 1024, b_voff[i], ks * b_step);
      }
    }
  };

  // A, x6: thread -> row t / 4, floats 4 * (t % 4) .. + 3 of the k-step; one 16-byte load, split, three 8-byte LDS stores
  // A, bf16: DMA pieces of 8 rows x 128 bytes, 16 pieces per stage, two per wave
  const int a_row = t >> 2, a_kq = t & 3;
  unsigned a_voff = OOB;
  unsigned a16_voff[2] = {OOB, OOB};
  if constexpr (NPL == 3) {
    if (m0 + a_row < p.M) a_voff = (unsigned)(((m0 + a_row) * p.x_ld + 4 * a_kq) * 4);
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = wave + 8 * i;
      const int row = g * RPP + lane / CPR, slot = lane % CPR;
      const int c = slot ^ G::swz(row);
      if (m0 + row < p.M) a16_voff[i] = (unsigned)(((m0 + row) * p.x_ld + 8 * c) * 2);
    }
  }
  const bool ktail = (p.K % BKW) != 0;
  u32x4_t ra = {0u, 0u, 0u, 0u};
  u32x4_t rx = {0u, 0u, 0u, 0u};   // BNB: the BatchNormalization's raw input beside its output gradient (ra)
  const __amdgpu_buffer_rsrc_t rsrc_x2 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNB ? p.bnb.x : p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_dz =
      __builtin_amdgcn_make_buffer_rsrc(BNB ? p.bnb.dz : const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  char* const tbl = smem + 2 * G::STAGE;   // BNB: [channel / 4][mean, invstd, gamma, beta, gamma invstd, dbeta / n, dgamma] x float4
  auto load_A = [&](int ks) {   // x6: registers
    ra = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)a_voff, ks * BKW * 4, 0);
    if constexpr (BNB) rx = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x2, (int)((p.bnb.relu & 4) ? OOB : a_voff), ks * BKW * 4, 0);   // (bit 2: timing experiment, no x load)
  };
  // BNB: the constants of this thread's four channels of k-step ks -> registers; the transform of ra (in place); the store
  f32x4 tq[7];
  auto bnb_consts = [&](int ks) {
    const f32x4* tp = reinterpret_cast<const f32x4*>(tbl + (ks * 4 + a_kq) * 112);
#pragma unroll
    for (int i = 0; i < 7; ++i) tq[i] = tp[i];
  };
  auto bnb_apply = [&](int e0, int ne = 2) {   // elements e0 .. e0 + ne - 1 of the chunk
    f32x4 g = __builtin_bit_cast(f32x4, ra);
    const f32x4 xv = __builtin_bit_cast(f32x4, rx);
#pragma unroll
    for (int e = e0; e < e0 + ne; ++e) {
      const float xh = (xv[e] - tq[0][e]) * tq[1][e];
      float gg = g[e];
      if (p.bnb.relu & 1) gg = fmaf(xh, tq[2][e], tq[3][e]) > 0.f ? gg : 0.f;   // bn_apply's own expression decides the mask
      float o = tq[4][e] * ((gg - tq[5][e]) - (xh * tq[6][e]) * p.bnb.inv_n);
      // the ROUNDED fp32 value is what the unfused pair multiplies: without this fence the compiler contracts this product into
      // the split's residual (a - bf16(a) as an fma of the unrounded product), the planes then carry more bits than the stored
      // dz and the input gradient differs from sg_bn_train_bwd_apply + sg_conv2d_dgrad in the last place (measured: 3e-7)
      asm volatile("" : "+v"(o));
      g[e] = o;
    }
    ra = __builtin_bit_cast(u32x4_t, g);
  };
  auto bnb_store = [&](int ks) {   // column tile 0 keeps the applied gradient for the filter gradient (others: out of range, dropped)
    const bool keep = tile_n == 0 && (ks * BKW + 4 * a_kq) < p.K && !(p.bnb.relu & 2);   // (bit 1 of relu: timing experiment, no store)
    __builtin_amdgcn_raw_buffer_store_b128(ra, rsrc_dz, (int)(keep ? a_voff : OOB), ks * BKW * 4, 0);
  };
  auto bnb_xform = [&](int ks) {
    bnb_consts(ks);
    bnb_apply(0);
    bnb_apply(2);
    bnb_store(ks);
  };
  auto store_A = [&](int stage) {
    const f32x4 f = __builtin_bit_cast(f32x4, ra);
    unsigned h0, m0_, l0, h1, m1, l1;
    split3_pair(f[0], f[1], h0, m0_, l0);
    split3_pair(f[2], f[3], h1, m1, l1);
    char* dst = smem + stage * G::STAGE + a_row * RB + (((a_kq >> 1) ^ G::swz(a_row)) << 4) + (a_kq & 1) * 8;
    *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){h0, h1};
    *reinterpret_cast<u32x2_t*>(dst + G::A_PLANE) = (u32x2_t){m0_, m1};
    *reinterpret_cast<u32x2_t*>(dst + 2 * G::A_PLANE) = (u32x2_t){l0, l1};
  };
  auto issue_A16 = [&](int ks, int stage) {   // bf16: DMA; chunks past K (a ragged last stage) are written as zeros
    char* base = smem + stage * G::STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int g = wave + 8 * i;
      const int row = g * RPP + lane / CPR, slot = lane % CPR;
      const int c = slot ^ G::swz(row);
      const bool kv = !ktail || (ks * BKW + 8 * c < p.K);
      pw_lds_dma16(rsrc_x, base + g * 1024, kv ? a16_voff[i] : OOB, ks * BKW * 2);
    }
  };

  // ---- MFMA side ---------------------------------------------------------------------------------------------------------
  const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
  const int lr = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int sw = G::swz(lr);   // wm, wn and the 32-row sub-tile offsets are multiples of 32: the swizzle depends on lr only
  const int a_lane = (wm + lr) * RB, b_lane = NPL * G::A_PLANE + (wn + lr) * RB;

  auto compute = [&](int stage, bool no_mfma = false) {
    const char* sb = smem + stage * G::STAGE;
    if constexpr (NPL == 3) {
      const int ko = (lh ^ sw) << 4;
      bf16x8_t af[TM][3], bf[TN][3];
      // reads in the order the MFMAs consume them
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) af[0][pl] = *reinterpret_cast<const bf16x8_t*>(sb + a_lane + pl * G::A_PLANE + ko);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          bf[j][pl] = *reinterpret_cast<const bf16x8_t*>(sb + b_lane + pl * G::B_PLANE + 32 * j * RB + ko);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) af[1][pl] = *reinterpret_cast<const bf16x8_t*>(sb + a_lane + pl * G::A_PLANE + 32 * RB + ko);
      if (no_mfma) {   // timing ablation: the reads are kept alive, nothing is multiplied
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          asm volatile("" ::"v"(af[0][pl]), "v"(af[1][pl]));
#pragma unroll
          for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bf[j][pl]));
        }
        return;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          // smallest terms first (conv_x6_kernel's order)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
        }
    } else {
      bf16x8_t af[2][TM], bf[2][TN];
      auto frags = [&](int ks, bf16x8_t (&a)[TM], bf16x8_t (&b)[TN]) {
        const int ko = ((2 * ks + lh) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(sb + a_lane + 32 * i * RB + ko);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(sb + b_lane + 32 * j * RB + ko);
      };
      frags(0, af[0], bf[0]);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) frags(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);   // one k-step of look-ahead
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- x6: one k-step with the staging of the next stage woven between its MFMAs, one small piece per MFMA, every piece
  // fenced (sched_barrier) so that the stream is exactly: fragment reads, then MFMA / piece / MFMA / piece ...  An MFMA holds
  // the vector issue for 8 of its 32 cycles; the rest of the gap takes the piece (conv_x6_kernel's fused_step, measured there:
  // left to itself the scheduler runs the phases back to back).  MODE 2: everything; 1: no fp32 load (last but one k-step);
  // 0: multiply only (last k-step).
  auto step3 = [&](int s, auto MODE_) {
    constexpr int MODE = decltype(MODE_)::value;
    const int cur = s & 1, nxt = cur ^ 1;
    const char* sb = smem + cur * G::STAGE;
    const int ko = (lh ^ sw) << 4;
    bf16x8_t af[TM][3], bf[TN][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) af[0][pl] = *reinterpret_cast<const bf16x8_t*>(sb + a_lane + pl * G::A_PLANE + ko);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8_t*>(sb + b_lane + pl * G::B_PLANE + 32 * j * RB + ko);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) af[1][pl] = *reinterpret_cast<const bf16x8_t*>(sb + a_lane + pl * G::A_PLANE + 32 * RB + ko);
    __builtin_amdgcn_sched_barrier(0);
    unsigned hh[2], mm[2], ll[2];
    char* adst = smem + nxt * G::STAGE + a_row * RB + (((a_kq >> 1) ^ G::swz(a_row)) << 4) + (a_kq & 1) * 8;
    char* bbase = smem + nxt * G::STAGE + NPL * G::A_PLANE;
    auto piece = [&](int w) {
      if (MODE == 0) return;
      if (w < 2) {
        const f32x4 f = __builtin_bit_cast(f32x4, ra);
        split3_pair(f[2 * w], f[2 * w + 1], hh[w], mm[w], ll[w]);
      } else if (w == 2) {
        *reinterpret_cast<u32x2_t*>(adst) = (u32x2_t){hh[0], hh[1]};
      } else if (w == 3) {
        *reinterpret_cast<u32x2_t*>(adst + G::A_PLANE) = (u32x2_t){mm[0], mm[1]};
      } else if (w == 4) {
        *reinterpret_cast<u32x2_t*>(adst + 2 * G::A_PLANE) = (u32x2_t){ll[0], ll[1]};
      } else if (w < 5 + NBW) {
        const int i = w - 5, g = wave + 8 * i;
        if (i < NBW - 1 || NBP % 8 == 0 || g < NBP) {   // only the last round of pieces is ragged (36 pieces, 8 waves)
          const int pl = g / PPP, rb = g - pl * PPP;
          pw_lds_dma16(rsrc_w, bbase + pl * G::B_PLANE + rb * 1024, b_voff[i], (s + 1) * b_step);
        }
      } else if (w == 5 + NBW) {
        if (MODE == 2) ra = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)a_voff, (s + 2) * BKW * 4, 0);
      }
    };
    int q = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first (conv_x6_kernel's order)
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA_[u]], bf[j][PB_[u]], acc[i][j], 0, 0, 0);
          piece(q);
          ++q;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    // B(s+1) is older than A(s+2) in the queue: vmcnt(1) retires it and leaves the fp32 load in flight across the barrier
    if (MODE == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // ---- x6, second form (p.stagger = SG_PW_VAR = 1, default): the barrier in the MIDDLE of the k-step (conv_x6w_kernel's finding:
  // with the barrier at the end, every wave opens the next k-step by waiting for its 15 fragment reads - with all eight waves in
  // step the matrix pipe idles for an LDS round trip per k-step).  A k-step's 36 MFMAs are two halves of 18 (row sub-tile 0, 1):
  //   first half   MFMAs on af[0] x bf[0..2]; bf[0] and af[0] were PREFETCHED in the previous k-step, the other nine fragments are
  //                read here, woven behind the first MFMAs (stage s is still whole);
  //   middle       wait: B(s+1)'s DMA (issued a whole k-step ago) and this thread's A(s+1) stores; barrier: stage s+1 is published,
  //                every wave holds all fragments of stage s in registers - its buffer is free;
  //   second half  MFMAs on af[1] x bf[0..2] with, woven: split + stores of A(s+2) and the DMA of B(s+2) into the buffer of stage
  //                s, the fp32 load of A(s+3), the six prefetch reads (af[0], bf[0]) of stage s+1.
  // Same products in the same order as step3: the results are bit-identical (tests/test_ops_gpu.py).
  // MODE 3: everything; 2: no fp32 load (s + 3 >= nk); 1: prefetch only (s + 2 >= nk); 0: the last k-step.
  auto step3m = [&](int s, auto MODE_, bf16x8_t (&ca0)[3], bf16x8_t (&cb0)[3], bf16x8_t (&na0)[3], bf16x8_t (&nb0)[3]) {
    constexpr int MODE = decltype(MODE_)::value;
    const int cur = s & 1, nxt = cur ^ 1;
    const char* sb = smem + cur * G::STAGE;
    const char* sn = smem + nxt * G::STAGE;
    const int ko = (lh ^ sw) << 4;
    bf16x8_t a1[3], bj[TN > 1 ? TN - 1 : 1][3];   // bj[j - 1]: column sub-tile j >= 1
    unsigned hh[2], mm[2], ll[2];
    char* adst = smem + cur * G::STAGE + a_row * RB + (((a_kq >> 1) ^ G::swz(a_row)) << 4) + (a_kq & 1) * 8;   // stage s+2 -> buffer of s
    char* bbase = smem + cur * G::STAGE + NPL * G::A_PLANE;
    constexpr int NREST = 3 * (TN - 1) + 3;
    auto rest = [&](int w) {   // first half: the fragments not prefetched, in the order the MFMAs want them
      if (w < 3 * (TN - 1)) bj[w / 3][w % 3] = *reinterpret_cast<const bf16x8_t*>(sb + b_lane + (w % 3) * G::B_PLANE + 32 * (w / 3 + 1) * RB + ko);
      else if (w < NREST) a1[w - 3 * (TN - 1)] = *reinterpret_cast<const bf16x8_t*>(sb + a_lane + (w - 3 * (TN - 1)) * G::A_PLANE + 32 * RB + ko);
    };
    // second half, in this order: the DMA pieces of B(s+2) (longest flight), the six prefetch reads of stage s+1 (the next k-step
    // opens with them), split + stores of A(s+2), the fp32 load of A(s+3)
    auto piece = [&](int w) {
      if constexpr (BNB) {
        // the constants of chunk A(s+2) are requested FIRST (their LDS round trip runs under the next MFMAs; right in front of
        // their use the wave waited for it: +21 us per launch), the transform follows the prefetch reads one element per gap
        if (w == 0) {
          if (MODE >= 2) bnb_consts(s + 2);
          return;
        }
        w -= 1;
      }
      if (w < NBW) {
        if (MODE >= 2) {
          const int i = w, g = wave + 8 * i;
          if (i < NBW - 1 || NBP % 8 == 0 || g < NBP) {   // only the last round of pieces is ragged (36 pieces, 8 waves)
            const int pl = g / PPP, rb = g - pl * PPP;
            pw_lds_dma16(rsrc_w, bbase + pl * G::B_PLANE + rb * 1024, b_voff[i], (s + 2) * b_step);
          }
        }
      } else if (w < NBW + 3) {
        if (MODE >= 1) na0[w - NBW] = *reinterpret_cast<const bf16x8_t*>(sn + a_lane + (w - NBW) * G::A_PLANE + ko);
      } else if (w < NBW + 6) {
        if (MODE >= 1) nb0[w - NBW - 3] = *reinterpret_cast<const bf16x8_t*>(sn + b_lane + (w - NBW - 3) * G::B_PLANE + ko);
      } else if (MODE >= 2) {
        int v = w - (NBW + 6);
        if constexpr (BNB) {   // five more pieces in front of the split: the chunk A(s+2) becomes the BatchNormalization's dx
          if (v < 4) { bnb_apply(v, 1); return; }
          if (v == 4) { bnb_store(s + 2); return; }
          v -= 5;
        }
        if (v < 2) {
          const f32x4 f = __builtin_bit_cast(f32x4, ra);
          split3_pair(f[2 * v], f[2 * v + 1], hh[v], mm[v], ll[v]);
        } else if (v == 2) {
          *reinterpret_cast<u32x2_t*>(adst) = (u32x2_t){hh[0], hh[1]};
        } else if (v == 3) {
          *reinterpret_cast<u32x2_t*>(adst + G::A_PLANE) = (u32x2_t){mm[0], mm[1]};
        } else if (v == 4) {
          *reinterpret_cast<u32x2_t*>(adst + 2 * G::A_PLANE) = (u32x2_t){ll[0], ll[1]};
        } else if (v == 5) {
          if (MODE == 3) load_A(s + 3);
        }
      }
    };
    constexpr int NPIECE = NBW + 6 + 6 + (BNB ? 6 : 0), NGAP = 6 * TN;   // the pieces are spread over the half's MFMA gaps (two in some at BN = 256)
    static_assert(NPIECE <= 2 * NGAP && NREST <= NGAP, "pieces per MFMA gap");
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first (conv_x6_kernel's order)
    int q = 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const bf16x8_t bv = j == 0 ? cb0[PB_[u]] : bj[j > 0 ? j - 1 : 0][PB_[u]];
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca0[PA_[u]], bv, acc[0][j], 0, 0, 0);
        rest(q);
        ++q;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (MODE >= 1) {
      // B(s+1) is older than the fp32 load of A(s+2) in the queue: vmcnt(1) retires it and leaves that load in flight
      // (BNB: behind B(s+1) the queue holds the store of dz(s+1) and TWO loads of A(s+2): vmcnt(3))
      if (MODE >= 2) {
        if constexpr (BNB) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    q = 0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const bf16x8_t bv = j == 0 ? cb0[PB_[u]] : bj[j > 0 ? j - 1 : 0][PB_[u]];
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[PA_[u]], bv, acc[1][j], 0, 0, 0);
#pragma unroll
        for (int w = (q * NPIECE) / NGAP; w < ((q + 1) * NPIECE) / NGAP; ++w) piece(w);
        ++q;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // ---- the k loop: one barrier per stage -------------------------------------------------------------------------------------
  // p.ablate (SG_PW_ABLATE; timing-only diagnostics, results wrong): 1 = no A path (load, split, LDS store), 2 = no B DMA,
  // 4 = no MFMAs (the fragment reads stay), 8 = no fragment reads and no MFMAs, 16 = no barrier
  const bool ab_a = (p.ablate & 1) != 0, ab_b = (p.ablate & 2) != 0, ab_mm = (p.ablate & 4) != 0, ab_rd = (p.ablate & 8) != 0;
  const bool ab_bar = (p.ablate & 16) != 0;
  if (nk > 0) {
    if constexpr (NPL == 3) {
      load_A(0);
      issue_B(0, 0);
      if constexpr (BNB) {
        // the constants table: channel c -> slot (c / 4) * 28 + param * 4 + c % 4; channels past K (the ragged last k-step) get
        // zeros, so that whatever the chunk holds there becomes 0 x (finite) = 0
        float* tf = reinterpret_cast<float*>(tbl);
        for (int c = t; c < nk * BKW; c += 512) {
          float v7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          if (c < p.K) {
            const float iv = p.bnb.invstd[c], gm = p.bnb.gamma[c];
            v7[0] = p.bnb.mean[c]; v7[1] = iv; v7[2] = gm; v7[3] = p.bnb.beta ? p.bnb.beta[c] : 0.f;
            v7[4] = gm * iv; v7[5] = p.bnb.dbeta[c] * p.bnb.inv_n; v7[6] = p.bnb.dgamma[c];
          }
#pragma unroll
          for (int i = 0; i < 7; ++i) tf[(c >> 2) * 28 + i * 4 + (c & 3)] = v7[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bnb_xform(0);
      }
      store_A(0);                    // (the compiler waits for ra here)
      if (nk > 1) load_A(1);
      if constexpr (BNB) asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // B(0), dz(0) done; the two loads of A(1) may fly
      else asm volatile("s_waitcnt vmcnt(1)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // B(0) has landed; A(1) may still fly
      if (nk == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (p.ablate == 0 && p.stagger == 1 && BN != 512) {   // the woven step with the barrier in the middle of the k-step (512-wide
                                                              // tiles: its two fragment sets do not fit beside 128 accumulator registers)
        bf16x8_t pa0[3], pb0[3], pa1[3], pb1[3];
        // stage 1 is staged before the loop (the loop stages s + 2 during k-step s), A(2) goes into flight
        if (nk > 1) {
          issue_B(1, 1);
          if constexpr (BNB) bnb_xform(1);
          store_A(1);                // (the compiler waits for ra = A(1) here)
          if (nk > 2) load_A(2);
        }
        {
          const char* sb = smem;
          const int ko = (lh ^ sw) << 4;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) pa0[pl] = *reinterpret_cast<const bf16x8_t*>(sb + a_lane + pl * G::A_PLANE + ko);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) pb0[pl] = *reinterpret_cast<const bf16x8_t*>(sb + b_lane + pl * G::B_PLANE + ko);
        }
        auto run = [&](int s, auto MODE_) {
          step3m(s, MODE_, pa0, pb0, pa1, pb1);
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {   // (register moves the allocator folds away where it can)
            pa0[pl] = pa1[pl];
            pb0[pl] = pb1[pl];
          }
        };
        int s = 0;
        for (; s + 3 < nk; ++s) run(s, IC<3>{});
        if (s + 2 < nk) { run(s, IC<2>{}); ++s; }
        if (s + 1 < nk) { run(s, IC<1>{}); ++s; }
        run(s, IC<0>{});
      } else if (p.ablate == 0) {    // the woven step, barrier at the end of the k-step (SG_PW_VAR=0; the plain loop below serves the timing ablations)
        int s = 0;
        for (; s + 2 < nk; ++s) step3(s, IC<2>{});
        if (s + 1 < nk) { step3(s, IC<1>{}); ++s; }
        step3(s, IC<0>{});
      } else
      for (int s = 0; s < nk; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < nk) {            // uniform
          if (!ab_a) store_A(nxt);   // A(s+1): loaded during the previous k-step
          __builtin_amdgcn_sched_barrier(0);
          if (!ab_b) issue_B(s + 1, nxt);       // stage nxt was last read before the previous barrier
          if (s + 2 < nk && !ab_a) load_A(s + 2);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (!ab_rd) compute(cur, ab_mm);
        __builtin_amdgcn_sched_barrier(0);
        // B(s+1) is older than A(s+2) in the queue: vmcnt(1) retires it and leaves the fp32 load in flight
        if (s + 2 < nk && !ab_a) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!ab_bar) __builtin_amdgcn_s_barrier();
      }
    } else if (p.stagger == 1) {
      // bf16 storage, second form: the barrier between k-steps 2 and 3 of the 64-deep stage (every fragment of the stage is in
      // registers or in flight by then), the DMA of stage s+2 and the first fragments of stage s+1 behind it, under the MFMAs of
      // k-step 3 - the next stage opens on fragments that are already there.  Same MFMAs in the same order as the first form.
      static_assert(NPL == 3 || KS == 4, "the bf16 stage has four k-steps");
      bf16x8_t af[2][TM], bf[2][TN];
      auto frags = [&](int stage, int ks, bf16x8_t (&a)[TM], bf16x8_t (&b)[TN]) {
        const char* sb = smem + stage * G::STAGE;
        const int ko = ((2 * ks + lh) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(sb + a_lane + 32 * i * RB + ko);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(sb + b_lane + 32 * j * RB + ko);
      };
      auto mfmas = [&](const bf16x8_t (&a)[TM], const bf16x8_t (&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      };
      issue_A16(0, 0);
      issue_B(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (nk > 1) {
        issue_A16(1, 1);
        issue_B(1, 1);
      }
      frags(0, 0, af[0], bf[0]);
      for (int s = 0; s < nk; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        frags(cur, 1, af[1], bf[1]);
        mfmas(af[0], bf[0]);
        frags(cur, 2, af[0], bf[0]);
        mfmas(af[1], bf[1]);
        frags(cur, 3, af[1], bf[1]);
        mfmas(af[0], bf[0]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < nk) {   // uniform; the buffer of stage s is free
          issue_A16(s + 2, cur);
          issue_B(s + 2, cur);
        }
        if (s + 1 < nk) frags(nxt, 0, af[0], bf[0]);
        mfmas(af[1], bf[1]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      issue_A16(0, 0);
      issue_B(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int s = 0; s < nk; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < nk) {
          issue_A16(s + 1, nxt);
          issue_B(s + 1, nxt);
          __builtin_amdgcn_sched_barrier(0);
        }
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------------
  const bool has_bias = (p.flags & SG_EPI_BIAS) != 0;
  const bool do_relu = (p.flags & SG_EPI_RELU) != 0;
  if constexpr (NPL == 3) {
    // Branch-free stores through a buffer descriptor: a lane outside the tensor (column >= Nout, row >= M) carries an
    // out-of-range offset and the hardware drops its store.  (A per-store `if` is a basic block of its own, and across
    // block boundaries the compiler's wait insertion falls back to vmcnt(0): 96 stores, each waiting for the one before.)
    const uint32_t y_bytes = (uint32_t)((((int64_t)p.M - 1) * p.y_ld + p.Nout) * 4);   // < 2^31: pw_wide_ok
    const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)y_bytes, 0x00020000);
    const bool full = m0 + PW_BM <= p.M;   // uniform: no row of this tile is past the end
    const unsigned row0 = (unsigned)(m0 + wm + 4 * lh);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + 32 * j + lr;
      const bool cv = col < p.Nout;
      const float bv = has_bias ? p.bias[cv ? col : p.Nout - 1] : 0.f;
      const unsigned cterm = cv ? (unsigned)col * 4u : OOB;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned row = row0 + 32 * i + (r & 3) + 8 * (r >> 2);
          float v = acc[i][j][r] + bv;
          v = do_relu ? fmaxf(v, 0.f) : v;
          unsigned voff = cterm + row * (unsigned)p.y_ld * 4u;
          if (!full) voff = row < (unsigned)p.M ? voff : OOB;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_y, (int)voff, 0, 0);
        }
    }
  } else {
    // the bf16 tile goes through LDS so that it leaves as 16-byte row chunks (conv_b16_kernel's epilogue)
    bf16_t* __restrict__ py = reinterpret_cast<bf16_t*>(p.y);
    constexpr int TP = BN * 2 + 16;
    const bool wide = (p.y_ld % 8 == 0) && (p.Nout % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0);
    if (wide) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cl = wn + 32 * j + lr;
        const int col = n0 + cl;
        const float bv = (has_bias && col < p.Nout) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float v = acc[i][j][r] + bv;
            if (do_relu) v = fmaxf(v, 0.f);
            *reinterpret_cast<unsigned short*>(smem + rl * TP + cl * 2) = f32_to_bf16_bits(v);
          }
      }
      __syncthreads();
      constexpr int CPT = BN / 8;
      for (int idx = t; idx < PW_BM * CPT; idx += 512) {
        const int rl = idx / CPT, c = idx - rl * CPT;
        const int row = m0 + rl, col = n0 + 8 * c;
        if (row < p.M && col < p.Nout)
          *reinterpret_cast<u32x4_t*>(py + (int64_t)row * p.y_ld + col) = *reinterpret_cast<const u32x4_t*>(smem + rl * TP + c * 16);
      }
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn + 32 * j + lr;
        const bool cv = col < p.Nout;
        const float bv = (has_bias && cv) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (cv && row < p.M) {
              float v = acc[i][j][r] + bv;
              if (do_relu) v = fmaxf(v, 0.f);
              st1<bf16_t>(py + (int64_t)row * p.y_ld + col, v);
            }
          }
      }
    }
  }

  // ---- BatchNormalization statistics of this 128-row tile (conv_x6_kernel's scheme and layout: stats[tile_m][2][Nout]) ---------
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem);   // [WGM][BN] partials, then [BN] tile means
    float* tmean = red + WGM * BN;
    const int wrow = wave / WGN;
    const int nvalid = (p.M - m0) < PW_BM ? (p.M - m0) : PW_BM;
    __syncthreads();   // every wave is done with the stage buffers / the staged output tile
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cl = wn + 32 * j + lr;
        const int col = n0 + cl;
        const float bv = (has_bias && col < p.Nout) ? p.bias[col] : 0.f;
        const float mu = pass ? tmean[cl] : 0.f;
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float dlt = acc[i][j][r] + bv - mu;
            if (row < p.M) sacc += pass ? dlt * dlt : dlt;
          }
        sacc += __shfl_xor(sacc, 32, 64);   // lanes l and l + 32 hold the same column
        if (lh == 0) red[wrow * BN + cl] = sacc;
      }
      __syncthreads();
      for (int cl = t; cl < BN; cl += 512) {
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

constexpr int PW_BNB_MAXK = 2048;   // channels the constants table of the BNB form is sized for (28 bytes each)

template <int NPL, typename TA, int BN = 384, bool BNB = false>
int launch_pw_wide(const IgemmParams& p, hipStream_t st) {
  using G = PwGeom<NPL, BN>;
  constexpr size_t stage_lds = 2 * (size_t)G::STAGE + (BNB ? (size_t)PW_BNB_MAXK * 28 : 0);
  constexpr size_t tile_lds = NPL == 1 ? (size_t)PW_BM * (BN * 2 + 16) : 0;
  constexpr size_t stat_lds = (size_t)3 * BN * sizeof(float);
  constexpr size_t m1 = stage_lds > tile_lds ? stage_lds : tile_lds;
  constexpr size_t lds = m1 > stat_lds ? m1 : stat_lds;
  static_assert(lds <= 160 * 1024, "LDS");
  static bool attr_done = false;
  if (!attr_done) {
    int rc = set_dyn_lds(pw_wide_kernel<NPL, TA, BN, BNB>, lds);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.M, PW_BM) * sg_cdiv(p.Nout, BN);
  if (tiles <= 0 || tiles > 0x7fffffff) {
    sg_set_error("pw_wide: bad tile count %lld", (long long)tiles);
    return SG_EINVAL;
  }
  if (p.x_bytes == 0 || (((int64_t)p.M - 1) * p.y_ld + p.Nout) * (int64_t)EL<TA>::BYTES >= (1ll << 31)) {
    sg_set_error("pw_wide: an operand of %lld rows does not fit one 2 GiB buffer descriptor", (long long)p.M);
    return SG_EINVAL;
  }
  IgemmParams q = p;
  {
    static int abl = -1;
    if (abl < 0) abl = getenv("SG_PW_ABLATE") ? atoi(getenv("SG_PW_ABLATE")) : 0;
    q.ablate = abl;
    static const int var = getenv("SG_PW_VAR") ? atoi(getenv("SG_PW_VAR")) : 1;
    q.stagger = var;   // (the field is free in this kernel) 1: barrier in the middle of the k-step, 0: at its end
  }
  if constexpr (BNB) {
    if (q.ablate != 0 || q.stagger != 1 || p.K + 16 > PW_BNB_MAXK || p.x_ld != p.K) {
      sg_set_error("pw_wide: the BatchNormalization-backward form needs the default schedule (SG_PW_VAR=1, no ablation), a dense "
                   "operand and at most %d channels", PW_BNB_MAXK - 16);
      return SG_EUNSUPPORTED;
    }
  }
  hipLaunchKernelGGL((pw_wide_kernel<NPL, TA, BN, BNB>), dim3((unsigned)tiles), dim3(512), lds, st, q);
  SG_LAUNCH_CHECK("pw_wide_kernel");
  return 0;
}

// ---- weight planes of the wide kernel: k-block-major [npl][K blocks][Npad][KD] bf16, KD = 16 (x6) / 64 (bf16 storage), Npad a
// multiple of 384, zero padded; sg_planes_job kind 3 (Ckp carries KD) / split3_weights_kernel(kd) ---------------------------
inline int pw_kd(int npl) { return npl == 3 ? 16 : 64; }
inline int pw_kpad(int K, int npl) { return (int)(sg_cdiv(K, pw_kd(npl)) * pw_kd(npl)); }
inline int pw_npad(int N, int bn = PW_BN) { return (int)(sg_cdiv(N, bn) * bn); }
inline size_t pw_planes_bytes(int K, int N, int npl, int bn = PW_BN) { return (size_t)npl * pw_kpad(K, npl) * pw_npad(N, bn) * 2; }

// Shapes the wide kernel takes: one tap (1x1), stride 1, no gather; a reduction deep enough to amortise the 128 x 384 tile's
// prologue; a column count whose last 384-wide tile is at least three quarters full (728 -> 2 tiles, 1024 -> 3, 1536 -> 4,
// 2048 -> 6; 256 stays on the 128-wide tiles); at least 6144 rows - 96 workgroups: below that the 128-wide tiles spread the
// work over more CUs and win (profiles/r03_pw_wide_mscan.txt: 4096 rows 72 vs 47 us, 6144 rows 75 vs 78 us); operands 16-byte
// aligned.  The weight planes are laid out for one kernel or the other when they are prepared (sg_conv2d_planes_job, which
// knows the batch), so the same rule must give the same answer at the launch: it reads nothing but the descriptor, and a
// launch cut into sub-batches (g_sub_batch) never takes the wide kernel.  Both kernels add the same products in the same
// order (16-deep k-steps ascending, the six x6 terms smallest first, one fp32 accumulator), so a tile's result does not
// depend on which of them its batch size selected (tests/test_fullsize_gpu.py: batch-slice invariance, bit exact).
// SG_PW_WIDE=0 switches it off, 2 takes every aligned 1x1 (tests).
// Tile width (round 4): 384 where the last 384-wide tile is at least three quarters full (728, 1536); else 256 where THAT fits
// (1024, 2048, 256) and the launch brings at least one 128 x 256 tile per CU (16384 rows x 256 columns are 128 tiles: those stay
// on conv_x6_kernel's 256 tiles of 128 x 128).  SG_PW_WIDE=3: 384 only.
inline int pw_wide_bn(const IgemmParams& p, int eb) {
  static int on = -1;
  if (on < 0) on = getenv("SG_PW_WIDE") ? atoi(getenv("SG_PW_WIDE")) : 1;
  if (!on) return 0;
  if (p.K != p.C || p.a_mul != 1 || p.div != 1 || p.off_h != 0 || p.off_w != 0) return 0;
  // (both activation tensors of a launch are below 2 GiB by construction: larger batches run as sub-batches of whole images,
  // images_per_2gib; launch_pw_wide re-checks)
  if (eb == 2 && (p.K % 8 != 0 || p.x_ld % 8 != 0 || (reinterpret_cast<uintptr_t>(p.x) & 15) != 0)) return 0;
  if (eb == 4 && (p.x_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(p.x) & 15) != 0)) return 0;
  if (g_sub_batch) return 0;
  if (on == 2) return PW_BN;
  if (p.K < 256 || p.M < 6144) return 0;
  // Width by a two-line cost model: one workgroup per CU, so a launch takes ceil(tiles / CUs) rounds of a time proportional to
  // the tile's width - 728 columns at 16384 rows: 384-wide tiles, one round; 1024 columns: 256-wide, two rounds (384-wide tiles
  // were 384 workgroups = two rounds with the second half empty: 167 -> 128 us); 2048 and 512 columns: 256-wide.  A width must
  // fill three quarters of its columns and, below 384, bring at least 192 tiles.  Ties go to 384, then 256.  SG_PW_WIDE=3: 384
  // only.  SG_PW_512=1 adds 512-wide tiles (fp32 only: 128 accumulator registers, barrier at the end of the k-step): one round of
  // them measured no better than two rounds of 256-wide ones (1024 -> 1024: 182 vs 178 us), so they are off.
  static const int w512 = getenv("SG_PW_512") ? atoi(getenv("SG_PW_512")) : 0;
  const int64_t ntm = sg_cdiv(p.M, PW_BM);
  int best = 0;
  int64_t best_cost = 0;
  const int widths[3] = {384, 256, 512};
  for (int i = 0; i < 3; ++i) {
    const int bn = widths[i];
    if (bn != 384 && on == 3) continue;
    if (bn == 512 && (!w512 || eb != 4)) continue;
    const int64_t ntn = sg_cdiv(p.Nout, bn), tiles = ntm * ntn;
    if ((double)p.Nout / (double)(ntn * bn) < 0.75) continue;
    if (bn != 384 && tiles < 192) continue;
    const int64_t cost = sg_cdiv(tiles, 256) * bn;
    if (!best || cost < best_cost) { best = bn; best_cost = cost; }
  }
  return best;
}
inline bool pw_wide_ok(const IgemmParams& p, int eb) { return pw_wide_bn(p, eb) != 0; }

// =====================================================================================================================
// The filter gradient of the same layers: dw[ci][co] = sum over pixels x[p][ci] * dy[p][co]   (1x1, stride 1; fp32, x6)
//
// wgrad_x6_kernel walks 128 x 128 tiles of dw: six row tiles x six column tiles for 728 -> 728, every element of x AND of dy
// split into its planes six times, 0.75 fragment reads per MFMA.  Here a workgroup owns 128 input channels x 384 output
// channels of dw (12 tiles) for a share of the pixels (21 splits: 252 workgroups, one per CU): x is split twice, dy six
// times - four 16-byte chunks per thread and 36 MFMAs per 16-pixel k-step instead of four per 24 - and a wave's 64 x 96
// sub-tile reads 15 fragments per 36 MFMAs.  Both operands are activations, [pixel][channel] in memory as in LDS; the
// fragments (8 consecutive pixels of one channel per lane) come from the transposed LDS read, exactly as in
// wgrad_x6_kernel (tr_frag, pitch = channel bytes + 64).  Two LDS stages of one k-step, one barrier per k-step, the
// split + LDS store of k-step s+1 and the loads of k-step s+2 woven between the MFMAs of k-step s.  Partial slabs
// [split][Cin][Cout] are summed by reduce_splits_kernel in fixed order, as for every other filter gradient.
constexpr int WPW_PA = 2 * PW_BM + 64;   // 320: LDS pitch of an A' pixel row (128 channels of bf16 + 64)
constexpr int WPW_PB = 2 * PW_BN + 64;   // 832
constexpr int WPW_STAGE = 3 * 16 * (WPW_PA + WPW_PB);   // 55296

template <int VAR>   // SG_WPW_VAR: 1 (default) = the barrier in the middle of the k-step (step_m below), 0 = at its end
__global__ __launch_bounds__(512, 2) void wgrad_pw_wide_kernel(const WgradParams p) {
  constexpr int WGN = 4, WM = 64, WN = 96, TM = 2, TN = 3;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages] x { A' [3][16][PA], B [3][16][PB] }
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint32_t ntn = (p.Cout + PW_BN - 1) / PW_BN;
  uint32_t bid, split;
  {
    const uint32_t lin = blockIdx.z * gridDim.x + blockIdx.x;
    const uint32_t o = xcd_remap(lin, gridDim.x * gridDim.z);
    split = o / gridDim.x;
    bid = o - split * gridDim.x;
  }
  const uint32_t tile_r = bid / ntn, tile_n = bid - tile_r * ntn;
  const int rbase = tile_r * PW_BM, n0 = tile_n * PW_BN;
  const int nks_total = (p.P + 15) / 16;
  const int ks_begin = (int)split * p.slabs_per_split;   // (k-steps of 16 pixels per split for this kernel)
  int ks_end = ks_begin + p.slabs_per_split;
  if (ks_end > nks_total) ks_end = nks_total;
  const int nk = ks_end - ks_begin;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
  // chunks of a k-step: A' 16 pixels x 32 chunks (one per thread), B 16 pixels x 96 chunks (three per thread)
  const int pa = t >> 5, ca = t & 31;
  const unsigned a_voff = (rbase + 4 * ca) < p.Cin ? (unsigned)((pa * p.x_ld + rbase + 4 * ca) * 4) : OOB;
  int pb[3], cb[3];
  unsigned b_voff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = t + 512 * i;
    pb[i] = idx / 96;
    cb[i] = idx - pb[i] * 96;
    b_voff[i] = (n0 + 4 * cb[i]) < p.Cout ? (unsigned)((pb[i] * p.y_ld + n0 + 4 * cb[i]) * 4) : OOB;
  }
  const bool ptail = (p.P & 15) != 0;
  u32x4_t ra, rb[3];
  auto load_piece = [&](int ks, int w) {   // w = 0: the A' chunk, 1..3: the B chunks of k-step ks (absolute)
    const int p0 = ks * 16;
    if (w == 0) {
      const bool v = !ptail || (p0 + pa < p.P);
      ra = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(v ? a_voff : OOB), p0 * p.x_ld * 4, 0);
    } else {
      const bool v = !ptail || (p0 + pb[w - 1] < p.P);
      rb[w - 1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)(v ? b_voff[w - 1] : OOB), p0 * p.y_ld * 4, 0);
    }
  };
  unsigned hh[2], mm[2], ll[2];
  auto split_piece = [&](const u32x4_t v, int half) {
    const f32x4 f = __builtin_bit_cast(f32x4, v);
    split3_pair(f[2 * half], f[2 * half + 1], hh[half], mm[half], ll[half]);
  };
  auto write_piece = [&](char* dst, int pitch) {   // the three planes of one chunk: 4 channels of one pixel, 8 bytes each
    *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){hh[0], hh[1]};
    *reinterpret_cast<u32x2_t*>(dst + 16 * pitch) = (u32x2_t){mm[0], mm[1]};
    *reinterpret_cast<u32x2_t*>(dst + 32 * pitch) = (u32x2_t){ll[0], ll[1]};
  };
  auto a_dst = [&](int stage) { return smem + stage * WPW_STAGE + pa * WPW_PA + ca * 8; };
  auto b_dst = [&](int stage, int i) { return smem + stage * WPW_STAGE + 3 * 16 * WPW_PA + pb[i] * WPW_PB + cb[i] * 8; };

  // ---- MFMA side (wgrad_x6_kernel's transposed reads): lane = 16 g + i supplies pixel row 8 (g >> 1) + (i >> 2) [+ 4 for
  // the second read], channels 16 (g & 1) + 4 (i & 3) .. + 3 of its 32-channel block
  const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int tg = lane >> 4, ti = lane & 15;
  const int tr_row = 8 * (tg >> 1) + (ti >> 2), tr_col = 16 * (tg & 1) + 4 * (ti & 3);
  const int a_lane = tr_row * WPW_PA + (wm + tr_col) * 2;
  const int b_lane = 3 * 16 * WPW_PA + tr_row * WPW_PB + (wn + tr_col) * 2;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one k-step; MODE 2: stage k-step s+1 and load s+2; 1: stage s+1; 0: multiply only
  auto step = [&](int s, auto MODE_) {
    constexpr int MODE = decltype(MODE_)::value;
    const int cur = s & 1, nxt = cur ^ 1;
    const char* sb = smem + cur * WPW_STAGE;
    bf16x8_t af[TM][3], bf[TN][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const char* a = sb + a_lane + pl * 16 * WPW_PA;
      af[0][pl] = tr_frag(a, a + 4 * WPW_PA);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const char* b = sb + b_lane + pl * 16 * WPW_PB + 64 * j;
        bf[j][pl] = tr_frag(b, b + 4 * WPW_PB);
      }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const char* a = sb + a_lane + pl * 16 * WPW_PA + 64;
      af[1][pl] = tr_frag(a, a + 4 * WPW_PA);
    }
    __builtin_amdgcn_sched_barrier(0);
    // pieces: chunk c (0 = A', 1..3 = B): split half 0, split half 1, write; then its reload - 4 pieces per chunk
    auto piece = [&](int w) {
      if (MODE == 0 || w >= 16) return;
      const int c = w >> 2, sub = w & 3;
      if (sub < 2) {
        split_piece(c == 0 ? ra : rb[c - 1 < 0 ? 0 : c - 1], sub);
      } else if (sub == 2) {
        if (c == 0) write_piece(a_dst(nxt), WPW_PA);
        else write_piece(b_dst(nxt, c - 1), WPW_PB);
      } else {
        if (MODE == 2) load_piece(ks_begin + s + 2, c);
      }
    };
    int q = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA_[u]], bf[j][PB_[u]], acc[i][j], 0, 0, 0);
          if ((q & 1) == 1) piece(q >> 1);   // one piece per two MFMAs: 16 pieces over 36 MFMAs
          ++q;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // raw: the loads of k-step s+2 stay in flight across it
  };

  // ---- second form (VAR = 1): the barrier in the MIDDLE of the k-step (pw_wide_kernel's step3m; conv_x6w_kernel's finding).
  //   first half   18 MFMAs on af[0] x bf[0..2]: af[0] and bf[0] were PREFETCHED in the previous k-step, the other nine fragments
  //                (18 transposed reads) are read here, one per MFMA; chunks 2 and 3 of stage s+1 are split and stored, their
  //                registers reloaded for stage s+2;
  //   middle       lgkmcnt(0) + barrier: stage s+1 is published, every wave holds all of stage s in registers: its buffer is free;
  //   second half  18 MFMAs on af[1] x bf[0..2]: the six prefetch fragments of stage s+1; chunks 0 and 1 of stage s+2 split and
  //                stored into the buffer of stage s, reloaded for stage s+3.
  // The same products in the same order: bit-identical to the first form.
  // MODE 3: s + 3 < nk (everything); 2: s + 2 < nk; 1: s + 1 < nk; 0: the last k-step.
  auto chunk_pieces = [&](int c, int sub, int stage, int ks_load, bool do_load) {   // sub 0, 1: split halves; 2: store; 3: reload
    if (sub < 2) {
      split_piece(c == 0 ? ra : rb[c - 1 < 0 ? 0 : c - 1], sub);
    } else if (sub == 2) {
      if (c == 0) write_piece(a_dst(stage), WPW_PA);
      else write_piece(b_dst(stage, c - 1), WPW_PB);
    } else if (do_load) {
      load_piece(ks_load, c);
    }
  };
  auto step_m = [&](int s, auto MODE_, bf16x8_t (&ca0)[3], bf16x8_t (&cb0)[3], bf16x8_t (&na0)[3], bf16x8_t (&nb0)[3]) {
    constexpr int MODE = decltype(MODE_)::value;
    const int cur = s & 1, nxt = cur ^ 1;
    const char* sb = smem + cur * WPW_STAGE;
    const char* sn = smem + nxt * WPW_STAGE;
    bf16x8_t a1[3], b1[3], b2[3];
    constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first
    // first half: gap q reads one of the nine remaining fragments (order of use: bf[1], bf[2], af[1]) on even q, and carries one
    // of the eight pieces of chunks 2, 3 of stage s+1 on odd q
    auto first = [&](int q) {
      if ((q & 1) == 0) {
        const int w = q >> 1;   // 0 .. 8
        if (w < 3) {
          const char* b = sb + b_lane + w * 16 * WPW_PB + 64 * 1;
          b1[w] = tr_frag(b, b + 4 * WPW_PB);
        } else if (w < 6) {
          const char* b = sb + b_lane + (w - 3) * 16 * WPW_PB + 64 * 2;
          b2[w - 3] = tr_frag(b, b + 4 * WPW_PB);
        } else {
          const char* a = sb + a_lane + (w - 6) * 16 * WPW_PA + 64;
          a1[w - 6] = tr_frag(a, a + 4 * WPW_PA);
        }
      } else if (MODE >= 1) {
        const int w = q >> 1;   // 0 .. 8: pieces 0 .. 7 used
        if (w < 8) chunk_pieces(2 + (w >> 2), w & 3, nxt, ks_begin + s + 2, MODE >= 2);
      }
    };
    // second half: the six prefetch fragments of stage s+1 first (the next k-step opens with them), then chunks 0, 1 of stage s+2
    auto second = [&](int q) {
      if (q < 6) {
        if (MODE >= 1) {
          if (q < 3) {
            const char* a = sn + a_lane + q * 16 * WPW_PA;
            na0[q] = tr_frag(a, a + 4 * WPW_PA);
          } else {
            const char* b = sn + b_lane + (q - 3) * 16 * WPW_PB;
            nb0[q - 3] = tr_frag(b, b + 4 * WPW_PB);
          }
        }
      } else if (MODE >= 2) {
        const int w = q - 6;   // 0 .. 11: pieces 0 .. 7 used
        if (w < 8) chunk_pieces(w >> 2, w & 3, cur, ks_begin + s + 3, MODE >= 3);
      }
    };
    int q = 0;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const bf16x8_t bv = j == 0 ? cb0[PB_[u]] : (j == 1 ? b1[PB_[u]] : b2[PB_[u]]);
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca0[PA_[u]], bv, acc[0][j], 0, 0, 0);
        first(q);
        ++q;
        __builtin_amdgcn_sched_barrier(0);
      }
    if (MODE >= 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // raw: the loads in flight stay in flight across it
    }
    __builtin_amdgcn_sched_barrier(0);
    q = 0;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const bf16x8_t bv = j == 0 ? cb0[PB_[u]] : (j == 1 ? b1[PB_[u]] : b2[PB_[u]]);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[PA_[u]], bv, acc[1][j], 0, 0, 0);
        second(q);
        ++q;
        __builtin_amdgcn_sched_barrier(0);
      }
  };

  if (nk > 0) {
#pragma unroll
    for (int w = 0; w < 4; ++w) load_piece(ks_begin, w);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      split_piece(c == 0 ? ra : rb[c - 1 < 0 ? 0 : c - 1], 0);
      split_piece(c == 0 ? ra : rb[c - 1 < 0 ? 0 : c - 1], 1);
      if (c == 0) write_piece(a_dst(0), WPW_PA);
      else write_piece(b_dst(0, c - 1), WPW_PB);
    }
    if (nk > 1) {
#pragma unroll
      for (int w = 0; w < 4; ++w) load_piece(ks_begin + 1, w);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (VAR == 0) {
      int s = 0;
      for (; s + 2 < nk; ++s) step(s, IC<2>{});
      if (s + 1 < nk) { step(s, IC<1>{}); ++s; }
      step(s, IC<0>{});
    } else {
      bf16x8_t pa0[3], pb0[3], pa1[3], pb1[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const char* a = smem + a_lane + pl * 16 * WPW_PA;
        pa0[pl] = tr_frag(a, a + 4 * WPW_PA);
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const char* b = smem + b_lane + pl * 16 * WPW_PB;
        pb0[pl] = tr_frag(b, b + 4 * WPW_PB);
      }
      if (nk > 1) {   // chunks 0, 1 of stage 1 before the loop (the loop stages them in the second half of k-step s - 2)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int sub = 0; sub < 4; ++sub) chunk_pieces(c, sub, 1, ks_begin + 2, nk > 2);
      }
      auto run = [&](int s, auto MODE_) {
        step_m(s, MODE_, pa0, pb0, pa1, pb1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {   // (register moves the allocator folds away where it can)
          pa0[pl] = pa1[pl];
          pb0[pl] = pb1[pl];
        }
      };
      int s = 0;
      for (; s + 3 < nk; ++s) run(s, IC<3>{});
      if (s + 2 < nk) { run(s, IC<2>{}); ++s; }
      if (s + 1 < nk) { run(s, IC<1>{}); ++s; }
      run(s, IC<0>{});
    }
  }

  // ---- the partial slab of this split: out[split][ci][co], branch-free buffer stores ---------------------------------------
  float* outp = p.out + (int64_t)split * p.K * p.Cout;
  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)((uint32_t)p.K * (uint32_t)p.Cout * 4u), 0x00020000);
  const unsigned row0 = (unsigned)(rbase + wm + 4 * lh);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn + 32 * j + lr;
    const unsigned cterm = col < p.Cout ? (unsigned)col * 4u : OOB;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned row = row0 + 32 * i + (r & 3) + 8 * (r >> 2);
        unsigned voff = cterm + row * (unsigned)p.Cout * 4u;
        voff = row < (unsigned)p.K ? voff : OOB;
        const float v = acc[i][j][r];   // (a named float: __builtin_bit_cast applied to the vector element itself reads lane 0 of
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_o, (int)voff, 0, 0);   // the vector)
      }
  }
}

// bf16 storage: the same 128 x 384 tile of dw and the same wave layout, but both operands ARE their single plane - no split, a
// thread's four 16-byte chunks (8 channels of one pixel each: one of x, three of dy) go from the buffer load to LDS unchanged.
// k-steps of 32 pixels (two MFMA k-blocks, 12 MFMAs per wave), two stages of 36 KB, one raw barrier per k-step, the four LDS
// stores of k-step s+1 and the four loads of k-step s+2 woven between the MFMAs.  wgrad_x6_kernel<128, .., bf16> reads x and dy
// six times each for the 728-wide layers (36 tiles of 128 x 128), this one twice and six times.
constexpr int WPB_KP = 32;                                      // pixels per k-step
constexpr int WPB_STAGE = WPB_KP * (WPW_PA + WPW_PB);           // 36864
__global__ __launch_bounds__(512, 2) void wgrad_pw_wide_b16_kernel(const WgradParams p) {
  constexpr int WGN = 4, WM = 64, WN = 96, TM = 2, TN = 3;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages] x { A' [32][PA], B [32][PB] }
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint32_t ntn = (p.Cout + PW_BN - 1) / PW_BN;
  uint32_t bid, split;
  {
    const uint32_t lin = blockIdx.z * gridDim.x + blockIdx.x;
    const uint32_t o = xcd_remap(lin, gridDim.x * gridDim.z);
    split = o / gridDim.x;
    bid = o - split * gridDim.x;
  }
  const uint32_t tile_r = bid / ntn, tile_n = bid - tile_r * ntn;
  const int rbase = tile_r * PW_BM, n0 = tile_n * PW_BN;
  const int nks_total = (p.P + WPB_KP - 1) / WPB_KP;
  const int ks_begin = (int)split * p.slabs_per_split;   // (k-steps of 32 pixels per split for this kernel)
  int ks_end = ks_begin + p.slabs_per_split;
  if (ks_end > nks_total) ks_end = nks_total;
  const int nk = ks_end - ks_begin;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
  // chunks of a k-step: A' 32 pixels x 16 chunks of 8 channels (one per thread), B 32 pixels x 48 chunks (three per thread)
  const int pa = t >> 4, ca = t & 15;
  const unsigned a_voff = (rbase + 8 * ca) < p.Cin ? (unsigned)((pa * p.x_ld + rbase + 8 * ca) * 2) : OOB;
  int pb[3], cb[3];
  unsigned b_voff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = t + 512 * i;
    pb[i] = idx / 48;
    cb[i] = idx - pb[i] * 48;
    b_voff[i] = (n0 + 8 * cb[i]) < p.Cout ? (unsigned)((pb[i] * p.y_ld + n0 + 8 * cb[i]) * 2) : OOB;
  }
  const bool ptail = (p.P & (WPB_KP - 1)) != 0;
  u32x4_t rr[4];
  auto load_piece = [&](int ks, int w) {   // w = 0: the A' chunk, 1..3: the B chunks of k-step ks (absolute)
    const int p0 = ks * WPB_KP;
    if (w == 0) {
      const bool v = !ptail || (p0 + pa < p.P);
      rr[0] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(v ? a_voff : OOB), p0 * p.x_ld * 2, 0);
    } else {
      const bool v = !ptail || (p0 + pb[w - 1] < p.P);
      rr[w] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)(v ? b_voff[w - 1] : OOB), p0 * p.y_ld * 2, 0);
    }
  };
  auto write_piece = [&](int stage, int w) {
    char* dst = smem + stage * WPB_STAGE + (w == 0 ? pa * WPW_PA + ca * 16 : WPB_KP * WPW_PA + pb[w - 1] * WPW_PB + cb[w - 1] * 16);
    *reinterpret_cast<u32x4_t*>(dst) = rr[w];
  };

  const int wm = (wave / WGN) * WM, wn = (wave % WGN) * WN;
  const int lr = lane & 31, lh = lane >> 5;
  const int tg = lane >> 4, ti = lane & 15;
  const int tr_row = 8 * (tg >> 1) + (ti >> 2), tr_col = 16 * (tg & 1) + 4 * (ti & 3);
  const int a_lane = tr_row * WPW_PA + (wm + tr_col) * 2;
  const int b_lane = WPB_KP * WPW_PA + tr_row * WPW_PB + (wn + tr_col) * 2;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one k-step; MODE 2: stage k-step s+1 and load s+2; 1: stage s+1; 0: multiply only
  auto step = [&](int s, auto MODE_) {
    constexpr int MODE = decltype(MODE_)::value;
    const int cur = s & 1, nxt = cur ^ 1;
    const char* sb = smem + cur * WPB_STAGE;
    bf16x8_t af[2][TM], bf[2][TN];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* a = sb + a_lane + kb * 16 * WPW_PA + 64 * i;
        af[kb][i] = tr_frag(a, a + 4 * WPW_PA);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const char* b = sb + b_lane + kb * 16 * WPW_PB + 64 * j;
        bf[kb][j] = tr_frag(b, b + 4 * WPW_PB);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    auto piece = [&](int w) {   // 0..3: LDS stores of k-step s+1, 4..7: loads of k-step s+2 into the registers just stored
      if (MODE == 0 || w >= 8) return;
      if (w < 4) write_piece(nxt, w);
      else if (MODE == 2) load_piece(ks_begin + s + 2, w - 4);
    };
    int q = 0;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kb][i], bf[kb][j], acc[i][j], 0, 0, 0);
          if (q < 8) piece(q);
          ++q;
          __builtin_amdgcn_sched_barrier(0);
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // raw: the loads of k-step s+2 stay in flight across it
  };

  if (nk > 0) {
#pragma unroll
    for (int w = 0; w < 4; ++w) load_piece(ks_begin, w);
#pragma unroll
    for (int w = 0; w < 4; ++w) write_piece(0, w);
    if (nk > 1) {
#pragma unroll
      for (int w = 0; w < 4; ++w) load_piece(ks_begin + 1, w);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int s = 0;
    for (; s + 2 < nk; ++s) step(s, IC<2>{});
    if (s + 1 < nk) { step(s, IC<1>{}); ++s; }
    step(s, IC<0>{});
  }

  float* outp = p.out + (int64_t)split * p.K * p.Cout;
  const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(outp, 0, (int)((uint32_t)p.K * (uint32_t)p.Cout * 4u), 0x00020000);
  const unsigned row0 = (unsigned)(rbase + wm + 4 * lh);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn + 32 * j + lr;
    const unsigned cterm = col < p.Cout ? (unsigned)col * 4u : OOB;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned row = row0 + 32 * i + (r & 3) + 8 * (r >> 2);
        unsigned voff = cterm + row * (unsigned)p.Cout * 4u;
        voff = row < (unsigned)p.K ? voff : OOB;
        const float v = acc[i][j][r];
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_o, (int)voff, 0, 0);
      }
  }
}

// plan of the wide filter gradient: tiles of 128 x 384, the pixel reduction cut into S shares of whole 16-pixel k-steps so
// that tiles x S fills the CUs once
inline bool wgrad_pw_wide_geom(const sg_conv_desc* d, int eb = 4) {
  static int on = -1;
  if (on < 0) on = getenv("SG_PW_WIDE") ? atoi(getenv("SG_PW_WIDE")) : 1;
  if (!on) return false;
  if (d->KH != 1 || d->KW != 1 || d->stride != 1 || d->pad_t != 0 || d->pad_l != 0) return false;
  const int xl = d->x_ld ? d->x_ld : d->Cin, yl = d->y_ld ? d->y_ld : d->Cout;
  const int ch = eb == 2 ? 8 : 4;   // channels per 16-byte chunk
  if (xl % ch != 0 || yl % ch != 0 || d->Cin % ch != 0 || d->Cout % ch != 0) return false;
  if (on == 2) return true;
  const int64_t P = (int64_t)d->N * d->Ho * d->Wo;
  if (P < 6144 || d->Cin < 256) return false;
  const int64_t ntn = sg_cdiv(d->Cout, PW_BN), ntr = sg_cdiv(d->Cin, PW_BM);
  return (double)d->Cout / (double)(ntn * PW_BN) >= 0.75 && (double)d->Cin / (double)(ntr * PW_BM) >= 0.75;
}

inline void wgrad_pw_wide_plan(int num_cus, const sg_conv_desc* d, int& S, int& ksteps_per_split, int kp = 16) {
  const int64_t tiles = sg_cdiv(d->Cin, PW_BM) * sg_cdiv(d->Cout, PW_BN);
  const int64_t nks = sg_cdiv((int64_t)d->N * d->Ho * d->Wo, kp);
  int64_t s = (int64_t)num_cus / tiles;
  if (s < 1) s = 1;
  if (s > nks / 8) s = nks / 8 > 0 ? nks / 8 : 1;   // at least 8 k-steps per share
  ksteps_per_split = (int)sg_cdiv(nks, s);
  S = (int)sg_cdiv(nks, ksteps_per_split);
}

inline int launch_wgrad_pw_wide(const WgradParams& p, int S, hipStream_t st) {
  constexpr size_t lds = 2 * (size_t)WPW_STAGE;
  static bool attr_done = false;
  if (!attr_done) {
    int rc = set_dyn_lds(wgrad_pw_wide_kernel<0>, lds);
    if (!rc) rc = set_dyn_lds(wgrad_pw_wide_kernel<1>, lds);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.Cin, PW_BM) * sg_cdiv(p.Cout, PW_BN);
  if (tiles <= 0 || tiles > 65535 || S < 1 || S > 65535 || (int64_t)p.K * p.Cout * 4 >= (1ll << 31)) {
    sg_set_error("wgrad_pw_wide: bad grid (%lld tiles, %d splits)", (long long)tiles, S);
    return SG_EINVAL;
  }
  static const int var = getenv("SG_WPW_VAR") ? atoi(getenv("SG_WPW_VAR")) : 1;
  if (var == 1) hipLaunchKernelGGL(wgrad_pw_wide_kernel<1>, dim3((unsigned)tiles, 1, (unsigned)S), dim3(512), lds, st, p);
  else hipLaunchKernelGGL(wgrad_pw_wide_kernel<0>, dim3((unsigned)tiles, 1, (unsigned)S), dim3(512), lds, st, p);
  SG_LAUNCH_CHECK("wgrad_pw_wide_kernel");
  return 0;
}

inline int launch_wgrad_pw_wide_b16(const WgradParams& p, int S, hipStream_t st) {
  constexpr size_t lds = 2 * (size_t)WPB_STAGE;
  static bool attr_done = false;
  if (!attr_done) {
    int rc = set_dyn_lds(wgrad_pw_wide_b16_kernel, lds);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.Cin, PW_BM) * sg_cdiv(p.Cout, PW_BN);
  if (tiles <= 0 || tiles > 65535 || S < 1 || S > 65535 || (int64_t)p.K * p.Cout * 4 >= (1ll << 31)) {
    sg_set_error("wgrad_pw_wide_b16: bad grid (%lld tiles, %d splits)", (long long)tiles, S);
    return SG_EINVAL;
  }
  hipLaunchKernelGGL(wgrad_pw_wide_b16_kernel, dim3((unsigned)tiles, 1, (unsigned)S), dim3(512), lds, st, p);
  SG_LAUNCH_CHECK("wgrad_pw_wide_b16_kernel");
  return 0;
}
