// fp32 ("x6") convolution with BOTH operands by LDS-DMA from bf16 planes: included by conv_igemm.hip after conv_b16w.h (same
// namespace, IgemmParams).  Round 4: the roofline kernel set's forward and dgrad (the dilated ASPP convolutions).
//
// conv_x6_kernel gathers its A operand as fp32 and splits it into the three bf16 planes on the VALU while it stages it: for a
// dilated 3x3 convolution 2048 -> 256 at 32 x 32 (M = 16384 pixels) every input element is loaded, split and written to LDS
// 9 taps x 2 column tiles = 18 times, and the split + LDS-store phase is as long as the MFMA phase (LAB_NOTEBOOK.md 4.2: of
// 1.03 ms, read + MFMA alone 0.66, split + store 0.39, loads 0.34).  Here the activation is split ONCE per launch into planes
// [3][pixels][C] (x6w_split_kernel: 1.5 x the bytes of x, ~75 us for the 128 MiB map) and the convolution kernel does no
// arithmetic on its operands at all - conv_b16w_kernel's structure with three planes per operand:
//   * 128 x 256 tile per workgroup, one workgroup per CU, 8 waves as 2 x 4 (64 x 64 per wave): per 16-deep k-step 12 fragment
//     reads for 24 MFMAs (the six x6 terms of four 32 x 32 blocks, smallest first);
//   * stages of 32 reduction channels: A 3 x 128 x 64 B + B 3 x 256 x 64 B = 72 KB, two stages, ONE barrier per stage; a stage is
//     72 LDS-DMA pieces of 1 KB (16 rows x 64 B): a wave issues 3 A pieces (the three planes of its 16 rows: one gather offset
//     per lane, the plane in the scalar offset) and 6 B pieces;
//   * the weight planes are conv_x6_kernel's own ([3][K / 32][Npad][32], prepared once per step);
//   * few tiles (the ASPP forward: 128) -> the K walk is cut in two shares, fp32 partial slabs behind the planes, summed (with
//     bias / ReLU and the BatchNormalization statistics per 128-row tile) by b16w_reduce_kernel<float>.  Shares and kernel
//     choice depend on one image's geometry only (x6w_plan).
// Order of additions of an output element: stages in the walk's order (padding taps skipped, channel-block order as in
// conv_x6_kernel), inside a stage two 16-deep k-steps, inside a k-step the six terms a3b1, a1b3, a2b2, a2b1, a1b2, a1b1 into one
// fp32 accumulator; shares added in order.  Deterministic and independent of the batch.
#pragma once

constexpr int XW_M = 128, XW_N = 256, XW_KD = 32, XW_RB = 64;
constexpr int XW_A = 3 * XW_M * XW_RB;                    // 24576
constexpr int XW_B = 3 * XW_N * XW_RB;                    // 49152
constexpr int XW_STAGE = XW_A + XW_B;                     // 73728
constexpr int XW_LDS = 2 * XW_STAGE;

__device__ __forceinline__ int xw_swz(int row) { return (row >> 2) & 3; }   // 64-byte rows: B16L<2>::swz

// x [rows][x_ld] fp32 (C channels used) -> planes [3][rows][C] bf16 (h, m, l of split3_pair); a thread takes 4 channels
__global__ __launch_bounds__(256) void x6w_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ planes, int64_t rows,
                                                        int C, int x_ld) {
  const int cq = C / 4;
  const int64_t total = rows * cq, plane = rows * (int64_t)C;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / cq;
    const int c = (int)(i - r * cq) * 4;
    const f32x4 f = *reinterpret_cast<const f32x4*>(x + r * x_ld + c);
    unsigned h0, m0, l0, h1, m1, l1;
    split3_pair(f[0], f[1], h0, m0, l0);
    split3_pair(f[2], f[3], h1, m1, l1);
    unsigned short* dst = planes + r * C + c;
    *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){h0, h1};
    *reinterpret_cast<u32x2_t*>(dst + plane) = (u32x2_t){m0, m1};
    *reinterpret_cast<u32x2_t*>(dst + 2 * plane) = (u32x2_t){l0, l1};
  }
}

// VAR (SG_X6W_VAR): 0 = wait + barrier at the END of a stage, DMA of the next stage at its top (the MFMA pipe idles from the last
// MFMA of a stage through barrier, DMA issue and the first fragment reads of the next: the timing ablations of
// LAB_NOTEBOOK.md 11.9 put that skeleton at 0.49 of the forward's 0.84 ms); 1 = the barrier in the MIDDLE of a stage - stage
// s + 1 is published between the two k-steps of stage s, its first fragments are read under the MFMAs of k-step 1, the DMA of
// stage s + 2 goes into the buffer just released: the pipe only waits for the barrier itself.
template <int VAR>
__global__ __launch_bounds__(512, 2) void conv_x6w_kernel(const IgemmParams p, const unsigned short* __restrict__ aq,
                                                          const uint32_t aq_bytes, const uint32_t a_plane_bytes, const int S,
                                                          float* __restrict__ part) {
  constexpr unsigned OOB = 0x80000000u;
  constexpr int WGN = 4, WM = 64, WN = 64, TM = 2, TN = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages] x { A [3][128][64 B], B [3][256][64 B] }
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const uint32_t ntn = (p.Nout + XW_N - 1) / XW_N;
  const uint32_t ntiles = gridDim.x / (uint32_t)S;
  const uint32_t o = xcd_remap(blockIdx.x, gridDim.x);
  const uint32_t split = o / ntiles, bid = o - split * ntiles;
  // grouped order as conv_x6_kernel's (row tile fastest inside groups of group_m row tiles) when there are several column tiles
  uint32_t tile_m, tile_n;
  if (p.group_m > 1) {
    const uint32_t ntm = ntiles / ntn, gm = (uint32_t)p.group_m;
    const uint32_t per = gm * ntn, g = bid / per, r = bid - g * per;
    const uint32_t left = ntm - g * gm, gsz = left < gm ? left : gm;
    tile_n = r / gsz;
    tile_m = g * gm + (r - tile_n * gsz);
  } else {
    tile_m = bid / ntn;
    tile_n = bid - tile_m * ntn;
  }
  const int m0 = tile_m * XW_M, n0 = tile_n * XW_N;

  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(aq), 0, (int)aq_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wq), 0, (int)p.w_bytes, 0x00020000);

  // ---- this lane's DMA pieces: A = the three planes of row block `wave` (one row per lane); B = pieces wave + 8 j, j < 6 ------
  const int a_row = wave * 16 + (lane >> 2);
  const int a_chunk = 16 * ((lane & 3) ^ xw_swz(a_row));
  int a_lin = 0, a_hw = (int)0x80008000u;
  {
    const int m = m0 + a_row;
    if (m < p.M) {
      uint32_t n, oh, ow;
      row_to_pixel(p, (uint32_t)m, n, oh, ow);
      const int ohs = (int)oh * p.a_mul + p.off_h, ows = (int)ow * p.a_mul + p.off_w;
      a_lin = (int)n * p.H * p.W + ohs * p.W + ows;
      a_hw = (ohs << 16) | (ows & 0xffff);
    }
  }
  unsigned b_voff[6];
  const int kblocks = p.Kpad >> 5;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int idx = wave + 8 * j;            // 0 .. 47
    const int pl = idx >> 4, rb = idx & 15;
    const int row = rb * 16 + (lane >> 2);
    const int c = (lane & 3) ^ xw_swz(row);
    const int nrow = n0 + row;
    b_voff[j] = nrow < p.Npad ? (unsigned)((((int64_t)pl * kblocks) * p.Npad + nrow) * XW_RB + 16 * c) : OOB;
  }
  auto tap_valid = [&](int dh, int dw, int& pix) -> bool {
    const int ohs = a_hw >> 16, ows = (int)(short)(a_hw & 0xffff);
    const int ih = ohs + dh, iw = ows + dw;
    pix = a_lin + dh * p.W + dw;
    return ((unsigned)ih < (unsigned)p.H) && ((unsigned)iw < (unsigned)p.W);
  };

  // ---- the K walk (conv_b16w_kernel's): active taps, channel-block order, this share's range ------------------------------------
  // The active taps are a 64-bit mask in SGPRs (x6w_plan: at most 64 taps), the position of the walk is (channel block, tap, slab
  // in the block), advanced with scalar selects only: nothing in the stage loop branches, reads LDS or divides for the walk.
  const int ntaps = p.K / p.C;
  const int spt = p.C / XW_KD;
  uint64_t tapmask = ntaps >= 64 ? ~0ull : ((1ull << ntaps) - 1);
  if (p.skip_taps) {   // uniform
    tapmask = 0;
    for (int tap = 0; tap < ntaps; ++tap) {
      uint32_t kh, kw;
      fd_divmod((uint32_t)tap, p.fd_kw, kh, kw);
      int pix;
      const bool any = tap_valid((int)kh * p.k_mul, (int)kw * p.k_mul, pix);
      if (__syncthreads_or(any ? 1 : 0)) tapmask |= 1ull << tap;
    }
  }
  // (a vote's result counts as divergent for the compiler: say it is uniform, or every LDS-DMA below becomes a waterfall loop)
  tapmask = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(tapmask >> 32)) << 32) |
            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tapmask);
  const int nact = __builtin_popcountll(tapmask);
  const int nstage = nact * spt;
  int it_run = spt;
  if (p.cb > 0) {
    it_run = p.cb;   // (p.cb counts 32-deep slabs = stages)
    if (it_run < 1) it_run = 1;
    if (spt % it_run != 0) it_run = spt;
  }
  const int per = it_run * nact;
  const int st_begin = __builtin_amdgcn_readfirstlane((int)(((int64_t)split * nstage) / S));
  const int st_end = __builtin_amdgcn_readfirstlane((int)(((int64_t)(split + 1) * nstage) / S));
  const int nk = nact > 0 ? st_end - st_begin : 0;
  auto tap_after = [&](int tap, bool& wrapped) -> int {   // the next active tap behind `tap` (cyclic)
    const uint64_t above = tap >= 63 ? 0ull : (tapmask & ~((2ull << tap) - 1));
    wrapped = above == 0;
    return __builtin_ctzll(wrapped ? tapmask : above);
  };
  int it_cb = 0, it_tap = 0, it_ci = 0;
  if (nk > 0) {
    it_cb = st_begin / per;
    const int rem = st_begin - it_cb * per;
    const int ti = rem / it_run;
    it_ci = rem - ti * it_run;
    it_tap = __builtin_ctzll(tapmask);
    for (int i = 0; i < ti; ++i) {
      bool w;
      it_tap = tap_after(it_tap, w);
    }
    it_cb = __builtin_amdgcn_readfirstlane(it_cb);
    it_tap = __builtin_amdgcn_readfirstlane(it_tap);
    it_ci = __builtin_amdgcn_readfirstlane(it_ci);
  }

  // the DMA of the NEXT stage of the walk into LDS buffer `stage`; !live: a stage behind the share's end (every piece out of range:
  // zeros into a buffer nobody reads again - cheaper than a branch that would cut the loop body into scheduling regions)
  auto issue = [&](bool first, int stage, bool live) {
    const int tap = it_tap, slab = it_cb * it_run + it_ci;
    {   // advance
      const bool run_end = it_ci + 1 == it_run;
      bool wrapped;
      const int nt = tap_after(it_tap, wrapped);
      it_ci = run_end ? 0 : it_ci + 1;
      it_tap = run_end ? nt : it_tap;
      it_cb = (run_end && wrapped) ? it_cb + 1 : it_cb;
    }
    uint32_t kh, kw;
    fd_divmod((uint32_t)tap, p.fd_kw, kh, kw);
    int pix;
    const bool ok = tap_valid((int)kh * p.k_mul, (int)kw * p.k_mul, pix);
    const unsigned a_voff = (ok && live) ? (unsigned)pix * (unsigned)p.C * 2u + (unsigned)a_chunk : OOB;   // dense planes: pitch C
    const int soff_a = slab * (XW_KD * 2);
    const int soff_b = (tap * spt + slab) * p.Npad * XW_RB;
    char* sa = smem + stage * XW_STAGE;
    char* sb = sa + XW_A;
    // experiment build only (-DSG_X6W_ABL; timing diagnostics, results wrong; SG_X6W_ABLATE): 1 = no A DMA behind the first stage,
    // 2 = no B DMA behind the first stage, 4 = no MFMAs (the fragment reads stay).  A run-time flag here would put a branch in front
    // of every MFMA of the product build.
#ifdef SG_X6W_ABL
    const int abl = p.ablate;
#else
    constexpr int abl = 0;
#endif
    if (!((abl & 1) && !first)) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        pw_lds_dma16(rsrc_a, sa + pl * (XW_M * XW_RB) + wave * 1024, a_voff, soff_a + pl * (int)a_plane_bytes);
    }
    if (!((abl & 2) && !first)) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int idx = wave + 8 * j;
        pw_lds_dma16(rsrc_w, sb + (idx >> 4) * (XW_N * XW_RB) + (idx & 15) * 1024, live ? b_voff[j] : OOB, soff_b);
      }
    }
  };

  // ---- MFMA side -----------------------------------------------------------------------------------------------------------
  const int wr = wave / WGN, wc = wave % WGN;
  const int wm = wr * WM, wn = wc * WN;
  const int lr = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int sw = xw_swz(lr);
  const int a_lane = (wm + lr) * XW_RB, b_lane = XW_A + (wn + lr) * XW_RB;
  auto frags = [&](int stage, int ks, bf16x8_t (&a)[TM][3], bf16x8_t (&b)[TN][3]) {
    const char* sbuf = smem + stage * XW_STAGE;
    const int ko = ((2 * ks + lh) ^ sw) << 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        a[i][pl] = *reinterpret_cast<const bf16x8_t*>(sbuf + a_lane + pl * (XW_M * XW_RB) + 32 * i * XW_RB + ko);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        b[j][pl] = *reinterpret_cast<const bf16x8_t*>(sbuf + b_lane + pl * (XW_N * XW_RB) + 32 * j * XW_RB + ko);
  };
  auto mfmas = [&](const bf16x8_t (&a)[TM][3], const bf16x8_t (&b)[TN][3]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first (conv_x6_kernel's order)
#pragma unroll
        for (int u = 0; u < 6; ++u) {
#ifdef SG_X6W_ABL
          if (p.ablate & 4) {
            asm volatile("" ::"v"(a[i][PA_[u]]), "v"(b[j][PB_[u]]));
            continue;
          }
#endif
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA_[u]], b[j][PB_[u]], acc[i][j], 0, 0, 0);
        }
      }
  };
  bf16x8_t af0[TM][3], bf0[TN][3], af1[TM][3], bf1[TN][3];

  if (nk > 0) {
    issue(true, 0, true);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (VAR == 0) {
      for (int s = 0; s < nk; ++s) {
        const int cur = s & 1;
        if (s + 1 < nk) {   // stage cur ^ 1 was last read before the previous barrier
          issue(false, cur ^ 1, true);
          __builtin_amdgcn_sched_barrier(0);
        }
        frags(cur, 0, af0, bf0);
        frags(cur, 1, af1, bf1);   // one k-step of look-ahead
        mfmas(af0, bf0);
        mfmas(af1, bf1);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    } else {
      // One basic block per stage, woven by hand (sched_group_barrier: 0x008 MFMA, 0x020 VMEM read, 0x100 DS read):
      //   A: the 24 MFMAs of k-step 0 with the 12 fragment reads of k-step 1 between them;
      //   wait + barrier: stage s + 1 (its DMA went out one whole stage ago) is published, stage s is free;
      //   B: the 24 MFMAs of k-step 1 with the 12 fragment reads of k-step 0 of stage s + 1 and the 9 DMA pieces of stage s + 2
      //      (into the buffer of stage s) between them.
      issue(false, 1, nk > 1);
      frags(0, 0, af0, bf0);
      for (int s = 0; s < nk; ++s) {
        const int cur = s & 1;
        __builtin_amdgcn_sched_barrier(0);
        frags(cur, 1, af1, bf1);
        mfmas(af0, bf0);
#pragma unroll
        for (int g = 0; g < 12; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        frags(cur ^ 1, 0, af0, bf0);   // (behind the last stage: stale bytes nobody uses)
        issue(false, cur, s + 2 < nk);   // (program order = the order the scheduler keeps between LDS reads and LDS-DMA writes)
        mfmas(af1, bf1);
#pragma unroll
        for (int g = 0; g < 12; ++g) {   // the reads first: the next stage's first MFMA waits for the last of them
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int g = 0; g < 9; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last (all out of range) DMA still writes zeros into LDS
    }
  }

  // ---- output: the fp32 tile (S == 1: bias / ReLU applied, into y; S > 1: this share's partial slab), branch-free buffer stores --
  const bool whole = S == 1;
  const bool has_bias = whole && (p.flags & SG_EPI_BIAS) != 0;
  const bool do_relu = whole && (p.flags & SG_EPI_RELU) != 0;
  {
    float* dst = whole ? p.y : part + (int64_t)split * ((int64_t)p.M * p.Nout);
    const int ld = whole ? p.y_ld : p.Nout;
    const uint32_t obytes = (uint32_t)((((int64_t)p.M - 1) * ld + p.Nout) * 4);   // < 2^31: launch_x6w
    const __amdgpu_buffer_rsrc_t rsrc_o = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)obytes, 0x00020000);
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
          unsigned voff = cterm + row * (unsigned)ld * 4u;
          voff = row < (unsigned)p.M ? voff : OOB;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_o, (int)voff, 0, 0);
        }
    }
  }

  // ---- BatchNormalization statistics of this 128-row tile (whole-K launches; conv_x6_kernel's scheme and layout) ---------------
  if (whole && p.stats) {
    float* red = reinterpret_cast<float*>(smem);   // [2][XW_N] partials, then [XW_N] tile means
    float* tmean = red + 2 * XW_N;
    const int nvalid = (p.M - m0) < XW_M ? (p.M - m0) : XW_M;
    const bool hb = (p.flags & SG_EPI_BIAS) != 0;
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int cl = wn + 32 * j + lr;
        const int col = n0 + cl;
        const float bv = (hb && col < p.Nout) ? p.bias[col] : 0.f;
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
        sacc += __shfl_xor(sacc, 32, 64);
        if (lh == 0) red[wr * XW_N + cl] = sacc;
      }
      __syncthreads();
      for (int cl = t; cl < XW_N; cl += 512) {
        const float tot = red[cl] + red[XW_N + cl];
        const int col = n0 + cl;
        if (pass == 0) tmean[cl] = tot / (float)nvalid;
        if (col < p.Nout) p.stats[((int64_t)tile_m * 2 + pass) * p.Nout + col] = tot;
      }
      __syncthreads();
    }
  }
}

// Which fp32 launches take it, and in how many K shares (0: none): the multi-tap convolutions with a long reduction (K >= 2048)
// and >= 192 output columns - the ASPP forward (two shares) and dgrad (whole), the 3x3 convolutions of the decoder at 256 / 512
// channels - from ONE image's geometry.  SG_X6_WIDE: 0 off, 1 the dilated ones only (round 4's first form), 2 (default) all of them
// (step -0.26 ms on alternating runs, profiles/r04_ab_runs.txt block r4t).
inline int x6w_plan(const IgemmParams& p) {
  static const int on = getenv("SG_X6_WIDE") ? atoi(getenv("SG_X6_WIDE")) : 2;
  if (!on) return 0;
  if (p.div != 1 || p.perm2 || p.res || p.C % XW_KD != 0 || p.K == p.C || p.K < 2048 || p.K / p.C > 64) return 0;
  if (on < 2 && !(p.k_mul > 1 || p.k_mul < -1)) return 0;   // 1: dilated taps only
  if (p.x_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(p.x) & 15) != 0 || p.x_bytes == 0 || p.Nout % 4 != 0) return 0;
  const int64_t ntn = sg_cdiv(p.Nout, XW_N);
  if ((double)p.Nout / (double)(ntn * XW_N) < 0.75) return 0;
  const int64_t img_px = (int64_t)p.OH * p.OW;
  if (img_px < XW_M || img_px % XW_M != 0) return 0;
  const int64_t img_wgs = ntn * (img_px / XW_M);
  if (img_wgs >= 16) return 1;
  if (img_wgs >= 8 && p.K / 2 >= 2048) return 2;
  return 0;
}

// planes of the A operand (3 x pixels x C bf16, 256-byte aligned) + partial slabs of a split-K launch
inline size_t x6w_a_planes_bytes(const IgemmParams& p) {
  const int64_t batch = p.M / ((int64_t)p.OH * p.OW);
  return (((size_t)3 * (size_t)batch * p.H * p.W * p.C * 2) + 255) & ~(size_t)255;
}
inline size_t x6w_scratch_bytes(const IgemmParams& p, int S) {
  return x6w_a_planes_bytes(p) + (S > 1 ? (size_t)S * (size_t)p.M * p.Nout * sizeof(float) : 0);
}

// scratch: [A planes][partial slabs]
inline int launch_x6w(const IgemmParams& p, int S, char* scratch, hipStream_t st) {
  static bool attr_done = false;
  if (!attr_done) {
    int rc = set_dyn_lds(conv_x6w_kernel<0>, (size_t)XW_LDS);
    if (!rc) rc = set_dyn_lds(conv_x6w_kernel<1>, (size_t)XW_LDS);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.M, XW_M) * sg_cdiv(p.Nout, XW_N);
  const size_t a_bytes = x6w_a_planes_bytes(p);
  const int64_t batch = p.M / ((int64_t)p.OH * p.OW), rows = batch * p.H * p.W;
  const int64_t plane = rows * p.C * 2;
  if (tiles <= 0 || tiles * S > 0x7fffffff || !scratch || 3 * plane >= (1ll << 31) || (((int64_t)p.M - 1) * p.y_ld + p.Nout) * 4 >= (1ll << 31) ||
      (int64_t)p.M * p.Nout * 4 >= (1ll << 31)) {
    sg_set_error("conv_x6w: bad launch (%lld tiles, %d shares) or an operand beyond one 2 GiB buffer descriptor", (long long)tiles, S);
    return SG_EINVAL;
  }
  // the activation planes: the caller's (sg_conv2d_fwd_stats_ap / _dgrad_ap: made once by sg_split_planes for every consumer of
  // the tensor and for its filter gradient) or, without them, split here into the scratch
  const unsigned short* aq = p.a_planes;
  if (!aq) {
    int64_t blocks = sg_cdiv(rows * (p.C / 4), 256);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(x6w_split_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p.x, (unsigned short*)scratch, rows, p.C, p.x_ld);
    SG_LAUNCH_CHECK("x6w_split_kernel");
    aq = (const unsigned short*)scratch;
  }
  float* part = S > 1 ? reinterpret_cast<float*>(scratch + a_bytes) : nullptr;
  static const int var = getenv("SG_X6W_VAR") ? atoi(getenv("SG_X6W_VAR")) : 1;
  if (var == 1)
    hipLaunchKernelGGL(conv_x6w_kernel<1>, dim3((unsigned)(tiles * S)), dim3(512), (size_t)XW_LDS, st, p, aq,
                       (uint32_t)(3 * plane), (uint32_t)plane, S, part);
  else
    hipLaunchKernelGGL(conv_x6w_kernel<0>, dim3((unsigned)(tiles * S)), dim3(512), (size_t)XW_LDS, st, p, aq,
                       (uint32_t)(3 * plane), (uint32_t)plane, S, part);
  SG_LAUNCH_CHECK("conv_x6w_kernel");
  if (S > 1) {
    dim3 grid((unsigned)sg_cdiv(p.M, 128), (unsigned)sg_cdiv(p.Nout, 64));
    hipLaunchKernelGGL(b16w_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)part, (p.flags & SG_EPI_BIAS) ? p.bias : nullptr,
                       p.y, p.stats, p.M, p.Nout, p.y_ld, S, (p.flags & SG_EPI_RELU) ? 1 : 0);
    SG_LAUNCH_CHECK("b16w_reduce_kernel<float>");
  }
  return 0;
}
