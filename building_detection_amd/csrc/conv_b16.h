// bf16-storage convolution with DEEP slabs: included by conv_igemm.hip after conv_x6.h (same namespace, same IgemmParams).
//
// conv_x6_kernel<NPL = 1, bf16> inherits the x6 kernel's 32-deep K slab: right for six MFMA passes per product (24 MFMAs per
// wave and slab), but with ONE pass a slab is 4 MFMAs per wave (128 matrix-pipe cycles) between two barriers, one LDS
// write -> read turn-around and one round of global loads: the fixed cost per slab (~900 cycles measured: the bf16 dilated
// set runs only 3.1x faster than the six-pass one for a sixth of the MFMAs) then dominates.  This kernel makes the slab
// 16 * KS deep (KS = 2, 4, 8 k-steps: 32, 64, 128 channels), so the same fixed cost buys up to four times the matrix work:
//   * LDS rows are the 32 * KS bytes of one row's k range; the 16-byte chunks of row r sit at slot chunk ^ swz(r) with
//     swz(r) = (r >> log2(256 / row bytes)) & (chunks - 1): the 16 lanes of a ds_read_b128 group hit 16 distinct bank
//     groups for every k-step (KS = 2 reproduces conv_x6.h's layout);
//   * a thread stages KS/2 A chunks and KS/2 B chunks per slab (16 bytes = 8 channels each, straight from the buffer load
//     to LDS: bf16 needs no conversion), prefetched into registers one slab ahead;
//   * the fragments of k-step s+1 are read while the MFMAs of k-step s issue (one k-step of look-ahead inside the slab);
//   * single LDS buffer, two workgroups per CU (64 KB at KS = 8, BN = 128): while one workgroup waits at its barriers and
//     stages, the other multiplies.
// Gather, padding-tap elimination, tile orders, epilogue and the BatchNormalization statistics are conv_x6_kernel's.
// Requires every slab inside one tap: C % (16 KS) == 0, a 1x1 kernel, or virtual channel padding to that depth.
#pragma once

template <int KS>
struct B16L {
  static constexpr int BKB = 16 * KS;           // channels per slab
  static constexpr int RB = 2 * BKB;            // bytes per LDS row
  static constexpr int CPR = RB / 16;           // 16-byte chunks per row
  static constexpr int WSH = (RB == 64) ? 2 : (RB == 128 ? 1 : 0);  // log2(rows per 256 bytes)
  __device__ static __forceinline__ int swz(int row) { return (row >> WSH) & (CPR - 1); }
};

// PD = register sets of prefetched slabs (1 or 2; SG_B16_PD, default 1).  Two sets (the loads of slab s + 2 in flight as
// well) were built on the theory that short-K layers run at the memory latency; measured (profiles/r02_b16_prefetch_ab.txt)
// they help the ASPP forward (308 -> 280 us) and cost everywhere else (728 -> 728: 38 -> 41 us, 304 -> 256 3x3: 528 -> 578 us,
// occupancy 6 -> 4 waves per SIMD).  The timing ablations of the same file say why: of the 728 -> 728 layer's 40 us, 12.5 are
// the two launches and the prologue, 3 the epilogue, and the K loop's 24 us are ~15 us of operand delivery alone (286 MB from
// L2 to the CUs: A six times, the weights 128 times) - it is bound by the L2 -> CU path, not by latency or the matrix pipe.
template <int BN, int WGM, int WGN, int KS, int PD = 1>
__global__ __launch_bounds__(64 * WGM * WGN, WGM * WGN / 2) void conv_b16_kernel(const IgemmParams p) {
  using L = B16L<KS>;
  constexpr int BKB = L::BKB, RB = L::RB, CPR = L::CPR;
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(NT % CPR == 0 && (BM * CPR) % NT == 0, "thread layout");
  constexpr int NA = (BM * CPR) / NT;       // A chunks per thread: rows r0 + RS * j
  constexpr int RS = NT / CPR;
  constexpr int NBC = BN * CPR;             // B chunks of a slab
  constexpr int NB = (NBC + NT - 1) / NT;
  static_assert(TM >= 1 && TN >= 1 && NA >= 1, "tile too small for the wave layout");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ap = smem;                                    // [BM][RB]
  char* Bp = Ap + BM * RB;                            // [BN][RB]
  int* tapinfo = reinterpret_cast<int*>(smem + (BM + BN) * RB);  // [64]
  int* row_lin_lds = tapinfo + 64;                    // [NA][NT]

  const int t = threadIdx.x;
  const uint32_t ntn = (p.Nout + BN - 1) / BN;
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

  // ---- A rows --------------------------------------------------------------------------------------------------------
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
    const int row = idx / CPR, c = idx % CPR;
    b_voff[i] = idx < NBC ? (unsigned)(((n0 + row) * BKB + c * 8) * 2) : OOB;   // k-block-major planes [Kpad / BKB][Npad][BKB]
  }
  const int ntaps = p.K / p.C;
  const bool ktail = (p.K % BKB) != 0;
  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.wq), 0, (int)p.w_bytes, 0x00020000);

  u32x4_t ra[PD][NA], rb[PD][NB];
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
        tap_voff[j] = ok ? (unsigned)off * 2u + 16u * kc : OOB;
      }
    }
    const int soff_a = (k0 - tap * p.C) * 2;
    const bool kvalid = !ktail || (k0 + 8 * kc < p.K);
#pragma unroll
    for (int j = 0; j < NA; ++j)
      ra[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(kvalid ? tap_voff[j] : OOB), soff_a, 0);
    const int soff_b = (k0 / BKB) * p.Npad * BKB * 2;
#pragma unroll
    for (int i = 0; i < NB; ++i) rb[S][i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, (int)b_voff[i], soff_b, 0);
  };
  auto store_AB = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int arow = r0 + RS * j;
      *reinterpret_cast<u32x4_t*>(Ap + arow * RB + ((kc ^ L::swz(arow)) << 4)) = ra[S][j];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + NT * i;
      if ((NBC % NT == 0) || idx < NBC) {
        const int row = idx / CPR, c = idx % CPR;
        *reinterpret_cast<u32x4_t*>(Bp + row * RB + ((c ^ L::swz(row)) << 4)) = rb[S][i];
      }
    }
  };

  // ---- MFMA side -----------------------------------------------------------------------------------------------------
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
  const char* a_lane = Ap + (wm + lr) * RB;
  const char* b_lane = Bp + (wn + lr) * RB;
  const int sw = L::swz(lr);  // wm, wn and the 32-row sub-tile offsets are multiples of 32: the swizzle depends on lr only

  auto frags = [&](int ks, bf16x8_t (&af)[TM], bf16x8_t (&bf)[TN]) {
    const int ko = ((2 * ks + lh) ^ sw) << 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(a_lane + (32 * i) * RB + ko);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8_t*>(b_lane + (32 * j) * RB + ko);
  };
  auto compute = [&]() {
    bf16x8_t af[2][TM], bf[2][TN];
    frags(0, af[0], bf[0]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) frags(ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);  // one k-step of look-ahead
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][i], bf[ks & 1][j], acc[i][j], 0, 0, 0);
    }
  };

  // ---- slab stream ----------------------------------------------------------------------------------------------------
  const int spt = p.C / BKB;
  int nslab = (p.K + BKB - 1) / BKB;
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
  // channel-block order (p.cb, in 32-deep slabs): the taps of one block of cb * 32 channels come together
  int it_run = spt;
  if (p.cb > 0) {
    it_run = (p.cb * 32) / BKB;
    if (it_run < 1) it_run = 1;
    if (spt % it_run != 0) it_run = spt;
  }
  int it_ci = 0, it_ti = 0, it_cb = 0, it_s = 0;
  auto next_k0 = [&]() -> int {
    int k0;
    if (it_lin) {
      k0 = it_s * BKB;
    } else {
      const int tap = use_map ? __builtin_amdgcn_readfirstlane(tapinfo[it_ti]) : it_ti;
      k0 = tap * p.C + (it_cb * it_run + it_ci) * BKB;
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
    if constexpr (PD == 1) {
      load_AB(next_k0(), IC<0>{});
#ifdef SG_B16_ABL  // experiment build (make ABL=1): timing-only ablations, results wrong; p.ablate from SG_B16_ABLATE
      const bool ab_ld = (p.ablate & 1) != 0, ab_st = (p.ablate & 2) != 0, ab_mm = (p.ablate & 4) != 0;
      for (int s = 0; s < nslab; ++s) {
        if (!ab_st) {
          __syncthreads();
          store_AB(IC<0>{});
          __syncthreads();
        }
        if (!ab_ld) load_AB(next_k0(), IC<0>{});
        if (!ab_mm) compute();
      }
#else
      for (int s = 0; s < nslab; ++s) {
        __syncthreads();  // every wave has finished reading the previous slab
        store_AB(IC<0>{});
        __syncthreads();
        load_AB(next_k0(), IC<0>{});  // the tail reloads the last slab (unused): no branch
        compute();
      }
#endif
    } else {
      load_AB(next_k0(), IC<0>{});
      load_AB(next_k0(), IC<1>{});
      for (int s = 0; s < nslab; s += 2) {
        __syncthreads();
        store_AB(IC<0>{});
        __syncthreads();
        load_AB(next_k0(), IC<0>{});  // slab s + 2 (the tail reloads the last slab, unused)
        compute();
        if (s + 1 >= nslab) break;
        __syncthreads();
        store_AB(IC<1>{});
        __syncthreads();
        load_AB(next_k0(), IC<1>{});
        compute();
      }
    }
  }

  // ---- epilogue: bias / ReLU, rounding to bf16; the tile goes through LDS so that it leaves as 16-byte row chunks (a
  // bf16 accumulator fragment is one 2-byte element per lane and store instruction: 32 store instructions per wave, and the
  // tail of a short-K launch is store-ISSUE bound) -----------------------------------------------------------------------
  const bool has_bias = (p.flags & SG_EPI_BIAS) != 0;
  const bool do_relu = (p.flags & SG_EPI_RELU) != 0;
  bf16_t* __restrict__ py = reinterpret_cast<bf16_t*>(p.y);
  constexpr int TP = BN * 2 + 16;  // LDS pitch of the staged tile: consecutive rows 16 bytes apart modulo 256
  const bool wide = (p.y_ld % 8 == 0) && (p.Nout % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0);
  if (wide) {
    __syncthreads();  // the slab buffers are free
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
    constexpr int CPT = BN / 8;  // 16-byte chunks per tile row
#ifdef SG_B16_ABL
    if (!(p.ablate & 8))
#endif
    for (int idx = t; idx < BM * CPT; idx += NT) {
      const int rl = idx / CPT, c = idx - rl * CPT;
      const int row = m0 + rl, col = n0 + 8 * c;
      if (row < p.M && col < p.Nout) {
        const int64_t yo = row_to_yoff(p, row) + col;
        u32x4_t o = *reinterpret_cast<const u32x4_t*>(smem + rl * TP + c * 16);
        if (p.res) {   // uniform: a collected gradient rides along (the tile's bf16 values + res in fp32, rounded again)
          f32x4 a0 = {__uint_as_float(o[0] << 16), __uint_as_float(o[0] & 0xffff0000u), __uint_as_float(o[1] << 16),
                      __uint_as_float(o[1] & 0xffff0000u)};
          f32x4 a1 = {__uint_as_float(o[2] << 16), __uint_as_float(o[2] & 0xffff0000u), __uint_as_float(o[3] << 16),
                      __uint_as_float(o[3] & 0xffff0000u)};
          f32x4 b0, b1;
          ld8_bf16(reinterpret_cast<const bf16_t*>(p.res) + yo, b0, b1);
#pragma unroll
          for (int e = 0; e < 4; ++e) { a0[e] += b0[e]; a1[e] += b1[e]; }
          st8_bf16(py + yo, a0, a1);
        } else {
          *reinterpret_cast<u32x4_t*>(py + yo) = o;
        }
      }
    }
  } else {
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
            const int64_t yo = row_to_yoff(p, row) + col;
            if (p.res) v += ld1<bf16_t>(reinterpret_cast<const bf16_t*>(p.res) + yo);
            st1<bf16_t>(py + yo, v);
          }
        }
      }
    }
  }

  if (p.stats) {  // per-tile BatchNormalization statistics, as in conv_x6_kernel
    float* red = reinterpret_cast<float*>(smem);  // [WGM][BN] partials, then [BN] tile means
    float* tmean = red + WGM * BN;
    const int wrow = wave / WGN;
    const int nvalid = (p.M - m0) < BM ? (p.M - m0) : BM;
    __syncthreads();
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
        sacc += __shfl_xor(sacc, 32, 64);
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

template <int BN, int WGM, int WGN, int KS, int PD>
int launch_b16_pd(const IgemmParams& p, hipStream_t st) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr size_t slab_lds = (size_t)(BM + BN) * B16L<KS>::RB;
  constexpr size_t stat_lds = (size_t)(WGM + 1) * BN * sizeof(float);
  constexpr size_t tile_lds = (size_t)BM * (BN * 2 + 16);  // the staged output tile of the epilogue
  constexpr size_t m1 = slab_lds > stat_lds ? slab_lds : stat_lds;
  constexpr size_t lds = (m1 > tile_lds ? m1 : tile_lds) + 256 + (size_t)BM * B16L<KS>::CPR * sizeof(int);
  (void)NT;
  static bool attr_done = false;
  if (!attr_done) {
    int rc = set_dyn_lds(conv_b16_kernel<BN, WGM, WGN, KS, PD>, lds);
    if (rc) return rc;
    attr_done = true;
  }
  const int64_t tiles = sg_cdiv(p.M, BM) * sg_cdiv(p.Nout, BN);
  if (tiles <= 0 || tiles > 0x7fffffff) {
    sg_set_error("conv_b16: bad tile count %lld", (long long)tiles);
    return SG_EINVAL;
  }
  hipLaunchKernelGGL((conv_b16_kernel<BN, WGM, WGN, KS, PD>), dim3((unsigned)tiles), dim3(64 * WGM * WGN), lds, st, p);
  SG_LAUNCH_CHECK("conv_b16_kernel");
  return 0;
}

// SG_B16_PD = 1 / 2: one / two register sets of prefetched slabs for every launch (A/B switch); default 0 = by shape: two
// sets for the long-K multi-tap FORWARD launches (the ASPP convolutions, K = 18432: 308 -> 280 us, 225 -> 195 us), one
// everywhere else (the same convolutions' dgrad loses 10 - 18 % with two, the short-K layers 8 %: profiles/r02_b16_prefetch_ab.txt)
template <int BN, int WGM, int WGN, int KS>
int launch_b16(const IgemmParams& p, hipStream_t st) {
  static const int pd = getenv("SG_B16_PD") ? atoi(getenv("SG_B16_PD")) : 0;
  const bool two = pd == 2 || (pd == 0 && p.k_mul > 0 && p.K != p.C && p.K >= 8192 && p.div == 1);
  return two ? launch_b16_pd<BN, WGM, WGN, KS, 2>(p, st) : launch_b16_pd<BN, WGM, WGN, KS, 1>(p, st);
}

// slab depth (k-steps of 16) for a launch whose taps are `c` channels deep: the deepest of 8 / 4 / 2 that keeps every slab
// inside one tap; 1x1 kernels (one tap, ragged tail masked) take the deepest.  SG_B16_KS forces a value (A/B runs).
inline int b16_ks(int c, bool one_tap) {
  static int force = -1;
  if (force < 0) force = getenv("SG_B16_KS") ? atoi(getenv("SG_B16_KS")) : 0;
  for (int ks : {4, 2}) {  // KS = 8 measured slower than 4 on most shapes (profiles/r02_b16_deep_ab.txt): not dispatched
    if (force && ks > force) continue;
    if (one_tap || c % (16 * ks) == 0) return ks;
  }
  return 2;
}

inline int dispatch_b16(const IgemmParams& p_in, int num_cus, hipStream_t st) {
  IgemmParams p = p_in;
  const int bn = pick_bn(p.M, p.Nout, num_cus);
  plan_common(p, true, bn, true, 2);
  p.stagger = 0;
  p.ablate = 0;
#ifdef SG_B16_ABL
  p.ablate = getenv("SG_B16_ABLATE") ? atoi(getenv("SG_B16_ABLATE")) : 0;
#endif
  const int ks = b16_ks(p.C, p.K == p.C);
  if (bn == 128) {
    if (ks == 8) return launch_b16<128, 2, 4, 8>(p, st);
    if (ks == 4) return launch_b16<128, 2, 4, 4>(p, st);
    return launch_b16<128, 2, 4, 2>(p, st);
  }
  if (bn == 64) {
    if (ks == 8) return launch_b16<64, 4, 2, 8>(p, st);
    if (ks == 4) return launch_b16<64, 4, 2, 4>(p, st);
    return launch_b16<64, 4, 2, 2>(p, st);
  }
  if (ks == 8) return launch_b16<32, 4, 1, 8>(p, st);
  if (ks == 4) return launch_b16<32, 4, 1, 4>(p, st);
  return launch_b16<32, 4, 1, 2>(p, st);
}

// k-block depth of the weight planes a launch will read = the slab depth of its kernel: 32 for conv_x6_kernel (both plane
// counts), 16 * KS for conv_b16_kernel (bf16 storage, SG_B16_DEEP).  `c` is the depth of one tap (after virtual padding).
inline int x6_plane_kd(bool b16_storage, int c, bool one_tap) {
  static const bool deep = !(getenv("SG_B16_DEEP") && atoi(getenv("SG_B16_DEEP")) == 0);
  return (b16_storage && deep) ? 16 * b16_ks(c, one_tap) : 32;
}
