// "Patch" form of the x6 convolution (conv_x6.h) for 3x3, stride 1, dilation 1, "same" convolutions with 32 or 64
// reduction channels per tap: the full-resolution entry / decoder / U-Net convolutions.  Included by
// conv_igemm.hip inside its anonymous namespace, after conv_x6.h.
//
// With N = 32..64 output columns the im2col tile of conv_x6_kernel stages - and splits into the three bf16 planes -
// every input element nine times (once per tap) for 32..64 columns of MFMA work: the VALU split and the LDS
// traffic, not the matrix pipe, set its rate (65-125 TFLOP/s).  Here a workgroup owns an 8 x 16
// pixel tile of one image (128 GEMM rows):
//   * its 10 x 18 x C input patch is loaded and split ONCE into LDS (three planes, [patch row][pixel][C] bf16,
//     the 16-byte chunks of a pixel XOR-swizzled by the pixel so that the 16 lanes of a ds_read_b128 group - 16
//     consecutive pixels - hit 16 distinct bank groups; patch rows are padded to a multiple of 256 bytes);
//   * the nine taps' A fragments are plain reads of that patch at a pixel offset: no staging in the K loop;
//   * the weights are split once per launch into FRAGMENT-major planes (split3_weights_frag_kernel: for every
//     16-deep k-step, 32-column block and plane the 64 lanes' 16-byte operands lie contiguous, 1 KB), and each
//     wave loads its B fragments straight from global memory / L2 into registers, one k-step ahead;
//   * so the K loop has NO barrier: four waves drift freely, each holding all 128 rows (4 x 32) of one
//     32-column block.  With BN = 32 / 64 the four waves split K four / two ways (k-step s belongs to wave class
//     s mod KS) and the partial accumulators are summed through LDS in a fixed order at the end (deterministic);
//   * per k-step a wave issues 12 ds_read_b128 + 3 global loads for 24 MFMAs, the reads of the next half k-step
//     in flight under the MFMAs of the current one.
// Serves forward and dgrad alike (IgemmParams: off = -1, k_mul = +1 / off = +1, k_mul = -1).
//
// Round 5, more than 64 reduction channels per tap (128 / 256: the decoder's 3x3 convolutions at 128 x 128 and 256 x 256 that the
// planes-in kernel does not take): the tile's reduction is walked in CHUNKS of 64 channels - stage the chunk's patch, run its nine
// taps, next chunk into the same LDS - with the accumulators kept across the chunks.  The im2col kernel staged and split every input
// element nine times for 64 / 128 output columns (256 x 256 x 128 -> 64: 1.22 ms at 127 TFLOP/s, the slowest launch of the step).
// Order of additions of an output element: chunk by chunk, inside a chunk tap by tap - another order than the im2col kernel's
// (tap-major over all channels), fixed and independent of the batch.
//
// UpSampling2D(2, nearest) -> Conv2D 3x3 (train_model/DeepLabv3plus.py:476-477: 256 x 256 x 64 -> 512 x 512 x 64 -> 32; round 5):
//   * forward, UP2 = true ("sub-pixel" form): output pixel (2i + a, 2j + b) of the up-sampled grid sees, through its nine
//     taps, only the 2 x 2 source pixels (i + a - 1 .. i + a, j + b - 1 .. j + b): rows {i - 1: w[0], i: w[1] + w[2]} for
//     a = 0, {i: w[0] + w[1], i + 1: w[2]} for a = 1, columns alike.  So the layer IS a convolution on the SOURCE grid with
//     4 x Cout output columns (phase-major) in which phase (a, b) uses four of the nine source taps, with the kernel's taps
//     summed beforehand (split3_weights_up2_frag_kernel): 4 / 9 of the MFMA work, the 10 x 18 patch staged once per
//     8 x 16 SOURCE pixels = 512 output pixels instead of once per 128, and no up-sampled tensor at all.  Wave w of the
//     workgroup is phase w (C = 64, BN = 128: four column blocks, no K split); the epilogue stores pixel-shuffled.  The
//     summed taps change the rounding (w1 x + w2 x vs (w1 + w2) x): within fp32 rounding of the unfused layer, not
//     bit-identical to it - the parity tests compare with the oracle, not with the unfused graph.
//   * dgrad, SG_EPI_DOWN2: the ordinary dgrad on the up-sampled grid whose epilogue adds each 2 x 2 cell - all four values sit
//     in ONE lane's accumulators - in up-sampling's backward order ((g00 + g01) + g10) + g11 and stores the SOURCE-sized
//     gradient: bit-identical to dgrad + sg_upsample_nearest_bwd, without the 4x tensor in between.
#pragma once

struct IgemmParams;
inline int x6w_plan(const IgemmParams& p);   // conv_x6w.h: launches of the planes-in kernel keep it (x6p_ok below)

template <int C>
struct X6P {
  static constexpr int PB = 2 * C;                        // bytes of one pixel in one plane
  static constexpr int NCH = PB / 16;                     // 16-byte chunks per pixel (4 / 8)
  static constexpr int SH = (C == 32) ? 2 : 1;            // log2(pixels per 256 bytes)
  static constexpr int ROWP = (C == 32) ? 1280 : 2304;    // bytes per patch row: 18 pixels, padded to k * 256
  static constexpr int PLANE = 10 * ROWP;
  static constexpr int PATCH = 3 * PLANE;                 // 38,400 / 69,120 bytes
  __device__ static __forceinline__ int chunk_slot(int chunk, int pc) { return (chunk ^ ((pc >> SH) & (NCH - 1))) << 4; }
};

// B[k][n] (k = tap*Ck + kk, see split3_weights_kernel) -> fragment-major planes:
//   out[((ks * NB32 + nb) * 3 + plane) * 512 + lane * 8 + e] = plane of B[ks*16 + (lane>>5)*8 + e][nb*32 + (lane&31)]
__global__ __launch_bounds__(256) void split3_weights_frag_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int K,
                                                                  int N, int Ck, int s_tap, int s_k, int s_n) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = (int)(gid & 63);
  const int64_t f = gid >> 6;
  const int NB32 = N / 32;
  if (f >= (int64_t)(K / 16) * NB32) return;
  const int ks = (int)(f / NB32), nb = (int)(f - (int64_t)ks * NB32);
  const int n = nb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = k0 + 2 * e + u;
      const int tap = k / Ck, kk = k - tap * Ck;
      v[u] = w[(int64_t)tap * s_tap + (int64_t)kk * s_k + (int64_t)n * s_n];
    }
    split3_pair(v[0], v[1], h[e], m[e], l[e]);
  }
  u32x4_t* o = reinterpret_cast<u32x4_t*>(out + (f * 3) * 512 + lane * 8);
  o[0] = (u32x4_t){h[0], h[1], h[2], h[3]};
  o[64] = (u32x4_t){m[0], m[1], m[2], m[3]};   // + 512 ushorts = 64 u32x4
  o[128] = (u32x4_t){l[0], l[1], l[2], l[3]};
}

// The sub-pixel form's weights (see the header): w[3][3][Ck][Cout] -> effective kernel B[tap4 * Ck + kk][phase * Cout + co],
// tap4 = 2 r + c the source tap (row a + r - 1, column b + c - 1) of phase (a, b) = the sum of the kernel's taps that read it
// (rows: a = 0: r = 0 <- {0}, r = 1 <- {1, 2}; a = 1: r = 0 <- {0, 1}, r = 1 <- {2}; columns alike; added kh-major in
// ascending order), split into the fragment-major planes of the patch kernel (NB32 = 4 phases, Cout = 32).
__global__ __launch_bounds__(256) void split3_weights_up2_frag_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Ck,
                                                                      int Cout) {
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = (int)(gid & 63);
  const int64_t f = gid >> 6;
  if (f >= (int64_t)(4 * Ck / 16) * 4) return;
  const int ks = (int)(f >> 2), ph = (int)(f & 3);
  const int a = ph >> 1, b = ph & 1;
  const int n = lane & 31, k0 = ks * 16 + (lane >> 5) * 8;
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = k0 + 2 * e + u;
      const int tap4 = k / Ck, kk = k - tap4 * Ck;
      const int r = tap4 >> 1, c = tap4 & 1;
      const int kh0 = (a == 0) ? (r == 0 ? 0 : 1) : (r == 0 ? 0 : 2), kh1 = (a == 0) ? (r == 0 ? 0 : 2) : (r == 0 ? 1 : 2);
      const int kw0 = (b == 0) ? (c == 0 ? 0 : 1) : (c == 0 ? 0 : 2), kw1 = (b == 0) ? (c == 0 ? 0 : 2) : (c == 0 ? 1 : 2);
      float sum = 0.f;
      for (int kh = kh0; kh <= kh1; ++kh)
        for (int kw = kw0; kw <= kw1; ++kw) sum += w[((int64_t)(kh * 3 + kw) * Ck + kk) * Cout + n];
      v[u] = sum;
    }
    split3_pair(v[0], v[1], h[e], m[e], l[e]);
  }
  u32x4_t* o = reinterpret_cast<u32x4_t*>(out + (f * 3) * 512 + lane * 8);
  o[0] = (u32x4_t){h[0], h[1], h[2], h[3]};
  o[64] = (u32x4_t){m[0], m[1], m[2], m[3]};
  o[128] = (u32x4_t){l[0], l[1], l[2], l[3]};
}

// ---- every weight of a model in ONE launch (sg_prepare_planes): block b finds its job by binary search over the jobs'
// first-block numbers and does that job's tile exactly as split3_weights_kernel / split3_weights_frag_kernel would
__global__ __launch_bounds__(256) void prepare_planes_kernel(const float* __restrict__ w_arena, char* __restrict__ planes_arena,
                                                              const sg_planes_job* __restrict__ jobs, int njobs) {
  __shared__ float tile[32][33];
  int lo = 0, hi = njobs - 1;
  const int b = (int)blockIdx.x;
  while (lo < hi) {  // last job with block0 <= b
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= b) lo = mid; else hi = mid - 1;
  }
  const sg_planes_job j = jobs[lo];
  const int lb = b - j.block0;
  if (lb >= j.nblocks) return;
  const float* __restrict__ w = w_arena + j.w_off;
  unsigned short* __restrict__ out = reinterpret_cast<unsigned short*>(planes_arena + j.out_off);
  if (j.kind == 1 || j.kind == 3) {  // row planes [npl][Npad][Kpad]; kind 3: k-block-major [npl][Kpad / KD][Npad][KD], KD = Ckp
    const int kd = j.kd;
    const int gx = (j.Kpad + 31) / 32;
    const int k0 = (lb % gx) * 32, n0 = (lb / gx) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
      const int ky = (j.s_n == 1) ? i : tx, nx = (j.s_n == 1) ? tx : i;
      const int k = k0 + ky, n = n0 + nx;
      float v = 0.f;
      if (k < j.K && n < j.N) {
        const int tap = k / j.Ckp, kk = k - tap * j.Ckp;
        if (kk < j.Ck) v = w[(int64_t)tap * j.s_tap + (int64_t)kk * j.s_k + (int64_t)n * j.s_n];
      }
      tile[ky][nx] = v;
    }
    __syncthreads();
    const int64_t plane = (int64_t)j.Npad * j.Kpad;
    // 128 threads write the tile: thread -> column n0 + q / 4, eight consecutive k: one 16-byte store per plane (the 2-byte
    // stores this replaced made the once-per-step preparation of all planes 0.38 ms for 370 MB).  Kpad is a multiple of 32
    // and kd of 8, so a run of eight never crosses a k block or the padded extent.
    if (threadIdx.x < 128) {
      const int i = threadIdx.x >> 2, kq = threadIdx.x & 3;
      const int n = n0 + i, k = k0 + 8 * kq;
      if (!(kd && (k >= j.Kpad || n >= j.Npad))) {
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3_pair(tile[8 * kq + 2 * e][i], tile[8 * kq + 2 * e + 1][i], h[e], m[e], l[e]);
        const int64_t o = kd ? ((int64_t)(k / kd) * j.Npad + n) * kd + (k % kd) : (int64_t)n * j.Kpad + k;
        *reinterpret_cast<u32x4_t*>(out + o) = (u32x4_t){h[0], h[1], h[2], h[3]};
        if (j.npl == 3) {
          *reinterpret_cast<u32x4_t*>(out + plane + o) = (u32x4_t){m[0], m[1], m[2], m[3]};
          *reinterpret_cast<u32x4_t*>(out + 2 * plane + o) = (u32x4_t){l[0], l[1], l[2], l[3]};
        }
      }
    }
  } else {  // fragment-major planes of the patch kernel
    const int64_t gid = (int64_t)lb * 256 + threadIdx.x;
    const int lane = (int)(gid & 63);
    const int64_t f = gid >> 6;
    const int NB32 = j.N / 32;
    if (f >= (int64_t)(j.K / 16) * NB32) return;
    const int ks = (int)(f / NB32), nb = (int)(f - (int64_t)ks * NB32);
    const int n = nb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = k0 + 2 * e + u;
        const int tap = k / j.Ck, kk = k - tap * j.Ck;
        v[u] = w[(int64_t)tap * j.s_tap + (int64_t)kk * j.s_k + (int64_t)n * j.s_n];
      }
      split3_pair(v[0], v[1], h[e], m[e], l[e]);
    }
    u32x4_t* o = reinterpret_cast<u32x4_t*>(out + (f * 3) * 512 + lane * 8);
    o[0] = (u32x4_t){h[0], h[1], h[2], h[3]};
    o[64] = (u32x4_t){m[0], m[1], m[2], m[3]};
    o[128] = (u32x4_t){l[0], l[1], l[2], l[3]};
  }
}

// Persistent: gridDim.x = (workgroups that fit one CU) x CUs; workgroup g walks tiles g, g + gridDim.x, ...
// The timing-only ablations add up (512x512 64->64: 0.46 ms without the K loop + 1.42 ms without the patch loads =
// the 1.87 ms of the whole), which looked like the workgroups of a CU marching in lockstep.  `delay` (100 MHz ticks,
// SG_X6P_DELAY, default 0) holds back the second (third) workgroup of each CU at the start to test that picture: it
// changed nothing - the in-kernel clocks (SG_X6P_ABLATE=4) show CU mates already out of phase, a lone K loop at ~60 %
// of the pipe and overlapped ones barely slower.
__device__ unsigned g_x6p_arrivals[2048];  // per CU: workgroups that have started there (never reset; used modulo)

// MULTI: p.C is a multiple of C and the reduction is walked chunk by chunk (the accumulators then live through the staging of a
// chunk - 60 - 80 more registers than the one-chunk instantiations, which stay as they were)
template <int C, int BN, bool UP2 = false, bool MULTI = false>
__global__ __launch_bounds__(256, 2) void conv_x6p_kernel(const IgemmParams p, int tiles_x, int tiles_y, int ntiles, int delay,
                                                          int wg_per_cu) {
  using L = X6P<C>;
  static_assert(!UP2 || (C == 64 && BN == 128), "the sub-pixel form: 64 source channels, 4 phases x 32 output channels");
  static_assert(!MULTI || (C == 64 && !UP2), "chunks of 64 channels");
  constexpr int NBLK = BN / 32;     // 32-column blocks of the tile: 1, 2, 4
  constexpr int KS = 4 / NBLK;      // K classes: 4, 2, 1
  constexpr int CS = C / 16;        // k-steps per tap
  constexpr int NKS = (UP2 ? 4 : 9) * CS;   // UP2: a phase reads four source taps
  constexpr int RED = (KS > 1) ? 4 * NBLK * (KS - 1) * 4096 : 0;          // bytes of the K-class exchange
  constexpr int STAT_OFF = (L::PATCH > RED) ? L::PATCH : RED;             // [2][4 waves][32] floats behind it
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);  // scalar: K class, k-steps, taps and B addresses stay on the SALU
  const int nblk = wave % NBLK, kc = wave / NBLK;
  const uint32_t ntn = (uint32_t)p.Nout / BN;
  const int NB32 = p.Nout / 32;
  const int64_t bstep = (int64_t)NB32 * 3 * 512;  // ushorts per k-step of the fragment-major weights
  const bool do_relu = (p.flags & SG_EPI_RELU) != 0;

  if (delay > 0) {  // see above; the phase is the order of arrival on this very CU (HW_ID: CU / SH / SE, XCC_ID)
    __shared__ int s_phase;
    if (t == 0) {
      const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
      s_phase = (int)(atomicAdd(&g_x6p_arrivals[((xcc & 7u) << 8) | ((hw >> 8) & 0xffu)], 1u) % (unsigned)wg_per_cu);
    }
    __syncthreads();
    const int phase = s_phase;
    if (phase > 0) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime(), wait = (uint64_t)phase * (uint64_t)delay;
      while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
    }
  }

  for (uint32_t tile = xcd_remap(blockIdx.x, gridDim.x); tile < (uint32_t)ntiles; tile += gridDim.x) {
    const uint32_t tile_m = tile / ntn, tile_n = tile - tile_m * ntn;
    const uint32_t bx = tile_m % (uint32_t)tiles_x, tq = tile_m / (uint32_t)tiles_x;
    const uint32_t by = tq % (uint32_t)tiles_y, img = tq / (uint32_t)tiles_y;
    // The thread index is laundered once per tile: everything derived from it (the patch chunk arithmetic, the 64
    // output offsets of the epilogue) is otherwise hoisted out of the tile loop and lives - 150+ registers, spills -
    // through the K loop.
    int tl = t;
    asm volatile("" : "+v"(tl));
    const int lane = tl & 63, lr = lane & 31, lh = lane >> 5;
    const int px = lr & 15, pyl = lr >> 4;
    int kcl = kc;  // likewise the K class: the per-step scalars (tap, offsets, B addresses) of all unrolled steps would
    asm volatile("" : "+s"(kcl));  // be hoisted and spill the SGPR file

    const bool dbg = (p.ablate & 4) != 0;  // timing diagnostics: phase durations into y instead of the result
    uint64_t ts[6];
    uint64_t rt0 = 0;
    if (dbg) {
      ts[0] = __builtin_amdgcn_s_memtime();
      rt0 = __builtin_amdgcn_s_memrealtime();
    }
    f32x16 acc[4];
    if constexpr (MULTI) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
    const int nch = MULTI ? p.C / C : 1;        // chunks of C reduction channels per tap (uniform)
    const int cst = MULTI ? p.C / 16 : CS;      // k-steps per tap over all chunks: a chunk's k-step ks is tap * cst + ch * CS + ks % CS
    for (int ch = 0; ch < nch; ++ch) {
    // ---- the patch: rows by*8-1 .. by*8+8, columns bx*16-1 .. bx*16+16 of image img, zero outside the image.
    // A wave pass covers PPW consecutive pixels of one patch row (C/4 lanes per pixel, one float4 each); the
    // 10 x NPASS (row, pass) slots go round-robin to the four waves, so row and pass - and with them the row's base
    // address, its bounds test and its LDS row - are scalar, and a lane's share is a few adds per slot.
    {
      constexpr int LPP = C / 4, PPW = 64 / LPP, NPASS = (18 + PPW - 1) / PPW, NSLOT = 10 * NPASS, SPW = (NSLOT + 3) / 4;
      const int y0 = (int)by * 8 - 1, x0 = (int)bx * 16 - 1;
      const float* xi = p.x + (int64_t)img * p.H * p.W * p.x_ld + ch * C;
      const int c4 = lane & (LPP - 1), pxl = lane / LPP;
      f32x4 v[SPW];
      unsigned okm = 0;   // bit k: slot k of this lane lies inside the image (the BatchNormalization below leaves the padding zero)
      // round 5: the BatchNormalization(+ReLU) in front of this layer applied while the patch is split (IgemmParams::bn; uniform)
      const bool bn_on = p.bn.mean != nullptr;
      f32x4 bm = {0.f, 0.f, 0.f, 0.f}, bi = bm, bg = bm, bb = bm;
      if (bn_on) {
        bm = *reinterpret_cast<const f32x4*>(p.bn.mean + ch * C + c4 * 4);
        bi = bn_in_inv(p.bn, ch * C + c4 * 4);
        bg = *reinterpret_cast<const f32x4*>(p.bn.gamma + ch * C + c4 * 4);
        bb = *reinterpret_cast<const f32x4*>(p.bn.beta + ch * C + c4 * 4);
      }
#pragma unroll
      for (int k = 0; k < SPW; ++k) {
        const int slot = wave + 4 * k;  // uniform
        const int pr = slot / NPASS, ps = slot - pr * NPASS;
        const int pc = ps * PPW + pxl, gy = y0 + pr, gx = x0 + pc;
        const bool ok = (NSLOT % 4 == 0 || slot < NSLOT) && pc < 18 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        v[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (ok) okm |= 1u << k;
        if (ok && !(p.ablate & 2)) v[k] = *reinterpret_cast<const f32x4*>(xi + ((int64_t)gy * p.W + gx) * p.x_ld + c4 * 4);
      }
      __syncthreads();  // the previous tile's LDS reads (K loop, K-class exchange, statistics) are done
      if (dbg) {
        ts[1] = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ts[2] = __builtin_amdgcn_s_memtime();
      }
#pragma unroll
      for (int k = 0; k < SPW; ++k) {
        const int slot = wave + 4 * k;
        const int pr = slot / NPASS, ps = slot - pr * NPASS;
        const int pc = ps * PPW + pxl;
        if ((NSLOT % 4 == 0 || slot < NSLOT) && pc < 18) {
          if (bn_on) {
            const bool in = (okm >> k) & 1u;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] = in ? bn_in_one(v[k][e], bm[e], bi[e], bg[e], bb[e], p.bn.relu) : 0.f;
          }
          unsigned h0, m0, l0, h1, m1, l1;
          split3_pair(v[k][0], v[k][1], h0, m0, l0);
          split3_pair(v[k][2], v[k][3], h1, m1, l1);
          char* dst = smem + pr * L::ROWP + pc * L::PB + L::chunk_slot(c4 >> 1, pc) + (c4 & 1) * 8;
          *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){h0, h1};
          *reinterpret_cast<u32x2_t*>(dst + L::PLANE) = (u32x2_t){m0, m1};
          *reinterpret_cast<u32x2_t*>(dst + 2 * L::PLANE) = (u32x2_t){l0, l1};
        }
      }
    }
    __syncthreads();
    if (dbg) ts[3] = __builtin_amdgcn_s_memtime();

    // ---- K loop: no barrier ---------------------------------------------------------------------------------
    const unsigned short* bq = p.wq + ((int64_t)(tile_n * NBLK + nblk) * 3) * 512 + lane * 8;
    if constexpr (!MULTI) {   // (one chunk: the accumulators start their life behind the staging, as before round 5)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
    // k-step ks of this chunk in the layer's fragment-major planes (all chunks: tap-major, then channels)
    auto gks = [&](int ks) -> int { return MULTI ? ks + (ks / CS) * (cst - CS) + ch * CS : ks; };

    // (B fragments are kept as integer vectors: bf16 vectors that cross the conditional steps are rebuilt element
    // by element - 24 shift / permute instructions per k-step)
    auto load_b = [&](int ks, u32x4_t (&b)[3]) {
      const unsigned short* q = bq + gks(ks) * bstep;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[pl] = *reinterpret_cast<const u32x4_t*>(q + pl * 512);
    };
    // byte offset (inside plane 0, row block 0) of this lane's A operand of k-step ks
    auto a_offset = [&](int ks) -> int {
      const int tap = ks / CS, cs = ks - tap * CS;
      int dyr, dxr;
      if constexpr (UP2) {  // wave = column block = phase (a, b): source taps (a + r - 1, b + c - 1), patch row 0 = source row - 1
        dyr = (nblk >> 1) + (tap >> 1);
        dxr = (nblk & 1) + (tap & 1);
      } else {
        const int kh = tap / 3, kw = tap - kh * 3;
        dyr = 1 + p.off_h + kh * p.k_mul;
        dxr = 1 + p.off_w + kw * p.k_mul;
      }
      const int q = px + dxr;
      return (pyl + dyr) * L::ROWP + q * L::PB + L::chunk_slot(cs * 2 + lh, q);
    };
    auto read_a = [&](int off, int half, bf16x8_t (&a)[2][3]) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        constexpr int APL[3] = {2, 0, 1};  // consumption order of the planes (a3, a1, a2)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          a[i][APL[u]] = *reinterpret_cast<const bf16x8_t*>(smem + APL[u] * L::PLANE + (2 * half + i) * 2 * L::ROWP + off);
      }
    };
    {
      // Fully unrolled (NJ <= 18 steps of 12 reads + 24 MFMAs): with a rolled loop the fragment reads are loop-carried
      // and the compiler waits lgkmcnt(0) before every first MFMA - on the reads it has just issued; straight-line code
      // gets counted waits, and the B double buffer needs no register copies.
      constexpr int NJ = (NKS + KS - 1) / KS;
      constexpr int PD = 2;  // B fragments are fetched PD k-steps ahead (an L2 round trip under load outlasts one k-step)
      bf16x8_t a[2][2][3];
      u32x4_t b[PD + 1][3];
      int off = a_offset(kcl);
#pragma unroll
      for (int d = 0; d < PD; ++d) {
        const int kd = kcl + d * KS;
        load_b(kd < NKS ? kd : kcl, b[d]);
      }
      read_a(off, 0, a[0]);
      // One piece of the next operands per MFMA, fenced, as in conv_x6_kernel's fused step: a wave issues in order, so
      // whatever follows a run of MFMAs starts only when the last of them has issued and must fit its 32 cycles or
      // the pipe idles; one read / load / address piece after each MFMA hides under that MFMA instead.  Half 0 carries
      // the six A reads of half 1 and the three B loads of step j + PD, half 1 the address of step j + 1 and its
      // first six A reads.  (No branch inside the unrolled steps other than the ragged last one: across basic blocks
      // the counted waits degrade to lgkmcnt(0) again.)
      auto read_a_piece = [&](int o, int half, int q, bf16x8_t (&fr)[2][3]) {
        constexpr int APL[3] = {2, 0, 1};
        const int u = q >> 1, i = q & 1;
        fr[i][APL[u]] = *reinterpret_cast<const bf16x8_t*>(smem + APL[u] * L::PLANE + (2 * half + i) * 2 * L::ROWP + o);
      };
      auto mfma_one = [&](int half, int m, bf16x8_t (&fr)[2][3], u32x4_t (&bb)[3]) {
        constexpr int PA_[6] = {2, 0, 1, 1, 0, 0}, PB_[6] = {0, 2, 1, 0, 1, 0};  // smallest terms first
        const int term = m >> 1, i = m & 1;
        acc[2 * half + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[i][PA_[term]], __builtin_bit_cast(bf16x8_t, bb[PB_[term]]),
                                                                    acc[2 * half + i], 0, 0, 0);
      };
      if (!(p.ablate & 1))
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int ks = kcl + j * KS;
        if (NKS % KS != 0 && j == NJ - 1 && ks >= NKS) continue;  // uniform (ragged K split: C = 32, KS = 4)
        const int kn = (ks + KS < NKS) ? ks + KS : ks;  // the tail re-reads the last step (unused)
        const int kp = (ks + PD * KS < NKS) ? ks + PD * KS : ks;
        const unsigned short* qb = bq + gks(kp) * bstep;
#pragma unroll
        for (int m = 0; m < 12; ++m) {
          mfma_one(0, m, a[0], b[j % (PD + 1)]);
          if (m < 6) read_a_piece(off, 1, m, a[1]);
          else if (m < 9 && j + PD < NJ) b[(j + PD) % (PD + 1)][m - 6] = *reinterpret_cast<const u32x4_t*>(qb + (m - 6) * 512);
          __builtin_amdgcn_sched_barrier(0);
        }
        int offn = off;
#pragma unroll
        for (int m = 0; m < 12; ++m) {
          mfma_one(1, m, a[1], b[j % (PD + 1)]);
          if (j + 1 < NJ) {
            if (m == 0) offn = a_offset(kn);
            else if (m <= 6) read_a_piece(offn, 0, m - 1, a[0]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        off = offn;
      }
    }

    }   // chunks of the reduction channels
    if (dbg) ts[4] = __builtin_amdgcn_s_memtime();
    // ---- K classes: 32-row block i of a column block is finalised by the wave of class i % KS, which sums the
    // classes' partials in the fixed order 0, 1, .. (its own from registers, the others' through LDS; the patch is dead)
    if constexpr (KS > 1) {
      float* red = reinterpret_cast<float*>(smem);  // [NBLK][4 blocks][KS - 1 foreign classes][16][64 lanes]
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int own = i % KS;
        if (kc != own) {
          float* dst = red + ((nblk * 4 + i) * (KS - 1) + (kc - (kc > own ? 1 : 0))) * 1024 + lane;
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[r * 64] = acc[i][r];
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int own = i % KS;
        if (kc == own) {
          f32x16 tot;
#pragma unroll
          for (int k = 0; k < KS; ++k) {
            f32x16 val;
            if (k == own) {
              val = acc[i];
            } else {
              const float* src = red + ((nblk * 4 + i) * (KS - 1) + (k - (k > own ? 1 : 0))) * 1024 + lane;
#pragma unroll
              for (int r = 0; r < 16; ++r) val[r] = src[r * 64];
            }
            if (k == 0) tot = val;
            else tot += val;
          }
          acc[i] = tot;
        }
      }
    }

    // ---- epilogue: each wave stores the blocks it finalised (all four when KS == 1) --------------------------------
    const int col = UP2 ? lr : (int)tile_n * BN + nblk * 32 + lr;   // UP2: the wave's 32 columns are the layer's 32 channels
    const float bv = (p.flags & SG_EPI_BIAS) ? p.bias[col] : 0.f;
    float s = 0.f;
    if (!UP2 && (p.flags & SG_EPI_DOWN2)) {
      // dgrad behind a 2x nearest up-sampling: block i holds tile rows 2i, 2i + 1 = ONE source row; the lane's register
      // r -> pixel (rb >> 4, rb & 15), rb = (r & 3) + 8 (r >> 2) + 4 lh, so the 2 x 2 cell of source column
      // (r >> 1 & 1) + 2 lh + 4 (r >> 2 & 1) is registers r0, r0 | 1 (next column), r0 | 8, r0 | 9 (next row)
      const int OWs = p.OW >> 1;
      float* yo = p.y + (((int64_t)img * (p.OH >> 1) + (int64_t)by * 4) * OWs + bx * 8) * p.y_ld + col;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KS == 1 || kc == i % KS) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int r0 = 2 * (q & 1) + 4 * (q >> 1);
            const float v = ((acc[i][r0] + acc[i][r0 | 1]) + acc[i][r0 | 8]) + acc[i][r0 | 9];   // sg_upsample_nearest_bwd's order
            if (!dbg && !(p.ablate & 8)) yo[((int64_t)i * OWs + (q & 1) + 2 * lh + 4 * (q >> 1)) * p.y_ld] = v;
          }
        }
      }
    } else {
      float* yo;
      int64_t rowp, colp;   // element strides of one tile row / tile column in y
      if constexpr (UP2) {  // source pixel (Y, X) of phase (a, b) -> pixel (2 Y + a, 2 X + b) of the 2 OH x 2 OW output
        const int64_t OW2 = 2 * (int64_t)p.OW;
        yo = p.y + ((((int64_t)img * 2 * p.OH + (int64_t)by * 16 + (nblk >> 1)) * OW2) + bx * 32 + (nblk & 1)) * p.y_ld + col;
        rowp = 2 * OW2 * p.y_ld;
        colp = 2 * (int64_t)p.y_ld;
      } else {
        yo = p.y + ((int64_t)img * p.OH * p.OW + (int64_t)by * 8 * p.OW + bx * 16) * p.y_ld + col;
        rowp = (int64_t)p.OW * p.y_ld;
        colp = p.y_ld;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KS == 1 || kc == i % KS) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rb = (r & 3) + 8 * (r >> 2) + 4 * lh;  // row of the 32-row block = pixel (2i + rb/16, rb%16)
            float v = acc[i][r] + bv;
            acc[i][r] = v;
            s += v;
            if (do_relu) v = fmaxf(v, 0.f);
            if (!dbg && !(p.ablate & 8)) yo[(int64_t)(2 * i + (rb >> 4)) * rowp + (rb & 15) * colp] = v;
          }
        }
      }
    }
    if (dbg) {
      ts[5] = __builtin_amdgcn_s_memtime();
      const uint32_t it = (tile - xcd_remap(blockIdx.x, gridDim.x)) / gridDim.x;
      const uint32_t bb = blockIdx.x, nc = gridDim.x / (uint32_t)wg_per_cu;  // slots 0..31: blocks 0..31, 32..63: their CU mates
      if (tl == 0 && (bb < 32 || (bb >= nc && bb < nc + 32)) && it < 16) {
        float* o = p.y + ((int64_t)(bb < 32 ? bb : bb - nc + 32) * 16 + it) * 12;
        for (int k = 0; k < 5; ++k) o[k] = (float)(ts[k + 1] - ts[k]);
        o[5] = (float)(ts[3] & 0xffffff);  // K-loop start
        o[6] = (float)(((__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u) << 8) | ((__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xffu));
        o[7] = (float)(ts[4] & 0xffffff);  // K-loop end
        o[8] = (float)(__builtin_amdgcn_s_memrealtime() - rt0);  // the tile in 10 ns ticks: shader clock = cycles / ticks x 100 MHz
      }
    }
    // BatchNormalization statistics of the tile (see conv_x6_kernel): sum, then centred sum of squares, over its 128
    // rows; with KS > 1 the rows of a column block are spread over KS waves, combined through LDS in a fixed order
    if (p.stats) {
      float* sl = reinterpret_cast<float*>(smem + STAT_OFF);
      s += __shfl_xor(s, 32, 64);
      if constexpr (KS > 1) {
        if (lh == 0) sl[wave * 32 + lr] = s;
        __syncthreads();
        s = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) s += sl[(k * NBLK + nblk) * 32 + lr];
      }
      const float mu = s * (1.f / 128.f);
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KS == 1 || kc == i % KS) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float d = acc[i][r] - mu;
            q = fmaf(d, d, q);
          }
        }
      }
      q += __shfl_xor(q, 32, 64);
      if constexpr (KS > 1) {
        if (lh == 0) sl[128 + wave * 32 + lr] = q;
        __syncthreads();
        q = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) q += sl[128 + (k * NBLK + nblk) * 32 + lr];
      }
      if (lh == 0 && kc == 0) {
        if constexpr (UP2) {  // a phase's 128 output pixels are one statistics tile: row 4 tile_m + phase of [tiles][2][32]
          p.stats[((int64_t)(tile_m * 4 + nblk) * 2) * 32 + col] = s;
          p.stats[((int64_t)(tile_m * 4 + nblk) * 2 + 1) * 32 + col] = q;
        } else {
          p.stats[((int64_t)tile_m * 2) * p.Nout + col] = s;
          p.stats[((int64_t)tile_m * 2 + 1) * p.Nout + col] = q;
        }
      }
    }
  }
}

inline bool x6p_enabled() {
  static const bool on = getenv("SG_X6_NOPATCH") == nullptr;
  return on;
}

// geometry the patch form covers; KH, KW are the filter's (IgemmParams carries only K)
inline bool x6p_ok(const IgemmParams& p, int KH, int KW) {
  if (!x6p_enabled() || KH != 3 || KW != 3) return false;
  // 32 or 64 reduction channels per tap in one patch; 128 / 256 / ... in chunks of 64 (round 5) unless the planes-in kernel takes
  // the launch (K >= 2048 and >= 192 output columns: conv_x6w.h)
  if (p.K != 9 * p.C) return false;
  if (!(p.C == 32 || p.C == 64)) {
    static const int multi = getenv("SG_X6P_CHUNKS") ? atoi(getenv("SG_X6P_CHUNKS")) : 1;
    if (!multi || p.C % 64 != 0 || p.C > 512) return false;
  }
  if (p.a_mul != 1 || p.div != 1) return false;
  const bool fwd = p.k_mul == 1 && p.off_h == -1 && p.off_w == -1, bwd = p.k_mul == -1 && p.off_h == 1 && p.off_w == 1;
  if (!fwd && !bwd) return false;
  if (p.OH != p.H || p.OW != p.W || (p.H % 8) || (p.W % 16)) return false;
  if (!(p.Nout == 32 || p.Nout == 64 || p.Nout % 128 == 0)) return false;
  if ((p.x_ld % 4) || (((uintptr_t)p.x) & 15)) return false;
  if (p.C > 64) {   // (the planes-in kernel declines a launch that adds a collected gradient; the weight planes were laid out
    IgemmParams q = p;   // for it all the same, so the answer here must not depend on `res`)
    q.res = nullptr;
    if (x6w_plan(q) > 0) return false;
  }
  return true;
}

template <int C, int BN, bool UP2 = false, bool MULTI = false>
int launch_x6p(const IgemmParams& p, int num_cus, hipStream_t st) {
  constexpr int KS = 4 / (BN / 32), NBLK = BN / 32;
  constexpr size_t red = (KS > 1) ? (size_t)4 * NBLK * (KS - 1) * 4096 : 0;
  constexpr size_t lds = (X6P<C>::PATCH > red ? (size_t)X6P<C>::PATCH : red) + 2 * 4 * 32 * sizeof(float);
  static int wg_per_cu = 0;
  if (!wg_per_cu) {
    int rc = set_dyn_lds(conv_x6p_kernel<C, BN, UP2, MULTI>, lds);
    if (rc) return rc;
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_x6p_kernel<C, BN, UP2, MULTI>, 256, lds);
    if (e != hipSuccess || nb < 1) nb = 1;
    wg_per_cu = nb > 4 ? 4 : nb;
  }
  const int tiles_x = p.OW / 16, tiles_y = p.OH / 8;
  const int64_t tiles = (int64_t)(p.M / 128) * (p.Nout / BN);
  if (tiles <= 0 || tiles > 0x7fffffff || (int64_t)(p.M / 128) % ((int64_t)tiles_x * tiles_y) != 0) {
    sg_set_error("conv_x6p: bad tile count %lld", (long long)tiles);
    return SG_EINVAL;
  }
  const int64_t slots = (int64_t)num_cus * wg_per_cu;
  const int64_t grid = tiles < slots ? tiles : slots;
  // Experiment switch, default OFF (measured: no effect, the workgroups of a CU are not in lockstep): SG_X6P_DELAY = start
  // offset of the CU's 2nd (3rd) workgroup in 10 ns ticks (-1: one K loop of a tile at ~1.1 TFLOP/s per CU); only when
  // every workgroup walks at least two tiles
  static int dly_env = -2;
  if (dly_env == -2) dly_env = getenv("SG_X6P_DELAY") ? atoi(getenv("SG_X6P_DELAY")) : 0;
  int delay = 0;
  if (dly_env != 0 && grid == slots && wg_per_cu > 1 && tiles >= 2 * slots)
    delay = dly_env > 0 ? dly_env : (int)(2.0 * 128 * BN * 9 * C / 1.1e6 * 100.0 + 0.5);
  hipLaunchKernelGGL((conv_x6p_kernel<C, BN, UP2, MULTI>), dim3((unsigned)grid), dim3(256), lds, st, p, tiles_x, tiles_y, (int)tiles, delay, wg_per_cu);
  SG_LAUNCH_CHECK("conv_x6p_kernel");
  return 0;
}

// split the weights fragment-major (in `ws`, x6_planes_bytes() is enough) and run the patch kernel
int run_x6p(IgemmParams& p, const float* w, bool dgrad, int Cin, int Cout, void* ws, int num_cus, hipStream_t st,
            bool prepared = false) {
  const int K = p.K, N = p.Nout;
  {
    static int abl = -1;  // SG_X6P_ABLATE (timing only, results wrong): 1 = no K loop, 2 = no patch loads, 4 = phase clocks into y, 8 = no y stores
    if (abl < 0) abl = getenv("SG_X6P_ABLATE") ? atoi(getenv("SG_X6P_ABLATE")) & 15 : 0;
    p.ablate = abl;
  }
  p.wq = (const unsigned short*)ws;
  if (!prepared) {
    const int64_t threads = (int64_t)(K / 16) * (N / 32) * 64;
    const dim3 grid((unsigned)sg_cdiv(threads, 256));
    if (!dgrad)
      hipLaunchKernelGGL(split3_weights_frag_kernel, grid, dim3(256), 0, st, w, (unsigned short*)ws, K, N, Cin, Cin * Cout, Cout, 1);
    else
      hipLaunchKernelGGL(split3_weights_frag_kernel, grid, dim3(256), 0, st, w, (unsigned short*)ws, K, N, Cout, Cin * Cout, 1, Cout);
    SG_LAUNCH_CHECK("split3_weights_frag_kernel");
  }
  const int bn = N >= 128 ? 128 : N;
  if (p.C == 32) {
    if (bn == 32) return launch_x6p<32, 32>(p, num_cus, st);
    if (bn == 64) return launch_x6p<32, 64>(p, num_cus, st);
    return launch_x6p<32, 128>(p, num_cus, st);
  }
  if (p.C > 64) {   // 128 / 256 / ... reduction channels per tap: chunks of 64
    if (bn == 32) return launch_x6p<64, 32, false, true>(p, num_cus, st);
    if (bn == 64) return launch_x6p<64, 64, false, true>(p, num_cus, st);
    return launch_x6p<64, 128, false, true>(p, num_cus, st);
  }
  if (bn == 32) return launch_x6p<64, 32>(p, num_cus, st);
  if (bn == 64) return launch_x6p<64, 64>(p, num_cus, st);
  return launch_x6p<64, 128>(p, num_cus, st);
}

// ---- UpSampling2D(2) -> Conv2D 3x3: which launches take the fused kernels (sg_conv2d_fwd* with SG_PRO_UP2, sg_conv2d_dgrad with
// SG_EPI_DOWN2).  `d` describes the convolution on the UP-SAMPLED grid (H x W = twice the source's).
inline bool x6p_up2_geom(const sg_conv_desc* d) {
  if (!x6p_enabled() || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dilation != 1) return false;
  if (d->pad_t != 1 || d->pad_l != 1 || d->Ho != d->H || d->Wo != d->W) return false;
  if (d->Cin != 64 || d->Cout != 32) return false;
  if ((d->H % 16) || (d->W % 32)) return false;   // 8 x 16 SOURCE tiles (forward), 8 x 16 output tiles (dgrad)
  return true;
}

// the sub-pixel forward: p describes the convolution on the SOURCE grid (H, W, OH, OW, M = source; Nout = 128 = 4 phases x 32)
inline int run_x6p_up2(IgemmParams& p, const float* w, void* ws, int num_cus, hipStream_t st) {
  p.ablate = 0;
  p.wq = (const unsigned short*)ws;
  hipLaunchKernelGGL(split3_weights_up2_frag_kernel, dim3(16), dim3(256), 0, st, w, (unsigned short*)ws, 64, 32);
  SG_LAUNCH_CHECK("split3_weights_up2_frag_kernel");
  return launch_x6p<64, 128, true>(p, num_cus, st);
}
inline size_t x6p_up2_ws_bytes() { return (size_t)16 * 4 * 3 * 512 * 2; }   // 16 k-steps x 4 phases x 3 planes x 1 KB
