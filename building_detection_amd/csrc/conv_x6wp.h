// "Patch" form of the filter gradient for 3x3, stride 1, dilation 1, "same" convolutions with 32 or 64 channels on both
// sides: the full-resolution entry / decoder convolutions (DeepLabv3+: 64->32 and 32->32 at 512x512, 32->64 and 64->64
// at 256x256).  Included by conv_igemm.hip inside its anonymous namespace, after conv_x6.h.
//
// wgrad_x6_kernel tiles the [9 Cin][Cout] gradient 128 rows at a time and streams the pixels past every tile: with
// Cin = 64 that is five row tiles, each re-reading x at its own taps (9 x in all) and dy (5 x) - 6 GB of L2 / fabric reads
// for 0.8 GB of tensors, and the kernel runs at that bandwidth (1.36 ms for 64->32 at 512x512, 70 TFLOP/s).  Here the
// roles are swapped.  A workgroup keeps the WHOLE gradient of its share of pixels in registers and walks 4 x 16 pixel
// tiles of the images:
//   * the 6 x 18 x Cin input patch and the 4 x 16 x Cout dy tile are loaded ONCE per tile and go to LDS as they lie
//     ([pixel][channel], split into the three bf16 planes on the way: the staging of wgrad_x6_kernel);
//   * a k-step is 16 consecutive pixels of one tile row; its A fragment for tap (kh, kw) is the transposed read
//     (ds_read_b64_tr_b16) of the patch at pixel offset (row + kh) * 18 + kw - nine fragments from one staging -, its B
//     fragment the transposed read of the dy tile, shared by the nine taps;
//   * a wave owns one (32 input channels, 32 output channels) pair and all nine taps: 9 accumulator blocks = 144
//     registers; with fewer than four such pairs the waves also split the tile's four rows (k classes), summed through
//     LDS in a fixed order when the workgroup has walked its last tile;
//   * pixel rows are 64 B (32 channels) or 128 B (64 channels, the two 64-byte halves swapped on pixels 2, 3 mod 4), so
//     the four pixel rows a transposed read touches per half wave lie in four different 64-byte bank groups at any
//     tap shift; no padding: 2 workgroups per CU fit (<= 66 KB each).
// Every workgroup writes one partial gradient slab; reduce_splits_kernel adds them in fixed order as before.
#pragma once

template <int CI, int CO, int NPL, typename TA>
struct X6WP {
  static constexpr int TH = 4, TW = 16, PW = TW + 2, NPX = (TH + 2) * PW;  // 108 patch pixels
  static constexpr int XP = CI * 2, YP = CO * 2;                          // bytes per pixel per plane
  static constexpr int XPLANE = NPX * XP, YPLANE = TH * TW * YP;
  static constexpr int YOFF = NPL * XPLANE;
  static constexpr int STAGE = NPL * (XPLANE + YPLANE);
  static constexpr int NCB = CI / 32, NOB = CO / 32, NG = NCB * NOB;      // (ci block, co block) pairs: 1, 2, 4
  static constexpr int KSPLIT = 4 / NG;                                   // k classes: 4, 2, 1
  static constexpr int RED = (KSPLIT > 1) ? NG * 3 * 16 * 64 * 4 : 0;      // three taps of every pair at a time
  static constexpr int LDS = STAGE > RED ? STAGE : RED;
};

template <int CI, int CO, int NPL, typename TA>
__global__ __launch_bounds__(256, 2) void wgrad_x6wp_kernel(const WgradParams p, int tiles_x, int tiles_y, int ntiles) {
  using L = X6WP<CI, CO, NPL, TA>;
  static_assert(NPL == 3 || NPL == 1, "planes");
  static_assert(NPL == 1 || std::is_same<TA, float>::value, "the three-plane split is the fp32 path");
  constexpr bool A16 = !std::is_same<TA, float>::value;
  constexpr int EB = EL<TA>::BYTES, CH = EL<TA>::CH;   // bytes per element, channels per 16-byte chunk
  constexpr int CPX = CI / CH, CPY = CO / CH;           // 16-byte chunks per pixel
  constexpr int NXC = L::NPX * CPX, NYC = L::TH * L::TW * CPY;
  constexpr int NXL = (NXC + 255) / 256, NYL = (NYC + 255) / 256;
  constexpr unsigned OOB = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int g = wave % L::NG, kc = wave / L::NG;
  const int cib = g % L::NCB, cob = g / L::NCB;

  const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);

  // ---- staging: chunk idx = t + 256 j of the patch -> (patch pixel, 16-byte chunk); its LDS place is tile-invariant
  u32x4_t rx[NXL], ry[NYL];
  unsigned okm = 0;   // bit j: patch chunk j of the tile in flight lies inside the image (WgradParams::bn keeps the padding zero)
  // round 5: the BatchNormalization(+ReLU) in front of the layer applied to x while the patch is split (fp32 storage; uniform).
  // A thread's chunks t + 256 j all hold the same four channels (256 is a multiple of the chunks per pixel).
  // (the four parameter vectors are re-read from L2 in every store_tile: kept in registers through the MFMA section they spill
  // the 64 x 64 instantiation, which sits at the 256-register cap)
  const bool bn_on = !A16 && p.bn.mean != nullptr;
  auto load_tile = [&](uint32_t tile) {
    const uint32_t bx = tile % (uint32_t)tiles_x, tq = tile / (uint32_t)tiles_x;
    const uint32_t by = tq % (uint32_t)tiles_y, img = tq / (uint32_t)tiles_y;
    const int y0 = (int)by * L::TH - 1, x0 = (int)bx * L::TW - 1;
    const int ibase = (int)img * p.H;
    const int up = p.up, ibase_s = (int)img * (p.H >> up), Ws = p.W >> up;   // SG_X_UP2: the patch is gathered from the source
    // the thread index is laundered per call: what derives from it (pixel, chunk, validity, LDS place of up to 11 chunks)
    // would otherwise be hoisted out of the tile loop and live - and spill - through the MFMA section
    int tl = t;
    asm volatile("" : "+v"(tl));
    okm = 0;
#pragma unroll
    for (int j = 0; j < NXL; ++j) {
      const int idx = tl + 256 * j;
      const int pp = idx / CPX, c = idx - pp * CPX;
      const int pr = pp / L::PW, pc = pp - pr * L::PW;
      const int gy = y0 + pr, gx = x0 + pc;
      const bool ok = (NXC % 256 == 0 || idx < NXC) && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      if (ok) okm |= 1u << j;
      const unsigned off = (unsigned)(((ibase_s + (gy >> up)) * Ws + (gx >> up)) * p.x_ld + c * CH) * (unsigned)EB;
      rx[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, (int)(ok ? off : OOB), 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NYL; ++j) {
      const int idx = tl + 256 * j;
      const int q = idx / CPY, c = idx - q * CPY;
      const int gy = (int)by * L::TH + (q >> 4), gx = (int)bx * L::TW + (q & 15);
      const bool ok = (NYC % 256 == 0 || idx < NYC);
      const unsigned off = (unsigned)(((ibase + gy) * p.W + gx) * p.y_ld + c * CH) * (unsigned)EB;
      ry[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_dy, (int)(ok ? off : OOB), 0, 0);
    }
  };
  // one 16-byte chunk -> its place in a [pixel][channel] plane image (PB bytes per pixel; 128-byte pixels swap their halves
  // on pixels 2, 3 mod 4)
  auto store_chunk = [&](const u32x4_t v, char* plane0, int plane_bytes, int pixel, int c, int PB) {
    const int cb = c * (A16 ? 16 : 8);  // byte offset of the chunk's bf16 image inside the pixel
    const int sw = (PB == 128) ? (((pixel >> 1) & 1) << 6) : 0;
    char* dst = plane0 + pixel * PB + (cb ^ sw);
    if constexpr (A16) {
      *reinterpret_cast<u32x4_t*>(dst) = v;
    } else {
      const f32x4 f = __builtin_bit_cast(f32x4, v);
      if constexpr (NPL == 3) {
        unsigned h0, m0, l0, h1, m1, l1;
        split3_pair(f[0], f[1], h0, m0, l0);
        split3_pair(f[2], f[3], h1, m1, l1);
        *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){h0, h1};
        *reinterpret_cast<u32x2_t*>(dst + plane_bytes) = (u32x2_t){m0, m1};
        *reinterpret_cast<u32x2_t*>(dst + 2 * plane_bytes) = (u32x2_t){l0, l1};
      } else {
        *reinterpret_cast<u32x2_t*>(dst) = (u32x2_t){pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3])};
      }
    }
  };
  auto store_tile = [&]() {
    int tl = t;
    asm volatile("" : "+v"(tl));
    f32x4 bm = {0.f, 0.f, 0.f, 0.f}, bi = bm, bg = bm, bb = bm;
    if (bn_on) {
      const int cb = (tl % CPX) * 4;
      bm = *reinterpret_cast<const f32x4*>(p.bn.mean + cb);
      bi = bn_in_inv(p.bn, cb);
      bg = *reinterpret_cast<const f32x4*>(p.bn.gamma + cb);
      bb = *reinterpret_cast<const f32x4*>(p.bn.beta + cb);
    }
#pragma unroll
    for (int j = 0; j < NXL; ++j) {
      const int idx = tl + 256 * j;
      if (NXC % 256 == 0 || idx < NXC) {
        u32x4_t v = rx[j];
        if (bn_on) {
          f32x4 f = __builtin_bit_cast(f32x4, v);
          const bool in = (okm >> j) & 1u;
#pragma unroll
          for (int e = 0; e < 4; ++e) f[e] = in ? bn_in_one(f[e], bm[e], bi[e], bg[e], bb[e], p.bn.relu) : 0.f;
          v = __builtin_bit_cast(u32x4_t, f);
        }
        store_chunk(v, smem, L::XPLANE, idx / CPX, idx % CPX, L::XP);
      }
    }
#pragma unroll
    for (int j = 0; j < NYL; ++j) {
      const int idx = tl + 256 * j;
      if (NYC % 256 == 0 || idx < NYC) store_chunk(ry[j], smem + L::YOFF, L::YPLANE, idx / CPY, idx % CPY, L::YP);
    }
  };

  // ---- fragments: lane = 16 tg + ti supplies pixel row 8 (tg>>1) + (ti>>2) [+4 for the second read], channels
  // 16 (tg&1) + 4 (ti&3) .. +3 of its 32-channel block (tr_frag, conv_x6.h)
  const int tg = lane >> 4, ti = lane & 15;
  const int tr_row = 8 * (tg >> 1) + (ti >> 2), tr_colb = (16 * (tg & 1) + 4 * (ti & 3)) * 2;
  auto a_frag = [&](int pl, int pix0) -> bf16x8_t {  // 16 patch pixels from pix0, this wave's input-channel block
    const int pix = pix0 + tr_row;
    const int sw = (L::XP == 128) ? (((pix >> 1) & 1) << 6) : 0;
    const char* a = smem + pl * L::XPLANE + pix * L::XP + (((cib << 6) + tr_colb) ^ sw);
    return tr_frag(a, a + 4 * L::XP);  // pixel + 4 has the same swap bit
  };
  auto b_frag = [&](int pl, int q0) -> bf16x8_t {
    const int q = q0 + tr_row;
    const int sw = (L::YP == 128) ? (((q >> 1) & 1) << 6) : 0;
    const char* b = smem + L::YOFF + pl * L::YPLANE + q * L::YP + (((cob << 6) + tr_colb) ^ sw);
    return tr_frag(b, b + 4 * L::YP);
  };

  f32x16 acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const uint32_t first = xcd_remap(blockIdx.x, gridDim.x);
  if (first < (uint32_t)ntiles) load_tile(first);
  for (uint32_t tile = first; tile < (uint32_t)ntiles; tile += gridDim.x) {
    __syncthreads();  // the previous tile's fragment reads are done
    store_tile();
    __syncthreads();
    if (tile + gridDim.x < (uint32_t)ntiles) load_tile(tile + gridDim.x);  // in flight under this tile's MFMAs
#pragma unroll
    for (int si = 0; si < 4 / L::KSPLIT; ++si) {
      const int s = kc + si * L::KSPLIT;  // tile row = k-step (uniform)
      bf16x8_t bf[NPL], af[2][NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) bf[pl] = b_frag(pl, s * 16);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) af[0][pl] = a_frag(pl, s * L::PW);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) {
          const int kh = (tap + 1) / 3, kw = (tap + 1) % 3;
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) af[(tap + 1) & 1][pl] = a_frag(pl, (s + kh) * L::PW + kw);
        }
        const bf16x8_t(&a)[NPL] = af[tap & 1];
        if constexpr (NPL == 1) {
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[0], acc[tap], 0, 0, 0);
        } else {
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], bf[0], acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[2], acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bf[1], acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], bf[0], acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[1], acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bf[0], acc[tap], 0, 0, 0);
        }
      }
    }
  }

  // ---- k classes: class 0 adds classes 1, 2, .. in that order (three taps of every pair per round through LDS)
  if constexpr (L::KSPLIT > 1) {
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int k = 1; k < L::KSPLIT; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        __syncthreads();
        if (kc == k) {
#pragma unroll
          for (int tt = 0; tt < 3; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((g * 3 + tt) * 16 + r) * 64 + lane] = acc[3 * c + tt][r];
        }
        __syncthreads();
        if (kc == 0) {
#pragma unroll
          for (int tt = 0; tt < 3; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[3 * c + tt][r] += red[((g * 3 + tt) * 16 + r) * 64 + lane];
        }
      }
  }
  if (kc == 0) {
    float* out = p.out + (int64_t)blockIdx.x * p.K * p.Cout;
    const int lr = lane & 31, lh = lane >> 5;
    const int col = cob * 32 + lr;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tap * CI + cib * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        out[(int64_t)row * CO + col] = acc[tap][r];
      }
  }
}

inline bool x6wp_enabled() {
  static const bool on = getenv("SG_X6_NOWPATCH") == nullptr;
  return on;
}

// geometry the patch wgrad covers (descriptor level: the plan and the launch must agree)
inline bool x6wp_geom(const sg_conv_desc* d) {
  if (!x6wp_enabled() || d->KH != 3 || d->KW != 3 || d->stride != 1 || d->dilation != 1) return false;
  if (d->pad_t != 1 || d->pad_l != 1 || d->Ho != d->H || d->Wo != d->W) return false;
  if (!(d->Cin == 32 || d->Cin == 64) || !(d->Cout == 32 || d->Cout == 64)) return false;
  if ((d->H % 4) || (d->W % 16)) return false;
  return true;
}
inline int64_t x6wp_tiles(const sg_conv_desc* d) { return (int64_t)d->N * (d->H / 4) * (d->W / 16); }
// workgroups = partial slabs: two per CU, fewer when there are fewer tiles
inline int x6wp_grid(int num_cus, const sg_conv_desc* d) {
  const int64_t tiles = x6wp_tiles(d), slots = 2 * (int64_t)num_cus;
  return (int)(tiles < slots ? tiles : slots);
}

template <int CI, int CO, int NPL, typename TA>
int launch_x6wp_t(const WgradParams& p, int grid, hipStream_t st) {
  using L = X6WP<CI, CO, NPL, TA>;
  static bool attr_done = false;
  if (!attr_done) {
    int rc = set_dyn_lds(wgrad_x6wp_kernel<CI, CO, NPL, TA>, (size_t)L::LDS);
    if (rc) return rc;
    attr_done = true;
  }
  const int tiles_x = p.W / 16, tiles_y = p.H / 4;
  const int64_t ntiles = (int64_t)(p.P / (p.OH * p.OW)) * tiles_x * tiles_y;
  if (ntiles <= 0 || ntiles > 0x7fffffff || grid < 1 || grid > ntiles) {
    sg_set_error("wgrad_x6wp: bad tile count %lld / grid %d", (long long)ntiles, grid);
    return SG_EINVAL;
  }
  hipLaunchKernelGGL((wgrad_x6wp_kernel<CI, CO, NPL, TA>), dim3((unsigned)grid), dim3(256), (size_t)L::LDS, st, p, tiles_x, tiles_y, (int)ntiles);
  SG_LAUNCH_CHECK("wgrad_x6wp_kernel");
  return 0;
}

template <int NPL, typename TA>
int launch_x6wp(const WgradParams& p, int grid, hipStream_t st) {
  if (p.Cin == 32) return p.Cout == 32 ? launch_x6wp_t<32, 32, NPL, TA>(p, grid, st) : launch_x6wp_t<32, 64, NPL, TA>(p, grid, st);
  return p.Cout == 32 ? launch_x6wp_t<64, 32, NPL, TA>(p, grid, st) : launch_x6wp_t<64, 64, NPL, TA>(p, grid, st);
}
