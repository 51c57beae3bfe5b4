// Mask clean-up of the ensemble (model_fuse.py:9-218 of the reference), on the GPU.
//
// The reference walks OpenCV contours object by object on the host.  Here the same decisions are taken from label
// maps (what each OpenCV call means at mask level is spelled out in oracle/cleanup.py):
//
//   stage A, image-wide (sg_mask_objects)            fill_and_delete, model_fuse.py:9-32
//     1. union-find labelling of the BACKGROUND (4-connected) with one extra node for "outside the image": every
//        background component not united with it is a hole and is filled (= cv.fillPoly of the external contours);
//     2. union-find labelling of the filled foreground (8-connected): the objects cv.findContours(RETR_EXTERNAL) reports;
//     3. per object, with integer atomics: bounding box and TWICE cv.contourArea, which for a hole-free region is
//        2*N4 + N3 over the 2x2 pixel quads (N4 quads fully inside, N3 with three pixels inside);
//     4. objects with contourArea > 1000 go to the object table, their pixels carry the table index.
//   stage B, one workgroup per object (sg_mask_split)  eroede_dilate_process, model_fuse.py:65-115, 173-218
//     the object's window (bounding box + 12 px) as a bitmask, 32 pixels per word; erosion / dilation by the 1x5 and 5x1
//     kernels (5 iterations) as shifted ANDs / ORs; pieces taken one at a time by a flood fill from the first set bit;
//     hole filling as a flood of the complement from the window's rim; contourArea of a piece by popcounts of the quad
//     patterns; the keep / replace-by-pieces / drop decision exactly as the reference's if-chain.
//
// HBM-bound integer / bit work: no matrix cores, 16-byte-free scalar accesses on small windows; stage A touches each
// pixel a handful of times, stage B works in L2-resident windows.  Deterministic: every atomic is an integer min / max /
// add, and the object order in the table does not influence the output mask.
#include "sg_common.h"

namespace {

constexpr int OBJ_COLS = 8;   // table row: root pixel, area2, x0, y0, x1, y1, kept, reserved
constexpr int MARGIN = 12;    // window margin: 10 px of dilation overshoot + a background rim for the hole fill

// ------------------------------------------------------------------------------------------------ union-find labelling
__device__ __forceinline__ int uf_find(const int* L, int x) {
  while (true) {
    const int p = __atomic_load_n(&L[x], __ATOMIC_RELAXED);  // other workgroups hang roots under smaller ones meanwhile
    if (p == x) return x;
    x = p;
  }
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  while (true) {
    a = uf_find(L, a);
    b = uf_find(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }  // a > b: hang a under b
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;
  }
}

// target pixels: foreground (fg = 1: m != 0) or background (fg = 0: m == 0); everything else gets parent -1
__global__ void ccl_init_kernel(const unsigned char* __restrict__ m, int* __restrict__ L, int64_t n, int fg) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n) return;
  if (i == n) { L[n] = (int)n; return; }  // the "outside" node of the background labelling
  const bool t = fg ? (m[i] != 0) : (m[i] == 0);
  L[i] = t ? (int)i : -1;
}

// conn8: also the two upper diagonals (8-connectivity); outside: unite image-border pixels with node n
__global__ void ccl_merge_kernel(int* __restrict__ L, int H, int W, int conn8, int outside) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)H * W;
  if (i >= n || L[i] < 0) return;
  const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
  if (x > 0 && L[i - 1] >= 0) uf_union(L, (int)i, (int)i - 1);
  if (y > 0) {
    if (L[i - W] >= 0) uf_union(L, (int)i, (int)(i - W));
    if (conn8) {
      if (x > 0 && L[i - W - 1] >= 0) uf_union(L, (int)i, (int)(i - W - 1));
      if (x < W - 1 && L[i - W + 1] >= 0) uf_union(L, (int)i, (int)(i - W + 1));
    }
  }
  if (outside && (x == 0 || y == 0 || x == W - 1 || y == H - 1)) uf_union(L, (int)i, (int)n);
}

// filled[i] = foreground, or background that is not united with the outside node (a hole)
__global__ void fill_holes_kernel(const unsigned char* __restrict__ m, const int* __restrict__ L, unsigned char* __restrict__ filled,
                                  int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (m[i] != 0) { filled[i] = 255; return; }
  filled[i] = (uf_find(L, (int)i) != uf_find(L, (int)n)) ? 255 : 0;
}

// L[i] <- root of i; every root takes a dense object number from `count` and stores it as -(number + 2)
__global__ void ccl_flatten_kernel(int* __restrict__ L, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || L[i] < 0) return;
  const int r = uf_find(L, (int)i);
  if (r != (int)i) L[i] = r;  // roots stay self-parented: concurrent finds of other pixels still terminate there
}

__global__ void number_roots_kernel(int* __restrict__ L, int64_t n, int* __restrict__ count, int* __restrict__ table, int max_objs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || L[i] != (int)i) return;
  const int k = atomicAdd(count, 1);
  if (k < max_objs) {
    int* row = table + (int64_t)k * OBJ_COLS;
    row[0] = (int)i; row[1] = 0; row[2] = 0x7fffffff; row[3] = 0x7fffffff; row[4] = -1; row[5] = -1; row[6] = 0; row[7] = 0;
  }
  L[i] = -(k + 2);
}

__device__ __forceinline__ int obj_of(const int* __restrict__ L, int64_t i) {
  const int p = L[i];
  if (p == -1) return -1;               // not a foreground pixel
  if (p < -1) return -(p + 2);          // a root
  return -(L[p] + 2);                   // its root carries the number
}

// bounding boxes and 2 x contourArea (quad counts) of every object, labels[i] = object number or -1
__global__ void obj_stats_kernel(const int* __restrict__ L, int H, int W, int* __restrict__ table, int max_objs,
                                 int* __restrict__ labels) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)H * W;
  if (i >= n) return;
  const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
  const int o = obj_of(L, i);
  labels[i] = o < max_objs ? o : -1;
  if (o >= 0 && o < max_objs) {
    int* row = table + (int64_t)o * OBJ_COLS;
    atomicMin(&row[2], x); atomicMin(&row[3], y); atomicMax(&row[4], x); atomicMax(&row[5], y);
  }
  if (x < W - 1 && y < H - 1) {  // the quad whose top-left pixel is (y, x)
    const int a = o, b = obj_of(L, i + 1), c = obj_of(L, i + W), d = obj_of(L, i + W + 1);
    const int cnt = (a >= 0) + (b >= 0) + (c >= 0) + (d >= 0);
    if (cnt >= 3) {  // the foreground pixels of a quad are mutually 8-adjacent: one object
      const int q = a >= 0 ? a : b;
      if (q < max_objs) atomicAdd(&table[(int64_t)q * OBJ_COLS + 1], cnt == 4 ? 2 : 1);
    }
  }
}

__global__ void obj_keep_kernel(int* __restrict__ table, int nobj, int min_area2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < nobj) table[(int64_t)k * OBJ_COLS + 6] = table[(int64_t)k * OBJ_COLS + 1] > min_area2 ? 1 : 0;
}

// gray_label of fill_and_delete: 255 on the pixels of kept objects
__global__ void kept_mask_kernel(const int* __restrict__ labels, const int* __restrict__ table, int64_t n,
                                 unsigned char* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int o = labels[i];
  out[i] = (o >= 0 && table[(int64_t)o * OBJ_COLS + 6]) ? 255 : 0;
}

// ------------------------------------------------------------------------------------------ stage B: window bit planes
struct Win {
  int ww, wh, wpr, nwords;
  unsigned lastmask;
  int blast;                 // bit index of the window's last column inside the last word
  bool eL, eR, eT, eB;       // the window edge is the IMAGE edge (pixels beyond it never erode)
};

constexpr int NT = 256;

// value of plane P at row r shifted so that bit j holds the pixel s columns to the RIGHT (s > 0) or LEFT (s < 0) of
// column 32k + j; columns beyond the window read `fill` (0 or 1)
__device__ __forceinline__ unsigned shx(const unsigned* __restrict__ P, const Win& w, int r, int k, int s, unsigned fill) {
  const unsigned* row = P + (int64_t)r * w.wpr;
  const unsigned cur = row[k];
  unsigned v;
  if (s > 0) {
    unsigned nxt = (k + 1 < w.wpr) ? row[k + 1] : (fill ? 0xffffffffu : 0u);
    if (k == w.wpr - 1 && fill) {  // the bits of the last word beyond the window are "fill" too
      const unsigned beyond = ~w.lastmask;
      v = ((cur | beyond) >> s) | (nxt << (32 - s));
    } else {
      if (k + 1 == w.wpr - 1 && fill) nxt |= ~w.lastmask;
      v = (cur >> s) | (nxt << (32 - s));
    }
  } else {
    const int t = -s;
    const unsigned prv = (k > 0) ? row[k - 1] : (fill ? 0xffffffffu : 0u);
    v = (cur << t) | (prv >> (32 - t));
  }
  return v;
}

__device__ __forceinline__ unsigned rowv(const unsigned* __restrict__ P, const Win& w, int r, int k, unsigned fillT, unsigned fillB) {
  if (r < 0) return fillT ? 0xffffffffu : 0u;
  if (r >= w.wh) return fillB ? 0xffffffffu : 0u;
  return P[(int64_t)r * w.wpr + k];
}

__device__ __forceinline__ unsigned clip_last(const Win& w, int k, unsigned v) { return k == w.wpr - 1 ? (v & w.lastmask) : v; }

// one erosion (AND) or dilation (OR) iteration by the 1x5 (axis 1) or 5x1 (axis 0) kernel: dst = op(src)
__device__ void morph5(const unsigned* __restrict__ src, unsigned* __restrict__ dst, const Win& w, int axis, bool erode) {
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) {
    const int r = idx / w.wpr, k = idx - r * w.wpr;
    unsigned v = src[idx];
    if (axis == 1) {
      const unsigned fl = (erode && w.eL) ? 1u : 0u, fr = (erode && w.eR) ? 1u : 0u;
      const unsigned a = shx(src, w, r, k, -1, fl), b = shx(src, w, r, k, -2, fl), c = shx(src, w, r, k, 1, fr), d = shx(src, w, r, k, 2, fr);
      v = erode ? (v & a & b & c & d) : (v | a | b | c | d);
    } else {
      const unsigned ft = (erode && w.eT) ? 1u : 0u, fb = (erode && w.eB) ? 1u : 0u;
      const unsigned a = rowv(src, w, r - 1, k, ft, fb), b = rowv(src, w, r - 2, k, ft, fb), c = rowv(src, w, r + 1, k, ft, fb),
                     d = rowv(src, w, r + 2, k, ft, fb);
      v = erode ? (v & a & b & c & d) : (v | a | b | c | d);
    }
    dst[idx] = clip_last(w, k, v);
  }
  __syncthreads();
}

// five iterations, ping-pong between a and b; the result ends in `a`'s partner after an odd count: returns the pointer
__device__ unsigned* morph5x5(unsigned* a, unsigned* b, const Win& w, int axis, bool erode) {
  unsigned* s = a;
  unsigned* d = b;
  for (int it = 0; it < 5; ++it) {
    morph5(s, d, w, axis, erode);
    unsigned* t = s; s = d; d = t;
  }
  return s;
}

__device__ void plane_copy(const unsigned* __restrict__ src, unsigned* __restrict__ dst, const Win& w) {
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) dst[idx] = src[idx];
  __syncthreads();
}
__device__ void plane_zero(unsigned* __restrict__ dst, const Win& w) {
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) dst[idx] = 0u;
  __syncthreads();
}
__device__ void plane_or(unsigned* __restrict__ dst, const unsigned* __restrict__ src, const Win& w) {
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) dst[idx] |= src[idx];
  __syncthreads();
}
__device__ void plane_andnot(unsigned* __restrict__ dst, const unsigned* __restrict__ src, const Win& w) {
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) dst[idx] &= ~src[idx];
  __syncthreads();
}

__device__ int block_min(int v, int* sm) {
  if (threadIdx.x == 0) *sm = 0x7fffffff;
  __syncthreads();
  if (v != 0x7fffffff) atomicMin(sm, v);
  __syncthreads();
  const int r = *sm;
  __syncthreads();
  return r;
}
__device__ int block_sum(int v, int* sm) {
  if (threadIdx.x == 0) *sm = 0;
  __syncthreads();
  if (v) atomicAdd(sm, v);
  __syncthreads();
  const int r = *sm;
  __syncthreads();
  return r;
}

// index of the first set bit of P in raster order, or -1
__device__ int first_bit(const unsigned* __restrict__ P, const Win& w, int* sm) {
  int best = 0x7fffffff;
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) {
    const unsigned v = P[idx];
    if (v) { best = idx * 32 + __builtin_ctz(v); break; }  // a thread's words come in increasing order
  }
  const int r = block_min(best, sm);
  return r == 0x7fffffff ? -1 : r;
}

// flood of `seed` inside `allowed` (both planes; seed is grown in place): 4- or 8-connected.  Every word is written by
// its owner thread only; neighbours are read while they grow, which only speeds the fixpoint up.
__device__ void flood(unsigned* __restrict__ seed, const unsigned* __restrict__ allowed, const Win& w, bool conn8, int* sm) {
  while (true) {
    int changed = 0;
    for (int idx = threadIdx.x; idx < w.nwords; idx += NT) {
      const int r = idx / w.wpr, k = idx - r * w.wpr;
      const unsigned al = allowed[idx];
      if (!al) continue;
      unsigned s = seed[idx];
      unsigned nb = shx(seed, w, r, k, -1, 0) | shx(seed, w, r, k, 1, 0);
      if (r > 0) {
        nb |= seed[idx - w.wpr];
        if (conn8) nb |= shx(seed, w, r - 1, k, -1, 0) | shx(seed, w, r - 1, k, 1, 0);
      }
      if (r < w.wh - 1) {
        nb |= seed[idx + w.wpr];
        if (conn8) nb |= shx(seed, w, r + 1, k, -1, 0) | shx(seed, w, r + 1, k, 1, 0);
      }
      unsigned g = (s | nb) & al;
      // run along the row inside this word to the fixpoint
      for (int t = 0; t < 32; ++t) {
        const unsigned g2 = (g | (g << 1) | (g >> 1)) & al;
        if (g2 == g) break;
        g = g2;
      }
      if (g != s) { seed[idx] = g; changed = 1; }
    }
    if (!block_sum(changed, sm)) break;
  }
}

// filled = src with its holes filled: the complement's 4-connected flood from the window rim marks the outside
__device__ void fill_holes_plane(const unsigned* __restrict__ src, unsigned* __restrict__ outside, unsigned* __restrict__ notsrc,
                                 unsigned* __restrict__ filled, const Win& w, int* sm) {
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) {
    const int r = idx / w.wpr, k = idx - r * w.wpr;
    const unsigned ns = clip_last(w, k, ~src[idx]);
    notsrc[idx] = ns;
    unsigned rim = 0u;
    if (r == 0 || r == w.wh - 1) rim = 0xffffffffu;
    if (k == 0) rim |= 1u;
    if (k == w.wpr - 1) rim |= 1u << w.blast;
    outside[idx] = ns & clip_last(w, k, rim);
  }
  __syncthreads();
  flood(outside, notsrc, w, false, sm);
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) {
    const int k = idx % w.wpr;
    filled[idx] = clip_last(w, k, ~outside[idx]);
  }
  __syncthreads();
}

// 2 x contourArea of a hole-free plane: 2 * (quads with 4 pixels) + (quads with 3)
__device__ int area2_plane(const unsigned* __restrict__ F, const Win& w, int* sm) {
  int acc = 0;
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) {
    const int r = idx / w.wpr, k = idx - r * w.wpr;
    if (r >= w.wh - 1) continue;
    const unsigned a = F[idx], a1 = shx(F, w, r, k, 1, 0), b = F[idx + w.wpr], b1 = shx(F, w, r + 1, k, 1, 0);
    const unsigned c4 = a & a1 & b & b1;
    const unsigned c3 = (a & a1 & b & ~b1) | (a & a1 & ~b & b1) | (a & ~a1 & b & b1) | (~a & a1 & b & b1);
    acc += 2 * __builtin_popcount(c4) + __builtin_popcount(c3);
  }
  return block_sum(acc, sm);
}

// erode_process / erode_process1 for the object in plane OBJ.  Returns 0 = None, 1 = False, 2 = list; for a list the
// union of the dilated-and-filled pieces is OR-ed into RES.
__device__ int split_axis(const unsigned* OBJ, unsigned* RES, unsigned* T0, unsigned* T1, unsigned* PIECE, unsigned* OUTS,
                          unsigned* NOTS, unsigned* FILL, unsigned* KEPT, const Win& w, int axis, int piece_area2, int* sm) {
  plane_copy(OBJ, T0, w);
  unsigned* ER = morph5x5(T0, T1, w, axis, true);
  unsigned* SCR = (ER == T0) ? T1 : T0;
  plane_zero(KEPT, w);
  int n = 0, nsmall = 0, nkept = 0;
  while (true) {
    const int fb = first_bit(ER, w, sm);
    if (fb < 0) break;
    plane_zero(PIECE, w);
    if (threadIdx.x == 0) PIECE[fb >> 5] = 1u << (fb & 31);
    __syncthreads();
    flood(PIECE, ER, w, true, sm);                           // the 8-connected piece: a top-level one (raster-first)
    fill_holes_plane(PIECE, OUTS, NOTS, FILL, w, sm);        // fill_small_target's fillPoly
    plane_andnot(ER, FILL, w);                               // the piece and whatever it encloses are dealt with
    const int a2 = area2_plane(FILL, w, sm);
    ++n;
    if (a2 <= piece_area2) ++nsmall;
    else { ++nkept; plane_or(KEPT, FILL, w); }
  }
  if (n == 1) return 0;
  if (nsmall > 0 && nkept == 0) return 1;
  // dilate_process: every kept piece on its own (they are not 8-adjacent, so they are the components of KEPT)
  while (true) {
    const int fb = first_bit(KEPT, w, sm);
    if (fb < 0) break;
    plane_zero(PIECE, w);
    if (threadIdx.x == 0) PIECE[fb >> 5] = 1u << (fb & 31);
    __syncthreads();
    flood(PIECE, KEPT, w, true, sm);
    plane_andnot(KEPT, PIECE, w);
    plane_copy(PIECE, SCR, w);
    unsigned* other = ER;                                    // ER is empty by now: free scratch
    unsigned* DIL = morph5x5(SCR, other, w, axis, false);
    fill_holes_plane(DIL, OUTS, NOTS, FILL, w, sm);          // drawContours(FILLED) of the dilated piece's external contour
    plane_or(RES, FILL, w);
  }
  return 2;
}

constexpr int NPLANES = 10;

__global__ __launch_bounds__(NT) void split_objects_kernel(const int* __restrict__ labels, const int* __restrict__ table,
                                                           const int* __restrict__ objs, const int64_t* __restrict__ offs, int H,
                                                           int W, int piece_area2, unsigned* __restrict__ ws,
                                                           unsigned char* __restrict__ out) {
  __shared__ int sm;
  const int o = objs[blockIdx.x];
  const int* row = table + (int64_t)o * OBJ_COLS;
  const int x0 = max(row[2] - MARGIN, 0), y0 = max(row[3] - MARGIN, 0), x1 = min(row[4] + MARGIN, W - 1), y1 = min(row[5] + MARGIN, H - 1);
  Win w;
  w.ww = x1 - x0 + 1; w.wh = y1 - y0 + 1; w.wpr = (w.ww + 31) >> 5; w.nwords = w.wh * w.wpr;
  w.blast = (w.ww - 1) & 31;
  w.lastmask = w.blast == 31 ? 0xffffffffu : ((1u << (w.blast + 1)) - 1u);
  w.eL = x0 == 0; w.eT = y0 == 0; w.eR = x1 == W - 1; w.eB = y1 == H - 1;
  unsigned* base = ws + offs[blockIdx.x];
  unsigned* P[NPLANES];
  for (int i = 0; i < NPLANES; ++i) P[i] = base + (int64_t)i * w.nwords;
  unsigned *OBJ = P[0], *RESH = P[1], *RESV = P[2], *T0 = P[3], *T1 = P[4], *PIECE = P[5], *OUTS = P[6], *NOTS = P[7], *FILL = P[8],
           *KEPT = P[9];
  // the object's pixels as bits
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) {
    const int r = idx / w.wpr, k = idx - r * w.wpr;
    unsigned v = 0u;
    const int64_t rowoff = (int64_t)(y0 + r) * W + x0 + 32 * k;
    const int nb = min(32, w.ww - 32 * k);
    for (int j = 0; j < nb; ++j) v |= (labels[rowoff + j] == o ? 1u : 0u) << j;
    OBJ[idx] = v;
    RESH[idx] = 0u;
    RESV[idx] = 0u;
  }
  __syncthreads();
  const int sh = split_axis(OBJ, RESH, T0, T1, PIECE, OUTS, NOTS, FILL, KEPT, w, 1, piece_area2, &sm);  // erode_process: 1x5
  const int sv = split_axis(OBJ, RESV, T0, T1, PIECE, OUTS, NOTS, FILL, KEPT, w, 0, piece_area2, &sm);  // erode_process1: 5x1
  // eroede_dilate_process's if-chain (model_fuse.py:184-216)
  const unsigned* R0 = nullptr;
  const unsigned* R1 = nullptr;
  if (sh == 1 || sv == 1) {
    // an axis whose pieces were all small: the object is dropped
  } else if (sh == 0 && sv == 0) {
    R0 = OBJ;
  } else {
    if (sh == 2) R0 = RESH;
    if (sv == 2) R1 = RESV;
  }
  for (int idx = threadIdx.x; idx < w.nwords; idx += NT) {
    unsigned v = (R0 ? R0[idx] : 0u) | (R1 ? R1[idx] : 0u);
    if (!v) continue;
    const int r = idx / w.wpr, k = idx - r * w.wpr;
    const int64_t rowoff = (int64_t)(y0 + r) * W + x0 + 32 * k;
    while (v) {
      const int j = __builtin_ctz(v);
      v &= v - 1;
      out[rowoff + j] = 255;  // windows of neighbouring objects may overlap: everybody writes the same value
    }
  }
}

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

extern "C" {

size_t sg_mask_objects_ws_bytes(int H, int W) {
  const int64_t n = (int64_t)H * W;
  if (n <= 0) return 0;
  return (size_t)(n + 1) * sizeof(int) + (size_t)n + 512;  // parent array (+ outside node), filled mask
}

int sg_mask_objects(sg_ctx* ctx, void* stream, int H, int W, const void* mask_u8, int min_area2, void* ws, size_t ws_bytes,
                    void* labels_i32, void* table_i32, int max_objs, void* count_i32, void* kept_mask_u8) {
  SG_CHECK_ARG(ctx && mask_u8 && ws && labels_i32 && table_i32 && count_i32 && H > 0 && W > 0 && max_objs > 0,
               "sg_mask_objects: bad argument");
  const int64_t n = (int64_t)H * W;
  SG_CHECK_ARG(n < (1ll << 31) - 2, "sg_mask_objects: image exceeds 2^31 pixels");
  if (ws_bytes < sg_mask_objects_ws_bytes(H, W)) {
    sg_set_error("sg_mask_objects: workspace %zu < %zu", ws_bytes, sg_mask_objects_ws_bytes(H, W));
    return SG_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int* L = (int*)ws;
  unsigned char* filled = (unsigned char*)ws + (((size_t)(n + 1) * sizeof(int) + 255) & ~(size_t)255);
  const unsigned char* m = (const unsigned char*)mask_u8;
  // 1. holes: background components (4-connected) that do not reach the image border
  hipLaunchKernelGGL(ccl_init_kernel, dim3(blocks_for(n + 1)), dim3(256), 0, st, m, L, n, 0);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(blocks_for(n)), dim3(256), 0, st, L, H, W, 0, 1);
  hipLaunchKernelGGL(fill_holes_kernel, dim3(blocks_for(n)), dim3(256), 0, st, m, (const int*)L, filled, n);
  // 2. objects: 8-connected components of the filled mask
  hipLaunchKernelGGL(ccl_init_kernel, dim3(blocks_for(n + 1)), dim3(256), 0, st, (const unsigned char*)filled, L, n, 1);
  hipLaunchKernelGGL(ccl_merge_kernel, dim3(blocks_for(n)), dim3(256), 0, st, L, H, W, 1, 0);
  hipLaunchKernelGGL(ccl_flatten_kernel, dim3(blocks_for(n)), dim3(256), 0, st, L, n);
  hipError_t e = hipMemsetAsync(count_i32, 0, sizeof(int), st);
  if (e != hipSuccess) { sg_set_error("sg_mask_objects: hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
  hipLaunchKernelGGL(number_roots_kernel, dim3(blocks_for(n)), dim3(256), 0, st, L, n, (int*)count_i32, (int*)table_i32, max_objs);
  // 3. bounding boxes and 2 x contourArea; 4. the keep flag
  hipLaunchKernelGGL(obj_stats_kernel, dim3(blocks_for(n)), dim3(256), 0, st, (const int*)L, H, W, (int*)table_i32, max_objs,
                     (int*)labels_i32);
  hipLaunchKernelGGL(obj_keep_kernel, dim3(blocks_for(max_objs)), dim3(256), 0, st, (int*)table_i32, max_objs, min_area2);
  if (kept_mask_u8)
    hipLaunchKernelGGL(kept_mask_kernel, dim3(blocks_for(n)), dim3(256), 0, st, (const int*)labels_i32, (const int*)table_i32, n,
                       (unsigned char*)kept_mask_u8);
  SG_LAUNCH_CHECK("sg_mask_objects");
  return 0;
}

int64_t sg_mask_split_words(int H, int W, int x0, int y0, int x1, int y1) {
  const int wx0 = x0 - MARGIN < 0 ? 0 : x0 - MARGIN, wy0 = y0 - MARGIN < 0 ? 0 : y0 - MARGIN;
  const int wx1 = x1 + MARGIN > W - 1 ? W - 1 : x1 + MARGIN, wy1 = y1 + MARGIN > H - 1 ? H - 1 : y1 + MARGIN;
  const int64_t wpr = (wx1 - wx0 + 1 + 31) / 32;
  return (int64_t)NPLANES * wpr * (wy1 - wy0 + 1);
}

int sg_mask_split(sg_ctx* ctx, void* stream, int H, int W, const void* labels_i32, const void* table_i32, const void* objs_i32,
                  const void* offsets_i64, int nobj, int piece_area2, void* ws, void* out_u8) {
  SG_CHECK_ARG(ctx && labels_i32 && table_i32 && out_u8 && H > 0 && W > 0 && nobj >= 0, "sg_mask_split: bad argument");
  if (nobj == 0) return 0;
  SG_CHECK_ARG(objs_i32 && offsets_i64 && ws, "sg_mask_split: null object list / workspace");
  hipLaunchKernelGGL(split_objects_kernel, dim3((unsigned)nobj), dim3(NT), 0, (hipStream_t)stream, (const int*)labels_i32,
                     (const int*)table_i32, (const int*)objs_i32, (const int64_t*)offsets_i64, H, W, piece_area2, (unsigned*)ws,
                     (unsigned char*)out_u8);
  SG_LAUNCH_CHECK("split_objects_kernel");
  return 0;
}

}  // extern "C"
