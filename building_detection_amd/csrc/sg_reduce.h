// Segmented column reductions over NHWC data: out[seg][o][c] = sum_{r < rows} f_o(seg, r, c).
//
// This one shape covers every "squeeze" on the path: BatchNorm statistics (1 segment, all pixels),
// GlobalAveragePooling / AveragePooling2D (segment = output pixel), the channel-gate gradients of cSE / BAM /
// SK (segment = image), the depthwise-conv kernel gradient (9 outputs per channel) and bias gradients.
//
// Mapping: a 256-thread block is TX x TY with TX threads on consecutive V-wide channel chunks (coalesced
// 16-byte loads along C) and TY threads on rows; each thread keeps NOUT x V partial sums in registers, the TY
// partials are combined through LDS in a fixed order, and when the rows are split over S workgroups
// (grid.z) the S partial rows are added in fixed order (in fp64) by the finalize kernel.  Results are
// therefore bit-reproducible run to run (no float atomics).
#pragma once
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <utility>
#include "sg_common.h"

// An Op may carry per-thread constants through the row loop (per-channel parameters loaded once instead of once per
// row): it then declares  template <int V> CtxType begin(int seg, int c) const  and takes the context as accum's last
// argument.
template <class Op, int V, class = void>
struct seg_has_ctx : std::false_type {};
template <class Op, int V>
struct seg_has_ctx<Op, V, std::void_t<decltype(std::declval<const Op&>().template begin<V>(0, 0))>> : std::true_type {};

// `fuse`: the LAST workgroup of a (segment, column block) to finish - found with one agent-scope counter per column
// block, cnt[seg * gridDim.x + blockIdx.x], zero before the launch and reset by that workgroup - also does the second
// stage for its channels (same lanes, same order of additions as seg_finalize_kernel: bit-identical), so that the
// few-microsecond finalize launch behind every reduction disappears.  With S == 1 (fuse = 1) there is nothing to wait
// for and no counter.  For S > 1 the partial rows travel between workgroups (and XCDs, whose L2s are not coherent):
//   fuse = 2 (rounds 2 / 3): plain stores, an agent-scope RELEASE fence in every workgroup before the counter increment,
//     an acquire after it.  Correct, and a loss: the release is buffer_wbl2 - every one of ~1000 workgroups per launch
//     writes back its XCD's whole L2, which holds the previous kernel's output = this kernel's input (step 79 -> 102 ms).
//   fuse = 3 (round 4): the hand-off of MI355X_MICROARCH.md "Valid forms" without any cache-wide operation - every
//     partial is stored WRITE-THROUGH (global_store sc1: __hip_atomic_store relaxed / agent), every storing wave drains
//     its stores (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane's agent-scope atomic add whose returned value tells the
//     last arriver, workgroup barrier, and the last arriver reads every partial with sc1 loads (bypassing its L1; an
//     agent-scope acquire - buffer_inv sc1, this CU's L1 only, ~2 us in that one workgroup - is kept in front of them
//     because several of these 256-thread workgroups share a CU, which is outside the table of measured sc1-only forms).
template <class Op, int V>
__global__ __launch_bounds__(256) void seg_reduce_kernel(const Op op, const int64_t rows, const int C, const int S,
                                                         float* __restrict__ part, unsigned* __restrict__ cnt, const int fuse) {
  constexpr int NO = Op::NOUT;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);  // [TY][TX][NO*V]
  const int TX = blockDim.x, TY = blockDim.y;
  const int tx = threadIdx.x, ty = threadIdx.y;
  const int c = (blockIdx.x * TX + tx) * V;
  const int seg = blockIdx.y, z = blockIdx.z;
  float acc[NO][V];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int v = 0; v < V; ++v) acc[o][v] = 0.f;
  if (c < C) {
    const int64_t per = (rows + S - 1) / S;
    const int64_t rb = (int64_t)z * per;
    int64_t re = rb + per;
    if (re > rows) re = rows;
    // four rows per trip so four independent 16-byte loads per operand are in flight per thread (the accum bodies are
    // branch-free for that reason: a run-time `if` around a load puts it in its own basic block behind a full wait)
    // (Ops whose single row already keeps a few dozen loads in flight - the depthwise filter gradient - set NOUT >= 9 and
    // are not unrolled: four interleaved rows would need 300 registers)
    constexpr int UNR = NO >= 9 ? 1 : 4;
    if constexpr (seg_has_ctx<Op, V>::value) {
      const auto ctx = op.template begin<V>(seg, c);
#pragma unroll UNR
      for (int64_t r = rb + ty; r < re; r += TY) op.template accum<V>(seg, r, c, acc, ctx);
    } else {
#pragma unroll UNR
      for (int64_t r = rb + ty; r < re; r += TY) op.template accum<V>(seg, r, c, acc);
    }
  }
  float* mine = red + ((size_t)(ty * TX + tx)) * (NO * V);
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int v = 0; v < V; ++v) mine[o * V + v] = acc[o][v];
  __syncthreads();
  const bool wt = fuse == 3 && S > 1;   // write-through partials (uniform)
  if (ty == 0 && c < C) {
#pragma unroll
    for (int o = 0; o < NO; ++o)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float s = 0.f;
        for (int y = 0; y < TY; ++y) s += red[((size_t)(y * TX + tx)) * (NO * V) + o * V + v];
        if (c + v < C) {
          float* dst = part + (((int64_t)seg * S + z) * NO + o) * C + c + v;
          if (wt) __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else *dst = s;
        }
      }
  }
  if (!fuse) return;
  if (S > 1) {
    __shared__ int s_last;
    if (wt) {
      // fuse == 3 (ADVICE r4 asked for a release here, or the reason why none is needed).  The reason: this is the
      // write-through hand-off of MI355X_MICROARCH.md, "Valid forms", first row of its table - EVERY handed-off byte is
      // stored sc1 (the agent-scope relaxed atomic stores above compile to global_store_dword ... sc1: they go through to the
      // fabric, nothing is left dirty in this XCD's L2 for a release to write back), every storing wave drains them
      // (s_waitcnt vmcnt(0)), the workgroup barrier below puts the ONE signalling lane behind every wave's drain, its
      // agent-scope atomic add returns the arrival count, and only the workgroup whose add came last reads - behind an agent
      // ACQUIRE (kept, stricter than the table's sc1 loads).  An agent-scope RELEASE on the add would put buffer_wbl2 (write
      // back this XCD's whole L2) into ~1000 workgroups per launch - the 25 ms that mode 2 costs and this mode exists to
      // avoid.  tests/test_ops_gpu.py::test_fused_second_stage_of_the_reductions_is_bit_identical runs mode 3 against mode 0
      // on multi-slab BatchNormalization / depthwise filter-gradient reductions, bit for bit.  Off by default (no gain).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores ...
    } else {
      // release only: round 2 used __threadfence() (acquire AND release at agent scope: write back and invalidate this XCD's
      // L2 in every workgroup; step 86 -> 117 ms).  Round 3 re-measured with the scoped pair: still 79 -> 102 ms (fp32), 34 ->
      // 52 ms (bf16) - the L2 writeback of ~1000 workgroups per launch is the cost, whatever it finds to write.
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    }
    __syncthreads();                                      // ... before the one lane that signals for all of them
    if (tx == 0 && ty == 0) {
      unsigned* slot = cnt + (size_t)seg * gridDim.x + blockIdx.x;
      const unsigned prev = __hip_atomic_fetch_add(slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_last = (prev == (unsigned)S - 1u) ? 1 : 0;
      if (s_last) __hip_atomic_store(slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every other workgroup of this column block has already arrived
    }
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the other workgroups' partial rows, not stale cache lines
    if (wt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the invalidate completes asynchronously)
      __syncthreads();
    }
  } else {
    __syncthreads();  // S == 1: the partial row just written by this workgroup's ty == 0 threads
  }
  // second stage for the TX * V channels of this column block, 64 at a time: lane zl of channel fx adds the partial rows
  // z = zl, zl + 4, .. in fp64, the four lane sums are combined ((0 + 1) + 2) + 3 - exactly seg_finalize_kernel
  double* dred = reinterpret_cast<double*>(smem_raw);  // [4][64][NO]
  const int tid = ty * TX + tx, fx = tid & 63, zl = tid >> 6;
  const int nch = TX * V, cb = blockIdx.x * nch;
  const float* __restrict__ rd = part;
  for (int ch0 = 0; ch0 < nch; ch0 += 64) {
    const int cc = cb + ch0 + fx;
    const bool act = (ch0 + fx < nch) && cc < C;
    double sum[NO];
#pragma unroll
    for (int o = 0; o < NO; ++o) sum[o] = 0.0;
    if (act) {
      constexpr int UNROLL = NO <= 2 ? 8 : 1;
#pragma unroll UNROLL
      for (int z2 = zl; z2 < S; z2 += 4) {
        const float* q = rd + (((int64_t)seg * S + z2) * NO) * C + cc;
#pragma unroll
        for (int o = 0; o < NO; ++o)
          sum[o] += (double)(wt ? __hip_atomic_load(q + (int64_t)o * C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : q[(int64_t)o * C]);
      }
    }
    __syncthreads();  // the first stage's (or the previous 64 channels') use of the LDS buffer is over
#pragma unroll
    for (int o = 0; o < NO; ++o) dred[(zl * 64 + fx) * NO + o] = sum[o];
    __syncthreads();
    if (zl == 0 && act) {
#pragma unroll
      for (int o = 0; o < NO; ++o)
        sum[o] = ((dred[(0 * 64 + fx) * NO + o] + dred[(1 * 64 + fx) * NO + o]) + dred[(2 * 64 + fx) * NO + o]) + dred[(3 * 64 + fx) * NO + o];
      op.finalize(seg, cc, sum);
    }
  }
}

// Second stage: a 64 (channels) x 4 (split lanes) block adds the S partial rows of its channels in fp64 — each
// lane a fixed subset z = lane, lane+4, ..., the four lane sums combined in fixed order through LDS — and
// hands the NOUT totals of each channel to op.finalize.  Reads are coalesced along C.
// LANES = 16 (round 4; seg_finalize_launch picks it for S >= 32): 16 channels x 16 lanes per block.  With 4 lanes a launch over
// 256 partial rows x 728 channels is 12 workgroups whose lanes each walk 64 rows one behind the other: 16 - 19 us for 1.5 MB
// (profiles/r04_bench_kernel_stats_final.csv: 140 such launches per step); 16 lanes walk 16 rows each in 46 workgroups.  The
// lane sums are added in lane order: deterministic, but another order than with 4 lanes - a launch's LANES depends on S only.
template <class Op, int LANES = 4>
__global__ __launch_bounds__(256) void seg_finalize_kernel(const Op op, const int nseg, const int C, const int S,
                                                           const float* __restrict__ part) {
  constexpr int NO = Op::NOUT;
  constexpr int CH = 256 / LANES;   // channels per block
  __shared__ double red[LANES][CH][NO];
  const int tx = threadIdx.x % CH, zl = threadIdx.x / CH;
  const int c = blockIdx.x * CH + tx, seg = blockIdx.y;
  double s[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) s[o] = 0.0;
  if (c < C) {
    // unrolled for the narrow ops so that the partial rows' loads are in flight together (the adds stay in order): BN
    // backward's finalize 15 -> 6 us; with NOUT = 9 (depthwise wgrad) the same unroll spills and runs 5x slower
    constexpr int UNROLL = NO <= 2 ? 8 : 1;
#pragma unroll UNROLL
    for (int z = zl; z < S; z += LANES) {
      const float* p = part + (((int64_t)seg * S + z) * NO) * C + c;
#pragma unroll
      for (int o = 0; o < NO; ++o) s[o] += (double)p[(int64_t)o * C];
    }
  }
#pragma unroll
  for (int o = 0; o < NO; ++o) red[zl][tx][o] = s[o];
  __syncthreads();
  if (zl == 0 && c < C) {
#pragma unroll
    for (int o = 0; o < NO; ++o) {
      double t = red[0][tx][o];
#pragma unroll
      for (int l = 1; l < LANES; ++l) t += red[l][tx][o];   // ((0 + 1) + 2) + 3 ...
      s[o] = t;
    }
    op.finalize(seg, c, s);
  }
}

// the finalize launch: 16 lanes per channel where there are many partial rows
template <class Op>
static inline void seg_finalize_launch(const Op& op, int nseg, int C, int S, const float* part, hipStream_t st) {
  static const int wide = getenv("SG_FINALIZE_LANES") ? atoi(getenv("SG_FINALIZE_LANES")) : 16;   // A/B switch: 4 = rounds 1 - 3
  if (S >= 32 && wide == 16)
    hipLaunchKernelGGL((seg_finalize_kernel<Op, 16>), dim3((unsigned)sg_cdiv(C, 16), (unsigned)nseg), dim3(256), 0, st, op, nseg, C, S, part);
  else
    hipLaunchKernelGGL((seg_finalize_kernel<Op, 4>), dim3((unsigned)sg_cdiv(C, 64), (unsigned)nseg), dim3(256), 0, st, op, nseg, C, S, part);
}

struct SegPlan {
  int V, TX, TY, gx, S;
  size_t part_bytes;
};

// wide8: the reduced tensor is bf16 and C % 8 == 0: a lane takes 8 channels (one 16-byte access) instead of 4 (8 bytes)
template <int NOUT>
static inline SegPlan seg_plan(int num_cus, int nseg, int64_t rows, int C, bool vec_ok, bool wide8 = false) {
  SegPlan pl;
  pl.V = (vec_ok && C % 4 == 0) ? 4 : 1;
  if (pl.V == 4 && wide8 && C % 8 == 0 && NOUT <= 2) pl.V = 8;
  const int chunks = C / pl.V;
  // 16 lanes x 16 B = one 256-byte run per row; narrow blocks keep gx (and so the block count) high without
  // a large row split S, which the second stage would have to add up again
  int tx = 1;
  while (tx < chunks && tx < 16) tx <<= 1;
  pl.TX = tx;
  pl.TY = 256 / tx;
  pl.gx = (int)sg_cdiv(chunks, tx);
  // S depends on the segment's own size only, never on how many segments (images) the launch carries: a
  // per-image reduction then adds in the same order whatever the batch, so inference is bit-exactly
  // batch-slice invariant (tests/test_fullsize_gpu.py)
  int64_t S = sg_cdiv((int64_t)4 * num_cus, (int64_t)pl.gx);
  const int64_t maxS = sg_cdiv(rows, (int64_t)pl.TY * 4);
  if (S > maxS) S = maxS;
  // short segments (<= 16 rows per thread: the per-tile BatchNorm statistics, pooled maps, gates): one workgroup per
  // column block, which then finalises its channels itself - one launch of a few microseconds instead of two
  if (rows <= (int64_t)pl.TY * 16) S = 1;
  if (S > 1024) S = 1024;  // few-channel maps at full resolution (C = 32: one column block) need the row split
                           // for occupancy: 1024 workgroups = 4 per CU
  if (S < 1) S = 1;
  pl.S = (int)S;
  pl.part_bytes = (size_t)nseg * pl.S * NOUT * C * sizeof(float);
  return pl;
}

// Arrival counters of the fused second stage: n zeroed words out of a per-device ring (this translation unit's own).
// The kernels leave their words zero again, and launches that may overlap (different streams, captured graphs) sit at
// different ring positions; nullptr (too many column blocks, no device memory, allocation refused inside a stream capture)
// sends the launch down the two-kernel path.
static inline unsigned* seg_counters(int64_t n) {
  constexpr unsigned RING = 1u << 18;  // words: 1 MiB
  struct Ring { unsigned* buf = nullptr; unsigned pos = 0; bool failed = false; };
  static Ring ring[16];
  static std::mutex mu;
  int dev = 0;
  if (n < 1 || n > 8192 || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  Ring& r = ring[dev];
  if (!r.buf) {
    if (r.failed) return nullptr;
    unsigned* q = nullptr;
    if (hipMalloc(&q, (size_t)RING * 4) != hipSuccess || hipMemset(q, 0, (size_t)RING * 4) != hipSuccess) {
      (void)hipGetLastError();
      if (q) (void)hipFree(q);
      r.failed = true;
      return nullptr;
    }
    r.buf = q;
  }
  if (r.pos + (unsigned)n > RING) r.pos = 0;
  unsigned* out = r.buf + r.pos;
  r.pos += (unsigned)n;
  return out;
}

// Launch the reduction (second stage fused into it, see seg_reduce_kernel; SG_SEG_FUSED=0: separate finalize launch).
// `part` must hold pl.part_bytes.
template <class Op>
static inline int seg_reduce_launch(const Op& op, const SegPlan& pl, int nseg, int64_t rows, int C, float* part,
                                    hipStream_t st, const char* name) {
  if (nseg <= 0 || rows <= 0 || C <= 0) return 0;
  // SG_SEG_FUSED: 0 = always the separate finalize launch, 1 = fused when S == 1 (no communication between workgroups),
  // 2 = fused for every S through the arrival counters with release / acquire fences, 3 = the same with write-through
  // partials and no cache-wide fence (seg_reduce_kernel's comment).  Mode 2 is correct (full GPU suite) but a LOSS on this
  // part: the agent-scope release / acquire pair makes every workgroup write back and invalidate its XCD's whole L2 -
  // which holds the previous kernel's output, i.e. this kernel's input - and the DeepLabv3+ step went 86 -> 117 ms (fp32),
  // 39 -> 63 ms (bf16).  Kept for re-measurement only.
  static const int fused_mode = getenv("SG_SEG_FUSED") ? atoi(getenv("SG_SEG_FUSED")) : 1;
  unsigned* cnt = nullptr;
  int fuse = 0;
  if (fused_mode >= 1 && pl.S == 1) fuse = 1;
  else if (fused_mode >= 2 && (cnt = seg_counters((int64_t)nseg * pl.gx)) != nullptr) fuse = fused_mode >= 3 ? 3 : 2;
  const size_t lds1 = (size_t)256 * Op::NOUT * pl.V * sizeof(float), lds2 = (size_t)4 * 64 * Op::NOUT * sizeof(double);
  const size_t lds = lds1 > lds2 ? lds1 : lds2;
  dim3 grid((unsigned)pl.gx, (unsigned)nseg, (unsigned)pl.S), block((unsigned)pl.TX, (unsigned)pl.TY);
  if (pl.V == 8) {
    if constexpr (Op::NOUT <= 2) hipLaunchKernelGGL((seg_reduce_kernel<Op, 8>), grid, block, lds, st, op, rows, C, pl.S, part, cnt, fuse);
  } else if (pl.V == 4)
    hipLaunchKernelGGL((seg_reduce_kernel<Op, 4>), grid, block, lds, st, op, rows, C, pl.S, part, cnt, fuse);
  else
    hipLaunchKernelGGL((seg_reduce_kernel<Op, 1>), grid, block, lds, st, op, rows, C, pl.S, part, cnt, fuse);
  SG_LAUNCH_CHECK(name);
  if (!fuse) {
    seg_finalize_launch(op, nseg, C, pl.S, (const float*)part, st);
    SG_LAUNCH_CHECK(name);
  }
  return 0;
}

// vector load/store helpers (V = 4: 16-byte access; V = 1: scalar; V = 8: two 16-byte accesses)
template <int V>
__device__ __forceinline__ void ldv(const float* __restrict__ p, float (&o)[V]) {
  if constexpr (V == 8) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  } else if constexpr (V == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
  } else {
    o[0] = p[0];
  }
}
template <int V>
__device__ __forceinline__ void stv(float* __restrict__ p, const float (&o)[V]) {
  if constexpr (V == 8) {
    const f32x4 a = {o[0], o[1], o[2], o[3]}, b = {o[4], o[5], o[6], o[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  } else if constexpr (V == 4) {
    f32x4 t = {o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  } else {
    p[0] = o[0];
  }
}

// bf16 storage: V = 4 is one 8-byte access, V = 1 a 2-byte one; values widen to fp32 in registers
template <int V>
__device__ __forceinline__ void ldv(const bf16_t* __restrict__ p, float (&o)[V]) {
  if constexpr (V == 8) {
    f32x4 a, b;
    ld8_bf16(p, a, b);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
  } else if constexpr (V == 4) {
    const f32x4 t = ld4<bf16_t>(p);
    o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3];
  } else {
    o[0] = ld1<bf16_t>(p);
  }
}
template <int V>
__device__ __forceinline__ void stv(bf16_t* __restrict__ p, const float (&o)[V]) {
  if constexpr (V == 8) {
    const f32x4 a = {o[0], o[1], o[2], o[3]}, b = {o[4], o[5], o[6], o[7]};
    st8_bf16(p, a, b);
  } else if constexpr (V == 4) {
    const f32x4 t = {o[0], o[1], o[2], o[3]};
    st4<bf16_t>(p, t);
  } else {
    st1<bf16_t>(p, o[0]);
  }
}

static inline bool sg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
