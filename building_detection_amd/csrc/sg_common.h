// Shared host/device helpers for libsegengine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/segengine.h"

struct sg_ctx {
  int device;
  int num_cus;
};

// thread-local error message, defined in api.hip
void sg_set_error(const char* fmt, ...);

#define SG_CHECK_ARG(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      sg_set_error(__VA_ARGS__);           \
      return SG_EINVAL;                    \
    }                                      \
  } while (0)

#define SG_LAUNCH_CHECK(name)                                             \
  do {                                                                    \
    hipError_t e__ = hipGetLastError();                                   \
    if (e__ != hipSuccess) {                                              \
      sg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return (int)e__;                                                    \
    }                                                                     \
  } while (0)

static inline int64_t sg_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- unsigned division by a runtime-constant divisor (n < 2^31), precomputed on the host -------------
struct FastDiv {
  uint32_t mul;
  uint32_t shift;
  uint32_t d;
};

static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;  // ceil(log2(d))
  f.shift = l;
  f.mul = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
  return f;
}

__device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
  return (__umulhi(n, f.mul) + n) >> f.shift;
}

__device__ __forceinline__ void fd_divmod(uint32_t n, const FastDiv& f, uint32_t& q, uint32_t& r) {
  q = fd_div(n, f);
  r = n - q * f.d;
}

// XCD-aware block remap (8 XCDs; blocks b and b+8 share an XCD): give each XCD a contiguous chunk of the
// logical tile order so neighbouring tiles hit the same L2.  Bijective for any nwg (guide §5 / T1).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
  const uint32_t q = nwg >> 3, r = nwg & 7;
  const uint32_t xcd = bid & 7, idx = bid >> 3;
  const uint32_t base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
