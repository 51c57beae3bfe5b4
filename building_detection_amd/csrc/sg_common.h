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

// ---- storage element types: fp32, or bf16 storage with fp32 arithmetic (SG_BF16) ---------------------------------
// Every kernel computes in fp32 registers; T only says how an activation element lies in HBM.  A "chunk" is the
// 16-byte access unit: 4 floats or 8 bf16.
typedef __bf16 bf16_t;
typedef unsigned int u32x2_c __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_c __attribute__((ext_vector_type(4)));

template <typename T>
struct EL;
template <>
struct EL<float> {
  static constexpr int BYTES = 4, CH = 4, DT = SG_F32;
};
template <>
struct EL<bf16_t> {
  static constexpr int BYTES = 2, CH = 8, DT = SG_BF16;
};

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }
// round-to-nearest-even through the hardware conversion (v_cvt_pk_bf16_f32 keeps NaN a NaN)
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  return __builtin_bit_cast(unsigned short, (bf16_t)f);
}
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
  typedef float f32x2_c __attribute__((ext_vector_type(2)));
  typedef bf16_t bf16x2_c __attribute__((ext_vector_type(2)));
  const f32x2_c v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_c));
}

template <typename T>
__device__ __forceinline__ float ld1(const T* __restrict__ p) {
  if constexpr (sizeof(T) == 4) return *p;
  else return bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(p));
}
template <typename T>
__device__ __forceinline__ void st1(T* __restrict__ p, float v) {
  if constexpr (sizeof(T) == 4) *p = v;
  else *reinterpret_cast<unsigned short*>(p) = f32_to_bf16_bits(v);
}
// four consecutive elements (16-byte or 8-byte aligned access)
template <typename T>
__device__ __forceinline__ f32x4 ld4(const T* __restrict__ p) {
  if constexpr (sizeof(T) == 4) {
    return *reinterpret_cast<const f32x4*>(p);
  } else {
    const u32x2_c r = *reinterpret_cast<const u32x2_c*>(p);
    f32x4 o = {__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
               __uint_as_float(r[1] & 0xffff0000u)};
    return o;
  }
}
template <typename T>
__device__ __forceinline__ void st4(T* __restrict__ p, const f32x4 v) {
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
    const u32x2_c r = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
    *reinterpret_cast<u32x2_c*>(p) = r;
  }
}
// eight consecutive elements of a bf16 tensor (one 16-byte access) <-> two f32x4
__device__ __forceinline__ void ld8_bf16(const bf16_t* __restrict__ p, f32x4& a, f32x4& b) {
  const u32x4_c r = *reinterpret_cast<const u32x4_c*>(p);
  a = (f32x4){__uint_as_float(r[0] << 16), __uint_as_float(r[0] & 0xffff0000u), __uint_as_float(r[1] << 16),
              __uint_as_float(r[1] & 0xffff0000u)};
  b = (f32x4){__uint_as_float(r[2] << 16), __uint_as_float(r[2] & 0xffff0000u), __uint_as_float(r[3] << 16),
              __uint_as_float(r[3] & 0xffff0000u)};
}
__device__ __forceinline__ void st8_bf16(bf16_t* __restrict__ p, const f32x4 a, const f32x4 b) {
  const u32x4_c r = {pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]), pack2_bf16(b[0], b[1]), pack2_bf16(b[2], b[3])};
  *reinterpret_cast<u32x4_c*>(p) = r;
}

// host-side dispatch on the ABI's dtype argument: CALL is a generic lambda taking a null pointer of the element type
#define SG_DTYPE_SWITCH(dtype, who, ...)                                          \
  do {                                                                            \
    if ((dtype) == SG_F32) { typedef float T; __VA_ARGS__; }                      \
    else if ((dtype) == SG_BF16) { typedef bf16_t T; __VA_ARGS__; }               \
    else { sg_set_error("%s: dtype %d", who, (int)(dtype)); return SG_EINVAL; }   \
  } while (0)
