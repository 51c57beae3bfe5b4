// Experiment: fp32 GEMM accuracy and MFMA rate when an fp32 product is emulated by bf16 MFMAs on a 3-way split
//   x = hi + mid + lo   (each a bf16, round-to-nearest-even splits),
//   a*b ~= hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi     (6 passes; the dropped terms are <= 2^-25 |a||b|)
// against the native v_mfma_f32_32x32x2_f32.  One wave computes one 32x32 tile of C = A[32xK] * B[Kx32].
// Build: hipcc -O3 --offload-arch=gfx950 scripts/exp_bf16x6.hip -o gpurun_out/exp_bf16x6 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ inline unsigned short bf16_rne(float x) {
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ inline float bf16_to_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// mode 0: native fp32 MFMA; 1: bf16x6 one accumulator; 2: bf16x6, low-order terms in a second accumulator;
// 3: bf16x3 (hi/mid only: hi*hi + hi*mid + mid*hi); 4: bf16x9 (all products)
__global__ void gemm_tile(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K, int mode) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc = {0}, acc2 = {0};
  if (mode == 0) {
    for (int k = 0; k < K; k += 2) {
      const float a = A[r * K + k + h], b = B[(k + h) * 32 + r];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  } else {
    for (int k = 0; k < K; k += 16) {
      u16x8 a1, a2, a3, b1, b2, b3;
      for (int j = 0; j < 8; ++j) {
        const float a = A[r * K + k + 8 * h + j], b = B[(k + 8 * h + j) * 32 + r];
        a1[j] = bf16_rne(a); float ra = a - bf16_to_f(a1[j]);
        a2[j] = bf16_rne(ra); ra -= bf16_to_f(a2[j]);
        a3[j] = bf16_rne(ra);
        b1[j] = bf16_rne(b); float rb = b - bf16_to_f(b1[j]);
        b2[j] = bf16_rne(rb); rb -= bf16_to_f(b2[j]);
        b3[j] = bf16_rne(rb);
      }
#define MF(x, y, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0)
      if (mode == 1) {
        MF(a1, b3, acc); MF(a3, b1, acc); MF(a2, b2, acc); MF(a1, b2, acc); MF(a2, b1, acc); MF(a1, b1, acc);
      } else if (mode == 2) {
        MF(a1, b3, acc2); MF(a3, b1, acc2); MF(a2, b2, acc2); MF(a1, b2, acc2); MF(a2, b1, acc2); MF(a1, b1, acc);
      } else if (mode == 3) {
        MF(a1, b2, acc); MF(a2, b1, acc); MF(a1, b1, acc);
      } else {
        MF(a3, b3, acc2); MF(a2, b3, acc2); MF(a3, b2, acc2); MF(a1, b3, acc2); MF(a3, b1, acc2); MF(a2, b2, acc2);
        MF(a1, b2, acc2); MF(a2, b1, acc2); MF(a1, b1, acc);
      }
    }
    for (int i = 0; i < 16; ++i) acc[i] += acc2[i];
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[row * 32 + r] = acc[i];
  }
}

// rate: each wave issues `iters` rounds of NM independent-accumulator MFMAs with operands in registers
template <int BF>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x16){0};
  const float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
  u16x8 ua, ub;
  for (int j = 0; j < 8; ++j) { ua[j] = (unsigned short)(0x3f80 + threadIdx.x + j); ub[j] = 0x3f00; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123.456f) out[0] = s;
}

int main() {
  const int K = 18432;  // the ASPP reduction length
  std::vector<float> A(32 * K), B(K * 32);
  srand(1);
  // activations-like: mixed signs, some large dynamic range
  for (auto& v : A) v = (float)((rand() / (double)RAND_MAX * 2 - 1) * std::exp((rand() / (double)RAND_MAX) * 4 - 2));
  for (auto& v : B) v = (float)((rand() / (double)RAND_MAX * 2 - 1) * 0.02);
  std::vector<double> ref(32 * 32), mag(32 * 32);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double s = 0, m = 0;
      for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + j]; m += std::fabs((double)A[i * K + k] * B[k * 32 + j]); }
      ref[i * 32 + j] = s; mag[i * 32 + j] = m;
    }
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 32 * 32 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  const char* names[] = {"native fp32 MFMA 32x32x2", "bf16x6, one accumulator", "bf16x6, low terms in a 2nd accumulator", "bf16x3 (hi,mid)", "bf16x9 (all products)"};
  std::vector<float> C(32 * 32);
  for (int mode = 0; mode < 5; ++mode) {
    hipLaunchKernelGGL(gemm_tile, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 32 * 32 * 4, hipMemcpyDeviceToHost));
    double emax = 0, esum = 0, rmax = 0;
    for (int i = 0; i < 1024; ++i) {
      const double e = std::fabs(C[i] - ref[i]) / mag[i];  // error relative to sum |a||b|
      emax = std::fmax(emax, e); esum += e;
      rmax = std::fmax(rmax, std::fabs(C[i] - ref[i]) / std::fmax(std::fabs(ref[i]), 1e-30));
    }
    printf("%-42s K=%d  max |err|/sum|ab| = %.3e   mean = %.3e   max |err|/|ref| = %.3e\n", names[mode], K, emax, esum / 1024, rmax);
  }
  // also a float CPU loop (sequential fp32 fma) for scale
  {
    double emax = 0, esum = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[k * 32 + j], s);
        const double e = std::fabs(s - ref[i * 32 + j]) / mag[i * 32 + j];
        emax = std::fmax(emax, e); esum += e;
      }
    printf("%-42s K=%d  max |err|/sum|ab| = %.3e   mean = %.3e\n", "CPU sequential fp32 fmaf", K, emax, esum / 1024);
  }
  // rates
  for (int bf = 0; bf < 2; ++bf) {
    const int iters = 20000, blocks = 256 * 4;  // 4 workgroups of 4 waves per CU => 4 waves per SIMD
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      if (bf) hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, dC, iters);
      else hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, dC, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (double)blocks * 4 * iters * 4 * (bf ? 32.0 * 32 * 16 * 2 : 32.0 * 32 * 2 * 2);
    printf("%s MFMA bare loop: %.2f ms, %.1f TFLOP/s (fp32-equivalent at 6 passes: %.1f)\n", bf ? "bf16 32x32x16" : "fp32 32x32x2", ms, flop / ms / 1e9,
           bf ? flop / ms / 1e9 / 6 : flop / ms / 1e9);
  }
  return 0;
}
