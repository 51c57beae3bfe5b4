// Experiment (round 3): does the x6 inner loop run faster on v_mfma_f32_16x16x32_bf16 than on v_mfma_f32_32x32x16_bf16?
// Both shapes take the same cycles per FLOP; MI355X_MICROARCH.md ("DVFS give-back", item 7) reports that the chip holds a
// higher clock on the 16x16x32 shape under load (1.12-1.15x the FLOP/s in bare loops on random data).  Same set-up as
// exp_x6_loop.hip: one workgroup = a 128 x 128 tile, planes [3][128 rows][80 B] for A and B in LDS, random operands, no global
// traffic, 8 waves of 64 x 32, fragments double-buffered (MODE 1 of exp_x6_loop.hip), one or two workgroups per CU; the two
// shapes alternate inside one process (rule 24).  A 32-deep slab is 2 k-steps of 32x32x16 (2 x 2 x 6 = 24 MFMAs of 32 cycles
// per wave) or 1 k-step of 16x16x32 (4 x 2 x 6 = 48 MFMAs of 16 cycles); LDS reads per slab are 18 ds_read_b128 either way.
// Build: hipcc -O3 --offload-arch=gfx950 scripts/exp_x6_shape.hip -o build/exp_x6_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BM = 128, BN = 128, PITCH = 80;

__device__ void fill_lds(char* smem, int nt) {
  for (int i = threadIdx.x; i < 3 * (BM + BN) * PITCH / 4; i += nt) {
    unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    reinterpret_cast<unsigned*>(smem)[i] = (h & 0x807f807fu) | 0x3f003f00u | ((h >> 3) & 0x00800080u);
  }
  __syncthreads();
}

// ---- 32x32x16: wave tile 64 x 32 = 2 x 1 blocks, two k-steps per slab
__global__ __launch_bounds__(512, 2) void loop32(float* out, int slabs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ap = smem; char* Bp = Ap + 3 * BM * PITCH;
  fill_lds(smem, 512);
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 31, lh = lane >> 5;
  const int wm = (wave / 4) * 64, wn = (wave % 4) * 32;
  const char* a_lane = Ap + (wm + lr) * PITCH + lh * 16;
  const char* b_lane = Bp + (wn + lr) * PITCH + lh * 16;
  f32x16 acc[2]; acc[0] = (f32x16){0}; acc[1] = (f32x16){0};
  auto rd = [&](bf16x8 (&af)[2][3], bf16x8 (&bf)[3], int ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(a_lane + (pl * BM + 32 * i) * PITCH + ks * 32);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) bf[pl] = *reinterpret_cast<const bf16x8*>(b_lane + (pl * BN) * PITCH + ks * 32);
  };
  auto mm = [&](bf16x8 (&af)[2][3], bf16x8 (&bf)[3]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[0], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[2], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[1], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[0], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[1], acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[0], acc[i], 0, 0, 0);
    }
  };
  bf16x8 af0[2][3], bf0[3], af1[2][3], bf1[3];
  rd(af0, bf0, 0);
  for (int s = 0; s < slabs; ++s) {
    rd(af1, bf1, 1);
    __builtin_amdgcn_sched_barrier(0);
    mm(af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    rd(af0, bf0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mm(af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
  }
  float sum = 0.f;
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
  if (sum == 1.2345f) out[0] = sum;
}

// ---- 16x16x32: wave tile 64 x 32 = 4 x 2 blocks of 16 x 16, one k-step per slab; lane -> row lane & 15, k chunk lane >> 4
__global__ __launch_bounds__(512, 2) void loop16(float* out, int slabs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ap = smem; char* Bp = Ap + 3 * BM * PITCH;
  fill_lds(smem, 512);
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63, lr = lane & 15, lq = lane >> 4;
  const int wm = (wave / 4) * 64, wn = (wave % 4) * 32;
  const char* a_lane = Ap + (wm + lr) * PITCH + lq * 16;
  const char* b_lane = Bp + (wn + lr) * PITCH + lq * 16;
  f32x4 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0};
  auto rd = [&](bf16x8 (&af)[4][3], bf16x8 (&bf)[2][3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(a_lane + (pl * BM + 16 * i) * PITCH);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8*>(b_lane + (pl * BN + 16 * j) * PITCH);
  };
  auto mm = [&](bf16x8 (&af)[4][3], bf16x8 (&bf)[2][3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
      }
  };
  bf16x8 af0[4][3], bf0[2][3], af1[4][3], bf1[2][3];
  rd(af0, bf0);
  for (int s = 0; s < slabs; s += 2) {
    rd(af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
    mm(af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    rd(af0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    mm(af1, bf1);
    __builtin_amdgcn_sched_barrier(0);
  }
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
  if (sum == 1.2345f) out[0] = sum;
}

template <typename K>
double run(K kern, const char* name, int wg_per_cu, float* dout) {
  const int slabs = 4000;
  const size_t lds = 3 * (BM + BN) * PITCH;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms = 0;
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kern, dim3(256 * wg_per_cu), dim3(512), lds, 0, dout, slabs);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 256.0 * wg_per_cu * slabs * 2.0 * 128 * 128 * 32 * 6;
  printf("%-10s %d WG/CU: %7.2f ms  bf16 %7.1f TFLOP/s  fp32-equivalent %6.1f TFLOP/s\n", name, wg_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 6);
  return ms;
}

int main() {
  float* dout; CK(hipMalloc(&dout, 64));
  for (int wg = 1; wg <= 2; ++wg) {
    run(loop32, "warm-up", wg, dout);
    double a = 0, b = 0;
    for (int rep = 0; rep < 4; ++rep) {   // interleaved rounds
      a += run(loop32, "32x32x16", wg, dout);
      b += run(loop16, "16x16x32", wg, dout);
    }
    printf("== %d WG/CU: 16x16x32 / 32x32x16 time ratio %.3f (FLOP/s ratio %.3f)\n", wg, b / a, a / b);
  }
  return 0;
}
