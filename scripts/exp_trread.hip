// Experiment: semantics of ds_read_b64_tr_b16 (transposed LDS read) as used by the x6 wgrad kernel.
// LDS image [32 rows][32 cols] of 16-bit values row*100 + col; lane l = 16 g + i supplies the address of row
// 8*(g>>1) + (i>>2), columns 16*(g&1) + 4*(i&3) .. +3 and should receive column 16*(g&1) + i of rows 8*(g>>1)+0..3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32 * 32];
  for (int i = threadIdx.x; i < 32 * 32; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, pp = i & 3;
  const int row = 8 * (g >> 1) + q, col = 16 * (g & 1) + 4 * pp;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + row * 32 + col));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short h[1024], o[256], *di, *dout;
  for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) h[r * 32 + c] = r * 100 + c;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, i = l & 15;
    for (int j = 0; j < 4; ++j) {
      const int want = (8 * (g >> 1) + j) * 100 + 16 * (g & 1) + i;
      if (o[l * 4 + j] != want) ++bad;
    }
    if (l % 8 == 0) printf("lane %2d: %4d %4d %4d %4d\n", l, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  }
  printf("transposed read: %d mismatches against the expected mapping\n", bad);
  return bad != 0;
}
