// Micro-test: semantics of ds_read_b64_tr_b16 (LDS transpose read) on gfx950.
// LDS holds tile[r][c] = 100*r + c as 16-bit integers, row stride RS elements.  Lane i of each
// 16-lane group points at 4 contiguous elements tile[k0 + i/4][c0 + 4*(i%4)]; the hardware is expected
// to hand lane l the column c0 + (l&15): tile[k0+0..3][c0 + l].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int RS = 40;  // elements per row (80 bytes, multiple of 8)
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short tile[16 * RS];
  for (int i = threadIdx.x; i < 16 * RS; i += 64) tile[i] = (short)(100 * (i / RS) + (i % RS));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  const int k0 = (g >> 1) * 8, c0 = (g & 1) * 16;      // groups: (k 0, c 0) (k 0, c 16) (k 8, c 0) (k 8, c 16)
  const short* p = &tile[(k0 + i / 4) * RS + c0 + 4 * (i % 4)];
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  return 0;
}
