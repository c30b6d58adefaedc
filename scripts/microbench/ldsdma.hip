// Micro-test: semantics of the LDS-DMA load (global_load_lds_dwordx4) on gfx950.
// Every lane passes its own global address (here: piece perm(lane) of a 1 KiB block of 16-bit counters)
// and the wave passes one LDS base; the test dumps where each lane's 16 bytes landed.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned* src, unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned tile[512];
  for (int i = threadIdx.x; i < 512; i += 64) tile[i] = 0xdeadbeefu;
  __syncthreads();
  const int l = threadIdx.x;
  const int piece = (l * 5) & 63;  // a permutation of the 64 sixteen-byte pieces
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 4),
                                   (__attribute__((address_space(3))) void*)(tile + 64), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = tile[i];
}
int main() {
  unsigned h[256];
  for (int i = 0; i < 256; ++i) h[i] = 1000u * (i / 4) + (i % 4);  // piece p holds 1000p .. 1000p+3
  unsigned *d, *o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 2048);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  unsigned r[512]; hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
  for (int i = 0; i < 512; i += 4)
    if (r[i] != 0xdeadbeefu) printf("lds dword %3d: %6u %6u %6u %6u   (piece %u)\n", i, r[i], r[i + 1], r[i + 2], r[i + 3], r[i] / 1000);
  return 0;
}
