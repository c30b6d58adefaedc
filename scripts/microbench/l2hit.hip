// Micro-benchmark: what does a CU get from its XCD's L2 when the data IS resident?  Every workgroup re-reads the same
// small buffer (REGION bytes, shared by all workgroups -> resident in every XCD's 4 MB L2 after the first pass) with
// 16-byte loads, NL loads in flight per thread, either into registers or straight into LDS (global_load_lds_dwordx4).
// Compare with loadwidth.hip (HBM stream: ~6 TB/s = ~10 B/clk/CU).
// Build: hipcc -O3 --offload-arch=gfx950 l2hit.hip -o l2hit ; run: ./l2hit
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NL, bool DMA>
__global__ void __launch_bounds__(256) k_l2(const u32x4* __restrict__ src, unsigned* __restrict__ out, int region16, int iters) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned acc = 0;
  const int tid = threadIdx.x, wave = tid >> 6;
  int pos = (blockIdx.x * 977) % region16;   // different workgroups start at different places of the region
  for (int it = 0; it < iters; ++it) {
    if constexpr (DMA) {
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        int i = pos + l * 256 + tid;
        i = i >= region16 ? i - region16 : i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i),
                                         (__attribute__((address_space(3))) void*)(smem + (l * 4 + wave) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[NL];
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        int i = pos + l * 256 + tid;
        i = i >= region16 ? i - region16 : i;
        v[l] = src[i];
      }
#pragma unroll
      for (int l = 0; l < NL; ++l) acc += v[l][0] ^ v[l][3];
    }
    pos += NL * 256;
    pos = pos >= region16 ? pos - region16 : pos;
  }
  if (DMA) acc = reinterpret_cast<unsigned*>(smem)[tid];
  if (acc == 0x12345678u) out[0] = acc;
}

template <int NL, bool DMA>
void run(const u32x4* src, unsigned* out, size_t region_bytes, int wgs, int iters, const char* name) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int lds = DMA ? NL * 4 * 1024 : 0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_l2<NL, DMA>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_l2<NL, DMA>), dim3(wgs), dim3(256), lds, 0, src, out, (int)(region_bytes / 16), iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
  }
  double bytes = (double)wgs * iters * 256 * NL * 16;
  printf("%-30s region %6zu KiB wgs %5d  %8.3f ms  %7.2f TB/s  = %5.1f B/clk/CU (2.4 GHz, 256 CUs)\n", name, region_bytes >> 10, wgs, ms,
         bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  u32x4* src; unsigned* out;
  hipMalloc(&src, 256 << 20); hipMalloc(&out, 64);
  hipMemset(src, 0, 256 << 20);
  for (size_t region : {(size_t)256 << 10, (size_t)2 << 20, (size_t)16 << 20, (size_t)128 << 20}) {
    for (int wgs : {512, 1024}) {
      run<4, false>(src, out, region, wgs, 4000, "regs, 4 x 16 B in flight");
      run<12, false>(src, out, region, wgs, 1500, "regs, 12 x 16 B in flight");
      run<8, true>(src, out, region, wgs, 2000, "LDS-DMA, 8 x 16 B in flight");
    }
  }
  return 0;
}
