// Micro-benchmark: how fast can a few resident workgroups per CU pull a contiguous stream from HBM
// into registers with 4-, 8- and 16-byte loads per lane (NLOADS loads in flight per thread)?
// Build: hipcc -O3 --offload-arch=gfx950 loadwidth.hip -o loadwidth ; run: ./loadwidth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int VEC, int NL, bool ALIGNED>
__global__ void __launch_bounds__(256) k_pull(const float* __restrict__ src, float* __restrict__ out, int chunk_floats, int iters) {
  // workgroup b streams floats [b*chunk, (b+1)*chunk); per iteration 256*NL*VEC floats
  const float* p = src + (size_t)blockIdx.x * chunk_floats + (ALIGNED ? 0 : 1);
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    float v[NL][VEC];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
      const float* q = p + (size_t)it * 256 * NL * VEC + (l * 256 + threadIdx.x) * VEC;
      if constexpr (VEC == 1) v[l][0] = q[0];
      else if constexpr (VEC == 2) { struct __attribute__((packed, aligned(4))) f2 { float a, b; }; f2 t = *reinterpret_cast<const f2*>(q); v[l][0] = t.a; v[l][1] = t.b; }
      else { struct __attribute__((packed, aligned(4))) f4 { float a, b, c, d; }; f4 t = *reinterpret_cast<const f4*>(q); v[l][0] = t.a; v[l][1] = t.b; v[l][2] = t.c; v[l][3] = t.d; }
    }
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int c = 0; c < VEC; ++c) acc += v[l][c];
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int VEC, int NL, bool AL>
void run(const float* src, float* out, size_t total_floats, int wgs, const char* name) {
  int chunk = (int)(total_floats / wgs) - 4;
  int iters = chunk / (256 * NL * VEC);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_pull<VEC, NL, AL>), dim3(wgs), dim3(256), 0, 0, src, out, chunk + 4, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
  }
  float ms; hipEventElapsedTime(&ms, a, b);
  double bytes = (double)wgs * iters * 256 * NL * VEC * 4;
  printf("%-28s wgs %5d  %8.3f ms  %7.2f TB/s\n", name, wgs, ms, bytes / ms / 1e9);
}

int main() {
  size_t total = (size_t)1 << 29;  // 2 GiB of floats
  float *src, *out;
  hipMalloc(&src, total * 4); hipMalloc(&out, 64);
  hipMemset(src, 0, total * 4);
  for (int wgs : {512, 1024, 2048}) {
    run<1, 12, true>(src, out, total, wgs, "dword   x12 in flight");
    run<1, 48, true>(src, out, total, wgs, "dword   x48 in flight");
    run<2, 24, true>(src, out, total, wgs, "dwordx2 x24 in flight");
    run<4, 12, true>(src, out, total, wgs, "dwordx4 x12 in flight");
    run<4, 12, false>(src, out, total, wgs, "dwordx4 x12 unaligned(+4B)");
    run<4, 3, true>(src, out, total, wgs, "dwordx4 x3  in flight");
  }
  return 0;
}
