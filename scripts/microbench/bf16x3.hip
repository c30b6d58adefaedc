// Micro-test: fp32-accurate GEMM out of bf16 MFMAs.  Each fp32 operand is split into three bf16
// terms (hi, mid, lo: 8+8+8 mantissa bits); a product keeps the six term pairs with i+j <= 2.
// Checks (1) the operand layout of v_mfma_f32_32x32x16_bf16, (2) the error against an fp64
// reference for the 6-product and the cheaper 3-product variants, (3) the issue rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf16_rn(float x) {  // round to nearest even
  unsigned u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  h = bf16_rn(x);
  float r = x - bf16_f(h);
  m = bf16_rn(r);
  r = r - bf16_f(m);
  l = bf16_rn(r);
}

union Frag { bf16x8 v; unsigned short s[8]; };

// one wave: C[32][32] = A[32][K] * B[K][32]; mode 6 = six products, 3 = three, 1 = plain bf16
__global__ void k_gemm(const float* A, const float* B, float* C, int K, int mode) {
  const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    Frag a[3], b[3];
    for (int j = 0; j < 8; ++j) {
      int k = k0 + 8 * lh + j;
      split3(A[l31 * K + k], a[0].s[j], a[1].s[j], a[2].s[j]);
      split3(B[k * 32 + l31], b[0].s[j], b[1].s[j], b[2].s[j]);
    }
    if (mode >= 6) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
    }
    if (mode >= 3) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[0].v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    C[row * 32 + l31] = acc[r];
  }
}

// issue-rate loop: NACC independent accumulators, 6 MFMAs per step from fixed registers
template <int NACC>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters) {
  Frag a[3], b[3];
  for (int p = 0; p < 3; ++p)
    for (int j = 0; j < 8; ++j) { a[p].s[j] = (unsigned short)(0x3f80 + threadIdx.x + p); b[p].s[j] = (unsigned short)(0x3f80 + j + p); }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) {
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[0].v, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[2].v, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[0].v, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[1].v, acc[n], 0, 0, 0);
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc[n], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
  if (s == 1234.5f) out[0] = s;
}

int main() {
  const int K = 4112;
  std::vector<float> A(32 * K), B(K * 32), C(1024);
  srand(1);
  for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  std::vector<double> R(1024, 0.0);
  std::vector<float> R32(1024, 0.f);
  double amax = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = 0; float s32 = 0.f;
    for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 32 + j]; s32 = fmaf(A[i * K + k], B[k * 32 + j], s32); }
    R[i * 32 + j] = s; R32[i * 32 + j] = s32; amax = fmax(amax, fabs(s));
  }
  double e32 = 0; for (int i = 0; i < 1024; ++i) e32 = fmax(e32, fabs(R32[i] - R[i]));
  printf("reference max |C| = %.4f ; sequential fp32 fma max err / max|C| = %.3e\n", amax, e32 / amax);
  for (int mode : {6, 3, 1}) {
    hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, mode);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    double e = 0; for (int i = 0; i < 1024; ++i) e = fmax(e, fabs(C[i] - R[i]));
    printf("bf16 split, %d products: max err / max|C| = %.3e\n", mode, e / amax);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<2>, dim3(256 * 2), dim3(256), 0, 0, dC, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double mf = 512.0 * 4 * iters * 2 * 6;  // MFMAs
  printf("rate: %.1f TFLOP/s bf16 dense (%.1f cycles per MFMA per SIMD at 2.33 GHz, 2 WGs/CU)\n", mf * 32768 / ms / 1e9, ms * 1e-3 * 2.33e9 / (iters * 2 * 6 * 2));
  return 0;
}
