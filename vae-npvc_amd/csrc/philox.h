// philox.h -- counter-based N(0,1) generator for the sampler (util/layers.py:154: tf.random_normal,
// unseeded in the reference).  Philox4x32-10 (Salmon et al., SC'11; the generator behind
// tf.random_normal's GPU kernel as well), Box-Muller on 24-bit uniforms.
//
// Element e of a [rows, cols] draw uses counter (e >> 2, offset) and key = seed; one Philox call yields the four
// normals of elements 4q .. 4q+3, so any thread can produce any element without state:
//   (x0, x1) -> n0 = r cos(2 pi u2), n1 = r sin(2 pi u2),  r = sqrt(-2 ln u1),  u = ((x >> 8) + 0.5) * 2^-24
//   (x2, x3) -> n2, n3 likewise.
// The test suite pins the block function against the published Random123 known-answer vectors and
// holds a NumPy restatement of the whole draw.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace vaenpvc {

struct PhiloxKey {
  uint32_t seed_lo, seed_hi;  // key
  uint32_t off_lo, off_hi;    // counter words 2, 3 (the step counter: a fresh draw per training step)
  const int64_t* d_off;       // optional device counter ADDED to the offset when the kernel runs (graph replay)
};
// offset resolved on the device (call once per thread)
__device__ __forceinline__ PhiloxKey philox_resolve(PhiloxKey k) {
  if (k.d_off) {
    uint64_t o = ((uint64_t)k.off_hi << 32 | k.off_lo) + (uint64_t)*k.d_off;
    k.off_lo = (uint32_t)o;
    k.off_hi = (uint32_t)(o >> 32);
  }
  return k;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0;
    c[1] = lo1;
    c[2] = n2;
    c[3] = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__device__ __forceinline__ float philox_u24(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// the normal of flat element index e
__device__ __forceinline__ float philox_normal(const PhiloxKey& k, uint64_t e) {
  const uint64_t q = e >> 2;
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), k.off_lo, k.off_hi};
  philox4x32_10(c, k.seed_lo, k.seed_hi);
  const int j = (int)(e & 3);
  const float u1 = philox_u24(j < 2 ? c[0] : c[2]), u2 = philox_u24(j < 2 ? c[1] : c[3]);
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincospif(2.0f * u2, &sn, &cs);
  return r * ((j & 1) ? sn : cs);
}

// U[0,1) of flat element index e (tf.random_uniform's range): word e & 3 of counter (e >> 2, offset), 24 bits
__device__ __forceinline__ float philox_uniform(const PhiloxKey& k, uint64_t e) {
  const uint64_t q = e >> 2;
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), k.off_lo, k.off_hi};
  philox4x32_10(c, k.seed_lo, k.seed_hi);
  return (float)(c[e & 3] >> 8) * (1.0f / 16777216.0f);
}

}  // namespace vaenpvc
