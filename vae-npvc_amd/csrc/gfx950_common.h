// gfx950_common.h -- device helpers shared by the tuned CDNA4 kernels.
//
// MFMA used throughout: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles/SIMD, 157 TF chip
// peak).  Operand maps (guides/cdna_hip_programming.md section 3):
//   A: lane l holds A[i = l&31][k = l>>5]        B: lane l holds B[k = l>>5][j = l&31]
//   C/D (16 regs): col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "runtime.h"

namespace vaenpvc {
namespace tuned {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LN_EPS 1e-5f
#define LEAK 0.02f
#define EPSILON 1e-6f

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}
// Streaming accesses: tensors of hundreds of MB that are touched once per kernel gain nothing from the caches; the
// nontemporal forms of the loads and stores (measured on the LayerNorm backward passes alone: 7.73 -> 7.64 ms per step)
// keep them from evicting what IS reused.  One switch per kernel group so that each can be A/B-built
// (scripts/build_variant.sh NAME "-DVAENPVC_NT_x=0", scripts/ab_libs.sh): A = LayerNorm backward passes (AS: their store), B = plane producers (measured: +35 us per step, the consumer finds freshly written planes in the Infinity Cache
// otherwise -- off), T = result stores of the 1025-tap layer's input gradient (-15 us).  Tried and removed: the fused conv kernels' result
// stores as nontemporal 4-byte stores (+310 us), their staging loads (no difference), encoder layer 0's store (+60 us) and
// its backward kernel's loads (no difference), the loads of the statistics + plane pass of decoder layer 2 (no difference),
// the result stores of the plane GEMMs (+60 us) and of the view GEMMs (+370 us).
#ifndef VAENPVC_NT_A
#define VAENPVC_NT_A 1
#endif
#ifndef VAENPVC_NT_B
#define VAENPVC_NT_B 0
#endif
#ifndef VAENPVC_NT_AS
#define VAENPVC_NT_AS 1
#endif
#ifndef VAENPVC_NT_T
#define VAENPVC_NT_T 1
#endif
// 16 bytes at 4-byte alignment (rows of 513 floats): the nontemporal builtins take vector types, not structs
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <int ON, class T>
__device__ __forceinline__ T ld_nt(const T* p) {
  if constexpr (ON) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int ON, class T>
__device__ __forceinline__ void st_nt(T* p, T v) {
  if constexpr (ON) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// row of accumulator register `reg` for this lane inside a 32x32 tile
__device__ __forceinline__ int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// make LDS writes of this wave visible to its own later LDS reads (cross-lane)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float lnact_v(float v, float mean, float rstd, float g, float b) {
  float n = (v - mean) * rstd * g + b;
  return fmaxf(n, LEAK * n);
}

// Sum over the 64 lanes of a wave, result in every lane.  Four DPP adds inside each row of 16 lanes (quad_perm, row
// mirrors), then the four row sums are read into scalar registers: ~15 instructions of a few cycles each, against six
// dependent ds_bpermute round trips through the LDS pipeline (~100 cycles each) for the butterfly of __shfl_xor -- the
// reductions sit on the per-frame latency chain of every kernel that takes LayerNorm statistics or sums.
// PRECONDITION: all 64 lanes active (full EXEC), i.e. every call site is wave-uniform.  With bound_ctrl a disabled lane
// contributes 0 to the DPP steps, but v_readlane of lanes 0 / 16 / 32 / 48 returns the stale register of an inactive lane:
// a call under a divergent branch returns a wrong sum without a sign.  -DVAENPVC_WAVE_SUM_ASSERT traps on a partial EXEC
// mask (debug builds); VAENPVC_WAVE_SUM_DPP=0 is the shuffle butterfly, which tolerates divergence.
#ifndef VAENPVC_WAVE_SUM_DPP
#define VAENPVC_WAVE_SUM_DPP 1
#endif
__device__ __forceinline__ float wave_sum(float v) {
#if VAENPVC_WAVE_SUM_DPP
#ifdef VAENPVC_WAVE_SUM_ASSERT
  if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
#endif
  auto dpp = [](float x, auto ctrl) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror: every lane holds the sum of its row of 16
  const int iv = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
  return (r0 + r1) + (r2 + r3);
#else
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
#endif
}

// Cooperative, latency-tolerant copy of `total` contiguous floats HBM -> (functor): every
// thread first issues BATCH independent 16-byte (VEC = 4, base 16-byte aligned, total % 4 == 0)
// or 4-byte loads, then hands the values to put(element_index, value).
template <int VEC, int BATCH, int NTHR = 256, class Put>
__device__ __forceinline__ void stage_range(const float* __restrict__ src, int total, Put&& put) {
  const int tid = threadIdx.x;
  if constexpr (VEC == 4) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    const int n4 = total >> 2;
    for (int i0 = tid; i0 < n4; i0 += NTHR * BATCH) {
      float4 v[BATCH];
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        int i = i0 + NTHR * b;
        v[b] = i < n4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        int i = i0 + NTHR * b;
        if (i < n4) {
          put(4 * i, v[b].x);
          put(4 * i + 1, v[b].y);
          put(4 * i + 2, v[b].z);
          put(4 * i + 3, v[b].w);
        }
      }
    }
  } else {
    for (int i0 = tid; i0 < total; i0 += NTHR * BATCH) {
      float v[BATCH];
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        int i = i0 + NTHR * b;
        v[b] = i < total ? src[i] : 0.f;
      }
#pragma unroll
      for (int b = 0; b < BATCH; ++b) {
        int i = i0 + NTHR * b;
        if (i < total) put(i, v[b]);
      }
    }
  }
}

constexpr int cmax_c(int a, int b) { return a > b ? a : b; }
constexpr int cmin_c(int a, int b) { return a < b ? a : b; }
// Row-wise staging HBM -> LDS for tensors whose rows (ROWLEN contiguous floats) keep their order in
// LDS: one wave copies RU rows per trip; the row index is wave-uniform, so row base addresses and
// the LayerNorm constants live in scalar registers (LN folded to one fma: y = lrelu(v*sc + sh),
// exactly the form tf.nn.batch_normalization uses).  rowinfo(r, src_off, dst_off, sc, sh).
template <int ROWLEN, int NWAVES, bool LN, class RowInfo>
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, float* __restrict__ dst, int nrows,
                                           RowInfo&& rowinfo) {
  constexpr int PER = (ROWLEN + 63) / 64;            // loads per row and lane
  constexpr int RU = cmax_c(2, cmin_c(16, 32 / PER)); // rows in flight per wave (~32 loads per lane)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int r0 = wave * RU; r0 < nrows; r0 += NWAVES * RU) {
    float v[RU][PER];
    int doff[RU];
    float sc[RU], sh[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      int r = r0 + u < nrows ? r0 + u : nrows - 1;   // clamp: duplicates are harmless
      int soff;
      rowinfo(r, soff, doff[u], sc[u], sh[u]);
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        int i = lane + 64 * p;
        v[u][p] = (i < ROWLEN) ? src[soff + i] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < RU; ++u)
#pragma unroll
      for (int p = 0; p < PER; ++p) {
        int i = lane + 64 * p;
        float x = v[u][p];
        if constexpr (LN) {
          x = x * sc[u] + sh[u];
          x = fmaxf(x, LEAK * x);
        }
        if (i < ROWLEN) dst[doff[u] + i] = x;
      }
  }
}

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int rup(int a, int b) { return cdiv(a, b) * b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }
constexpr int cmin_(int a, int b) { return a < b ? a : b; }
// smallest value >= v that is congruent to r modulo 32 (LDS bank period for 4-byte accesses)
constexpr int next_mod32(int v, int r) { return v + ((r - v) % 32 + 32) % 32; }

}  // namespace tuned
}  // namespace vaenpvc
