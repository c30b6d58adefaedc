// gfx950_toep_bf16.h -- the 1025-tap last decoder layer on the bf16 matrix cores at fp32 accuracy.
//
// v_mfma_f32_32x32x16_bf16 runs 16x the MAC rate of the exact-fp32 v_mfma_f32_32x32x2_f32.  Every
// fp32 operand x is split into NPL bf16 terms (planes), x = t0 + t1 (+ t2), 8 mantissa bits each, and a
// product keeps the term pairs (i, j) with i + j < NPL, accumulated in fp32 inside the MFMA:
//   NPL = 3: six products (dropped terms <= 2^-24 relative): fp32-exact, measured 1.3e-6 of max|C| at K = 4112
//            against fp64 (scripts/microbench/bf16x3.hip) -- better than a sequential fp32 fma chain (2.3e-6);
//            6 bf16 MFMAs replace 8 fp32 MFMAs of the same tile => 2.67x the fp32-MFMA peak;
//   NPL = 2: three products, operands carry 16 mantissa bits (relative error <= 2^-17 per operand):
//            ~1e-5 of max|C|, inside the 1e-4 parity bar; 5.3x the fp32-MFMA peak.  The default;
//   NPL = 1: plain bf16 operands with fp32 accumulation (the bf16 training mode, BASELINE config 2).
// Selected per context (vaenpvc_set_precision).
//
// Toeplitz operand.  The weight matrix of this layer is T[k][n] = w[c][k - n + 512] (input-gradient
// direction: k = output bin p, n = input bin i).  A B fragment of the MFMA (lane: column n0+l31,
// eight consecutive k) is therefore eight consecutive taps starting at u0 = k0 + 8*lh - n0 - l31 + 512:
// contiguous, but its start moves by ONE bf16 per lane.  LDS reads need 16-byte alignment, so the
// kernel keeps eight copies of the channel's tap row, copy s shifted by s elements: lane reads the
// 16-byte chunk u0>>3 of copy u0&7 -- one ds_read_b128 per fragment and plane; the copy index
// depends on the lane only (all other address terms are multiples of 8).
#pragma once
#include "gfx950_common.h"

#ifndef VAENPVC_PROF
#define VAENPVC_PROF 0
#endif

namespace vaenpvc {
namespace tuned {

#if VAENPVC_PROF
// developer instrumentation: cycle sums per wave: [0] waves, [1] tap-copy load, [2] A store + barriers,
// [3] k loop, [4] epilogue, [5] total
__device__ unsigned long long g_tb_prof[8];
__device__ unsigned long long g_tw_prof[8];  // wgrad: waves, gload, loadF, mm, lstore, barrier, epilogue, total
#define TBPROF_T(var) const long long var = (long long)__builtin_amdgcn_s_memtime()
#else
#define TBPROF_T(var)
#endif

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
struct __attribute__((packed, aligned(4))) packed4 {
  float x, y, z, w;
};  // 16 bytes = 8 bf16 (native vector: stays in registers)

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// developer switch (variant libraries): waves raise their scheduling priority while they issue a cluster of MFMAs, so that on a SIMD shared by
// two waves the one in its matrix phase is not interleaved with the other's address arithmetic.  Bits: 1 k_gemm_nt, 2 k_cgemm*, 4 k_fconv_r,
// 8 k_fconv.  Measured in round 5: see DESIGN.md section 6.
#ifndef VAENPVC_MFMA_PRIO
#define VAENPVC_MFMA_PRIO 0
#endif
template <int BIT>
__device__ __forceinline__ void mfma_prio(bool hi) {
  if constexpr ((VAENPVC_MFMA_PRIO & BIT) != 0) {
    if (hi) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
  }
}

// fp32 -> bf16, round to nearest even: v_cvt_pk_bf16_f32 converts TWO values in one instruction (the integer recipe
// u += 0x7fff + ((u >> 16) & 1); u >>= 16 costs four; in the conversion-heavy fused kernels the VALU was the limit)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ unsigned bf16_rn(float x) { return cvt_pk_bf16(x, 0.f); }   // as 16 bits (upper half zero)
// bf16 ACTIVATION STORAGE (precision "bf16", round 4): in the bf16 training mode the large tensors of the thin decoder layers
// (pre-LN outputs of layers 1 and 2, the gradients at their activated outputs) live in HBM as bf16 -- half the bytes of
// the passes that are bound by them.  Same workspace regions, rows [C][pitch] with the pitch rounded up to an even number
// of elements so that pairs / quads of neighbouring positions stay 4-byte aligned.  fp32 everywhere else (LayerNorm
// statistics, accumulation, parameters, Adam).
__device__ __forceinline__ float bf16_f32(unsigned short u) { return __uint_as_float((unsigned)u << 16); }
constexpr int act_pitch(bool bf, int h) { return bf ? ((h + 1) & ~1) : h; }
template <bool BF>
__device__ __forceinline__ float act_ld(const float* base, int64_t i) {
  if constexpr (BF) return bf16_f32(reinterpret_cast<const unsigned short*>(base)[i]);
  else return base[i];
}
template <bool BF>
__device__ __forceinline__ void act_st(float* base, int64_t i, float v) {
  if constexpr (BF) reinterpret_cast<unsigned short*>(base)[i] = (unsigned short)bf16_rn(v);
  else base[i] = v;
}
// (x, y) -> per plane one packed pair (x's term in the low half): 3 instructions per element with two planes
template <int NPL>
__device__ __forceinline__ void split_pair(float x, float y, unsigned (&pk)[NPL]) {
  float rx = x, ry = y;
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    pk[p] = cvt_pk_bf16(rx, ry);
    if (p + 1 < NPL) {
      rx -= __uint_as_float(pk[p] << 16);
      ry -= __uint_as_float(pk[p] & 0xffff0000u);
    }
  }
}
// eight consecutive values -> one 16-byte piece per plane
template <int NPL>
__device__ __forceinline__ void pack8(const float (&v)[8], u32x4 (&out)[NPL]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned pk[NPL];
    split_pair<NPL>(v[2 * q], v[2 * q + 1], pk);
#pragma unroll
    for (int p = 0; p < NPL; ++p) out[p][q] = pk[p];
  }
}
// x = t[0] + t[1] + ... (each term the bf16 rounding of what the previous terms left over)
template <int NPL>
__device__ __forceinline__ void split_n(float x, unsigned (&t)[NPL]) {
  float r = x;
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    t[p] = bf16_rn(r);
    if (p + 1 < NPL) r = r - __uint_as_float(t[p] << 16);
  }
}
// term pairs (plane of A, plane of B) kept by a product, smallest first
template <int NPL>
struct Prod;
template <>
struct Prod<3> {
  static constexpr int N = 6;
  static constexpr int A[6] = {2, 1, 0, 1, 0, 0};
  static constexpr int B[6] = {0, 1, 2, 0, 1, 0};
};
template <>
struct Prod<2> {
  static constexpr int N = 3;
  static constexpr int A[3] = {1, 0, 0};
  static constexpr int B[3] = {0, 1, 0};
};
template <>
struct Prod<1> {
  static constexpr int N = 1;
  static constexpr int A[1] = {0};
  static constexpr int B[1] = {0};
};

constexpr int TB_H = 513;             // bins
constexpr int TB_KP = 528;            // bins padded to a multiple of 16 (k-steps of the bf16 MFMA)
constexpr int TB_C = 8;               // channels of the layer's input
constexpr int TB_T = 1025;            // taps
constexpr int TB_CPY = 8;             // shifted copies of a tap row
constexpr int TB_CHUNKS = 133;        // 16-byte chunks per copy (1064 taps; odd => the 8 copies a
                                      // quarter-wave reads from sit in different bank groups)
constexpr int TB_CPYB = TB_CHUNKS * 16;                       // bytes per copy
constexpr int tb_wch(int npl) { return npl * TB_CPY * TB_CPYB; }  // bytes of one channel's copies (npl planes)
constexpr int TB_WFLOATS = TB_C * tb_wch(3) / 4;              // floats reserved for the packed block (all channels)

// ---- ONE pass over d(xh) [F][513] for everything the last layer's backward needs from it besides the GEMMs:
//      (1) its NPL bf16 planes dst[f][plane][528], (2) column i = 512 of the input gradient
//      dY[f][c][512] = sum_p G[f][p] * W[p][c], (3) the bias gradient sum_f sum_p G[f][p] (one atomic per
//      workgroup).  One wave per frame, lane l owns bins 8l .. 8l+7 (lane 0 also bin 512); waves walk the frames
//      with a grid stride.
template <int NPL, bool BOUT = false>   // BOUT: dY is stored as bf16 with rows of 514 (bf16 activation storage)
__global__ void __launch_bounds__(256) k_dxh_post(const float* __restrict__ G, const float* __restrict__ W,
                                                  unsigned short* __restrict__ dst, float* __restrict__ dY,
                                                  float* __restrict__ dbias, int F,
                                                  int dyp = TB_H) {   // floats per row of the fp32 dY (516: rows padded to 16 bytes, round 5)
  __shared__ float sm[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // taps of this lane's 8 bins, all 8 channels: W[(8l + j)*8 + c], 64 contiguous floats
  float wt[8][TB_C];
  {
    const float* wp = W + (size_t)lane * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < TB_C; ++c) wt[j][c] = wp[j * TB_C + c];
  }
  float wl[TB_C];  // bin 512 (lane 0)
#pragma unroll
  for (int c = 0; c < TB_C; ++c) wl[c] = W[512 * TB_C + c];
  float bsum = 0.f;
  for (int f = blockIdx.x * 4 + wv; f < F; f += gridDim.x * 4) {
    const float* gf = G + (int64_t)f * TB_H;
    packed4 p0 = *reinterpret_cast<const packed4*>(gf + 8 * lane);
    packed4 p1 = *reinterpret_cast<const packed4*>(gf + 8 * lane + 4);
    const float g[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    const float gt = lane == 0 ? gf[512] : 0.f;
    unsigned tm[8][NPL];
    float dot[TB_C];
#pragma unroll
    for (int c = 0; c < TB_C; ++c) dot[c] = gt * wl[c];
    float sfr = gt;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      split_n<NPL>(g[j], tm[j]);
      sfr += g[j];
#pragma unroll
      for (int c = 0; c < TB_C; ++c) dot[c] += g[j] * wt[j][c];
    }
    bsum += sfr;
    unsigned short* d = dst + (int64_t)f * (NPL * TB_KP);
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      u32x4 pk;
#pragma unroll
      for (int k = 0; k < 4; ++k) pk[k] = tm[2 * k][p] | (tm[2 * k + 1][p] << 16);
      *reinterpret_cast<u32x4*>(d + p * TB_KP + 8 * lane) = pk;
    }
    if (lane == 0) {  // bin 512 and the zero padding 513..527
      unsigned tt[NPL];
      split_n<NPL>(gt, tt);
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        u32x4 t = z;
        t[0] = tt[p];
        *reinterpret_cast<u32x4*>(d + p * TB_KP + 512) = t;
        *reinterpret_cast<u32x4*>(d + p * TB_KP + 520) = z;
      }
    }
#pragma unroll
    for (int c = 0; c < TB_C; ++c) {
      float v = wave_sum(dot[c]);
      if (lane == 0) act_st<BOUT>(dY, ((int64_t)f * TB_C + c) * (BOUT ? act_pitch(true, TB_H) : dyp) + 512, v);
    }
  }
  bsum = wave_sum(bsum);
  if (lane == 0) sm[wv] = bsum;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dbias, (sm[0] + sm[1]) + (sm[2] + sm[3]));
}

// ---- the loss (util/layers.py:159-167 with log_var = 0, model/vae.py:112-128) AND everything k_dxh_post derives from d(xh), in ONE
//      pass over x and xh: per-frame log-density, d(xh) = (xh - x) / ((1 + 1e-6) F) as fp32 and as NPL bf16 planes, column 512 of the
//      last layer's input gradient, the per-workgroup part of its bias gradient (bpart[blockIdx.x]: the gradient buffer is zeroed
//      when the backward pass starts, which adds the parts then).  Replaces k_nll + k_dxh_post (the second read of d(xh)).
template <int NPL>
__global__ void __launch_bounds__(256) k_nll_dxh_post(const float* __restrict__ x, const float* __restrict__ xh, float* __restrict__ nll_f,
                                                      float* __restrict__ G, const float* __restrict__ W, unsigned short* __restrict__ dst,
                                                      float* __restrict__ dY, float* __restrict__ bpart, int F, float invF,
                                                      const float* __restrict__ y2,      // activated output of the layer in front ([F][8][513]: bin 512 is read)
                                                      float* __restrict__ wpart,         // [gridDim.x][513 * 8]: this workgroup's part of the edge term
                                                      int dyp) {                         // floats per row of dY (513, or 516: rows padded to 16 bytes)
  //  dW[t][c] += sum_f y2[f][c][512] * d(xh)[f][t], t <= 512, of the last layer's weight gradient (k_toep_wgrad_row512 re-read d(xh) for it)
  __shared__ float sm[4];
  __shared__ float wsum[65 * 64];   // [j * 8 + c][lane] (+ row 64: bin 512)
  constexpr float LOG2PI = 1.8378770664093453f;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // taps of this lane's 8 bins, all 8 channels (W[(8l + j)*8 + c]): the same for the four waves -> one LDS copy [j][c / 4][lane][4],
  // read back as 16-byte pieces per frame (in registers they cost a wave per SIMD: 177 registers with the edge-term accumulators)
  __shared__ __attribute__((aligned(16))) float wts[16 * 64 * 4];
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int l = i >> 6, jc = i & 63;          // consecutive threads: consecutive floats of W
    wts[((jc >> 2) * 64 + l) * 4 + (jc & 3)] = W[i];
  }
  __syncthreads();
  float wl[TB_C];  // bin 512 (lane 0)
#pragma unroll
  for (int c = 0; c < TB_C; ++c) wl[c] = W[512 * TB_C + c];
  const float gs = -invF / (1.0f + EPSILON);
  float bsum = 0.f;
  float wacc[8][TB_C], wtl[TB_C];
#pragma unroll
  for (int c = 0; c < TB_C; ++c) {
    wtl[c] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) wacc[j][c] = 0.f;
  }
  for (int f = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wv); f < F; f += gridDim.x * 4) {
    asm volatile("" ::: "memory");   // (keeps the tap reads below inside the loop: hoisted they are 64 registers again)
    float yv[TB_C];
#pragma unroll
    for (int c = 0; c < TB_C; ++c) yv[c] = y2[(int64_t)f * (TB_C * TB_H) + c * TB_H + 512];   // wave-uniform
    const float* xf = x + (int64_t)f * TB_H;
    const float* hf = xh + (int64_t)f * TB_H;
    const packed4 a0 = *reinterpret_cast<const packed4*>(xf + 8 * lane), a1 = *reinterpret_cast<const packed4*>(xf + 8 * lane + 4);
    const packed4 b0 = *reinterpret_cast<const packed4*>(hf + 8 * lane), b1 = *reinterpret_cast<const packed4*>(hf + 8 * lane + 4);
    const float dt = lane == 0 ? xf[512] - hf[512] : 0.f;
    const float d[8] = {a0.x - b0.x, a0.y - b0.y, a0.z - b0.z, a0.w - b0.w, a1.x - b1.x, a1.y - b1.y, a1.z - b1.z, a1.w - b1.w};
    float nl = lane == 0 ? -0.5f * (LOG2PI + (dt * dt) / (1.0f + EPSILON)) : 0.f;
    float g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      nl += -0.5f * (LOG2PI + (d[j] * d[j]) / (1.0f + EPSILON));
      g[j] = d[j] * gs;
    }
    const float gt = dt * gs;
    nl = wave_sum(nl);
    if (G) {   // (uniform; nullptr: only the planes below are read downstream)
      float* gf = G + (int64_t)f * TB_H;
      *reinterpret_cast<packed4*>(gf + 8 * lane) = packed4{g[0], g[1], g[2], g[3]};
      *reinterpret_cast<packed4*>(gf + 8 * lane + 4) = packed4{g[4], g[5], g[6], g[7]};
      if (lane == 0) gf[512] = gt;
    }
    if (lane == 0) nll_f[f] = nl;
    unsigned tm[8][NPL];
    float dot[TB_C];
#pragma unroll
    for (int c = 0; c < TB_C; ++c) dot[c] = gt * wl[c];
    float sfr = gt;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      split_n<NPL>(g[j], tm[j]);
      sfr += g[j];
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(&wts[((j * 2) * 64 + lane) * 4]), w1 = *reinterpret_cast<const f32x4*>(&wts[((j * 2 + 1) * 64 + lane) * 4]);
      const float wtj[TB_C] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
      for (int c = 0; c < TB_C; ++c) {
        dot[c] += g[j] * wtj[c];
        wacc[j][c] += g[j] * yv[c];
      }
    }
#pragma unroll
    for (int c = 0; c < TB_C; ++c) wtl[c] += gt * yv[c];
    bsum += sfr;
    unsigned short* dd = dst + (int64_t)f * (NPL * TB_KP);
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      u32x4 pk;
#pragma unroll
      for (int k = 0; k < 4; ++k) pk[k] = tm[2 * k][p] | (tm[2 * k + 1][p] << 16);
      *reinterpret_cast<u32x4*>(dd + p * TB_KP + 8 * lane) = pk;
    }
    if (lane == 0) {  // bin 512 and the zero padding 513..527
      unsigned tt[NPL];
      split_n<NPL>(gt, tt);
      const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        u32x4 t = z;
        t[0] = tt[p];
        *reinterpret_cast<u32x4*>(dd + p * TB_KP + 512) = t;
        *reinterpret_cast<u32x4*>(dd + p * TB_KP + 520) = z;
      }
    }
#pragma unroll
    for (int c = 0; c < TB_C; ++c) {
      const float v = wave_sum(dot[c]);
      if (lane == 0) dY[((int64_t)f * TB_C + c) * dyp + 512] = v;
    }
  }
  bsum = wave_sum(bsum);
  if (lane == 0) sm[wv] = bsum;
  // the four waves' edge-term partials meet in LDS one after the other (plain read-modify-write, conflict-free rows)
#pragma unroll 1
  for (int w4 = 0; w4 < 4; ++w4) {
    if (wv == w4) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int c = 0; c < TB_C; ++c) {
          float* r = &wsum[(j * TB_C + c) * 64 + lane];
          *r = w4 == 0 ? wacc[j][c] : *r + wacc[j][c];
        }
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < TB_C; ++c) wsum[64 * 64 + c] = w4 == 0 ? wtl[c] : wsum[64 * 64 + c] + wtl[c];
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) bpart[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
  float* wp = wpart + (int64_t)blockIdx.x * (513 * TB_C);
  for (int e = threadIdx.x; e < 513 * TB_C; e += 256) {   // e = t * 8 + c, t = 8 * lane + j
    const int t = e >> 3, c = e & 7;
    wp[e] = t < 512 ? wsum[((t & 7) * TB_C + c) * 64 + (t >> 3)] : wsum[64 * 64 + c];
  }
}
// out[i] += sum over parts (atomic: other kernels add to the same gradient): 64 columns per workgroup, sixteen waves deal the parts
__global__ void __launch_bounds__(1024) k_sum_parts_add(const float* __restrict__ part, int nparts, int n, float* __restrict__ out) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (i < n) {
#pragma unroll 8
    for (int c = wave; c < nparts; c += 16) acc += part[(int64_t)c * n + i];
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w][lane];
    atomicAdd(out + i, t);
  }
}

// ---- shifted bf16 copies of the tap rows: dst[c][plane][s][m] = plane(w[c][m + s]), m < 8*TB_CHUNKS.
//      REV = false: w[c][u] = W[u][c] (input gradient);  REV = true: w[c][u] = W[1024 - u][c] (forward).
template <bool REV, int NPL>
struct PackToepBf16Job {  // job of k_pack_multi (gfx950_elem.h)
  const float* W;
  unsigned short* dst;
  int count;  // TB_C * TB_CPY * 8 * TB_CHUNKS
  __device__ void run(int i) const {
    constexpr int PER = 8 * TB_CHUNKS;
    int c = i / (TB_CPY * PER), r = i - c * (TB_CPY * PER);
    int sh = r / PER, m = r - sh * PER;
    int u = m + sh;
    unsigned t[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) t[p] = 0;
    if (u < TB_T) split_n<NPL>(W[(REV ? TB_T - 1 - u : u) * TB_C + c], t);
    unsigned short* d = dst + (size_t)c * (tb_wch(NPL) / 2) + sh * PER + m;
#pragma unroll
    for (int p = 0; p < NPL; ++p) d[p * TB_CPY * PER] = (unsigned short)t[p];
  }
};

// ---- input gradient:  dY[f][c][i] = sum_p G[f][p] * W[p - i + 512][c]   (i < 512; column 512: k_dxh_post)
//
// Workgroup = 64 frames x all 512 bins of every channel (channels in sequence); 4 waves, wave w owns
// bins [128w, 128w+128) = 4 column tiles, both row tiles => 8 accumulators.  The reduction index p
// runs in 3 chunks of 176 bins (11 k-steps): the chunk of the G planes is staged through registers into
// LDS (prefetched one chunk ahead); the channel's tap copies are loaded at every channel switch.
constexpr int DG_M = 64;                               // frames per workgroup
constexpr int DG_KC = 176, DG_NKC = TB_KP / DG_KC;     // bins per chunk, chunks
constexpr int DG_ROWB = DG_KC * 2 + 16;                // bytes per LDS row (368 = 16 * 23, odd)
constexpr int DG_APL = DG_M * DG_ROWB;                 // bytes per plane of the A tile
constexpr int dg_lds(int npl) { return npl * DG_APL + tb_wch(npl); }  // NPL = 3: 70 656 + 51 072 bytes

//
// FWD = true is the forward direction of the same layer with the same machinery:
//     xh[f][p] = bias + sum_c sum_i y[f][c][i] * W[p - i + 512][c]        (p < 512; column 512: the plane producer)
// k = input bin i, column = output bin p, taps read from the REVERSED copies (u = 512 - p + i); the A
// planes are per channel ([F][3][8][528]) and the accumulators run over all 8 channels.
// (second launch bound = waves per SIMD.  One plane: two workgroups per CU, -10..13 %; two planes need 344 - 388
//  registers and spill at 256: measured equal or slower, so they keep one workgroup per CU)
// DYP (input gradient only, round 5): floats per row of the fp32 result.  The canonical [F][8][513] rows start at every 4-byte alignment, and
// 16-byte stores that are not 16-byte aligned retire at about HALF the rate (this kernel with one plane: 538 MB in 230 us = 2.3 TB/s, the
// merge GEMM's 1 539-float rows the same); with rows of 516 floats every store of this epilogue is aligned.  Only the fused backward
// kernel of decoder layer 2 reads the result (gfx950_fbwd.h, same pitch).
// LNA (round 6, forward only): the A operand is NOT read as planes.  The kernel stages decoder layer 2's fp32 pre-LN output itself and applies
// LayerNorm + lrelu (statistics from the layer's own forward kernel, k_fconv<..., OST>) and the split into bf16 terms on the way into LDS --
// what the separate pass k_ln_stats_act_planes did with a read of the 0.54 GB tensor and a write of 0.55 GB of planes (237 us per step).  Every
// A element is staged exactly once per launch (a workgroup owns 64 frames x all 512 columns), so nothing is converted twice.  On the way:
//   * the converted pieces also leave as the operand planes yp[F][NPL][8][528] the weight-gradient kernel reads by LDS-DMA (it cannot convert
//     on load); nullable: the conversion path has no backward pass;
//   * output column 512 (a dot product of the activated frame with the reversed taps, the pass computed it) is accumulated per staged piece
//     against an fp32 LDS copy of the taps and summed per frame in a FIXED order at the end (bitwise repeatable);
//   * bin 512 of the activated tensor leaves as fp32 (dec_y[f][c][512]: the loss kernel's edge term reads it).
// Thread map: piece id = tid + 256 i (i < 6), piece = (frame row id / 22, 8 bins id % 22) of the 64 x 176 chunk: consecutive lanes read
// consecutive 32-byte runs of an fp32 row and write consecutive 16-byte pieces of a plane row.
#ifndef VAENPVC_LNA_NT
#define VAENPVC_LNA_NT 1   // the planes yp leave with non-temporal stores: 0.55 GB that the weight-gradient kernel reads a whole forward tail + loss later;
                           // cached they push x / xh out of the Infinity Cache in front of the loss kernel (measured: loss 88 -> 111 us)
#endif
struct ToepLna {
  const float* a = nullptr;      // [F][8][513] fp32 pre-LN output of decoder layer 2
  const float* st = nullptr;     // [F][2] its LayerNorm statistics (mean, rstd)
  const float* gamma = nullptr;  // [8]
  const float* beta = nullptr;
  const float* wc = nullptr;     // [8][1040] fp32 taps, wc[c][8 + t] = W[t][c] (k_ln_stats_act_planes' table)
  unsigned short* yp = nullptr;  // out, nullable: planes of the activated tensor [F][NPL][8][528] (zero padding 513 .. 527)
  float* decy = nullptr;         // out, nullable: fp32 [F][8][513], only bin 512 of every channel is written
};
constexpr int DG_LNA_LDS = TB_C * TB_KP * 4 + 2 * TB_C * 4 + DG_M * (DG_KC / 8) * 4 + DG_M * 4;   // taps + LN parameters + per-piece dot sums + row sums
template <bool FWD, int NPL, bool BOUT = false, int DYP = TB_H, bool LNA = false>   // BOUT (input gradient only): dY stored as bf16, rows of 514
__global__ void __launch_bounds__(256, (NPL == 1 && !LNA ? 2 : 1)) k_toep_gemm_bf16(const unsigned short* __restrict__ gp,   // A planes
                                                            const unsigned short* __restrict__ wcp,  // packed tap copies
                                                            const float* __restrict__ bias,          // FWD: [1]
                                                            float* __restrict__ dY,  // dgrad: [F][8][513]; fwd: [F][513]
                                                            int F, ToepLna ln) {
  static_assert(!LNA || (FWD && !BOUT && DYP == TB_H), "LayerNorm on load: the forward direction");
  constexpr int A_PL = (FWD ? TB_C : 1) * TB_KP * 2;  // bytes between planes of a frame
  constexpr int A_FR = NPL * A_PL;                    // bytes per frame
  constexpr int TB_WCH = tb_wch(NPL);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // LNA: TWO A tiles -- the chunk being multiplied and the one being converted into (no staging registers for the converted terms, one
  // barrier per chunk)
  constexpr int NABUF = LNA ? 2 : 1;
  unsigned char* sA = smem;             // the tile the MFMAs read (LNA: toggles between the two)
  unsigned char* sAn = smem + (LNA ? NPL * DG_APL : 0);   // LNA: the tile the conversion writes
  unsigned char* sW = smem + NABUF * NPL * DG_APL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int f0 = blockIdx.x * DG_M;
  // ---- LNA staging state
  constexpr int LN_NPC = 6, LN_PPR = DG_KC / 8, LN_NP = DG_M * LN_PPR;   // pieces per thread / row / chunk (22, 1408)
  float* wrev = reinterpret_cast<float*>(smem + NABUF * NPL * DG_APL + TB_WCH);   // [8][528]: wrev[c][i] = W[1024 - i][c], zero from bin 513 on
  float* lnpar = wrev + TB_C * TB_KP;                                     // [2][8] gamma | beta
  float* pdot = lnpar + 2 * TB_C;                                         // [1408] per-piece dot sums (end of the kernel)
  float* rdot = pdot + LN_NP;                                             // [64]
  int ln_aoff[LN_NPC], ln_loff[LN_NPC], ln_yoff[LN_NPC], ln_pc[LN_NPC];
  bool ln_ok[LN_NPC], ln_st[LN_NPC];
  float ln_mean[LN_NPC], ln_rstd[LN_NPC], ln_dot[LN_NPC];
  float ln_A[LN_NPC], ln_B[LN_NPC];   // n = v * A + B for the chunk being converted (A = rstd * gamma_c, B = beta_c - mean * A)
  float raw[LN_NPC][8];
  // the last chunk's pieces 20 (bin 512 + padding) and 21 (padding) of a row are written by thread `row` (tid < 64), not by the piece slots
  float sp_mean = 0.f, sp_rstd = 0.f, sp_a512 = 0.f, sp_dot = 0.f;
  bool ln_spec[LN_NPC];
  if constexpr (LNA) {
#pragma unroll
    for (int i = 0; i < LN_NPC; ++i) {
      const int id0 = tid + 256 * i, id = id0 < LN_NP ? id0 : LN_NP - 1, row = id / LN_PPR, pc = id - row * LN_PPR;
      const int fr = f0 + row < F ? row : F - 1 - f0;     // rows past the batch end: duplicates of its last frame, never stored
      ln_pc[i] = pc;
      ln_aoff[i] = fr * (TB_C * TB_H) + pc * 8;
      ln_loff[i] = row * DG_ROWB + pc * 16;
      ln_yoff[i] = fr * (NPL * TB_C * TB_KP) + pc * 8;
      ln_st[i] = id0 < LN_NP;                              // this slot holds a piece of the chunk (i = 5: the first 128 threads)
      ln_ok[i] = id0 < LN_NP && f0 + row < F;              // ... of a frame of the batch
      ln_mean[i] = ln.st[2 * (f0 + fr)];
      ln_rstd[i] = ln.st[2 * (f0 + fr) + 1];
      ln_dot[i] = 0.f;
      ln_spec[i] = pc >= 20;
    }
    if (tid < DG_M) {
      const int fr = f0 + tid < F ? tid : F - 1 - f0;
      sp_mean = ln.st[2 * (f0 + fr)];
      sp_rstd = ln.st[2 * (f0 + fr) + 1];
    }
    for (int i = tid; i < TB_C * TB_KP; i += 256) {
      const int c = i / TB_KP, b = i - c * TB_KP;
      wrev[i] = b < TB_H ? ln.wc[c * 1040 + 8 + (TB_T - 1) - b] : 0.f;
    }
    if (tid < 2 * TB_C) lnpar[tid] = tid < TB_C ? ln.gamma[tid] : ln.beta[tid - TB_C];
    // (visible behind the first barrier of the channel loop)
  }
  // fp32 pieces of chunk (c, kc) -> raw.  The last chunk's pieces 20 (bins 512 .. 519: only bin 512 exists) and 21 (padding) read bins 505 .. 512
  // instead: no access leaves the frame's row set; the conversion picks bin 512 out of element 7
  auto gload_lna = [&](int c, int kc, int i0 = 0, int i1 = LN_NPC) __attribute__((always_inline)) {   // piece slots [i0, i1)
    if constexpr (LNA) {
      const float* ab = ln.a + (int64_t)f0 * (TB_C * TB_H) + c * TB_H + kc * DG_KC;
#pragma unroll
      for (int i = i0; i < i1; ++i) {
        const int back = (kc == DG_NKC - 1 && ln_pc[i] >= 20) ? (ln_pc[i] == 20 ? 7 : 15) : 0;
        const float* p = ab + ln_aoff[i] - back;
        const packed4 p0 = *reinterpret_cast<const packed4*>(p), p1 = *reinterpret_cast<const packed4*>(p + 4);
        raw[i][0] = p0.x; raw[i][1] = p0.y; raw[i][2] = p0.z; raw[i][3] = p0.w;
        raw[i][4] = p1.x; raw[i][5] = p1.y; raw[i][6] = p1.z; raw[i][7] = p1.w;
      }
      if (i1 == LN_NPC && kc == DG_NKC - 1 && tid < DG_M) {   // (with the second half of a channel's last chunk: one chunk before convert_special uses it)
        const int fr = f0 + tid < F ? tid : F - 1 - f0;
        sp_a512 = ln.a[((int64_t)(f0 + fr) * TB_C + c) * TB_H + (TB_H - 1)];
      }
    }
  };
  // raw (chunk (c, kc)) -> LayerNorm + lrelu -> bf16 terms (lpk); dot products with the reversed taps; yp / bin 512 out.  PAIRWISE: pair q of
  // piece i is ~25 vector instructions, small enough to issue in the shadow of the MFMAs of ONE (k-step, column tile) step of the chunk in
  // front -- as one serial block at the top of a chunk the conversion cost 100 us per launch (measured, 328 -> 430 us: a wave issues in order
  // and one wave runs per SIMD).  `live` = false: the chunk is the redundant wrap-around prefetch behind the last chunk of the launch.
  u32x4 cpk[NPL];   // the piece being converted (four pairs = four steps), then stored
  // per chunk, before its first pair: the LayerNorm coefficients of the six piece slots
  auto convert_begin = [&](float g, float b) __attribute__((always_inline)) {
    if constexpr (LNA) {
#pragma unroll
      for (int i = 0; i < LN_NPC; ++i) {
        ln_A[i] = ln_rstd[i] * g;
        ln_B[i] = b - ln_mean[i] * ln_A[i];
      }
    }
  };
  // pair q of piece slot i of chunk (c, kc): LayerNorm + lrelu (n = v A + B), the pair's term of output column 512, the split; behind the
  // fourth pair the piece goes to the OTHER A tile in LDS and (live) to yp.  The special pieces of a channel's last chunk (20: bin 512 +
  // padding, 21: padding) convert what they loaded (bins 505 .. 512) and are neither stored nor summed: convert_special writes them.
  auto convert_pair = [&](int i, int q, int c, int kc, bool live) __attribute__((always_inline)) {
    if constexpr (LNA) {
      const float n0 = fmaf(raw[i][2 * q], ln_A[i], ln_B[i]), n1 = fmaf(raw[i][2 * q + 1], ln_A[i], ln_B[i]);
      const float o0 = fmaxf(n0, LEAK * n0), o1 = fmaxf(n1, LEAK * n1);
      const bool reg = !(kc == DG_NKC - 1 && ln_spec[i]);
      const float* wr = wrev + c * TB_KP + kc * DG_KC + ln_pc[i] * 8 + 2 * q;
      const float d = o0 * wr[0] + o1 * wr[1];
      ln_dot[i] += (live && reg) ? d : 0.f;
      unsigned pk2[NPL];
      split_pair<NPL>(o0, o1, pk2);
#pragma unroll
      for (int p = 0; p < NPL; ++p) cpk[p][q] = pk2[p];
      if (q == 3) {
        if (ln_st[i] && reg) {
#pragma unroll
          for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(sAn + p * DG_APL + ln_loff[i]) = cpk[p];
        }
        if (live && reg && ln.yp && ln_ok[i]) {
          unsigned short* ypb = ln.yp + (int64_t)f0 * (NPL * TB_C * TB_KP) + c * TB_KP + kc * DG_KC;
#pragma unroll
          for (int p = 0; p < NPL; ++p) st_nt<VAENPVC_LNA_NT>(reinterpret_cast<u32x4*>(ypb + p * (TB_C * TB_KP) + ln_yoff[i]), cpk[p]);
        }
      }
    }
  };
  // the last chunk of channel c: thread `row` (tid < 64) writes the row's pieces 20 and 21 -- bin 512 of the activated tensor (also to yp /
  // dec_y, and its term of output column 512) and the zero padding 513 .. 527
  auto convert_special = [&](int c, bool live) __attribute__((always_inline)) {
    if constexpr (LNA) {
      if (tid < DG_M) {
        const float y = lnact_v(sp_a512, sp_mean, sp_rstd, lnpar[c], lnpar[TB_C + c]);
        sp_dot += live ? y * wrev[c * TB_KP + (TB_H - 1)] : 0.f;
        unsigned t[NPL];
        split_n<NPL>(y, t);
        const u32x4 z = {0u, 0u, 0u, 0u};
        const bool out = live && f0 + tid < F;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          u32x4 v = z;
          v[0] = t[p];
          *reinterpret_cast<u32x4*>(sAn + p * DG_APL + tid * DG_ROWB + 20 * 16) = v;
          *reinterpret_cast<u32x4*>(sAn + p * DG_APL + tid * DG_ROWB + 21 * 16) = z;
          if (out && ln.yp) {
            unsigned short* yr = ln.yp + ((int64_t)(f0 + tid) * NPL + p) * (TB_C * TB_KP) + c * TB_KP + (TB_H - 1);
            *reinterpret_cast<u32x4*>(yr) = v;
            *reinterpret_cast<u32x4*>(yr + 8) = z;
          }
        }
        if (out && ln.decy) ln.decy[((int64_t)(f0 + tid) * TB_C + c) * TB_H + (TB_H - 1)] = y;
      }
    }
  };
  // Conversion schedule (round 6, last version): the 24 pairs of the NEXT chunk are spread over ALL 44 (k-step, column tile) steps of the chunk being
  // multiplied -- pair p at step floor(44 p / 24) -- so a step carries ~10 vector instructions beside its 6 MFMAs instead of ~18 in the last 24
  // steps (a wave hides ~5 other instructions per MFMA; the fragment reads already use 1 - 2).  The fp32 pieces arrive in two halves on ONE set
  // of staging registers: piece slots 0 - 2 (converted in steps 0 - 20) are refilled at step 22 with the chunk AFTER next, slots 3 - 5 (converted
  // in steps 22 - 42) at step 0 with the next chunk: every load has half a chunk (~2 us) to land.
  constexpr int LN_NP2 = 4 * LN_NPC;   // pairs per chunk
  auto ln_pair_at = [](int step) constexpr { for (int p = 0; p < 24; ++p) if ((44 * p) / 24 == step) return p; return -1; };
  static_assert(LN_NP2 == 24, "schedule written for six piece slots");

  // staging map: thread -> (row = tid >> 2, part = tid & 3) copies the 16-byte pieces part + 4*q
  // (q < 6; a row of a chunk has 22 pieces) of all three planes: every address is a per-thread
  // base plus a compile-time offset
  constexpr int PPR = DG_KC * 2 / 16;  // 22
  const int srow = tid >> 2, spart = tid & 3;
  const int sfr = f0 + srow < F ? f0 + srow : F - 1;  // clamp: rows past the batch end are never stored
  const unsigned char* gsrc = reinterpret_cast<const unsigned char*>(gp) + (size_t)sfr * A_FR + spart * 16;
  unsigned char* sdst = sA + srow * DG_ROWB + spart * 16;
  u32x4 st[6 * NPL];
  auto gload = [&](int c, int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        int c16 = spart + 4 * q;
        c16 = c16 < PPR ? c16 : PPR - 1;  // (q = 5, parts 2 and 3: duplicate load, not stored)
        st[pl * 6 + q] = *reinterpret_cast<const u32x4*>(gsrc + pl * A_PL + (FWD ? c * (TB_KP * 2) : 0) + kc * (DG_KC * 2) + (c16 - spart) * 16);
      }
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int q = 0; q < 6; ++q)
        if (spart + 4 * q < PPR) *reinterpret_cast<u32x4*>(sdst + pl * DG_APL + q * 64) = st[pl * 6 + q];
  };

  // fragment addresses
  //   A: plane*DG_APL + (32*mb + l31)*DG_ROWB + (16*ks + 8*lh)*2
  //   B: plane*(8*TB_CPYB) + s*TB_CPYB + 16*q,  u0 = 176*kc + 16*ks + 8*lh - bin + 512
  const int aoff = l31 * DG_ROWB + lh * 16;
  //      column tile nb of wave w = bins 128*w + 4*l31 + nb (lane-interleaved, so that a lane ends up
  //      with 4 CONSECUTIVE bins of a row: one 16-byte store instead of four 4-byte ones)
  int boff[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int u_lane = 8 * lh - 128 * wave - 4 * l31 - nb + 512;  // + 176*kc + 16*ks
    const int s_cpy = u_lane & 7;
    boff[nb] = s_cpy * TB_CPYB + ((u_lane - s_cpy) >> 3) * 16;  // + (22*kc + 2*ks)*16
  }
  // fragments: A of a whole k-step (2 row tiles x 3 planes), B of ONE column tile (3 planes); both
  // ping-pong: while the 12 MFMAs of column tile nb run, the B fragments of tile nb+1 (or, for the
  // last tile, A and B(0) of the next k-step) are already being read from LDS
  u32x4 fa[2][2][NPL], fb[2][NPL];
  auto loadA = [&](int set, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        fa[set][mb][pl] = *reinterpret_cast<const u32x4*>(sA + pl * DG_APL + mb * 32 * DG_ROWB + aoff + ks * 32);
  };
  auto loadB = [&](int set, int kc, int ks, int nb) __attribute__((always_inline)) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
      fb[set][pl] = *reinterpret_cast<const u32x4*>(sW + pl * (TB_CPY * TB_CPYB) + boff[nb] + (22 * kc + 2 * ks) * 16);
  };
  f32x16 acc[2][4];
  auto mm = [&](int sa, int sb, int nb) __attribute__((always_inline)) {
    // term pairs, smallest first
    using PR = Prod<NPL>;
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) acc[mb][nb] = mfma_bf16(fa[sa][mb][PR::A[t]], fb[sb][PR::B[t]], acc[mb][nb]);
  };

#if VAENPVC_PROF
  long long pc[4] = {0, 0, 0, 0};
  TBPROF_T(k0);
#endif
  // channel range of this workgroup: gridDim.y groups share the 8 channels (small batches: more workgroups; the
  // forward direction then accumulates its partial sums over channel groups with atomics into a zeroed output)
  const int cpg = TB_C / (int)gridDim.y, c_lo = (int)blockIdx.y * cpg, c_hi = c_lo + cpg;
  const bool split_out = FWD && gridDim.y > 1;
  // Input gradient: the channels are independent outputs, so every workgroup starts at a different one (rotation by
  // its index): neighbouring workgroups then write different channel rows at the same time.  (Forward: the channels
  // accumulate, order kept.)  Spreading the stores of a channel over the next channel's k-steps (a second accumulator
  // set) was tried and dropped: 128 more live registers and the unrolled chunk loop spill ~100 registers, 340 -> 394 us.
  const int crot = FWD ? 0 : (int)(blockIdx.x % (unsigned)cpg);
  auto chan = [&](int ci) { return c_lo + (ci - c_lo + crot) % cpg; };   // ci = loop position -> channel
  if constexpr (LNA) {
    gload_lna(chan(c_lo), 0);
    __syncthreads();   // wrev / lnpar are in LDS
    convert_begin(lnpar[chan(c_lo)], lnpar[TB_C + chan(c_lo)]);
#pragma unroll
    for (int i = 0; i < LN_NPC; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) convert_pair(i, q, chan(c_lo), 0, true);   // the first chunk: once per workgroup, serial
    unsigned char* t_ = sA; sA = sAn; sAn = t_;   // (visible behind the barrier at the top of the channel loop + the one behind the tap copies)
    gload_lna(chan(c_lo), 1 < DG_NKC ? 1 : 0, 0, LN_NPC / 2);   // first half of the second chunk (steady state: requested at step 22 of the chunk in front)
  } else gload(chan(c_lo), 0);
  // tap copies of a channel: the global loads are issued one channel ahead, BEFORE the epilogue stores of the channel
  // in front (vector memory completes in order and loads and stores share one counter on this ISA: loads issued behind
  // 64 stores per lane wait for all of them -- measured: the input-gradient epilogue, 538 MB, cost its full 100 us on
  // top of the MFMA time); the LDS stores follow after the barrier at the top of the channel
  constexpr int NW16 = TB_WCH / 16, WPT = (NW16 + 255) / 256;  // NPL = 3: 3192 pieces, 13 per thread
  u32x4 wv[WPT];
  auto wload = [&](int c) __attribute__((always_inline)) {
    const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(wcp) + (size_t)c * TB_WCH);
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      int i = tid + 256 * k;
      wv[k] = src[i < NW16 ? i : NW16 - 1];
    }
  };
  if constexpr (!FWD) wload(chan(c_lo));
#pragma unroll 1
  for (int ci = c_lo; ci < c_hi; ++ci) {
    const int c = chan(ci);
    TBPROF_T(t0);
    __syncthreads();  // previous channel fully consumed (tap copies and A tile)
    if constexpr (FWD) wload(c);   // (forward: no epilogue between the channels; loading here keeps the registers short-lived)
#pragma unroll
    for (int k = 0; k < WPT; ++k) {
      int i = tid + 256 * k;
      if (i < NW16) reinterpret_cast<u32x4*>(sW)[i] = wv[k];
    }
    if (!FWD || c == c_lo) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = zero16();
    }
    if constexpr (LNA) __syncthreads();   // the channel's tap copies are in LDS (without LNA the barrier behind the first lstore publishes them)
    TBPROF_T(t1);
#if VAENPVC_PROF
    pc[0] += t1 - t0;
#endif
    for (int kc = 0; kc < DG_NKC; ++kc) {
      TBPROF_T(t2);
      if constexpr (!LNA) {
        lstore();  // chunk kc (prefetched)
        __syncthreads();
      }
      TBPROF_T(t3);
      // prefetch the next chunk (same rows, next bins; wraps to chunk 0 for the next channel)
      const int kn = kc + 1 < DG_NKC ? kc + 1 : 0;
      const int cn = kc + 1 < DG_NKC ? c : (ci + 1 < c_hi ? chan(ci + 1) : c);
      const bool nlive = kc + 1 < DG_NKC || ci + 1 < c_hi;   // (false: the wrap-around prefetch behind the last chunk)
      // ... and the chunk after next (LNA: its first half is requested in the middle of this chunk)
      const int kn2 = kn + 1 < DG_NKC ? kn + 1 : 0;
      const int cn2 = kn + 1 < DG_NKC ? cn : ((kc + 1 < DG_NKC ? ci : ci + 1) + 1 < c_hi ? chan((kc + 1 < DG_NKC ? ci : ci + 1) + 1) : cn);
      if constexpr (LNA) {
        gload_lna(cn, kn, LN_NPC / 2, LN_NPC);      // second half of the next chunk
        convert_begin(lnpar[cn], lnpar[TB_C + cn]);
      } else gload(cn, kn);
      __builtin_amdgcn_sched_barrier(0);
      loadA(0, 0);
      loadB(0, kc, 0, 0);
#pragma unroll
      for (int ks = 0; ks < 11; ++ks) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int sa = ks & 1, sb = (4 * ks + nb) & 1;
          if (nb < 3) {
            loadB(sb ^ 1, kc, ks, nb + 1);
          } else if (ks + 1 < 11) {
            loadA(sa ^ 1, ks + 1);
            loadB(sb ^ 1, kc, ks + 1, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          mm(sa, sb, nb);
          if constexpr (LNA) {   // one conversion pair of the NEXT chunk in the shadow of this step's MFMAs
            constexpr int NMM = 2 * Prod<NPL>::N;
            const int pr = ln_pair_at(4 * ks + nb);
            if (4 * ks + nb == 22) gload_lna(cn2, kn2, 0, LN_NPC / 2);   // slots 0 - 2 are converted: first half of the chunk after next
            if (pr >= 0) {
              convert_pair(pr >> 2, pr & 3, cn, kn, nlive);
#pragma unroll
              for (int _m = 0; _m < NMM; ++_m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (NMM >= 12 ? 2 : NMM >= 6 ? 4 : 12), 0);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#if VAENPVC_PROF
      asm volatile("s_nop 0" ::: "memory");
#endif
      TBPROF_T(t4);
      if constexpr (LNA) {
        if (kn == DG_NKC - 1) convert_special(cn, nlive);   // uniform
      }
      __syncthreads();  // chunk consumed (LNA: and the next one converted)
      if constexpr (LNA) { unsigned char* t_ = sA; sA = sAn; sAn = t_; }
#if VAENPVC_PROF
      TBPROF_T(t5);
      pc[1] += (t3 - t2) + (t5 - t4);
      pc[2] += t4 - t3;
#endif
    }
    TBPROF_T(t6);
    if (!FWD && ci + 1 < c_hi) wload(chan(ci + 1));
    if (FWD && ci + 1 < c_hi) continue;
    // epilogue: rows = frames, lanes = 32 consecutive bins -> 128-byte stores; one uniform base
    // per workgroup and channel, 32-bit lane offsets, column tiles through the immediate offset
    {
      constexpr int OP = FWD ? TB_H : DYP;          // floats per output row
      constexpr int ORS = (FWD ? 1 : TB_C) * OP;    // floats between consecutive frames of the output
      float* ob = dY + ((int64_t)f0 * (FWD ? 1 : TB_C) + (FWD ? 0 : c)) * OP;
      const int lo = (4 * lh) * ORS + 128 * wave + 4 * l31;
      const bool full = f0 + DG_M <= F;  // uniform
      const float bb = (FWD && c_lo == 0) ? bias[0] : 0.f;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int r = mb * 32 + (reg & 3) + 8 * (reg >> 2);  // + 4*lh
          float* o = ob + lo + r * ORS;
          if (full || f0 + r + 4 * lh < F) {
            if (split_out) {  // uniform: partial sum of this channel group (the plane producer zeroed the row)
#pragma unroll
              for (int nb = 0; nb < 4; ++nb) atomicAdd(o + nb, acc[mb][nb][reg] + bb);
            } else if constexpr (BOUT) {  // four bf16: one 8-byte store (rows of 514 elements: 4-byte aligned)
              static_assert(!BOUT || !FWD, "bf16 output: input gradient only");
              unsigned short* o2 = reinterpret_cast<unsigned short*>(dY) + ((int64_t)f0 * TB_C + c) * act_pitch(true, TB_H) +
                                   (4 * lh + r) * (TB_C * act_pitch(true, TB_H)) + 128 * wave + 4 * l31;
              const u32x2_a4 pk = {cvt_pk_bf16(acc[mb][0][reg], acc[mb][1][reg]), cvt_pk_bf16(acc[mb][2][reg], acc[mb][3][reg])};
              st_nt<VAENPVC_NT_T>(reinterpret_cast<u32x2_a4*>(o2), pk);
            } else {  // rows are only 4-byte aligned (513 bins): packed 16-byte store
              if constexpr (VAENPVC_NT_T && !FWD)
                st_nt<1>(reinterpret_cast<f32x4_a4*>(o), f32x4_a4{acc[mb][0][reg] + bb, acc[mb][1][reg] + bb, acc[mb][2][reg] + bb, acc[mb][3][reg] + bb});
              else
              *reinterpret_cast<packed4*>(o) = packed4{acc[mb][0][reg] + bb, acc[mb][1][reg] + bb, acc[mb][2][reg] + bb, acc[mb][3][reg] + bb};
            }
          }
        }
    }
#if VAENPVC_PROF
    asm volatile("s_nop 0" ::: "memory");
    TBPROF_T(t7);
    pc[3] += t7 - t6;
#endif
  }
  if constexpr (LNA) {
    // output column 512: the pieces' dot sums of a frame row added in a fixed order (22 pieces x 24 chunks each)
#pragma unroll
    for (int i = 0; i < LN_NPC; ++i)
      if (ln_st[i]) pdot[tid + 256 * i] = ln_dot[i];
    __syncthreads();
    if (tid < DG_M && f0 + tid < F) {
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < LN_PPR; ++k) sum += pdot[tid * LN_PPR + k];
      dY[(int64_t)(f0 + tid) * TB_H + (TB_H - 1)] = (sum + sp_dot) + bias[0];
    }
  }
#if VAENPVC_PROF
  {
    TBPROF_T(k1);
    if (!FWD && (threadIdx.x & 63) == 0) {   // (the input gradient only: the forward instances would add into the same sums)
      atomicAdd(g_tb_prof + 0, 1ull);
      for (int i = 0; i < 4; ++i) atomicAdd(g_tb_prof + 1 + i, (unsigned long long)pc[i]);
      atomicAdd(g_tb_prof + 5, (unsigned long long)(k1 - k0));
    }
  }
#endif
}

// ---- producer of the forward operand: LayerNorm statistics of the layer in front (8 x 513 per frame),
//      its activated output as fp32 (still read by the weight-gradient GEMM) AND as three bf16 planes
//      yp[f][plane][c][528], plus column p = 512 of the forward result, which is a plain dot product
//      of the activated frame with the taps W[1024 - i][c] (the GEMM kernel covers p < 512).
//      One wave per frame; lane l owns bins 8l .. 8l+7 of every channel, lanes 0..7 also bin 512 of
//      channel l.
template <int NPL, bool BIN = false>
__global__ void __launch_bounds__(256) k_ln_stats_act_planes(const float* __restrict__ a, float* __restrict__ st,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ y, unsigned short* __restrict__ yp,
                                                             const float* __restrict__ Wc,  // [8][WROW]: Wc[c][8 + t]
                                                             const float* __restrict__ bias, float* __restrict__ xh, int F,
                                                             int write_y,    // 0: only bin 512 of y (fp32) is stored
                                                             int zero_xh) {  // 1: bins 0..511 of xh are zeroed (the forward
                                                                             // GEMM accumulates channel groups into them)
  constexpr int WROWC = 1040;
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= F) return;
  constexpr int PIN = act_pitch(BIN, TB_H);   // row pitch of the input (bf16 storage: 514 elements of 2 bytes)
  const float* af = a + (int64_t)f * (TB_C * TB_H);
  const unsigned short* ah = reinterpret_cast<const unsigned short*>(a) + (int64_t)f * (TB_C * PIN);
  float v[TB_C][8], vt = 0.f;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < TB_C; ++c) {
    if constexpr (BIN) {   // 8 bf16 = one 16-byte piece (rows are 4-byte aligned: the pitch is even)
      const u32x4 q = *reinterpret_cast<const u32x4_a4*>(ah + c * PIN + 8 * lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[c][2 * j] = __uint_as_float(q[j] << 16);
        v[c][2 * j + 1] = __uint_as_float(q[j] & 0xffff0000u);
      }
    } else {
      packed4 p0 = *reinterpret_cast<const packed4*>(af + c * TB_H + 8 * lane);
      packed4 p1 = *reinterpret_cast<const packed4*>(af + c * TB_H + 8 * lane + 4);
      v[c][0] = p0.x; v[c][1] = p0.y; v[c][2] = p0.z; v[c][3] = p0.w;
      v[c][4] = p1.x; v[c][5] = p1.y; v[c][6] = p1.z; v[c][7] = p1.w;
    }
    s += ((v[c][0] + v[c][1]) + (v[c][2] + v[c][3])) + ((v[c][4] + v[c][5]) + (v[c][6] + v[c][7]));
  }
  if (lane < TB_C) {
    vt = BIN ? bf16_f32(ah[lane * PIN + 512]) : af[lane * TB_H + 512];
    s += vt;
  }
  const float mean = wave_sum(s) / (TB_C * TB_H);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < TB_C; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float d = v[c][j] - mean;
      q += d * d;
    }
  if (lane < TB_C) q += (vt - mean) * (vt - mean);
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (TB_C * TB_H) + LN_EPS);
  if (lane == 0) {
    st[2 * f] = mean;
    st[2 * f + 1] = rstd;
  }
  float* yf = y + (int64_t)f * (TB_C * TB_H);
  unsigned short* ypf = yp + (int64_t)f * (NPL * TB_C * TB_KP);
  float dot = 0.f;
  // taps W[1024 - i][c] for i = 8*lane + j  =  Wc[c][8 + 1024 - 8*lane - j]: one ascending block per channel, reversed.  ALL
  // channels' taps are fetched before the first plane store (a load between the stores would order the later stores behind
  // its round trip: round 4)
  packed4 wq[TB_C][2];
#pragma unroll
  for (int c = 0; c < TB_C; ++c) {
    const float* wp = Wc + c * WROWC + 8 + 1017 - 8 * lane;
    wq[c][0] = *reinterpret_cast<const packed4*>(wp);
    wq[c][1] = *reinterpret_cast<const packed4*>(wp + 4);
  }
  float wlast = 0.f;
  if (lane < TB_C) wlast = Wc[lane * WROWC + 8 + 512];
#pragma unroll
  for (int c = 0; c < TB_C; ++c) {
    const float g = gamma[c], b = beta[c];
    const packed4 w0 = wq[c][0], w1 = wq[c][1];
    const float wr[8] = {w1.w, w1.z, w1.y, w1.x, w0.w, w0.z, w0.y, w0.x};
    float o[8];
    unsigned tm[8][NPL];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o[j] = lnact_v(v[c][j], mean, rstd, g, b);
      dot += o[j] * wr[j];
    }
    if (write_y) {  // uniform; the fp32 copy is only read by the exact-fp32 weight-gradient kernel
      *reinterpret_cast<packed4*>(yf + c * TB_H + 8 * lane) = packed4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<packed4*>(yf + c * TB_H + 8 * lane + 4) = packed4{o[4], o[5], o[6], o[7]};
    }
    u32x4 pkv[NPL];
    pack8<NPL>(o, pkv);
#pragma unroll
    for (int p = 0; p < NPL; ++p) st_nt<VAENPVC_NT_B>(reinterpret_cast<u32x4*>(ypf + (p * TB_C + c) * TB_KP + 8 * lane), pkv[p]);
  }
  if (lane < TB_C) {  // bin 512 of channel `lane`, and the zero padding 513..527 of its plane rows
    const int c = lane;
    float o = lnact_v(vt, mean, rstd, gamma[c], beta[c]);
    yf[c * TB_H + 512] = o;
    dot += o * wlast;
    unsigned tt[NPL];
    split_n<NPL>(o, tt);
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      u32x4 t = z;
      t[0] = tt[p];
      *reinterpret_cast<u32x4*>(ypf + (p * TB_C + c) * TB_KP + 512) = t;
      *reinterpret_cast<u32x4*>(ypf + (p * TB_C + c) * TB_KP + 520) = z;
    }
  }
  dot = wave_sum(dot);
  if (lane == 0) xh[(int64_t)f * TB_H + 512] = dot + bias[0];
  if (zero_xh) {  // uniform
    float* xr = xh + (int64_t)f * TB_H + 8 * lane;
    *reinterpret_cast<packed4*>(xr) = packed4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<packed4*>(xr + 4) = packed4{0.f, 0.f, 0.f, 0.f};
  }
}

// ---- weight gradient:  dW[t][c] = sum_f sum_i y[f][c][i] * G[f][i + t - 512]
//
// Computed as the cross product P_c[i][q] = sum_f y[f][c][i] * G[f][q] (same MAC count, plain GEMM with
// the FRAME as reduction index) followed by a sum along the diagonals t = q - i + 512.  Both operands
// exist as row-major bf16 planes (toep_yp, toep_gp: rows = frames); a row-major [32 frames][columns] tile in
// LDS feeds the MFMA through ds_read_b64_tr_b16 (hardware 4x16 transpose: lane l receives 4 consecutive
// FRAMES of column l&15; semantics pinned by scripts/microbench/trread.hip).
//
// Workgroup = channel c, 128 bins i x 256 bins q, frames [z*fchunk, (z+1)*fchunk); 8 waves as 2 (i) x 4 (q),
// wave tile 64 x 64 (2 x 2 MFMA tiles).  The workgroups of the upper q half also own the column q = 512: a
// strip of four 32 x 32 tiles (only their first column is non-zero) dealt to the waves wc < 2 on even
// k-steps and wc >= 2 on odd ones (one extra accumulator per wave).  Row i = 512 of y: k_toep_wgrad_row512.
constexpr int WG_KF = 16;                       // frames per staged chunk = one k-step; two LDS buffers
constexpr int WG_RSA = 128 * 2 + 64;            // bytes per LDS row, A tile (320 = 64 mod 256: the 4 rows of a
constexpr int WG_RSB = 256 * 2 + 64;            //   transpose read sit in different bank quarters); B: 576
constexpr int WG_APL = WG_KF * WG_RSA, WG_BPL = WG_KF * WG_RSB;
constexpr int wg_buf(int npl) { return npl * (WG_APL + WG_BPL); }  // NPL = 3: 15 360 + 27 648 bytes per buffer
constexpr int wg_lds(int npl) { return 2 * wg_buf(npl); }

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 tr_read8(const unsigned char* p, int row4_bytes) {
  // two transpose reads: frames +0..3 and +4..7 of this lane's column
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + row4_bytes));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);  // register pairs side by side, no ALU work
  return __builtin_bit_cast(u32x4, ab);
}

template <int NPL>
__global__ void __launch_bounds__(512, 2) k_toep_wgrad_bf16(const unsigned short* __restrict__ yp,  // [F][NPL][8][528]
                                                             const unsigned short* __restrict__ gp,  // [F][NPL][528]
                                                             float* __restrict__ dW,                 // [1025][8] atomicAdd
                                                             int F, int fchunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WG_BUF = wg_buf(NPL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  const int i0 = (blockIdx.x >> 1) * 128, q0 = (blockIdx.x & 1) * 256, c = blockIdx.y;
  const bool strip = (blockIdx.x & 1) != 0;  // uniform: this workgroup also owns q = 512
  const int fb = blockIdx.z * fchunk, fe = min(F, fb + fchunk);

  // ---- staging (16-byte pieces) of one 16-frame chunk into buffer `buf`:
  //      A 3 planes x 16 rows x 16: threads < 256 -> (row tid>>4, piece tid&15), plane k;
  //      B 3 x 16 x 32: thread -> (row tid>>5, piece tid&31), plane k;
  //      strip: bins 512..527 = pieces 64, 65 of the G row -> tile pieces 32, 33: threads < 32 * NPL
  const int arow = (tid >> 4) & 15, apc = tid & 15;
  const int brow = tid >> 5, bpc = tid & 31;
  const int epl = tid >> 5, erow = (tid >> 1) & 15, epc = tid & 1;
  u32x4 sta[NPL], stb[NPL], ste;
  // per-thread source pointers of chunk fb, advanced by 16 frames per chunk (address arithmetic in the
  // staging phases competes with the partner wave's MFMAs: keep it to three 64-bit adds per chunk)
  const unsigned char* pa = reinterpret_cast<const unsigned char*>(yp) + ((size_t)(fb + arow) * NPL * TB_C + c) * (TB_KP * 2) + (i0 * 2 + apc * 16);
  const unsigned char* pb = reinterpret_cast<const unsigned char*>(gp) + (size_t)(fb + brow) * NPL * (TB_KP * 2) + q0 * 2 + bpc * 16;
  const unsigned char* pe = reinterpret_cast<const unsigned char*>(gp) + ((size_t)(fb + erow) * NPL + epl) * (TB_KP * 2) + (64 + epc) * 16;
  auto gload = [&](int f0) __attribute__((always_inline)) {
    if (f0 + WG_KF <= F) {  // uniform: all 16 frames exist
      if (tid < 256) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) sta[pl] = *reinterpret_cast<const u32x4*>(pa + (size_t)pl * TB_C * TB_KP * 2);
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) stb[pl] = *reinterpret_cast<const u32x4*>(pb + pl * (TB_KP * 2));
      if (strip && tid < 32 * NPL) ste = *reinterpret_cast<const u32x4*>(pe);
    } else {  // last chunk of the batch: clamp the frame index (rows past the end are zeroed in lstore)
      if (tid < 256) {
        int f = f0 + arow;
        f = f < F ? f : F - 1;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(yp) + ((size_t)f * NPL * TB_C + c) * (TB_KP * 2) + (i0 * 2 + apc * 16);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) sta[pl] = *reinterpret_cast<const u32x4*>(src + (size_t)pl * TB_C * TB_KP * 2);
      }
      {
        int f = f0 + brow;
        f = f < F ? f : F - 1;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(gp) + (size_t)f * NPL * (TB_KP * 2) + q0 * 2 + bpc * 16;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) stb[pl] = *reinterpret_cast<const u32x4*>(src + pl * (TB_KP * 2));
      }
      if (strip && tid < 32 * NPL) {
        int f = f0 + erow;
        f = f < F ? f : F - 1;
        ste = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(gp) + ((size_t)f * NPL + epl) * (TB_KP * 2) + (64 + epc) * 16);
      }
    }
    pa += (size_t)WG_KF * NPL * TB_C * TB_KP * 2;
    pb += (size_t)WG_KF * NPL * TB_KP * 2;
    pe += (size_t)WG_KF * NPL * TB_KP * 2;
  };
  auto lstore = [&](int f0, int buf) __attribute__((always_inline)) {
    unsigned char* sA = smem + buf * WG_BUF;
    unsigned char* sB = sA + NPL * WG_APL;
    const u32x4 z = {0u, 0u, 0u, 0u};
    const bool tail = f0 + WG_KF > fe;  // uniform: frames past the chunk contribute zero (A rows zeroed)
    if (tid < 256) {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        *reinterpret_cast<u32x4*>(sA + pl * WG_APL + arow * WG_RSA + apc * 16) = (tail && f0 + arow >= fe) ? z : sta[pl];
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x4*>(sB + pl * WG_BPL + brow * WG_RSB + bpc * 16) = stb[pl];
    if (strip && tid < 32 * NPL) *reinterpret_cast<u32x4*>(sB + epl * WG_BPL + erow * WG_RSB + (32 + epc) * 16) = ste;
  };

  // ---- fragment addresses (transpose reads): lane -> (frame row (l&15)>>2 (+8*lh), column quad 4*(l&3) + 16*((l>>4)&1))
  const int trow = ((lane & 15) >> 2) + 8 * lh, tcol = 4 * (lane & 3) + 16 * ((lane >> 4) & 1);
  const int aoff = trow * WG_RSA + (64 * wr + tcol) * 2;                 // + buf + plane*WG_APL + 64*ri
  const int boff = NPL * WG_APL + trow * WG_RSB + (64 * wc + tcol) * 2;    // + buf + plane*WG_BPL + 64*cj
  const int eoff = NPL * WG_APL + trow * WG_RSB + (256 + tcol) * 2;        // the q = 512 strip (tile columns 256..287)
  const int eri = wc & 1;                                                // this wave's strip row tile = 2*wr + eri, on chunks of parity wc>>1

  f32x16 acc[2][2], acce;
#pragma unroll
  for (int ri = 0; ri < 2; ++ri)
#pragma unroll
    for (int cj = 0; cj < 2; ++cj) acc[ri][cj] = zero16();
  acce = zero16();

  u32x4 fa[2][NPL], fbq[2][NPL], fe3[NPL];
  auto loadF = [&](int buf, bool mine) __attribute__((always_inline)) {
    const unsigned char* sb = smem + buf * WG_BUF;
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[ri][pl] = tr_read8(sb + pl * WG_APL + aoff + ri * 64, 4 * WG_RSA);
#pragma unroll
    for (int cj = 0; cj < 2; ++cj)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fbq[cj][pl] = tr_read8(sb + pl * WG_BPL + boff + cj * 64, 4 * WG_RSB);
    if (mine) {  // wave-uniform: this wave's turn on the q = 512 strip
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fe3[pl] = tr_read8(sb + pl * WG_BPL + eoff, 4 * WG_RSB);
    }
  };
  auto mm = [&](bool mine) __attribute__((always_inline)) {
    using PR = Prod<NPL>;
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int ri = 0; ri < 2; ++ri)
#pragma unroll
        for (int cj = 0; cj < 2; ++cj) acc[ri][cj] = mfma_bf16(fa[ri][PR::A[t]], fbq[cj][PR::B[t]], acc[ri][cj]);
    if (mine) {  // wave-uniform
      u32x4 af[NPL];
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) af[pl] = eri == 0 ? fa[0][pl] : fa[1][pl];
#pragma unroll
      for (int t = 0; t < PR::N; ++t) acce = mfma_bf16(af[PR::A[t]], fe3[PR::B[t]], acce);
    }
  };

  // one barrier per chunk, two LDS buffers: a wave that finished the MFMAs of chunk n stores chunk n+1
  // into the other buffer while slower waves still read buffer n (the waves of a SIMD drift apart and
  // overlap each other's load and MFMA phases)
#if VAENPVC_PROF
  long long pc[6] = {0, 0, 0, 0, 0, 0};
  TBPROF_T(k0);
#endif
  if (fb < fe) {
    gload(fb);
    lstore(fb, 0);
  }
  __syncthreads();
  int n = 0;
  for (int f0 = fb; f0 < fe; f0 += WG_KF, ++n) {
    const bool more = f0 + WG_KF < fe;
    const bool mine = strip && (wc >> 1) == (n & 1);
    TBPROF_T(t0);
    if (more) gload(f0 + WG_KF);
    __builtin_amdgcn_sched_barrier(0);
    TBPROF_T(t1);
    loadF(n & 1, mine);
#if VAENPVC_PROF
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
    TBPROF_T(t2);
    mm(mine);
#if VAENPVC_PROF
    asm volatile("s_nop 0" ::: "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
    TBPROF_T(t3);
    if (more) lstore(f0 + WG_KF, (n + 1) & 1);
#if VAENPVC_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    TBPROF_T(t4);
    __syncthreads();
#if VAENPVC_PROF
    TBPROF_T(t5);
    pc[0] += t1 - t0; pc[1] += t2 - t1; pc[2] += t3 - t2; pc[3] += t4 - t3; pc[4] += t5 - t4;
#endif
  }
  TBPROF_T(k1);

  // ---- epilogue: diagonals.  Wave tile = rows i0+64wr .. +63, columns q0+64wc .. +63: d = col - row in [-63, 63]
  // all 8 wave tiles of the workgroup (128 x 256) share 383 diagonals: reduce them in LDS first,
  // then ONE global atomic per diagonal and workgroup
  float* dg = reinterpret_cast<float*>(smem);
  if (tid < 384) dg[tid] = 0.f;
  __syncthreads();
  const int dbase = 64 * wc - 64 * wr + 127;  // + (cl - row) in [-63, 63]  ->  [0, 382]
#pragma unroll
  for (int ri = 0; ri < 2; ++ri)
#pragma unroll
    for (int cj = 0; cj < 2; ++cj)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int row = ri * 32 + acc_row(reg, lane);
        int cl = cj * 32 + l31;
        atomicAdd(&dg[dbase + cl - row], acc[ri][cj][reg]);
      }
  __syncthreads();
  if (tid < 383) {
    int t = q0 - i0 + (tid - 127) + 512;  // in [1, 1023]
    atomicAdd(dW + t * TB_C + c, dg[tid]);
  }
  // the q = 512 strip: column 512 is lane l31 == 0 of the strip tile; t = 512 - i + 512
  if (strip && l31 == 0) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      int i = i0 + 64 * wr + 32 * eri + acc_row(reg, lane);
      atomicAdd(dW + (1024 - i) * TB_C + c, acce[reg]);
    }
  }
#if VAENPVC_PROF
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TBPROF_T(k2);
    if ((threadIdx.x & 63) == 0) {
      atomicAdd(g_tw_prof + 0, 1ull);
      for (int i = 0; i < 5; ++i) atomicAdd(g_tw_prof + 1 + i, (unsigned long long)pc[i]);
      atomicAdd(g_tw_prof + 6, (unsigned long long)(k2 - k1));
      atomicAdd(g_tw_prof + 7, (unsigned long long)(k2 - k0));
    }
  }
#endif
}


// ---- the same weight gradient with 32-frame chunks (NPL <= 2: two planes leave the LDS room): two MFMA k-steps
//      (24 MFMAs per wave at NPL = 2) between barriers instead of one.  Same tile, same fragment reads, same epilogue.
constexpr int W2_KF = 32;
constexpr int w2_buf(int npl) { return npl * W2_KF * (WG_RSA + WG_RSB); }
constexpr int w2_lds(int npl) { return 2 * w2_buf(npl); }   // 114 688 bytes at NPL = 2

#ifndef VAENPVC_WG_FENCE
#define VAENPVC_WG_FENCE 1
#endif
// scheduling hint for the regions of the pipelined loop below: N times (one MFMA, then CNT instructions of class MASK:
// 0x100 LDS read, 0x200 LDS write) -- LDS traffic issues in the shadow of the 32-cycle MFMAs instead of between their bursts
#if VAENPVC_WG_FENCE
#define WG_INTERLEAVE(N, MASK, CNT)                          \
  _Pragma("unroll") for (int _i = 0; _i < (N); ++_i) {       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        \
    __builtin_amdgcn_sched_group_barrier((MASK), (CNT), 0);   \
  }
#else
#define WG_INTERLEAVE(N, MASK, CNT)
#endif
template <int NPL>
__global__ void __launch_bounds__(512, 2) k_toep_wgrad_bf16_k32(const unsigned short* __restrict__ yp,  // [F][NPL][8][528]
                                                                 const unsigned short* __restrict__ gp,  // [F][NPL][528]
                                                                 float* __restrict__ dW,                 // [1025][8] atomicAdd
                                                                 int F, int fchunk) {
  static_assert(NPL <= 2, "three planes do not fit two 32-frame buffers");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int APL = W2_KF * WG_RSA, BPL = W2_KF * WG_RSB, BUF = w2_buf(NPL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wr = wave >> 2, wc = wave & 3;
  // XCD-aware order (1-D grid; workgroup b runs on XCD b % 8): XCD = channel, and on an XCD the 8 tiles of one frame
  // chunk run back to back -- a channel's y planes are then fetched into ONE L2 (both q tiles of an i range share
  // them there) instead of two; the small d(xh) planes are fetched by every XCD from the Infinity Cache
  const int c = blockIdx.x & 7, tl = (blockIdx.x >> 3) & 7, zc = blockIdx.x >> 6;
  const int i0 = (tl >> 1) * 128, q0 = (tl & 1) * 256;
  const bool strip = (tl & 1) != 0;  // uniform: this workgroup also owns q = 512
  const int fb = zc * fchunk, fe = min(F, fb + fchunk);
  // staging of one 32-frame chunk (16-byte pieces): A 32 rows x 16 pieces: one per thread and plane;
  // B 32 rows x 32 pieces: two per thread and plane (rows brow, brow + 16); strip: 32 rows x 2 pieces x NPL: threads < 64*NPL
  const int arow = tid >> 4, apc = tid & 15;
  const int brow = tid >> 5, bpc = tid & 31;
  const int epl = tid >> 6, erow = (tid >> 1) & 31, epc = tid & 1;
  u32x4 sta[NPL], stb[NPL][2], ste;
  const unsigned char* Y8 = reinterpret_cast<const unsigned char*>(yp);
  const unsigned char* G8 = reinterpret_cast<const unsigned char*>(gp);
  auto gload = [&](int f0) __attribute__((always_inline)) {
    const int fa_ = min(f0 + arow, F - 1), fb0 = min(f0 + brow, F - 1), fb1 = min(f0 + brow + 16, F - 1);
    const unsigned char* pa = Y8 + ((size_t)fa_ * NPL * TB_C + c) * (TB_KP * 2) + (i0 * 2 + apc * 16);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) sta[pl] = *reinterpret_cast<const u32x4*>(pa + (size_t)pl * TB_C * TB_KP * 2);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      stb[pl][0] = *reinterpret_cast<const u32x4*>(G8 + ((size_t)fb0 * NPL + pl) * (TB_KP * 2) + q0 * 2 + bpc * 16);
      stb[pl][1] = *reinterpret_cast<const u32x4*>(G8 + ((size_t)fb1 * NPL + pl) * (TB_KP * 2) + q0 * 2 + bpc * 16);
    }
    if (strip && tid < 64 * NPL) {
      const int fe_ = min(f0 + erow, F - 1);
      ste = *reinterpret_cast<const u32x4*>(G8 + ((size_t)fe_ * NPL + epl) * (TB_KP * 2) + (64 + epc) * 16);
    }
  };
  auto lstore = [&](int f0, int buf) __attribute__((always_inline)) {
    unsigned char* sA = smem + buf * BUF;
    unsigned char* sB = sA + NPL * APL;
    const u32x4 z = {0u, 0u, 0u, 0u};
    const bool zero = f0 + arow >= fe;   // frames past the chunk contribute zero (A rows zeroed)
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x4*>(sA + pl * APL + arow * WG_RSA + apc * 16) = zero ? z : sta[pl];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      *reinterpret_cast<u32x4*>(sB + pl * BPL + brow * WG_RSB + bpc * 16) = stb[pl][0];
      *reinterpret_cast<u32x4*>(sB + pl * BPL + (brow + 16) * WG_RSB + bpc * 16) = stb[pl][1];
    }
  };
  auto lstore_strip = [&](int buf) __attribute__((always_inline)) {   // (a divergent branch: kept out of the interleaved regions)
    unsigned char* sB = smem + buf * BUF + NPL * APL;
    if (strip && tid < 64 * NPL) *reinterpret_cast<u32x4*>(sB + epl * BPL + erow * WG_RSB + (32 + epc) * 16) = ste;
  };
  // fragment addresses (transpose reads) inside a 16-frame half of the chunk
  const int trow = ((lane & 15) >> 2) + 8 * lh, tcol = 4 * (lane & 3) + 16 * ((lane >> 4) & 1);
  const int aoff = trow * WG_RSA + (64 * wr + tcol) * 2;
  const int boff = NPL * APL + trow * WG_RSB + (64 * wc + tcol) * 2;
  const int eoff = NPL * APL + trow * WG_RSB + (256 + tcol) * 2;
  const int eri = wc & 1;   // this wave's strip row tile = 2*wr + eri, on k-steps of parity wc >> 1
  f32x16 acc[2][2], acce;
#pragma unroll
  for (int ri = 0; ri < 2; ++ri)
#pragma unroll
    for (int cj = 0; cj < 2; ++cj) acc[ri][cj] = zero16();
  acce = zero16();
  // Software pipeline.  Measured by ablation on the plain loop (load / 2 k-steps / store / barrier): with every wave of
  // the workgroup in the same phase at the same time the phases simply ADD -- 186 us MFMA + 101 fragment reads + 57 LDS
  // stores + 119 waiting for the prefetched chunk + 123 prologue / epilogue = 587.  Here: two fragment sets, so that
  // the transposed reads of the next k-step are in flight during the MFMAs of the current one, across the barrier
  // too; the staging registers are refilled right after they were written to LDS (a chunk is requested more than a full
  // iteration ahead); and inside a region the LDS instructions are issued BETWEEN the MFMAs (sched_group_barrier), which
  // needs straight-line code: iterations with a successor run in a branch-free loop body, the strip column's three
  // MFMAs and its 16-byte store sit at the region boundaries.
  u32x4 fa[2][2][NPL], fbq[2][2][NPL], fe3[2][NPL];
  int n = 0;   // chunk counter
  const bool mine0 = strip && (wc >> 1) == 0, mine1 = strip && (wc >> 1) == 1;   // wave-uniform: owner of the strip tile on k-step 0 / 1
  auto readF = [&](int set, int buf, int ks) __attribute__((always_inline)) {
    const unsigned char* sb = smem + buf * BUF;
#pragma unroll
    for (int ri = 0; ri < 2; ++ri)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[set][ri][pl] = tr_read8(sb + pl * APL + ks * 16 * WG_RSA + aoff + ri * 64, 4 * WG_RSA);
#pragma unroll
    for (int cj = 0; cj < 2; ++cj)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fbq[set][cj][pl] = tr_read8(sb + pl * BPL + ks * 16 * WG_RSB + boff + cj * 64, 4 * WG_RSB);
    // (read by every wave: the strip columns of non-strip workgroups are ordinary finite LDS contents, never used)
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) fe3[set][pl] = tr_read8(sb + pl * BPL + ks * 16 * WG_RSB + eoff, 4 * WG_RSB);
  };
  // MFMAs of one k-step for the row tiles [r0, r1)
  auto mma = [&](int set, int r0, int r1) __attribute__((always_inline)) {
    using PR = Prod<NPL>;
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int ri = r0; ri < r1; ++ri)
#pragma unroll
        for (int cj = 0; cj < 2; ++cj) acc[ri][cj] = mfma_bf16(fa[set][ri][PR::A[t]], fbq[set][cj][PR::B[t]], acc[ri][cj]);
  };
  auto mma_strip = [&](int set) __attribute__((always_inline)) {
    using PR = Prod<NPL>;
    u32x4 af[NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) af[pl] = eri == 0 ? fa[set][0][pl] : fa[set][1][pl];
#pragma unroll
    for (int t = 0; t < PR::N; ++t) acce = mfma_bf16(af[PR::A[t]], fe3[set][PR::B[t]], acce);
  };
  if (fb < fe) {
    gload(fb);
    lstore(fb, 0);
    lstore_strip(0);
    if (fb + W2_KF < fe) gload(fb + W2_KF);   // registers: chunk 1
  }
  __syncthreads();
  if (fb < fe) readF(0, 0, 0);
  int f0 = fb;
  for (; f0 + W2_KF < fe; f0 += W2_KF, ++n) {   // iterations with a successor
    const int buf = n & 1;
    // region A: fragment reads of k-step 1 under the MFMAs of k-step 0
    readF(1, buf, 1);
    mma(0, 0, 2);
    WG_INTERLEAVE(12, 0x100, 2);
    __builtin_amdgcn_sched_barrier(0);
    if (mine0) mma_strip(0);
    lstore_strip(buf ^ 1);
    // region B: LDS stores of the next chunk (requested one iteration ago) under the first half of k-step 1
    lstore(f0 + W2_KF, buf ^ 1);
    mma(1, 0, 1);
    WG_INTERLEAVE(6, 0x200, 2);
    __builtin_amdgcn_sched_barrier(0);
    if (f0 + 2 * W2_KF < fe) gload(f0 + 2 * W2_KF);   // the staging registers go straight back to work
    __syncthreads();   // chunk n + 1 is in LDS; every wave holds its fragments of chunk n in registers
    // region C: fragment reads of the next chunk's k-step 0 under the second half of k-step 1
    readF(0, buf ^ 1, 0);
    mma(1, 1, 2);
    WG_INTERLEAVE(6, 0x100, 3);
    __builtin_amdgcn_sched_barrier(0);
    if (mine1) mma_strip(1);
  }
  if (f0 < fe) {   // last chunk
    readF(1, n & 1, 1);
    mma(0, 0, 2);
    if (mine0) mma_strip(0);
    mma(1, 0, 2);
    if (mine1) mma_strip(1);
  }
  __syncthreads();
  // ---- epilogue (as k_toep_wgrad_bf16): 383 diagonals reduced in LDS, one global atomic per diagonal and workgroup
  float* dg = reinterpret_cast<float*>(smem);
  if (tid < 384) dg[tid] = 0.f;
  __syncthreads();
  const int dbase = 64 * wc - 64 * wr + 127;
#pragma unroll
  for (int ri = 0; ri < 2; ++ri)
#pragma unroll
    for (int cj = 0; cj < 2; ++cj)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int row = ri * 32 + acc_row(reg, lane);
        int cl = cj * 32 + l31;
        atomicAdd(&dg[dbase + cl - row], acc[ri][cj][reg]);
      }
  __syncthreads();
  if (tid < 383) {
    int t = q0 - i0 + (tid - 127) + 512;
    atomicAdd(dW + t * TB_C + c, dg[tid]);
  }
  if (strip && l31 == 0) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      int i = i0 + 64 * wr + 32 * eri + acc_row(reg, lane);
      atomicAdd(dW + (1024 - i) * TB_C + c, acce[reg]);
    }
  }
}


// ---- the weight gradient with 128 x 128 WAVE tiles: four waves (one per SIMD, 2 x 2) on a 256 (i) x 256 (q) tile.
// k_toep_wgrad_bf16_k32 reads 9 KB of fragments from LDS per 12 MFMAs and wave and writes the staged chunk through
// registers: ~125 bytes of LDS traffic per CU and clock at the MFMA rate, i.e. it is bound by the LDS (128 B/clk), not
// by the matrix pipe.  Here a wave reads 18 KB per 51 MFMAs (65 B/clk per CU at the MFMA rate) and nothing passes
// through staging registers:
//   * operands arrive by LDS-DMA (global_load_lds_dwordx4, lane-linear 1-KiB blocks = two frames of 256 bins), into a
//     ring of four 16-frame stages; a stage is requested four iterations before it is multiplied and waited for with a
//     COUNTED vmcnt (the two newest requests may still be in flight), one bare s_barrier per stage;
//   * rows are unpadded (512 B); the 16-byte piece index is XOR-ed with (frame & 3) << 2 on the way in (each lane picks
//     its global piece) and on the way out, which puts the four frames of a transpose read into different 64-byte bank
//     quarters;
//   * the fragments of stage t + 1 are read between the MFMAs of stage t (two register sets, sched_group_barrier);
//   * the q = 512 column is a matrix-VECTOR product: it runs on the vector ALU in the shadow of the MFMAs (12
//     v_dot2_f32_bf16 per stage and wave on the y fragments the wave holds anyway, d(xh)[.][512] broadcast from the lane
//     that read it), one fp32 register instead of a 17th accumulator tile (16 tiles = all 256 AGPRs).  The two q tiles
//     of an i range split its eight row tiles between them, so all workgroups carry the same work (in k32 the odd q
//     tiles carry 12.5 % more and the others wait for them at the end of the launch).  The wave's row tiles are rotated
//     so that its strip tile is fragment slot 0 (the fragment addresses are per-lane registers anyway).
#ifndef VAENPVC_W4_ABL
#define VAENPVC_W4_ABL 0   // developer ablation of the main loop (wrong results): 1 no requests, 2 no fragment reads, 4 no wait / barrier, 8 no strip
#endif
constexpr int W4_KF = 16;                 // frames per stage = one MFMA k-step
constexpr int W4_RS = 512;                // bytes per LDS row (256 bins)
constexpr int W4_APL = W4_KF * W4_RS;     // bytes per plane, operand and stage
constexpr int W4_E = 4096;                // strip: bins 512..575 of both planes (only 512 is used), one 1-KiB block per wave
constexpr int W4_NS = 4;                  // ring slots
constexpr int w4_stage(int npl) { return npl * 2 * W4_APL + W4_E; }   // 36 864 bytes at NPL = 2
constexpr int W4_EPI_LDS = 4 * 7 * 32 * 33 * 4;                       // the epilogue's blocks: 118 272 bytes
constexpr int w4_lds(int npl) { return W4_NS * w4_stage(npl) > W4_EPI_LDS ? W4_NS * w4_stage(npl) : W4_EPI_LDS; }   // 147 456 at two planes
template <int N>
struct IntC {
  static constexpr int value = N;
};

// LDS-DMA: every lane passes its own global address, the wave ONE LDS byte address (in M0); lane l's 16 bytes land at
// base + 16 l (scripts/microbench/ldsdma.hip); completion is counted by vmcnt.  Written as inline assembly: behind the
// builtin the compiler orders every later LDS read after the transfer with s_waitcnt vmcnt(0), which would undo the ring.
__device__ __forceinline__ void lds_dma16(const unsigned char* g, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_wave_base), "v"(g) : "memory");
}
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The requests run up to five stages (80 frames) past the last frame of the batch without clamping: the plane buffers are
// sized for three planes (model.cpp), this kernel runs with at most two, so the addresses stay inside the buffers; rows of
// frames >= F are cleared in LDS before they are multiplied (clear_tail), later stages are never multiplied.
template <int NPL>
__global__ void __launch_bounds__(256) k_toep_wgrad_bf16_w4(const unsigned short* __restrict__ yp,  // [F][NPL][8][528]
                                                            const unsigned short* __restrict__ gp,  // [F][NPL][528]
                                                            float* __restrict__ dW,                 // [1025][8] atomicAdd
                                                            int F, int fchunk) {
  static_assert(NPL <= 2, "ring of four stages, slack of the plane buffers: two planes");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using PR = Prod<NPL>;
  constexpr int STAGE = w4_stage(NPL), BOFF = NPL * W4_APL, EOFF = 2 * NPL * W4_APL;
  constexpr int NB = 2 * NPL;            // 1-KiB blocks per wave, operand and stage
  constexpr int NDMA = 2 * NB + 1;       // DMA instructions per wave and stage
  constexpr int NT8 = 9 * NPL;           // fragment reads (2 x ds_read_b64_tr_b16 each) per wave and stage
  constexpr int NM = 16 * PR::N;         // MFMAs per wave and stage
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
#ifndef VAENPVC_W4_XCD
#define VAENPVC_W4_XCD 1
#endif
  // XCD-aware order (workgroup b runs on XCD b % 8).  XCD = frame chunk: the 32 workgroups of a chunk (8 channels x 4 tiles)
  // walk the same frames in step, so a chunk's y planes AND its d(xh) planes pass through one L2, once (with XCD = channel
  // every XCD fetched all of d(xh): 1.96 x the algorithmic bytes)
#if VAENPVC_W4_XCD
  const int zc = blockIdx.x % (gridDim.x >> 5), rest = blockIdx.x / (gridDim.x >> 5), tl = rest & 3, c = rest >> 2;
#else
  const int c = blockIdx.x & 7, tl = (blockIdx.x >> 3) & 3, zc = blockIdx.x >> 5;
#endif
  const int it = tl >> 1, qt = tl & 1, i0 = it * 256, q0 = qt * 256;
  const int fb = zc * fchunk, fe = min(F, fb + fchunk);
  if (fb >= fe) return;
  const int nst = (fe - fb + W4_KF - 1) / W4_KF;
  const int eri = 2 * qt + wc;           // this wave's strip row tile; fragment slot ri holds row tile (ri + eri) & 3

  // ---- DMA: block k of this wave = (plane, frame pair) of the stage; lane -> (frame 2 fp + (lane >> 5), LDS piece lane & 31);
  //      the piece fetched is (lane & 31) ^ ((frame & 3) << 2).  Strip block of wave w: frames 4 w ..+3, both planes, bins
  //      512..575: lane -> (plane lane >> 5, frame (lane >> 3) & 3, piece lane & 7)
  constexpr size_t YF = (size_t)NPL * TB_C * TB_KP * 2, GF = (size_t)NPL * TB_KP * 2;   // bytes per frame
  const unsigned char* Y8 = reinterpret_cast<const unsigned char*>(yp);
  const unsigned char* G8 = reinterpret_cast<const unsigned char*>(gp);
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned char* src[NDMA];        // per-lane addresses of stage 0, advanced by a scalar per stage
  unsigned dst[NDMA];                    // wave-uniform LDS offsets inside a stage
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int bid = wave * NB + k, pl = bid >> 3, fp = bid & 7, fr = 2 * fp + lh;
    const int gpc = l31 ^ ((fr & 3) << 2);
    src[2 * k] = Y8 + (size_t)(fb + fr) * YF + ((size_t)pl * TB_C + c) * (TB_KP * 2) + i0 * 2 + gpc * 16;
    src[2 * k + 1] = G8 + (size_t)(fb + fr) * GF + (size_t)pl * (TB_KP * 2) + q0 * 2 + gpc * 16;
    dst[2 * k] = pl * W4_APL + fp * 1024;
    dst[2 * k + 1] = BOFF + pl * W4_APL + fp * 1024;
  }
  {
    const int pl = min(lh, NPL - 1), fr = 4 * wave + ((lane >> 3) & 3);
    src[2 * NB] = G8 + (size_t)(fb + fr) * GF + (size_t)pl * (TB_KP * 2) + (64 + (lane & 7)) * 16;
    dst[2 * NB] = EOFF + wave * 1024;
  }
  // request g of stage s into ring slot `slot`
  auto dma = [&](int g, int s, int slot) __attribute__((always_inline)) {
    const size_t off = (size_t)s * (W4_KF * ((g & 1) || g == 2 * NB ? GF : YF));
    lds_dma16(src[g] + off, lds0 + slot * STAGE + dst[g]);
  };
  // rows of frames >= F (nv valid rows) of a landed stage are cleared: y planes, d(xh) planes and the strip
  auto clear_tail = [&](int slot, int nv) __attribute__((always_inline)) {
    unsigned char* sb = smem + slot * STAGE;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < 2 * NPL * W4_KF * 32; i += 256)   // [operand][plane][row][32 pieces]
      if (((i >> 5) & (W4_KF - 1)) >= nv) *reinterpret_cast<u32x4*>(sb + i * 16) = z;
    for (int i = tid; i < W4_E / 16; i += 256)              // [wave][plane][row & 3][8 pieces]
      if (4 * (i >> 6) + ((i >> 3) & 3) >= nv) *reinterpret_cast<u32x4*>(sb + EOFF + i * 16) = z;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  const int last_nv = (fe - fb) - (nst - 1) * W4_KF;   // valid frames of the last stage (16: nothing to clear)

  // ---- fragment addresses (transpose reads): lane -> frame row ((l & 15) >> 2) + 8 lh (second read: + 4), bins 4 (l & 3) + 16 ((l >> 4) & 1) ..+3
  const int trow = ((lane & 15) >> 2) + 8 * lh, tcol = 4 * (lane & 3) + 16 * ((lane >> 4) & 1), s3 = trow & 3;
  int a_off[4], b_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int rt = (r + eri) & 3;
    a_off[r] = trow * W4_RS + (16 * wr + ((rt ^ s3) << 2) + (tcol >> 3)) * 16 + (tcol & 7) * 2;
    b_off[r] = BOFF + trow * W4_RS + (16 * wc + ((r ^ s3) << 2) + (tcol >> 3)) * 16 + (tcol & 7) * 2;
  }
  const int e_off = EOFF + (trow >> 2) * 1024 + (trow & 3) * 128 + (tcol & 15) * 2;   // (+ 512 per plane, + 1024 for frames + 4)

  f32x16 acc[4][4];
#pragma unroll
  for (int ri = 0; ri < 4; ++ri)
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) acc[ri][cj] = zero16();
  float accl = 0.f, acch = 0.f;   // row l31 of the strip tile times frames 0..7 / 8..15 of every stage (lanes of half lh keep the one that matches their fragment)
  u32x4 fa[2][4][NPL], fq[2][4][NPL], fx[2][NPL];
  // fragment read j of a stage: y row tiles, d(xh) column tiles, strip
  auto rd = [&](int set, int slot, int j) __attribute__((always_inline)) {
    const unsigned char* sb = smem + slot * STAGE;
    if (j < 4 * NPL)
      fa[set][j / NPL][j % NPL] = tr_read8(sb + (j % NPL) * W4_APL + a_off[j / NPL], 4 * W4_RS);
    else if (j < 8 * NPL)
      fq[set][(j - 4 * NPL) / NPL][j % NPL] = tr_read8(sb + (j % NPL) * W4_APL + b_off[(j - 4 * NPL) / NPL], 4 * W4_RS);
    else if (j < NT8)
      fx[set][j - 8 * NPL] = tr_read8(sb + (j - 8 * NPL) * 512 + e_off, 1024);
  };
  // MFMA m of a stage: product t = m >> 4 outermost (consecutive MFMAs touch different accumulators)
  auto mm = [&](int set, int m) __attribute__((always_inline)) {
    const int t = m >> 4, ri = (m >> 2) & 3, cj = m & 3;
    acc[ri][cj] = mfma_bf16(fa[set][ri][PR::A[t]], fq[set][cj][PR::B[t]], acc[ri][cj]);
  };
  // ---- prologue: stages 0 .. 3 requested, 0 and 1 landed
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int g = 0; g < NDMA; ++g) dma(g, s, s);
  wait_vmcnt<2 * NDMA>();
  __builtin_amdgcn_s_barrier();
  if (last_nv < W4_KF && nst <= 2) clear_tail(nst - 1, last_nv);
#pragma unroll
  for (int j = 0; j < NT8; ++j) rd(0, 0, j);

  // iteration t (slot S = t & 3, fragment set S & 1): the fragments of stage t are in registers, so its slot is free:
  // stage t + 4 is requested into it; stage t is multiplied while the fragments of stage t + 1 are read; at the end
  // stage t + 2 must have landed (t + 3 and t + 4 may be in flight: a stage has three iterations to arrive).  A wave
  // issues in order and one wave runs per SIMD: whatever stands between two MFMAs must fit the 32 cycles of the first, so
  // the requests are spread over the stage, ONE between two pairs of MFMAs (with two fragment reads and a broadcast)
  auto body = [&](auto S_, int t) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value;
    // slots: every MFMA is followed by ONE small piece of other work -- a fragment read (two ds_read_b64_tr_b16), or a
    // request (address add, M0, transfer)
#pragma unroll
    for (int g = 0; g < NDMA; ++g) {
      mm(S & 1, 3 * g);
      if (!(VAENPVC_W4_ABL & 2)) rd((S + 1) & 1, (S + 1) & 3, 2 * g);
      __builtin_amdgcn_sched_barrier(0);
      mm(S & 1, 3 * g + 1);
      if (!(VAENPVC_W4_ABL & 2)) rd((S + 1) & 1, (S + 1) & 3, 2 * g + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(S & 1, 3 * g + 2);
      if (!(VAENPVC_W4_ABL & 1)) dma(g, t + 4, S);
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(2 * NDMA >= NT8 && 3 * NDMA <= NM, "slots cover the reads");
#pragma unroll
    for (int m = 3 * NDMA; m < NM; ++m) mm(S & 1, m);
    if (!(VAENPVC_W4_ABL & 8)) {
      // strip: lane (l31, lh) of a B fragment holds bin l31, frames 8 lh ..+7 -- bin 512 sits in lanes 0 and 32.  Both go
      // to scalar registers (no LDS cross-lane traffic); every lane multiplies with both, the half is chosen at the end
#pragma unroll
      for (int tp = 0; tp < PR::N; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)fx[S & 1][PR::B[tp]][r], 0);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)fx[S & 1][PR::B[tp]][r], 32);
          accl = dot2_bf16(fa[S & 1][0][PR::A[tp]][r], lo, accl);
          acch = dot2_bf16(fa[S & 1][0][PR::A[tp]][r], hi, acch);
        }
      asm volatile("" : "+v"(accl), "+v"(acch));   // keeps the dot products in this stage (they would be sunk past the barrier otherwise)
      WG_INTERLEAVE(NM - 3 * NDMA - 1, 0x002, 2);  // two vector instructions behind each of the remaining MFMAs
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(VAENPVC_W4_ABL & 4)) {
      wait_vmcnt<2 * NDMA>();
      __builtin_amdgcn_s_barrier();
    }
    if (last_nv < W4_KF && t + 2 == nst - 1) clear_tail((S + 2) & 3, last_nv);   // uniform, once per launch at most
  };
  int t = 0;
  for (; t + 4 <= nst; t += 4) {
    body(IntC<0>{}, t);
    body(IntC<1>{}, t + 1);
    body(IntC<2>{}, t + 2);
    body(IntC<3>{}, t + 3);
  }
  if (t < nst) {
    body(IntC<0>{}, t);
    if (t + 1 < nst) {
      body(IntC<1>{}, t + 1);
      if (t + 2 < nst) body(IntC<2>{}, t + 2);
    }
  }
  wait_vmcnt<0>();   // requests past the last stage are still writing into the ring
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // ---- epilogue: the 511 diagonals of the 256 x 256 tile, WITHOUT LDS atomics (a wave-level ds_add_f32 takes ~150 cycles:
  //      the 256 per lane of the straightforward reduction cost 80 us of a 400 us launch).  (1) In registers: the 16 tiles
  //      of a wave lie on 7 tile diagonals (column tile - row tile); tiles of one tile diagonal are added element-wise.
  //      (2) The 7 sums go to LDS as 32 x 32 blocks (row pitch 33 floats).  (3) Thread D adds up diagonal D of the tile
  //      from the (at most two) blocks of every wave that cross it and issues the ONE global atomic of that diagonal.
  constexpr int BLK = 32 * 33;
  float* blk = reinterpret_cast<float*>(smem);   // [wave][7][32][33]: 118 272 bytes of the (now idle) ring
  {
    f32x16 R[7];
    auto reduce = [&](auto E_) __attribute__((always_inline)) {
      constexpr int E = decltype(E_)::value;
#pragma unroll
      for (int d = 0; d < 7; ++d) {
        bool first = true;
#pragma unroll
        for (int ri = 0; ri < 4; ++ri)
#pragma unroll
          for (int cj = 0; cj < 4; ++cj)
            if (cj - ((ri + E) & 3) + 3 == d) {
              R[d] = first ? acc[ri][cj] : R[d] + acc[ri][cj];
              first = false;
            }
      }
    };
    if (eri == 0) reduce(IntC<0>{});
    else if (eri == 1) reduce(IntC<1>{});
    else if (eri == 2) reduce(IntC<2>{});
    else reduce(IntC<3>{});
#pragma unroll
    for (int d = 0; d < 7; ++d)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) blk[(wave * 7 + d) * BLK + acc_row(reg, lane) * 33 + l31] = R[d][reg];
  }
  __syncthreads();
  for (int D = tid; D < 511; D += 256) {
    const int e = D - 255;   // column - row inside the workgroup tile
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int x = e - 128 * ((w & 1) - (w >> 1));   // = 32 td + bd inside wave w's tile, |td| <= 3, |bd| <= 31
      const int td1 = (x + 512) / 32 - 16;            // floor(x / 32)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int td = td1 + k, bd = x - 32 * td;
        if (td >= -3 && td <= 3 && bd >= -31 && bd <= 31) {
          const float* bp = blk + (w * 7 + td + 3) * BLK + bd;
          const int r0 = bd < 0 ? -bd : 0, r1 = bd > 0 ? 32 - bd : 32;
          for (int r = r0; r < r1; ++r) v += bp[r * 34];
        }
      }
    }
    atomicAdd(dW + (q0 - i0 + e + 512) * TB_C + c, v);
  }
  {   // the q = 512 column: t = 1024 - i; the two halves of the wave hold the two frame octets of row l31
    const float acce = lh ? acch : accl;
    const float tot = acce + __shfl_xor(acce, 32);
    if (lh == 0) atomicAdd(dW + (1024 - (i0 + 128 * wr + 32 * eri + l31)) * TB_C + c, tot);
  }
}

}  // namespace tuned
}  // namespace vaenpvc
