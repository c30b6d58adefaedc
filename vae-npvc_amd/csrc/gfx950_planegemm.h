// gfx950_planegemm.h -- plain GEMMs on the bf16 matrix cores for the dense-shaped layers (encoder heads,
// merge FC, encoder layer 4 written as a dense layer) in all three directions.
//
// Operands are "plane" tensors: an fp32 matrix X[rows][K] split into NPL bf16 terms X = t0 + t1 (+ t2),
// stored plane-major as unsigned short [NPL][rows][Kp] (Kp = K padded with zeros to a multiple of 64, so the
// kernels have no K tail).  Products keep the term pairs (i, j) with i + j < NPL and accumulate in fp32 inside
// v_mfma_f32_32x32x16_bf16 (gfx950_toep_bf16.h explains the arithmetic; NPL is the context's precision).
//
//   k_split_planes : fp32 activation / gradient tensor -> planes (optionally LayerNorm + lrelu on the way,
//                    util/layers.py:32-44,149, or the [a | b] concatenation of two tensors); HBM-bound
//   PackPlanesJob  : weight matrices -> planes, inside the step's single packing launch
//   k_gemm_nt      : C[m][n] = sum_k A[m][k] B[n][k] (+ bias[n] + T[idx[m]][n])     forward / input gradient
//   k_gemm_tn      : C[m][n] += sum_f A[f][m] B[f][n]                                weight gradient
//
// Encoder layer 4 (conv k7 s3 SAME on 7 positions -> 3 positions, util/layers.py:56-64) touches every input
// position from nearly every output position, so it is run as the dense layer [F, 128*7] x [896, 256*3] whose
// weight matrix holds W[h - 3j + 3][c][o] (zero where the tap index leaves 0..6: 6 of 21 (h, j) pairs): 29 % more
// MACs than the conv, at 5-16x the MAC rate, with the canonical [F][C][H] tensors as operands and results.
#pragma once
#include "gfx950_common.h"
#include "gfx950_toep_bf16.h"  // split_n, Prod, mfma_bf16, u32x4, packed4, tr_read8
#include "philox.h"

namespace vaenpvc {
namespace tuned {

constexpr int PG_KA = 64;  // K granule of every plane tensor

// row r of a plane tensor: element offset (r / R) * fs + x0 + (r % R) * step.  Plain matrices: R = INT_MAX, step = leading
// dimension; channel-last conv tensors: R rows per frame, fs elements per frame, rows may overlap (see k_cgemm below)
struct RowView {
  int R, fs, x0, step;
};
__device__ __forceinline__ int64_t view_off(const RowView& v, int r) {
  const int f = r / v.R, q = r - f * v.R;
  return (int64_t)f * v.fs + v.x0 + q * v.step;
}
static inline RowView plain_rows(int ld) { return RowView{0x7fffffff, 0, 0, ld}; }

// ---------------------------------------------------------------- producers
struct SplitArgs {
  const float* src;    // [rows][ld1], columns [0, k1)
  const float* src2;   // optional: columns [k1, K) come from src2[rows][ld2]
  int k1, ld1, ld2;
  const float* st;     // LN: per-row (mean, rstd)
  const float* gamma;  // LN: per channel, channel = column / lndiv
  const float* beta;
  int lndiv;
  int K, Kp;
  int64_t rows;
  unsigned short* dst;  // [NPL][rows][Kp]
};

// one thread = 8 consecutive columns of one row: two 16-byte loads, NPL 16-byte stores
template <int NPL, bool LN>
__global__ void __launch_bounds__(256) k_split_planes(SplitArgs a) {
  const int g8 = a.Kp >> 3;
  const int64_t total = a.rows * g8;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / g8;
    const int k0 = (int)(i - r * g8) * 8;
    float v[8];
    if (k0 + 8 <= a.k1) {
      const float* p = a.src + r * a.ld1 + k0;
      packed4 p0 = *reinterpret_cast<const packed4*>(p), p1 = *reinterpret_cast<const packed4*>(p + 4);
      v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
    } else if (a.src2 && k0 >= a.k1 && k0 + 8 <= a.K) {
      const float* p = a.src2 + r * a.ld2 + (k0 - a.k1);
      packed4 p0 = *reinterpret_cast<const packed4*>(p), p1 = *reinterpret_cast<const packed4*>(p + 4);
      v[0] = p0.x; v[1] = p0.y; v[2] = p0.z; v[3] = p0.w; v[4] = p1.x; v[5] = p1.y; v[6] = p1.z; v[7] = p1.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        v[j] = k < a.k1 ? a.src[r * a.ld1 + k] : ((a.src2 && k < a.K) ? a.src2[r * a.ld2 + (k - a.k1)] : 0.f);
      }
    }
    if constexpr (LN) {
      const float mean = a.st[2 * r], rstd = a.st[2 * r + 1];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const int c = (k < a.K ? k : a.K - 1) / a.lndiv;
        const float y = lnact_v(v[j], mean, rstd, a.gamma[c], a.beta[c]);
        v[j] = k < a.K ? y : 0.f;
      }
    }
    u32x4 pk[NPL];
    pack8<NPL>(v, pk);
#pragma unroll
    for (int p = 0; p < NPL; ++p) st_nt<VAENPVC_NT_B>(reinterpret_cast<u32x4*>(a.dst + ((int64_t)p * a.rows + r) * a.Kp + k0), pk[p]);
  }
}

// LayerNorm statistics of a layer's pre-LN output (N = C*H floats per frame, util/layers.py:32) AND the planes of its
// activated output in the same pass (one wave per frame, the frame stays in registers): replaces k_ln_stats_fast
// + k_split_planes<LN> for the tensors the dense-shaped layers consume (saves a read of the tensor)
template <int N, int H, int NPL>
__global__ void __launch_bounds__(256) k_ln_stats_planes(const float* __restrict__ a, float* __restrict__ st,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         unsigned short* __restrict__ dst, int F) {
  static_assert(N % 64 == 0, "plane rows are padded to 64 columns");
  constexpr int NV = N / 4, PER = cdiv(NV, 64);
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= F) return;
  const float4* p = reinterpret_cast<const float4*>(a + (int64_t)f * N);
  float4 v[PER];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    int idx = lane + 64 * i;
    v[i] = idx < NV ? p[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (lane + 64 * i < NV) {
      float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / N + LN_EPS);
  if (lane == 0) {
    st[2 * f] = mean;
    st[2 * f + 1] = rstd;
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int idx = lane + 64 * i;
    if (idx < NV) {
      const int e = 4 * idx;
      const float y4[4] = {lnact_v(v[i].x, mean, rstd, gamma[e / H], beta[e / H]),
                           lnact_v(v[i].y, mean, rstd, gamma[(e + 1) / H], beta[(e + 1) / H]),
                           lnact_v(v[i].z, mean, rstd, gamma[(e + 2) / H], beta[(e + 2) / H]),
                           lnact_v(v[i].w, mean, rstd, gamma[(e + 3) / H], beta[(e + 3) / H])};
      unsigned t01[NPL], t23[NPL];
      split_pair<NPL>(y4[0], y4[1], t01);
      split_pair<NPL>(y4[2], y4[3], t23);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        uint2 pk;
        pk.x = t01[pl];
        pk.y = t23[pl];
        *reinterpret_cast<uint2*>(dst + ((int64_t)pl * F + f) * N + e) = pk;
      }
    }
  }
}

template <int NPL>
inline void launch_split(const SplitArgs& a, hipStream_t s) {
  const int64_t total = a.rows * (a.Kp >> 3);
  const unsigned blocks = (unsigned)cmin_((int)((total + 255) / 256), 8192);
  if (a.st)
    hipLaunchKernelGGL((k_split_planes<NPL, true>), dim3(blocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_split_planes<NPL, false>), dim3(blocks), dim3(256), 0, s, a);
}

// merge backward: the planes of d(h) (what k_split_planes writes) AND the per-speaker column sums S[y_f][n] += d(h)[f][n]
// (what k_segsum_atomic adds, models/vae.py merge: the bias / embedding gradients derive from S) in ONE pass over d(h).
// grid (ceil(Kp/512), frame chunks), 4 waves: a lane owns 8 consecutive columns, a wave a quarter of the chunk's frames
// (the speaker of a frame is wave-uniform: NY x 8 register partials behind a uniform switch); the four waves meet in an LDS
// image of S's slab, flushed with one atomic per nonzero entry.
#ifndef VAENPVC_SS_CHUNKS
#define VAENPVC_SS_CHUNKS 128
#endif
#ifndef VAENPVC_SS_FIF
#define VAENPVC_SS_FIF 4   // frames in flight per wave
#endif
template <int NPL, int NY>
__global__ void __launch_bounds__(256) k_split_segsum(const float* __restrict__ d, const int64_t* __restrict__ y, int N, int Kp, int F,
                                                      int fchunk, unsigned short* __restrict__ dst, float* __restrict__ parts) {
  __shared__ float red[NY * 8 * 64];   // [k][j][lane]: conflict-free for the owners, read transposed by the flush
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.x * 512 + lane * 8;
  const int fq = fchunk >> 2;
  const int fb = blockIdx.y * fchunk + wave * fq, fe = min(F, fb + fq);
  float acc[NY][8];
#pragma unroll
  for (int k = 0; k < NY; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
  if (k0 < Kp) {
    const bool full = k0 + 8 <= N;
    constexpr int FIF = VAENPVC_SS_FIF;
    for (int f0 = fb; f0 < fe; f0 += FIF) {
      float v[FIF][8];
      int yk[FIF];
#pragma unroll
      for (int u = 0; u < FIF; ++u) {
        const int f = f0 + u < fe ? f0 + u : fe - 1;
        const float* p = d + (int64_t)f * N + k0;
        if (full) {
          packed4 p0 = *reinterpret_cast<const packed4*>(p), p1 = *reinterpret_cast<const packed4*>(p + 4);
          v[u][0] = p0.x; v[u][1] = p0.y; v[u][2] = p0.z; v[u][3] = p0.w; v[u][4] = p1.x; v[u][5] = p1.y; v[u][6] = p1.z; v[u][7] = p1.w;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[u][j] = k0 + j < N ? p[j] : 0.f;
        }
        const int64_t yy = y[f];
        yk[u] = (int)(yy < 0 ? 0 : (yy >= NY ? NY - 1 : yy));   // ids are clamped (vaenpvc_validate_ids reports them)
      }
#pragma unroll
      for (int u = 0; u < FIF; ++u) {
        if (f0 + u >= fe) break;
        const int ku = __builtin_amdgcn_readfirstlane(yk[u]);
#pragma unroll
        for (int k = 0; k < NY; ++k)
          if (ku == k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k][j] += v[u][j];
          }
        u32x4 pk[NPL];
        pack8<NPL>(v[u], pk);
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          st_nt<VAENPVC_NT_B>(reinterpret_cast<u32x4*>(dst + ((int64_t)p * F + (f0 + u)) * Kp + k0), pk[p]);
      }
    }
  }
  // the four waves meet in LDS one after the other (plain read-modify-write: LDS atomics cost ~150 cycles per wave op)
#pragma unroll 1
  for (int w4 = 0; w4 < 4; ++w4) {
    if (wave == w4) {
#pragma unroll
      for (int k = 0; k < NY; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float* r = &red[(k * 8 + j) * 64 + lane];
          *r = w4 == 0 ? acc[k][j] : *r + acc[k][j];
        }
    }
    __syncthreads();
  }
  // (plain stores of the chunk's partial image: device-scope atomics on one address serialise at the memory side)
  float* part = parts + (int64_t)blockIdx.y * NY * N;
  for (int i = threadIdx.x; i < NY * 512; i += 256) {
    const int k = i >> 9, c = i & 511, n = blockIdx.x * 512 + c;
    if (n < N) part[k * N + n] = red[(k * 8 + (c & 7)) * 64 + (c >> 3)];
  }
}
// out[i] = sum over c of part[c][i] (fixed order: repeatable), 64 columns per workgroup, sixteen waves deal the parts (all of a
// wave's loads in flight together: with four waves the 32-deep load chain per wave cost more than the bytes)
__global__ void __launch_bounds__(1024) k_sum_parts(const float* __restrict__ part, int nparts, int n, float* __restrict__ out) {
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
    out[i] = t;
  }
}
// parts: scratch of (frame chunks) x NY x N floats; returns the number of chunks (<= max(1, min(F/64, VAENPVC_SS_CHUNKS)))
template <int NPL, int NY>
inline int launch_split_segsum(const float* d, const int64_t* y, int N, int Kp, int64_t F, unsigned short* dst, float* parts, hipStream_t s) {
  const int ch = cmax(1, cmin_(cdiv((int)F, 64), VAENPVC_SS_CHUNKS));
  const int fc = 4 * cdiv(cdiv((int)F, ch), 4);
  const int nch = cdiv((int)F, fc);
  hipLaunchKernelGGL((k_split_segsum<NPL, NY>), dim3((unsigned)cdiv(Kp, 512), (unsigned)nch), dim3(256), 0, s, d, y, N, Kp, (int)F, fc, dst, parts);
  return nch;
}
// The per-speaker column sums alone, from the bf16 operand PLANES of d(h) (round 5: the input-gradient kernel of decoder layer 0 writes the
// planes itself, gfx950_fconv_r.h POUT, and no fp32 d(h) exists): value = sum of the element's terms (two planes: the fp32 value to 16 - 17
// mantissa bits, rounded to nearest -- the sums of 3 000+ such values per speaker move by < 1e-6 of their magnitude scale).  Same grid,
// ownership and chunk partials as k_split_segsum; read-only, eight frames in flight per wave.
template <int NPL, int NY>
__global__ void __launch_bounds__(256) k_segsum_planes(const unsigned short* __restrict__ pl, int64_t plane, const int64_t* __restrict__ y,
                                                       int N, int Kp, int F, int fchunk, float* __restrict__ parts) {
  __shared__ float red[NY * 8 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int k0 = blockIdx.x * 512 + lane * 8;
  const int fq = fchunk >> 2;
  const int fb = blockIdx.y * fchunk + wave * fq, fe = min(F, fb + fq);
  float acc[NY][8];
#pragma unroll
  for (int k = 0; k < NY; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
  if (k0 < N) {
    constexpr int FIF = 8;
    for (int f0 = fb; f0 < fe; f0 += FIF) {
      u32x4 v[FIF][NPL];
      int yk[FIF];
#pragma unroll
      for (int u = 0; u < FIF; ++u) {
        const int f = f0 + u < fe ? f0 + u : fe - 1;
#pragma unroll
        for (int p = 0; p < NPL; ++p) v[u][p] = *reinterpret_cast<const u32x4*>(pl + p * plane + (int64_t)f * Kp + k0);
        const int64_t yy = y[f];
        yk[u] = (int)(yy < 0 ? 0 : (yy >= NY ? NY - 1 : yy));   // ids are clamped (vaenpvc_validate_ids reports them)
      }
#pragma unroll
      for (int u = 0; u < FIF; ++u) {
        if (f0 + u >= fe) break;
        float x[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float lo = 0.f, hi = 0.f;
#pragma unroll
          for (int p = NPL - 1; p >= 0; --p) {   // smallest term first
            lo += __uint_as_float(v[u][p][q] << 16);
            hi += __uint_as_float(v[u][p][q] & 0xffff0000u);
          }
          x[2 * q] = lo;
          x[2 * q + 1] = hi;
        }
        const int ku = __builtin_amdgcn_readfirstlane(yk[u]);
#pragma unroll
        for (int k = 0; k < NY; ++k)
          if (ku == k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k][j] += x[j];
          }
      }
    }
  }
#pragma unroll 1
  for (int w4 = 0; w4 < 4; ++w4) {
    if (wave == w4) {
#pragma unroll
      for (int k = 0; k < NY; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float* r = &red[(k * 8 + j) * 64 + lane];
          *r = w4 == 0 ? acc[k][j] : *r + acc[k][j];
        }
    }
    __syncthreads();
  }
  float* part = parts + (int64_t)blockIdx.y * NY * N;
  for (int i = threadIdx.x; i < NY * 512; i += 256) {
    const int k = i >> 9, c = i & 511, n = blockIdx.x * 512 + c;
    if (n < N) part[k * N + n] = red[(k * 8 + (c & 7)) * 64 + (c >> 3)];
  }
}
template <int NPL, int NY>
inline int launch_segsum_planes(const unsigned short* pl, int64_t plane, const int64_t* y, int N, int Kp, int64_t F, float* parts, hipStream_t s) {
  const int ch = cmax(1, cmin_(cdiv((int)F, 64), VAENPVC_SS_CHUNKS));
  const int fc = 4 * cdiv(cdiv((int)F, ch), 4);
  const int nch = cdiv((int)F, fc);
  hipLaunchKernelGGL((k_segsum_planes<NPL, NY>), dim3((unsigned)cdiv(N, 512), (unsigned)nch), dim3(256), 0, s, pl, plane, y, N, Kp, (int)F, fc, parts);
  return nch;
}
inline void launch_sum_parts(const float* parts, int nch, int n, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)cdiv(n, 64)), dim3(1024), 0, s, parts, nch, n, out);
}

// sampler + per-frame KL (generic k_reparam: util/layers.py:152-156,170-183 with mu2 = lv2 = 0) for z = 128, one WAVE per frame (a lane
// owns two neighbouring elements), writing z as fp32 AND as the bf16 planes the merge GEMM reads ([NPL][F][128]: the split pass
// over z goes away).  eps: injected draw, or (draw) drawn here with Philox and stored to eps_out for the backward pass.
template <int NPL>
__global__ void __launch_bounds__(256) k_reparam_planes(const float* __restrict__ zmu, const float* __restrict__ zlv, const float* __restrict__ eps,
                                                        float* __restrict__ z, float* __restrict__ kl_f, unsigned short* __restrict__ pl, int F,
                                                        PhiloxKey key, int draw, float* __restrict__ eps_out) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (draw) key = philox_resolve(key);
  const int64_t plane = (int64_t)F * 128;
  for (int f = blockIdx.x * 4 + wv; f < F; f += gridDim.x * 4) {
    const int64_t i = (int64_t)f * 128 + 2 * lane;
    const f32x2 mu = *reinterpret_cast<const f32x2*>(zmu + i), lv = *reinterpret_cast<const f32x2*>(zlv + i);
    f32x2 e = {0.f, 0.f};
    if (draw) {
      e[0] = philox_normal(key, (uint64_t)i);
      e[1] = philox_normal(key, (uint64_t)i + 1);
      *reinterpret_cast<f32x2*>(eps_out + i) = e;
    } else if (eps) {
      struct __attribute__((packed, aligned(4))) p2 { float x, y; };   // (a caller's pointer: only float alignment is promised)
      const p2 t = *reinterpret_cast<const p2*>(eps + i);
      e[0] = t.x;
      e[1] = t.y;
    }
    f32x2 zz;
    float kl = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float v = expf(lv[j]);
      zz[j] = (draw || eps) ? mu[j] + e[j] * sqrtf(v) : mu[j];
      kl += 0.5f * ((0.f - lv[j]) + (v + mu[j] * mu[j]) / (1.0f + EPSILON) - 1.0f);
    }
    *reinterpret_cast<f32x2*>(z + i) = zz;
    unsigned pk[NPL];
    split_pair<NPL>(zz[0], zz[1], pk);
#pragma unroll
    for (int p = 0; p < NPL; ++p) *reinterpret_cast<unsigned*>(pl + p * plane + i) = pk[p];
    kl = wave_sum(kl);
    if (lane == 0) kl_f[f] = kl;
  }
}

// autodiff of the sampler + KL (k_reparam_bwd_colsum, gfx950_elem.h: util/layers.py:152-156,170-183) whose two results leave as the
// bf16 planes [dz_mu | dz_lv] ([NPL][F][256]) the two head GEMMs read -- no fp32 copies, no split pass -- with four frames in
// flight per thread; the two head-bias gradients as per-workgroup parts added by k_colsum_part2.
template <int NPL>
__global__ void __launch_bounds__(256) k_reparam_bwd_planes(const float* __restrict__ dz, const float* __restrict__ zmu,
                                                            const float* __restrict__ zlv, const float* __restrict__ eps,
                                                            unsigned short* __restrict__ pl, float* __restrict__ part,
                                                            int F, int fchunk, float invF) {
  __shared__ float sm[2][128];
  const int k = threadIdx.x & 127, ph = threadIdx.x >> 7;
  const int fb = blockIdx.x * fchunk, fe = min(F, fb + fchunk);
  const int64_t plane = (int64_t)F * 256;
  float smu = 0.f, slv = 0.f;
  for (int f0 = fb + ph; f0 < fe; f0 += 8) {
    float mu[4], lv[4], g[4], e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = f0 + 2 * u < fe ? f0 + 2 * u : fb;
      const int64_t i = (int64_t)f * 128 + k;
      mu[u] = zmu[i];
      lv[u] = zlv[i];
      g[u] = dz[i];
      e[u] = eps[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = f0 + 2 * u;
      if (f >= fe) break;
      const float v = expf(lv[u]);
      const float a = g[u] + mu[u] / (1.0f + EPSILON) * invF;
      const float b = g[u] * (0.5f * e[u] * sqrtf(v)) + 0.5f * (v / (1.0f + EPSILON) - 1.0f) * invF;
      smu += a;
      slv += b;
      unsigned ta[NPL], tb[NPL];
      split_n<NPL>(a, ta);
      split_n<NPL>(b, tb);
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        pl[p * plane + (int64_t)f * 256 + k] = (unsigned short)ta[p];
        pl[p * plane + (int64_t)f * 256 + 128 + k] = (unsigned short)tb[p];
      }
    }
  }
  if (ph == 1) {
    sm[0][k] = smu;
    sm[1][k] = slv;
  }
  __syncthreads();
  // (parts, not atomics: a thousand workgroups adding to the same 256 addresses serialise at the memory side -- the chain was the
  //  kernel's duration)
  if (ph == 0) {
    part[(int64_t)blockIdx.x * 256 + k] = smu + sm[0][k];
    part[(int64_t)blockIdx.x * 256 + 128 + k] = slv + sm[1][k];
  }
}
// out1[col] += sum_r part[r][col] (col < 128), out2[col - 128] += ... (col >= 128): one workgroup per column
__global__ void __launch_bounds__(256) k_colsum_part2(const float* __restrict__ part, int rows, float* __restrict__ out1, float* __restrict__ out2) {
  __shared__ float sm[4];
  const int col = blockIdx.x;
  float s = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) s += part[(int64_t)r * 256 + col];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(col < 128 ? out1 + col : out2 + (col - 128), (sm[0] + sm[1]) + (sm[2] + sm[3]));
}

static inline SplitArgs split_args(const float* src, int K, int Kp, int64_t rows, unsigned short* dst) {
  SplitArgs a;
  a.src = src;
  a.src2 = nullptr;
  a.k1 = K;
  a.ld1 = K;
  a.ld2 = 0;
  a.st = a.gamma = a.beta = nullptr;
  a.lndiv = 1;
  a.K = K;
  a.Kp = Kp;
  a.rows = rows;
  a.dst = dst;
  return a;
}

// weight matrix B[n][k] = fn(n, k) (zero outside the logical N x K) -> planes [NPL][Np][Kp]; job of k_pack_multi
template <class Fn, int NPL>
struct PackPlanesJob {
  Fn fn;
  unsigned short* dst;
  int Np, Kp;
  int count;  // Np * Kp
  __device__ void run(int i) const {
    const int n = i / Kp, k = i - n * Kp;
    unsigned t[NPL];
    split_n<NPL>(fn(n, k), t);
#pragma unroll
    for (int p = 0; p < NPL; ++p) dst[(size_t)p * Np * Kp + i] = (unsigned short)t[p];
  }
};
template <int NPL, class Fn>
inline PackPlanesJob<Fn, NPL> planes_job(Fn fn, float* dst, int Np, int Kp) {
  return PackPlanesJob<Fn, NPL>{fn, reinterpret_cast<unsigned short*>(dst), Np, Kp, Np * Kp};
}

// the weight matrices of the dense-shaped layers (TF layouts: dense [in][out], conv [t][cin][cout])
struct WHeadsF {  // z = y4 [F,768] x B^T : B[n][k] = n < 128 ? Wmu[k][n] : Wlv[k][n - 128]
  const float *Wmu, *Wlv;
  __device__ float operator()(int n, int k) const {
    if (n >= 256 || k >= 768) return 0.f;
    return n < 128 ? Wmu[k * 128 + n] : Wlv[k * 128 + (n - 128)];
  }
};
struct WHeadsB {  // dy4 [F,768] = [dz_mu | dz_lv] [F,256] x B^T : B[n][k] = k < 128 ? Wmu[n][k] : Wlv[n][k - 128]
  const float *Wmu, *Wlv;
  __device__ float operator()(int n, int k) const {
    if (n >= 768 || k >= 256) return 0.f;
    return k < 128 ? Wmu[n * 128 + k] : Wlv[n * 128 + (k - 128)];
  }
};
struct WMergeF {  // h = z [F,128] x B^T : B[n][k] = Wz[k][n]
  const float* Wz;
  int M;
  __device__ float operator()(int n, int k) const { return (n < M && k < 128) ? Wz[(int64_t)k * M + n] : 0.f; }
};
struct WMergeB {  // dz [F,128] = dh [F,M] x B^T : B[n][k] = Wz[n][k]
  const float* Wz;
  int M;
  __device__ float operator()(int n, int k) const { return (n < 128 && k < M) ? Wz[(int64_t)n * M + k] : 0.f; }
};
// encoder layer 4 as a dense layer: column (o, j) = o*3 + j, row (c, h) = c*7 + h, tap t = h - 3j + 3
struct WEnc4F {  // a4 [F,768] = y3 [F,896] x B^T : B[n = (o,j)][k = (c,h)]
  const float* W;  // [7][128][256]
  __device__ float operator()(int n, int k) const {
    if (n >= 768 || k >= 896) return 0.f;
    const int o = n / 3, j = n - 3 * o, c = k / 7, h = k - 7 * c, t = h - 3 * j + 3;
    return (t >= 0 && t < 7) ? W[(t * 128 + c) * 256 + o] : 0.f;
  }
};
struct WEnc4B {  // dy3 [F,896] = da4 [F,768] x B^T : B[n = (c,h)][k = (o,j)]
  const float* W;
  __device__ float operator()(int n, int k) const {
    if (n >= 896 || k >= 768) return 0.f;
    const int o = k / 3, j = k - 3 * o, c = n / 7, h = n - 7 * c, t = h - 3 * j + 3;
    return (t >= 0 && t < 7) ? W[(t * 128 + c) * 256 + o] : 0.f;
  }
};
struct PackRepeat3 {  // bias of layer 4 per dense column (o, j): b[o]
  const float* b;
  __device__ float operator()(int i) const { return i < 768 ? b[i / 3] : 0.f; }
};

// ---------------------------------------------------------------- C = A B^T
// Workgroup = 128 x 128 tile, 4 waves as 2 x 2, wave tile 64 x 64 (2 x 2 MFMA tiles).  K runs in chunks of 64:
// the chunk of both operands is prefetched from HBM / L2 into registers while the previous one is multiplied
// from LDS (rows padded to 144 bytes: the 16 rows a quarter-wave reads sit in different bank quads).
struct NtArgs {
  const unsigned short* A;  // planes [NPL][M][Kp]
  const unsigned short* B;  // planes [NPL][Np][Kp], Np = number of 128-column tiles * 128
  int64_t a_plane, b_plane;  // elements between planes
  int M, N, Kp;
  float* C;   // [M][ldc]            (columns n <  split)
  float* C2;  // [M][ldc] or nullptr  (columns n >= split, stored at n - split)
  int split, ldc;
  const float* bias;      // [N] or nullptr
  const float* rowbias;   // [nrb][ldrb] or nullptr: row idx[m] is added to output row m
  const int64_t* idx;
  int nrb, ldrb;
  int lep;                // 1: result tile through LDS, 16-byte stores (set by the launcher: Runtime::nt_lep, env VAENPVC_NT_LEP)
};
constexpr int NT_BM = 128, NT_BN = 128;
constexpr int NT_MAXRB = 16;   // row-bias table rows the epilogue keeps in LDS (speakers; VCC2016: 10)
// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch order), each XCD has its own L2.  The linear
// id is remapped (bijectively, any grid size) so that every XCD walks a CONTIGUOUS range of tiles; with the column
// tile as the fast index the workgroups resident on one XCD share their 128 rows of A, which are then fetched into
// that L2 once instead of once per column tile from HBM / Infinity Cache.
__device__ __forceinline__ int xcd_contiguous(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
// K chunk: 64 with up to two planes, 32 with three (LDS and prefetch registers stay at two workgroups per CU; the
// MFMAs between two barriers are the same 48 per wave either way)
#ifndef VAENPVC_NT_BK2
#define VAENPVC_NT_BK2 64
#endif
constexpr int nt_bk(int npl) { return npl >= 3 ? 32 : VAENPVC_NT_BK2; }
// the result tile goes through LDS on its way out (round 5, EVERY plane count): [128][NT_EP_PITCH] floats behind the speaker table
constexpr int NT_EP_PITCH = NT_BN + 4, NT_EP_OFF = NT_MAXRB * 128 * 4 + 128 * 4, NT_EP_LDS = NT_EP_OFF + NT_BM * NT_EP_PITCH * 4;   // 76 288 bytes
// One plane INCLUDED on purpose: its 36 KB of staging alone would fit three workgroups per CU and the 76 KB tile leaves two, and the
// tile still won the A/B (bf16 mode 4.46 -> 4.39 ms per step with VAENPVC_NT_LEP, DESIGN.md section 6 round 5: these sites are bound by
// their store instructions, not by the third workgroup).
constexpr bool nt_lep(int npl) { return npl >= 1; }
constexpr int nt_lds(int npl) {
  const int st = npl * (NT_BM + NT_BN) * (nt_bk(npl) * 2 + 16);  // 73 728 (2 planes) / 61 440 (3)
  return nt_lep(npl) && NT_EP_LDS > st ? NT_EP_LDS : st;
}

#ifndef VAENPVC_NT_CST
#define VAENPVC_NT_CST 0   // 1: non-temporal result stores (the result streams past the L2 that holds the weight tiles)
#endif
#ifndef VAENPVC_NT_WPS
#define VAENPVC_NT_WPS 2
#endif
// PERS (round 5): PERSISTENT workgroups.  The grid is two workgroups per CU; a workgroup walks the tiles of its XCD's contiguous range
// (workgroup b runs on XCD b % 8: the workgroups resident on one XCD at any time still work on adjacent tiles and share their rows of A
// in that L2), and the first K chunk of tile i + 1 is requested BEFORE the result stores of tile i are issued: the prologue round trip of
// a tile (35 us of launch + first loads + epilogue per one-tile workgroup) then runs under the previous tile's stores.
template <int NPL, bool PERS = false>
__global__ void __launch_bounds__(256, VAENPVC_NT_WPS) k_gemm_nt(NtArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NT_BK = nt_bk(NPL), NT_RS = NT_BK * 2 + 16, NQ = NT_BK / 16;   // NQ: 16-byte pieces per thread, plane, operand
  constexpr int APL = NT_BM * NT_RS, BPL = NT_BN * NT_RS;
  unsigned char* sA = smem;
  unsigned char* sB = smem + NPL * APL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = cdiv(a.N, NT_BN);
  // tiles of this workgroup: one (the launch has a workgroup per tile), or every (gridDim.x / 8)-th tile of the XCD's range
  int tile, tile_end, tile_step;
  if constexpr (PERS) {
    const int ntiles = cdiv(a.M, NT_BM) * ntn, x = blockIdx.x & 7, q = ntiles >> 3, r = ntiles & 7;
    const int lo = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    tile = lo + (blockIdx.x >> 3);
    tile_end = lo + q + (x < r ? 1 : 0);
    tile_step = gridDim.x >> 3;      // (the grid is a multiple of 8)
  } else {
    tile = xcd_contiguous(blockIdx.x, gridDim.x);
    tile_end = tile + 1;
    tile_step = 1;
  }
  if (tile >= tile_end) return;      // (uniform)
  int m0 = (tile / ntn) * NT_BM, n0 = (tile % ntn) * NT_BN;
  // staging: thread -> (row tid >> 1, half tid & 1 of the row's chunk) of both tiles: NQ pieces of 16 bytes per plane and operand
  const int srow = tid >> 1, shalf = tid & 1;
  const unsigned char *ga, *gb;
  auto point = [&](int m0_, int n0_) __attribute__((always_inline)) {
    const int arow = m0_ + srow < a.M ? m0_ + srow : a.M - 1;  // rows past the end: duplicates, never stored
    ga = reinterpret_cast<const unsigned char*>(a.A) + ((size_t)arow * a.Kp) * 2 + shalf * NT_BK;
    gb = reinterpret_cast<const unsigned char*>(a.B) + ((size_t)(n0_ + srow) * a.Kp) * 2 + shalf * NT_BK;
  };
  point(m0, n0);
  unsigned char* da = sA + srow * NT_RS + shalf * NT_BK;
  unsigned char* db = sB + srow * NT_RS + shalf * NT_BK;
  u32x4 ra[NPL][NQ], rb[NPL][NQ];
  auto gload = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        ra[p][q] = *reinterpret_cast<const u32x4*>(ga + (size_t)p * a.a_plane * 2 + kc * (NT_BK * 2) + q * 16);
        rb[p][q] = *reinterpret_cast<const u32x4*>(gb + (size_t)p * a.b_plane * 2 + kc * (NT_BK * 2) + q * 16);
      }
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        *reinterpret_cast<u32x4*>(da + p * APL + q * 16) = ra[p][q];
        *reinterpret_cast<u32x4*>(db + p * BPL + q * 16) = rb[p][q];
      }
  };
  f32x16 acc[2][2];
  const int aoff = (wm * 64 + l31) * NT_RS + lh * 16;
  const int boff = (wn * 64 + l31) * NT_RS + lh * 16;
  u32x4 fa[2][2][NPL], fb[2][2][NPL];
  auto loadF = [&](int set, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        fa[set][t][p] = *reinterpret_cast<const u32x4*>(sA + p * APL + aoff + t * 32 * NT_RS + ks * 32);
        fb[set][t][p] = *reinterpret_cast<const u32x4*>(sB + p * BPL + boff + t * 32 * NT_RS + ks * 32);
      }
  };
  auto mm = [&](int set) __attribute__((always_inline)) {
    using PR = Prod<NPL>;
    mfma_prio<1>(true);
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(fa[set][i][PR::A[t]], fb[set][j][PR::B[t]], acc[i][j]);
    mfma_prio<1>(false);
  };
#ifndef VAENPVC_NT_ABL
#define VAENPVC_NT_ABL 0   // developer ablation (wrong results): 1 no global loads after the first chunk, 2 no MFMAs, 4 no result stores, 8 no LDS traffic
#endif
  const int nch = a.Kp / NT_BK;
  gload(0);
  for (;;) {   // tiles of this workgroup
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  for (int kc = 0; kc < nch; ++kc) {
    if (!(VAENPVC_NT_ABL & 8)) lstore();  // chunk kc (prefetched)
    __syncthreads();
    if (kc + 1 < nch && !(VAENPVC_NT_ABL & 1)) gload(kc + 1);
    __builtin_amdgcn_sched_barrier(0);
    if (!(VAENPVC_NT_ABL & 8) || kc == 0) loadF(0, 0);
#pragma unroll
    for (int ks = 0; ks < NT_BK / 16; ++ks) {
      if (ks + 1 < NT_BK / 16 && (!(VAENPVC_NT_ABL & 8) || kc == 0)) loadF((ks + 1) & 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (!(VAENPVC_NT_ABL & 2)) mm(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();  // chunk consumed
  }
  const int tile_next = tile + tile_step;
  const int m0c = m0, n0c = n0;      // the tile whose results sit in the accumulators
  // persistent: the next tile's first chunk is requested ahead of this tile's result stores (the staging registers are free) -- but
  // BEHIND the epilogue's own loads (speaker table, bias): loads return in order, a load issued behind the request would wait for it
  auto prefetch_next = [&]() __attribute__((always_inline)) {
    if constexpr (PERS) {
      if (tile_next < tile_end) {
        m0 = (tile_next / ntn) * NT_BM;
        n0 = (tile_next % ntn) * NT_BN;
        point(m0, n0);
        gload(0);
      }
    }
  };
  if (VAENPVC_NT_ABL & 4) {
    prefetch_next();
    float sacc = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (sacc == 12345.678f) a.C[0] = sacc;
    if (PERS && tile_next < tile_end) { tile = tile_next; continue; }
    return;
  }
  // the speaker table rows of this tile (T[k][n0 .. n0+127], k < nrb <= NT_MAXRB) and the tile's 128 speaker ids go
  // through the (now free) LDS: the epilogue then reads LDS instead of issuing 2 x 64 dependent global loads per lane
  // (measured: 124 of the merge forward's 159 us were this epilogue)
  float* Ts = reinterpret_cast<float*>(smem);              // [nrb][128]
  int* ys = reinterpret_cast<int*>(smem + NT_MAXRB * 128 * 4);   // [128]
  const bool rb_lds = a.rowbias && a.nrb <= NT_MAXRB;   // uniform
  if (rb_lds) {
    for (int i = tid; i < a.nrb * 128; i += 256) {
      const int k = i >> 7, nl = i & 127, n = n0c + nl;
      Ts[i] = n < a.N ? a.rowbias[(int64_t)k * a.ldrb + n] : 0.f;
    }
    if (tid < 128) {
      const int m = m0c + tid;
      int64_t r = a.idx[m < a.M ? m : a.M - 1];
      ys[tid] = (int)(r < 0 ? 0 : (r >= a.nrb ? a.nrb - 1 : r));
    }
    __syncthreads();
  }
  // ---- epilogue through LDS (round 5): the 128 x 128 tile is parked as fp32 rows, then every thread stores 16-byte pieces of whole
  //      512-byte row runs -- 16 store instructions per lane instead of 64 four-byte ones.  The one-tile launches spent a quarter of their
  //      time issuing those (ablation without result stores: 176 -> 144 us, DESIGN.md section 6: a CU retires ~4 bytes per clock of
  //      4-byte-per-lane stores).  Bias and the speaker's table row are added on the way out (bias values fetched before the first store).
  if (nt_lep(NPL) && a.lep && (!a.rowbias || rb_lds) && (!a.C2 || (a.split % NT_BN) == 0)) {   // uniform
    float* ot = reinterpret_cast<float*>(smem + NT_EP_OFF);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          ot[(wm * 64 + i * 32 + acc_row(reg, lane)) * NT_EP_PITCH + wn * 64 + j * 32 + l31] = acc[i][j][reg];
    const int pc = tid & 31, r0 = tid >> 5, n = n0c + pc * 4;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
      for (int k = 0; k < 4; ++k) bv[k] = a.bias[min(n + k, a.N - 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
    prefetch_next();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    const bool second = a.C2 && n0c >= a.split;
    float* cb = (second ? a.C2 : a.C) + (n - (second ? a.split : 0));
    const bool whole = n + 3 < a.N;
    // (Round 6, measured and removed: pieces SHIFTED to each row's 16-byte boundary for rows that are not a multiple of 16 bytes -- the merge
    //  forward's 1 539-float rows -- with thread 31 storing the wrap-around floats singly: every 16-byte store aligned, and SLOWER, 94 -> 150 us:
    //  the shifted columns cost eight 4-byte LDS reads + a select chain per row where this loop has two 16-byte reads.)
#pragma unroll
    for (int i = 0; i < NT_BM / 8; ++i) {
      const int row = r0 + 8 * i, m = m0c + row;
      f32x4 v = *reinterpret_cast<const f32x4*>(ot + row * NT_EP_PITCH + pc * 4);
      v += bv;
      if (rb_lds) v += *reinterpret_cast<const f32x4*>(Ts + ys[row] * 128 + pc * 4);
      if (m >= a.M) continue;
      float* o = cb + (int64_t)m * a.ldc;
      if (whole) {
        st_nt<VAENPVC_NT_CST != 0>(reinterpret_cast<f32x4_a4*>(o), f32x4_a4{v[0], v[1], v[2], v[3]});
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (n + k < a.N) o[k] = v[k];
      }
    }
    if (!PERS || tile_next >= tile_end) break;
    tile = tile_next;
    __syncthreads();   // the tile and the table are read before the next tile's chunks overwrite the LDS
    continue;
  }
  // epilogue: lanes = 32 consecutive columns of a row -> 128-byte stores (both bias values first: a load issued between the
  // stores would order the later ones behind its round trip)
  float bb2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0c + wn * 64 + j * 32 + l31;
    bb2[j] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
  }
  __builtin_amdgcn_sched_barrier(0);
  prefetch_next();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0c + wn * 64 + j * 32 + l31;
    if (n >= a.N) continue;
    const float bb = bb2[j];
    float* cb = a.C;
    int nn = n;
    if (a.C2 && n >= a.split) {
      cb = a.C2;
      nn = n - a.split;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float rbv[16];
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        rbv[reg] = 0.f;
        if (rb_lds) {  // uniform
          const int ml = wm * 64 + i * 32 + acc_row(reg, lane);
          rbv[reg] = Ts[ys[ml] * 128 + (n - n0c)];
        } else if (a.rowbias) {  // uniform
          const int m = m0c + wm * 64 + i * 32 + acc_row(reg, lane);
          int64_t r = a.idx[m < a.M ? m : a.M - 1];
          r = r < 0 ? 0 : (r >= a.nrb ? a.nrb - 1 : r);
          rbv[reg] = a.rowbias[r * a.ldrb + n];
        }
      }
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0c + wm * 64 + i * 32 + acc_row(reg, lane);
        if (m < a.M) st_nt<VAENPVC_NT_CST != 0>(cb + (int64_t)m * a.ldc + nn, acc[i][j][reg] + bb + rbv[reg]);
      }
    }
  }
  if (!PERS || tile_next >= tile_end) break;
  tile = tile_next;
  if (rb_lds) __syncthreads();   // the epilogue's table reads are done before the next tile's chunks overwrite the LDS
  }  // tiles
}

inline bool gemm_nt_ring_serves(const NtArgs& a);                 // gfx950_ntring.h (round 6)
inline void launch_gemm_nt_ring(const NtArgs& a, hipStream_t s);
struct CgArgs;
struct CgSfArgs;
inline bool cgemm_sf_ring_serves(const CgArgs& a);
inline void launch_cgemm_sf_ring(const CgSfArgs& b, hipStream_t s);
struct CgLnbArgs;
inline bool cgemm_pf_ring_serves(const CgArgs& a);
inline int launch_cgemm_pf_ring_lnb(const CgArgs& a, const CgLnbArgs& lb, hipStream_t s);
inline bool gemm_nt_ar_serves(const NtArgs& a);
template <int NPL>
inline void launch_gemm_nt_ar(const NtArgs& a, hipStream_t s);
template <int NPL>
inline void launch_gemm_nt(const NtArgs& a_, hipStream_t s) {
  NtArgs a = a_;
  a.lep = rt().nt_lep ? 1 : 0;
  if constexpr (NPL == 2) {   // (two planes: the result tile is parked in the B buffer, which one plane does not fill)
    if (rt().nt_ar && gemm_nt_ar_serves(a) && (rt().nt_ar > 1 || a.M / NT_BM >= 192)) {   // short K, many rows: the A-resident kernel (Runtime::nt_ar)
      launch_gemm_nt_ar<NPL>(a, s);
      return;
    }
  }
  if constexpr (NPL == 2) {
    // K-long sites on the four-wave ring kernel (Runtime::nt_ring: 1 = from 128 tiles of 256 x 128 on, 2 = whenever served -- parity tests)
    if (rt().nt_ring && gemm_nt_ring_serves(a) && (rt().nt_ring > 1 || cdiv(a.M, 256) * cdiv(a.N, 128) >= 128)) {
      launch_gemm_nt_ring(a, s);
      return;
    }
  }
  const int ntiles = cdiv(a.M, NT_BM) * cdiv(a.N, NT_BN);
  constexpr int SLOTS = 256 * VAENPVC_NT_WPS;     // resident workgroups of the chip
  // (one plane: the one-tile kernel runs two workgroups per CU -- its LDS result tile, see nt_lep -- at 138 registers; the persistent form
  //  needs 194 and was only measured with two planes and more: not used)
  if (NPL >= 2 && rt().nt_persist > 0 && ntiles >= rt().nt_persist) {   // (Runtime::nt_persist, env VAENPVC_NT_PERSIST)
    rt().ensure_lds(reinterpret_cast<const void*>(&k_gemm_nt<NPL, true>), nt_lds(NPL));
    hipLaunchKernelGGL((k_gemm_nt<NPL, true>), dim3(SLOTS), dim3(256), nt_lds(NPL), s, a);
    return;
  }
  rt().ensure_lds(reinterpret_cast<const void*>(&k_gemm_nt<NPL, false>), nt_lds(NPL));
  hipLaunchKernelGGL((k_gemm_nt<NPL, false>), dim3((unsigned)ntiles), dim3(256), nt_lds(NPL), s, a);
}

// ---------------------------------------------------------------- C = A B^T with the A tile RESIDENT (short K; round 5)
// The merge layer's forward GEMM (model/vae.py:51-61: h = z Wz + T[y], M = frames, N = 1539, K = 128) is a STORE stream: 201 MB of
// results for 16 MB of operands and 17 GFLOP.  On the one-tile kernel above every 128 x 128 tile was its own workgroup: 3 328 prologues
// (first loads, table, barriers) for two K chunks each, the A rows fetched 13 times -- 131-146 us against ~45 us for the stores alone.
// Here ONE workgroup per CU owns 128 rows: their whole K run (Kp <= 128) stays in LDS, the column tiles of B stream through a second
// LDS buffer (next tile requested into registers before the current one is multiplied), and the workgroup writes 128 complete rows of
// C.  Full tiles issue their 64 result stores per lane UNCONDITIONALLY (M % 128 == 0 is required, the ragged last column tile takes
// the predicated path): a static store count lets the wait for the prefetched tile be a counted one, so the stores drain under the
// next tile's staging and MFMAs instead of being acknowledged first (loads and stores share one in-order counter).
constexpr int NTA_KP = 128, NTA_RS = NTA_KP * 2 + 16;
constexpr int nta_lds(int npl) { return npl * (NT_BM + NT_BN) * NTA_RS + NT_MAXRB * 128 * 4 + 128 * 4; }   // 148 480 bytes with two planes
static_assert(NT_BM * NT_EP_PITCH * 4 <= 2 * NT_BN * NTA_RS, "the result tile fits the B buffer (two planes)");
template <int NPL, bool RB>
__global__ void __launch_bounds__(256, 1) k_gemm_nt_ar(NtArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int APL = NT_BM * NTA_RS, BPL = NT_BN * NTA_RS, NQ = NTA_KP * 2 / 16 / 2;   // NQ: 16-byte pieces per thread, plane, operand (two threads per row)
  unsigned char* sA = smem;
  unsigned char* sB = smem + NPL * APL;
  float* Ts = reinterpret_cast<float*>(smem + NPL * (APL + BPL));   // [NT_MAXRB][128]
  int* ys = reinterpret_cast<int*>(Ts + NT_MAXRB * 128);            // [128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = cdiv(a.N, NT_BN), m0 = blockIdx.x * NT_BM;
  const int srow = tid >> 1, shalf = tid & 1;
  constexpr bool rb = RB;                 // a.rowbias != nullptr (nrb <= NT_MAXRB: checked by the launcher); a template parameter: a uniform
                                          // runtime flag left one branch per result store in the epilogue
  constexpr int NTB = NT_MAXRB * 128 / 256;   // table floats per thread
  {  // the A tile, once
    const unsigned char* ga = reinterpret_cast<const unsigned char*>(a.A) + ((size_t)(m0 + srow) * a.Kp) * 2 + shalf * (NTA_KP);
    u32x4 ra[NPL][NQ];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int q = 0; q < NQ; ++q) ra[p][q] = *reinterpret_cast<const u32x4*>(ga + (size_t)p * a.a_plane * 2 + q * 16);
    if (rb && tid < 128) {
      const int64_t r = a.idx[m0 + tid];
      ys[tid] = (int)(r < 0 ? 0 : (r >= a.nrb ? a.nrb - 1 : r));
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int q = 0; q < NQ; ++q) *reinterpret_cast<u32x4*>(sA + p * APL + srow * NTA_RS + shalf * NTA_KP + q * 16) = ra[p][q];
  }
  u32x4 rbq[NPL][NQ];
  float tb[NTB];
  f32x4 bv4;
  auto gload = [&](int t) __attribute__((always_inline)) {   // column tile t: its rows of B, its slice of the speaker table, its bias values
    const int n0 = t * NT_BN;
    const unsigned char* gb = reinterpret_cast<const unsigned char*>(a.B) + ((size_t)(n0 + srow) * a.Kp) * 2 + shalf * NTA_KP;
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int q = 0; q < NQ; ++q) rbq[p][q] = *reinterpret_cast<const u32x4*>(gb + (size_t)p * a.b_plane * 2 + q * 16);
    if (rb) {
#pragma unroll
      for (int u = 0; u < NTB; ++u) {
        const int i = tid + 256 * u, k = min(i >> 7, a.nrb - 1), n = min(n0 + (i & 127), a.N - 1);   // (clamped: entries past nrb / N are never read)
        tb[u] = a.rowbias[(int64_t)k * a.ldrb + n];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) bv4[k] = a.bias ? a.bias[min(n0 + (tid & 31) * 4 + k, a.N - 1)] : 0.f;
  };
  const int aoff = (wm * 64 + l31) * NTA_RS + lh * 16;
  const int boff = (wn * 64 + l31) * NTA_RS + lh * 16;
  f32x16 acc[2][2];
  u32x4 fa[2][2][NPL], fb[2][2][NPL];
  auto loadF = [&](int set, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        fa[set][t][p] = *reinterpret_cast<const u32x4*>(sA + p * APL + aoff + t * 32 * NTA_RS + ks * 32);
        fb[set][t][p] = *reinterpret_cast<const u32x4*>(sB + p * BPL + boff + t * 32 * NTA_RS + ks * 32);
      }
  };
  auto mm = [&](int set) __attribute__((always_inline)) {
    using PR = Prod<NPL>;
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(fa[set][i][PR::A[t]], fb[set][j][PR::B[t]], acc[i][j]);
  };
  constexpr int NKS = NTA_KP / 16;
  // One column tile.  The loop over the FULL tiles is straight-line code (the ragged last tile is peeled, the request for the next tile
  // is unconditional with a clamped index): only then can the wait for the prefetched registers at the top of the next tile be a counted
  // one that leaves this tile's 64 result stores in flight (a branch anywhere in between makes the compiler wait for everything).
  auto tile_body = [&](int t, auto full_) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_)::value;
    const int n0 = t * NT_BN;
    // the prefetched tile into LDS (the previous tile's fragment / table reads ended at the barrier that closed its iteration)
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int q = 0; q < NQ; ++q) *reinterpret_cast<u32x4*>(sB + p * BPL + srow * NTA_RS + shalf * NTA_KP + q * 16) = rbq[p][q];
    if constexpr (rb) {
#pragma unroll
      for (int u = 0; u < NTB; ++u) Ts[tid + 256 * u] = tb[u];
    }
    const f32x4 bv = bv4;
    __syncthreads();
    gload(min(t + 1, ntn - 1));     // (the last tile requests itself again: no branch)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
    loadF(0, 0);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      if (ks + 1 < NKS) loadF((ks + 1) & 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue through LDS (as k_gemm_nt): the tile parked as fp32 rows in the B buffer, then 16-byte pieces of whole 512-byte row runs
    __syncthreads();   // every wave is done with this tile's B fragments
    float* ot = reinterpret_cast<float*>(sB);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          ot[(wm * 64 + i * 32 + acc_row(reg, lane)) * NT_EP_PITCH + wn * 64 + j * 32 + l31] = acc[i][j][reg];
    __syncthreads();
    const int pc = tid & 31, r0 = tid >> 5, n = n0 + pc * 4;
    float* cb = a.C + (int64_t)m0 * a.ldc + n;
#pragma unroll
    for (int i = 0; i < NT_BM / 8; ++i) {
      const int row = r0 + 8 * i;
      f32x4 v = *reinterpret_cast<const f32x4*>(ot + row * NT_EP_PITCH + pc * 4);
      v += bv;
      if constexpr (rb) v += *reinterpret_cast<const f32x4*>(Ts + ys[row] * 128 + pc * 4);
      float* o = cb + (int64_t)row * a.ldc;
      if (FULL || n + 3 < a.N) {
        *reinterpret_cast<f32x4_a4*>(o) = f32x4_a4{v[0], v[1], v[2], v[3]};
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (n + k < a.N) o[k] = v[k];
      }
    }
    __syncthreads();   // tile and table reads are done before the next tile overwrites them
  };
  const int nfull = a.N / NT_BN;
  gload(0);
  // (the first tile is peeled as well: the loop is then ENTERED in the state its back edge leaves -- one tile request and 64 stores
  //  outstanding -- and the wait at its top stays counted; entered straight from gload(0) the compiler merges both states to "wait for all")
  tile_body(0, std::true_type{});
  for (int t = 1; t < nfull; ++t) tile_body(t, std::true_type{});
  if (nfull < ntn) tile_body(nfull, std::false_type{});
}
// served: short K resident in LDS, whole 128-row tiles, one output tensor, a table that fits
inline bool gemm_nt_ar_serves(const NtArgs& a) {
  return a.Kp == NTA_KP && a.M % NT_BM == 0 && !a.C2 && (!a.rowbias || (a.nrb <= NT_MAXRB && a.idx)) && a.N >= NT_BN;
}
template <int NPL>
inline void launch_gemm_nt_ar(const NtArgs& a, hipStream_t s) {
  if (a.rowbias) {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_gemm_nt_ar<NPL, true>), nta_lds(NPL));
    hipLaunchKernelGGL((k_gemm_nt_ar<NPL, true>), dim3((unsigned)(a.M / NT_BM)), dim3(256), nta_lds(NPL), s, a);
  } else {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_gemm_nt_ar<NPL, false>), nta_lds(NPL));
    hipLaunchKernelGGL((k_gemm_nt_ar<NPL, false>), dim3((unsigned)(a.M / NT_BM)), dim3(256), nta_lds(NPL), s, a);
  }
}

// (An LDS-DMA variant of this kernel -- global_load_lds_dwordx4 into a ring of 2-4 unpadded, XOR-swizzled stage
//  buffers, counted vmcnt waits, bare s_barrier -- was measured and dropped: with two workgroups per CU it ran within
//  +-3 % of the register-staged loop on all six sites (2 and 1 planes), with one workgroup per CU and 3-4 stages 30-50 %
//  slower.  The staging method is not the limit of the 128 x 128 two-barrier structure)

// ---------------------------------------------------------------- C += A^T B   (reduction over frames)
// Workgroup = 128 (m) x 256 (n) tile over the frames [z*fchunk, (z+1)*fchunk); 8 waves as 2 x 4, wave tile 64 x 64.
// Both operands are frame-major planes; a row-major [16 frames][columns] LDS tile feeds the MFMA through
// ds_read_b64_tr_b16 (gfx950_toep_bf16.h: k_toep_wgrad_bf16 is the same machine with a diagonal epilogue).
// Partial sums over frame chunks are combined with fp32 global atomics.
enum { TN_EPI_PLAIN = 0, TN_EPI_ENC4 = 1, TN_EPI_TRANS = 2 };
struct TnpArgs {
  const unsigned short* A;  // planes; row r (the reduction index) of A at view_off(av, r), M columns from there
  const unsigned short* B;  // planes; row r of B at view_off(bv, r), N columns
  int64_t a_plane, b_plane;
  RowView av, bv;
  int lda, ldb;   // readable elements per row (loads are clamped to the row: ragged last column tile)
  int xcd;               // 1: XCD-aware tile order (the tiles of a row chunk on one XCD)
  int tn4;               // 1: the four-wave kernel may be used (bit 16 of the backward mask)
  int M, N, F, fchunk;   // F = number of reduction rows
  float* C;   // PLAIN: C[m*ldc + n] (n < split) ; ENC4: the TF kernel tensor [7][128][256] ; TRANS: C[n*ldc + m]
  float* C2;  // PLAIN: columns n >= split at n - split
  int split, ldc;
};
constexpr int TP_KF = 16;
// Tile: 8 waves as 2 x 4, a wave owns TI x TJ MFMA tiles: <2,2> = 128 x 256 (the dense layers), <1,2> = 64 x 256,
// <1,1> = 64 x 128 (thin conv layers).  LDS row strides padded by 64 bytes (see WG_RSA / WG_RSB).
template <int NPL, int TI, int TJ>
struct TnTile {
  static constexpr int BM = 64 * TI, BN = 128 * TJ;
  // LDS row pitch: the transposing fragment reads take 8 bytes of four consecutive ROWS per lane group, so consecutive rows must fall on
  // different banks: pitch / 4 = 16 or 48 (mod 64).  + 64 bytes does that for the power-of-two tiles; 96 and 288 columns need no pad
  static constexpr int tn32_pitch(int w) { return ((w * 2 + 64) / 4) % 32 == 16 ? w * 2 + 64 : w * 2; }
  static constexpr int RSA = tn32_pitch(BM), RSB = tn32_pitch(BN);
  static_assert((RSA / 4) % 32 == 16 && (RSB / 4) % 32 == 16 && RSA % 16 == 0 && RSB % 16 == 0, "conflict-free row pitches");
  static constexpr int APL = TP_KF * RSA, BPL = TP_KF * RSB;
  static constexpr int BUF = NPL * (APL + BPL), LDS = 2 * BUF;
  static constexpr int APC = BM / 8, BPC = BN / 8;   // 16-byte pieces per row
};

// (second launch bound = waves per SIMD: 4 = two 8-wave workgroups per CU, i.e. at most 128 registers per lane)
template <int NPL, int EPI, int TI = 2, int TJ = 2>
__global__ void __launch_bounds__(512, NPL <= 2 ? 4 : 2) k_gemm_tn(TnpArgs a) {
  using T = TnTile<NPL, TI, TJ>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BUF = T::BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lh = lane >> 5, l31 = lane & 31;
  const int wr = wave >> 2, wc = wave & 3;
  // XCD-aware order: the tiles of one row chunk read the same rows of A and B -- they run on one XCD (one L2)
  const int ntm = cdiv(a.M, T::BM), ntt = ntm * cdiv(a.N, T::BN);
  const int wg = a.xcd ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int zc = wg / ntt, tl = wg - zc * ntt;
  const int m0 = (tl % ntm) * T::BM, n0 = (tl / ntm) * T::BN;
  const int fb = zc * a.fchunk, fe = min(a.F, fb + a.fchunk);
  // staging of one 16-row chunk: A 16 rows x APC pieces, B 16 rows x BPC pieces (the first 16*APC / 16*BPC threads)
  const bool a_thr = tid < TP_KF * T::APC, b_thr = tid < TP_KF * T::BPC;
  const int arow = (tid / T::APC) & 15, apc = tid % T::APC;
  const int brow = (tid / T::BPC) & 15, bpc = tid % T::BPC;
  // pieces past the row end (last column tile of a ragged N) are clamped to the row's last piece: finite data, masked
  // in the epilogue
  const int acol = cmin_(m0 * 2 + apc * 16, a.lda * 2 - 16);
  const int bcol = cmin_(n0 * 2 + bpc * 16, a.ldb * 2 - 16);
  const unsigned char* A8 = reinterpret_cast<const unsigned char*>(a.A);
  const unsigned char* B8 = reinterpret_cast<const unsigned char*>(a.B);
  u32x4 sta[NPL], stb[NPL];
  // Row addresses advance incrementally: 16 rows on = df whole frames + dq rows (uniform) + one conditional carry, no
  // division and no branch in the loop.  An iterator stops at its last valid row (rows past the end of the tensor
  // re-read it: finite data, and their products vanish because the A rows are zeroed in lstore).
  struct RowIt {   // (byte offsets within a plane fit 32 bits)
    unsigned off;  // row offset + column, bytes
    int q;         // row within the frame
  };
  auto row_init = [&](const RowView& v, int r, int colbytes) {
    RowIt it;
    const int f = r / v.R;
    it.q = r - f * v.R;
    it.off = (unsigned)(f * v.fs + v.x0 + it.q * v.step) * 2u + (unsigned)colbytes;
    return it;
  };
  const int adf = TP_KF / a.av.R, adq = TP_KF - adf * a.av.R, bdf = TP_KF / a.bv.R, bdq = TP_KF - bdf * a.bv.R;
  auto row_next = [&](const RowView& v, int df, int dq, RowIt& it) __attribute__((always_inline)) {
    int q = it.q + dq;
    const int carry = q >= v.R ? 1 : 0;
    q -= carry ? v.R : 0;
    it.off += (unsigned)(((df + carry) * v.fs + (q - it.q) * v.step) * 2);
    it.q = q;
  };
  RowIt ia = row_init(a.av, min(fb + arow, a.F - 1), acol), ib = row_init(a.bv, min(fb + brow, a.F - 1), bcol);
  auto gload = [&](int f0) __attribute__((always_inline)) {   // rows f0 + arow / f0 + brow: loads, then advances the iterators
    if (a_thr) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) sta[p] = *reinterpret_cast<const u32x4*>(A8 + (size_t)p * a.a_plane * 2 + ia.off);
      if (f0 + TP_KF + arow < a.F) row_next(a.av, adf, adq, ia);
    }
    if (b_thr) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) stb[p] = *reinterpret_cast<const u32x4*>(B8 + (size_t)p * a.b_plane * 2 + ib.off);
      if (f0 + TP_KF + brow < a.F) row_next(a.bv, bdf, bdq, ib);
    }
  };
  auto lstore = [&](int f0, int buf) __attribute__((always_inline)) {
    unsigned char* sA = smem + buf * BUF;
    unsigned char* sB = sA + NPL * T::APL;
    const u32x4 z = {0u, 0u, 0u, 0u};
    const bool tail = f0 + TP_KF > fe;  // uniform: rows past the chunk contribute zero (A rows zeroed)
    if (a_thr) {
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        *reinterpret_cast<u32x4*>(sA + p * T::APL + arow * T::RSA + apc * 16) = (tail && f0 + arow >= fe) ? z : sta[p];
    }
    if (b_thr) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(sB + p * T::BPL + brow * T::RSB + bpc * 16) = stb[p];
    }
  };
  // fragment addresses (transpose reads): lane -> (row (l&15)>>2 (+8*lh), column quad 4*(l&3) + 16*((l>>4)&1))
  const int trow = ((lane & 15) >> 2) + 8 * lh, tcol = 4 * (lane & 3) + 16 * ((lane >> 4) & 1);
  const int aoff = trow * T::RSA + (32 * TI * wr + tcol) * 2;
  const int boff = NPL * T::APL + trow * T::RSB + (32 * TJ * wc + tcol) * 2;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = zero16();
  u32x4 fa[TI][NPL], fbq[TJ][NPL];
  auto loadF = [&](int buf) __attribute__((always_inline)) {
    const unsigned char* sb = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int p = 0; p < NPL; ++p) fa[i][p] = tr_read8(sb + p * T::APL + aoff + i * 64, 4 * T::RSA);
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int p = 0; p < NPL; ++p) fbq[j][p] = tr_read8(sb + p * T::BPL + boff + j * 64, 4 * T::RSB);
  };
  auto mm = [&]() __attribute__((always_inline)) {
    using PR = Prod<NPL>;
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          // TRANS: the tile is accumulated transposed (rows = n, lanes = m) so that the atomics of a wave fall on
          // consecutive addresses of C[n*ldc + m]
          if constexpr (EPI == TN_EPI_TRANS) acc[i][j] = mfma_bf16(fbq[j][PR::B[t]], fa[i][PR::A[t]], acc[i][j]);
          else acc[i][j] = mfma_bf16(fa[i][PR::A[t]], fbq[j][PR::B[t]], acc[i][j]);
        }
  };
  if (fb < fe) {
    gload(fb);
    lstore(fb, 0);
  }
  __syncthreads();
  int nq = 0;
  for (int f0 = fb; f0 < fe; f0 += TP_KF, ++nq) {
    const bool more = f0 + TP_KF < fe;
    if (more) gload(f0 + TP_KF);
    __builtin_amdgcn_sched_barrier(0);
    loadF(nq & 1);
    __builtin_amdgcn_sched_barrier(0);
    mm();
    __builtin_amdgcn_sched_barrier(0);
    if (more) lstore(f0 + TP_KF, (nq + 1) & 1);
    __syncthreads();
  }
  // epilogue: acc[i][j][reg] = C[m0 + 32 TI wr + 32 i + acc_row(reg)][n0 + 32 TJ wc + 32 j + l31]  (TRANS: transposed)
  if constexpr (EPI == TN_EPI_TRANS) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int m = m0 + 32 * TI * wr + 32 * i + l31;
        if (m >= a.M) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int n = n0 + 32 * TJ * wc + 32 * j + acc_row(reg, lane);
          if (n < a.N) atomicAdd(a.C + (int64_t)n * a.ldc + m, acc[i][j][reg]);
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int n = n0 + 32 * TJ * wc + 32 * j + l31;
      if (n >= a.N) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + 32 * TI * wr + 32 * i + acc_row(reg, lane);
        if (m >= a.M) continue;
        const float v = acc[i][j][reg];
        if constexpr (EPI == TN_EPI_ENC4) {
          // m = (c, h) = c*7 + h, n = (o, j3) = o*3 + j3: dW[t][c][o] with t = h - 3*j3 + 3
          const int c = m / 7, h = m - 7 * c, o = n / 3, j3 = n - 3 * o, t = h - 3 * j3 + 3;
          if (t >= 0 && t < 7) atomicAdd(a.C + ((t * 128 + c) * 256 + o), v);
        } else {
          if (a.C2 && n >= a.split) atomicAdd(a.C2 + (int64_t)m * a.ldc + (n - a.split), v);
          else atomicAdd(a.C + (int64_t)m * a.ldc + n, v);
        }
      }
    }
}

template <int NPL, int EPI, int TI, int TJ, int WR, int WC>
inline void launch_gemm_tn32(TnpArgs a, int target_wgs, hipStream_t s);
template <int NPL, int EPI, int TI = 2, int TJ = 2>
inline void launch_gemm_tn(TnpArgs a, int target_wgs, hipStream_t s) {
  if constexpr (NPL <= 2 && EPI != TN_EPI_TRANS && TI == 2 && TJ == 2) {
    // the four-wave kernel where a row chunk has many 256 x 256 tiles (encoder layer 4: 4 x 3) and the rows are plain
    if (a.tn4 && a.F >= 4096 && a.av.R >= a.F && a.bv.R >= a.F && cdiv(a.M, 256) * cdiv(a.N, 256) >= rt().tn_w4_tiles)
      return launch_gemm_tn4<NPL, EPI>(a, s);
  }
  if constexpr (NPL <= 2) {
    // measured (32 768 frames): one plane -27 % over six sites; two planes -17..24 % on the sites with few tiles per row
    // chunk (merge, heads, encoder layer 3), equal on decoder layer 0, +10 % on encoder layer 4 (21 tiles: stays here)
    const bool many_tiles = cdiv(a.M, 64 * TI) * cdiv(a.N, 128 * TJ) > 8;
    if (!rt().tn_k16 && (NPL == 1 || !many_tiles)) return launch_gemm_tn32<NPL, EPI, TI, TJ, 2, 4>(a, target_wgs, s);
  }
  using T = TnTile<NPL, TI, TJ>;
  rt().ensure_lds(reinterpret_cast<const void*>(&k_gemm_tn<NPL, EPI, TI, TJ>), T::LDS);
  const int tiles = cdiv(a.M, T::BM) * cdiv(a.N, T::BN);
  const int zc = cmax(1, cmin_(cdiv(a.F, 64), cdiv(target_wgs, tiles)));
  a.fchunk = rup(cdiv(a.F, zc), TP_KF);
  dim3 grid((unsigned)(cdiv(a.M, T::BM) * cdiv(a.N, T::BN) * cdiv(a.F, a.fchunk)));
  hipLaunchKernelGGL((k_gemm_tn<NPL, EPI, TI, TJ>), grid, dim3(512), T::LDS, s, a);
}

// ---------------------------------------------------------------- C += A^T B, pipelined (up to two planes)
// The same tile and epilogues on the schedule of the Toeplitz weight gradient (gfx950_toep_bf16.h:
// k_toep_wgrad_bf16_k32), chosen after the same ablation (five sites, 820 us: prologue / epilogue / atomics 356, MFMAs
// 243, waiting for the one-chunk prefetch 177, LDS stores 83 -- all additive): 32-row chunks (two k-steps per barrier),
// two fragment sets with the transposed reads issued between the MFMAs, the staging registers refilled right after
// they were written to LDS, one 8-wave workgroup per CU and HALF as many workgroups (half the atomics).
constexpr int TP32_KF = 32;
// (round 5) the wave grid is a template parameter: WR x WC waves of TI x TJ MFMA tiles each -- 2 x 4 waves (512 threads) as before, or the
// 3 x 3 grid of 1 x 3 tiles (576 threads) whose 96 x 288 tile fits decoder layer 0's weight gradient (M = 81, N = 9 taps x 32 = 288):
// on the 128 x 256 tile that site ran 2 column tiles of which 36 % of the MFMA work was useful (the kernel is MFMA-bound there).
template <int NPL, int TI, int TJ, int WR = 2, int WC = 4>
struct Tn32Tile {
  static constexpr int BM = 32 * TI * WR, BN = 32 * TJ * WC, NTHR = 64 * WR * WC;
  // LDS row pitch: the transposing fragment reads take 8 bytes of four consecutive ROWS per lane group, so consecutive rows must fall on
  // different banks: pitch / 4 = 16 or 48 (mod 64).  + 64 bytes does that for the power-of-two tiles; 96 and 288 columns need no pad
  static constexpr int tn32_pitch(int w) { return ((w * 2 + 64) / 4) % 32 == 16 ? w * 2 + 64 : w * 2; }
  static constexpr int RSA = tn32_pitch(BM), RSB = tn32_pitch(BN);
  static_assert((RSA / 4) % 32 == 16 && (RSB / 4) % 32 == 16 && RSA % 16 == 0 && RSB % 16 == 0, "conflict-free row pitches");
  static constexpr int APL = TP32_KF * RSA, BPL = TP32_KF * RSB;
  static constexpr int BUF = NPL * (APL + BPL), LDS = 2 * BUF;
  static constexpr int APC = BM / 8, BPC = BN / 8;   // 16-byte pieces per row
};
#define TN32_INTERLEAVE(N, MASK, CNT)                        \
  _Pragma("unroll") for (int _i = 0; _i < (N); ++_i) {       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);        \
    __builtin_amdgcn_sched_group_barrier((MASK), (CNT), 0);   \
  }

template <int NPL, int EPI, int TI = 2, int TJ = 2, int WR = 2, int WC = 4>
__global__ void __launch_bounds__((Tn32Tile<NPL, TI, TJ, WR, WC>::NTHR), (WR * WC <= 8 ? 2 : 1)) k_gemm_tn32(TnpArgs a) {
  static_assert(NPL <= 2, "three planes do not fit two 32-row buffers");
  using T = Tn32Tile<NPL, TI, TJ, WR, WC>;
  static_assert(TP32_KF * T::APC <= T::NTHR && 16 * T::BPC <= T::NTHR, "one staging piece of A, two of B per thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BUF = T::BUF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lh = lane >> 5, l31 = lane & 31;
  const int wr = wave / WC, wc = wave % WC;
  const int ntm = cdiv(a.M, T::BM), ntt = ntm * cdiv(a.N, T::BN);
  const int wg = a.xcd ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int zc = wg / ntt, tl = wg - zc * ntt;
  const int m0 = (tl % ntm) * T::BM, n0 = (tl / ntm) * T::BN;
  const int fb = zc * a.fchunk, fe = min(a.F, fb + a.fchunk);
  // staging of one 32-row chunk: A 32 rows x APC pieces (one per thread), B 32 rows x BPC pieces (rows brow and brow + 16)
  const bool a_thr = tid < TP32_KF * T::APC, b_thr = tid < 16 * T::BPC;
  const int arow = (tid / T::APC) & 31, apc = tid % T::APC;
  const int brow = (tid / T::BPC) & 15, bpc = tid % T::BPC;
  const int acol = cmin_(m0 * 2 + apc * 16, a.lda * 2 - 16);
  const int bcol = cmin_(n0 * 2 + bpc * 16, a.ldb * 2 - 16);
  const unsigned char* A8 = reinterpret_cast<const unsigned char*>(a.A);
  const unsigned char* B8 = reinterpret_cast<const unsigned char*>(a.B);
  u32x4 sta[NPL], stb[NPL][2];
  struct RowIt {
    unsigned off;
    int q;
  };
  auto row_init = [&](const RowView& v, int r, int colbytes) {
    RowIt it;
    const int f = r / v.R;
    it.q = r - f * v.R;
    it.off = (unsigned)(f * v.fs + v.x0 + it.q * v.step) * 2u + (unsigned)colbytes;
    return it;
  };
  const int adf = TP32_KF / a.av.R, adq = TP32_KF - adf * a.av.R, bdf = TP32_KF / a.bv.R, bdq = TP32_KF - bdf * a.bv.R;
  auto row_next = [&](const RowView& v, int df, int dq, RowIt& it) __attribute__((always_inline)) {
    int q = it.q + dq;
    const int carry = q >= v.R ? 1 : 0;
    q -= carry ? v.R : 0;
    it.off += (unsigned)(((df + carry) * v.fs + (q - it.q) * v.step) * 2);
    it.q = q;
  };
  RowIt ia = row_init(a.av, min(fb + arow, a.F - 1), acol);
  RowIt ib0 = row_init(a.bv, min(fb + brow, a.F - 1), bcol), ib1 = row_init(a.bv, min(fb + brow + 16, a.F - 1), bcol);
  auto gload = [&](int f0) __attribute__((always_inline)) {   // rows of chunk f0: loads, then advances the iterators
    if (a_thr) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) sta[p] = *reinterpret_cast<const u32x4*>(A8 + (size_t)p * a.a_plane * 2 + ia.off);
      if (f0 + TP32_KF + arow < a.F) row_next(a.av, adf, adq, ia);
    }
    if (b_thr) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        stb[p][0] = *reinterpret_cast<const u32x4*>(B8 + (size_t)p * a.b_plane * 2 + ib0.off);
        stb[p][1] = *reinterpret_cast<const u32x4*>(B8 + (size_t)p * a.b_plane * 2 + ib1.off);
      }
      if (f0 + TP32_KF + brow < a.F) row_next(a.bv, bdf, bdq, ib0);
      if (f0 + TP32_KF + brow + 16 < a.F) row_next(a.bv, bdf, bdq, ib1);
    }
  };
  auto lstore = [&](int f0, int buf) __attribute__((always_inline)) {
    unsigned char* sA = smem + buf * BUF;
    unsigned char* sB = sA + NPL * T::APL;
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (a_thr) {
      const bool zero = f0 + arow >= fe;   // rows past the chunk contribute zero (A rows zeroed)
#pragma unroll
      for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(sA + p * T::APL + arow * T::RSA + apc * 16) = zero ? z : sta[p];
    }
    if (b_thr) {
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        *reinterpret_cast<u32x4*>(sB + p * T::BPL + brow * T::RSB + bpc * 16) = stb[p][0];
        *reinterpret_cast<u32x4*>(sB + p * T::BPL + (brow + 16) * T::RSB + bpc * 16) = stb[p][1];
      }
    }
  };
  const int trow = ((lane & 15) >> 2) + 8 * lh, tcol = 4 * (lane & 3) + 16 * ((lane >> 4) & 1);
  const int aoff = trow * T::RSA + (32 * TI * wr + tcol) * 2;
  const int boff = NPL * T::APL + trow * T::RSB + (32 * TJ * wc + tcol) * 2;
  f32x16 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) acc[i][j] = zero16();
  u32x4 fa[2][TI][NPL], fbq[2][TJ][NPL];
  auto readF = [&](int set, int buf, int ks) __attribute__((always_inline)) {
    const unsigned char* sb = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int p = 0; p < NPL; ++p) fa[set][i][p] = tr_read8(sb + p * T::APL + ks * 16 * T::RSA + aoff + i * 64, 4 * T::RSA);
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int p = 0; p < NPL; ++p) fbq[set][j][p] = tr_read8(sb + p * T::BPL + ks * 16 * T::RSB + boff + j * 64, 4 * T::RSB);
  };
  // MFMAs of one k-step for the tiles of parity `half` (all tiles when the wave owns a single one and half == 0)
  constexpr int NTILE = TI * TJ, NHALF = NTILE >= 2 ? NTILE / 2 : 1;
  auto mma = [&](int set, int half) __attribute__((always_inline)) {
    using PR = Prod<NPL>;
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
          const int id = i * TJ + j;
          if (NTILE >= 2 ? ((id & 1) != half) : (half != 0)) continue;
          if constexpr (EPI == TN_EPI_TRANS) acc[i][j] = mfma_bf16(fbq[set][j][PR::B[t]], fa[set][i][PR::A[t]], acc[i][j]);
          else acc[i][j] = mfma_bf16(fa[set][i][PR::A[t]], fbq[set][j][PR::B[t]], acc[i][j]);
        }
  };
  constexpr int NPROD = Prod<NPL>::N, NRD = (TI + TJ) * NPL * 2;   // MFMAs per tile and k-step; LDS reads per fragment set
  if (fb < fe) {
    gload(fb);
    lstore(fb, 0);
    if (fb + TP32_KF < fe) gload(fb + TP32_KF);
  }
  __syncthreads();
  if (fb < fe) readF(0, 0, 0);
  int n = 0, f0 = fb;
  for (; f0 + TP32_KF < fe; f0 += TP32_KF, ++n) {   // iterations with a successor: branch-free regions
    const int buf = n & 1;
    // region A: fragment reads of k-step 1 under the MFMAs of k-step 0
    readF(1, buf, 1);
    mma(0, 0);
    mma(0, 1);
    TN32_INTERLEAVE(NTILE * NPROD, 0x100, cdiv(NRD, NTILE * NPROD));
    __builtin_amdgcn_sched_barrier(0);
    // region B: LDS stores of the next chunk under the first half of k-step 1
    lstore(f0 + TP32_KF, buf ^ 1);
    mma(1, 0);
    TN32_INTERLEAVE(NHALF * NPROD, 0x200, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (f0 + 2 * TP32_KF < fe) gload(f0 + 2 * TP32_KF);   // the staging registers go straight back to work
    __syncthreads();
    // region C: fragment reads of the next chunk's k-step 0 under the second half of k-step 1
    readF(0, buf ^ 1, 0);
    mma(1, 1);
    TN32_INTERLEAVE(NHALF * NPROD, 0x100, cdiv(NRD, NHALF * NPROD));
    __builtin_amdgcn_sched_barrier(0);
  }
  if (f0 < fe) {   // last chunk
    readF(1, n & 1, 1);
    mma(0, 0);
    mma(0, 1);
    mma(1, 0);
    mma(1, 1);
  }
  // epilogue: acc[i][j][reg] = C[m0 + 32 TI wr + 32 i + acc_row(reg)][n0 + 32 TJ wc + 32 j + l31]  (TRANS: transposed)
  if constexpr (EPI == TN_EPI_TRANS) {
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int m = m0 + 32 * TI * wr + 32 * i + l31;
        if (m >= a.M) continue;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int nn = n0 + 32 * TJ * wc + 32 * j + acc_row(reg, lane);
          if (nn < a.N) atomicAdd(a.C + (int64_t)nn * a.ldc + m, acc[i][j][reg]);
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int nn = n0 + 32 * TJ * wc + 32 * j + l31;
      if (nn >= a.N) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + 32 * TI * wr + 32 * i + acc_row(reg, lane);
        if (m >= a.M) continue;
        const float v = acc[i][j][reg];
        if constexpr (EPI == TN_EPI_ENC4) {
          const int c = m / 7, h = m - 7 * c, o = nn / 3, j3 = nn - 3 * o, t = h - 3 * j3 + 3;
          if (t >= 0 && t < 7) atomicAdd(a.C + ((t * 128 + c) * 256 + o), v);
        } else {
          if (a.C2 && nn >= a.split) atomicAdd(a.C2 + (int64_t)m * a.ldc + (nn - a.split), v);
          else atomicAdd(a.C + (int64_t)m * a.ldc + nn, v);
        }
      }
    }
}

template <int NPL, int EPI, int TI, int TJ, int WR = 2, int WC = 4>
inline void launch_gemm_tn32(TnpArgs a, int target_wgs, hipStream_t s) {
  using T = Tn32Tile<NPL, TI, TJ, WR, WC>;
  rt().ensure_lds(reinterpret_cast<const void*>(&k_gemm_tn32<NPL, EPI, TI, TJ, WR, WC>), T::LDS);
  const int tiles = cdiv(a.M, T::BM) * cdiv(a.N, T::BN);
  const int zc = cmax(1, cmin_(cdiv(a.F, 128), cdiv(cmax(target_wgs / 2, 1), tiles)));   // one workgroup per CU
  a.fchunk = rup(cdiv(a.F, zc), TP32_KF);
  dim3 grid((unsigned)(tiles * cdiv(a.F, a.fchunk)));
  hipLaunchKernelGGL((k_gemm_tn32<NPL, EPI, TI, TJ, WR, WC>), grid, dim3(T::NTHR), T::LDS, s, a);
}

// ---------------------------------------------------------------- C += A^T B on four waves (plain row views, up to two planes)
// The schedule of k_toep_wgrad_bf16_w4 (gfx950_toep_bf16.h) on a 256 x 256 tile: one wave per SIMD with a 128 x 128 wave
// tile (16 accumulator tiles = all 256 AGPRs; a fragment byte read from LDS feeds twice the MFMAs of the 64 x 64 wave
// tiles above), operands by LDS-DMA into a ring of four 16-row stages (rows unpadded, 16-byte pieces XOR-swizzled with
// the row so that the transpose reads stay conflict-free; requested four iterations ahead, counted vmcnt, one bare
// barrier per stage), one piece of side work -- a fragment read or a request -- behind every MFMA.  Rows are PLAIN
// (R = INT_MAX): a stage is a scalar step in both operands.  The requests run up to five stages past the last row without
// clamping: the plane buffers are sized for three planes (model.cpp), this kernel runs with at most two; rows >= F are
// cleared in LDS before they are multiplied.  Workgroup order: XCD = row chunk (all tiles of a chunk share one L2).
#ifndef VAENPVC_TN4_XCD
#define VAENPVC_TN4_XCD 1
#endif
constexpr int tn4_stage(int npl) { return npl * 2 * W4_APL; }
constexpr int tn4_lds(int npl) { return W4_NS * tn4_stage(npl) > 131072 ? W4_NS * tn4_stage(npl) : 131072; }   // ring (131 072 bytes at two planes) / the folded epilogue's four 32-KB tiles
template <int NPL, int EPI>
__global__ void __launch_bounds__(256) k_gemm_tn4(TnpArgs a) {
  static_assert(NPL <= 2 && EPI != TN_EPI_TRANS, "two planes; plain and folded epilogues");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using PR = Prod<NPL>;
  constexpr int STAGE = tn4_stage(NPL), BOFF = NPL * W4_APL;
  constexpr int NB = 2 * NPL, NDMA = 2 * NB, NT8 = 8 * NPL, NM = 16 * PR::N;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int ntm = cdiv(a.M, 256), ntt = ntm * cdiv(a.N, 256), nzc = gridDim.x / ntt;
  // every XCD walks a contiguous range of (chunk, tile) with the tile as the fast index: the tiles of a frame chunk -- they read the same
  // rows of A and B -- share one L2 (the old order, chunk = blockIdx % chunks, spread them over all eight whenever the number of chunks
  // was not a multiple of 8: encoder layer 4's weight gradient fetched 664 MB for 217 MB of operands)
  const int wgx = VAENPVC_TN4_XCD ? xcd_contiguous(blockIdx.x, gridDim.x) : -1;
  const int zc = VAENPVC_TN4_XCD ? wgx / ntt : blockIdx.x % nzc, tl = VAENPVC_TN4_XCD ? wgx - zc * ntt : blockIdx.x / nzc;
  const int m0 = (tl % ntm) * 256, n0 = (tl / ntm) * 256;
  const int fb = zc * a.fchunk, fe = min(a.F, fb + a.fchunk);
  if (fb >= fe) return;
  const int nst = (fe - fb + W4_KF - 1) / W4_KF;

  // ---- DMA: block k of this wave = (plane, row pair) of the stage; lane -> (row 2 fp + (lane >> 5), LDS piece lane & 31);
  //      the piece fetched is (lane & 31) ^ ((row & 3) << 2), clamped to the row (ragged last column tile: finite data,
  //      masked in the epilogue)
  const unsigned char* A8 = reinterpret_cast<const unsigned char*>(a.A);
  const unsigned char* B8 = reinterpret_cast<const unsigned char*>(a.B);
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;
  const size_t arow = (size_t)a.lda * 2, brow = (size_t)a.ldb * 2;   // bytes per row
  const unsigned char* src[NDMA];
  unsigned dst[NDMA];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const int bid = wave * NB + k, pl = bid >> 3, fp = bid & 7, fr = 2 * fp + lh;
    const int gpc = l31 ^ ((fr & 3) << 2);
    src[2 * k] = A8 + (size_t)pl * a.a_plane * 2 + (size_t)(fb + fr) * arow + cmin_(m0 * 2 + gpc * 16, a.lda * 2 - 16);
    src[2 * k + 1] = B8 + (size_t)pl * a.b_plane * 2 + (size_t)(fb + fr) * brow + cmin_(n0 * 2 + gpc * 16, a.ldb * 2 - 16);
    dst[2 * k] = pl * W4_APL + fp * 1024;
    dst[2 * k + 1] = BOFF + pl * W4_APL + fp * 1024;
  }
  auto dma = [&](int g, int s, int slot) __attribute__((always_inline)) {
    const size_t off = (size_t)s * (W4_KF * ((g & 1) ? brow : arow));
    lds_dma16(src[g] + off, lds0 + slot * STAGE + dst[g]);
  };
  auto clear_tail = [&](int slot, int nv) __attribute__((always_inline)) {
    unsigned char* sb = smem + slot * STAGE;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < 2 * NPL * W4_KF * 32; i += 256)   // [operand][plane][row][32 pieces]
      if (((i >> 5) & (W4_KF - 1)) >= nv) *reinterpret_cast<u32x4*>(sb + i * 16) = z;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };
  const int last_nv = (fe - fb) - (nst - 1) * W4_KF;

  const int trow = ((lane & 15) >> 2) + 8 * lh, tcol = 4 * (lane & 3) + 16 * ((lane >> 4) & 1), s3 = trow & 3;
  int a_off[4], b_off[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    a_off[r] = trow * W4_RS + (16 * wr + ((r ^ s3) << 2) + (tcol >> 3)) * 16 + (tcol & 7) * 2;
    b_off[r] = BOFF + trow * W4_RS + (16 * wc + ((r ^ s3) << 2) + (tcol >> 3)) * 16 + (tcol & 7) * 2;
  }
  f32x16 acc[4][4];
#pragma unroll
  for (int ri = 0; ri < 4; ++ri)
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) acc[ri][cj] = zero16();
  u32x4 fa[2][4][NPL], fq[2][4][NPL];
  auto rd = [&](int set, int slot, int j) __attribute__((always_inline)) {
    const unsigned char* sb = smem + slot * STAGE;
    if (j < 4 * NPL)
      fa[set][j / NPL][j % NPL] = tr_read8(sb + (j % NPL) * W4_APL + a_off[j / NPL], 4 * W4_RS);
    else if (j < NT8)
      fq[set][(j - 4 * NPL) / NPL][j % NPL] = tr_read8(sb + (j % NPL) * W4_APL + b_off[(j - 4 * NPL) / NPL], 4 * W4_RS);
  };
  auto mm = [&](int set, int m) __attribute__((always_inline)) {
    const int t = m >> 4, ri = (m >> 2) & 3, cj = m & 3;
    acc[ri][cj] = mfma_bf16(fa[set][ri][PR::A[t]], fq[set][cj][PR::B[t]], acc[ri][cj]);
  };
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int g = 0; g < NDMA; ++g) dma(g, s, s);
  wait_vmcnt<2 * NDMA>();
  __builtin_amdgcn_s_barrier();
  if (last_nv < W4_KF && nst <= 2) clear_tail(nst - 1, last_nv);
#pragma unroll
  for (int j = 0; j < NT8; ++j) rd(0, 0, j);
  auto body = [&](auto S_, int t) __attribute__((always_inline)) {
    constexpr int S = decltype(S_)::value;
#pragma unroll
    for (int g = 0; g < NDMA; ++g) {
      mm(S & 1, 3 * g);
      rd((S + 1) & 1, (S + 1) & 3, 2 * g);
      __builtin_amdgcn_sched_barrier(0);
      mm(S & 1, 3 * g + 1);
      rd((S + 1) & 1, (S + 1) & 3, 2 * g + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(S & 1, 3 * g + 2);
      dma(g, t + 4, S);
      __builtin_amdgcn_sched_barrier(0);
    }
    static_assert(2 * NDMA >= NT8 && 3 * NDMA <= NM, "slots cover the reads");
#pragma unroll
    for (int m = 3 * NDMA; m < NM; ++m) mm(S & 1, m);
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<2 * NDMA>();
    __builtin_amdgcn_s_barrier();
    if (last_nv < W4_KF && t + 2 == nst - 1) clear_tail((S + 2) & 3, last_nv);
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
  // ---- epilogue: acc[ri][cj][reg] = C[m0 + 128 wr + 32 ri + acc_row(reg)][n0 + 128 wc + 32 cj + l31]
  if constexpr (EPI == TN_EPI_ENC4) {
    // m = (c, h) = 7 c + h, n = (o, j3) = 3 o + j3 -> dW[t][c][o], t = h - 3 j3 + 3: up to three entries of the tile fold onto one
    // output, and along n the outputs of a tap lie three apart.  One atomic per tile entry (16.5 M per launch, five partly
    // filled lines per wave instruction) cost 80 us at the end of the launch, when every workgroup arrives together.  Here
    // a wave parks 64 rows of its tile in LDS (its 32 KB of the idle ring), folds them -- lanes along o: stride-3 reads,
    // conflict-free -- and issues one atomic per (t, c, o) it holds a part of: half as many, on 2-3 full lines per instruction.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* tile = reinterpret_cast<float*>(smem) + wave * (64 * 128);
    const int nw0 = n0 + 128 * wc, o_lo = nw0 / 3, n_o = cmin_(nw0 + 127, a.N - 1) / 3 - o_lo + 1;   // (at most 44 <= 64 lanes)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int mw0 = m0 + 128 * wr + 64 * half;
      wave_lds_sync();   // (the previous half has been read)
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int cj = 0; cj < 4; ++cj)
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) tile[(32 * r2 + acc_row(reg, lane)) * 128 + 32 * cj + l31] = acc[2 * half + r2][cj][reg];
      wave_lds_sync();
      if (mw0 < a.M && nw0 < a.N) {
        const int c_lo = mw0 / 7, c_hi = cmin_(mw0 + 63, a.M - 1) / 7;
        const int o = o_lo + lane;
        for (int c = c_lo; c <= c_hi; ++c)
#pragma unroll
          for (int tt = 0; tt < 7; ++tt) {
            float v = 0.f;
            bool any = false;
#pragma unroll
            for (int j3 = 0; j3 < 3; ++j3) {
              const int h = tt - 3 + 3 * j3;
              if (h < 0 || h >= 7) continue;
              const int m = 7 * c + h - mw0, n = 3 * o + j3 - nw0;     // uniform m
              if (m >= 0 && m < 64 && 7 * c + h < a.M && lane < n_o && n >= 0 && n < 128 && 3 * o + j3 < a.N) {
                v += tile[m * 128 + n];
                any = true;
              }
            }
            if (any) atomicAdd(a.C + ((tt * 128 + c) * 256 + o), v);
          }
      }
    }
    return;
  }
#pragma unroll
  for (int ri = 0; ri < 4; ++ri)
#pragma unroll
    for (int cj = 0; cj < 4; ++cj) {
      const int nn = n0 + 128 * wc + 32 * cj + l31;
      if (nn >= a.N) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + 128 * wr + 32 * ri + acc_row(reg, lane);
        if (m >= a.M) continue;
        const float v = acc[ri][cj][reg];
        if constexpr (EPI == TN_EPI_ENC4) {
          const int c = m / 7, h = m - 7 * c, o = nn / 3, j3 = nn - 3 * o, tt = h - 3 * j3 + 3;
          if (tt >= 0 && tt < 7) atomicAdd(a.C + ((tt * 128 + c) * 256 + o), v);
        } else {
          if (a.C2 && nn >= a.split) atomicAdd(a.C2 + (int64_t)m * a.ldc + (nn - a.split), v);
          else atomicAdd(a.C + (int64_t)m * a.ldc + nn, v);
        }
      }
    }
}
template <int NPL, int EPI>
inline void launch_gemm_tn4(TnpArgs a, hipStream_t s) {
  rt().ensure_lds(reinterpret_cast<const void*>(&k_gemm_tn4<NPL, EPI>), tn4_lds(NPL));
  const int tiles = cdiv(a.M, 256) * cdiv(a.N, 256);
  const int zc = cmax(1, cmin_(cdiv(a.F, 64), 256 / tiles));   // one workgroup per CU
  a.fchunk = rup(cdiv(a.F, zc), W4_KF);
  hipLaunchKernelGGL((k_gemm_tn4<NPL, EPI>), dim3((unsigned)(tiles * cdiv(a.F, a.fchunk))), dim3(256), tn4_lds(NPL), s, a);
}

// =====================================================================================================
// Conv layers on the same machine.  Activations / gradients as CHANNEL-LAST planes with zero halo rows,
//     X[p][f][pos][c]   (unsigned short [NPL][F][HP][CP], HP = HLO + H + HHI, CP = channels padded to 8),
// turn every 1-D conv of the network into a plain GEMM over an overlapping-row VIEW of that tensor: the K-run
// of output row (f, q) -- taps x channels -- is ONE contiguous stretch of X starting at
//     f*HP*CP + x0 + q*xstep          (xstep = stride*CP for a conv, CP for a phase of a transposed conv),
// so there is no im2col, no gather and no shifted copy: rows simply overlap in memory.
//   conv forward / conv_transpose input gradient (util/layers.py:56-64, model/vae.py:96-99 backward):
//       out[f][o][j] = sum_{t,c} X[f][S j + t][c] W[t][c][o]                      K = T*CP, xstep = S*CP
//   conv_transpose forward / conv input gradient: S phases r, p = S q + r - PAD, taps t = S d + r, d < ceil(T/S):
//       out[f][o][p] = sum_{d,c} X[f][q - d][c] W[S d + r][..]                     K = ceil(T/S)*CP, xstep = CP
//   weight gradients: C[(t,c)][o] += sum_{(f,j)} X[f][S j + t][c] G[f][j][o]  -- k_gemm_tn with the view as A.
// The weights are the MFMA "A" operand (accumulator rows = output channels) and the view the "B" operand
// (lanes = positions), so results land in the canonical [F][C][H] fp32 tensors along the position axis.

struct CgArgs {
  const unsigned short* W;   // weight planes [NPL][Mp][Kp]
  const unsigned short* X;   // activation planes [NPL][...]
  int64_t w_plane, x_plane;  // elements
  RowView xv;                // rows of the GEMM's N index n = f*R + q
  int Kp, M, N;              // M = GEMM rows (output channels, or phase * mdiv + channel), N = F * R
  // out[f*ofs + ch*om + pos], ch = m % mdiv, pos = q*oq + o0 + (m / mdiv)*o0s, stored when 0 <= pos < OH and ch < C
  // (transposed convs put their S output phases into M: one pass over the taps, S consecutive positions per row q)
  float* out;
  int mdiv, C, ofs, om, oq, o0, o0s, OH;
  const float* bias;         // [C] or nullptr
};

// Tile: 4 waves as WM x (4 / WM); a wave owns MT x 2 MFMA tiles.  <2,2>: 128 x 128 (>= 96 GEMM rows); <1,2>: 64 x 256;
// <1,1>: 32 x 256 (thin layers).  K chunks of 64 for the square tile with <= 2 planes, 32 otherwise.
template <int NPL, int WM, int MT>
struct CgTile {
  static constexpr int WN = 4 / WM, BM = WM * MT * 32, BN = WN * 64, ROWS = BM + BN;
  static constexpr int BK = (WM == 2 && MT == 2) ? nt_bk(NPL) : 32;
  static constexpr int RS = BK * 2 + 16;
  static constexpr int PIECES = ROWS * (BK * 2 / 16), PPT = cdiv(PIECES, 256);
  static constexpr int LDS = NPL * ROWS * RS;
};

template <int NPL, int WM, int MT>
__global__ void __launch_bounds__(256, 2) k_cgemm(CgArgs a) {
  using T = CgTile<NPL, WM, MT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PLB = T::ROWS * T::RS;   // bytes per plane of the LDS image: weight rows, then view rows
  constexpr int PPR = T::BK * 2 / 16;    // 16-byte pieces per row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave / T::WN, wn = wave % T::WN;
  const int ntm = cdiv(a.M, T::BM), wg = xcd_contiguous(blockIdx.x, gridDim.x);   // neighbours on one XCD: adjacent view rows
  const int n0 = (wg / ntm) * T::BN, m0 = (wg % ntm) * T::BM;
  // staging: piece id = tid + 256*i -> (row id / PPR, piece id % PPR)
  const unsigned char* gp[T::PPT];
  int gplane2[T::PPT];   // bytes between planes / 2 (fits an int: planes are < 4 GB apart in elements)
  int lofs[T::PPT];
#pragma unroll
  for (int i = 0; i < T::PPT; ++i) {
    int id = tid + 256 * i;
    id = id < T::PIECES ? id : T::PIECES - 1;
    const int row = id / PPR, pc = id - row * PPR;
    lofs[i] = row * T::RS + pc * 16;
    if (row < T::BM) {
      gp[i] = reinterpret_cast<const unsigned char*>(a.W) + ((size_t)(m0 + row) * a.Kp) * 2 + pc * 16;
      gplane2[i] = (int)a.w_plane;
    } else {
      int r = n0 + row - T::BM;
      r = r < a.N ? r : a.N - 1;   // rows past the end: duplicates, never stored
      gp[i] = reinterpret_cast<const unsigned char*>(a.X) + (size_t)view_off(a.xv, r) * 2 + pc * 16;
      gplane2[i] = (int)a.x_plane;
    }
  }
  u32x4 rg[NPL][T::PPT];
  auto gload = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int i = 0; i < T::PPT; ++i)
        rg[p][i] = *reinterpret_cast<const u32x4*>(gp[i] + (size_t)p * (size_t)gplane2[i] * 2 + kc * (T::BK * 2));
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int i = 0; i < T::PPT; ++i)
        if (T::PIECES % 256 == 0 || tid + 256 * i < T::PIECES) *reinterpret_cast<u32x4*>(smem + p * PLB + lofs[i]) = rg[p][i];
  };
  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  const int aoff = (wm * MT * 32 + l31) * T::RS + lh * 16;
  const int boff = (T::BM + wn * 64 + l31) * T::RS + lh * 16;
  u32x4 fa[2][MT][NPL], fb[2][2][NPL];
  auto loadF = [&](int set, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int p = 0; p < NPL; ++p) fa[set][t][p] = *reinterpret_cast<const u32x4*>(smem + p * PLB + aoff + t * 32 * T::RS + ks * 32);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < NPL; ++p) fb[set][t][p] = *reinterpret_cast<const u32x4*>(smem + p * PLB + boff + t * 32 * T::RS + ks * 32);
  };
  auto mm = [&](int set) __attribute__((always_inline)) {
    using PR = Prod<NPL>;
    mfma_prio<2>(true);
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(fa[set][i][PR::A[t]], fb[set][j][PR::B[t]], acc[i][j]);
    mfma_prio<2>(false);
  };
  const int nch = a.Kp / T::BK;
  gload(0);
  for (int kc = 0; kc < nch; ++kc) {
    lstore();
    __syncthreads();
    if (kc + 1 < nch) gload(kc + 1);
    __builtin_amdgcn_sched_barrier(0);
    loadF(0, 0);
#pragma unroll
    for (int ks = 0; ks < T::BK / 16; ++ks) {
      if (ks + 1 < T::BK / 16) loadF((ks + 1) & 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  // epilogue: accumulator rows = GEMM rows m, lanes = 32 consecutive (frame, position) rows.  The bias values of the rows a
  // lane stores are fetched BEFORE the first store (a load between stores orders the later ones behind its round trip)
  // (phase, channel) of a row: with mdiv a multiple of 32 the 32 rows of an MFMA tile share their phase -- ONE uniform division per
  // tile instead of one per accumulator register and lane (a runtime integer division is ~28 vector instructions: the 2 x MT x 16
  // of them, 3 400 instructions, took longer than the tile's MFMA loop; round 4)
  const bool m32 = (a.mdiv & 31) == 0;   // uniform
  int pimI[MT], chbI[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int mb = __builtin_amdgcn_readfirstlane(m0 + wm * MT * 32 + i * 32);
    pimI[i] = mb / a.mdiv;
    chbI[i] = mb - pimI[i] * a.mdiv;
  }
  auto row_of = [&](int i, int reg, int& pim, int& ch) __attribute__((always_inline)) {
    if (m32) {
      pim = pimI[i];
      ch = chbI[i] + acc_row(reg, lane);
    } else {
      const int m = m0 + wm * MT * 32 + i * 32 + acc_row(reg, lane);
      pim = m / a.mdiv;
      ch = m - pim * a.mdiv;
    }
  };
  float bvc[MT][16];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int m = m0 + wm * MT * 32 + i * 32 + acc_row(reg, lane);
      int pim, ch;
      row_of(i, reg, pim, ch);
      bvc[i][reg] = (a.bias && m < a.M && ch < a.C) ? a.bias[ch] : 0.f;
    }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + l31;
    if (n >= a.N) continue;
    const int f = n / a.xv.R, q = n - f * a.xv.R;
    float* ob = a.out + (int64_t)f * a.ofs;
    const int pbase = q * a.oq + a.o0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int m = m0 + wm * MT * 32 + i * 32 + acc_row(reg, lane);
        if (m >= a.M) continue;
        int pim, ch;
        row_of(i, reg, pim, ch);
        const int pos = pbase + pim * a.o0s;
        if (ch < a.C && pos >= 0 && pos < a.OH) ob[(int64_t)ch * a.om + pos] = acc[i][j][reg] + bvc[i][reg];
      }
  }
}

template <int NPL, int WM, int MT>
inline void launch_cgemm(const CgArgs& a, hipStream_t s) {
  using T = CgTile<NPL, WM, MT>;
  rt().ensure_lds(reinterpret_cast<const void*>(&k_cgemm<NPL, WM, MT>), T::LDS);
  dim3 grid((unsigned)(cdiv(a.N, T::BN) * cdiv(a.M, T::BM)));
  hipLaunchKernelGGL((k_cgemm<NPL, WM, MT>), grid, dim3(256), T::LDS, s, a);
}
// tile by the number of GEMM rows
template <int NPL>
inline void launch_cgemm_auto(const CgArgs& a, hipStream_t s) {
  // a row count that leaves a half-empty 128-row tile (192 rows: the input gradient of encoder layer 3) runs on
  // 64 x 256 tiles: three full tiles instead of two with a quarter of the MFMAs wasted (269 -> 246 us)
  const bool half_tile_tail = a.M % 128 > 0 && a.M % 128 <= 64;
  if (a.M <= 32) launch_cgemm<NPL, 1, 1>(a, s);
  else if (a.M <= 64 || half_tile_tail) launch_cgemm<NPL, 1, 2>(a, s);
  else launch_cgemm<NPL, 2, 2>(a, s);
}

// ---------------------------------------------------------------- P-type site on a tile that owns whole frames (round 5)
// Encoder layer 3's input gradient (conv k7 s3 backward: util/layers.py:56-64 autodiff) has M = 3 phases x 64 channels = 192
// GEMM rows and R = 8 view rows per frame.  On the tiles above it ran as three 64-row tiles per 256 columns: every view row was
// fetched three times, and the three phases of one (channel, row q) -- three CONSECUTIVE output positions -- left from three
// different workgroups as 4-byte stores at a 12-byte stride (246 us at 20 % matrix-pipe busy: the worst GEMM of the step).
// Here a workgroup owns ALL 192 rows of 128 columns = 16 WHOLE frames: a wave holds the 3 phases x 32 channels x 64 columns,
// the view rows are staged once, and the finished tile is re-ordered through LDS into the canonical [frame][channel][position]
// fp32 layout -- 16 frames x 4 864 bytes are ONE contiguous run of the output tensor, copied out as 16-byte pieces (19 per
// thread).  LDS frame pitch 1 224 floats: the 32 lanes of a half-wave (4 frames x 8 rows q, stride 3) hit 32 different banks.
template <int NPL>
struct CgPfTile {
  static constexpr int PH = 3, MDIV = 64, BM = PH * MDIV, BN = 128, ROWS = BM + BN, BK = 32, RS = BK * 2 + 16;
  static constexpr int R = 8, TF = BN / R, OH = 19, FOUT = MDIV * OH, FPITCH = FOUT + 8;     // floats per output frame / its LDS pitch
  static constexpr int PIECES = ROWS * (BK * 2 / 16), PPT = cdiv(PIECES, 256);
  static constexpr int LDS_GEMM = NPL * ROWS * RS, LDS_OUT = TF * FPITCH * 4, LDS = LDS_GEMM > LDS_OUT ? LDS_GEMM : LDS_OUT;
  static_assert(PIECES % 256 == 0 && (TF * FOUT / 4) % 256 == 0, "staging / copy-out are whole rounds");
};
// LNB (round 5): the tile's 16 frames ARE the gradient at encoder layer 2's activated output, so the LayerNorm + lrelu backward of that
// layer (autodiff of util/layers.py:32-44,149; the arithmetic of k_ln_bwd_fused, gfx950_elem.h) runs on the frames in LDS, one wave per frame:
// a.out receives d(pre-LN output of layer 2) instead of d(activated output), the pre-LN tensor is read here (requested before the tile
// is re-ordered, so the loads fly under the LDS traffic), and the per-channel sums for d gamma / d beta / d bias leave as one row of
// `part` per workgroup (second stage: k_ln_bwd_reduce).  The gradient never exists in HBM as d(activated output): one write and one
// read of the tensor and the separate pass (k_ln_bwd_fused<64, 19>, 85 us) are gone.
struct CgLnbArgs {
  const float* a2;      // [F][64][19] pre-LN output of the layer whose LayerNorm is differentiated
  const float* st;      // [F][2] its statistics
  const float* gamma;   // [64]
  const float* beta;
  float* part;          // [workgroups][3][64]: sum dn xhat | sum dn | sum du
};
template <int NPL, bool LNB = false>
__global__ void __launch_bounds__(256, 2) k_cgemm_pf(CgArgs a, CgLnbArgs lb) {
  using T = CgPfTile<NPL>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PLB = T::ROWS * T::RS, PPR = T::BK * 2 / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wc = wave >> 1, wn = wave & 1;     // channel half, column half
  const int n0 = xcd_contiguous(blockIdx.x, gridDim.x) * T::BN;
  const unsigned char* gp[T::PPT];
  int gplane2[T::PPT], lofs[T::PPT];
#pragma unroll
  for (int i = 0; i < T::PPT; ++i) {
    const int id = tid + 256 * i, row = id / PPR, pc = id - row * PPR;
    lofs[i] = row * T::RS + pc * 16;
    if (row < T::BM) {
      gp[i] = reinterpret_cast<const unsigned char*>(a.W) + ((size_t)row * a.Kp) * 2 + pc * 16;
      gplane2[i] = (int)a.w_plane;
    } else {
      int r = n0 + row - T::BM;
      r = r < a.N ? r : a.N - 1;   // rows past the end: duplicates, never stored
      gp[i] = reinterpret_cast<const unsigned char*>(a.X) + (size_t)view_off(a.xv, r) * 2 + pc * 16;
      gplane2[i] = (int)a.x_plane;
    }
  }
  u32x4 rg[NPL][T::PPT];
  auto gload = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int i = 0; i < T::PPT; ++i)
        rg[p][i] = *reinterpret_cast<const u32x4*>(gp[i] + (size_t)p * (size_t)gplane2[i] * 2 + kc * (T::BK * 2));
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int i = 0; i < T::PPT; ++i) *reinterpret_cast<u32x4*>(smem + p * PLB + lofs[i]) = rg[p][i];
  };
  f32x16 acc[T::PH][2];
#pragma unroll
  for (int i = 0; i < T::PH; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  const int aoff = (wc * 32 + l31) * T::RS + lh * 16;                    // + phase * MDIV rows
  const int boff = (T::BM + wn * 64 + l31) * T::RS + lh * 16;            // + t * 32 rows
  u32x4 fa[2][T::PH][NPL], fb[2][2][NPL];
  auto loadF = [&](int set, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < T::PH; ++t)
#pragma unroll
      for (int p = 0; p < NPL; ++p) fa[set][t][p] = *reinterpret_cast<const u32x4*>(smem + p * PLB + aoff + t * T::MDIV * T::RS + ks * 32);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < NPL; ++p) fb[set][t][p] = *reinterpret_cast<const u32x4*>(smem + p * PLB + boff + t * 32 * T::RS + ks * 32);
  };
  auto mm = [&](int set) __attribute__((always_inline)) {
    using PR = Prod<NPL>;
    mfma_prio<2>(true);
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int i = 0; i < T::PH; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(fa[set][i][PR::A[t]], fb[set][j][PR::B[t]], acc[i][j]);
    mfma_prio<2>(false);
  };
  const int nch = a.Kp / T::BK;
  gload(0);
  for (int kc = 0; kc < nch; ++kc) {
    lstore();
    __syncthreads();
    if (kc + 1 < nch) gload(kc + 1);
    __builtin_amdgcn_sched_barrier(0);
    loadF(0, 0);
#pragma unroll
    for (int ks = 0; ks < T::BK / 16; ++ks) {
      if (ks + 1 < T::BK / 16) loadF((ks + 1) & 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  // ---- epilogue: the tile through LDS as [frame][channel][position] fp32, then one contiguous run of the output tensor
  float* ot = reinterpret_cast<float*>(smem);
  const int f0 = n0 / T::R, nf = min(T::TF, a.N / T::R - f0);
  constexpr int P16 = T::FOUT / 4;     // 16-byte pieces per frame
  constexpr int PPL = cdiv(P16, 64), FPW = T::TF / 4;   // LNB: pieces per lane, frames per wave
  f32x4 av[LNB ? FPW : 1][LNB ? PPL : 1];
  float gm[LNB ? PPL : 1][4], bt[LNB ? PPL : 1][4], fmean[LNB ? FPW : 1], frstd[LNB ? FPW : 1];
  if constexpr (LNB) {
#pragma unroll
    for (int k = 0; k < FPW; ++k) {
      const int f = min(f0 + wave + 4 * k, f0 + nf - 1);
      fmean[k] = lb.st[2 * f];
      frstd[k] = lb.st[2 * f + 1];
#pragma unroll
      for (int u = 0; u < PPL; ++u) {
        const int pc = min(lane + 64 * u, P16 - 1);
        av[k][u] = *reinterpret_cast<const f32x4*>(lb.a2 + (int64_t)f * T::FOUT + pc * 4);
      }
    }
#pragma unroll
    for (int u = 0; u < PPL; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ch = (min(lane + 64 * u, P16 - 1) * 4 + k) / T::OH;
        gm[u][k] = lb.gamma[ch];
        bt[u][k] = lb.beta[ch];
      }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nl = wn * 64 + j * 32 + l31, fl = nl / T::R, q = nl - fl * T::R;
#pragma unroll
    for (int ph = 0; ph < T::PH; ++ph) {
      const int pos = q * a.oq + a.o0 + ph;
      if (pos < 0 || pos >= T::OH) continue;
      float* ob = ot + fl * T::FPITCH + pos;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) ob[(wc * 32 + acc_row(reg, lane)) * T::OH] = acc[ph][j][reg];
    }
  }
  __syncthreads();
  if constexpr (LNB) {
    constexpr float INVN = 1.0f / T::FOUT;
    float su[PPL][4], sw[PPL][4], sd[PPL][4];
#pragma unroll
    for (int u = 0; u < PPL; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) su[u][k] = sw[u][k] = sd[u][k] = 0.f;
#pragma unroll
    for (int kf = 0; kf < FPW; ++kf) {
      const int fl = wave + 4 * kf;
      if (fl >= nf) break;       // (uniform per wave)
      const float mean = fmean[kf], rstd = frstd[kf];
      float dn[PPL][4], xh[PPL][4];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int u = 0; u < PPL; ++u) {
        const int pc = lane + 64 * u;
        const bool ok = pc < P16;
        const f32x4 dy = *reinterpret_cast<const f32x4*>(ot + fl * T::FPITCH + (ok ? pc : 0) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xh[u][k] = (av[kf][u][k] - mean) * rstd;
          const float nn = xh[u][k] * gm[u][k] + bt[u][k];
          dn[u][k] = ok ? dy[k] * (nn >= 0.f ? 1.0f : LEAK) : 0.f;
          const float dx = dn[u][k] * gm[u][k];
          s1 += dx;
          s2 += dx * xh[u][k];
        }
      }
      s1 = wave_sum(s1) * INVN;
      s2 = wave_sum(s2) * INVN;
      float* og = a.out + (int64_t)(f0 + fl) * T::FOUT;
#pragma unroll
      for (int u = 0; u < PPL; ++u) {
        const int pc = lane + 64 * u;
        if (pc >= P16) continue;
        f32x4 d;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          d[k] = rstd * (dn[u][k] * gm[u][k] - s1 - xh[u][k] * s2);
          su[u][k] += dn[u][k] * xh[u][k];
          sw[u][k] += dn[u][k];
          sd[u][k] += d[k];
        }
        *reinterpret_cast<f32x4*>(og + pc * 4) = d;
      }
    }
    __syncthreads();    // every wave is done with the tile: the LDS now carries the per-element sums [wave][3][FOUT]
    float* ps = ot + wave * (3 * T::FOUT);
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
      const int pc = lane + 64 * u;
      if (pc >= P16) continue;
      *reinterpret_cast<f32x4*>(ps + pc * 4) = f32x4{su[u][0], su[u][1], su[u][2], su[u][3]};
      *reinterpret_cast<f32x4*>(ps + T::FOUT + pc * 4) = f32x4{sw[u][0], sw[u][1], sw[u][2], sw[u][3]};
      *reinterpret_cast<f32x4*>(ps + 2 * T::FOUT + pc * 4) = f32x4{sd[u][0], sd[u][1], sd[u][2], sd[u][3]};
    }
    __syncthreads();
    if (tid < 3 * T::MDIV) {
      const int which = tid / T::MDIV, c = tid - which * T::MDIV;
      float v = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4)
        for (int h = 0; h < T::OH; ++h) v += ot[w4 * (3 * T::FOUT) + which * T::FOUT + c * T::OH + h];
      lb.part[(int64_t)blockIdx.x * (3 * T::MDIV) + tid] = v;
    }
    return;
  }
  const u32x4* ot4 = reinterpret_cast<const u32x4*>(smem);
  u32x4* og = reinterpret_cast<u32x4*>(a.out + (int64_t)f0 * T::FOUT);
#pragma unroll
  for (int i = 0; i < T::TF * P16 / 256; ++i) {
    const int id = tid + 256 * i, fl = id / P16, pc = id - fl * P16;
    if (fl < nf) og[(int64_t)fl * P16 + pc] = ot4[fl * (T::FPITCH / 4) + pc];
  }
}
// serves the site? (geometry of CV_E3G; no bias: an input gradient)
inline bool cgemm_pf_serves(const CgArgs& a) {
  return a.M == 192 && a.mdiv == 64 && a.C == 64 && a.xv.R == 8 && a.OH == 19 && a.om == 19 && a.ofs == 64 * 19 && a.oq == 3 && a.o0s == 1 &&
         !a.bias && a.Kp % 32 == 0 && a.N % 8 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
}
template <int NPL>
inline void launch_cgemm_pf(const CgArgs& a, hipStream_t s) {
  using T = CgPfTile<NPL>;
  rt().ensure_lds(reinterpret_cast<const void*>(&k_cgemm_pf<NPL, false>), T::LDS);
  hipLaunchKernelGGL((k_cgemm_pf<NPL, false>), dim3((unsigned)cdiv(a.N, T::BN)), dim3(256), T::LDS, s, a, CgLnbArgs{});
}
// ... with the LayerNorm backward of the layer below in its epilogue; returns the number of rows written to lb.part
template <int NPL>
inline int launch_cgemm_pf_lnb(const CgArgs& a, const CgLnbArgs& lb, hipStream_t s) {
  using T = CgPfTile<NPL>;
  static_assert(4 * 3 * T::FOUT * 4 <= T::LDS, "the per-element sums of four waves fit the tile's LDS");
  const int nwg = cdiv(a.N, T::BN);
  rt().ensure_lds(reinterpret_cast<const void*>(&k_cgemm_pf<NPL, true>), T::LDS);
  hipLaunchKernelGGL((k_cgemm_pf<NPL, true>), dim3((unsigned)nwg), dim3(256), T::LDS, s, a, lb);
  return nwg;
}

// ---------------------------------------------------------------- S-type site on a tile that owns whole frames (round 5)
// Encoder layer 3 FORWARD (conv k7 s3: util/layers.py:56-64; M = 128 output channels, R = 7 rows per frame, K = 7 x 64).  The
// 128 x 128 tile takes 126 columns = 18 WHOLE frames (two columns idle), so the workgroup holds complete pre-LN frames a3[f][128][7]:
//   * the tile goes through LDS in the canonical [frame][channel][position] order (frame pitch 904 floats: the 32 lanes of a
//     half-wave -- consecutive (frame, row) columns -- fall on different banks) and leaves as 16-byte pieces of ONE contiguous
//     run of the tensor (the one-tile kernel above stored 28-byte runs, 4 bytes per lane);
//   * the layer's LayerNorm statistics (util/layers.py:32-44: mean, biased variance, two-pass) are taken from the frame in LDS,
//     one wave per frame, and the ACTIVATED frame lrelu(LN(a3)) leaves as the bf16 operand planes pl_y3[NPL][F][896] that
//     encoder layer 4's GEMMs read: the separate pass over the tensor (k_ln_stats_planes<896, 7>: 43 us, 100 MB read) is gone.
struct CgSfArgs {
  CgArgs g;                 // the GEMM (W, X, planes, view, Kp, N = F * 7); g.out = a3 [F][128][7], g.bias = conv bias [128]
  float* st;                // [F][2] LayerNorm statistics of a3 (mean, rstd)
  const float* gamma;       // [128]
  const float* beta;
  unsigned short* planes;   // [NPL][F][896] or nullptr
  int F;
};
template <int NPL>
struct CgSfTile {
  static constexpr int BM = 128, BN = 128, ROWS = BM + BN, BK = 64, RS = BK * 2 + 16;
  static constexpr int R = 7, TF = 18, NCOL = TF * R, OH = 7, FOUT = BM * OH, FPITCH = FOUT + 8;
  static constexpr int PIECES = ROWS * (BK * 2 / 16), PPT = cdiv(PIECES, 256);
  static constexpr int LDS_GEMM = NPL * ROWS * RS, LDS_OUT = TF * FPITCH * 4, LDS = LDS_GEMM > LDS_OUT ? LDS_GEMM : LDS_OUT;
  static constexpr int P8 = FOUT / 8, PPL = cdiv(P8, 64);   // 8-element pieces per frame, per lane
  static_assert(PIECES % 256 == 0 && FOUT % 8 == 0 && NCOL <= BN, "tile geometry");
};
template <int NPL>
__global__ void __launch_bounds__(256, 2) k_cgemm_sf(CgSfArgs b) {
  using T = CgSfTile<NPL>;
  const CgArgs& a = b.g;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PLB = T::ROWS * T::RS, PPR = T::BK * 2 / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_contiguous(blockIdx.x, gridDim.x), n0 = tile * T::NCOL, f0 = tile * T::TF;
  const unsigned char* gp[T::PPT];
  int gplane2[T::PPT], lofs[T::PPT];
#pragma unroll
  for (int i = 0; i < T::PPT; ++i) {
    const int id = tid + 256 * i, row = id / PPR, pc = id - row * PPR;
    lofs[i] = row * T::RS + pc * 16;
    if (row < T::BM) {
      gp[i] = reinterpret_cast<const unsigned char*>(a.W) + ((size_t)row * a.Kp) * 2 + pc * 16;
      gplane2[i] = (int)a.w_plane;
    } else {
      int nl = row - T::BM;
      nl = nl < T::NCOL ? nl : T::NCOL - 1;    // the two idle columns and rows past the end: duplicates, never stored
      int r = n0 + nl;
      r = r < a.N ? r : a.N - 1;
      gp[i] = reinterpret_cast<const unsigned char*>(a.X) + (size_t)view_off(a.xv, r) * 2 + pc * 16;
      gplane2[i] = (int)a.x_plane;
    }
  }
  u32x4 rg[NPL][T::PPT];
  auto gload = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int i = 0; i < T::PPT; ++i)
        rg[p][i] = *reinterpret_cast<const u32x4*>(gp[i] + (size_t)p * (size_t)gplane2[i] * 2 + kc * (T::BK * 2));
  };
  auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
      for (int i = 0; i < T::PPT; ++i) *reinterpret_cast<u32x4*>(smem + p * PLB + lofs[i]) = rg[p][i];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
  const int aoff = (wm * 64 + l31) * T::RS + lh * 16;
  const int boff = (T::BM + wn * 64 + l31) * T::RS + lh * 16;
  u32x4 fa[2][2][NPL], fb[2][2][NPL];
  auto loadF = [&](int set, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        fa[set][t][p] = *reinterpret_cast<const u32x4*>(smem + p * PLB + aoff + t * 32 * T::RS + ks * 32);
        fb[set][t][p] = *reinterpret_cast<const u32x4*>(smem + p * PLB + boff + t * 32 * T::RS + ks * 32);
      }
  };
  auto mm = [&](int set) __attribute__((always_inline)) {
    using PR = Prod<NPL>;
    mfma_prio<2>(true);
#pragma unroll
    for (int t = 0; t < PR::N; ++t)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma_bf16(fa[set][i][PR::A[t]], fb[set][j][PR::B[t]], acc[i][j]);
    mfma_prio<2>(false);
  };
  const int nch = a.Kp / T::BK;
  gload(0);
  for (int kc = 0; kc < nch; ++kc) {
    lstore();
    __syncthreads();
    if (kc + 1 < nch) gload(kc + 1);
    __builtin_amdgcn_sched_barrier(0);
    loadF(0, 0);
#pragma unroll
    for (int ks = 0; ks < T::BK / 16; ++ks) {
      if (ks + 1 < T::BK / 16) loadF((ks + 1) & 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  // ---- the tile as [frame][channel][position] fp32 in LDS (raw conv sums; the bias is added in the frame pass)
  float* ot = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int nl = wn * 64 + j * 32 + l31;
    if (nl >= T::NCOL) continue;
    const int fl = nl / T::R, q = nl - fl * T::R;
    float* ob = ot + fl * T::FPITCH + q;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) ob[(wm * 64 + i * 32 + acc_row(reg, lane)) * T::OH] = acc[i][j][reg];
  }
  // per-lane constants of the frame pass: lane owns the 8-element pieces lane, lane + 64 of a frame's 896 = 112 x 8 elements;
  // element e = (channel e / 7, position e % 7)
  float gm[T::PPL][8], bt[T::PPL][8], bs[T::PPL][8];
#pragma unroll
  for (int u = 0; u < T::PPL; ++u)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int pc = lane + 64 * u, e = (pc < T::P8 ? pc : 0) * 8 + k, ch = e / T::OH;
      gm[u][k] = b.gamma[ch];
      bt[u][k] = b.beta[ch];
      bs[u][k] = a.bias ? a.bias[ch] : 0.f;
    }
  __syncthreads();
  const int nf = min(T::TF, b.F - f0);
  constexpr float INVN = 1.0f / T::FOUT;
  for (int fl = wave; fl < nf; fl += 4) {
    const int f = f0 + fl;
    float v[T::PPL][8];
    float sm = 0.f;
#pragma unroll
    for (int u = 0; u < T::PPL; ++u) {
      const int pc = lane + 64 * u;
      const bool ok = pc < T::P8;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(ot + fl * T::FPITCH + (ok ? pc : 0) * 8);
      const f32x4 t1 = *reinterpret_cast<const f32x4*>(ot + fl * T::FPITCH + (ok ? pc : 0) * 8 + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[u][k] = ok ? t0[k] + bs[u][k] : 0.f;
        v[u][4 + k] = ok ? t1[k] + bs[u][4 + k] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) sm += v[u][k];
    }
    const float mean = wave_sum(sm) * INVN;
    float q2 = 0.f;
#pragma unroll
    for (int u = 0; u < T::PPL; ++u) {
      const bool ok = lane + 64 * u < T::P8;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float d = v[u][k] - mean;
        q2 += ok ? d * d : 0.f;
      }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q2) * INVN + LN_EPS);
    if (lane == 0) {
      b.st[2 * f] = mean;
      b.st[2 * f + 1] = rstd;
    }
    float* og = a.out + (int64_t)f * T::FOUT;
#pragma unroll
    for (int u = 0; u < T::PPL; ++u) {
      const int pc = lane + 64 * u;
      if (pc >= T::P8) continue;
      *reinterpret_cast<f32x4*>(og + pc * 8) = f32x4{v[u][0], v[u][1], v[u][2], v[u][3]};
      *reinterpret_cast<f32x4*>(og + pc * 8 + 4) = f32x4{v[u][4], v[u][5], v[u][6], v[u][7]};
      if (b.planes) {   // uniform
        float y8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y8[k] = lnact_v(v[u][k], mean, rstd, gm[u][k], bt[u][k]);
        u32x4 pk[NPL];
        pack8<NPL>(y8, pk);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(b.planes + ((int64_t)p * b.F + f) * T::FOUT + pc * 8) = pk[p];
      }
    }
  }
}
inline bool cgemm_sf_serves(const CgArgs& a) {   // geometry of CV_E3F
  return a.M == 128 && a.mdiv == 128 && a.C == 128 && a.xv.R == 7 && a.OH == 7 && a.om == 7 && a.ofs == 128 * 7 && a.oq == 1 && a.o0 == 0 &&
         a.Kp % 64 == 0 && a.N % 7 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
}
template <int NPL>
inline void launch_cgemm_sf(const CgSfArgs& b, hipStream_t s) {
  using T = CgSfTile<NPL>;
  rt().ensure_lds(reinterpret_cast<const void*>(&k_cgemm_sf<NPL>), T::LDS);
  hipLaunchKernelGGL((k_cgemm_sf<NPL>), dim3((unsigned)cdiv(b.F, T::TF)), dim3(256), T::LDS, s, b);
}

// (the producers of the channel-last planes live in gfx950_viewconv.h: k_cl_produce)

}  // namespace tuned
}  // namespace vaenpvc

// (A large-tile variant -- 8 waves, 256 x 256 or 128 x 512 per workgroup, 128 accumulator registers per lane, one
//  workgroup per CU -- was measured and dropped: enc4 forward 194 us against 157 us for the 128 x 128 kernel above,
//  merge forward 250 against 150.  With 8 MFMA tiles per wave there is no room left for a second fragment set, so a
//  wave's LDS reads and MFMAs no longer overlap, and one workgroup per CU leaves nothing to cover the barriers.)
