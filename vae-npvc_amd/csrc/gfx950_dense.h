// gfx950_dense.h -- dense layers (encoder heads, speaker/latent merge) forward and
// input-gradient:  out[f][n] = bias[n] + sum_k A'[f][k] * B[k][n]
//
// One workgroup = 32*MB frames (MB MFMA row tiles) x up to 4*NBW column tiles.  A' is staged
// through LDS in K-chunks of KCH (coalesced HBM reads; LN+lrelu on load, or the
// [z | E[y]] / [dz_mu | dz_lv] concatenation), B (packed [KP][NP] by gfx950_prep) streams
// from L2 into the MFMA operand register (128 contiguous bytes per half-wave).  The
// accumulator layout has lanes along n, so results are stored straight from registers.
#pragma once
#include "gfx950_common.h"
#include "gfx950_convgemm.h"  // IN_* kinds

namespace vaenpvc {
namespace tuned {

struct DenseArgs {
  const float* in;     // [F][K]  (IN_CONCAT2: first half [F][K/2])
  const float* in2;    // IN_CONCAT2: second half rows
  const int64_t* idx;  // IN_CONCAT2: optional row gather of in2
  const float* st;     // IN_LN
  const float* gamma;
  const float* beta;
  const float* Bp;  // packed [KP][NP], zero padded
  const float* rowbias;  // optional [nrb][ldo] table: row idx[f] is added to output row f (merge: T = E Wy + biases)
  int nrb;               // rows of the table (idx is clamped to it)
  const float* bias;
  float* out;   // [F][ldo]            (columns n <  split)
  float* out2;  // [F][ldo] or nullptr  (columns n >= split, stored at n - split)
  int split;
  int ldo;
  int F;
};

template <int K_, int N_, int KCH_, int NBW_, int INKIND_, int LNDIV_, int MB_ = 2>
struct DenseCfg {
  static constexpr int K = K_, N = N_, KCH = KCH_, NBW = NBW_, INKIND = INKIND_, LNDIV = LNDIV_;
  // row tiles (of 32 frames) per workgroup: every weight fragment streamed from L2 feeds MB MFMAs (a CU
  // sustains only ~10 B/clk of L1 misses; at one 256-byte fragment per MFMA that caps the chip near half
  // of the fp32-MFMA rate)
  static constexpr int MB = MB_, ROWS = 32 * MB_;
  static constexpr int NP = rup(N, 32), NT = NP / 32;
  static constexpr int NCHUNK = cdiv(K, KCH);
  static constexpr int KP = NCHUNK * KCH;  // packed B rows (zero padded)
  static constexpr int ASTR = KCH + 1;     // odd -> conflict-free gathers (lane <-> frame)
  static constexpr int NSPLIT = cdiv(NT, 4 * NBW);
  static constexpr int LDS_BYTES = ROWS * ASTR * 4;
  static_assert(KCH % 64 == 0, "chunk = multiple of the prefetch depth and of the staging batch");
};

template <class C>
__global__ void __launch_bounds__(256) k_densegemm(DenseArgs a) {
  extern __shared__ __attribute__((aligned(16))) float tA[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int f0 = blockIdx.x * C::ROWS;
  const int nt0 = (blockIdx.y * 4 + wave) * C::NBW;  // first column tile of this wave
  f32x16 acc[C::MB][C::NBW];
#pragma unroll
  for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < C::NBW; ++nb) acc[mb][nb] = zero16();
  const float* ap = tA + l31 * C::ASTR + lh;
  constexpr int U = 8;
  // gridDim.z > 1 (small batches): the K chunks are dealt to gridDim.z workgroups per tile; the partial results are
  // added with atomics into an output the caller zeroed, biases by the first one
  const bool ksplit = gridDim.z > 1;
  for (int ch = blockIdx.z; ch < C::NCHUNK; ch += gridDim.z) {
    const int kc0 = ch * C::KCH;
    __syncthreads();
    if constexpr (C::INKIND != IN_LN && C::KCH == 256) {
      // row-wise staging: one wave per frame row, lane l <-> floats 4l .. 4l+3 of the 256-float chunk (one
      // 16-byte load per row and lane; rows are only 4-byte aligned).  The row index and the gathered
      // embedding row are wave-uniform; all loads of a wave's rows are issued before the LDS stores.
      constexpr int RPW = C::ROWS / 4;  // rows per wave
      struct __attribute__((packed, aligned(4))) p4 { float x, y, z, w; };
      const int wv = __builtin_amdgcn_readfirstlane(wave);
      p4 v[RPW];
      const int k = kc0 + 4 * lane;
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int f = f0 + wv + 4 * r;
        const int fc = f < a.F ? f : a.F - 1;
        const float* src;
        if constexpr (C::INKIND == IN_CONCAT2) {
          constexpr int HALF = C::K / 2;  // (K == KCH == 256: lanes 0..31 first half, 32..63 second half)
          const int64_t g = a.idx ? a.idx[fc] : (int64_t)fc;
          src = lane < 32 ? a.in + (int64_t)fc * HALF + 4 * lane : a.in2 + g * HALF + 4 * (lane - 32);
        } else {
          src = a.in + (int64_t)fc * C::K + (k + 3 < C::K ? k : 0);
        }
        v[r] = *reinterpret_cast<const p4*>(src);
      }
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const int fl = wv + 4 * r;
        const bool okf = f0 + fl < a.F;
        float x[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
        if constexpr (C::INKIND != IN_CONCAT2) {
          if (k + 3 >= C::K) {  // last chunk: tail of the row (re-read element-wise, clamped)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = (k + j < C::K && okf) ? a.in[(int64_t)(f0 + fl) * C::K + k + j] : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tA[fl * C::ASTR + 4 * lane + j] = okf ? x[j] : 0.f;
      }
    } else {
    constexpr int BT = 8;
    for (int e0 = tid; e0 < C::ROWS * C::KCH; e0 += 256 * BT) {
      float v[BT];
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        int e = e0 + 256 * b;
        int fl = e / C::KCH, kk = e - fl * C::KCH;
        int k = kc0 + kk, f = f0 + fl;
        v[b] = 0.f;
        if (k < C::K && f < a.F) {
          if constexpr (C::INKIND == IN_CONCAT2) {
            constexpr int HALF = C::K / 2;
            if (k < HALF) {
              v[b] = a.in[(int64_t)f * HALF + k];
            } else {
              int64_t g = a.idx ? a.idx[f] : (int64_t)f;
              v[b] = a.in2[g * HALF + (k - HALF)];
            }
          } else {
            v[b] = a.in[(int64_t)f * C::K + k];
          }
        }
      }
#pragma unroll
      for (int b = 0; b < BT; ++b) {
        int e = e0 + 256 * b;
        int fl = e / C::KCH, kk = e - fl * C::KCH;
        int k = kc0 + kk, f = f0 + fl;
        float x = v[b];
        if constexpr (C::INKIND == IN_LN) {
          if (k < C::K && f < a.F) {
            int c = k / C::LNDIV;
            x = lnact_v(x, a.st[2 * f], a.st[2 * f + 1], a.gamma[c], a.beta[c]);
          }
        }
        tA[fl * C::ASTR + kk] = x;
      }
    }
    }
    __syncthreads();
    if (nt0 < C::NT) {  // wave-uniform
      // B streams from L2 into two ping-pong register sets, one sub-chunk (U k-steps) ahead
      // (unconditional loads: column tiles beyond NT are clamped and never stored; the last
      //  prefetch of the last chunk over-reads into the zero padding / next packed matrix)
      int ncol[C::NBW];
#pragma unroll
      for (int nb = 0; nb < C::NBW; ++nb) ncol[nb] = (nt0 + nb < C::NT ? nt0 + nb : C::NT - 1) * 32 + l31;
      const float* bp = a.Bp + (kc0 + lh) * C::NP;
      constexpr int NSUB = C::KCH / (2 * U);
      auto loadB = [&](float (&b)[U][C::NBW], int c8) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int nb = 0; nb < C::NBW; ++nb) b[u][nb] = bp[(c8 * U + u) * 2 * C::NP + ncol[nb]];
      };
      auto compute = [&](const float (&b)[U][C::NBW], int c8) {
        float av[U][C::MB];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int mb = 0; mb < C::MB; ++mb) av[u][mb] = ap[mb * 32 * C::ASTR + (c8 * U + u) * 2];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < C::NBW; ++nb) acc[mb][nb] = mfma32(av[u][mb], b[u][nb], acc[mb][nb]);
      };
      float b0[U][C::NBW], b1[U][C::NBW];
      loadB(b0, 0);
#pragma unroll
      for (int c8 = 0; c8 < NSUB; c8 += 2) {
        if (c8 + 1 < NSUB) loadB(b1, c8 + 1);
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch AHEAD of the MFMAs it overlaps
        compute(b0, c8);
        __builtin_amdgcn_sched_barrier(0);
        if (c8 + 1 < NSUB) {
          if (c8 + 2 < NSUB) loadB(b0, c8 + 2);
          __builtin_amdgcn_sched_barrier(0);
          compute(b1, c8 + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  // row offsets into the per-row bias table (merge: the speaker's row of T), fetched once for all column tiles
  const bool first = blockIdx.z == 0;
  int rbo[C::MB][16];
  if (a.rowbias && first) {  // uniform
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int f = f0 + mb * 32 + acc_row(reg, lane);
        int64_t r = a.idx[f < a.F ? f : a.F - 1];
        r = r < 0 ? 0 : (r >= a.nrb ? a.nrb - 1 : r);
        rbo[mb][reg] = (int)r * a.ldo;
      }
  }
#pragma unroll
  for (int nb = 0; nb < C::NBW; ++nb) {
    int n = (nt0 + nb) * 32 + l31;
    if (nt0 + nb < C::NT && n < C::N) {
      float bb = (a.bias && first) ? a.bias[n] : 0.f;
#pragma unroll
      for (int mb = 0; mb < C::MB; ++mb) {
        float rb[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) rb[reg] = (a.rowbias && first) ? a.rowbias[rbo[mb][reg] + n] : 0.f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          int f = f0 + mb * 32 + acc_row(reg, lane);
          if (f < a.F) {
            float* o = (a.out2 && n >= a.split) ? a.out2 + (int64_t)f * a.ldo + (n - a.split) : a.out + (int64_t)f * a.ldo + n;
            const float v = acc[mb][nb][reg] + bb + ((a.out2 && n >= a.split) ? 0.f : rb[reg]);
            if (ksplit) atomicAdd(o, v);
            else *o = v;
          }
        }
      }
    }
  }
}

// ksplit > 1: the caller has zeroed the output(s)
template <class C>
inline void launch_densegemm(const DenseArgs& a, hipStream_t s, int ksplit = 1) {
  rt().ensure_lds(reinterpret_cast<const void*>(&k_densegemm<C>), C::LDS_BYTES);
  dim3 grid((unsigned)cdiv(a.F, C::ROWS), (unsigned)C::NSPLIT, (unsigned)cmax(1, cmin_(ksplit, C::NCHUNK)));
  hipLaunchKernelGGL(k_densegemm<C>, grid, dim3(256), C::LDS_BYTES, s, a);
}

}  // namespace tuned
}  // namespace vaenpvc
