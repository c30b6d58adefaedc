// gfx950_toeplitz.h -- the last decoder layer (conv2d_transpose, k = 1025, s = 1, 8 -> 1
// channels, model/vae.py:96-99 with architecture-vae-vcc2016.json:16-18) as a dense
// Toeplitz GEMM on fp32 MFMA.  Every (p, j) pair is a valid tap (p - j + 512 in [0,1024]),
// so xh = y2 (F x 4104) . T (4104 x 513) with T[(c,j)][p] = W[p-j+512][c]: the Toeplitz
// operand is never materialised -- a fragment is ONE contiguous ds_read_b32 from the
// channel-major weight copy Wc[c][t] held in LDS.
//
//   forward : xh[f][p]     = b + sum_c sum_j Wc[c][p-j+512] * y2[f][c][j]   (p < 512 here;
//             y2 = lrelu(LN(a2)) is materialised once by the LN-statistics kernel of layer 2)
//   dgrad   : dy2[f][c][j] =     sum_p     Wc[c][p-j+512] * dxh[f][p]       (all j)
// 513 = 16*32 + 1: the MFMA part covers 512 columns; forward column p = 512 is a side
// kernel (k_toep_fwd_lastcol), dgrad column j = 512 is a wave reduction inside the kernel.
#pragma once
#include "gfx950_common.h"

namespace vaenpvc {
namespace tuned {

constexpr int TOEP_H = 513, TOEP_C = 8, TOEP_T = 1025;
constexpr int WROW = 1032;  // Wc row: 4 zero floats + 1025 taps + 3 zero floats
constexpr int WPRE = 4;

// ---------------------------------------------------------------- forward
// grid (ceil(F/32), NSPLIT); 256 threads; each wave owns 16/NSPLIT/4 column tiles.
// K = (channel c, j) is walked in 16 chunks (8 channels x 2 halves of 258 j).  Double buffered:
// while the MFMAs consume chunk i from LDS buffer i&1, the global loads of chunk i+1 are in
// flight into registers; they are written to the other buffer after the MFMA block, one
// __syncthreads per chunk.  The 1032-float weight row of the chunk's channel travels with it.
constexpr int TF_JC = 258;           // j-chunk (2 chunks cover 516 >= 513)
constexpr int TF_ASTR = TF_JC + 1;   // odd row stride -> conflict-free A gathers
constexpr int TF_BUF = 32 * TF_ASTR + WROW;   // floats per buffer: A chunk + weight row
constexpr int TF_LDS = 2 * TF_BUF * 4;
constexpr int TF_EPT = (32 * TF_JC + 255) / 256;  // staged A elements per thread
constexpr int TF_WPT = (WROW + 255) / 256;

template <int NBW>  // column tiles per wave (4 -> NSPLIT 1, 2 -> NSPLIT 2, 1 -> NSPLIT 4)
__global__ void __launch_bounds__(256) k_toep_fwd(const float* __restrict__ y2, const float* __restrict__ Wc,
                                                  const float* __restrict__ bias, float* __restrict__ xh, int F) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int f0 = blockIdx.x * 32;
  const int p0 = (blockIdx.y * 4 + wave) * NBW * 32;
  f32x16 acc[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) acc[nb] = zero16();
  float ra[TF_EPT], rw[TF_WPT];
  auto gload = [&](int chunk) {
    const int c = chunk >> 1, jc0 = (chunk & 1) * TF_JC;
#pragma unroll
    for (int k = 0; k < TF_EPT; ++k) {
      int e = tid + 256 * k;
      int fl = e / TF_JC, jj = e - fl * TF_JC;
      int j = jc0 + jj, f = f0 + fl;
      ra[k] = (e < 32 * TF_JC && j < TOEP_H && f < F) ? y2[(int64_t)f * (TOEP_C * TOEP_H) + c * TOEP_H + j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < TF_WPT; ++k) {
      int i = tid + 256 * k;
      rw[k] = i < WROW ? Wc[c * WROW + i] : 0.f;
    }
  };
  auto lstore = [&](int buf) {
    float* tA = lds + buf * TF_BUF;
    float* tW = tA + 32 * TF_ASTR;
#pragma unroll
    for (int k = 0; k < TF_EPT; ++k) {
      int e = tid + 256 * k;
      int fl = e / TF_JC, jj = e - fl * TF_JC;
      if (e < 32 * TF_JC) tA[fl * TF_ASTR + jj] = ra[k];
    }
#pragma unroll
    for (int k = 0; k < TF_WPT; ++k) {
      int i = tid + 256 * k;
      if (i < WROW) tW[i] = rw[k];
    }
  };
  gload(0);
  lstore(0);
  __syncthreads();
  for (int chunk = 0; chunk < 2 * TOEP_C; ++chunk) {
    if (chunk + 1 < 2 * TOEP_C) gload(chunk + 1);
    __builtin_amdgcn_sched_barrier(0);
    {
      const float* tA = lds + (chunk & 1) * TF_BUF;
      const float* tW = tA + 32 * TF_ASTR;
      const int jc0 = (chunk & 1) * TF_JC;
      const float* ap = tA + l31 * TF_ASTR + lh;
      // B index: Wc[c][p - j + 512], p = p0 + nb*32 + l31, j = jc0 + 2*s + lh
      const float* wp = tW + WPRE + 512 + p0 + l31 - lh - jc0;
#pragma unroll 3
      for (int s = 0; s < TF_JC / 2; ++s) {
        float av = ap[2 * s];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[nb] = mfma32(av, wp[nb * 32 - 2 * s], acc[nb]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (chunk + 1 < 2 * TOEP_C) lstore((chunk + 1) & 1);
    __syncthreads();
  }
  const float bb = bias[0];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      int f = f0 + acc_row(reg, lane);
      if (f < F) xh[(int64_t)f * TOEP_H + p0 + nb * 32 + l31] = acc[nb][reg] + bb;
    }
}

// forward column p = 512: xh[f][512] = b + sum_{c,j} Wc[c][1024-j] * y2[f][c][j]; one wave per frame
__global__ void __launch_bounds__(256) k_toep_fwd_lastcol(const float* __restrict__ y2, const float* __restrict__ Wc,
                                                          const float* __restrict__ bias, float* __restrict__ xh, int F) {
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= F) return;
  float s = 0.f;
  for (int c = 0; c < TOEP_C; ++c) {
    const float* row = y2 + (int64_t)f * (TOEP_C * TOEP_H) + c * TOEP_H;
    const float* w = Wc + c * WROW + WPRE + 1024;
    for (int j = lane; j < TOEP_H; j += 64) s += row[j] * w[-j];
  }
  s = wave_sum(s);
  if (lane == 0) xh[(int64_t)f * TOEP_H + 512] = s + bias[0];
}

// ---------------------------------------------------------------- input gradient
// grid (ceil(F/32), 8 channels); 256 threads; wave w owns column tiles j0 = (4w..4w+3)*32.
constexpr int TD_ASTR = 515;  // dxh row in LDS: 513 values + 1 zero (K padded to 514) ; odd stride
constexpr int TD_LDS = (32 * TD_ASTR + WROW) * 4;

__global__ void __launch_bounds__(256) k_toep_dgrad(const float* __restrict__ dxh, const float* __restrict__ Wc,
                                                    float* __restrict__ dy, int F) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tA = lds;
  float* tW = lds + 32 * TD_ASTR;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int f0 = blockIdx.x * 32, c = blockIdx.y;
  for (int i = tid; i < WROW; i += 256) tW[i] = Wc[c * WROW + i];
  {
    constexpr int BT = 8;
    for (int e0 = tid; e0 < 32 * TD_ASTR; e0 += 256 * BT) {
      float v[BT];
#pragma unroll
      for (int bb = 0; bb < BT; ++bb) {
        int e = e0 + 256 * bb;
        int fl = e / TD_ASTR, p = e - fl * TD_ASTR;
        v[bb] = (e < 32 * TD_ASTR && p < TOEP_H && f0 + fl < F) ? dxh[(int64_t)(f0 + fl) * TOEP_H + p] : 0.f;
      }
#pragma unroll
      for (int bb = 0; bb < BT; ++bb) {
        int e = e0 + 256 * bb;
        if (e < 32 * TD_ASTR) tA[e] = v[bb];
      }
    }
  }
  __syncthreads();
  f32x16 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = zero16();
  const int j0 = wave * 128;
  const float* ap = tA + l31 * TD_ASTR + lh;
  // B[k = p][n = j] = Wc[c][p - j + 512], p = 2*s + lh, j = j0 + nb*32 + l31
  const float* wp = tW + WPRE + 512 + lh - j0 - l31;
#pragma unroll 3
  for (int s = 0; s < 257; ++s) {
    float av = ap[2 * s];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[nb] = mfma32(av, wp[2 * s - nb * 32], acc[nb]);
  }
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      int f = f0 + acc_row(reg, lane);
      if (f < F) dy[(int64_t)f * (TOEP_C * TOEP_H) + c * TOEP_H + j0 + nb * 32 + l31] = acc[nb][reg];
    }
  // column j = 512: dy[f][c][512] = sum_p Wc[c][p] * dxh[f][p]; wave w reduces frames 8w..8w+7
  for (int i = 0; i < 8; ++i) {
    int fl = wave * 8 + i;
    float s = 0.f;
    for (int p = lane; p < TOEP_H; p += 64) s += tA[fl * TD_ASTR + p] * tW[WPRE + p];
    s = wave_sum(s);
    if (lane == 0 && f0 + fl < F) dy[(int64_t)(f0 + fl) * (TOEP_C * TOEP_H) + c * TOEP_H + 512] = s;
  }
}

}  // namespace tuned
}  // namespace vaenpvc
