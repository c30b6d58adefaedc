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
// 513 = 16*32 + 1: the MFMA part covers 512 columns; forward column p = 512 is accumulated
// inside k_toep_fwd by the workgroups with blockIdx.y == 0, dgrad column j = 512 is a wave reduction
// inside the kernel.
// These exact-fp32 kernels serve batches below 8192 frames (and VAENPVC_TOEP=f32); larger batches
// use the bf16 matrix cores with a 3-term operand split, gfx950_toep_bf16.h.
#pragma once
#include "gfx950_common.h"

namespace vaenpvc {
namespace tuned {

constexpr int TOEP_H = 513, TOEP_C = 8, TOEP_T = 1025;
constexpr int WROW = 1040;  // Wc row: 8 zero floats + 1025 taps + 7 zero floats
constexpr int WPRE = 8;

// ---------------------------------------------------------------- forward
// grid (ceil(F/32), NSPLIT); 256 threads; each wave owns 16/NSPLIT/4 column tiles.
// K = (channel c, j) is walked in 16 chunks (8 channels x 2 halves of 260 j).  Double buffered:
// while the MFMAs consume chunk i from LDS buffer i&1, the global loads of chunk i+1 are in
// flight into registers; they are written to the other buffer after the MFMA block, one
// __syncthreads per chunk.  The 1040-float weight row of the chunk's channel travels with it.
// Staging is row-wise (wave w copies frames w, w+4, ...: no index arithmetic beyond one add),
// and inside the MFMA block the fragments of k-step s+1 are read before the MFMAs of step s.
constexpr int TF_JC = 260;           // j-chunk (2 chunks cover 520 >= 513), 130 k-steps (even)
constexpr int TF_ASTR = TF_JC + 1;   // odd row stride -> conflict-free A gathers
constexpr int TF_BUF = 32 * TF_ASTR + WROW;   // floats per buffer: A chunk + weight row
constexpr int TF_LDS = 2 * TF_BUF * 4;
constexpr int TF_RPW = 8;                         // frames (rows) per wave
constexpr int TF_LPR = (TF_JC + 63) / 64;         // loads per row and lane (5)
constexpr int TF_WPT = (WROW + 255) / 256;

template <int NBW>  // column tiles per wave (4 -> NSPLIT 1, 2 -> NSPLIT 2, 1 -> NSPLIT 4)
__global__ void __launch_bounds__(256) k_toep_fwd(const float* __restrict__ y2, const float* __restrict__ Wc,
                                                  const float* __restrict__ bias, float* __restrict__ xh, int F) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f0 = blockIdx.x * 32;
  const int p0 = (blockIdx.y * 4 + wave) * NBW * 32;
  f32x16 acc[NBW];
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb) acc[nb] = zero16();
  float ra[TF_RPW][TF_LPR], rw[TF_WPT];
  // column p = 512 (513 = 16*32 + 1): xh[f][512] = b + sum_{c,j} Wc[c][1024-j] y2[f][c][j] is a plain
  // dot product per frame; the workgroups with blockIdx.y == 0 accumulate it from the staged
  // chunks (wave w: frames w, w+4, ...; lanes stride over j), 1.2 % extra LDS reads, no extra HBM pass.
  float last[TF_RPW];
#pragma unroll
  for (int r = 0; r < TF_RPW; ++r) last[r] = 0.f;
  auto gload = [&](int chunk) {
    const int c = chunk >> 1, jc0 = (chunk & 1) * TF_JC;
#pragma unroll
    for (int r = 0; r < TF_RPW; ++r) {
      const int fl = wave + 4 * r;                       // wave-uniform
      const int f = f0 + fl < F ? f0 + fl : F - 1;       // clamped; rows past F are never stored
      const float* row = y2 + (int64_t)f * (TOEP_C * TOEP_H) + c * TOEP_H + jc0;
#pragma unroll
      for (int p = 0; p < TF_LPR; ++p) {
        int jj = lane + 64 * p;
        ra[r][p] = (jj < TF_JC && jc0 + jj < TOEP_H) ? row[jj] : 0.f;
      }
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
    for (int r = 0; r < TF_RPW; ++r)
#pragma unroll
      for (int p = 0; p < TF_LPR; ++p) {
        int jj = lane + 64 * p;
        if (jj < TF_JC) tA[(wave + 4 * r) * TF_ASTR + jj] = ra[r][p];
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
      // software pipeline: fragments of step s+1 (s+2) are in flight while step s computes
      float a0 = ap[0], a1;
      float b0[NBW], b1[NBW];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) b0[nb] = wp[nb * 32];
      for (int s = 0; s < TF_JC / 2; s += 2) {
        a1 = ap[2 * (s + 1)];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) b1[nb] = wp[nb * 32 - 2 * (s + 1)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[nb] = mfma32(a0, b0[nb], acc[nb]);
        __builtin_amdgcn_sched_barrier(0);
        a0 = ap[2 * (s + 2)];   // last trip reads one k-step past the chunk: still inside the buffer
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) b0[nb] = wp[nb * 32 - 2 * (s + 2)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[nb] = mfma32(a1, b1[nb], acc[nb]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (blockIdx.y == 0) {
      const float* tA = lds + (chunk & 1) * TF_BUF;
      const float* tW = tA + 32 * TF_ASTR;
      const int jc0 = (chunk & 1) * TF_JC;
#pragma unroll
      for (int p = 0; p < TF_LPR; ++p) {
        int jj = lane + 64 * p;
        // zero-padded on both sides: j >= 513 has y2 = 0, the weight index stays inside the row
        float wv = jj < TF_JC ? tW[WPRE + 1024 - jc0 - jj] : 0.f;
#pragma unroll
        for (int r = 0; r < TF_RPW; ++r)
          if (jj < TF_JC) last[r] += tA[(wave + 4 * r) * TF_ASTR + jj] * wv;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (chunk + 1 < 2 * TOEP_C) lstore((chunk + 1) & 1);
    __syncthreads();
  }
  const float bb = bias[0];
  if (blockIdx.y == 0) {
#pragma unroll
    for (int r = 0; r < TF_RPW; ++r) {
      float sum = wave_sum(last[r]);
      int f = f0 + wave + 4 * r;
      if (lane == 0 && f < F) xh[(int64_t)f * TOEP_H + 512] = sum + bb;
    }
  }
#pragma unroll
  for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      int f = f0 + acc_row(reg, lane);
      if (f < F) xh[(int64_t)f * TOEP_H + p0 + nb * 32 + l31] = acc[nb][reg] + bb;
    }
}

// ---------------------------------------------------------------- input gradient
// grid (ceil(F/32), 8/CPW); 256 threads; wave w owns column tiles j0 = (4w..4w+3)*32.
// The dxh tile of 32 frames (66 KB) is staged ONCE and reused for CPW channels (CPW = 8 for large
// batches: staging drops to ~2 % of the workgroup's time; CPW = 1 keeps small batches parallel).
constexpr int TD_K = 516;     // reduction length p padded to an even number of k-steps (258)
constexpr int TD_ASTR = 517;  // dxh row in LDS: 513 values + zeros up to TD_K ; odd stride
constexpr int TD_LDS = (32 * TD_ASTR + 2 * WROW) * 4;   // A tile + two weight rows (ping-pong)
constexpr int TD_LPR = (TOEP_H + 63) / 64;  // loads per row and lane (9; the 9th covers p = 512)

template <int CPW, int NWV>
__global__ void __launch_bounds__(NWV * 64) k_toep_dgrad(const float* __restrict__ dxh, const float* __restrict__ Wc,
                                                    float* __restrict__ dy, int F) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tA = lds;
  float* tW = lds + 32 * TD_ASTR;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int f0 = blockIdx.x * 32, c0 = blockIdx.y * CPW;
  constexpr int NT = NWV * 64, NBW = 16 / NWV, RPW = 32 / NWV;   // threads, column tiles and frames per wave
  for (int i = tid; i < WROW; i += NT) tW[i] = Wc[c0 * WROW + i];
  // row-wise staging: wave w copies frames w, w+4, ... (4 rows = 36 loads in flight per lane)
#pragma unroll
  for (int half = 0; half < RPW / 4; ++half) {
    float v[4][TD_LPR];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int fl = wave + NWV * (half * 4 + r);
      const int f = f0 + fl < F ? f0 + fl : F - 1;
      const float* row = dxh + (int64_t)f * TOEP_H;
#pragma unroll
      for (int p = 0; p < TD_LPR; ++p) {
        int i = lane + 64 * p;
        v[r][p] = (i < TOEP_H && f0 + fl < F) ? row[i] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int fl = wave + NWV * (half * 4 + r);
#pragma unroll
      for (int p = 0; p < TD_LPR; ++p) {
        int i = lane + 64 * p;
        if (i < TD_ASTR) tA[fl * TD_ASTR + i] = v[r][p];   // i in [513, 517) gets the zero padding
      }
    }
  }
  __syncthreads();
  const int j0 = wave * (NBW * 32);
  const float* ap = tA + l31 * TD_ASTR + lh;
  for (int ci = 0; ci < CPW; ++ci) {
    const int c = c0 + ci;
    const float* tWc = tW + (ci & 1) * WROW;
    // next channel's weight row into the other buffer (read by nobody until the barrier below)
    if (ci + 1 < CPW)
      for (int i = tid; i < WROW; i += NT) tW[((ci + 1) & 1) * WROW + i] = Wc[(c + 1) * WROW + i];
    f32x16 acc[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) acc[nb] = zero16();
    // B[k = p][n = j] = Wc[c][p - j + 512], p = 2*s + lh, j = j0 + nb*32 + l31
    const float* wp = tWc + WPRE + 512 + lh - j0 - l31;
    // software pipeline: fragments of the next k-step are read before the MFMAs of the current one
    float a0 = ap[0], a1;
    float b0[NBW], b1[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) b0[nb] = wp[-nb * 32];
    for (int s = 0; s < TD_K / 2; s += 2) {
      a1 = ap[2 * (s + 1)];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) b1[nb] = wp[2 * (s + 1) - nb * 32];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) acc[nb] = mfma32(a0, b0[nb], acc[nb]);
      __builtin_amdgcn_sched_barrier(0);
      const int s2 = s + 2 < TD_K / 2 ? s + 2 : s;   // last trip: re-read a valid step (unused)
      a0 = ap[2 * s2];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) b0[nb] = wp[2 * s2 - nb * 32];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) acc[nb] = mfma32(a1, b1[nb], acc[nb]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int f = f0 + acc_row(reg, lane);
        if (f < F) dy[(int64_t)f * (TOEP_C * TOEP_H) + c * TOEP_H + j0 + nb * 32 + l31] = acc[nb][reg];
      }
    // column j = 512: dy[f][c][512] = sum_p Wc[c][p] * dxh[f][p]; wave w reduces its RPW frames
    for (int i = 0; i < RPW; ++i) {
      int fl = wave * RPW + i;
      float sum = 0.f;
      for (int p = lane; p < TOEP_H; p += 64) sum += tA[fl * TD_ASTR + p] * tWc[WPRE + p];
      sum = wave_sum(sum);
      if (lane == 0 && f0 + fl < F) dy[(int64_t)(f0 + fl) * (TOEP_C * TOEP_H) + c * TOEP_H + 512] = sum;
    }
    __syncthreads();  // next channel's weight row complete; this channel's row no longer read
  }
}

}  // namespace tuned
}  // namespace vaenpvc
