// gfx950_tngemm.h -- C[m][n] += sum_f X[f][m] * Y[f][n]  (reduction over FRAMES).
//
// Both operands are contiguous along their M / N index in HBM and the reduction index
// (frame) is the slow axis, which is exactly the fp32-MFMA operand shape: lanes 0..31 of
// an A (B) fragment read 128 contiguous bytes of frame f, lanes 32..63 of frame f+1.  So
// operands go HBM/L2 -> VGPR -> MFMA with no LDS staging at all.
//
// Used for  * dense weight gradients: dW_heads = y4^T dz, dW_merge = [z|e]^T dh
//           * the weight gradient of the 1025-tap last decoder layer: the correlation
//             dW[t][c] = sum_f sum_j y2[f,c,j] dxh[f,j+t-512] is computed as the dense
//             cross-product P[(c,j)][q] = sum_f y2[f,c,j] dxh[f,q] (same MAC count, plain
//             GEMM) whose 64x64 wave tiles are summed along diagonals t = q-j+512 in LDS
//             before ONE global atomic per diagonal (TOEP mode).
// Partial sums over frame chunks (blockIdx.z) are combined with fp32 global atomics.
#pragma once
#include "gfx950_common.h"

namespace vaenpvc {
namespace tuned {

struct TnArgs {
  const float* X;       // [rows][ldx]
  const int64_t* xidx;  // optional row gather for X (speaker-embedding rows)
  const float* st;      // optional LN-on-load of X: per-frame (mean, rstd)
  const float* gamma;
  const float* beta;
  int lndiv;  // channel of column m = m / lndiv
  int ldx;
  const float* Y;  // [F][ldy]
  int ldy;
  int M, N, F;
  float* C;  // C[m*ldc + n] (atomicAdd)   | TOEP: dW[t*8 + c]
  int ldc;
  int fchunk;  // frames per blockIdx.z
};

// 256 threads = 2x2 waves, wave tile 64x64 (2x2 MFMA tiles), workgroup tile 128x128.
template <bool TOEP>
__global__ void __launch_bounds__(256) k_tngemm(TnArgs a) {
  __shared__ float diag[4][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * 128 + (wave >> 1) * 64;
  const int n0 = blockIdx.y * 128 + (wave & 1) * 64;
  const int fb = blockIdx.z * a.fchunk;
  const int fe = min(a.F, fb + a.fchunk);
  int xoff[2], noff[2];
  bool mok[2], nok[2];
  float g[2] = {1.f, 1.f}, b[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int m = m0 + i * 32 + l31;
    mok[i] = m < a.M;
    int mm = mok[i] ? m : 0;
    xoff[i] = TOEP ? mm + (mm >> 9) : mm;  // (c, j<512) -> c*513 + j
    if (a.st) {
      int ch = TOEP ? (mm >> 9) : mm / a.lndiv;
      g[i] = a.gamma[ch];
      b[i] = a.beta[ch];
    }
    int n = n0 + i * 32 + l31;
    nok[i] = n < a.N;
    noff[i] = nok[i] ? n : 0;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

  // operand loads run one chunk (U k-steps = 2U frames) ahead of the MFMAs
  constexpr int U = 4;
  float xr[U][2], yr[U][2], mr[U], rr[U];
  float xn[U][2], yn[U][2], mn[U], rn[U];
  auto load_chunk = [&](int f, float (&x)[U][2], float (&y)[U][2], float (&mm)[U], float (&rs)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int ff = f + 2 * u + lh;
      int fc = ff < fe ? ff : fb;
      int64_t xr_ = a.xidx ? a.xidx[fc] : (int64_t)fc;
      mm[u] = 0.f;
      rs[u] = 1.f;
      if (a.st) {
        mm[u] = a.st[2 * fc];
        rs[u] = a.st[2 * fc + 1];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        x[u][i] = a.X[xr_ * a.ldx + xoff[i]];
        y[u][i] = a.Y[(int64_t)fc * a.ldy + noff[i]];
      }
    }
  };
  load_chunk(fb, xr, yr, mr, rr);
  for (int f = fb; f < fe; f += 2 * U) {
    const bool more = f + 2 * U < fe;
    if (more) load_chunk(f + 2 * U, xn, yn, mn, rn);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool fok = f + 2 * u + lh < fe;
      float xa[2], yb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v = xr[u][i];
        if (a.st) v = lnact_v(v, mr[u], rr[u], g[i], b[i]);
        xa[i] = (fok && mok[i]) ? v : 0.f;
        yb[i] = (fok && nok[i]) ? yr[u][i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(xa[i], yb[j], acc[i][j]);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        mr[u] = mn[u];
        rr[u] = rn[u];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          xr[u][i] = xn[u][i];
          yr[u][i] = yn[u][i];
        }
      }
    }
  }

  if constexpr (!TOEP) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          int m = m0 + i * 32 + acc_row(reg, lane);
          int n = n0 + j * 32 + l31;
          if (m < a.M && n < a.N) atomicAdd(a.C + (int64_t)m * a.ldc + n, acc[i][j][reg]);
        }
  } else {
    // wave tile rows m0..m0+63 lie in ONE channel (512 % 64 == 0); diagonal d = col - row
    float* dg = diag[wave];
    dg[lane] = 0.f;
    dg[lane + 64] = 0.f;
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          int row = i * 32 + acc_row(reg, lane);
          int col = j * 32 + l31;
          atomicAdd(&dg[col - row + 63], acc[i][j][reg]);
        }
    wave_lds_sync();
    const int c = m0 >> 9, j0 = m0 & 511;
    for (int d = lane; d < 127; d += 64) {
      int t = n0 - j0 + (d - 63) + 512;  // always in [1, 1023]
      atomicAdd(a.C + t * 8 + c, dg[d]);
    }
  }
}

inline void launch_tngemm(const TnArgs& a, bool toep, int kchunks, hipStream_t s) {
  TnArgs b = a;
  b.fchunk = ((a.F + kchunks - 1) / kchunks + 7) / 8 * 8;
  dim3 grid((unsigned)cdiv(a.M, 128), (unsigned)cdiv(a.N, 128), (unsigned)cdiv(a.F, b.fchunk));
  if (toep)
    hipLaunchKernelGGL(k_tngemm<true>, grid, dim3(256), 0, s, b);
  else
    hipLaunchKernelGGL(k_tngemm<false>, grid, dim3(256), 0, s, b);
}

// Edge terms of the Toeplitz weight gradient not covered by the 512x512 MFMA part:
//   t <= 512 : dW[t][c] += sum_f y2[f,c,512] * dxh[f,t]        (row j = 512, all q)
//   t >  512 : dW[t][c] += sum_f y2[f,c,1024-t] * dxh[f,512]   (column q = 512, j < 512)
// One thread per (t, c); grid (ceil(8200/256), frame chunks).
__global__ void __launch_bounds__(256) k_toep_wgrad_edges(const float* __restrict__ a2, const float* __restrict__ st,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ dxh, float* __restrict__ dW,
                                                          int F, int fchunk) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 1025 * 8) return;
  int t = idx >> 3, c = idx & 7;
  int j = t <= 512 ? 512 : 1024 - t;
  int q = t <= 512 ? t : 512;
  const float g = gamma[c], b = beta[c];
  int fb = blockIdx.y * fchunk, fe = min(F, fb + fchunk);
  float s = 0.f;
  for (int f = fb; f < fe; f += 4) {
    float x[4], d[4], m4[4], r4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int ff = f + u < fe ? f + u : fb;
      x[u] = a2[(int64_t)ff * 4104 + c * 513 + j];
      d[u] = dxh[(int64_t)ff * 513 + q];
      m4[u] = st[2 * ff];
      r4[u] = st[2 * ff + 1];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (f + u < fe) s += lnact_v(x[u], m4[u], r4[u], g, b) * d[u];
  }
  atomicAdd(dW + idx, s);
}

}  // namespace tuned
}  // namespace vaenpvc
