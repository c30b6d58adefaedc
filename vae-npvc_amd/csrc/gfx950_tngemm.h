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
//   TOEP   : X = y2 (activated dec-2 output, [F][4104]), rows m = (c, j<512); diagonal epilogue
//   EDGE   : M, N or the frame count are not multiples of the tile -> guarded operands
//   LN     : LayerNorm+lrelu applied to X on load (heads: X = pre-LN output of encoder layer 4)
//   GATHER : X rows gathered through xidx (speaker-embedding rows)
// Operand loads run one chunk (U k-steps = 2U frames) ahead of the MFMAs, ping-pong register sets.
template <bool TOEP, bool EDGE, bool LN, bool GATHER>
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
    if constexpr (LN) {
      int ch = mm / a.lndiv;
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

  constexpr int U = 4;
  struct Chunk {
    float x[U][2], y[U][2], mean[U], rstd[U];
  };
  const float* __restrict__ X = a.X;
  const float* __restrict__ Y = a.Y;
  auto load = [&](int f, Chunk& c) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int ff = f + 2 * u + lh;
      if constexpr (EDGE) ff = ff < fe ? ff : fb;
      int xrow = ff;
      if constexpr (GATHER) xrow = (int)a.xidx[ff];
      if constexpr (LN) {
        c.mean[u] = a.st[2 * ff];
        c.rstd[u] = a.st[2 * ff + 1];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        c.x[u][i] = X[xrow * a.ldx + xoff[i]];
        c.y[u][i] = Y[ff * a.ldy + noff[i]];
      }
    }
  };
  auto compute = [&](int f, const Chunk& c) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float xa[2], yb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v = c.x[u][i];
        if constexpr (LN) v = lnact_v(v, c.mean[u], c.rstd[u], g[i], b[i]);
        float w = c.y[u][i];
        if constexpr (EDGE) {
          const bool fok = f + 2 * u + lh < fe;
          v = (fok && mok[i]) ? v : 0.f;
          w = (fok && nok[i]) ? w : 0.f;
        }
        xa[i] = v;
        yb[i] = w;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(xa[i], yb[j], acc[i][j]);
    }
  };
  Chunk c0, c1;
  load(fb, c0);
  int f = fb;
  while (true) {
    if (f + 2 * U < fe) load(f + 2 * U, c1);
    compute(f, c0);
    f += 2 * U;
    if (f >= fe) break;
    if (f + 2 * U < fe) load(f + 2 * U, c0);
    compute(f, c1);
    f += 2 * U;
    if (f >= fe) break;
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
  const bool edge = (a.M % 128) || (a.N % 128) || (a.F % 8);
  const bool ln = a.st != nullptr, gather = a.xidx != nullptr;
#define VAENPVC_TN(T, E, L, G) hipLaunchKernelGGL((k_tngemm<T, E, L, G>), grid, dim3(256), 0, s, b)
  if (toep) {
    if (edge) VAENPVC_TN(true, true, false, false); else VAENPVC_TN(true, false, false, false);
  } else if (gather) {
    VAENPVC_TN(false, true, false, true);
  } else if (ln) {
    if (edge) VAENPVC_TN(false, true, true, false); else VAENPVC_TN(false, false, true, false);
  } else {
    if (edge) VAENPVC_TN(false, true, false, false); else VAENPVC_TN(false, false, false, false);
  }
#undef VAENPVC_TN
}

// Edge terms of the Toeplitz weight gradient not covered by the 512x512 MFMA part:
//   t <= 512 : dW[t][c] += sum_f y2[f,c,512] * dxh[f,t]        (row j = 512, all q)
//   t >  512 : dW[t][c] += sum_f y2[f,c,1024-t] * dxh[f,512]   (column q = 512, j < 512)
// One thread per (t, c); grid (ceil(8200/256), frame chunks).
__global__ void __launch_bounds__(256) k_toep_wgrad_edges(const float* __restrict__ y2, const float* __restrict__ dxh,
                                                          float* __restrict__ dW, int F, int fchunk) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 1025 * 8) return;
  int t = idx >> 3, c = idx & 7;
  int j = t <= 512 ? 512 : 1024 - t;
  int q = t <= 512 ? t : 512;
  int fb = blockIdx.y * fchunk, fe = min(F, fb + fchunk);
  float s = 0.f;
  for (int f = fb; f < fe; f += 4) {
    float x[4], d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int ff = f + u < fe ? f + u : fb;
      x[u] = y2[(int64_t)ff * 4104 + c * 513 + j];
      d[u] = dxh[(int64_t)ff * 513 + q];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (f + u < fe) s += x[u] * d[u];
  }
  atomicAdd(dW + idx, s);
}

}  // namespace tuned
}  // namespace vaenpvc
