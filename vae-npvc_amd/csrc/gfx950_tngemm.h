// gfx950_tngemm.h -- C[m][n] += sum_f X[f][m] * Y[f][n]  (reduction over FRAMES).
//
// Both operands are contiguous along their M / N index in HBM and the reduction index
// (frame) is the slow axis, which is exactly the fp32-MFMA operand shape: lanes 0..31 of
// an A (B) fragment read 32 consecutive floats of frame f, lanes 32..63 of frame f+1.  Chunks
// of 32 frames are staged once per workgroup through LDS (register-prefetch double buffer);
// a first version streamed HBM/L2 -> VGPR -> MFMA directly and stalled the L1 on duplicate
// in-flight lines (TCP_PENDING_STALL 46 %).
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
//   LN     : LayerNorm+lrelu applied to X while staging (heads: X = pre-LN output of enc layer 4)
//   GATHER : X rows gathered through xidx (speaker-embedding rows)
// The reduction runs over chunks of KF frames.  Each chunk's 128 X columns and 128 Y columns
// are read ONCE per workgroup (coalesced 512-byte rows) into registers while the MFMAs consume
// the previous chunk from LDS, then written to the other LDS buffer (one barrier per chunk).
// Fragments: lanes 0..31 read 32 consecutive floats of frame k, lanes 32..63 of frame k+1.
constexpr int TN_KF = 32;                       // frames per chunk
constexpr int TN_BUF = 2 * TN_KF * 128;         // floats per buffer: X[KF][128] then Y[KF][128]
constexpr int TN_EPT = TN_KF * 128 / 256;       // staged elements per thread and operand

template <bool TOEP, bool LN, bool GATHER>
__global__ void __launch_bounds__(256) k_tngemm(TnArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * TN_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int mb0 = blockIdx.x * 128, nb0 = blockIdx.y * 128;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const int fb = blockIdx.z * a.fchunk;
  const int fe = min(a.F, fb + a.fchunk);
  // staging: thread -> column (tid & 127), rows (tid >> 7) + 2k
  const int col = tid & 127, row0 = tid >> 7;
  const int m = mb0 + col, n = nb0 + col;
  const bool mok = m < a.M, nok = n < a.N;
  const int mm = mok ? m : 0, nn = nok ? n : 0;
  const int xoff = TOEP ? mm + (mm >> 9) : mm;  // (c, j<512) -> c*513 + j
  float g = 1.f, b = 0.f;
  if constexpr (LN) {
    g = a.gamma[mm / a.lndiv];
    b = a.beta[mm / a.lndiv];
  }
  float rx[TN_EPT], ry[TN_EPT];
  // TOEP edge term (column q = 512 of dxh, which the 512-wide MFMA part does not cover):
  //   dW[1024-j][c] += sum_f y2[f,c,j] * dxh[f,512]   for j < 512
  // is accumulated from the X values staged anyway by the workgroups with blockIdx.y == 0.
  float edge = 0.f;
  const bool do_edge = TOEP && blockIdx.y == 0;
  const float* __restrict__ X = a.X;
  const float* __restrict__ Y = a.Y;
  const int ldx = a.ldx, ldy = a.ldy;
  const bool tile_full = (mb0 + 128 <= a.M) && (nb0 + 128 <= a.N);
  // fast path: whole chunk inside [fb, fe) and a full 128x128 tile -> no guards, offsets advance
  // by a constant (one add per load); guarded path only for edge tiles / the chunk tail
  auto gload = [&](int f0) {
    if (!GATHER && tile_full && f0 + TN_KF <= fe) {
      int ox = (f0 + row0) * ldx + xoff, oy = (f0 + row0) * ldy + nn;
#pragma unroll
      for (int k = 0; k < TN_EPT; ++k) {
        float v = X[ox];
        if constexpr (LN) {
          int f = f0 + row0 + 2 * k;
          v = lnact_v(v, a.st[2 * f], a.st[2 * f + 1], g, b);
        }
        rx[k] = v;
        ry[k] = Y[oy];
        ox += 2 * ldx;
        oy += 2 * ldy;
      }
      if constexpr (TOEP) {
        if (do_edge) {  // workgroup-uniform; one batched block of loads, then the fmas
          float d[TN_EPT];
          int oe = (f0 + row0) * ldy + 512;
#pragma unroll
          for (int k = 0; k < TN_EPT; ++k) {
            d[k] = Y[oe];
            oe += 2 * ldy;
          }
#pragma unroll
          for (int k = 0; k < TN_EPT; ++k) edge += rx[k] * d[k];
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < TN_EPT; ++k) {
        int f = f0 + row0 + 2 * k;
        bool fok = f < fe;
        int ff = fok ? f : fb;
        int xrow = ff;
        if constexpr (GATHER) xrow = (int)a.xidx[ff];
        float v = X[xrow * ldx + xoff];
        if constexpr (LN) v = lnact_v(v, a.st[2 * ff], a.st[2 * ff + 1], g, b);
        rx[k] = (fok && mok) ? v : 0.f;
        float w = Y[ff * ldy + nn];
        ry[k] = (fok && nok) ? w : 0.f;
        if constexpr (TOEP) {
          if (do_edge && fok && mok) edge += v * Y[ff * ldy + 512];
        }
      }
    }
  };
  auto lstore = [&](int buf) {
    float* px = lds + buf * TN_BUF;
    float* py = px + TN_KF * 128;
#pragma unroll
    for (int k = 0; k < TN_EPT; ++k) {
      px[(row0 + 2 * k) * 128 + col] = rx[k];
      py[(row0 + 2 * k) * 128 + col] = ry[k];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

  gload(fb);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int f0 = fb; f0 < fe; f0 += TN_KF, buf ^= 1) {
    const bool more = f0 + TN_KF < fe;
    if (more) gload(f0 + TN_KF);
    __builtin_amdgcn_sched_barrier(0);
    {
      const float* px = lds + buf * TN_BUF + lh * 128 + wm + l31;
      const float* py = lds + buf * TN_BUF + TN_KF * 128 + lh * 128 + wn + l31;
      // fragments of k-step k+1 are read before the MFMAs of k-step k (LDS latency off the MFMA chain)
      float x0 = px[0], x1 = px[32], y0 = py[0], y1 = py[32];
      float u0, u1, v0, v1;
#pragma unroll
      for (int k = 0; k < TN_KF / 2; k += 2) {
        u0 = px[(k + 1) * 256];
        u1 = px[(k + 1) * 256 + 32];
        v0 = py[(k + 1) * 256];
        v1 = py[(k + 1) * 256 + 32];
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = mfma32(x0, y0, acc[0][0]);
        acc[0][1] = mfma32(x0, y1, acc[0][1]);
        acc[1][0] = mfma32(x1, y0, acc[1][0]);
        acc[1][1] = mfma32(x1, y1, acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        const int k2 = k + 2 < TN_KF / 2 ? k + 2 : k;
        x0 = px[k2 * 256];
        x1 = px[k2 * 256 + 32];
        y0 = py[k2 * 256];
        y1 = py[k2 * 256 + 32];
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = mfma32(u0, v0, acc[0][0]);
        acc[0][1] = mfma32(u0, v1, acc[0][1]);
        acc[1][0] = mfma32(u1, v0, acc[1][0]);
        acc[1][1] = mfma32(u1, v1, acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) lstore(buf ^ 1);
    __syncthreads();
  }

  const int m0 = mb0 + wm, n0 = nb0 + wn;
  if constexpr (!TOEP) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          int mo = m0 + i * 32 + acc_row(reg, lane);
          int no = n0 + j * 32 + l31;
          if (mo < a.M && no < a.N) atomicAdd(a.C + (int64_t)mo * a.ldc + no, acc[i][j][reg]);
        }
  } else {
    // wave tile rows m0..m0+63 lie in ONE channel (512 % 64 == 0); diagonal d = col - row
    __syncthreads();
    float* dg = lds + wave * 128;
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
          int cl = j * 32 + l31;
          atomicAdd(&dg[cl - row + 63], acc[i][j][reg]);
        }
    wave_lds_sync();
    const int c = m0 >> 9, j0 = m0 & 511;
    for (int d = lane; d < 127; d += 64) {
      int t = n0 - j0 + (d - 63) + 512;  // always in [1, 1023]
      atomicAdd(a.C + t * 8 + c, dg[d]);
    }
    if (do_edge && mok) atomicAdd(a.C + (1024 - (mm & 511)) * 8 + (mm >> 9), edge);
  }
}

inline void launch_tngemm(const TnArgs& a, bool toep, int kchunks, hipStream_t s) {
  TnArgs b = a;
  b.fchunk = ((a.F + kchunks - 1) / kchunks + TN_KF - 1) / TN_KF * TN_KF;
  dim3 grid((unsigned)cdiv(a.M, 128), (unsigned)cdiv(a.N, 128), (unsigned)cdiv(a.F, b.fchunk));
  const bool ln = a.st != nullptr, gather = a.xidx != nullptr;
  if (toep)
    hipLaunchKernelGGL((k_tngemm<true, false, false>), grid, dim3(256), 0, s, b);
  else if (gather)
    hipLaunchKernelGGL((k_tngemm<false, false, true>), grid, dim3(256), 0, s, b);
  else if (ln)
    hipLaunchKernelGGL((k_tngemm<false, true, false>), grid, dim3(256), 0, s, b);
  else
    hipLaunchKernelGGL((k_tngemm<false, false, false>), grid, dim3(256), 0, s, b);
}

// Remaining edge term of the Toeplitz weight gradient (row j = 512 of y2, all q):
//   dW[t][c] += sum_f y2[f,c,512] * dxh[f,t]        for t <= 512
// grid (frame chunks, 3): thread <-> ONE tap t = 256*blockIdx.y + tid with 8 channel accumulators.  The
// 8 y2 values of 64 frames sit in 8 registers (lane <-> frame) and are broadcast with v_readlane, so a
// frame costs one coalesced load and 8 FMAs.  Few, long chunks: the result is 513 x 8 atomics per
// workgroup onto the same 4104 addresses (with 512 chunks those atomics WERE the kernel's run time).
#ifndef TW512_U
#define TW512_U 16
#endif
__global__ void __launch_bounds__(256) k_toep_wgrad_row512(const float* __restrict__ y2, const float* __restrict__ dxh,
                                                           float* __restrict__ dW, int F, int fchunk) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int t = 256 * blockIdx.y + tid;
  const bool tok = t <= 512;
  const int fb = blockIdx.x * fchunk, fe = min(F, fb + fchunk);
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  for (int f0 = fb; f0 < fe; f0 += 64) {
    const int nf = min(64, fe - f0);
    float yv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) yv[c] = lane < nf ? y2[(int64_t)(f0 + lane) * 4104 + c * 513 + 512] : 0.f;
    // (sixteen rows of d(xh) in flight per thread: with four the kernel was a chain of 16 round trips per 64 frames)
    for (int j = 0; j < nf; j += TW512_U) {
      float d[TW512_U];
#pragma unroll
      for (int u = 0; u < TW512_U; ++u) d[u] = (tok && j + u < nf) ? dxh[(int64_t)(f0 + j + u) * 513 + t] : 0.f;
#pragma unroll
      for (int u = 0; u < TW512_U; ++u)
#pragma unroll
        for (int c = 0; c < 8; ++c)
          acc[c] += d[u] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yv[c]), (j + u) & 63));
    }
  }
  if (tok)
#pragma unroll
    for (int c = 0; c < 8; ++c) atomicAdd(dW + t * 8 + c, acc[c]);
}

}  // namespace tuned
}  // namespace vaenpvc
