// gfx950_convwgrad.h -- weight gradients of the strided conv / conv_transpose layers.
//
//   dW[m = (t, xc)][n] += sum_f sum_r X'[f, xc, S*r - PAD + t] * Y'[f, n, r]
//
//   conv (encoder):   X = layer input (LN+lrelu on load), Y = d(pre-LN output)   -> dW[t][cin][cout]
//   conv_transpose:   X = d(pre-LN output),               Y = layer input        -> dW[t][cout][cin]
// i.e. in both cases the result lands directly in the TF kernel layout of the flat
// gradient buffer.  GEMM view: M = T*XC, N = YC, K = (frame, r).  Both operand tiles of TF
// frames are staged in LDS (coalesced HBM reads, zero halos); fragments are gathered with
// lane <-> channel (odd channel stride -> conflict free), k-step = the same position r of two
// consecutive frames.  Waves split M (and K when M is small); partial sums over frame
// chunks are combined with fp32 global atomics.
#pragma once
#include "gfx950_common.h"
#include "gfx950_stage.h"

#ifndef VAENPVC_PROF
#define VAENPVC_PROF 0
#endif
#ifndef VAENPVC_WABL
#define VAENPVC_WABL 0  // developer ablations: 1 no MFMA loop, 2 no LDS store of the prefetch, 3 no prefetch loads
#endif

namespace vaenpvc {
namespace tuned {

#if VAENPVC_PROF
// developer instrumentation: [slot][0..7] = waves, stage, barrier1, compute, barrier2, epilogue, total
__device__ unsigned long long g_wg_prof[16][8];
inline int g_wg_prof_next = 0;
#define WPROF_T(var) const long long var = (long long)__builtin_amdgcn_s_memtime()
#else
#define WPROF_T(var)
#endif

constexpr int odd_up(int v) { return v | 1; }

template <int XC_, int XH_, int YC_, int YH_, int T_, int S_, int PAD_, bool XLN_, bool YLN_, int TF_, int NTW_,
          int NWV_ = 4, int WM_ = 0, int WPE_ = 2, bool T16_ = false>
struct WgCfg {
  static constexpr int WPE = WPE_;  // waves per SIMD the register allocation must allow
  // T16: 16x16x4 MFMA tiles (v_mfma_f32_16x16x4_f32, same MAC rate) for layers with <= 16 output channels
  // or very few (tap, channel) rows: a 32-wide tile would be half / three quarters padding.  k = 4 =
  // two frames x two consecutive positions.
  static constexpr bool T16 = T16_;
  static constexpr int XC = XC_, XH = XH_, YC = YC_, YH = YH_, T = T_, S = S_, PAD = PAD_, TF = TF_, NTW = NTW_;
  static constexpr bool XLN = XLN_, YLN = YLN_;
  static constexpr int NWV = NWV_, NTHR = NWV_ * 64;
  static constexpr int M = T * XC, MTL = cdiv(M, T16_ ? 16 : 32);
  static constexpr int NTL = cdiv(YC, 32), NSPLIT = cdiv(NTL, NTW);
  // waves = WM (split of the M tiles) x WK (split of the k-steps)
  static constexpr int WM = WM_ > 0 ? WM_ : (MTL >= 4 ? 4 : (MTL >= 2 ? 2 : 1));
  static constexpr int WK = NWV / WM, MTW = cdiv(MTL, WM);
  static_assert(WM * WK == NWV, "waves = WM x WK");
  static constexpr int HLO = PAD, HHI = cmax(0, S * (YH - 1) - PAD + T - 1 - (XH - 1));
  // A gathers: lane <-> m = (t, xc) reads xc*CSTRX + t.  XC >= 32: one tap per fragment, odd
  // stride.  XC < 32: a fragment spans 32/XC taps, stride == 32/XC (mod 32) keeps all 32 banks distinct.
  static constexpr int CSTRX = XC >= 32 ? odd_up(HLO + XH + HHI) : next_mod32(HLO + XH + HHI, 32 / XC);
  static constexpr int CSTRY = odd_up(YH);
  static constexpr int FSTRX = XC * CSTRX, FSTRY = NTW * 32 * CSTRY;
  static constexpr int XT = rup(TF * FSTRX, 4), YT = rup(TF * FSTRY, 4);
  static_assert(TF % 2 == 0, "k-steps pair two frames");
  static constexpr int HP = TF / 2;
  // positions r handled per trip of the k loop (>= ~8 MFMAs between two batches of LDS reads)
  static constexpr int RU = cmax(1, cmin_(4, 8 / (HP * MTW * NTW)));
};

struct WgArgs {
  const float* X;
  const float* xst;  // LN of X (XLN)
  const float* xg;
  const float* xb;
  const float* Y;
  const float* yst;  // LN of Y (YLN)
  const float* yg;
  const float* yb;
  float* dW;  // [M][YC] atomicAdd
  int F;
  int fchunk;  // frames per blockIdx.x (multiple of TF)
#if VAENPVC_PROF
  int slot;
#endif
};

// Software pipeline per workgroup: the global loads of sub-tile t+1 are issued before the MFMAs
// of sub-tile t (both MFMA operands come from LDS, so nothing in the k loop waits on them) and
// are written to the single LDS tile pair after the compute phase; two barriers per sub-tile.
template <class C>
__global__ void __launch_bounds__(C::NTHR, C::WPE) k_convwgrad(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using OX = TileStager<C::XC, C::XC, C::XH, C::CSTRX, C::FSTRX, C::HLO, C::XLN, C::TF, C::NWV>;
  using OY = TileStager<C::NTW * 32, C::YC, C::YH, C::CSTRY, C::FSTRY, 0, C::YLN, C::TF, C::NWV>;
  float* tX = lds;
  float* tY = lds + C::XT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave % C::WM, wk = wave / C::WM;
  const int nc0 = blockIdx.y * C::NTW * 32;              // first Y channel of this workgroup
  const int ych = min(C::NTW * 32, C::YC - nc0);         // valid Y channels
  for (int i = tid; i < (C::XT + C::YT) / 4; i += C::NTHR) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int TS = C::T16 ? 16 : 32;                       // tile edge
  constexpr int NTS = C::T16 ? cdiv(cmin_(C::YC, C::NTW * 32), 16) : C::NTW;  // column tiles per workgroup
  const int lt = C::T16 ? (lane & 15) : l31;                 // lane's row / column inside a tile
  const int g = lane >> 4;                                   // T16: k-group = (frame g&1, position +(g>>1))
  int baseA[C::MTW];
  bool aok[C::MTW];
#pragma unroll
  for (int i = 0; i < C::MTW; ++i) {
    int m = (wm + i * C::WM) * TS + lt;
    aok[i] = m < C::M;
    int mm = aok[i] ? m : 0;
    int t = mm / C::XC, xc = mm - t * C::XC;
    baseA[i] = C::T16 ? (g & 1) * C::FSTRX + xc * C::CSTRX + C::HLO - C::PAD + t + C::S * (g >> 1)
                      : lh * C::FSTRX + xc * C::CSTRX + C::HLO - C::PAD + t;
  }
  int baseB[NTS];
#pragma unroll
  for (int j = 0; j < NTS; ++j)
    baseB[j] = C::T16 ? (g & 1) * C::FSTRY + (j * 16 + lt) * C::CSTRY + (g >> 1) : lh * C::FSTRY + (j * 32 + l31) * C::CSTRY;
  f32x16 acc[C::T16 ? 1 : C::MTW][C::T16 ? 1 : C::NTW];
  f32x4 acc4[C::T16 ? C::MTW : 1][C::T16 ? NTS : 1];
#pragma unroll
  for (int i = 0; i < (C::T16 ? 1 : C::MTW); ++i)
#pragma unroll
    for (int j = 0; j < (C::T16 ? 1 : C::NTW); ++j) acc[i][j] = zero16();
#pragma unroll
  for (int i = 0; i < (C::T16 ? C::MTW : 1); ++i)
#pragma unroll
    for (int j = 0; j < (C::T16 ? NTS : 1); ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fb = blockIdx.x * a.fchunk;
  const int fe = min(a.F, fb + a.fchunk);
  OX ox;
  OY oy;
  ox.init(tX, a.xg, a.xb, 0, C::XC);
  oy.init(tY, a.yg, a.yb, nc0, ych);
#if VAENPVC_PROF
  long long pc[6] = {0, 0, 0, 0, 0, 0};
  WPROF_T(k0);
#endif
  ox.gload(a.X, a.xst, fb, min(C::TF, fe - fb), 0, C::XC);
  oy.gload(a.Y, a.yst, fb, min(C::TF, fe - fb), nc0, ych);
  __syncthreads();  // zero fill done
  ox.lstore(tX, min(C::TF, fe - fb), C::XC);
  oy.lstore(tY, min(C::TF, fe - fb), ych);
  __syncthreads();
  for (int f0 = fb; f0 < fe; f0 += C::TF) {
    const int fn = f0 + C::TF;
    WPROF_T(w0);
#if VAENPVC_WABL != 3
    if (fn < fe) {
      ox.gload(a.X, a.xst, fn, min(C::TF, fe - fn), 0, C::XC);
      oy.gload(a.Y, a.yst, fn, min(C::TF, fe - fn), nc0, ych);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
    WPROF_T(w1);
    // k-steps: RU positions r per trip x frame pairs fp; the WK waves that share an M range take
    // interleaved groups of positions.  All fragment reads of a trip are issued before its MFMAs.
#if VAENPVC_WABL == 1
    if (a.F < 0)
#endif
    if constexpr (C::T16) {
      // position PAIRS (2*rp, 2*rp+1) x frame pairs; RU16 pairs per trip
      constexpr int NRP = cdiv(C::YH, 2), RU16 = 4;
      for (int rp0 = wk * RU16; rp0 < NRP; rp0 += C::WK * RU16) {
        float av[RU16][C::HP][C::MTW], bv[RU16][C::HP][NTS];
#pragma unroll
        for (int u = 0; u < RU16; ++u) {
          const int rp = rp0 + u < NRP ? rp0 + u : NRP - 1;
          const bool rok = rp0 + u < NRP && 2 * rp + (g >> 1) < C::YH;  // (odd YH: the last pair has one position)
#pragma unroll
          for (int fp = 0; fp < C::HP; ++fp) {
#pragma unroll
            for (int i = 0; i < C::MTW; ++i) {
              float v = tX[baseA[i] + C::S * 2 * rp + fp * 2 * C::FSTRX];
              av[u][fp][i] = aok[i] ? v : 0.f;
            }
#pragma unroll
            for (int j = 0; j < NTS; ++j) {
              float v = tY[baseB[j] + 2 * rp + fp * 2 * C::FSTRY];
              bv[u][fp][j] = rok ? v : 0.f;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < RU16; ++u)
#pragma unroll
          for (int fp = 0; fp < C::HP; ++fp)
#pragma unroll
            for (int i = 0; i < C::MTW; ++i)
#pragma unroll
              for (int j = 0; j < NTS; ++j)
                acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][fp][i], bv[u][fp][j], acc4[i][j], 0, 0, 0);
      }
    } else
    for (int r0 = wk * C::RU; r0 < C::YH; r0 += C::WK * C::RU) {
      float av[C::RU][C::HP][C::MTW], bv[C::RU][C::HP][C::NTW];
#pragma unroll
      for (int u = 0; u < C::RU; ++u) {
        const bool rok = r0 + u < C::YH;  // wave-uniform
        const int r = rok ? r0 + u : C::YH - 1;
#pragma unroll
        for (int fp = 0; fp < C::HP; ++fp) {
#pragma unroll
          for (int i = 0; i < C::MTW; ++i) {
            float v = tX[baseA[i] + C::S * r + fp * 2 * C::FSTRX];
            av[u][fp][i] = aok[i] ? v : 0.f;
          }
#pragma unroll
          for (int j = 0; j < C::NTW; ++j) {
            float v = tY[baseB[j] + r + fp * 2 * C::FSTRY];
            bv[u][fp][j] = rok ? v : 0.f;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < C::RU; ++u)
#pragma unroll
        for (int fp = 0; fp < C::HP; ++fp)
#pragma unroll
          for (int i = 0; i < C::MTW; ++i)
#pragma unroll
            for (int j = 0; j < C::NTW; ++j) acc[i][j] = mfma32(av[u][fp][i], bv[u][fp][j], acc[i][j]);
    }
#if VAENPVC_PROF
    asm volatile("s_nop 0" ::: "memory");
#endif
    WPROF_T(w2);
    __syncthreads();  // sub-tile consumed
    WPROF_T(w3);
#if VAENPVC_PROF
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WPROF_T(w3b);
    pc[5] += w3b - w3;  // landing of the prefetch
#endif
#if VAENPVC_WABL != 2
    if (fn < fe) {
      ox.lstore(tX, min(C::TF, fe - fn), C::XC);
      oy.lstore(tY, min(C::TF, fe - fn), ych);
    }
#endif
#if VAENPVC_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    WPROF_T(w4);
    __syncthreads();
#if VAENPVC_PROF
    WPROF_T(w5);
    pc[0] += (w1 - w0);  // gload issue
    pc[4] += (w4 - w3);  // lstore (incl. landing of the loads)
    pc[1] += w3 - w2;                // barrier after compute
    pc[2] += w2 - w1;                // compute
#endif
  }
  WPROF_T(k1);
  if constexpr (C::T16) {
    // 16x16 accumulator: column = lane & 15, row = 4*(lane >> 4) + reg
#pragma unroll
    for (int i = 0; i < C::MTW; ++i)
#pragma unroll
      for (int j = 0; j < NTS; ++j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          int m = (wm + i * C::WM) * 16 + 4 * g + reg;
          int n = nc0 + j * 16 + lt;
          if (m < C::M && n < C::YC) atomicAdd(a.dW + (int64_t)m * C::YC + n, acc4[i][j][reg]);
        }
  } else {
#pragma unroll
  for (int i = 0; i < C::MTW; ++i)
#pragma unroll
    for (int j = 0; j < C::NTW; ++j)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        int m = (wm + i * C::WM) * 32 + acc_row(reg, lane);
        int n = nc0 + j * 32 + l31;
        if (m < C::M && n < C::YC) atomicAdd(a.dW + (int64_t)m * C::YC + n, acc[i][j][reg]);
      }
  }
#if VAENPVC_PROF
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WPROF_T(k2);
    if ((threadIdx.x & 63) == 0) {
      unsigned long long* g = g_wg_prof[a.slot];
      atomicAdd(g + 0, 1ull);
      for (int i = 0; i < 4; ++i) atomicAdd(g + 1 + i, (unsigned long long)pc[i]);
      atomicAdd(g + 5, (unsigned long long)(k2 - k1));
      atomicAdd(g + 6, (unsigned long long)(k2 - k0));
      atomicAdd(g + 7, (unsigned long long)pc[4]);
      atomicAdd(g + 4, (unsigned long long)pc[5]);  // (reuses the barrier-2 column)
    }
  }
#endif
}

template <class C>
inline void launch_convwgrad(const WgArgs& a0, int target_wgs, hipStream_t s) {
  constexpr int LDS_BYTES = (C::XT + C::YT) * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "operand tiles exceed the LDS of a CU");
  rt().ensure_lds(reinterpret_cast<const void*>(&k_convwgrad<C>), LDS_BYTES);
  WgArgs a = a0;
  int chunks = cmax(1, target_wgs / C::NSPLIT);
  a.fchunk = rup(cmax(1, cdiv(a.F, chunks)), C::TF);
  dim3 grid((unsigned)cdiv(a.F, a.fchunk), (unsigned)C::NSPLIT);
#if VAENPVC_PROF
  static int slot = -1;
  if (slot < 0) {
    slot = g_wg_prof_next++;
    fprintf(stderr, "WPROF slot %d grid %u x %u fchunk %d lds %d : %s\n", slot, grid.x, grid.y, a.fchunk, LDS_BYTES, __PRETTY_FUNCTION__);
  }
  a.slot = slot;
#endif
  hipLaunchKernelGGL(k_convwgrad<C>, grid, dim3(C::NTHR), LDS_BYTES, s, a);
}

}  // namespace tuned
}  // namespace vaenpvc
