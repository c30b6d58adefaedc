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

namespace vaenpvc {
namespace tuned {

constexpr int odd_up(int v) { return v | 1; }

template <int XC_, int XH_, int YC_, int YH_, int T_, int S_, int PAD_, bool XLN_, bool YLN_, int TF_, int NTW_>
struct WgCfg {
  static constexpr int XC = XC_, XH = XH_, YC = YC_, YH = YH_, T = T_, S = S_, PAD = PAD_, TF = TF_, NTW = NTW_;
  static constexpr bool XLN = XLN_, YLN = YLN_;
  static constexpr int M = T * XC, MTL = cdiv(M, 32);
  static constexpr int NTL = cdiv(YC, 32), NSPLIT = cdiv(NTL, NTW);
  static constexpr int WM = MTL >= 4 ? 4 : (MTL >= 2 ? 2 : 1), WK = 4 / WM, MTW = cdiv(MTL, WM);
  static constexpr int HLO = PAD, HHI = cmax(0, S * (YH - 1) - PAD + T - 1 - (XH - 1));
  // A gathers: lane <-> m = (t, xc) reads xc*CSTRX + t.  XC >= 32: one tap per fragment, odd
  // stride.  XC < 32: a fragment spans 32/XC taps, stride == 32/XC (mod 32) keeps all 32 banks distinct.
  static constexpr int CSTRX = XC >= 32 ? odd_up(HLO + XH + HHI) : next_mod32(HLO + XH + HHI, 32 / XC);
  static constexpr int CSTRY = odd_up(YH);
  static constexpr int FSTRX = XC * CSTRX, FSTRY = NTW * 32 * CSTRY;
  static constexpr int XT = rup(TF * FSTRX, 4), YT = rup(TF * FSTRY, 4);
  static constexpr int LDS_BYTES = (XT + YT) * 4;
  static_assert(TF % 2 == 0, "k-steps pair two frames");
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
};

// Row-wise staging (see stage_rows); here the scale/shift is ALWAYS applied so that rows outside
// the chunk / channel range are written as zeros (sc = sh = 0).
template <int ROWLEN, bool LN, class RowInfo>
__device__ __forceinline__ void wg_stage_rows(const float* __restrict__ src, float* __restrict__ dst, int nrows,
                                              RowInfo&& rowinfo) {
  constexpr int PER = (ROWLEN + 63) / 64;
  constexpr int RU = cmax_c(2, cmin_c(16, 32 / PER));  // ~32 loads in flight per lane
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int r0 = wave * RU; r0 < nrows; r0 += 4 * RU) {
    float v[RU][PER];
    int doff[RU];
    float sc[RU], sh[RU];
#pragma unroll
    for (int u = 0; u < RU; ++u) {
      int r = r0 + u < nrows ? r0 + u : nrows - 1;
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
        float x = v[u][p] * sc[u] + sh[u];
        if constexpr (LN) x = fmaxf(x, LEAK * x);
        if (i < ROWLEN) dst[doff[u] + i] = x;
      }
  }
}
template <class C, class RowInfo>
__device__ __forceinline__ void stage_rows_x(const float* src, float* dst, int nrows, int, RowInfo&& ri) {
  wg_stage_rows<C::XH, C::XLN>(src, dst, nrows, ri);
}
template <class C, class RowInfo>
__device__ __forceinline__ void stage_rows_y(const float* src, float* dst, int nrows, int, RowInfo&& ri) {
  wg_stage_rows<C::YH, C::YLN>(src, dst, nrows, ri);
}

template <class C>
__global__ void __launch_bounds__(256) k_convwgrad(WgArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tX = lds;
  float* tY = lds + C::XT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave % C::WM, wk = wave / C::WM;
  const int nc0 = blockIdx.y * C::NTW * 32;  // first Y channel of this workgroup
  for (int i = tid; i < (C::XT + C::YT) / 4; i += 256) reinterpret_cast<float4*>(lds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);

  int baseA[C::MTW];
  bool aok[C::MTW];
#pragma unroll
  for (int i = 0; i < C::MTW; ++i) {
    int m = (wm + i * C::WM) * 32 + l31;
    aok[i] = m < C::M;
    int mm = aok[i] ? m : 0;
    int t = mm / C::XC, xc = mm - t * C::XC;
    baseA[i] = lh * C::FSTRX + xc * C::CSTRX + C::HLO - C::PAD + t;
  }
  int baseB[C::NTW];
#pragma unroll
  for (int j = 0; j < C::NTW; ++j) baseB[j] = lh * C::FSTRY + (j * 32 + l31) * C::CSTRY;
  f32x16 acc[C::MTW][C::NTW];
#pragma unroll
  for (int i = 0; i < C::MTW; ++i)
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) acc[i][j] = zero16();

  const int fb = blockIdx.x * a.fchunk;
  const int fe = min(a.F, fb + a.fchunk);
  constexpr int XPER = C::XC * C::XH, YPER = C::NTW * 32 * C::YH;
  for (int f0 = fb; f0 < fe; f0 += C::TF) {
    __syncthreads();  // previous sub-tile consumed (first pass: zero fill done)
    const int nfr = min(C::TF, fe - f0);
    // Both tiles are staged row-wise (row = one channel of one frame, wave-uniform index);
    // frames past the chunk end and channels past YC are clamped on load and zeroed by scale 0.
    if constexpr (C::XH >= 32) {
      const float* src = a.X + (int64_t)f0 * XPER;
      auto rowx = [&](int r, int& soff, int& doff, float& sc, float& sh) {
        int f = r / C::XC, xc = r - f * C::XC;
        bool ok = f < nfr;
        int fs = ok ? f : 0;
        soff = (fs * C::XC + xc) * C::XH;
        doff = f * C::FSTRX + xc * C::CSTRX + C::HLO;
        sc = ok ? 1.f : 0.f;
        sh = 0.f;
        if constexpr (C::XLN) {
          float mean = a.xst[2 * (f0 + fs)], rstd = a.xst[2 * (f0 + fs) + 1];
          sc = ok ? rstd * a.xg[xc] : 0.f;
          sh = ok ? a.xb[xc] - mean * sc : 0.f;
        }
      };
      stage_rows_x<C>(src, tX, C::TF * C::XC, nfr, rowx);
    } else
    {  // X tile: nfr whole frames are one contiguous HBM range
      auto putx = [&](int e, float v) {
        int f = e / XPER, rem = e - f * XPER;
        int xc = rem / C::XH, i = rem - xc * C::XH;
        if constexpr (C::XLN) v = lnact_v(v, a.xst[2 * (f0 + f)], a.xst[2 * (f0 + f) + 1], a.xg[xc], a.xb[xc]);
        tX[f * C::FSTRX + xc * C::CSTRX + C::HLO + i] = v;
      };
      stage_range<(XPER % 4 == 0) ? 4 : 1, 8>(a.X + (int64_t)f0 * XPER, nfr * XPER, putx);
      if (nfr < C::TF)  // tail of the chunk: frames beyond it must read as zero
        for (int e = nfr * XPER + tid; e < C::TF * XPER; e += 256) {
          int f = e / XPER, rem = e - f * XPER;
          int xc = rem / C::XH, i = rem - xc * C::XH;
          tX[f * C::FSTRX + xc * C::CSTRX + C::HLO + i] = 0.f;
        }
    }
    if constexpr (C::YH >= 32) {
      constexpr int YFR = C::YC * C::YH;
      const float* src = a.Y + (int64_t)f0 * YFR + (int64_t)nc0 * C::YH;
      const int ych = min(C::NTW * 32, C::YC - nc0);  // valid channels of this workgroup
      auto rowy = [&](int r, int& soff, int& doff, float& sc, float& sh) {
        int f = r / (C::NTW * 32), nl = r - f * (C::NTW * 32);
        bool ok = f < nfr && nl < ych;
        int fs = ok ? f : 0, ns = ok ? nl : 0;
        soff = fs * YFR + ns * C::YH;
        doff = f * C::FSTRY + nl * C::CSTRY;
        sc = ok ? 1.f : 0.f;
        sh = 0.f;
        if constexpr (C::YLN) {
          float mean = a.yst[2 * (f0 + fs)], rstd = a.yst[2 * (f0 + fs) + 1];
          sc = ok ? rstd * a.yg[nc0 + ns] : 0.f;
          sh = ok ? a.yb[nc0 + ns] - mean * sc : 0.f;
        }
      };
      stage_rows_y<C>(src, tY, C::TF * C::NTW * 32, nfr, rowy);
    } else
    {  // Y tile: per frame the valid channels of this workgroup are one contiguous run
      constexpr int YFR = C::YC * C::YH;              // floats per frame of Y
      const int ych = min(C::NTW * 32, C::YC - nc0);  // valid channels
      const int yper = ych * C::YH;
      const int ytot = C::TF * yper;
      constexpr int BT = 16;
      for (int e0 = tid; e0 < ytot; e0 += 256 * BT) {
        float v[BT];
#pragma unroll
        for (int b = 0; b < BT; ++b) {
          int e = e0 + 256 * b;
          int f = e / yper, rem = e - f * yper;
          v[b] = (e < ytot && f < nfr) ? a.Y[(int64_t)(f0 + f) * YFR + (int64_t)nc0 * C::YH + rem] : 0.f;
        }
#pragma unroll
        for (int b = 0; b < BT; ++b) {
          int e = e0 + 256 * b;
          if (e < ytot) {
            int f = e / yper, rem = e - f * yper;
            int nl = rem / C::YH, r = rem - nl * C::YH;
            float x = v[b];
            if constexpr (C::YLN) {
              if (f < nfr) x = lnact_v(x, a.yst[2 * (f0 + f)], a.yst[2 * (f0 + f) + 1], a.yg[nc0 + nl], a.yb[nc0 + nl]);
            }
            tY[f * C::FSTRY + nl * C::CSTRY + r] = x;
          }
        }
      }
    }
    __syncthreads();
    // k-steps: position r (outer) x frame pair fp (inner, unrolled); the WK waves that share
    // an M range take interleaved positions r.  Fragment reads of one r are all issued before
    // its MFMAs (addresses differ only by compile-time offsets).
    constexpr int HP = C::TF / 2;
    for (int r = wk; r < C::YH; r += C::WK) {
      float av[HP][C::MTW], bv[HP][C::NTW];
#pragma unroll
      for (int fp = 0; fp < HP; ++fp) {
#pragma unroll
        for (int i = 0; i < C::MTW; ++i) {
          float v = tX[baseA[i] + C::S * r + fp * 2 * C::FSTRX];
          av[fp][i] = aok[i] ? v : 0.f;
        }
#pragma unroll
        for (int j = 0; j < C::NTW; ++j) bv[fp][j] = tY[baseB[j] + r + fp * 2 * C::FSTRY];
      }
#pragma unroll
      for (int fp = 0; fp < HP; ++fp)
#pragma unroll
        for (int i = 0; i < C::MTW; ++i)
#pragma unroll
          for (int j = 0; j < C::NTW; ++j) acc[i][j] = mfma32(av[fp][i], bv[fp][j], acc[i][j]);
    }
  }
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

template <class C>
inline void launch_convwgrad(const WgArgs& a0, int target_wgs, hipStream_t s) {
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_convwgrad<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              C::LDS_BYTES);
    once = true;
  }
  WgArgs a = a0;
  int chunks = cmax(1, target_wgs / C::NSPLIT);
  a.fchunk = rup(cmax(1, cdiv(a.F, chunks)), C::TF);
  dim3 grid((unsigned)cdiv(a.F, a.fchunk), (unsigned)C::NSPLIT);
  hipLaunchKernelGGL(k_convwgrad<C>, grid, dim3(256), C::LDS_BYTES, s, a);
}

}  // namespace tuned
}  // namespace vaenpvc
