// gfx950_fbwd.h -- the WHOLE backward step of a thin decoder layer in one kernel (round 4): LayerNorm + lrelu backward,
// input gradient and weight gradient, the layer's five parameter-gradient tensors.  The layered path ran three kernels
// per layer (k_ln_bwd_fused -> HBM -> k_fconv input gradient + k_fwgrad weight gradient) and moved the gradient at the
// layer's pre-LN output three times (one write, two reads); here it never leaves the chip:
//   1. a workgroup owns TF whole frames.  The gradient at the layer's ACTIVATED output (dy) and the pre-LN output (a) are
//      read once, coalesced, a lane owning one position and walking the channels in registers (the staging layout of
//      gfx950_fconv.h).  The LayerNorm backward (autodiff of util/layers.py:32-44,149, written out above k_ln_bwd_fused in
//      gfx950_elem.h) runs in those registers: two sums per frame through wave reductions + LDS partials, then
//      du = rstd (dn gamma - mean(dn gamma) - xhat mean(dn gamma xhat)); the per-channel sums for d gamma / d beta / d bias
//      are carried per lane to the end of the kernel.
//   2. du goes, split into NPL bf16 terms, channel-last with the conv's zero rows in front into LDS -- ONE image that is
//      both the S-type view operand of the input-gradient site (gfx950_viewconv.h: CV_D*G) and the view operand of the
//      weight-gradient site (CW_D*); the activated input of the layer (lrelu(LN(output of the layer below)), rebuilt on
//      load) is the weight gradient's plain-row operand.
//   3. both GEMMs run from LDS on the bf16 matrix cores: the input gradient leaves as canonical fp32 [F][C][H] (it is
//      the dy of the layer below), the weight-gradient tile stays in the accumulators of the persistent workgroup until
//      one flush of atomics.
// HBM traffic per frame: dy + a + input activation + input gradient (decoder layer 2: 54.6 KB against 103.8 KB).
// Reference: autodiff of model/vae.py:96-102 (conv2d_transpose + Layernorm + lrelu), trainer/vae.py:24.
#pragma once
#include "gfx950_fwgrad.h"

namespace vaenpvc {
namespace tuned {

enum { FB_D2, FB_D1, FB_COUNT };
constexpr int fb_gsite(int l) { return l == FB_D2 ? CV_D2G : CV_D1G; }
constexpr int fb_wsite(int l) { return l == FB_D2 ? CW_D2 : CW_D1; }

// frames per group and register prefetch of the next group, per layer.  The staging registers hold a lane's position of
// every channel of both tensors (dy, a) for all items of the wave: with two frames per group, or with one frame and the
// next group's loads in flight during the GEMMs, decoder layer 2 does not fit 256 registers (measured: 474 / 39 spilled).
#ifndef VAENPVC_FB_TF
#define VAENPVC_FB_TF 1
#endif
#ifndef VAENPVC_FB_PREFETCH
#define VAENPVC_FB_PREFETCH 0
#endif
template <int NPL, int L>
struct FbCfg {
  static constexpr CvSite V = CVS[fb_gsite(L)];
  static constexpr CwSite WS = CWS[fb_wsite(L)];
  static constexpr ClDesc UD = CLD[WS.b], XD = CLD[WS.a];    // U: gradient at this layer's pre-LN output; X: its activated input
  static constexpr int CU = UD.C, HU = UD.H, CX = XD.C, HX = XD.H, CPX = XD.CP;
  static constexpr int R = WS.R, R16 = rup(R, 16), TAPS = WS.T, S = 3, PAD = UD.HLO, NU = CU * HU;
  static constexpr int CPLU = (CU == 32 || CU == 64 || CU == 128) ? CU + 8 : CU;
  static constexpr int CPLX = (CPX == 32 || CPX == 64 || CPX == 128) ? CPX + 8 : CPX;
  static constexpr int ROWSU = S * (R16 - 1) + TAPS + 1;
  static constexpr int FSU = ROWSU * CPLU, FSX = R16 * CPLX;       // elements per frame
  static constexpr int TF = VAENPVC_FB_TF;
  static constexpr bool PREFETCH = VAENPVC_FB_PREFETCH != 0;
  static constexpr int UPL = TF * FSU + 64, XPL = TF * FSX;        // elements per plane
  // input-gradient site (S-type view of U)
  static constexpr int KS = cdiv(V.NT * CU, 16), MT = cdiv(V.M, 32), WP = V.Kp + 8, WPL = MT * 32 * WP;
  static constexpr int RSTEP = (V.step / CU) * CPLU;
  // weight-gradient tile: rows n = (tap, channel of U), columns m = channel of X
  static constexpr int N = TAPS * CU, M = WS.M, NT = cdiv(N, 32), MTW = cdiv(M, 32);
  static constexpr int KSPLIT = NT <= 2 ? 2 : 1, WN = 4 / KSPLIT, NTW = cdiv(NT, WN);
  static constexpr int NCHU = cdiv(HU, 64), NITU = TF * NCHU, IPWU = cdiv(NITU, 4);
  static constexpr int LDS = NPL * (UPL + XPL + WPL) * 2;
  static_assert(V.x == WS.b && UD.CP == CU && HX == R && V.R == R && V.PH == 0 && V.M == CX && MT == 1, "layer not served");
  static_assert(KS * 16 <= V.Kp && (S * (R - 1)) * CPLU + KS * 16 <= FSU, "view runs past the frame image");
};

struct FbArgs {
  const float* dy;       // [F][CU][HU] gradient at the layer's activated output
  const float* a;        // [F][CU][HU] the layer's pre-LN output
  const float* st;       // its LayerNorm statistics (mean, rstd) per frame
  const float* gamma;    // [CU]
  const float* beta;
  const float* xa;       // [F][CX][HX] pre-LN output of the layer below
  const float* xst;
  const float* xgamma;   // [CX]
  const float* xbeta;
  const unsigned short* W;   // weight planes of the input-gradient site [NPL][Mp][Kp] (cv_job)
  float* dx;             // [F][CX][HX] out: gradient at the activated output of the layer below
  float* dW;             // [TAPS * CU][CX], atomicAdd
  float* dgamma;         // [CU], atomicAdd
  float* dbeta;
  float* dbias;
  int F;
};

template <int NPL, int L>
__global__ void __launch_bounds__(256, 2) k_fbwd(FbArgs a) {
  using T = FbCfg<NPL, L>;
  constexpr CvSite V = T::V;
  constexpr int CU = T::CU, HU = T::HU, NCHU = T::NCHU, NITU = T::NITU, IPWU = T::IPWU;
  extern __shared__ __attribute__((aligned(16))) unsigned short bsm[];
  __shared__ float part[2][FbCfg<NPL, L>::NITU];
  __shared__ float red[4][3 * FbCfg<NPL, L>::CU];
  unsigned short* us = bsm;                       // [NPL][UPL]  du, channel-last, PAD zero rows in front
  unsigned short* xs = bsm + NPL * T::UPL;        // [NPL][XPL]  activated input, plain rows
  unsigned short* ws = xs + NPL * T::XPL;         // [NPL][32][WP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int ngroups = cdiv(a.F, T::TF);
  float vd[IPWU][CU], va[IPWU][CU], mean[IPWU], rstd[IPWU];
  bool uok[IPWU];
  FwStage<NPL, T::CX, T::CPX, T::CPLX, T::HX, T::TF, T::FSX, 0, T::XPL> sx;
  float su[CU], sw[CU], sd[CU];
#pragma unroll
  for (int c = 0; c < CU; ++c) su[c] = sw[c] = sd[c] = 0.f;

  auto uload = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < IPWU; ++u) {
      const int it = wave + 4 * u, fl = it / NCHU, k = it - fl * NCHU;
      const int f = g * T::TF + fl, h = 64 * k + lane;
      const bool fok = it < NITU && f < a.F;
      uok[u] = fok;
      // (addresses clamped into the tensor instead of predicated loads: no branch per load, the loads issue back to back)
      const int64_t fo = (int64_t)(fok ? f : 0) * T::NU + (h < HU ? h : HU - 1);
      mean[u] = a.st[2 * (fok ? f : 0)];
      rstd[u] = a.st[2 * (fok ? f : 0) + 1];
      const float* pd = a.dy + fo;
      const float* pa = a.a + fo;
#pragma unroll
      for (int c = 0; c < CU; ++c) {
        vd[u][c] = pd[c * HU];
        va[u][c] = pa[c * HU];
      }
      const bool ok = fok && h < HU;
#pragma unroll
      for (int c = 0; c < CU; ++c) {
        vd[u][c] = ok ? vd[u][c] : 0.f;
        va[u][c] = ok ? va[u][c] : mean[u];
      }
    }
  };
  // LayerNorm + lrelu backward, first half: dn = dy lrelu'(n), xhat, and the frame's two sums (partials per item)
  auto upass1 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < IPWU; ++u) {
      const int it = wave + 4 * u;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < CU; ++c) {
        const float xh = (va[u][c] - mean[u]) * rstd[u];
        const float nn = xh * a.gamma[c] + a.beta[c];
        const float dn = vd[u][c] * (nn >= 0.f ? 1.0f : LEAK);
        const float dxh = dn * a.gamma[c];
        s1 += dxh;
        s2 += dxh * xh;
        vd[u][c] = dn;
        va[u][c] = xh;
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0 && it < NITU) {
        part[0][it] = s1;
        part[1][it] = s2;
      }
    }
  };
  // second half: du, the per-channel sums, and du as bf16 terms into the U image (frames past the batch end: zeros)
  auto upass2 = [&]() __attribute__((always_inline)) {
    constexpr float INVN = 1.0f / T::NU;
#pragma unroll
    for (int u = 0; u < IPWU; ++u) {
      const int it = wave + 4 * u, fl = (it < NITU ? it : 0) / NCHU, k = it - fl * NCHU;
      const int h = 64 * k + lane;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int kk = 0; kk < NCHU; ++kk) {
        s1 += part[0][fl * NCHU + kk];
        s2 += part[1][fl * NCHU + kk];
      }
      s1 *= INVN;
      s2 *= INVN;
      const bool pos_ok = it < NITU && h < HU, live = pos_ok && uok[u];
#pragma unroll
      for (int c = 0; c < CU; ++c) {
        const float d = rstd[u] * (vd[u][c] * a.gamma[c] - s1 - va[u][c] * s2);
        su[c] += live ? vd[u][c] * va[u][c] : 0.f;
        sw[c] += live ? vd[u][c] : 0.f;
        sd[c] += live ? d : 0.f;
        vd[u][c] = live ? d : 0.f;
      }
      if (!pos_ok) continue;
      unsigned short* dxp = us + fl * T::FSU + (T::PAD + h) * T::CPLU;
#pragma unroll
      for (int g8 = 0; g8 < CU / 8; ++g8) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = vd[u][8 * g8 + j];
        u32x4 pk[NPL];
        pack8<NPL>(v8, pk);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dxp + p * T::UPL + 8 * g8) = pk[p];
      }
    }
  };

  int g = blockIdx.x;
  if (T::PREFETCH && g < ngroups) {
    uload(g);
    sx.load(a.xa, a.xst, g, a.F, wave, lane);
  }
  {  // once per workgroup: zero both images (pad rows, rows past R, tails stay zero), copy the input-gradient weights
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < NPL * (T::UPL + T::XPL) / 8; i += 256) reinterpret_cast<u32x4*>(bsm)[i] = z;
    constexpr int WROW8 = V.Kp / 8, WPIECES = NPL * 32 * WROW8;
    for (int i = tid; i < WPIECES; i += 256) {
      const int p = i / (32 * WROW8), r = i - p * (32 * WROW8), m = r / WROW8, c8 = r - m * WROW8;
      *reinterpret_cast<u32x4*>(ws + p * T::WPL + m * T::WP + c8 * 8) =
          *reinterpret_cast<const u32x4*>(a.W + ((size_t)p * V.Mp + m) * V.Kp + c8 * 8);
    }
  }
  __syncthreads();
  // weight-gradient tiles of this wave (gfx950_fwgrad.h): n tiles wn, wn + WN, ...; k-chunks of parity kpar
  const int wn = wave % T::WN, kpar = wave / T::WN;
  f32x16 wacc[T::NTW][T::MTW];
#pragma unroll
  for (int i = 0; i < T::NTW; ++i)
#pragma unroll
    for (int j = 0; j < T::MTW; ++j) wacc[i][j] = zero16();
  const int trow = ((lane & 15) >> 2) + 8 * lh, tcol = 4 * (lane & 3) + 16 * ((lane >> 4) & 1);
  int bcol[T::NTW], acol[T::MTW];
#pragma unroll
  for (int i = 0; i < T::NTW; ++i) {
    int n = 32 * (wn + i * T::WN) + tcol;
    n = n < T::N ? n : 0;
    bcol[i] = (n / CU) * T::CPLU + n % CU;
  }
#pragma unroll
  for (int j = 0; j < T::MTW; ++j) {
    const int m = 32 * j + tcol;
    acol[j] = m < T::CPX ? m : 0;
  }
  const int woff = l31 * T::WP + lh * 8;
  for (; g < ngroups; g += gridDim.x) {
    const int f0 = g * T::TF, nf = min(T::TF, a.F - f0);
    if (!T::PREFETCH) {   // (unconditional: the staging registers must be dead across the GEMMs, not loop-carried)
      uload(g);
      sx.load(a.xa, a.xst, g, a.F, wave, lane);
    }
    upass1();
    sx.store(xs, true, a.xgamma, a.xbeta, wave, lane);
    __syncthreads();   // the partial sums of every item are visible
    upass2();
    __syncthreads();   // both images are complete
    if (T::PREFETCH && g + (int)gridDim.x < ngroups) {
      uload(g + gridDim.x);
      sx.load(a.xa, a.xst, g + gridDim.x, a.F, wave, lane);
    }
    // ---- input gradient: GEMM rows n = fl * R + q (32 per step), steps dealt round-robin to the waves
    const int nrows = nf * V.R, nsteps = cdiv(nrows, 32);
    for (int s = wave; s < nsteps; s += 4) {
      int n = s * 32 + l31;
      const bool nok = n < nrows;
      n = nok ? n : 0;
      const int fl = n / V.R, q = n - fl * V.R;
      const int xoff = fl * T::FSU + q * T::RSTEP;
      f32x16 acc = zero16();
#pragma unroll
      for (int ks = 0; ks < T::KS; ++ks) {
        u32x4 fa[NPL], fb[NPL];
        const int ko = fc_koff<CU, T::CPLU>(ks, lh);
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          fa[p] = *reinterpret_cast<const u32x4*>(ws + p * T::WPL + woff + ks * 16);
          fb[p] = *reinterpret_cast<const u32x4*>(us + p * T::UPL + xoff + ko);
        }
        using PR = Prod<NPL>;
#pragma unroll
        for (int t = 0; t < PR::N; ++t) acc = mfma_bf16(fa[PR::A[t]], fb[PR::B[t]], acc);
      }
      if (nok) {
        float* ob = a.dx + (int64_t)(f0 + fl) * (V.OC * V.OH) + q;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = acc_row(reg, lane);
          if (m < V.M) ob[m * V.OH] = acc[reg];
        }
      }
    }
    // ---- weight gradient: k-chunks (frame, 16 rows j) of parity kpar
    constexpr int CPF = T::R16 / 16;
    for (int kc = kpar; kc < T::TF * CPF; kc += T::KSPLIT) {
      const int fl = kc / CPF, j0 = (kc - fl * CPF) * 16 + trow;
      const unsigned short* pa0 = xs + fl * T::FSX + j0 * T::CPLX;
      const unsigned short* pb0 = us + fl * T::FSU + (T::S * j0) * T::CPLU;
      u32x4 fa[T::MTW][NPL], fb[T::NTW][NPL];
#pragma unroll
      for (int j = 0; j < T::MTW; ++j)
#pragma unroll
        for (int p = 0; p < NPL; ++p) fa[j][p] = tr_read8_2(pa0 + p * T::XPL + acol[j], pa0 + p * T::XPL + acol[j] + 4 * T::CPLX);
#pragma unroll
      for (int i = 0; i < T::NTW; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          fb[i][p] = tr_read8_2(pb0 + p * T::UPL + bcol[i], pb0 + p * T::UPL + bcol[i] + 4 * T::S * T::CPLU);
      using PR = Prod<NPL>;
#pragma unroll
      for (int t = 0; t < PR::N; ++t)
#pragma unroll
        for (int i = 0; i < T::NTW; ++i)
#pragma unroll
          for (int j = 0; j < T::MTW; ++j) wacc[i][j] = mfma_bf16(fb[i][PR::B[t]], fa[j][PR::A[t]], wacc[i][j]);
    }
    __syncthreads();   // all fragment reads of this group are done before the next one overwrites the images
  }
  // ---- flush: the weight-gradient tile (rows n = (tap, channel of U), lanes m: consecutive addresses of dW[n][m]) ...
#pragma unroll
  for (int i = 0; i < T::NTW; ++i)
#pragma unroll
    for (int j = 0; j < T::MTW; ++j) {
      const int m = 32 * j + l31;
      if (m >= T::M) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = 32 * (wn + i * T::WN) + acc_row(reg, lane);
        if (n < T::N && wn + i * T::WN < T::NT) atomicAdd(a.dW + n * T::M + m, wacc[i][j][reg]);
      }
    }
  // ... and the three per-channel sums: d gamma, d beta (LayerNorm parameters), d bias (the conv's)
#pragma unroll
  for (int c = 0; c < CU; ++c) {
    const float u = wave_sum(su[c]), w = wave_sum(sw[c]), d = wave_sum(sd[c]);
    if (lane == 0) {
      red[wave][c] = u;
      red[wave][CU + c] = w;
      red[wave][2 * CU + c] = d;
    }
  }
  __syncthreads();
  if (tid < 3 * CU) {
    const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    float* dst = tid < CU ? a.dgamma + tid : tid < 2 * CU ? a.dbeta + (tid - CU) : a.dbias + (tid - 2 * CU);
    atomicAdd(dst, v);
  }
}

// ---------------------------------------------------------------- the same step with LDS-DMA landing buffers
// k_fbwd above stages through registers: a workgroup's loads, its LayerNorm passes and its GEMMs run one after the other,
// and what overlaps is what two workgroups per CU happen to interleave (measured 3.2 TB/s of the kernel's own bytes; the
// register prefetch of the next frame does not fit 256 registers beside two frames or the GEMMs).  Here ONE eight-wave
// workgroup per CU keeps the memory system busy all the time instead: the three fp32 tensors of the NEXT frame (dy, a,
// input activation: 43.8 / 29.2 KB) are requested by LDS-DMA (global_load_lds_dwordx4, no registers) into the other of two
// raw landing buffers before the passes of the current frame start; every wave then reads its positions of the current
// frame from LDS (lane = position: conflict-free) and the step continues as above -- passes, bf16 images, both GEMMs.
template <int NPL, int L>
struct FbdCfg : FbCfg<NPL, L> {
  using B = FbCfg<NPL, L>;
  static constexpr int NW = 8;                                      // waves
  static constexpr int NX = B::CX * B::HX;
  static constexpr int RAWF = 2 * B::NU + NX;                       // floats per landing buffer: dy | a | xa
  static constexpr int P1 = B::NU / 4, P2 = NX / 4, PIECES = 2 * P1 + P2;   // 16-byte pieces
  static constexpr int NDMA = cdiv(PIECES, 64 * NW);                // requests per wave and frame
  static constexpr int UPL1 = B::FSU + 64, XPL1 = B::FSX;           // one frame per group
  static constexpr int IMG = NPL * (UPL1 + XPL1 + B::WPL) * 2;      // bytes: U image, X image, weights
  static constexpr int LDS = 2 * RAWF * 4 + IMG;
  static constexpr int NCHX = cdiv(B::HX, 64);
  static constexpr int IPWU = cdiv(B::NCHU, NW), IPWX = cdiv(NCHX, NW);
  static constexpr int KSPLIT = B::NT <= 2 ? 4 : 2, WN = NW / KSPLIT, NTW = cdiv(B::NT, WN);
  static_assert(B::NU % 4 == 0 && NX % 4 == 0 && (RAWF * 4) % 16 == 0, "frames are whole 16-byte pieces");
  static_assert(LDS <= 160 * 1024, "landing buffers + images must fit the LDS");
};

template <int NPL, int L>
__global__ void __launch_bounds__(512, 1) k_fbwd_dma(FbArgs a) {
  using T = FbdCfg<NPL, L>;
  constexpr CvSite V = T::V;
  constexpr int CU = T::CU, HU = T::HU, NCHU = T::NCHU, IPWU = T::IPWU, NW = T::NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char dsm[];
  __shared__ float part[2][FbCfg<NPL, L>::NCHU];
  __shared__ float red[FbdCfg<NPL, L>::NW][3 * FbCfg<NPL, L>::CU];
  float* raw = reinterpret_cast<float*>(dsm);                                          // [2][RAWF]
  unsigned short* us = reinterpret_cast<unsigned short*>(dsm + 2 * T::RAWF * 4);       // [NPL][UPL1]
  unsigned short* xs = us + NPL * T::UPL1;                                             // [NPL][XPL1]
  unsigned short* ws = xs + NPL * T::XPL1;                                             // [NPL][32][WP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const unsigned raw_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dsm;
  float su[CU], sw[CU], sd[CU];
#pragma unroll
  for (int c = 0; c < CU; ++c) su[c] = sw[c] = sd[c] = 0.f;
  // request frame f into landing buffer `slot`: piece p of the frame's 16-byte pieces (dy | a | xa) lands at 16 p
  auto dma = [&](int f, int slot) __attribute__((always_inline)) {
    const unsigned char* g0 = reinterpret_cast<const unsigned char*>(a.dy + (int64_t)f * T::NU);
    const unsigned char* g1 = reinterpret_cast<const unsigned char*>(a.a + (int64_t)f * T::NU);
    const unsigned char* g2 = reinterpret_cast<const unsigned char*>(a.xa + (int64_t)f * T::NX);
#pragma unroll
    for (int k = 0; k < T::NDMA; ++k) {
      const int pb = (k * NW + wave) * 64, p = pb + lane;
      if (pb >= T::PIECES) break;                       // (wave-uniform)
      const unsigned char* src = p < T::P1 ? g0 + 16 * (int64_t)p : p < 2 * T::P1 ? g1 + 16 * (int64_t)(p - T::P1) : g2 + 16 * (int64_t)(p - 2 * T::P1);
      if (p < T::PIECES) lds_dma16(src, raw_base + (unsigned)(slot * T::RAWF * 4 + pb * 16));
    }
  };
  int f = blockIdx.x;
  if (f < a.F) dma(f, 0);
  {  // once per workgroup: zero both images (pad rows, rows past R, tails stay zero), copy the input-gradient weights
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < NPL * (T::UPL1 + T::XPL1) / 8; i += 64 * NW) reinterpret_cast<u32x4*>(us)[i] = z;
    constexpr int WROW8 = V.Kp / 8, WPIECES = NPL * 32 * WROW8;
    for (int i = tid; i < WPIECES; i += 64 * NW) {
      const int p = i / (32 * WROW8), r = i - p * (32 * WROW8), m = r / WROW8, c8 = r - m * WROW8;
      *reinterpret_cast<u32x4*>(ws + p * T::WPL + m * T::WP + c8 * 8) =
          *reinterpret_cast<const u32x4*>(a.W + ((size_t)p * V.Mp + m) * V.Kp + c8 * 8);
    }
  }
  const int wn = wave % T::WN, kpar = wave / T::WN;
  f32x16 wacc[T::NTW][T::MTW];
#pragma unroll
  for (int i = 0; i < T::NTW; ++i)
#pragma unroll
    for (int j = 0; j < T::MTW; ++j) wacc[i][j] = zero16();
  const int trow = ((lane & 15) >> 2) + 8 * lh, tcol = 4 * (lane & 3) + 16 * ((lane >> 4) & 1);
  int bcol[T::NTW], acol[T::MTW];
#pragma unroll
  for (int i = 0; i < T::NTW; ++i) {
    int n = 32 * (wn + i * T::WN) + tcol;
    n = n < T::N ? n : 0;
    bcol[i] = (n / CU) * T::CPLU + n % CU;
  }
#pragma unroll
  for (int j = 0; j < T::MTW; ++j) {
    const int m = 32 * j + tcol;
    acol[j] = m < T::CPX ? m : 0;
  }
  const int woff = l31 * T::WP + lh * 8;
  int slot = 0;
  for (; f < a.F; f += gridDim.x, slot ^= 1) {
    wait_vmcnt<0>();   // this wave's pieces of frame f have landed (and its result stores of the frame before are done)
    __syncthreads();   // ... and everybody else's; the images are free (all GEMMs of the frame before are finished)
    if (f + (int)gridDim.x < a.F) dma(f + gridDim.x, slot ^ 1);
    const float* rd = raw + slot * T::RAWF;      // dy
    const float* ra = rd + T::NU;                // a
    const float* rx = ra + T::NU;                // xa
    const float mean = a.st[2 * f], rstd = a.st[2 * f + 1];
    // ---- LayerNorm + lrelu backward, first half (items = 64-position chunks dealt to the waves)
    float vd[IPWU][CU], va[IPWU][CU];
#pragma unroll
    for (int u = 0; u < IPWU; ++u) {
      const int it = wave + NW * u, h = 64 * it + lane;
      const bool ok = it < NCHU && h < HU;
      const int hh = ok ? h : 0;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < CU; ++c) {
        const float dyv = ok ? rd[c * HU + hh] : 0.f, av = ok ? ra[c * HU + hh] : mean;
        const float xh = (av - mean) * rstd;
        const float nn = xh * a.gamma[c] + a.beta[c];
        const float dn = dyv * (nn >= 0.f ? 1.0f : LEAK);
        const float dxh = dn * a.gamma[c];
        s1 += dxh;
        s2 += dxh * xh;
        vd[u][c] = dn;
        va[u][c] = xh;
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0 && it < NCHU) {
        part[0][it] = s1;
        part[1][it] = s2;
      }
    }
    // ---- the activated input of the layer -> X image (plain rows)
    {
      const float xmean = a.xst[2 * f], xrstd = a.xst[2 * f + 1];
#pragma unroll
      for (int u = 0; u < T::IPWX; ++u) {
        const int it = wave + NW * u, h = 64 * it + lane;
        if (!(it < T::NCHX && h < T::HX)) continue;
        unsigned short* dxp = xs + h * T::CPLX;
#pragma unroll
        for (int g8 = 0; g8 < T::CPX / 8; ++g8) {
          float v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = 8 * g8 + j;
            v8[j] = c < T::CX ? lnact_v(rx[c * T::HX + h], xmean, xrstd, a.xgamma[c], a.xbeta[c]) : 0.f;
          }
          u32x4 pk[NPL];
          pack8<NPL>(v8, pk);
#pragma unroll
          for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dxp + p * T::XPL1 + 8 * g8) = pk[p];
        }
      }
    }
    __syncthreads();   // the partial sums of every item are visible
    {
      constexpr float INVN = 1.0f / T::NU;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int kk = 0; kk < NCHU; ++kk) {
        s1 += part[0][kk];
        s2 += part[1][kk];
      }
      s1 *= INVN;
      s2 *= INVN;
#pragma unroll
      for (int u = 0; u < IPWU; ++u) {
        const int it = wave + NW * u, h = 64 * it + lane;
        const bool live = it < NCHU && h < HU;
#pragma unroll
        for (int c = 0; c < CU; ++c) {
          const float d = rstd * (vd[u][c] * a.gamma[c] - s1 - va[u][c] * s2);
          su[c] += live ? vd[u][c] * va[u][c] : 0.f;
          sw[c] += live ? vd[u][c] : 0.f;
          sd[c] += live ? d : 0.f;
          vd[u][c] = d;
        }
        if (!live) continue;
        unsigned short* dxp = us + (T::PAD + h) * T::CPLU;
#pragma unroll
        for (int g8 = 0; g8 < CU / 8; ++g8) {
          float v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = vd[u][8 * g8 + j];
          u32x4 pk[NPL];
          pack8<NPL>(v8, pk);
#pragma unroll
          for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dxp + p * T::UPL1 + 8 * g8) = pk[p];
        }
      }
    }
    __syncthreads();   // both images are complete
    // ---- input gradient: GEMM rows q (32 per step), steps dealt round-robin to the waves
    constexpr int NSTEPS = cdiv(V.R, 32);
    for (int s = wave; s < NSTEPS; s += NW) {
      int q = s * 32 + l31;
      const bool nok = q < V.R;
      q = nok ? q : 0;
      const int xoff = q * T::RSTEP;
      f32x16 acc = zero16();
#pragma unroll
      for (int ks = 0; ks < T::KS; ++ks) {
        u32x4 fa[NPL], fb[NPL];
        const int ko = fc_koff<CU, T::CPLU>(ks, lh);
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          fa[p] = *reinterpret_cast<const u32x4*>(ws + p * T::WPL + woff + ks * 16);
          fb[p] = *reinterpret_cast<const u32x4*>(us + p * T::UPL1 + xoff + ko);
        }
        using PR = Prod<NPL>;
#pragma unroll
        for (int t = 0; t < PR::N; ++t) acc = mfma_bf16(fa[PR::A[t]], fb[PR::B[t]], acc);
      }
      if (nok) {
        float* ob = a.dx + (int64_t)f * (V.OC * V.OH) + q;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = acc_row(reg, lane);
          if (m < V.M) ob[m * V.OH] = acc[reg];
        }
      }
    }
    // ---- weight gradient: k-chunks (16 rows j) of parity kpar
    constexpr int CPF = T::R16 / 16;
    for (int kc = kpar; kc < CPF; kc += T::KSPLIT) {
      const int j0 = kc * 16 + trow;
      const unsigned short* pa0 = xs + j0 * T::CPLX;
      const unsigned short* pb0 = us + (T::S * j0) * T::CPLU;
      u32x4 fa[T::MTW][NPL], fb[T::NTW][NPL];
#pragma unroll
      for (int j = 0; j < T::MTW; ++j)
#pragma unroll
        for (int p = 0; p < NPL; ++p) fa[j][p] = tr_read8_2(pa0 + p * T::XPL1 + acol[j], pa0 + p * T::XPL1 + acol[j] + 4 * T::CPLX);
#pragma unroll
      for (int i = 0; i < T::NTW; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          fb[i][p] = tr_read8_2(pb0 + p * T::UPL1 + bcol[i], pb0 + p * T::UPL1 + bcol[i] + 4 * T::S * T::CPLU);
      using PR = Prod<NPL>;
#pragma unroll
      for (int t = 0; t < PR::N; ++t)
#pragma unroll
        for (int i = 0; i < T::NTW; ++i)
#pragma unroll
          for (int j = 0; j < T::MTW; ++j) wacc[i][j] = mfma_bf16(fb[i][PR::B[t]], fa[j][PR::A[t]], wacc[i][j]);
    }
  }
  // ---- flush (as k_fbwd)
#pragma unroll
  for (int i = 0; i < T::NTW; ++i)
#pragma unroll
    for (int j = 0; j < T::MTW; ++j) {
      const int m = 32 * j + l31;
      if (m >= T::M) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = 32 * (wn + i * T::WN) + acc_row(reg, lane);
        if (n < T::N && wn + i * T::WN < T::NT) atomicAdd(a.dW + n * T::M + m, wacc[i][j][reg]);
      }
    }
#pragma unroll
  for (int c = 0; c < CU; ++c) {
    const float u = wave_sum(su[c]), w = wave_sum(sw[c]), d = wave_sum(sd[c]);
    if (lane == 0) {
      red[wave][c] = u;
      red[wave][CU + c] = w;
      red[wave][2 * CU + c] = d;
    }
  }
  __syncthreads();
  if (tid < 3 * CU) {
    float v = 0.f;
#pragma unroll
    for (int w8 = 0; w8 < NW; ++w8) v += red[w8][tid];
    float* dst = tid < CU ? a.dgamma + tid : tid < 2 * CU ? a.dbeta + (tid - CU) : a.dbias + (tid - 2 * CU);
    atomicAdd(dst, v);
  }
}

template <int NPL, int L>
static void launch_fbwd(const FbArgs& a, hipStream_t s) {
  if (rt().fb_dma) {   // VAENPVC_FB_DMA=0: the register-staged kernel (A/B)
    using D = FbdCfg<NPL, L>;
    rt().ensure_lds(reinterpret_cast<const void*>(&k_fbwd_dma<NPL, L>), D::LDS);
    hipLaunchKernelGGL((k_fbwd_dma<NPL, L>), dim3((unsigned)cmin_(a.F, 256)), dim3(512), D::LDS, s, a);
    return;
  }
  using T = FbCfg<NPL, L>;
  rt().ensure_lds(reinterpret_cast<const void*>(&k_fbwd<NPL, L>), T::LDS);
  const unsigned grid = (unsigned)cmin_(cdiv(a.F, T::TF), T::LDS > 78 * 1024 ? 256 : 512);
  hipLaunchKernelGGL((k_fbwd<NPL, L>), dim3(grid), dim3(256), T::LDS, s, a);
}
template <int NPL>
static bool fbwd(int layer, const FbArgs& a, hipStream_t s) {
  if constexpr (NPL <= 2) {
    switch (layer) {
      case FB_D2: launch_fbwd<NPL, FB_D2>(a, s); return true;
      case FB_D1: launch_fbwd<NPL, FB_D1>(a, s); return true;
    }
  }
  return false;
}

}  // namespace tuned
}  // namespace vaenpvc
