// gfx950_fbwd.h -- the WHOLE backward step of a thin conv layer in one kernel (round 4): LayerNorm + lrelu backward,
// input gradient and weight gradient, the layer's five parameter-gradient tensors.  The layered path ran three kernels
// per layer (k_ln_bwd_fused -> HBM -> k_fconv input gradient + k_fwgrad weight gradient) and moved the gradient at the
// layer's pre-LN output three times (one write, two reads); here it never leaves the chip:
//   1. a workgroup owns one whole frame at a time.  The gradient at the layer's ACTIVATED output (dy) and the pre-LN
//      output (a) are read once, coalesced, a lane owning one position and walking a group of 8 channels in registers (the
//      staging layout of gfx950_fconv.h; items = (64-position chunk, channel group) dealt to the four waves).  The
//      LayerNorm backward (autodiff of util/layers.py:32-44,149, written out above k_ln_bwd_fused in gfx950_elem.h) runs
//      in those registers: two sums per frame through wave reductions + LDS partials, then
//      du = rstd (dn gamma - mean(dn gamma) - xhat mean(dn gamma xhat)); the per-channel sums for d gamma / d beta / d bias
//      are carried per lane to the end of the kernel (a wave always owns the same channel group).
//   2. du goes, split into NPL bf16 terms, channel-last with zero halo rows into LDS -- ONE image that serves both GEMMs:
//        decoder layer (conv_transpose): the S-type view operand of the input-gradient site (CV_D*G) and the view operand
//                                        of the weight-gradient site (CW_D*);
//        encoder layer (conv):           the P-type (phase-stacked) view operand of the input-gradient site (CV_E*G) and
//                                        the plain-row operand of the weight-gradient site (CW_E*);
//      the activated input of the layer (lrelu(LN(output of the layer below)), rebuilt on load) is the weight gradient's
//      other operand (plain rows for a decoder layer, the stride-3 view for an encoder layer).
//   3. both GEMMs run from LDS on the bf16 matrix cores: the input gradient leaves as canonical fp32 [F][C][H] (it is
//      the dy of the layer below), the weight-gradient tile stays in the accumulators of the persistent workgroup until
//      one flush of atomics.
// HBM traffic per frame: dy + a + input activation + input gradient (decoder layer 2: 54.6 KB against 103.8 KB).
// Measured and not kept (round 4, same-box A/Bs): two frames per group or a register prefetch of the
// next frame beside decoder layer 2's 80 staging registers (474 / 39 registers spilled); ONE eight-wave workgroup per CU
// whose next frame arrives by LDS-DMA in a second landing buffer while the current one is processed (correct, 724 / 683 us
// against 552 / 385: the per-frame chain scalar loads -> passes -> three barriers -> dependent LDS reads -> MFMAs is ~5 us
// of latency, and one workgroup per CU has nothing to hide it with).
// Reference: autodiff of util/layers.py:47-66 / model/vae.py:96-102 (conv / conv_transpose + Layernorm + lrelu), trainer/vae.py:24.
#pragma once
#include "gfx950_fwgrad.h"

namespace vaenpvc {
namespace tuned {

enum { FB_D2, FB_D1, FB_E1, FB_COUNT };
constexpr int FB_DY2_PITCH = 516;   // padded rows of decoder layer 2's dy (513 bins -> 16-byte aligned rows; cl_layout.h: DY2_PITCH)
constexpr bool fb_enc(int l) { return l == FB_E1; }
constexpr int fb_gsite(int l) { return l == FB_D2 ? CV_D2G : l == FB_D1 ? CV_D1G : CV_E1G; }
constexpr int fb_wsite(int l) { return l == FB_D2 ? CW_D2 : l == FB_D1 ? CW_D1 : CW_E1; }
// channel groups of the LayerNorm-backward items (a lane walks C / groups = 8 channels of its position)
constexpr int fb_cgr(int l) { return l == FB_D2 ? 1 : l == FB_D1 ? 2 : 4; }
// register prefetch of the next frame during the GEMMs, per layer: 3 = everything (dy, a, the input activation), 2 = the pre-LN
// tensor a only, 1 = dy only (decoder layer 2: all of it does not fit 256 registers; its halves fit and change nothing: 514 - 526 us
// with none / dy / a prefetched, same box), 0 = nothing
#ifndef VAENPVC_FB_PFW
#define VAENPVC_FB_PFW 0x330   // hex digit l = what layer FB_* l prefetches
#endif
constexpr int fb_pfw(int l) { return (VAENPVC_FB_PFW >> (4 * l)) & 0xf; }

#ifndef VAENPVC_FB_OTL
#define VAENPVC_FB_OTL 1   // 0: input gradient stored straight from the accumulators (A/B)
#endif
#ifndef VAENPVC_FB_ABL
#define VAENPVC_FB_ABL 0   // developer ablation (wrong results): 1 no input-gradient GEMM, 2 no weight-gradient GEMM, 4 no result stores, 8 no global loads, 16 no flush of the weight-gradient tile / channel sums
#endif
// which wave stages the input-activation items (item i goes to wave (i - rot) & 3): away from the waves that carry the most
// LayerNorm items and input-gradient steps
#ifndef VAENPVC_FB_XROT
#define VAENPVC_FB_XROT 0x313   // hex digit l = rotation of layer FB_* l
#endif
constexpr int fb_xrot(int l) { return (VAENPVC_FB_XROT >> (4 * l)) & 3; }
constexpr int fb_max(int a, int b) { return a > b ? a : b; }
template <int NPL, int L>
struct FbCfg {
  static constexpr bool ENC = fb_enc(L);
  static constexpr int PFW = fb_pfw(L), XROT = fb_xrot(L);
  static constexpr CvSite V = CVS[fb_gsite(L)];
  static constexpr CwSite WS = CWS[fb_wsite(L)];
  // G: gradient at this layer's pre-LN output (du); X: its activated input
  static constexpr ClDesc GD = CLD[ENC ? WS.a : WS.b], XD = CLD[ENC ? WS.b : WS.a];
  static constexpr int CG = GD.C, HG = GD.H, NG = CG * HG, CX = XD.C, HX = XD.H;
  static constexpr int R = WS.R, R16 = rup(R, 16), TAPS = WS.T, S = 3;
  static constexpr int CPLG = (CG == 32 || CG == 64 || CG == 128) ? CG + 8 : CG;
  static constexpr int CPLX = (CX == 32 || CX == 64 || CX == 128) ? CX + 8 : CX;
  static constexpr int VIEWROWS = S * (R16 - 1) + TAPS + 1;
  static constexpr int ROW0G = GD.HLO, ROWSG = ENC ? fb_max(GD.HP, GD.HLO + R16) : VIEWROWS;
  static constexpr int ROW0X = ENC ? XD.HLO : 0, ROWSX = ENC ? VIEWROWS : R16;
  static constexpr int GPL = ROWSG * CPLG + 64, XPL = ROWSX * CPLX + 64;   // elements per plane (one frame) + zero tail
  // input-gradient site (view of G)
  static constexpr int KS = cdiv(V.NT * CG, 16), MT = cdiv(V.M, 32), WP = V.Kp + 8, WPL = MT * 32 * WP;
  static constexpr int RSTEP = (V.step / CG) * CPLG;
  // weight-gradient tile: rows n = (tap, channel of the VIEW operand), columns m = channel of the PLAIN operand
  static constexpr int CVW = ENC ? CX : CG, CPLV = ENC ? CPLX : CPLG, CPN = ENC ? CG : CX, CPLN = ENC ? CPLG : CPLX;
  static constexpr int N = TAPS * CVW, M = WS.M, NT = cdiv(N, 32), MTW = cdiv(M, 32);
  static constexpr int KSPLIT = NT <= 2 ? 2 : 1, WN = 4 / KSPLIT, NTW = cdiv(NT, WN);
  // LayerNorm-backward items
  static constexpr int CGR = fb_cgr(L), CUG = CG / CGR, NCHG = cdiv(HG, 64), NITG = NCHG * CGR, IPWG = cdiv(NITG, 4);
  // OTL (round 5): the frame's input gradient goes through an LDS tile ([CX][HX] fp32, the canonical order) and leaves as 16-byte ALIGNED
  // pieces of one contiguous run (a frame is a multiple of 16 bytes although its rows of 171 / 57 floats are not): straight from the
  // accumulators it left as 4- / 12-byte stores at unaligned row starts
  static constexpr bool OTL = VAENPVC_FB_OTL != 0;
  // LNB2 (round 5, decoder layer 1): the LayerNorm + lrelu backward of the layer BELOW (decoder layer 0) runs on the input gradient while it
  // sits in the LDS tile -- its normalised pre-LN values were in this kernel's registers anyway (they are the activation the weight gradient
  // multiplies) and are parked in a second fp32 tile; the result leaves as the channel-last operand planes of that layer's gradient GEMMs
  // (CL_GD0), its per-channel sums as one row of `part0` per workgroup.  The gradient at decoder layer 0's activated output never reaches
  // HBM and the separate pass (k_ln_bwd_planes<32, 57>: 131 us, 0.5 GB read + 0.26 GB written) is gone.
  static constexpr bool LNB2_OK = OTL && L == FB_D1 && NPL <= 2;
  static constexpr ClDesc PD0 = CLD[CL_GD0];
  static constexpr int P0_CPL = PD0.CP + 8, P0_IMG = PD0.HP * P0_CPL;   // LDS pitch / elements of one plane image
  static constexpr int OFR = CX * HX, LDS_IMG = NPL * (GPL + XPL + WPL) * 2, LDS = LDS_IMG + (OTL ? OFR * 4 : 0);
  static constexpr int LDS_LNB2 = LDS + OFR * 4 + 2 * NPL * P0_IMG * 2;   // + the normalised values + two parities of plane images
  static_assert(!OTL || (LDS_IMG % 16 == 0 && OFR % 4 == 0), "result tile: aligned, whole pieces");
  static_assert(V.x == (ENC ? WS.a : WS.b) && GD.CP == CG && XD.CP == CX && M == CPN && CUG == 8 && 4 % CGR == 0, "layer not served");
  static_assert(V.OC == CX && V.OH == HX && V.PH == (ENC ? 1 : 0) && (ENC || (V.R == R && MT == 1)), "layer not served");
  static_assert(KS * 16 <= V.Kp && (V.R - 1) * RSTEP + (KS * 16 / CG + 1) * CPLG <= ROWSG * CPLG + 64, "view runs past the frame image");
};

struct FbArgs {
  const float* dy;       // [F][CG][HG] gradient at the layer's activated output
  const float* a;        // [F][CG][HG] the layer's pre-LN output
  const float* st;       // its LayerNorm statistics (mean, rstd) per frame
  const float* gamma;    // [CG]
  const float* beta;
  const float* xa;       // [F][CX][HX] pre-LN output of the layer below
  const float* xst;
  const float* xgamma;   // [CX]
  const float* xbeta;
  const unsigned short* W;   // weight planes of the input-gradient site [NPL][Mp][Kp] (cv_job)
  float* dx;             // [F][CX][HX] out: gradient at the activated output of the layer below
  float* dW;             // [TAPS * CVW][M], atomicAdd (the TF kernel tensor of the layer)
  float* dgamma;         // [CG], atomicAdd
  float* dbeta;
  float* dbias;
  int F;
  bool bf16_act = false;   // bf16 activation storage of the decoder tensors (launch_fbwd picks the layer's pattern)
  int dy_pitch = 0;        // decoder layer 2: floats per row of dy when its producer padded the rows to 16 bytes (516; 0 = the tensor's own rows)
  // LNB2 (decoder layer 1): non-null = LayerNorm backward of the layer below in the epilogue; dx is then NOT written
  unsigned short* pl0 = nullptr;   // channel-last planes of d(pre-LN output of the layer below) [NPL][pl0_plane] (cl_layout.h: CL_GD0)
  int64_t pl0_plane = 0;
  float* part0 = nullptr;          // [gridDim.x][3][CX]: sum dn xhat | sum dn | sum du per channel of the layer below
};

// BFM: bf16 activation storage (precision "bf16"): bit 0 = dy and a, bit 1 = the input activation xa, bit 2 = the result dx
// DYP: floats per row of dy when they differ from the tensor's (the 1025-tap layer's input gradient writes rows of 516 floats so that
// its 16-byte stores are aligned, gfx950_toep_bf16.h); 0 = rows of the tensor
#ifndef VAENPVC_FB_OCC3
#define VAENPVC_FB_OCC3 0   // bit l: layer FB_* l compiled for three workgroups per CU (168 registers).  Measured (round 5): decoder layer 2 spills 22 registers
                            // there and runs 453 -> 488 us; the other two layers do not fit three workgroups in LDS
#endif
#ifndef VAENPVC_FB_MAXPL
#define VAENPVC_FB_MAXPL 3   // operand planes the fused layer-backward kernels serve (round 6: the fp32-exact 3-term mode as well; 2 = rounds 4-5)
#endif
constexpr int fb_occ(int l, bool lnb2) { return (!lnb2 && ((VAENPVC_FB_OCC3 >> l) & 1)) ? 3 : 2; }
template <int NPL, int L, int BFM = 0, int DYP = 0, bool LNB2 = false>
__global__ void __launch_bounds__(256, fb_occ(L, LNB2)) k_fbwd(FbArgs a) {
  using T = FbCfg<NPL, L>;
  static_assert(!LNB2 || (T::LNB2_OK && BFM == 0), "LayerNorm backward of the layer below: decoder layer 1, fp32 storage");
  constexpr bool BFG = BFM & 1, BFX = (BFM >> 1) & 1, BFO = (BFM >> 2) & 1;
  constexpr int PG = act_pitch(BFG, T::HG), PO = act_pitch(BFO, T::V.OH), PD = DYP ? DYP : PG;
  static_assert(DYP == 0 || (BFM == 0 && DYP >= T::HG), "padded rows: fp32 storage");
  static_assert(BFM == 0 || (NPL == 1 && !T::ENC), "bf16 storage: decoder layers of the bf16 mode");
  constexpr CvSite V = T::V;
  constexpr int CUG = T::CUG, HG = T::HG, CGR = T::CGR, NITG = T::NITG, IPWG = T::IPWG;
  extern __shared__ __attribute__((aligned(16))) unsigned short bsm[];
  __shared__ float part[2][FbCfg<NPL, L>::NITG];
  __shared__ float red[4][3 * FbCfg<NPL, L>::CUG];
  __shared__ float lnx[2][FbCfg<NPL, L>::CX];   // LayerNorm parameters of the input activation (copied once)
  __shared__ float red0[2][4][2];                 // LNB2: the frame's two sums, per parity and wave
  __shared__ float lnq0[2][LNB2 ? FbCfg<NPL, L>::CX : 1];
  float* xh0 = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(bsm) + T::LDS);              // LNB2: [CX][HX] normalised values of the layer below
  unsigned short* img0 = reinterpret_cast<unsigned short*>(xh0 + T::OFR);                               // LNB2: [2][NPL][P0_IMG]
  unsigned short* gs = bsm;                       // [NPL][GPL]  du, channel-last, zero halo rows
  unsigned short* xs = bsm + NPL * T::GPL;        // [NPL][XPL]  activated input
  unsigned short* ws = xs + NPL * T::XPL;         // [NPL][MT * 32][WP]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, lh = lane >> 5;
  const int cg = wave % CGR;                      // this wave's channel group in every item it owns
  const int xwave = (wave + T::XROT) & 3;         // its index for the input-activation items
  // the LayerNorm parameters of this wave's 8 channels, once, as wave-uniform values (scalar registers): read through the
  // pointers inside the frame loop the compiler re-fetched them with vector loads in every pass -- one exposed memory round
  // trip per frame in the second pass
  float gam[CUG], bet[CUG];
#pragma unroll
  for (int c = 0; c < CUG; ++c) {
    gam[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.gamma[cg * CUG + c])));
    bet[c] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.beta[cg * CUG + c])));
  }
  float vd[IPWG][CUG], va[IPWG][CUG], mean = 0.f, rstd = 1.f;
  FwStage<NPL, T::CX, T::CX, T::CPLX, T::HX, 1, 0, T::ROW0X, T::XPL, BFX> sx;
  float su[CUG], sw[CUG], sd[CUG];
#pragma unroll
  for (int c = 0; c < CUG; ++c) su[c] = sw[c] = sd[c] = 0.f;

  // WHICH: bit 0 = the gradient dy, bit 1 = the pre-LN tensor a (+ the frame's statistics)
  auto uload = [&](int f, auto which_) __attribute__((always_inline)) {
    constexpr int WHICH = decltype(which_)::value;
    if constexpr ((WHICH & 2) != 0) {
      mean = a.st[2 * f];
      rstd = a.st[2 * f + 1];
    }
#pragma unroll
    for (int u = 0; u < IPWG; ++u) {
      const int it = wave + 4 * u, h = 64 * (it / CGR) + lane;
      const bool ok = it < NITG && h < HG;
      // (addresses clamped into the tensor instead of predicated loads: no branch per load, the loads issue back to back)
      const int64_t fo = (int64_t)f * (T::CG * PG) + cg * CUG * PG + (ok ? h : 0);
      const int64_t fd = (int64_t)f * (T::CG * PD) + cg * CUG * PD + (ok ? h : 0);
#pragma unroll
      for (int c = 0; c < CUG; ++c) {
        if constexpr ((WHICH & 1) != 0) vd[u][c] = (VAENPVC_FB_ABL & 8) ? 0.5f : act_ld<BFG>(a.dy, fd + c * PD);
        if constexpr ((WHICH & 2) != 0) va[u][c] = (VAENPVC_FB_ABL & 8) ? 0.25f : act_ld<BFG>(a.a, fo + c * PG);
      }
      // (no fill of the idle lanes HERE: a select on a register a load is still writing waits for that load -- and, through `mean`, for the
      //  statistics -- in the middle of the request sequence: `s_waitcnt vmcnt(0)` between the loads of every wave that holds a tail item
      //  (bin 512: lane 0 alone), once per group.  upass1 masks the idle lanes when it consumes the values.)
    }
  };
  // LayerNorm + lrelu backward, first half: dn = dy lrelu'(n), xhat, and the frame's two sums (partials per item)
  auto upass1 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < IPWG; ++u) {
      const int it = wave + 4 * u;
      const bool ok = it < NITG && 64 * (it / CGR) + lane < HG;   // (idle lanes hold the values of a clamped address: xhat = 0, dn = 0 for them)
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < CUG; ++c) {
        const float xh = ok ? (va[u][c] - mean) * rstd : 0.f;
        const float nn = xh * gam[c] + bet[c];
        const float dn = ok ? vd[u][c] * (nn >= 0.f ? 1.0f : LEAK) : 0.f;
        const float dxh = dn * gam[c];
        s1 += dxh;
        s2 += dxh * xh;
        vd[u][c] = dn;
        va[u][c] = xh;
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      if (lane == 0 && it < NITG) {
        part[0][it] = s1;
        part[1][it] = s2;
      }
    }
  };
  // second half: du, the per-channel sums, and du as bf16 terms into the G image
  auto upass2 = [&]() __attribute__((always_inline)) {
    constexpr float INVN = 1.0f / T::NG;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NITG; ++k) {
      s1 += part[0][k];
      s2 += part[1][k];
    }
    s1 *= INVN;
    s2 *= INVN;
#pragma unroll
    for (int u = 0; u < IPWG; ++u) {
      const int it = wave + 4 * u, h = 64 * (it / CGR) + lane;
      const bool live = it < NITG && h < HG;
      float v8[8];
#pragma unroll
      for (int c = 0; c < CUG; ++c) {
        const float d = rstd * (vd[u][c] * gam[c] - s1 - va[u][c] * s2);
        su[c] += live ? vd[u][c] * va[u][c] : 0.f;
        sw[c] += live ? vd[u][c] : 0.f;
        sd[c] += live ? d : 0.f;
        v8[c] = d;
      }
      if (!live) continue;
      u32x4 pk[NPL];
      pack8<NPL>(v8, pk);
      unsigned short* dxp = gs + (T::ROW0G + h) * T::CPLG + cg * CUG;
#pragma unroll
      for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dxp + p * T::GPL) = pk[p];
    }
  };

  int f = blockIdx.x;
  using W3 = std::integral_constant<int, 3>;
  using WPF = std::integral_constant<int, T::PFW>;          // what is prefetched during the GEMMs ...
  using WTOP = std::integral_constant<int, 3 & ~T::PFW>;    // ... and what is loaded at the top of the iteration
  if (T::PFW && f < a.F) {
    uload(f, WPF{});
    if (T::PFW == 3) sx.load(a.xa, a.xst, f, a.F, xwave, lane);
  }
  if (tid < T::CX) {
    lnx[0][tid] = a.xgamma[tid];
    lnx[1][tid] = a.xbeta[tid];
  }
  {  // once per workgroup: zero both images (halo rows, rows past the tensor, tails stay zero), copy the input-gradient weights
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < NPL * (T::GPL + T::XPL) / 8; i += 256) reinterpret_cast<u32x4*>(bsm)[i] = z;
    constexpr int WROW8 = V.Kp / 8, WPIECES = NPL * T::MT * 32 * WROW8;
    for (int i = tid; i < WPIECES; i += 256) {
      const int p = i / (T::MT * 32 * WROW8), r = i - p * (T::MT * 32 * WROW8), m = r / WROW8, c8 = r - m * WROW8;
      *reinterpret_cast<u32x4*>(ws + p * T::WPL + m * T::WP + c8 * 8) =
          *reinterpret_cast<const u32x4*>(a.W + ((size_t)p * V.Mp + m) * V.Kp + c8 * 8);
    }
  }
  // LNB2: a thread owns ONE channel of the layer below (8 threads per channel, positions t8, t8 + 8, ...): its LayerNorm parameters and its
  // three per-channel sums are scalars per thread
  const int c0 = tid >> 3, t8 = tid & 7;
  float g0 = 0.f, b0 = 0.f, su0 = 0.f, sw0 = 0.f, sd0 = 0.f;
  if constexpr (LNB2) {
    g0 = a.xgamma[c0];
    b0 = a.xbeta[c0];
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < 2 * NPL * T::P0_IMG / 8; i += 256) reinterpret_cast<u32x4*>(img0)[i] = z;   // (halo rows stay zero)
    if (blockIdx.x == 0) {   // zero tails behind the planes (as k_cl_produce)
      const int64_t used = (int64_t)a.F * T::PD0.HP * T::PD0.CP;
      for (int64_t i = used + tid; i < a.pl0_plane; i += 256)
#pragma unroll
        for (int p = 0; p < NPL; ++p) a.pl0[p * a.pl0_plane + i] = 0;
    }
  }
  // the plane image of frame fp (parity par) -> its [HP][CP] frame of every plane, consecutive threads = consecutive 16-byte pieces
  auto copy_out0 = [&](int fp, int par) __attribute__((always_inline)) {
    constexpr int G8 = T::PD0.CP / 8, PPP = T::PD0.HP * G8;
    const unsigned short* im = img0 + par * NPL * T::P0_IMG;
    unsigned short* df = a.pl0 + (int64_t)fp * T::PD0.HP * T::PD0.CP;
#pragma unroll
    for (int r = 0; r < cdiv(NPL * PPP, 256); ++r) {
      const int it = min(tid + 256 * r, NPL * PPP - 1);      // (rounds past the end repeat the last piece: unconditional stores)
      const int p = it / PPP, q = it - p * PPP, hp = q / G8, g8 = q - hp * G8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(im + p * T::P0_IMG + hp * T::P0_CPL + 8 * g8);
      *reinterpret_cast<u32x4*>(df + p * a.pl0_plane + (int64_t)q * 8) = v;
    }
  };
  int nfr = 0;      // frames this workgroup has processed (parity of the plane images)
  int fprev = -1;
  __syncthreads();
  // weight-gradient tiles of this wave (gfx950_fwgrad.h): n tiles wn, wn + WN, ...; k-chunks of parity kpar
  const int wn = wave % T::WN, kpar = wave / T::WN;
  f32x16 wacc[T::NTW][T::MTW];
#pragma unroll
  for (int i = 0; i < T::NTW; ++i)
#pragma unroll
    for (int j = 0; j < T::MTW; ++j) wacc[i][j] = zero16();
  const int trow = ((lane & 15) >> 2) + 8 * lh, tcol = 4 * (lane & 3) + 16 * ((lane >> 4) & 1);
  int vcol[T::NTW], pcol[T::MTW];
#pragma unroll
  for (int i = 0; i < T::NTW; ++i) {
    int n = 32 * (wn + i * T::WN) + tcol;
    n = n < T::N ? n : 0;
    vcol[i] = (n / T::CVW) * T::CPLV + n % T::CVW;
  }
#pragma unroll
  for (int j = 0; j < T::MTW; ++j) {
    const int m = 32 * j + tcol;
    pcol[j] = m < T::CPN ? m : 0;
  }
  const unsigned short* vimg = T::ENC ? xs : gs;      // view operand (rows 3 j + tap), plain operand (row j)
  const unsigned short* pimg = T::ENC ? gs + T::ROW0G * T::CPLG : xs;
  constexpr int VPL = T::ENC ? T::XPL : T::GPL, PPL = T::ENC ? T::GPL : T::XPL;
  const int woff = l31 * T::WP + lh * 8;
  for (; f < a.F; f += gridDim.x) {
    if (T::PFW != 3) {   // (unconditional: these staging registers must be dead across the GEMMs, not loop-carried)
      if constexpr (T::PFW != 3) uload(f, WTOP{});
      sx.load(a.xa, a.xst, f, a.F, xwave, lane);
    }
    upass1();
    sx.store(xs, true, lnx[0], lnx[1], xwave, lane, LNB2 ? xh0 : nullptr);
    __syncthreads();   // the partial sums of every item are visible
    upass2();
    __syncthreads();   // both images are complete
    if (T::PFW && f + (int)gridDim.x < a.F) {
      uload(f + gridDim.x, WPF{});
      if (T::PFW == 3) sx.load(a.xa, a.xst, f + gridDim.x, a.F, xwave, lane);
    }
    // ---- input gradient: GEMM rows q (32 per step), steps dealt round-robin to the waves
    constexpr int NSTEPS = cdiv(V.R, 32);
    // (decoder layer 2, whose weight-gradient k-chunks are split over wave pairs: steps dealt from the LAST wave down, so the waves with
    //  one chunk more get one step less, 463 -> 452 us; the layers with two steps keep them on the first waves: the other order cost
    //  them 40 - 60 us, same-box A/B)
    for (int s = (T::KSPLIT > 1 ? 3 - wave : wave); s < ((VAENPVC_FB_ABL & 1) ? 0 : NSTEPS); s += 4) {
      int q = s * 32 + l31;
      const bool nok = q < V.R;
      q = nok ? q : 0;
      const int xoff = q * T::RSTEP;
      f32x16 acc[T::MT];
#pragma unroll
      for (int i = 0; i < T::MT; ++i) acc[i] = zero16();
#pragma unroll
      for (int ks = 0; ks < T::KS; ++ks) {
        u32x4 fa[T::MT][NPL], fb[NPL];
        const int ko = fc_koff<T::CG, T::CPLG>(ks, lh);
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
          for (int i = 0; i < T::MT; ++i) fa[i][p] = *reinterpret_cast<const u32x4*>(ws + p * T::WPL + i * 32 * T::WP + woff + ks * 16);
          fb[p] = *reinterpret_cast<const u32x4*>(gs + p * T::GPL + xoff + ko);
        }
        using PR = Prod<NPL>;
#pragma unroll
        for (int t = 0; t < PR::N; ++t)
#pragma unroll
          for (int i = 0; i < T::MT; ++i) acc[i] = mfma_bf16(fa[i][PR::A[t]], fb[PR::B[t]], acc[i]);
      }
      if (!nok || ((VAENPVC_FB_ABL & 4) && a.F > 0)) continue;
      float* ob = a.dx + (int64_t)f * (V.OC * V.OH);
      if constexpr (T::OTL && BFM == 0) {
        float* ot = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(bsm) + T::LDS_IMG);
        if constexpr (!T::ENC) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            const int m = acc_row(reg, lane);
            if (m < V.M) ot[m * V.OH + q] = acc[0][reg];
          }
        } else {
          const int pbase = q * V.oq + V.o0;
#pragma unroll
          for (int cs = 0; cs < V.mdiv / 2; ++cs) {
            const int chb = (cs & 3) + 8 * (cs >> 2), ch = chb + 4 * lh;
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3) {
              const int mb = p3 * V.mdiv + chb, ti = mb / 32, row = mb % 32, reg = (row & 3) + 4 * (row >> 3);
              if (pbase + p3 >= 0 && pbase + p3 < V.OH) ot[ch * V.OH + pbase + p3] = acc[ti < T::MT ? ti : 0][reg];
            }
          }
        }
        continue;
      }
      if constexpr (!T::ENC) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int m = acc_row(reg, lane);
          if (m < V.M) act_st<BFO>(a.dx, (int64_t)f * (V.OC * PO) + m * PO + q, acc[0][reg]);
        }
      } else {
        // phase-stacked rows m = phase * mdiv + channel: the three phases of (channel, row q) sit in three registers of the
        // same lane and are three consecutive positions 3 q + o0 .. + 2 (gfx950_fconv.h): one 12-byte store
        static_assert(!T::ENC || (V.S == 3 && V.mdiv % 8 == 0 && V.O == V.mdiv), "phase-stacked epilogue");
        struct __attribute__((packed, aligned(4))) f3 { float x, y, z; };
        const int pbase = q * V.oq + V.o0;
        const bool inner = pbase >= 0 && pbase + 2 < V.OH;
#pragma unroll
        for (int cs = 0; cs < V.mdiv / 2; ++cs) {
          const int chb = (cs & 3) + 8 * (cs >> 2), ch = chb + 4 * lh;
          float ph[3];
#pragma unroll
          for (int p3 = 0; p3 < 3; ++p3) {
            const int mb = p3 * V.mdiv + chb, ti = mb / 32, row = mb % 32, reg = (row & 3) + 4 * (row >> 3);
            ph[p3] = acc[ti < T::MT ? ti : 0][reg];
          }
          float* o = ob + ch * V.OH + pbase;
          if (inner) {
            *reinterpret_cast<f3*>(o) = f3{ph[0], ph[1], ph[2]};
          } else {
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3)
              if (pbase + p3 >= 0 && pbase + p3 < V.OH) o[p3] = ph[p3];
          }
        }
      }
    }
    // ---- weight gradient: k-chunks (16 rows j) of parity kpar
    constexpr int CPF = T::R16 / 16;
    for (int kc = kpar; kc < ((VAENPVC_FB_ABL & 2) ? 0 : CPF); kc += T::KSPLIT) {
      const int j0 = kc * 16 + trow;
      const unsigned short* pp0 = pimg + j0 * T::CPLN;
      const unsigned short* pv0 = vimg + (T::S * j0) * T::CPLV;
      u32x4 fp[T::MTW][NPL], fv[T::NTW][NPL];
#pragma unroll
      for (int j = 0; j < T::MTW; ++j)
#pragma unroll
        for (int p = 0; p < NPL; ++p) fp[j][p] = tr_read8_2(pp0 + p * PPL + pcol[j], pp0 + p * PPL + pcol[j] + 4 * T::CPLN);
#pragma unroll
      for (int i = 0; i < T::NTW; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          fv[i][p] = tr_read8_2(pv0 + p * VPL + vcol[i], pv0 + p * VPL + vcol[i] + 4 * T::S * T::CPLV);
      using PR = Prod<NPL>;
      // (the term pairs of Prod are symmetric in the two operands: view x plain)
#pragma unroll
      for (int t = 0; t < PR::N; ++t)
#pragma unroll
        for (int i = 0; i < T::NTW; ++i)
#pragma unroll
          for (int j = 0; j < T::MTW; ++j) wacc[i][j] = mfma_bf16(fv[i][PR::B[t]], fp[j][PR::A[t]], wacc[i][j]);
    }
    __syncthreads();   // all fragment reads of this frame are done before the next one overwrites the images
    if constexpr (LNB2) {
      // ---- LayerNorm + lrelu backward of the layer below on the tile (arithmetic of k_ln_bwd_planes, gfx950_lnb_planes.h)
      const float* ot = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(bsm) + T::LDS_IMG);
      const float rstd0 = a.xst[2 * f + 1];
      constexpr int HX = T::HX, EPT0 = cdiv(HX, 8);
      float dn[EPT0], xh[EPT0];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < EPT0; ++j) {
        const int h = t8 + 8 * j;
        const bool ok = h < HX;
        const int i = c0 * HX + (ok ? h : 0);
        xh[j] = xh0[i];
        const float nn = xh[j] * g0 + b0;
        dn[j] = ok ? ot[i] * (nn >= 0.f ? 1.0f : LEAK) : 0.f;
        const float dxv = dn[j] * g0;
        s1 += dxv;
        s2 += dxv * xh[j];
      }
      s1 = wave_sum(s1);
      s2 = wave_sum(s2);
      const int par = nfr & 1;
      if (lane == 0) {
        red0[par][wave][0] = s1;
        red0[par][wave][1] = s2;
      }
      __syncthreads();
      constexpr float INVN0 = 1.0f / T::OFR;
      s1 = ((red0[par][0][0] + red0[par][1][0]) + (red0[par][2][0] + red0[par][3][0])) * INVN0;
      s2 = ((red0[par][0][1] + red0[par][1][1]) + (red0[par][2][1] + red0[par][3][1])) * INVN0;
      if (fprev >= 0) copy_out0(fprev, par ^ 1);     // the previous frame's image is complete: every thread has passed this frame's barrier
      unsigned short* im = img0 + par * NPL * T::P0_IMG;
#pragma unroll
      for (int j = 0; j < EPT0; ++j) {
        const int h = t8 + 8 * j;
        if (h >= HX) continue;
        const float d = rstd0 * (dn[j] * g0 - s1 - xh[j] * s2);
        su0 += dn[j] * xh[j];
        sw0 += dn[j];
        sd0 += d;
        unsigned t[NPL];
        split_n<NPL>(d, t);
#pragma unroll
        for (int p = 0; p < NPL; ++p) im[p * T::P0_IMG + (T::PD0.HLO + h) * T::P0_CPL + c0] = (unsigned short)t[p];
      }
      fprev = f;
      ++nfr;
    } else if constexpr (T::OTL && BFM == 0) {
      // the frame's input gradient: one aligned contiguous run (unconditional stores, the rounds past the end repeat the last piece: a static
      // store count keeps the wait for the next frame's prefetched registers a counted one).  The tile is rewritten two barriers from here.
      if (!((VAENPVC_FB_ABL & (1 | 4)) && a.F > 0)) {
        const f32x4* ot4 = reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(bsm) + T::LDS_IMG);
        f32x4* og = reinterpret_cast<f32x4*>(a.dx + (int64_t)f * T::OFR);
        constexpr int N4 = T::OFR / 4;
#pragma unroll
        for (int r = 0; r < cdiv(N4, 256); ++r) {
          const int i = min(tid + 256 * r, N4 - 1);
          og[i] = ot4[i];
        }
      }
    }
  }
  if constexpr (LNB2) {
    __syncthreads();
    if (fprev >= 0) copy_out0(fprev, (nfr - 1) & 1);
    // the three per-channel sums of the layer below: the 8 threads of a channel are 8 consecutive lanes
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      su0 += __shfl_xor(su0, o);
      sw0 += __shfl_xor(sw0, o);
      sd0 += __shfl_xor(sd0, o);
    }
    if (t8 == 0) {
      float* pp = a.part0 + (int64_t)blockIdx.x * (3 * T::CX);
      pp[c0] = su0;
      pp[T::CX + c0] = sw0;
      pp[2 * T::CX + c0] = sd0;
    }
  }
  if ((VAENPVC_FB_ABL & 16) && a.F > 0) return;
  // ---- flush: the weight-gradient tile (rows n = (tap, channel of the view operand), lanes m: consecutive addresses of dW[n][m]) ...
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
  for (int c = 0; c < CUG; ++c) {
    const float u = wave_sum(su[c]), w = wave_sum(sw[c]), d = wave_sum(sd[c]);
    if (lane == 0) {
      red[wave][c] = u;
      red[wave][CUG + c] = w;
      red[wave][2 * CUG + c] = d;
    }
  }
  __syncthreads();
  if (tid < 3 * T::CG) {
    const int which = tid / T::CG, c = tid - which * T::CG, cgi = c / CUG, cc = c - cgi * CUG;
    float v = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4)
      if (w4 % CGR == cgi) v += red[w4][which * CUG + cc];
    float* dst = which == 0 ? a.dgamma + c : which == 1 ? a.dbeta + c : a.dbias + c;
    atomicAdd(dst, v);
  }
}

// the grid launch_fbwd uses: also the number of rows of per-workgroup parts the LNB2 form leaves in `part0`, so the
// reduce kernel behind it must be given THIS number (round-5 advisor: the caller had its own copy of the formula)
template <int NPL, int L>
static unsigned fbwd_grid(int64_t F) {
  using T = FbCfg<NPL, L>;
  return (unsigned)cmin_((int)F, T::LDS > 78 * 1024 ? 256 : (fb_occ(L, false) == 3 && 3 * T::LDS <= 156 * 1024) ? 768 : 512);
}
static unsigned fbwd_grid_d1(int npl, int64_t F) { return npl == 1 ? fbwd_grid<1, FB_D1>(F) : fbwd_grid<2, FB_D1>(F); }

template <int NPL, int L>
static void launch_fbwd(const FbArgs& a, hipStream_t s) {
  using T = FbCfg<NPL, L>;
  const unsigned grid = fbwd_grid<NPL, L>(a.F);
  if constexpr (NPL == 1 && (L == FB_D2 || L == FB_D1)) {
    if (a.bf16_act) {   // bf16 activation storage: layer 2 reads and writes bf16 throughout, layer 1 reads (dy, a) as bf16
      constexpr int BFM = L == FB_D2 ? 7 : 1;
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fbwd<NPL, L, BFM>), T::LDS);
      hipLaunchKernelGGL((k_fbwd<NPL, L, BFM>), dim3(grid), dim3(256), T::LDS, s, a);
      return;
    }
  }
  if constexpr (T::LNB2_OK) {
    if (a.pl0) {
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fbwd<NPL, L, 0, 0, true>), T::LDS_LNB2);
      hipLaunchKernelGGL((k_fbwd<NPL, L, 0, 0, true>), dim3(grid), dim3(256), T::LDS_LNB2, s, a);
      return;
    }
  }
  if constexpr (L == FB_D2) {
    if (a.dy_pitch == FB_DY2_PITCH) {
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fbwd<NPL, L, 0, FB_DY2_PITCH>), T::LDS);
      hipLaunchKernelGGL((k_fbwd<NPL, L, 0, FB_DY2_PITCH>), dim3(grid), dim3(256), T::LDS, s, a);
      return;
    }
  }
  rt().ensure_lds(reinterpret_cast<const void*>(&k_fbwd<NPL, L>), T::LDS);
  hipLaunchKernelGGL((k_fbwd<NPL, L>), dim3(grid), dim3(256), T::LDS, s, a);
}
template <int NPL>
static bool fbwd(int layer, const FbArgs& a, hipStream_t s) {
  if constexpr (NPL <= VAENPVC_FB_MAXPL) {
    switch (layer) {
      case FB_D2: launch_fbwd<NPL, FB_D2>(a, s); return true;
      case FB_D1: launch_fbwd<NPL, FB_D1>(a, s); return true;
      case FB_E1: launch_fbwd<NPL, FB_E1>(a, s); return true;
    }
  }
  return false;
}

}  // namespace tuned
}  // namespace vaenpvc
