// gfx950_fconv.h -- the THIN conv / conv_transpose sites (<= 64 GEMM rows, K <= 128) as fused view GEMMs on the bf16
// matrix cores: the geometry of gfx950_viewconv.h (overlapping-row view of channel-last planes), but the planes never
// exist in HBM.  A workgroup owns TF whole frames:
//   1. the fp32 frames ([C][H], optionally LayerNorm + lrelu with the stored statistics) are read ONCE, coalesced along
//      H -- a lane owns one position and walks the channels in registers -- split into NPL bf16 terms and written as
//      channel-last rows with zero halo into LDS (16-byte stores, no transposition scratch);
//   2. the site's weight planes ([M][K], a few KB) are copied into LDS once;
//   3. every wave then walks its share of the TF * R GEMM rows with NO barrier: A fragments from the resident weights,
//      B fragments straight from the resident frames at (row, tap, channel) offsets -- the overlap between neighbouring
//      rows (7/3 for a stride-3 conv, 3x for a transposed-conv phase stack) costs LDS reads, not HBM or L1 traffic;
//   4. results leave the accumulators as canonical fp32 [F][C][H] (+ bias), as from the exact-fp32 engines.
// HBM traffic = input + output, once.  These layers are HBM-bound: the exact-fp32 engines they
// replace reach 2.1 - 3.8 TB/s because 32x32x2 MFMA tiles are mostly padding at 8 - 32 channels.
// Reference: util/layers.py:56-64 (conv2d SAME), model/vae.py:96-99 (conv2d_transpose SAME) and their autodiff.
#pragma once
#include "gfx950_viewconv.h"

namespace vaenpvc {
namespace tuned {

// frames per workgroup and 32-row tiles per wave step, per site (0 = site not served by this kernel)
#ifndef VAENPVC_FC_W2
#define VAENPVC_FC_W2 0     // bit set of CV_* sites on TWO-wave workgroups with 2-frame groups, four per CU: the same registers and LDS per CU as
                            // two four-wave workgroups with 4-frame groups, but four independent load -> convert -> GEMM -> store pipelines
#endif
constexpr bool fc_w2(int site) { return (VAENPVC_FC_W2 >> site) & 1; }
#ifndef VAENPVC_FC_OCC3
#define VAENPVC_FC_OCC3 0x20   // bit set of CV_* sites on 2-frame groups with three workgroups per CU (168 registers) instead of 4-frame groups with two:
                               // decoder layer 2 forward (the largest frames) 268 -> 244 us; the other thin sites lose 15 - 50 % (same-box A/B, round 4)
#endif
constexpr bool fc_occ3(int site) { return (VAENPVC_FC_OCC3 >> site) & 1; }
constexpr int fc_tf(int site) {
  return fc_occ3(site) || fc_w2(site) ? 2 : site == CV_E1F || site == CV_D1F || site == CV_D2F || site == CV_E1G || site == CV_D1G ? 4
         : site == CV_D2G ? 2
         : site == CV_E2G ? 6   // a medium site: weights + frames fill most of the LDS (one workgroup per CU with 2 planes)
                          : 0;
}
// (encoder layer 2 FORWARD was tried the same way and dropped: 417 us against 255 us on the fp32 engine -- K = 224
//  against 64 rows: two waves per M tile re-read every B fragment, one workgroup per CU)
constexpr int fc_nj(int site) { return site == CV_D2G || site == CV_E2G || (site == CV_D2F && fc_occ3(site)) ? 1 : 2; }   // (decoder layer 2 forward at three workgroups per CU: 32-row steps, three 16-register accumulators per wave instead of two 32-register ones)

#ifndef VAENPVC_FC_DEFER
#define VAENPVC_FC_DEFER 1   // result stores of a group issued at the top of the next iteration (0: at the end of the GEMM phase; A/B)
#endif
#ifndef VAENPVC_FC_ABL
#define VAENPVC_FC_ABL 0   // developer ablation (wrong results): 1 no global loads, 2 no conversion / LDS stores, 4 no fragment reads / MFMAs, 8 no result stores
#endif
template <int NPL, int SITE>
struct FcCfg {
  static constexpr CvSite V = CVS[SITE];
  static constexpr ClDesc X = CLD[V.x];
  static constexpr int C = X.C, H = X.H, CP = X.CP, HLO = X.HLO, HP = X.HP;
  // LDS row pitch: +8 elements where the row stride would otherwise be a multiple of 64 bytes (bank conflicts)
  static constexpr int CPL = (CP == 32 || CP == 64 || CP == 128) ? CP + 8 : CP;
  static constexpr int FS = HP * CPL;                  // elements per frame
  static constexpr int TF = fc_tf(SITE), NJ = fc_nj(SITE), SROWS = 32 * NJ;
  static constexpr int NWV = fc_w2(SITE) ? 2 : 4, NTHR = 64 * NWV;      // waves / threads per workgroup
  static constexpr int XPL = TF * FS + 64;             // elements per plane (+ zero tail: K runs rounded up to 16)
  static constexpr int K = V.NT * CP, KS = cdiv(K, 16);
  static constexpr int MT = cdiv(V.M, 32), WP = V.Kp + 8, WPL = MT * 32 * WP;   // weight rows padded by 16 bytes
  static constexpr int RSTEP = (V.step / CP) * CPL;    // elements between GEMM rows
  static constexpr int LDS = NPL * (XPL + WPL) * 2;
  static constexpr int OCC = fc_occ3(SITE) && 3 * LDS <= 156 * 1024 ? 3 : 2;   // waves per SIMD the kernel is compiled for
  static constexpr int WGS_PER_CU = fc_w2(SITE) ? (4 * LDS <= 160 * 1024 ? 4 : 3) : OCC;
  static_assert(TF > 0 && C <= 64 && KS * 16 <= V.Kp, "site not served");
};

struct FcArgs {
  const float* src;     // [F][C][H] fp32
  const float* st;      // LN statistics (mean, rstd) per frame, or nullptr
  float* st_out;        // non-null: the statistics are computed HERE from the staged frames (two-pass, in registers) and
                        // stored -- the workgroup owns whole frames, so the separate statistics pass over the tensor goes away
  const float* gamma;
  const float* beta;
  const unsigned short* W;   // weight planes [NPL][Mp][Kp] (cv_job)
  const float* bias;    // [O] or nullptr
  float* out;           // [F][OC][OH]
  int F;
  bool bf_in = false, bf_out = false;   // bf16 activation storage of the input / output tensor (precision "bf16", decoder layers 1 - 2)
  // k_fconv_r, decoder layer 0 forward: non-null = the staged bf16 terms of the input ALSO leave as its channel-last planes
  // ([NPL][cl_plane], frames of [HP][CP], cl_layout.h) -- the operand the layer's weight-gradient GEMM reads in the backward pass
  unsigned short* cl_out = nullptr;
  int64_t cl_plane = 0;
  // k_fconv_r, decoder layer 0 input gradient: non-null = the operand is read as its channel-last planes ([NPL][cl_plane], written by
  // the LayerNorm backward in front, gfx950_lnb_planes.h) instead of as the fp32 tensor `src`: a straight copy into the LDS image
  const unsigned short* cl_in = nullptr;
  // k_fconv_r, encoder layer 2 forward (round 5): non-null = the LayerNorm statistics of the RESULT are taken from the result tile in LDS
  // (one wave per frame) and stored here, and the activated result lrelu(LN(out)) ALSO leaves as the channel-last planes the next layer's
  // view GEMM reads ([NPL][cl2_plane], frames of [HP][CP] with zero halo rows, cl_layout.h: CL_Y2): the separate statistics + planes pass
  // over the tensor (k_cl_produce<LN = 2>, 73 us) goes away
  float* st2_out = nullptr;
  const float* gamma2 = nullptr;
  const float* beta2 = nullptr;
  unsigned short* cl2_out = nullptr;
  int64_t cl2_plane = 0;
  // k_fconv, decoder layer 2 forward with the DECODER TAIL in its epilogue (round 5, TAIL): st2_out / gamma2 / beta2 as above, and
  unsigned short* yp = nullptr;     // the 1025-tap layer's forward operand: planes of lrelu(LN(out)) [F][NPL][8][528] (zero padding 513..527)
  float* decy = nullptr;            // fp32 [F][8][513]: only bin 512 of every channel is written (read by the loss kernel's edge term)
  const float* wc = nullptr;        // tap table of the 1025-tap layer [8][1040]: Wc[c][8 + t] (k_ln_stats_act_planes)
  const float* bias3 = nullptr;     // its bias [1]
  float* xh = nullptr;              // [F][513]: column 512 of the forward result = dot(activated frame, reversed taps) + bias
  int zero_xh = 0;                  // 1: bins 0 .. 511 of xh are zeroed (the 1025-tap forward GEMM accumulates channel groups into them at small batches)
  // k_fconv_r, decoder layer 0 input gradient (round 5, POUT): non-null = the result rows d(h) leave as the bf16 operand planes of the two
  // merge GEMMs ([NPL][pl_plane], rows of pl_kp elements, columns behind the row's 1539 values zero) INSTEAD of as the fp32 tensor `out`:
  // the split pass over d(h) (k_split_segsum: 200 MB read, 210 MB written) shrinks to a read-only pass for the per-speaker sums
  unsigned short* pl_out = nullptr;
  int64_t pl_plane = 0;
  int pl_kp = 0;
};

template <int CP, int CPL>
__device__ __forceinline__ int fc_koff(int ks, int lh) {
  const int k0 = 16 * ks + 8 * lh;
  if constexpr (CPL == CP) return k0;
  else return (k0 / CP) * CPL + (k0 % CP);
}

// LN: 0 plain input, 1 LayerNorm + lrelu with given statistics, 2 ... with statistics computed here
// TAIL (round 5, decoder layer 2 forward only): the workgroup owns whole result frames, so the work of the separate pass between this layer and
// the 1025-tap layer (k_ln_stats_act_planes: 240 us, a re-read of the 0.54 GB result) runs on the accumulators in the deferred epilogue:
// LayerNorm statistics of the result (two-pass: per-wave partials through LDS, two barriers), the activated values as bf16 terms into an LDS
// image of the frames' operand planes [frame][plane][8][528] (copied out as 16-byte pieces behind a third barrier), bin 512 of the activated
// tensor as fp32, and output column 512 of the 1025-tap layer (a dot product of the activated frame with the reversed taps).  Two workgroups per
// CU (the image does not fit beside three), up to 256 registers.
constexpr int fc_tail_img_bytes(int npl, int tf) { return tf * npl * TB_C * TB_KP * 2; }
// OST (round 6, decoder layer 2 forward only): the LayerNorm statistics of the RESULT (mean, rstd per frame -> st2_out) are taken from the
// accumulators in the deferred epilogue and NOTHING else of the tail: per wave and frame the triple (count, sum, centred square sum about the
// wave's own mean) by wave reductions, the four triples of a frame combined behind the barrier the loop has anyway (Chan's formula: two-pass
// quality without a second barrier).  No LDS image, no third workgroup lost (the TAIL variant's cost): with the statistics known the
// 1025-tap layer's forward kernel normalises the fp32 tensor while it stages it (k_toep_gemm_bf16<..., LNA>) and the pass in between
// (k_ln_stats_act_planes: 237 us, a re-read of the 0.54 GB tensor + 0.55 GB of planes written) is gone.
template <int NPL, int SITE, int LN, bool BIN = false, bool BOUT = false, bool TAIL = false, bool OST = false>
__global__ void __launch_bounds__((FcCfg<NPL, SITE>::NTHR), (TAIL ? 2 : FcCfg<NPL, SITE>::OCC)) k_fconv(FcArgs a) {
  using T = FcCfg<NPL, SITE>;
  constexpr CvSite V = T::V;
  static_assert(!OST || (!TAIL && SITE == CV_D2F && !BOUT && V.PH && V.mdiv == 8 && T::NJ == 1 && T::MT == 1 && T::TF == 2 && T::NWV == 4 && VAENPVC_FC_DEFER),
                "result statistics: decoder layer 2 forward on 2-frame groups, fp32 output, deferred epilogue");
  static_assert(!TAIL || (SITE == CV_D2F && !BIN && !BOUT && V.PH && V.O == TB_C && V.OH == TB_H && T::NJ == 1 && T::MT == 1 && VAENPVC_FC_DEFER),
                "decoder tail: decoder layer 2 forward, fp32 storage, deferred epilogue");
  extern __shared__ __attribute__((aligned(16))) unsigned short fsm[];
  __shared__ float tpart[3][4][FcCfg<NPL, SITE>::TF];   // TAIL: per-wave partials of the frames' sum / centred square sum / dot product
  __shared__ float part[2][FcCfg<NPL, SITE>::TF * cdiv(FcCfg<NPL, SITE>::H, 64)];   // per staging item: sum, sum of squared deviations
  // LayerNorm parameters of the input, copied once: read through the argument pointers inside the group loop they were re-fetched
  // with vector loads by every group, right behind the result stores (one exposed round trip per group; round 4)
  __shared__ float lnp[2][FcCfg<NPL, SITE>::CP];
  unsigned short* xs = fsm;                       // [NPL][XPL]
  unsigned short* ws = fsm + NPL * T::XPL;        // [NPL][MT*32][WP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  constexpr int NCH = cdiv(T::H, 64), NWV = T::NWV, NTHR = T::NTHR;
  const int ngroups = cdiv(a.F, T::TF);
  // staging items = (frame of the group, 64-position chunk), dealt round-robin to the four waves; an item's C coalesced
  // 256-byte loads go to registers one step ahead: the loads of group g + 1 fly during the GEMM of group g, the
  // conversion and the 16-byte LDS stores happen at the top of the next iteration
  constexpr int NIT = T::TF * NCH, IPW = cdiv(NIT, NWV);
  float v[IPW][T::CP];
  float mean[IPW], rstd[IPW];
  auto fload = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const int it = wave + NWV * u, fl = it / NCH, k = it - fl * NCH;
      const int f = g * T::TF + fl, h = 64 * k + lane;
      const bool fok = it < NIT && f < a.F;
      constexpr int PIN = act_pitch(BIN, T::H);
      const int64_t sfo = (int64_t)(fok ? f : 0) * (T::C * PIN);
      if constexpr (LN == 1) {
        mean[u] = a.st[2 * (fok ? f : 0)];
        rstd[u] = a.st[2 * (fok ? f : 0) + 1];
      }
#pragma unroll
      for (int c = 0; c < T::CP; ++c) v[u][c] = (!(VAENPVC_FC_ABL & 1) && c < T::C && h < T::H && fok) ? act_ld<BIN>(a.src, sfo + c * PIN + h) : 0.f;
    }
  };
  // LayerNorm statistics of the group's frames from the registers (LN == 2): per item partial sums through LDS, two
  // workgroup barriers; mean first, then the centred second moment (as k_ln_stats_fast)
  auto fstats = [&](int g) __attribute__((always_inline)) {
    constexpr float INVN = 1.0f / (T::C * T::H);
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const int it = wave + NWV * u;
      float sm = 0.f;
#pragma unroll
      for (int c = 0; c < T::C; ++c) sm += v[u][c];       // (invalid lanes / frames hold zeros)
      sm = wave_sum(sm);
      if (lane == 0 && it < NIT) part[0][it] = sm;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const int it = wave + NWV * u, fl = (it < NIT ? it : 0) / NCH, k = it - fl * NCH;
      float sm = 0.f;
#pragma unroll
      for (int kk = 0; kk < NCH; ++kk) sm += part[0][fl * NCH + kk];
      mean[u] = sm * INVN;
      const bool ok = 64 * k + lane < T::H;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < T::C; ++c) {
        const float d = v[u][c] - mean[u];
        q += ok ? d * d : 0.f;
      }
      q = wave_sum(q);
      if (lane == 0 && it < NIT) part[1][it] = q;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const int it = wave + NWV * u, fl = (it < NIT ? it : 0) / NCH, k = it - fl * NCH;
      float q = 0.f;
#pragma unroll
      for (int kk = 0; kk < NCH; ++kk) q += part[1][fl * NCH + kk];
      rstd[u] = 1.0f / sqrtf(q * INVN + LN_EPS);
      const int f = g * T::TF + fl;
      if (lane == 0 && it < NIT && k == 0 && f < a.F) {
        a.st_out[2 * f] = mean[u];
        a.st_out[2 * f + 1] = rstd[u];
      }
    }
  };
  auto fstore = [&](int g) __attribute__((always_inline)) {
    if constexpr ((VAENPVC_FC_ABL & 2) != 0) return;
    if constexpr (LN == 2) fstats(g);
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const int it = wave + NWV * u, fl = it / NCH, k = it - fl * NCH;
      const int h = 64 * k + lane;
      if (!(it < NIT && g * T::TF + fl < a.F && h < T::H)) continue;
      if constexpr (LN != 0) {
#pragma unroll
        for (int c = 0; c < T::C; ++c) v[u][c] = lnact_v(v[u][c], mean[u], rstd[u], lnp[0][c], lnp[1][c]);
      }
      unsigned short* dx = xs + fl * T::FS + (T::HLO + h) * T::CPL;
#pragma unroll
      for (int g8 = 0; g8 < T::CP / 8; ++g8) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = v[u][8 * g8 + j];
        u32x4 pk[NPL];
        pack8<NPL>(v8, pk);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dx + p * T::XPL + 8 * g8) = pk[p];
      }
    }
  };
  int g = blockIdx.x;
  if (g < ngroups) fload(g);
  // ---- once per workgroup: zero the frame tile (halo rows, channel padding, tail stay zero), copy the weights
  if constexpr (LN != 0) {
    if (tid < T::C) {
      lnp[0][tid] = a.gamma[tid];
      lnp[1][tid] = a.beta[tid];
    }
  }
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < NPL * T::XPL / 8; i += NTHR) reinterpret_cast<u32x4*>(xs)[i] = z;
    constexpr int WROW8 = V.Kp / 8, WPIECES = NPL * T::MT * 32 * WROW8, WPT = cdiv(WPIECES, NTHR);
    u32x4 wr[WPT];
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      int i = tid + NTHR * u;
      i = i < WPIECES ? i : WPIECES - 1;
      const int p = i / (T::MT * 32 * WROW8), r = i - p * (T::MT * 32 * WROW8), m = r / WROW8, c8 = r - m * WROW8;
      wr[u] = *reinterpret_cast<const u32x4*>(a.W + ((size_t)p * V.Mp + m) * V.Kp + c8 * 8);
    }
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
      int i = tid + NTHR * u;
      i = i < WPIECES ? i : WPIECES - 1;
      const int p = i / (T::MT * 32 * WROW8), r = i - p * (T::MT * 32 * WROW8), m = r / WROW8, c8 = r - m * WROW8;
      *reinterpret_cast<u32x4*>(ws + p * T::WPL + m * T::WP + c8 * 8) = wr[u];
    }
  }
  __syncthreads();
  const int woff = l31 * T::WP + lh * 8;
  // The result stores of a group are DEFERRED to the top of the next iteration, behind the wait for that group's prefetched
  // loads (VAENPVC_FC_DEFER, round 4): loads and stores share one in-order counter, so stores issued at the end of the GEMM
  // phase had to be acknowledged before the wave could touch the next group's loaded registers -- their whole round trip
  // was exposed once per group (the ablation without result stores ran 44 - 59 % faster).  Issued behind
  // that wait they fly during the conversion, the LDS stores, the barrier and the next GEMM instead.  The accumulators of a
  // wave's steps (NSW of them) stay live across the barrier; nothing else changes.
  constexpr int NSW = cdiv(cdiv(T::TF * V.R, T::SROWS), NWV);
  f32x16 acc[NSW][T::MT][T::NJ];
  // bias of the rows a lane stores, once per kernel (a load inside the epilogue would order every later store behind it)
  constexpr int NBV = V.PH ? V.mdiv / 2 : T::MT * 16;
  float bv[NBV];
#pragma unroll
  for (int k = 0; k < NBV; ++k) {
    int ch;
    bool ok;
    if constexpr (V.PH) {
      ch = (k & 3) + 8 * (k >> 2) + 4 * lh;
      ok = true;
    } else {
      const int m = (k >> 4) * 32 + acc_row(k & 15, lane);
      ch = m % V.mdiv;
      ok = m < V.M && ch < V.O;
    }
    bv[k] = (a.bias && ok) ? a.bias[ch] : 0.f;
  }
  auto epilogue = [&](int f0, int nf) __attribute__((always_inline)) {
    const int nrows = nf * V.R;
#pragma unroll
    for (int sidx = 0; sidx < NSW; ++sidx) {
      const int s = wave + NWV * sidx;
      // accumulator rows = GEMM rows m (phase * mdiv + channel), lanes = 32 consecutive (frame, position) rows
#pragma unroll
      for (int j = 0; j < T::NJ; ++j) {
        const int n = s * T::SROWS + j * 32 + l31;
        if (n >= nrows || ((VAENPVC_FC_ABL & 8) && a.F > 0)) continue;
        const int fl = n / V.R, q = n - fl * V.R;
        float* ob = a.out + (int64_t)(f0 + fl) * (V.OC * V.OH);
        const int pbase = q * V.oq + V.o0;
        if constexpr (V.PH) {
          // transposed conv: the three output phases of (channel, row q) sit in three registers of the SAME lane
          // (mdiv is a multiple of 8) and are three consecutive positions: one 12-byte store per (lane, channel), 32
          // lanes = 384 contiguous bytes, instead of three 4-byte stores at a 12-byte stride
          static_assert(!V.PH || (V.S == 3 && V.mdiv % 8 == 0 && V.O == V.mdiv), "phase-stacked epilogue");
          struct __attribute__((packed, aligned(4))) f3 { float x, y, z; };
          const bool inner = pbase >= 0 && pbase + 2 < V.OH;
#pragma unroll
          for (int cs = 0; cs < V.mdiv / 2; ++cs) {
            const int chb = (cs & 3) + 8 * (cs >> 2);        // + 4 * lh
            const int ch = chb + 4 * lh;
            float ph[3];
#pragma unroll
            for (int p3 = 0; p3 < 3; ++p3) {
              const int mb = p3 * V.mdiv + chb;               // row of the lh = 0 lanes; lh = 1: + 4 (same tile, same register)
              const int ti = mb / 32, row = mb % 32, reg = (row & 3) + 4 * (row >> 3);
              ph[p3] = acc[sidx][ti][j][reg] + bv[cs];
            }
            if constexpr (BOUT) {
              // bf16 storage, rows of an even pitch: the three positions leave as one aligned pair + one single
              constexpr int PO = act_pitch(true, V.OH);
              unsigned short* o2 = reinterpret_cast<unsigned short*>(a.out) + ((int64_t)(f0 + fl) * V.OC + ch) * PO + pbase;
              if (inner) {
                if (pbase & 1) {
                  o2[0] = (unsigned short)bf16_rn(ph[0]);
                  *reinterpret_cast<unsigned*>(o2 + 1) = cvt_pk_bf16(ph[1], ph[2]);
                } else {
                  *reinterpret_cast<unsigned*>(o2) = cvt_pk_bf16(ph[0], ph[1]);
                  o2[2] = (unsigned short)bf16_rn(ph[2]);
                }
              } else {
#pragma unroll
                for (int p3 = 0; p3 < 3; ++p3)
                  if (pbase + p3 >= 0 && pbase + p3 < V.OH) o2[p3] = (unsigned short)bf16_rn(ph[p3]);
              }
              continue;
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
        } else {
#pragma unroll
          for (int i = 0; i < T::MT; ++i)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int m = i * 32 + acc_row(reg, lane);
              if (m >= V.M) continue;
              const int pim = m / V.mdiv, ch = m - pim * V.mdiv;
              const int pos = pbase + pim;
              if (ch < V.O && pos >= 0 && pos < V.OH) ob[ch * V.OH + pos] = acc[sidx][i][j][reg] + bv[i * 16 + reg];
            }
        }
      }
    }
  };
  // ---- TAIL epilogue (see above).  Lane: row n = (frame fl, view row q) of each of its NSW steps; 4 channels (cs + 4 lh) x 3 positions.
  float g2[4], b2[4];
  if constexpr (TAIL) {
#pragma unroll
    for (int cs = 0; cs < 4; ++cs) {
      g2[cs] = a.gamma2[cs + 4 * lh];
      b2[cs] = a.beta2[cs + 4 * lh];
    }
  }
  auto epilogue_tail = [&](int f0, int nf) __attribute__((always_inline)) {
   if constexpr (TAIL) {
    static_assert(!TAIL || T::TF == 2, "two frames per group");
    const int nrows = nf * V.R;
    unsigned short* img = fsm + NPL * (T::XPL + T::WPL);      // [TF][NPL][8][528]
    // taps of this lane's positions (before any store; rows past the end / positions out of range read tap 0 of a valid address)
    float wt[NSW][4][3];
    int fl_[NSW], pb_[NSW];
    bool ok_[NSW];
#pragma unroll
    for (int sidx = 0; sidx < NSW; ++sidx) {
      const int n = (wave + NWV * sidx) * T::SROWS + l31;
      ok_[sidx] = n < nrows;
      const int nn = ok_[sidx] ? n : 0;
      fl_[sidx] = nn / V.R;
      pb_[sidx] = (nn - fl_[sidx] * V.R) * V.oq + V.o0;
#pragma unroll
      for (int cs = 0; cs < 4; ++cs)
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) {
          const int pos = min(max(pb_[sidx] + p3, 0), TB_H - 1);
          wt[sidx][cs][p3] = a.wc[(cs + 4 * lh) * 1040 + 8 + 1024 - pos];
        }
    }
    // values (conv + bias) in place of the accumulators; positions outside the tensor / rows past the group are masked everywhere below
    // accumulator register of (channel slot cs, phase p3): tile row = phase * mdiv + channel
#define FC_TAIL_VAL(sidx, cs, p3) acc[sidx][0][0][(((p3) * V.mdiv + (cs)) % 32 & 3) + 4 * ((((p3) * V.mdiv + (cs)) % 32) >> 3)]
    auto live = [&](int sidx, int p3) __attribute__((always_inline)) {
      const int pos = pb_[sidx] + p3;
      return ok_[sidx] && pos >= 0 && pos < TB_H;
    };
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int sidx = 0; sidx < NSW; ++sidx)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs)
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) {
          FC_TAIL_VAL(sidx, cs, p3) += bv[cs];
          const float t = live(sidx, p3) ? FC_TAIL_VAL(sidx, cs, p3) : 0.f;
          s0 += fl_[sidx] == 0 ? t : 0.f;
          s1 += fl_[sidx] == 1 ? t : 0.f;
        }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    if (lane == 0) {
      tpart[0][wave][0] = s0;
      tpart[0][wave][1] = s1;
    }
    __syncthreads();
    constexpr float INVN = 1.0f / (TB_C * TB_H);
    const float mean0 = ((tpart[0][0][0] + tpart[0][1][0]) + (tpart[0][2][0] + tpart[0][3][0])) * INVN;
    const float mean1 = ((tpart[0][0][1] + tpart[0][1][1]) + (tpart[0][2][1] + tpart[0][3][1])) * INVN;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int sidx = 0; sidx < NSW; ++sidx)
#pragma unroll
      for (int cs = 0; cs < 4; ++cs)
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) {
          const float d = FC_TAIL_VAL(sidx, cs, p3) - (fl_[sidx] == 0 ? mean0 : mean1);
          const float t = live(sidx, p3) ? d * d : 0.f;
          q0 += fl_[sidx] == 0 ? t : 0.f;
          q1 += fl_[sidx] == 1 ? t : 0.f;
        }
    q0 = wave_sum(q0);
    q1 = wave_sum(q1);
    if (lane == 0) {
      tpart[1][wave][0] = q0;
      tpart[1][wave][1] = q1;
    }
    __syncthreads();
    const float rstd0 = 1.0f / sqrtf(((tpart[1][0][0] + tpart[1][1][0]) + (tpart[1][2][0] + tpart[1][3][0])) * INVN + LN_EPS);
    const float rstd1 = 1.0f / sqrtf(((tpart[1][0][1] + tpart[1][1][1]) + (tpart[1][2][1] + tpart[1][3][1])) * INVN + LN_EPS);
    if (tid < nf) {
      a.st2_out[2 * (f0 + tid)] = tid == 0 ? mean0 : mean1;
      a.st2_out[2 * (f0 + tid) + 1] = tid == 0 ? rstd0 : rstd1;
    }
    // pre-LN result (fp32, as the plain epilogue), activated values -> image / bin 512 / dot product
    float d0 = 0.f, d1 = 0.f;
    struct __attribute__((packed, aligned(4))) f3 { float x, y, z; };
#pragma unroll
    for (int sidx = 0; sidx < NSW; ++sidx) {
      if (!ok_[sidx]) continue;
      const int fl = fl_[sidx], pbase = pb_[sidx];
      const float mean = fl == 0 ? mean0 : mean1, rstd = fl == 0 ? rstd0 : rstd1;
      float* ob = a.out + (int64_t)(f0 + fl) * (V.OC * V.OH);
      const bool inner = pbase >= 0 && pbase + 2 < V.OH;
      float dsum = 0.f;
#pragma unroll
      for (int cs = 0; cs < 4; ++cs) {
        const int ch = cs + 4 * lh;
        const float p0 = FC_TAIL_VAL(sidx, cs, 0), p1 = FC_TAIL_VAL(sidx, cs, 1), p2 = FC_TAIL_VAL(sidx, cs, 2);
        float* o = ob + ch * V.OH + pbase;
        if (inner) {
          *reinterpret_cast<f3*>(o) = f3{p0, p1, p2};
        } else {
          if (pbase >= 0 && pbase < V.OH) o[0] = p0;
          if (pbase + 1 >= 0 && pbase + 1 < V.OH) o[1] = p1;
          if (pbase + 2 >= 0 && pbase + 2 < V.OH) o[2] = p2;
        }
        unsigned short* ir = img + (fl * NPL * TB_C + ch) * TB_KP;      // + plane * TB_C * TB_KP
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) {
          const int pos = pbase + p3;
          if (pos < 0) continue;
          unsigned t[NPL];
          if (pos < TB_H) {
            const float y = lnact_v(FC_TAIL_VAL(sidx, cs, p3), mean, rstd, g2[cs], b2[cs]);
            dsum += y * wt[sidx][cs][p3];
            split_n<NPL>(y, t);
            if (pos == TB_H - 1) a.decy[(int64_t)(f0 + fl) * (TB_C * TB_H) + ch * TB_H + pos] = y;
          } else {
#pragma unroll
            for (int p = 0; p < NPL; ++p) t[p] = 0u;
          }
#pragma unroll
          for (int p = 0; p < NPL; ++p) ir[p * TB_C * TB_KP + pos] = (unsigned short)t[p];
        }
        if (pbase + 2 >= TB_H) {   // the last view row of the frame (positions 511, 512, 513): the zero padding 514 .. 527 of the plane rows
#pragma unroll
          for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int z = TB_H + 1; z < TB_KP; z += 2) *reinterpret_cast<unsigned*>(ir + p * TB_C * TB_KP + z) = 0u;
        }
      }
      d0 += fl == 0 ? dsum : 0.f;
      d1 += fl == 1 ? dsum : 0.f;
    }
    d0 = wave_sum(d0);
    d1 = wave_sum(d1);
    if (lane == 0) {
      tpart[2][wave][0] = d0;
      tpart[2][wave][1] = d1;
    }
    __syncthreads();
    if (tid < nf) a.xh[(int64_t)(f0 + tid) * TB_H + (TB_H - 1)] = ((tpart[2][0][tid] + tpart[2][1][tid]) + (tpart[2][2][tid] + tpart[2][3][tid])) + a.bias3[0];
    if (a.zero_xh)   // uniform
      for (int i = tid; i < nf * (TB_H - 1); i += NTHR) a.xh[(int64_t)(f0 + i / (TB_H - 1)) * TB_H + i % (TB_H - 1)] = 0.f;
    // the image: the frames' planes are one contiguous run of yp
    {
      constexpr int PPF = NPL * TB_C * TB_KP / 8;       // 16-byte pieces per frame
      const u32x4* im4 = reinterpret_cast<const u32x4*>(img);
      u32x4* og = reinterpret_cast<u32x4*>(a.yp + (int64_t)f0 * (NPL * TB_C * TB_KP));
      for (int i = tid; i < nf * PPF; i += NTHR) og[i] = im4[i];
    }
#undef FC_TAIL_VAL
   }
  };
  // ---- OST: statistics of the group's result frames from the accumulators (+ bias), see the comment above the kernel
  auto ost_stats = [&](int nf) __attribute__((always_inline)) {
   if constexpr (OST) {
    // ONE pass over the accumulators: sums of (x - K) and (x - K)^2 about a pivot K taken from the data (this wave's first value), so the
    // centred square sum M2 = Q - S^2 / n loses nothing to the frame's mean; counts from ballots (scalar registers).  Few live registers and
    // few instructions on purpose: the kernel is compiled for three workgroups per CU (168 registers) with the next group's staged loads live
    // here, and it is bound by memory operations in flight -- serial vector work per group delays the next group's requests (the first,
    // two-pass version of this: dec2_fwd 204 -> 241 us).
    const int nrows = nf * V.R;
    const float K = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[0][0][0][0] + bv[0])));
    float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
    int c0 = 0, c1 = 0;
#pragma unroll
    for (int sidx = 0; sidx < NSW; ++sidx) {
      const int n = (wave + NWV * sidx) * T::SROWS + l31;
      const int nn = n < nrows ? n : 0;
      const int fl = nn / V.R;
      const int pbase = n < nrows ? (nn - fl * V.R) * V.oq + V.o0 : -8;    // rows past the group: every position out of range
#pragma unroll
      for (int p3 = 0; p3 < 3; ++p3) {
        const bool lv = pbase + p3 >= 0 && pbase + p3 < V.OH;
        float t = 0.f, u = 0.f;
#pragma unroll
        for (int cs = 0; cs < 4; ++cs) {
          const float d = (acc[sidx][0][0][cs + 4 * p3] + bv[cs]) - K;
          t += d;
          u = fmaf(d, d, u);
        }
        const bool a0 = lv && fl == 0, a1 = lv && fl == 1;
        s0 += a0 ? t : 0.f;
        q0 += a0 ? u : 0.f;
        s1 += a1 ? t : 0.f;
        q1 += a1 ? u : 0.f;
        c0 += 4 * __builtin_popcountll(__ballot(a0));
        c1 += 4 * __builtin_popcountll(__ballot(a1));
      }
    }
    s0 = wave_sum(s0);
    q0 = wave_sum(q0);
    s1 = wave_sum(s1);
    q1 = wave_sum(q1);
    if (lane == 0) {
      const float n0 = (float)(c0 > 0 ? c0 : 1), n1 = (float)(c1 > 0 ? c1 : 1);
      tpart[0][wave][0] = (float)c0;
      tpart[0][wave][1] = (float)c1;
      tpart[1][wave][0] = K + s0 / n0;          // the wave's mean
      tpart[1][wave][1] = K + s1 / n1;
      tpart[2][wave][0] = q0 - s0 * s0 / n0;    // centred square sum about it
      tpart[2][wave][1] = q1 - s1 * s1 / n1;
    }
   }
  };
  // ... combined behind the next barrier by one thread per frame (Chan): n = sum n_w, mean = sum n_w m_w / n, M2 = sum (M2_w + n_w (m_w - mean)^2)
  auto ost_combine = [&](int f0, int nf) __attribute__((always_inline)) {
   if constexpr (OST) {
    if (tid < nf) {
      float n = 0.f, sm = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        n += tpart[0][w4][tid];
        sm += tpart[0][w4][tid] * tpart[1][w4][tid];
      }
      const float mean = sm / n;
      float m2 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        const float d = tpart[1][w4][tid] - mean;
        m2 += tpart[2][w4][tid] + tpart[0][w4][tid] * d * d;
      }
      a.st2_out[2 * (f0 + tid)] = mean;
      a.st2_out[2 * (f0 + tid) + 1] = 1.0f / sqrtf(m2 / n + LN_EPS);
    }
   }
  };
  int pf0 = -1, pnf = 0;   // the group whose results are still in the accumulators
  for (; g < ngroups; g += gridDim.x) {
    const int f0 = g * T::TF, nf = min(T::TF, a.F - f0);
    if (VAENPVC_FC_DEFER && pf0 >= 0) {
      // the prefetched loads of this group must have landed BEFORE the deferred stores are issued (a wait behind them would
      // wait for them too): an empty use of every staging register puts the compiler's wait here
#pragma unroll
      for (int u = 0; u < IPW; ++u)
#pragma unroll
        for (int c = 0; c < T::CP; ++c) asm volatile("" ::"v"(v[u][c]));
      if constexpr (TAIL) epilogue_tail(pf0, pnf);
      else {
        epilogue(pf0, pnf);   // the stores first: they fly while the statistics are taken from the same accumulators
        ost_stats(pnf);
      }
    }
    fstore(g);
    __syncthreads();   // the group's frames are in LDS
    if (OST && pf0 >= 0) ost_combine(pf0, pnf);   // (the next triples are written behind the barrier at the end of this iteration)
    if (g + (int)gridDim.x < ngroups) fload(g + gridDim.x);
    // ---- GEMM rows n = fl * R + q, SROWS per step, steps dealt round-robin to the waves
    const int nrows = nf * V.R, nsteps = cdiv(nrows, T::SROWS);
#pragma unroll
    for (int sidx = 0; sidx < NSW; ++sidx) {
      const int s = wave + NWV * sidx;
      if (s >= nsteps) break;
      int xoff[T::NJ];
#pragma unroll
      for (int j = 0; j < T::NJ; ++j) {
        int n = s * T::SROWS + j * 32 + l31;
        n = n < nrows ? n : 0;                       // rows past the end: duplicates, never stored
        const int fl = n / V.R, q = n - fl * V.R;
        xoff[j] = fl * T::FS + q * T::RSTEP;
      }
#pragma unroll
      for (int i = 0; i < T::MT; ++i)
#pragma unroll
        for (int j = 0; j < T::NJ; ++j) acc[sidx][i][j] = zero16();
#pragma unroll
      for (int ks = 0; ks < ((VAENPVC_FC_ABL & 4) ? 0 : T::KS); ++ks) {
        u32x4 fa[T::MT][NPL], fb[T::NJ][NPL];
        const int ko = fc_koff<T::CP, T::CPL>(ks, lh);
#pragma unroll
        for (int i = 0; i < T::MT; ++i)
#pragma unroll
          for (int p = 0; p < NPL; ++p) fa[i][p] = *reinterpret_cast<const u32x4*>(ws + p * T::WPL + i * 32 * T::WP + woff + ks * 16);
#pragma unroll
        for (int j = 0; j < T::NJ; ++j)
#pragma unroll
          for (int p = 0; p < NPL; ++p) fb[j][p] = *reinterpret_cast<const u32x4*>(xs + p * T::XPL + xoff[j] + ko);
        using PR = Prod<NPL>;
        mfma_prio<8>(true);
#pragma unroll
        for (int t = 0; t < PR::N; ++t)
#pragma unroll
          for (int i = 0; i < T::MT; ++i)
#pragma unroll
            for (int j = 0; j < T::NJ; ++j) acc[sidx][i][j] = mfma_bf16(fa[i][PR::A[t]], fb[j][PR::B[t]], acc[sidx][i][j]);
        mfma_prio<8>(false);
      }
    }
    if (!VAENPVC_FC_DEFER) epilogue(f0, nf);
    pf0 = f0;
    pnf = nf;
    __syncthreads();   // all fragment reads of this group are done before the next one overwrites the tile
  }
  if constexpr (TAIL) {
    if (pf0 >= 0) epilogue_tail(pf0, pnf);
  } else {
    if (VAENPVC_FC_DEFER && pf0 >= 0) {
      epilogue(pf0, pnf);
      ost_stats(pnf);
      if constexpr (OST) {
        __syncthreads();
        ost_combine(pf0, pnf);
      }
    }
  }
}

template <int NPL, int SITE>
static void launch_fconv(const FcArgs& a, hipStream_t s) {
  using T = FcCfg<NPL, SITE>;
  const unsigned grid = (unsigned)cmin_(cdiv(a.F, T::TF), T::LDS > 80 * 1024 ? 256 : 256 * T::WGS_PER_CU);   // persistent: two (three) workgroups per CU walk the frame groups
  if constexpr (NPL == 1 && (SITE == CV_D1F || SITE == CV_D2F)) {
    if (a.bf_out) {   // bf16 activation storage (the statistics of the input are taken inside: LN = 2)
      constexpr bool BIN = SITE == CV_D2F;
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv<NPL, SITE, 2, BIN, true>), T::LDS);
      hipLaunchKernelGGL((k_fconv<NPL, SITE, 2, BIN, true>), dim3(grid), dim3(T::NTHR), T::LDS, s, a);
      return;
    }
  }
  if constexpr (SITE == CV_D2F && NPL <= 2 && fc_occ3(SITE)) {
    if (a.yp && a.st_out) {   // decoder tail in the epilogue: two workgroups per CU (frames + weights + plane image)
      constexpr int LDS_T = T::LDS + fc_tail_img_bytes(NPL, T::TF);
      const unsigned gridt = (unsigned)cmin_(cdiv(a.F, T::TF), 512);
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv<NPL, SITE, 2, false, false, true>), LDS_T);
      hipLaunchKernelGGL((k_fconv<NPL, SITE, 2, false, false, true>), dim3(gridt), dim3(T::NTHR), LDS_T, s, a);
      return;
    }
  }
  if constexpr (SITE == CV_D2F && fc_occ3(SITE)) {
    if (a.st2_out && !a.yp && a.st_out) {   // statistics of the RESULT from the accumulators (OST; those of the input in the staging: LN = 2)
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv<NPL, SITE, 2, false, false, false, true>), T::LDS);
      hipLaunchKernelGGL((k_fconv<NPL, SITE, 2, false, false, false, true>), dim3(grid), dim3(T::NTHR), T::LDS, s, a);
      return;
    }
  }
  if (a.st_out) {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv<NPL, SITE, 2>), T::LDS);
    hipLaunchKernelGGL((k_fconv<NPL, SITE, 2>), dim3(grid), dim3(T::NTHR), T::LDS, s, a);
  } else if (a.st) {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv<NPL, SITE, 1>), T::LDS);
    hipLaunchKernelGGL((k_fconv<NPL, SITE, 1>), dim3(grid), dim3(T::NTHR), T::LDS, s, a);
  } else {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv<NPL, SITE, 0>), T::LDS);
    hipLaunchKernelGGL((k_fconv<NPL, SITE, 0>), dim3(grid), dim3(T::NTHR), T::LDS, s, a);
  }
}
// site dispatch (only the thin sites are instantiated)
template <int NPL>
static bool fconv(int site, const FcArgs& a, hipStream_t s) {
  switch (site) {
    case CV_E1F: launch_fconv<NPL, CV_E1F>(a, s); return true;
    case CV_D1F: launch_fconv<NPL, CV_D1F>(a, s); return true;
    case CV_D2F: launch_fconv<NPL, CV_D2F>(a, s); return true;
    case CV_E1G: launch_fconv<NPL, CV_E1G>(a, s); return true;
    case CV_D1G: launch_fconv<NPL, CV_D1G>(a, s); return true;
    case CV_D2G: launch_fconv<NPL, CV_D2G>(a, s); return true;
    case CV_E2G:
      if constexpr (NPL <= 2) {
        launch_fconv<NPL, CV_E2G>(a, s);
        return true;
      }
      return false;
  }
  return false;
}
// served with `npl` operand planes (the medium site does not fit the LDS with three)
constexpr bool fconv_serves(int site, int npl) { return fc_tf(site) > 0 && !(site == CV_E2G && npl > 2); }

}  // namespace tuned
}  // namespace vaenpvc
