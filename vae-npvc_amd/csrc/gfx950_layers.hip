// gfx950_layers.hip -- the tuned MI355X path for the VCC2016 geometry
// (architecture-vae-vcc2016.json): layer configurations of the four engines and the
// per-step orchestration of forward and backward.  Every step can be switched back to
// the geometry-generic kernel (set_masks) to isolate a fault on the GPU.
//
// Step list and the reference code each step replaces:
//   e0..e4  conv2d_nchw_layernorm x5        util/layers.py:47-66 (model/vae.py:74-78)
//   heads   two dense heads                  model/vae.py:79-81
//   merge   embedding lookup + _merge        model/vae.py:51-61,89-90
//   d0..d2  conv2d_transpose + LN + lrelu    model/vae.py:96-102
//   d3      conv2d_transpose k=1025          model/vae.py:96-99
//   backward steps = autodiff of the same    trainer/vae.py:24
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "gfx950_convgemm.h"
#include "gfx950_convwgrad.h"
#include "gfx950_dense.h"
#include "gfx950_elem.h"
#include "gfx950_tngemm.h"
#include "gfx950_toeplitz.h"
#include "gfx950_toep_bf16.h"
#include "gfx950_planegemm.h"
#include "gfx950_ntring.h"
#include "gfx950_viewconv.h"
#include "gfx950_fconv.h"
#include "gfx950_fwgrad.h"
#include "gfx950_fconv_r.h"
#include "gfx950_fbwd.h"
#include "gfx950_lnb_planes.h"
#ifndef VAENPVC_LNB_PLANES
#define VAENPVC_LNB_PLANES 15   // LayerNorm backward writing its consumers' operand planes itself: 1 encoder layer 4, 2 encoder layer 3, 4 decoder layer 0,
                               // 8 decoder layer 0 without the fp32 copy (its input-gradient kernel reads the planes)
#endif
#ifndef VAENPVC_SPLIT_SEGSUM
#define VAENPVC_SPLIT_SEGSUM 1
#endif
#include "kernels.h"

namespace vaenpvc {
namespace tuned {

// ---------------------------------------------------------------- configurations
// ConvCfg<KC, HIN, N, HOUT, T, S, PAD, kind, TF, in-kind, lndiv, MB, NB, NW>; the tiling (TF = frames per tile,
// MB x NB = MFMA tiles per work item, NW = waves) is overridable per layer for tuning sweeps
// (scripts/build_variant.sh NAME "-DE2F_T=6,1,1,8"):  X_T = TF, MB, NB, NW
#ifndef E1F_T
#define E1F_T 4, 1, 1, 8
#endif
#ifndef E2F_T
#define E2F_T 12, 2, 1, 8
#endif
#ifndef E3F_T
#define E3F_T 18, 2, 1, 8
#endif
#ifndef E4F_T
#define E4F_T 21, 2, 1, 8
#endif
#ifndef D0F_T
#define D0F_T 13, 2, 1, 8
#endif
#ifndef D1F_T
#define D1F_T 4, 1, 2, 8
#endif
#ifndef D2F_T
#define D2F_T 4, 3, 1, 8
#endif
#ifndef GD2_T
#define GD2_T 4, 3, 1, 8
#endif
#ifndef GD1_T
#define GD1_T 4, 1, 1, 8
#endif
#ifndef GD0_T
#define GD0_T 13, 1, 1, 8
#endif
#ifndef GE4_T
#define GE4_T 16, 1, 1, 8
#endif
#ifndef GE3_T
#define GE3_T 18, 2, 1, 8
#endif
#ifndef GE2_T
#define GE2_T 13, 2, 1, 8
#endif
#ifndef GE1_T
#define GE1_T 4, 1, 2, 8
#endif
template <int KC, int HIN, int N, int HOUT, int T, int S, int PAD, int KIND, int INKIND, int TF, int MB, int NB, int NW>
using ConvT = ConvCfg<KC, HIN, N, HOUT, T, S, PAD, KIND, TF, INKIND, 1, MB, NB, NW>;
using E1F = ConvT<16, 171, 32, 57, 7, 3, 2, CONV_S, IN_LN, E1F_T>;
using E2F = ConvT<32, 57, 64, 19, 7, 3, 2, CONV_S, IN_LN, E2F_T>;
using E3F = ConvT<64, 19, 128, 7, 7, 3, 3, CONV_S, IN_LN, E3F_T>;
using E4F = ConvT<128, 7, 256, 3, 7, 3, 3, CONV_S, IN_LN, E4F_T>;
using D0F = ConvT<81, 19, 32, 57, 9, 3, 3, CONV_P, IN_PLAIN, D0F_T>;
using D1F = ConvT<32, 57, 16, 171, 7, 3, 2, CONV_PM, IN_LN, D1F_T>;
using D2F = ConvT<16, 171, 8, 513, 7, 3, 2, CONV_PM, IN_LN, D2F_T>;
// input gradients: conv_transpose layers (S-type) and conv layers (transposed kinds)
using GD2 = ConvT<8, 513, 16, 171, 7, 3, 2, CONV_S, IN_PLAIN, GD2_T>;
using GD1 = ConvT<16, 171, 32, 57, 7, 3, 2, CONV_S, IN_PLAIN, GD1_T>;
using GD0 = ConvT<32, 57, 81, 19, 9, 3, 3, CONV_S, IN_PLAIN, GD0_T>;
using GE4 = ConvT<256, 3, 128, 7, 7, 3, 3, CONV_P, IN_PLAIN, GE4_T>;
using GE3 = ConvT<128, 7, 64, 19, 7, 3, 3, CONV_P, IN_PLAIN, GE3_T>;
using GE2 = ConvT<64, 19, 32, 57, 7, 3, 2, CONV_P, IN_PLAIN, GE2_T>;
using GE1 = ConvT<32, 57, 16, 171, 7, 3, 2, CONV_PM, IN_PLAIN, GE1_T>;

// small-batch tilings (F < SMALL_BATCH_FRAMES): fewer frames per tile = more workgroups; the packed weights
// do not depend on the tiling, so both variants of a layer read the same copy
constexpr int64_t SMALL_BATCH_FRAMES = 2048;
using E3Fs = ConvT<64, 19, 128, 7, 7, 3, 3, CONV_S, IN_LN, 9, 1, 1, 8>;
using D0Fs = ConvT<81, 19, 32, 57, 9, 3, 3, CONV_P, IN_PLAIN, 8, 1, 1, 8>;
using GD0s = ConvT<32, 57, 81, 19, 9, 3, 3, CONV_S, IN_PLAIN, 8, 1, 1, 8>;
using GE3s = ConvT<128, 7, 64, 19, 7, 3, 3, CONV_P, IN_PLAIN, 18, 1, 1, 8>;
using GE2s = ConvT<64, 19, 32, 57, 7, 3, 2, CONV_P, IN_PLAIN, 8, 1, 1, 8>;
static_assert(E3Fs::BTOTAL == E3F::BTOTAL && D0Fs::BTOTAL == D0F::BTOTAL && GD0s::BTOTAL == GD0::BTOTAL &&
                  GE3s::BTOTAL == GE3::BTOTAL && GE2s::BTOTAL == GE2::BTOTAL,
              "both tilings of a layer share one packed weight copy");

using HeadsF = DenseCfg<768, 256, 256, 2, IN_LN, 3>;
using HeadsB = DenseCfg<256, 768, 256, 3, IN_CONCAT2, 1>;
// merge: h = z Wz + T[y] (K = 128, the speaker's table row added in the epilogue); dz = dh Wz^T (N = 128)
using MergeF = DenseCfg<128, 1539, 128, 2, IN_PLAIN, 1>;
using MergeB = DenseCfg<1539, 128, 256, 1, IN_PLAIN, 1>;
using HeadsFs = DenseCfg<768, 256, 128, 2, IN_LN, 3, 1>;   // (K chunks of 128: six-way split-K at small batches)
using HeadsBs = DenseCfg<256, 768, 64, 3, IN_CONCAT2, 1, 1>;    // (K chunks of 64: four-way split-K)
using MergeFs = DenseCfg<128, 1539, 128, 2, IN_PLAIN, 1, 1>;
using MergeBs = DenseCfg<1539, 128, 256, 1, IN_PLAIN, 1, 1>;
constexpr int MERGE_NY = 10;   // speakers of the VCC2016 geometry (the tuned path is selected for it only)
//                     XC  XH   YC  YH  T  S PAD  XLN    YLN   TF NTW
// weight-gradient tilings: trailing parameters = TF (frames per sub-tile), NTW (column tiles per workgroup),
// NWV (waves), WM (waves along M, 0 = auto), WPE (waves per SIMD the register budget must allow);
// overridable per layer for tuning sweeps (scripts/build_variant.sh NAME "-DWD2_T=4,1,4,0,2")
#ifndef WD2_T
#define WD2_T 2, 1, 4, 2, 2, true
#endif
#ifndef WD1_T
#define WD1_T 4, 1, 8, 0, 2
#endif
#ifndef WD0_T
#define WD0_T 4, 3, 4, 0, 1
#endif
#ifndef WE4_T
#define WE4_T 8, 1, 4, 0, 2
#endif
#ifndef WE3_T
#define WE3_T 8, 2, 4, 0, 2
#endif
#ifndef WE2_T
#define WE2_T 4, 2, 4, 0, 2
#endif
#ifndef WE1_T
#define WE1_T 4, 1, 8, 0, 2
#endif
#ifndef WE0_T
#define WE0_T 2, 1, 4, 1, 2, true
#endif
using WD2 = WgCfg<8, 513, 16, 171, 7, 3, 2, false, true, WD2_T>;
using WD1 = WgCfg<16, 171, 32, 57, 7, 3, 2, false, true, WD1_T>;
using WD0 = WgCfg<32, 57, 81, 19, 9, 3, 3, false, false, WD0_T>;
using WE4 = WgCfg<128, 7, 256, 3, 7, 3, 3, true, false, WE4_T>;
using WE3 = WgCfg<64, 19, 128, 7, 7, 3, 3, true, false, WE3_T>;
using WE2 = WgCfg<32, 57, 64, 19, 7, 3, 2, true, false, WE2_T>;
using WE1 = WgCfg<16, 171, 32, 57, 7, 3, 2, true, false, WE1_T>;
using WE0 = WgCfg<1, 513, 16, 171, 7, 3, 2, false, false, WE0_T>;

// packed-weight scratch layout (float offsets)
struct Pk {
  static constexpr int heads_f = 0;
  static constexpr int heads_b = heads_f + HeadsF::KP * HeadsF::NP;
  static constexpr int merge_f = heads_b + HeadsB::KP * HeadsB::NP;
  static constexpr int merge_b = merge_f + MergeF::KP * MergeF::NP;
  static constexpr int merge_tb = merge_b + MergeB::KP * MergeB::NP;   // T[ny][1539]: E Wy + the three biases
  static constexpr int merge_s = merge_tb + MERGE_NY * 1539 + 2;        // S[ny][1539]: per-speaker sums of d(h)
  static constexpr int d0f = merge_s + MERGE_NY * 1539 + 2;
  static constexpr int d1f = d0f + D0F::BTOTAL;
  static constexpr int d2f = d1f + D1F::BTOTAL;
  static constexpr int gd2 = d2f + D2F::BTOTAL;
  static constexpr int gd0 = gd2 + GD2::BTOTAL;
  static constexpr int ge4 = gd0 + GD0::BTOTAL;
  static constexpr int ge3 = ge4 + GE4::BTOTAL;
  static constexpr int ge2 = ge3 + GE3::BTOTAL;
  static constexpr int ge1 = ge2 + GE2::BTOTAL;
  static constexpr int wc = ge1 + GE1::BTOTAL;
  static constexpr int lnpart = wc + TOEP_C * WROW;  // [LWGS][3][C] partial sums of the LN backward
  // BORROWED between the forward and the backward pass of a train step: the fused loss kernel (loss_fwd_post) parks the last layer's
  // per-workgroup BIAS parts here; backward() adds them (k_colsum_part) before its first LayerNorm pass re-uses the region.  Valid
  // while Runtime::dxh_post_F == F; a kernel placed between the loss and the backward pass must not touch it.
  static constexpr int wdg = lnpart + 2048 * 3 * 256;  // bf16 tap copies, input-gradient direction
  static constexpr int wfw = wdg + TB_WFLOATS;  // bf16 tap copies, forward direction (reversed)
  static constexpr int heads_bias = wfw + TB_WFLOATS;  // [b_mu | b_lv]
  // weight planes of the dense-shaped layers (gfx950_planegemm.h): up to 3 planes of [Np][Kp] unsigned short
  static constexpr int pg_headsf = heads_bias + 256;            // [256][768]
  static constexpr int pg_headsb = pg_headsf + 3 * 256 * 768 / 2;   // [768][256]
  static constexpr int pg_mergef = pg_headsb + 3 * 768 * 256 / 2;   // [1664][128]
  static constexpr int pg_mergeb = pg_mergef + 3 * 1664 * 128 / 2;  // [128][1600]
  static constexpr int pg_enc4f = pg_mergeb + 3 * 128 * 1600 / 2;   // [768][896]
  static constexpr int pg_enc4b = pg_enc4f + 3 * 768 * 896 / 2;     // [896][768]
  static constexpr int pg_bias4 = pg_enc4b + 3 * 896 * 768 / 2;     // bias of layer 4 per dense column (o, j)
  static constexpr int cvw = pg_bias4 + 768;                        // weight planes of the conv view-GEMM sites
  static constexpr int cvwr_d0f = cvw + CV_WTOTAL;                  // phase-permuted weight planes of the register-weight fused kernel
  static constexpr int cvwr_e2g = cvwr_d0f + fcr_wfloats(CV_D0F);
  static constexpr int enc0part_ = cvwr_e2g + fcr_wfloats(CV_E2G);
  static constexpr int enc0part = enc0part_;                  // [512][7*16] partial weight gradients of encoder layer 0
  static constexpr int total = enc0part + 512 * 7 * 16;
};
static_assert(Pk::total <= 8 * 939162 + 65536, "packed weights must fit the scratch region");
// the dense-shaped layers (heads, merge, encoder layer 4) on the plane GEMM kernels: bit 29 of the masks, and enough
// frames to fill 128-row tiles (below, the exact-fp32 kernels with their 32-frame tiles spread better)
constexpr int64_t PLANEGEMM_MIN_FRAMES = 1024;
// (bit 28 cleared: at any batch size -- parity tests)
static inline bool pg_on(unsigned mask, int64_t F) { return ((mask >> 29) & 1u) && (F >= PLANEGEMM_MIN_FRAMES || !((mask >> 28) & 1u)); }
static inline bool pg_fwd(int64_t F) { return pg_on(rt().fwd_mask, F); }
// conv layers as view GEMMs: bit 27 (same batch-size rule)
static inline bool cg_fwd(int64_t F) { return ((rt().fwd_mask >> 27) & 1u) && pg_on(rt().fwd_mask | (1u << 29), F); }
static inline bool cg_bwd(int64_t F) { return ((rt().bwd_mask >> 27) & 1u) && pg_on(rt().bwd_mask | (1u << 29), F); }
static inline bool pg_bwd(int64_t F) { return pg_on(rt().bwd_mask, F); }
// ... per site: the context's site set (runtime.h: cv_sites), or every site when bit 26 of the mask is cleared
static inline bool cv_sel(unsigned mask, int site) { return !((mask >> 26) & 1u) || ((rt().cv_sites() >> site) & 1u); }
static inline bool cv_fwd(int site, int64_t F) { return cg_fwd(F) && cv_sel(rt().fwd_mask, site); }
static inline bool cv_bwd(int site, int64_t F) { return cg_bwd(F) && cv_sel(rt().bwd_mask, site); }
// thin conv sites on the fused kernel (gfx950_fconv.h): the context's site set from FCONV_MIN_FRAMES frames on, or every
// served site at any batch size when bit 25 of the mask is cleared (parity tests)
constexpr int64_t FCONV_MIN_FRAMES = 1024;
constexpr int64_t ENC0_WAVE_MIN_FRAMES = 1024;   // encoder layer 0 on the wave-per-frame kernels (fewer frames: too few waves)
// (bit 23 of a mask cleared: at any batch size -- parity tests)
static inline bool enc0_wave(unsigned mask, int64_t F) { return F >= ENC0_WAVE_MIN_FRAMES || !((mask >> 23) & 1u); }
static inline bool fc_on(unsigned mask, int site, int64_t F) {
  if (!fconv_serves(site, rt().dense_planes ? rt().dense_planes : rt().planes)) return false;
  if (!((mask >> 25) & 1u)) return true;
  return ((rt().fc_sites() >> site) & 1u) && F >= FCONV_MIN_FRAMES;
}
// medium sites on the register-weight fused kernel (gfx950_fconv_r.h): bit 22 of a mask cleared = at any batch size
static inline bool fcr_on(unsigned mask, int site, int64_t F) {
  if (!fcr_serves_site(site) || (rt().dense_planes ? rt().dense_planes : rt().planes) > 2) return false;
  if (!((mask >> 22) & 1u)) return true;
  return ((rt().fcr_sites() >> site) & 1u) && F >= FCONV_MIN_FRAMES;
}
static inline bool fcr_fwd(int site, int64_t F) { return fcr_on(rt().fwd_mask, site, F); }
static inline bool fcr_bwd(int site, int64_t F) { return fcr_on(rt().bwd_mask, site, F); }
static inline bool fc_fwd(int site, int64_t F) { return !fcr_fwd(site, F) && fc_on(rt().fwd_mask, site, F); }
static inline bool fc_bwd(int site, int64_t F) { return !fcr_bwd(site, F) && fc_on(rt().bwd_mask, site, F); }
// thin weight gradients on the fused kernel (gfx950_fwgrad.h): bit 24 of the backward mask cleared = every served site at
// any batch size (parity tests)
static inline bool fw_bwd(int wsite, int64_t F) {
  if (!fwgrad_serves(wsite) || (rt().dense_planes ? rt().dense_planes : rt().planes) > 2) return false;
  if (!((rt().bwd_mask >> 24) & 1u)) return true;
  return ((rt().fw_sites() >> wsite) & 1u) && F >= FCONV_MIN_FRAMES;
}
// whole backward step of a thin decoder layer in one kernel (gfx950_fbwd.h): bit 15 of the backward mask (default set) and
// the context's layer set from FCONV_MIN_FRAMES frames on; bit 14 cleared = at any batch size (parity tests)
static inline bool fb_bwd(int layer, int64_t F) {
  if ((rt().dense_planes ? rt().dense_planes : rt().planes) > VAENPVC_FB_MAXPL || !((rt().bwd_mask >> 15) & 1u)) return false;
  if (!((rt().fb_layers() >> layer) & 1u)) return false;
  return F >= FCONV_MIN_FRAMES || !((rt().bwd_mask >> 14) & 1u);
}
static bool toep_bf16_for(int64_t F);
// bf16 ACTIVATION STORAGE (precision "bf16" only, gfx950_toep_bf16.h: act_pitch): the pre-LN outputs of decoder layers 1 and 2
// and the gradients at their activated outputs live in HBM as bf16 when every producer and consumer of them is one of the
// kernels that knows the format -- the default selection from FCONV_MIN_FRAMES frames on (VAENPVC_ACT_BF16=0: fp32 storage)
static inline bool act_bf16(int64_t F) {
  Runtime& r = rt();
  if (!r.act_bf16 || r.planes != 1 || (r.dense_planes && r.dense_planes != 1) || F < FCONV_MIN_FRAMES) return false;
  const unsigned need = (1u << 8) | (1u << 9) | (1u << 10);
  if ((r.fwd_mask & need) != need || (r.bwd_mask & need) != need) return false;
  return fc_fwd(CV_D1F, F) && fc_fwd(CV_D2F, F) && toep_bf16_for(F) && fb_bwd(FB_D2, F) && fb_bwd(FB_D1, F);
}
static inline bool fc_any(int64_t F) {
  for (int i = 0; i < CV_COUNT; ++i)
    if (fc_fwd(i, F) || fc_bwd(i, F) || fcr_fwd(i, F) || fcr_bwd(i, F)) return true;
  return false;
}
// the forward pass left the channel-last planes of a site's input behind (view GEMM selected and not overridden by the fused kernel)
static inline bool fwd_planes(int bit, int site, int64_t F) {
  return ((rt().fwd_mask >> bit) & 1u) && cv_fwd(site, F) && !fc_fwd(site, F) && !fcr_fwd(site, F);
}
static inline bool cw_bwd(int site, int64_t F) { return cg_bwd(F) && cv_sel(rt().bwd_mask, CV_COUNT + site); }
#ifndef VAENPVC_E2F_STATS
#define VAENPVC_E2F_STATS 1   // encoder layer 2's fused forward kernel computes the LayerNorm statistics of its input itself
#endif
#ifndef VAENPVC_DZ_PLANES
#define VAENPVC_DZ_PLANES 1   // the sampler backward writes the planes of [dz_mu | dz_lv] itself
#endif
#ifndef VAENPVC_TN_DENSE_XCD
#define VAENPVC_TN_DENSE_XCD 1   // dense-shaped C += A^T B sites: the tiles of a row chunk on one XCD (they share the narrow operand)
#endif
#ifndef VAENPVC_D0F_CLOUT
#define VAENPVC_D0F_CLOUT 1
#endif
// decoder layer 0's fused forward kernel also writes the channel-last planes of its input h (operand of the layer's view weight gradient)
static inline bool d0f_leaves_planes(int64_t F) {
  return VAENPVC_D0F_CLOUT && F >= 1024 && ((rt().fwd_mask >> 7) & 1u) && fcr_fwd(CV_D0F, F) && ((rt().bwd_mask >> 7) & 1u) && cw_bwd(CW_D0, F);
}
static inline unsigned short* us(float* p) { return reinterpret_cast<unsigned short*>(p); }
static NtArgs nt_args(const float* Ap, int M, int Kp, const float* Bp, int Np, int N, float* C, int ldc) {
  NtArgs a;
  memset(&a, 0, sizeof a);
  a.A = reinterpret_cast<const unsigned short*>(Ap);
  a.B = reinterpret_cast<const unsigned short*>(Bp);
  a.a_plane = (int64_t)M * Kp;
  a.b_plane = (int64_t)Np * Kp;
  a.M = M;
  a.N = N;
  a.Kp = Kp;
  a.C = C;
  a.ldc = ldc;
  return a;
}
// frame chunks of the heads / merge weight gradients (two-dimensional tilings with 6 / 7 tiles): every chunk ends in one
// atomic per tile entry, all chunks at the same moment.  Measured at 32 768 frames (heads / merge, us): 512 -> 83.5 / 89.4,
// 384 -> 69.2 / 80.2, 256 -> 75.3 / 89.7
#ifndef TN_WGS_DENSE
#define TN_WGS_DENSE 384
#endif
static TnpArgs tnp_args(const float* Ap, int lda, const float* Bp, int ldb, int M, int N, int F, float* C, int ldc) {
  TnpArgs a;
  memset(&a, 0, sizeof a);
  a.A = reinterpret_cast<const unsigned short*>(Ap);
  a.B = reinterpret_cast<const unsigned short*>(Bp);
  a.a_plane = (int64_t)F * lda;
  a.b_plane = (int64_t)F * ldb;
  a.av = plain_rows(lda);
  a.bv = plain_rows(ldb);
  a.lda = lda;
  a.ldb = ldb;
  a.M = M;
  a.N = N;
  a.F = F;
  a.C = C;
  a.ldc = ldc;
  // XCD-aware tile order: measured -15 % where a row chunk has many tiles sharing both operands (layer 4: 7 x 3), +5..10 %
  // on the one-dimensional tilings (merge 1 x 7, heads 6 x 1)
  a.xcd = rt().tn_xcd >= 0 ? rt().tn_xcd : (cdiv(M, 128) * cdiv(N, 256) > 1 ? VAENPVC_TN_DENSE_XCD : 0);
  a.tn4 = 0;
  return a;
}
// layers whose TF kernel tensor IS the packed operand (no copy)
static_assert(E1F::BTOTAL == 7 * 16 * 32 && E2F::BTOTAL == 7 * 32 * 64 && E3F::BTOTAL == 7 * 64 * 128 &&
                  E4F::BTOTAL == 7 * 128 * 256 && GD1::BTOTAL == 7 * 16 * 32,
              "direct-use layers must not be padded");

// Kernel selection lives in the calling context's Runtime (runtime.h): no process-global state.
//   rt().toep_f32 (VAENPVC_TOEP=f32 at context creation) selects the exact-fp32 MFMA kernels of the last decoder
//   layer instead of the bf16-split ones (kept for A/B measurements).
// The bf16 kernels own 64 frames per workgroup: below ~8k frames they cannot fill the chip and the
// fp32 kernels (32 frames per workgroup, bins split over more workgroups) are faster.  Clearing bit 30
// of the forward mask forces them at any batch size (parity tests).
constexpr int64_t TOEP_BF16_MIN_FRAMES = 16;
// channel groups of the bf16 Toeplitz GEMM kernels: each workgroup owns 64 frames, so small batches spread the 8
// channels over up to 8 workgroups per frame tile (>= 256 workgroups where the batch allows it)
static int toep_groups(int64_t F) {
  const int wgs = cdiv((int)F, DG_M);
  int g = 1;
  while (g < TB_C && wgs * g < 256) g *= 2;
  return g;
}
// the forward direction combines its channel groups with fp32 atomics (order-dependent rounding): only inside a
// train / loss step; a stand-alone decode (conversion path) stays bitwise reproducible with one group
static int toep_fwd_groups(int64_t F, bool in_step) { return in_step ? toep_groups(F) : 1; }
bool available() { return true; }
static inline bool fwd_on(int bit) { return (rt().fwd_mask >> bit) & 1u; }
static inline bool bwd_on(int bit) { return (rt().bwd_mask >> bit) & 1u; }
static bool toep_bf16_for(int64_t F) { return !rt().toep_f32 && (F >= TOEP_BF16_MIN_FRAMES || !fwd_on(30)); }
// the weight gradient of that layer reads both operands as bf16 planes (then the fp32 copy of y is not stored)
static bool toep_wgrad_bf16_for(int64_t F) { return toep_bf16_for(F) && fwd_on(9) && fwd_on(10) && !rt().toep_wgrad_f32; }
static inline void read_env() {}


// dispatch on the context's operand precision: fn(std::integral_constant<int, NPL>), NPL = bf16 terms per fp32 operand
// of every kernel on the bf16 matrix cores.  2 (default): 16 mantissa bits per operand -- measured 1.3e-5 of a
// gradient tensor's scale at the worst (against the 2e-4 bar, with the lrelu kink branches pinned: an unpinned
// comparison at small batch sizes measures which side of a kink a 1e-5 rounding error falls on, not arithmetic);
// 3: fp32-exact; 1: plain bf16 (the bf16 mode).  VAENPVC_DENSE_PLANES overrides the dense-shaped layers (experiments).
template <class Fn>
static void for_planes(Fn&& fn) {
  switch (rt().planes) {
    case 1: fn(std::integral_constant<int, 1>{}); break;
    case 3: fn(std::integral_constant<int, 3>{}); break;
    default: fn(std::integral_constant<int, 2>{}); break;
  }
}
constexpr int dense_planes(int npl) { return npl; }
static inline int dense_planes_now() { return rt().dense_planes ? rt().dense_planes : dense_planes(rt().planes); }
template <class Fn>
static void for_dense_planes(Fn&& fn) {
  const int p = rt().dense_planes ? rt().dense_planes : dense_planes(rt().planes);
  if (p == 1) fn(std::integral_constant<int, 1>{});
  else if (p == 2) fn(std::integral_constant<int, 2>{});
  else fn(std::integral_constant<int, 3>{});
}

// ---------------------------------------------------------------- weight packing
static void prep(const Model& m, const float* P, const Ws& w, int64_t F, hipStream_t s) {
  float* S = w.scratch;
  constexpr int NTB = TB_C * TB_CPY * 8 * TB_CHUNKS;
  // one launch for all packed copies (see k_pack_multi)
  for_planes([&](auto npl) {
  constexpr int NPL = decltype(npl)::value;
  for_dense_planes([&](auto npd) {
  constexpr int NPD = decltype(npd)::value;
  // (plane copies of the dense-shaped layers only when the plane GEMMs run at this batch size)
  const bool dense_planes_used = pg_fwd(F) || pg_bwd(F);
  auto pj = [&](auto j) {
    if (!dense_planes_used) j.count = 0;
    return j;
  };
  VAENPVC_TIMED("prep", s, launch_pack_multi(
      s,
      pack_job(PackDense{P + m.wmu_off, P + m.wlv_off, 0, 768, 256, HeadsF::NP, 128, 128}, S + Pk::heads_f, HeadsF::KP * HeadsF::NP),
      pack_job(PackDense{P + m.wmu_off, P + m.wlv_off, 2, 256, 768, HeadsB::NP, 128, 128}, S + Pk::heads_b, HeadsB::KP * HeadsB::NP),
      pack_job(PackDense{P + m.wz_off, P + m.wz_off, 1, 128, 1539, MergeF::NP, 128, 0}, S + Pk::merge_f, MergeF::KP * MergeF::NP),
      pack_job(PackDense{P + m.wz_off, P + m.wz_off, 3, 1539, 128, MergeB::NP, 128, 0}, S + Pk::merge_b, MergeB::KP * MergeB::NP),
      pack_job(PackMergeTable{P + m.emb_off, P + m.wy_off, P + m.bz_off, P + m.by_off, P + m.bm_off, m.z, m.merge},
               S + Pk::merge_tb, MERGE_NY * 1539),
      pack_job(PackCat2{P + m.bmu_off, P + m.blv_off, 128}, S + Pk::heads_bias, 256),
      // conv_transpose forward: B[t][k=cin][n=cout] from TF [t][cout][cin]  -> transposed
      pack_job(PackConv<D0F>{P + m.dec[0].w_off, true}, S + Pk::d0f, D0F::BTOTAL),
      pack_job(PackConv<D1F>{P + m.dec[1].w_off, true}, S + Pk::d1f, D1F::BTOTAL),
      pack_job(PackConv<D2F>{P + m.dec[2].w_off, true}, S + Pk::d2f, D2F::BTOTAL),
      // conv_transpose input-gradient: B[t][k=cout][n=cin] = TF layout (padded copies only)
      pack_job(PackConv<GD2>{P + m.dec[2].w_off, false}, S + Pk::gd2, GD2::BTOTAL),
      pack_job(PackConv<GD0>{P + m.dec[0].w_off, false}, S + Pk::gd0, GD0::BTOTAL),
      // conv input-gradient: B[t][k=cout][n=cin] from TF [t][cin][cout] -> transposed
      pack_job(PackConv<GE4>{P + m.enc[4].w_off, true}, S + Pk::ge4, GE4::BTOTAL),
      pack_job(PackConv<GE3>{P + m.enc[3].w_off, true}, S + Pk::ge3, GE3::BTOTAL),
      pack_job(PackConv<GE2>{P + m.enc[2].w_off, true}, S + Pk::ge2, GE2::BTOTAL),
      pack_job(PackConv<GE1>{P + m.enc[1].w_off, true}, S + Pk::ge1, GE1::BTOTAL),
      pack_job(PackToep{P + m.dec[3].w_off}, S + Pk::wc, TOEP_C * WROW),
      // shifted bf16 tap copies of the last layer (count 0 = skipped when the fp32 kernels are selected)
      PackToepBf16Job<false, NPL>{P + m.dec[3].w_off, reinterpret_cast<unsigned short*>(S + Pk::wdg), !rt().toep_f32 ? NTB : 0},
      PackToepBf16Job<true, NPL>{P + m.dec[3].w_off, reinterpret_cast<unsigned short*>(S + Pk::wfw), !rt().toep_f32 ? NTB : 0},
      // weight planes of the dense-shaped layers
      pj(planes_job<NPD>(WHeadsF{P + m.wmu_off, P + m.wlv_off}, S + Pk::pg_headsf, 256, 768)),
      pj(planes_job<NPD>(WHeadsB{P + m.wmu_off, P + m.wlv_off}, S + Pk::pg_headsb, 768, 256)),
      pj(planes_job<NPD>(WMergeF{P + m.wz_off, 1539}, S + Pk::pg_mergef, 1664, 128)),
      pj(planes_job<NPD>(WMergeB{P + m.wz_off, 1539}, S + Pk::pg_mergeb, 128, 1600)),
      pj(planes_job<NPD>(WEnc4F{P + m.enc[4].w_off}, S + Pk::pg_enc4f, 768, 896)),
      pj(planes_job<NPD>(WEnc4B{P + m.enc[4].w_off}, S + Pk::pg_enc4b, 896, 768)),
      pack_job(PackRepeat3{P + m.enc[4].b_off}, S + Pk::pg_bias4, 768)));
  // weight planes of the conv view-GEMM sites.  TF layouts: conv [T][Cin][Cout], conv_transpose [T][Cout][Cin];
  // (s_t, s_o, s_c) = strides of (tap, GEMM output channel, contracted channel)
  if (cg_fwd(F) || cg_bwd(F) || fc_any(F) || fb_bwd(FB_D2, F) || fb_bwd(FB_D1, F) || fb_bwd(FB_E1, F)) {
    // only the sites some kernel of this step reads (count 0 = job skipped)
    auto used = [&](int site, bool fwd_dir) {
      if (!fwd_dir && ((site == CV_D2G && fb_bwd(FB_D2, F)) || (site == CV_D1G && fb_bwd(FB_D1, F)) || (site == CV_E1G && fb_bwd(FB_E1, F)))) return true;   // gfx950_fbwd.h
      return fwd_dir ? (cv_fwd(site, F) || fc_fwd(site, F) || fcr_fwd(site, F)) : (cv_bwd(site, F) || fc_bwd(site, F) || fcr_bwd(site, F));
    };
    auto job = [&](int site, bool fwd_dir, const ConvL& l, int s_o, int s_c) {
      auto j = cv_job<NPD>(site, P + l.w_off, l.cin * l.cout, s_o, s_c, S + Pk::cvw + cv_woff(site));
      if (!used(site, fwd_dir)) j.count = 0;
      return j;
    };
    auto ef = [&](int site, int i) { const ConvL& l = m.enc[i]; return job(site, true, l, 1, l.cout); };
    auto eg = [&](int site, int i) { const ConvL& l = m.enc[i]; return job(site, false, l, l.cout, 1); };
    auto df = [&](int site, int i) { const ConvL& l = m.dec[i]; return job(site, true, l, l.cin, 1); };
    auto dg = [&](int site, int i) { const ConvL& l = m.dec[i]; return job(site, false, l, 1, l.cin); };
    {  // phase-permuted copies for the register-weight kernel
      auto pd = fcr_perm_job<NPD>(CV_D0F, P + m.dec[0].w_off, m.dec[0].cin * m.dec[0].cout, m.dec[0].cin, 1, S + Pk::cvwr_d0f);
      auto pe = fcr_perm_job<NPD>(CV_E2G, P + m.enc[2].w_off, m.enc[2].cin * m.enc[2].cout, m.enc[2].cout, 1, S + Pk::cvwr_e2g);
      if (!fcr_fwd(CV_D0F, F)) pd.count = 0;
      if (!fcr_bwd(CV_E2G, F)) pe.count = 0;
      if (pd.count || pe.count) VAENPVC_TIMED("prep", s, launch_pack_multi(s, pd, pe));
    }
    VAENPVC_TIMED("prep", s, launch_pack_multi(s, ef(CV_E1F, 1), ef(CV_E2F, 2), ef(CV_E3F, 3), df(CV_D0F, 0), df(CV_D1F, 1), df(CV_D2F, 2),
                      eg(CV_E3G, 3), eg(CV_E2G, 2), eg(CV_E1G, 1), dg(CV_D0G, 0), dg(CV_D1G, 1), dg(CV_D2G, 2)));
  }
  });
  });
}

// statistics of encoder layer 3 / 4 outputs; when the next layer runs on the plane GEMM kernels the activated planes
// (canonical [F][C*H] order) are written in the same pass
template <int N, int H>
static void stats_planes(const float* a, float* st, const float* gamma, const float* beta, float* planes, bool want, int F,
                         hipStream_t s) {
  if (!want) {
    hipLaunchKernelGGL(k_ln_stats_fast<N>, dim3((unsigned)cdiv(F, 4)), dim3(256), 0, s, a, st, F);
    return;
  }
  for_dense_planes([&](auto npl) {
    hipLaunchKernelGGL((k_ln_stats_planes<N, H, decltype(npl)::value>), dim3((unsigned)cdiv(F, 4)), dim3(256), 0, s, a, st, gamma,
                       beta, reinterpret_cast<unsigned short*>(planes), F);
  });
}
template <int N>
static void stats(const float* a, float* st, int F, hipStream_t s) {
  hipLaunchKernelGGL(k_ln_stats_fast<N>, dim3((unsigned)cdiv(F, 4)), dim3(256), 0, s, a, st, F);
}

// column-block split of a conv GEMM: 1 for large batches, up to all blocks when F/TF workgroups
// would leave most of the 256 CUs idle (each extra split re-stages the input tile, which is cheap)
template <class C>
static int nsplit_for(int F) {
  constexpr int NBLK = cdiv(C::NT, C::NB);
  int wgs = cdiv(F, C::TF);
  int want = cdiv(512, wgs);
  return cmax(1, cmin_(NBLK, want));
}

static DenseArgs dense_args(const float* in, const float* Bp, float* out, int ldo, int F) {
  DenseArgs a;
  memset(&a, 0, sizeof a);
  a.in = in;
  a.Bp = Bp;
  a.out = out;
  a.ldo = ldo;
  a.F = F;
  return a;
}

static ConvArgs conv_args(const float* in, const float* st, const float* gamma, const float* beta, const float* Bp,
                          const float* bias, float* out, int F) {
  ConvArgs a;
  a.in = in;
  a.in2 = nullptr;
  a.idx = nullptr;
  a.st = st;
  a.gamma = gamma;
  a.beta = beta;
  a.Bp = Bp;
  a.bias = bias;
  a.out = out;
  a.F = F;
  return a;
}

// ---------------------------------------------------------------- forward
void encoder_fwd(const Model& m, const float* P, const float* x, int64_t F64, const Ws& w, hipStream_t s) {
  read_env();
  const int F = (int)F64;
  prep(m, P, w, F, s);
  // e0: Cin = 1, K = 7 -- 0.4 % of the MACs, HBM-bound: VALU conv fused with its LN statistics
  if (fwd_on(0)) {
    int fch = cmax(1, cdiv(F, 4096));
    if (enc0_wave(rt().fwd_mask, F)) {   // one wave per frame, no workgroup barriers
      VAENPVC_TIMED("enc0_fwd", s, hipLaunchKernelGGL(k_enc0_fwd_wave, dim3((unsigned)cmin_(cdiv(F, 4), 2048)), dim3(256), 0, s, x,
                                                      P + m.enc[0].w_off, P + m.enc[0].b_off, w.enc_a[0], w.enc_st[0], F));
    } else
    VAENPVC_TIMED("enc0_fwd", s, hipLaunchKernelGGL(k_enc0_fwd, dim3((unsigned)cdiv(F, fch)), dim3(256), 0, s, x, P + m.enc[0].w_off,
                                                    P + m.enc[0].b_off, w.enc_a[0], w.enc_st[0], F, fch));
  } else {
    generic::enc_layer_fwd(m, P, x, F, w, s, 0);
  }
  auto lnp = [&](int i) { return conv_args(w.enc_a[i - 1], w.enc_st[i - 1], P + m.enc[i - 1].gamma_off, P + m.enc[i - 1].beta_off,
                                           P + m.enc[i].w_off, P + m.enc[i].b_off, w.enc_a[i], F); };
  // a conv site as a GEMM over the overlapping-row view of channel-last planes (gfx950_viewconv.h): producer of the
  // activated input planes (LayerNorm + lrelu of layer i - 1), then the GEMM
  auto enc_view = [&](int site, int cl, int i, bool have, const char* tsplit, const char* tgemm) {
    for_dense_planes([&](auto npl) {
      constexpr int NPL = decltype(npl)::value;
      if (!have)
        VAENPVC_TIMED(tsplit, s, cv_split<NPL>(cl, w.enc_a[i - 1], w.enc_st[i - 1], P + m.enc[i - 1].gamma_off, P + m.enc[i - 1].beta_off,
                                               w.cl[cl], F, s));
      VAENPVC_TIMED(tgemm, s, cv_gemm<NPL>(site, w.scratch + Pk::cvw + cv_woff(site), w.cl[cl], w.enc_a[i], P + m.enc[i].b_off, F, s));
    });
  };
  // LayerNorm statistics of layer i's output; when the next layer is a view-GEMM site its activated input planes are
  // written in the same pass over the tensor
  auto enc_stats = [&](int i, int next_site, int cl, const char* tag) {
    const bool fuse = fwd_on(i + 1) && cv_fwd(next_site, F) && !fc_fwd(next_site, F) && !fcr_fwd(next_site, F);
    if (fuse)
      for_dense_planes([&](auto npl) {
        VAENPVC_TIMED(tag, s, cv_stats_split<decltype(npl)::value>(cl, w.enc_a[i], w.enc_st[i], P + m.enc[i].gamma_off, P + m.enc[i].beta_off,
                                                                   w.cl[cl], F, s));
      });
    return fuse;
  };
  // thin site on the fused kernel: fp32 input (+ LayerNorm + lrelu of layer `ln`) -> fp32 output, no planes in HBM
  auto fused = [&](int site, const float* src, const float* st, float* st_out, const ConvL* ln, const float* bias, float* out, const char* tag) {
    for_dense_planes([&](auto npl) {
      FcArgs fa{src, st, st_out, ln ? P + ln->gamma_off : nullptr, ln ? P + ln->beta_off : nullptr,
                reinterpret_cast<const unsigned short*>(w.scratch + Pk::cvw + cv_woff(site)), bias, out, F};
      VAENPVC_TIMED(tag, s, fconv<decltype(npl)::value>(site, fa, s));
    });
  };
  auto fused_r = [&](int site, const float* wpl, const float* src, const float* st, const ConvL* ln, const float* bias, float* out, const char* tag,
                     float* st_out = nullptr) {
    for_dense_planes([&](auto npl) {
      FcArgs fa{src, st, st_out, ln ? P + ln->gamma_off : nullptr, ln ? P + ln->beta_off : nullptr,
                reinterpret_cast<const unsigned short*>(wpl), bias, out, F};
      VAENPVC_TIMED(tag, s, fconv_r<decltype(npl)::value>(site, fa, s));
    });
  };
  bool have_y1 = false, have_y2 = false;
  const bool e2_takes_stats = VAENPVC_E2F_STATS && fwd_on(1) && fwd_on(2) && fcr_fwd(CV_E2F, F);
  if (fwd_on(1)) {
    if (fc_fwd(CV_E1F, F)) fused(CV_E1F, w.enc_a[0], w.enc_st[0], nullptr, &m.enc[0], P + m.enc[1].b_off, w.enc_a[1], "enc1_fwd");
    else if (cv_fwd(CV_E1F, F)) enc_view(CV_E1F, CL_Y0, 1, false, "enc1_split", "enc1_fwd");
    else VAENPVC_TIMED("enc1_fwd", s, launch_convgemm<E1F>(lnp(1), nsplit_for<E1F>(F), s));
    if (e2_takes_stats) {}   // (layer 2's fused kernel takes the statistics of its input in its staging)
    else if (!(have_y1 = enc_stats(1, CV_E2F, CL_Y1, "enc2_split"))) VAENPVC_TIMED("stats_enc1", s, stats<1824>(w.enc_a[1], w.enc_st[1], F, s));
  } else generic::enc_layer_fwd(m, P, x, F, w, s, 1);
  if (fwd_on(2)) {
    // (round 5) layer 2's fused kernel also leaves the statistics of its RESULT and the activated channel-last planes layer 3's view GEMM reads
    const bool e2_makes_y2 = rt().e2_osp && e2_takes_stats && dense_planes_now() <= 2 && fcr_otl(CV_E2F, 2) && fwd_on(3) && cv_fwd(CV_E3F, F) &&
                             !fc_fwd(CV_E3F, F) && !fcr_fwd(CV_E3F, F);
    if (fcr_fwd(CV_E2F, F) && e2_makes_y2) {
      for_dense_planes([&](auto npl) {
        FcArgs fa{w.enc_a[1], nullptr, w.enc_st[1], P + m.enc[1].gamma_off, P + m.enc[1].beta_off,
                  reinterpret_cast<const unsigned short*>(w.scratch + Pk::cvw + cv_woff(CV_E2F)), P + m.enc[2].b_off, w.enc_a[2], F};
        fa.st2_out = w.enc_st[2];
        fa.gamma2 = P + m.enc[2].gamma_off;
        fa.beta2 = P + m.enc[2].beta_off;
        fa.cl2_out = us(w.cl[CL_Y2]);
        fa.cl2_plane = cl_plane(CL_Y2, F);
        VAENPVC_TIMED("enc2_fwd", s, fconv_r<decltype(npl)::value>(CV_E2F, fa, s));
      });
      have_y2 = true;
    } else {
    if (fcr_fwd(CV_E2F, F)) fused_r(CV_E2F, w.scratch + Pk::cvw + cv_woff(CV_E2F), w.enc_a[1], e2_takes_stats ? nullptr : w.enc_st[1], &m.enc[1], P + m.enc[2].b_off, w.enc_a[2], "enc2_fwd",
                                    e2_takes_stats ? w.enc_st[1] : nullptr);
    else if (cv_fwd(CV_E2F, F)) enc_view(CV_E2F, CL_Y1, 2, have_y1, "enc2_split", "enc2_fwd");
    else VAENPVC_TIMED("enc2_fwd", s, launch_convgemm<E2F>(lnp(2), nsplit_for<E2F>(F), s));
    if (!(have_y2 = enc_stats(2, CV_E3F, CL_Y2, "enc3_split"))) VAENPVC_TIMED("stats_enc2", s, stats<1216>(w.enc_a[2], w.enc_st[2], F, s));
    }
  } else generic::enc_layer_fwd(m, P, x, F, w, s, 2);
  if (fwd_on(3) && cv_fwd(CV_E3F, F)) {
    // the frame-owning tile (k_cgemm_sf, round 5): conv + statistics + activated planes of a3 in one kernel
    bool e3_whole = false;
    if (rt().cg_sf)
      for_dense_planes([&](auto npl) {
        constexpr int NPL = decltype(npl)::value;
        if constexpr (NPL <= 2) {
          if (!have_y2)
            VAENPVC_TIMED("enc3_split", s, cv_split<NPL>(CL_Y2, w.enc_a[2], w.enc_st[2], P + m.enc[2].gamma_off, P + m.enc[2].beta_off, w.cl[CL_Y2], F, s));
          have_y2 = true;
          VAENPVC_TIMED("enc3_fwd", s, e3_whole = cv_gemm_stats_planes<NPL>(CV_E3F, w.scratch + Pk::cvw + cv_woff(CV_E3F), w.cl[CL_Y2], w.enc_a[3], P + m.enc[3].b_off,
                                                                            w.enc_st[3], P + m.enc[3].gamma_off, P + m.enc[3].beta_off,
                                                                            (fwd_on(4) && pg_fwd(F)) ? w.pl_y3 : nullptr, F, s));
        }
      });
    if (!e3_whole) {
      enc_view(CV_E3F, CL_Y2, 3, have_y2, "enc3_split", "enc3_fwd");
      VAENPVC_TIMED("stats_enc3", s, stats_planes<896, 7>(w.enc_a[3], w.enc_st[3], P + m.enc[3].gamma_off, P + m.enc[3].beta_off, w.pl_y3,
                           fwd_on(4) && pg_fwd(F), F, s));
    }
  } else if (fwd_on(3)) {
    VAENPVC_TIMED("enc3_fwd", s, (F < SMALL_BATCH_FRAMES ? launch_convgemm<E3Fs>(lnp(3), nsplit_for<E3Fs>(F), s) : launch_convgemm<E3F>(lnp(3), nsplit_for<E3F>(F), s)));
    VAENPVC_TIMED("stats_enc3", s, stats_planes<896, 7>(w.enc_a[3], w.enc_st[3], P + m.enc[3].gamma_off, P + m.enc[3].beta_off, w.pl_y3,
                         fwd_on(4) && pg_fwd(F), F, s));
  } else generic::enc_layer_fwd(m, P, x, F, w, s, 3);
  if (fwd_on(4) && pg_fwd(F)) {
    // layer 4 as the dense layer [F, 896] x [896, 768] on the bf16 matrix cores (gfx950_planegemm.h)
    for_dense_planes([&](auto npl) {
      constexpr int NPL = decltype(npl)::value;
      if (!fwd_on(3)) {  // (layer 3 ran on the generic kernel: its statistics pass did not write the planes)
        SplitArgs sa = split_args(w.enc_a[3], 896, 896, F, us(w.pl_y3));
        sa.st = w.enc_st[3];
        sa.gamma = P + m.enc[3].gamma_off;
        sa.beta = P + m.enc[3].beta_off;
        sa.lndiv = 7;
        VAENPVC_TIMED("enc4_split", s, launch_split<NPL>(sa, s));
      }
      NtArgs a = nt_args(w.pl_y3, F, 896, w.scratch + Pk::pg_enc4f, 768, 768, w.enc_a[4], 768);
      a.bias = w.scratch + Pk::pg_bias4;
      VAENPVC_TIMED("enc4_fwd", s, launch_gemm_nt<NPL>(a, s));
    });
    VAENPVC_TIMED("stats_enc4", s, stats_planes<768, 3>(w.enc_a[4], w.enc_st[4], P + m.enc[4].gamma_off, P + m.enc[4].beta_off, w.pl_y4,
                         fwd_on(5) && pg_fwd(F), F, s));
  } else if (fwd_on(4)) {
    VAENPVC_TIMED("enc4_fwd", s, launch_convgemm<E4F>(lnp(4), nsplit_for<E4F>(F), s));
    VAENPVC_TIMED("stats_enc4", s, stats_planes<768, 3>(w.enc_a[4], w.enc_st[4], P + m.enc[4].gamma_off, P + m.enc[4].beta_off, w.pl_y4,
                         fwd_on(5) && pg_fwd(F), F, s));
  } else generic::enc_layer_fwd(m, P, x, F, w, s, 4);
  if (fwd_on(5) && pg_fwd(F)) {
    for_dense_planes([&](auto npl) {
      constexpr int NPL = decltype(npl)::value;
      if (!fwd_on(4)) {  // (layer 4 ran on the generic kernel)
        SplitArgs sa = split_args(w.enc_a[4], 768, 768, F, us(w.pl_y4));
        sa.st = w.enc_st[4];
        sa.gamma = P + m.enc[4].gamma_off;
        sa.beta = P + m.enc[4].beta_off;
        sa.lndiv = 3;
        VAENPVC_TIMED("heads_split", s, launch_split<NPL>(sa, s));
      }
      NtArgs a = nt_args(w.pl_y4, F, 768, w.scratch + Pk::pg_headsf, 256, 256, w.z_mu, 128);
      a.C2 = w.z_lv;
      a.split = 128;
      a.bias = w.scratch + Pk::heads_bias;
      VAENPVC_TIMED("heads_fwd", s, launch_gemm_nt<NPL>(a, s));
    });
  } else if (fwd_on(5)) {
    DenseArgs a = dense_args(w.enc_a[4], w.scratch + Pk::heads_f, w.z_mu, 128, F);
    a.st = w.enc_st[4];
    a.gamma = P + m.enc[4].gamma_off;
    a.beta = P + m.enc[4].beta_off;
    a.out2 = w.z_lv;
    a.split = 128;
    a.bias = w.scratch + Pk::heads_bias;  // [b_mu | b_lv], packed by prep()
    if (F < SMALL_BATCH_FRAMES) {   // few frames: the three K chunks on three workgroups per tile (split-K into zeroed outputs)
      if (w.z_lv == w.z_mu + (size_t)F * 128) {   // (adjacent workspace regions: one fill)
        (void)hipMemsetAsync(w.z_mu, 0, (size_t)F * 256 * 4, s);
      } else {
        (void)hipMemsetAsync(w.z_mu, 0, (size_t)F * 128 * 4, s);
        (void)hipMemsetAsync(w.z_lv, 0, (size_t)F * 128 * 4, s);
      }
      VAENPVC_TIMED("heads_fwd", s, launch_densegemm<HeadsFs>(a, s, 6));
    } else
    VAENPVC_TIMED("heads_fwd", s, launch_densegemm<HeadsF>(a, s));
  } else generic::heads_fwd(m, P, F, w, s);
}

void decoder_fwd(const Model& m, const float* P, const float* z, const int64_t* y, int64_t F64, const Ws& w,
                 float* xh_out, hipStream_t s, bool weights_packed) {
  read_env();
  const int F = (int)F64;
  if (!weights_packed) prep(m, P, w, F, s);
  const bool z_planes_done = rt().plz_F == F && z == w.z;   // (this step's sampler kernel wrote them: reparam_fwd_planes)
  rt().plz_F = -1;
  if (fwd_on(6) && pg_fwd(F)) {
    for_dense_planes([&](auto npl) {
      constexpr int NPL = decltype(npl)::value;
      if (!z_planes_done) VAENPVC_TIMED("merge_split", s, launch_split<NPL>(split_args(z, 128, 128, F, us(w.pl_z)), s));
      NtArgs a = nt_args(w.pl_z, F, 128, w.scratch + Pk::pg_mergef, 1664, 1539, w.h, 1539);
      a.rowbias = w.scratch + Pk::merge_tb;   // + T[y_f]: the speaker's row of E Wy + (bz + by + b)
      a.idx = y;
      a.nrb = MERGE_NY;
      a.ldrb = 1539;
      VAENPVC_TIMED("merge_fwd", s, launch_gemm_nt<NPL>(a, s));
    });
  } else if (fwd_on(6)) {
    DenseArgs a = dense_args(z, w.scratch + Pk::merge_f, w.h, m.merge, F);
    a.idx = y;                                // + T[y_f]: the speaker's row of E Wy + (bz + by + b)
    a.rowbias = w.scratch + Pk::merge_tb;
    a.nrb = MERGE_NY;
    VAENPVC_TIMED("merge_fwd", s, (F < SMALL_BATCH_FRAMES ? launch_densegemm<MergeFs>(a, s) : launch_densegemm<MergeF>(a, s)));
  } else generic::merge_fwd(m, P, z, y, F, w, s);
  const bool abf = act_bf16(F);    // bf16 storage of dec_a[1], dec_a[2] (and of their gradients in the backward pass)
  auto fused = [&](int site, const float* src, const float* st, float* st_out, const ConvL* ln, const float* bias, float* out, const char* tag,
                   bool bf_in = false, bool bf_out = false) {
    for_dense_planes([&](auto npl) {
      FcArgs fa{src, st, st_out, ln ? P + ln->gamma_off : nullptr, ln ? P + ln->beta_off : nullptr,
                reinterpret_cast<const unsigned short*>(w.scratch + Pk::cvw + cv_woff(site)), bias, out, F};
      fa.bf_in = bf_in;
      fa.bf_out = bf_out;
      VAENPVC_TIMED(tag, s, fconv<decltype(npl)::value>(site, fa, s));
    });
  };
  auto dec_view = [&](int site, int cl, int i, bool have, const float* src, const char* tsplit, const char* tgemm) {
    for_dense_planes([&](auto npl) {
      constexpr int NPL = decltype(npl)::value;
      if (have) {}
      else if (i == 0) VAENPVC_TIMED(tsplit, s, cv_split<NPL>(cl, src, nullptr, nullptr, nullptr, w.cl[cl], F, s));
      else VAENPVC_TIMED(tsplit, s, cv_split<NPL>(cl, src, w.dec_st[i - 1], P + m.dec[i - 1].gamma_off, P + m.dec[i - 1].beta_off, w.cl[cl], F, s));
      VAENPVC_TIMED(tgemm, s, cv_gemm<NPL>(site, w.scratch + Pk::cvw + cv_woff(site), w.cl[cl], w.dec_a[i], P + m.dec[i].b_off, F, s));
    });
  };
  auto dec_stats = [&](int i, int next_site, int cl, const char* tag) {
    const bool fuse = fwd_on(8 + i) && cv_fwd(next_site, F) && !fc_fwd(next_site, F);
    if (fuse)
      for_dense_planes([&](auto npl) {
        VAENPVC_TIMED(tag, s, cv_stats_split<decltype(npl)::value>(cl, w.dec_a[i], w.dec_st[i], P + m.dec[i].gamma_off, P + m.dec[i].beta_off,
                                                                   w.cl[cl], F, s));
      });
    return fuse;
  };
  bool have_yd0 = false, have_yd1 = false;
  // a fused consumer takes the LayerNorm statistics of its input itself (it owns whole frames): no statistics pass
  const bool d1_fused = fwd_on(8) && fc_fwd(CV_D1F, F), d2_fused = fwd_on(9) && fc_fwd(CV_D2F, F);
  if (fwd_on(7) && fcr_fwd(CV_D0F, F)) {
    for_dense_planes([&](auto npl) {
      FcArgs fa{w.h, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<const unsigned short*>(w.scratch + Pk::cvwr_d0f),
                P + m.dec[0].b_off, w.dec_a[0], F};
      if (w.d_h && d0f_leaves_planes(F)) {   // train mode: the planes of h the weight-gradient GEMM of this layer reads leave the staging
        fa.cl_out = us(w.cl[CL_H]);
        fa.cl_plane = cl_plane(CL_H, F);
      }
      VAENPVC_TIMED("dec0_fwd", s, fconv_r<decltype(npl)::value>(CV_D0F, fa, s));
    });
    if (!d1_fused && !(have_yd0 = dec_stats(0, CV_D1F, CL_YD0, "dec1_split"))) VAENPVC_TIMED("stats_dec0", s, stats<1824>(w.dec_a[0], w.dec_st[0], F, s));
  } else if (fwd_on(7) && cv_fwd(CV_D0F, F)) {
    dec_view(CV_D0F, CL_H, 0, false, w.h, "dec0_split", "dec0_fwd");
    if (!d1_fused && !(have_yd0 = dec_stats(0, CV_D1F, CL_YD0, "dec1_split"))) VAENPVC_TIMED("stats_dec0", s, stats<1824>(w.dec_a[0], w.dec_st[0], F, s));
  } else if (fwd_on(7)) {
    VAENPVC_TIMED("dec0_fwd", s, (F < SMALL_BATCH_FRAMES ? launch_convgemm<D0Fs>(conv_args(w.h, nullptr, nullptr, nullptr, w.scratch + Pk::d0f,
                                                                P + m.dec[0].b_off, w.dec_a[0], F), nsplit_for<D0Fs>(F), s) : launch_convgemm<D0F>(conv_args(w.h, nullptr, nullptr, nullptr, w.scratch + Pk::d0f,
                                                                P + m.dec[0].b_off, w.dec_a[0], F), nsplit_for<D0F>(F), s)));
    if (!d1_fused && !(have_yd0 = dec_stats(0, CV_D1F, CL_YD0, "dec1_split"))) VAENPVC_TIMED("stats_dec0", s, stats<1824>(w.dec_a[0], w.dec_st[0], F, s));
  } else generic::dec_layer_fwd(m, P, F, w, xh_out, s, 0);
  if (fwd_on(8) && fc_fwd(CV_D1F, F)) {
    fused(CV_D1F, w.dec_a[0], nullptr, w.dec_st[0], &m.dec[0], P + m.dec[1].b_off, w.dec_a[1], "dec1_fwd", false, abf);
    if (!d2_fused && !(have_yd1 = dec_stats(1, CV_D2F, CL_YD1, "dec2_split"))) VAENPVC_TIMED("stats_dec1", s, stats<2736>(w.dec_a[1], w.dec_st[1], F, s));
  } else if (fwd_on(8) && cv_fwd(CV_D1F, F)) {
    dec_view(CV_D1F, CL_YD0, 1, have_yd0, w.dec_a[0], "dec1_split", "dec1_fwd");
    if (!d2_fused && !(have_yd1 = dec_stats(1, CV_D2F, CL_YD1, "dec2_split"))) VAENPVC_TIMED("stats_dec1", s, stats<2736>(w.dec_a[1], w.dec_st[1], F, s));
  } else if (fwd_on(8)) {
    VAENPVC_TIMED("dec1_fwd", s, launch_convgemm<D1F>(conv_args(w.dec_a[0], w.dec_st[0], P + m.dec[0].gamma_off,
                                                                P + m.dec[0].beta_off, w.scratch + Pk::d1f,
                                                                P + m.dec[1].b_off, w.dec_a[1], F), nsplit_for<D1F>(F), s));
    if (!d2_fused && !(have_yd1 = dec_stats(1, CV_D2F, CL_YD1, "dec2_split"))) VAENPVC_TIMED("stats_dec1", s, stats<2736>(w.dec_a[1], w.dec_st[1], F, s));
  } else generic::dec_layer_fwd(m, P, F, w, xh_out, s, 1);
  // (round 6) no pass between decoder layer 2 and the 1025-tap layer: statistics out of layer 2's epilogue (k_fconv<OST>), LayerNorm + lrelu +
  // operand split in the 1025-tap forward kernel's staging (k_toep_gemm_bf16<LNA>).  One channel group per frame tile (large batches).
  const bool d2_lna = rt().d2_lna && rt().planes <= 2 && !rt().d2_tail && fwd_on(9) && fwd_on(10) && fc_fwd(CV_D2F, F) && fc_occ3(CV_D2F) && toep_bf16_for(F) && !abf &&
                      (rt().d2_lna >= 2 || (F >= FCONV_MIN_FRAMES && toep_fwd_groups(F, weights_packed) == 1)) &&
                      (!weights_packed || (toep_wgrad_bf16_for(F) && w.dec_y && w.toep_yp));
  if (fwd_on(9)) {
    // (round 5) the decoder tail in layer 2's epilogue: statistics of its result, the 1025-tap layer's operand planes, bin 512 of the activated
    // tensor and output column 512 leave the kernel that computed the frames; k_ln_stats_act_planes and its re-read of the tensor are gone
    const bool d2_tail = rt().d2_tail && fc_fwd(CV_D2F, F) && fwd_on(10) && toep_bf16_for(F) && !abf && dense_planes_now() <= 2 &&
                         dense_planes_now() == rt().planes && toep_wgrad_bf16_for(F) && w.dec_y && w.toep_yp &&
                         fc_occ3(CV_D2F) && F >= FCONV_MIN_FRAMES;
    if (d2_lna) {
      // (round 6) layer 2's kernel leaves the statistics of its result; the 1025-tap forward kernel normalises the fp32 tensor in its staging
      for_dense_planes([&](auto npl) {
        FcArgs fa{w.dec_a[1], nullptr, w.dec_st[1], P + m.dec[1].gamma_off, P + m.dec[1].beta_off,
                  reinterpret_cast<const unsigned short*>(w.scratch + Pk::cvw + cv_woff(CV_D2F)), P + m.dec[2].b_off, w.dec_a[2], F};
        fa.st2_out = w.dec_st[2];
        VAENPVC_TIMED("dec2_fwd", s, fconv<decltype(npl)::value>(CV_D2F, fa, s));
      });
    } else
    if (d2_tail) {
      for_dense_planes([&](auto npl) {
        FcArgs fa{w.dec_a[1], nullptr, w.dec_st[1], P + m.dec[1].gamma_off, P + m.dec[1].beta_off,
                  reinterpret_cast<const unsigned short*>(w.scratch + Pk::cvw + cv_woff(CV_D2F)), P + m.dec[2].b_off, w.dec_a[2], F};
        fa.st2_out = w.dec_st[2];
        fa.gamma2 = P + m.dec[2].gamma_off;
        fa.beta2 = P + m.dec[2].beta_off;
        fa.yp = reinterpret_cast<unsigned short*>(w.toep_yp);
        fa.decy = w.dec_y;
        fa.wc = w.scratch + Pk::wc;
        fa.bias3 = P + m.dec[3].b_off;
        fa.xh = xh_out;
        fa.zero_xh = toep_fwd_groups(F, weights_packed) > 1 ? 1 : 0;
        VAENPVC_TIMED("dec2_fwd", s, fconv<decltype(npl)::value>(CV_D2F, fa, s));
      });
    } else
    if (fc_fwd(CV_D2F, F)) fused(CV_D2F, w.dec_a[1], nullptr, w.dec_st[1], &m.dec[1], P + m.dec[2].b_off, w.dec_a[2], "dec2_fwd", abf, abf);
    else if (cv_fwd(CV_D2F, F)) dec_view(CV_D2F, CL_YD1, 2, have_yd1, w.dec_a[1], "dec2_split", "dec2_fwd");
    else
    VAENPVC_TIMED("dec2_fwd", s, launch_convgemm<D2F>(conv_args(w.dec_a[1], w.dec_st[1], P + m.dec[1].gamma_off,
                                                                P + m.dec[1].beta_off, w.scratch + Pk::d2f,
                                                                P + m.dec[2].b_off, w.dec_a[2], F), nsplit_for<D2F>(F), s));
    if (d2_tail || d2_lna) {}
    else if (toep_bf16_for(F) && fwd_on(10))
      for_planes([&](auto npl) {
        constexpr int NPL_ = decltype(npl)::value;
        auto launch_sap = [&](auto kern) {
          VAENPVC_TIMED("dec2_stats_planes", s, hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(F, 4)), dim3(256), 0, s, w.dec_a[2], w.dec_st[2],
                           P + m.dec[2].gamma_off, P + m.dec[2].beta_off, w.dec_y, reinterpret_cast<unsigned short*>(w.toep_yp),
                           w.scratch + Pk::wc, P + m.dec[3].b_off, xh_out, (int)F, toep_wgrad_bf16_for(F) ? 0 : 1,
                           toep_fwd_groups(F, weights_packed) > 1 ? 1 : 0));
        };
        if constexpr (NPL_ == 1) {
          if (abf) { launch_sap(k_ln_stats_act_planes<1, true>); return; }
        }
        launch_sap(k_ln_stats_act_planes<NPL_, false>);
      });
    else
      VAENPVC_TIMED("dec2_stats_planes", s, hipLaunchKernelGGL((k_ln_stats_act<4104, 513>), dim3((unsigned)cdiv(F, 4)), dim3(256), 0, s, w.dec_a[2], w.dec_st[2],
                         P + m.dec[2].gamma_off, P + m.dec[2].beta_off, w.dec_y, F));
  } else {
    generic::dec_layer_fwd(m, P, F, w, xh_out, s, 2);
    int64_t tot = (int64_t)F * 4104;
    VAENPVC_TIMED("dec2_stats_planes", s, hipLaunchKernelGGL(k_act_from_stats, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, w.dec_a[2], w.dec_st[2],
                       P + m.dec[2].gamma_off, P + m.dec[2].beta_off, w.dec_y, tot, 4104, 513));
  }
  if (fwd_on(10)) {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_toep_fwd<4>), TF_LDS);
    rt().ensure_lds(reinterpret_cast<const void*>(&k_toep_fwd<1>), TF_LDS);
    if (toep_bf16_for(F) && fwd_on(9)) {  // (the planes come from the tuned layer-9 epilogue kernel)
      for_planes([&](auto npl) {
        constexpr int NPL = decltype(npl)::value;
        rt().ensure_lds(reinterpret_cast<const void*>(&k_toep_gemm_bf16<true, NPL>), dg_lds(NPL));
        if constexpr (NPL <= 2) if (d2_lna) {   // LayerNorm + lrelu + split of decoder layer 2's fp32 output in the staging (no plane producer pass in front)
          constexpr int LDS_LNA = dg_lds(NPL) + NPL * DG_APL + DG_LNA_LDS;   // two A tiles: 151 104 bytes with two planes (three would not fit)
          ToepLna ln;
          ln.a = w.dec_a[2];
          ln.st = w.dec_st[2];
          ln.gamma = P + m.dec[2].gamma_off;
          ln.beta = P + m.dec[2].beta_off;
          ln.wc = w.scratch + Pk::wc;
          ln.yp = weights_packed ? reinterpret_cast<unsigned short*>(w.toep_yp) : nullptr;   // (a train step: the weight gradient reads the planes)
          ln.decy = weights_packed ? w.dec_y : nullptr;
          rt().ensure_lds(reinterpret_cast<const void*>(&k_toep_gemm_bf16<true, NPL, false, TB_H, true>), LDS_LNA);
          VAENPVC_TIMED("dec3_fwd", s, hipLaunchKernelGGL((k_toep_gemm_bf16<true, NPL, false, TB_H, true>), dim3((unsigned)cdiv(F, DG_M), 1u), dim3(256),
                                                          LDS_LNA, s, nullptr, reinterpret_cast<const unsigned short*>(w.scratch + Pk::wfw),
                                                          P + m.dec[3].b_off, xh_out, (int)F, ln));
          return;
        }
        VAENPVC_TIMED("dec3_fwd", s, hipLaunchKernelGGL((k_toep_gemm_bf16<true, NPL>), dim3((unsigned)cdiv(F, DG_M), (unsigned)toep_fwd_groups(F, weights_packed)), dim3(256), dg_lds(NPL), s,
                                                        reinterpret_cast<const unsigned short*>(w.toep_yp),
                                                        reinterpret_cast<const unsigned short*>(w.scratch + Pk::wfw),
                                                        P + m.dec[3].b_off, xh_out, (int)F, ToepLna{}));
      });
    } else
    if (F >= 2048) {
      VAENPVC_TIMED("dec3_fwd", s, hipLaunchKernelGGL(k_toep_fwd<4>, dim3((unsigned)cdiv(F, 32), 1), dim3(256), TF_LDS, s, w.dec_y,
                                                      w.scratch + Pk::wc, P + m.dec[3].b_off, xh_out, F));
    } else {
      VAENPVC_TIMED("dec3_fwd", s, hipLaunchKernelGGL(k_toep_fwd<1>, dim3((unsigned)cdiv(F, 32), 4), dim3(256), TF_LDS, s, w.dec_y,
                                                      w.scratch + Pk::wc, P + m.dec[3].b_off, xh_out, F));
    }
    // (column p = 512 is produced inside k_toep_fwd by the workgroups with blockIdx.y == 0)
  } else generic::dec_layer_fwd(m, P, F, w, xh_out, s, 3);
}

// ---------------------------------------------------------------- backward
static TnArgs tn_args(const float* X, int ldx, const float* Y, int ldy, int M, int N, int F, float* C, int ldc) {
  TnArgs a;
  a.X = X;
  a.xidx = nullptr;
  a.st = a.gamma = a.beta = nullptr;
  a.lndiv = 1;
  a.ldx = ldx;
  a.Y = Y;
  a.ldy = ldy;
  a.M = M;
  a.N = N;
  a.F = F;
  a.C = C;
  a.ldc = ldc;
  a.fchunk = 0;
  return a;
}
// ---- side stream for the weight-gradient kernels --------------------------------------
// Every weight gradient depends only on activations of the forward pass and on ONE gradient
// tensor of the serial chain dgrad_i -> LN-backward_{i-1} -> dgrad_{i-1} ...; none of them is
// on that chain.  They are forked onto a second stream (event fork at the point where their
// gradient tensor is complete, one join at the end) so that they fill the MFMA pipes and the
// kernel tails the chain leaves idle.  VAENPVC_SIDE_STREAM=0 disables the fork.
// The stream and its event ring belong to the context (Runtime::side_stream, created lazily on the context's
// device, destroyed with it).
constexpr int SIDE_STREAM_MAX_FRAMES = 16384;
static int kchunks_for(int F, int tiles) { return cmax(1, cmin_(cdiv(F, 64), 512 / tiles)); }  // 2 workgroups (64 KB LDS) per CU

#ifndef VAENPVC_Z_PLANES
#define VAENPVC_Z_PLANES 1
#endif
bool reparam_fwd_planes(const Model& m, const float* eps, const PhiloxKey* key, int64_t F64, const Ws& w, hipStream_t s) {
  read_env();
  const int F = (int)F64;
  if (!VAENPVC_Z_PLANES || m.z != 128 || F < 1024 || !w.pl_z || !fwd_on(6) || !pg_fwd(F)) return false;
  const PhiloxKey k = key ? *key : PhiloxKey{0, 0, 0, 0, nullptr};
  for_dense_planes([&](auto npl) {
    hipLaunchKernelGGL((k_reparam_planes<decltype(npl)::value>), dim3((unsigned)cmin_(2048, cdiv(F, 4))), dim3(256), 0, s, w.z_mu, w.z_lv, eps, w.z, w.kl_f,
                       us(w.pl_z), F, k, key ? 1 : 0, w.eps);
  });
  rt().plz_F = F;
  return true;
}
#ifndef VAENPVC_NLL_POST
#define VAENPVC_NLL_POST 1
#endif
static inline int nll_post_blocks(int64_t F) { return cmin_(1024, cdiv((int)F, 4)); }   // (each leaves a 16 KB part of the last layer's edge term)
// floats per row of d(activated output of decoder layer 2) in dy_tmp: padded to 16 bytes when its producer is the bf16 Toeplitz input-gradient
// GEMM and its only consumer the fused backward kernel of decoder layer 2 (VAENPVC_DY2_PAD=0: the tensor's own 513-float rows, A/B)
static inline int dy2_pitch(int64_t F) {
  static_assert(FB_DY2_PITCH == DY2_PITCH, "one constant");
  return (rt().dy2_pad && toep_bf16_for(F) && bwd_on(9) && bwd_on(10) && fb_bwd(FB_D2, F) && !act_bf16(F) && !frame_bwd_on(F)) ? DY2_PITCH : TB_H;
}
bool loss_fwd_post(const Model& m, const float* P, const float* x, int64_t F64, const Ws& w, float* loss3, hipStream_t s) {
  const int F = (int)F64;
  if (!VAENPVC_NLL_POST || !w.d_xh || !w.toep_gp || !w.dy_tmp || !w.dec_y || !w.d_dec_a[0] || !fwd_on(9) || !fwd_on(10) || F < 1024 || frame_bwd_on(F64) || !bwd_on(10) || !toep_bf16_for(F) || act_bf16(F)) return false;
  for_planes([&](auto npl) {
    constexpr int NPL = decltype(npl)::value;
    // (d(xh) as fp32 has no reader when the weight-gradient GEMM takes the planes too: 67 MB per step not written)
    hipLaunchKernelGGL((k_nll_dxh_post<NPL>), dim3((unsigned)nll_post_blocks(F)), dim3(256), 0, s, x, w.xh, w.nll_f,
                       rt().dxh_skip && toep_wgrad_bf16_for(F) ? nullptr : w.d_xh, P + m.dec[3].w_off,
                       reinterpret_cast<unsigned short*>(w.toep_gp), w.dy_tmp, w.scratch + Pk::lnpart, F, 1.0f / (float)F, w.dec_y,
                       w.d_dec_a[0],    // (the edge-term parts wait in d(a0)'s buffer: nothing writes it before the backward pass has added them)
                       dy2_pitch(F));
  });
  generic::loss_reduce(F, w, loss3, s);
  rt().dxh_post_F = F;
  return true;
}

void backward(const Model& m, const float* P, const float* x, const int64_t* y, const float* eps, int64_t F64,
              const Ws& w, float* G, hipStream_t s) {
  read_env();
  const int F = (int)F64;
  const bool abf = act_bf16(F64);   // bf16 storage of dec_a[1], dec_a[2] and of the gradients at their activated outputs
  (void)hipMemsetAsync(G, 0, (size_t)m.n_params * 4, s);
  // (bit 30 of the backward mask cleared = no fork for this call: serialised kernels, used by bench.py to
  //  time single kernels without concurrent neighbours)
  // (with two operand planes the fork LOSES from SIDE_STREAM_MAX_FRAMES frames on: the weight-gradient GEMMs hold a whole CU's LDS for
  //  100+ us each and the chain's fused kernels (78 - 150 KB of LDS per workgroup) cannot move in beside them.  Measured same-box, step
  //  with / without the second stream: 1 024 frames 1.07 / 1.16 ms, 4 096: 1.68 / 1.72, 8 192: 2.39 / 2.41, 16 384: 3.69 / 3.69,
  //  32 768: 6.79 / 6.59; with one plane (bf16 mode) 4.74 / 4.86 and with three 10.37 / 10.81 at 32 768: the fork stays there)
  hipStream_t side = bwd_on(30) && (F < SIDE_STREAM_MAX_FRAMES || dense_planes_now() != 2 || rt().side_forced) ? rt().side_stream() : nullptr;
  const bool fork = side != nullptr;
  hipStream_t s2 = fork ? side : s;   // weight-gradient stream
  auto ready = [&]() { if (fork) rt().stream_dep(s, s2); };   // "the tensors produced so far on s are ready for s2"
  // gradient range [off, end) of the flat buffer is complete once everything enqueued so far has run: hand it to
  // the data-parallel host (vaenpvc_set_bucket_callback) on the weight-gradient stream, which then holds all of it
  auto bucket = [&](int64_t off, int64_t end) {
    Runtime& r = rt();
    if (!r.bucket_cb) return;
    ready();
    r.bucket_cb(r.bucket_user, r.bucket_next++, off, end - off, (void*)s2);
  };
  // one pass over d(xh) for the bf16 kernels of the last layer: its three planes (operand of the input-gradient
  // and weight-gradient GEMMs), column 512 of the input gradient, the bias gradient
  const bool toep_planes = bwd_on(10) && toep_bf16_for(F);
  const bool post_done = toep_planes && rt().dxh_post_F == F;   // (the loss kernel of this step did it: loss_fwd_post)
  rt().dxh_post_F = -1;
  if (post_done)   // bias parts and the parts of the weight gradient's edge term (row 512 of y) -> the gradient
    VAENPVC_TIMED("dxh_post", s, hipLaunchKernelGGL(k_colsum_part, dim3(1), dim3(256), 0, s, w.scratch + Pk::lnpart, nll_post_blocks(F), 1, G + m.dec[3].b_off);
                  hipLaunchKernelGGL(k_sum_parts_add, dim3((unsigned)cdiv(513 * 8, 64)), dim3(1024), 0, s, w.d_dec_a[0], nll_post_blocks(F), 513 * 8, G + m.dec[3].w_off));
  else if (toep_planes)
    for_planes([&](auto npl) {
      constexpr int NPL_ = decltype(npl)::value;
      auto launch_dp = [&](auto kern) {
        VAENPVC_TIMED("dxh_post", s, hipLaunchKernelGGL(kern, dim3((unsigned)cmin_(2048, cdiv((int)F, 4))), dim3(256), 0, s, w.d_xh, P + m.dec[3].w_off,
                           reinterpret_cast<unsigned short*>(w.toep_gp), w.dy_tmp, G + m.dec[3].b_off, (int)F, dy2_pitch(F)));
      };
      if constexpr (NPL_ == 1) {
        if (abf) { launch_dp(k_dxh_post<1, true>); return; }
      }
      launch_dp(k_dxh_post<NPL_, false>);
    });
  ready();
  bool dec_bias_done[4] = {false, false, false, false};
  bool enc_bias_done[5] = {false, false, false, false, false};
  const int WGS = 512;    // workgroups of the chunked weight-gradient reductions: 2 per CU (LDS-bound residency);
                          // more chunks only deepen the same-address atomic chains on the small weight tensors
  const int LWGS = 2048;  // ... of the HBM-bound LayerNorm backward
  // second stages of the LayerNorm backward kernels: queued and flushed once per gradient bucket at small batches (the
  // partial rows of all queued layers must fit the scratch region: <= 2048 * 3 * 256 floats; large batches need it per layer)
  LnReduceList lnq_store;
  lnq_store.n = 0;
  lnq_store.used = 0;
  lnq_store.capacity = (int64_t)2048 * 3 * 256;
  // (only when every step runs its tuned kernel: a generic fallback STORES the bias gradient its layer's LayerNorm
  //  backward has added to, which is only right if that addition happened first)
  LnReduceList* lnq = (F <= 512 && (rt().bwd_mask & 0x7ffu) == 0x7ffu) ? &lnq_store : nullptr;
  // encoder layer 0: one fused backward kernel (gfx950_elem.h: k_enc0_bwd_wave) from ENC0_WAVE_MIN_FRAMES frames on;
  // bit 19 of the backward mask cleared = the separate LayerNorm-backward and weight-gradient passes
  const bool enc0_fused = bwd_on(0) && bwd_on(1) && bwd_on(19) && F >= ENC0_WAVE_MIN_FRAMES && !lnq;

  // ---- conv layers on the view GEMMs (gfx950_viewconv.h): producers / consumers of the channel-last planes
  auto gsplit = [&](int cl, const float* src, const char* tag) {   // gradient tensor -> planes
    for_dense_planes([&](auto npl) {
      VAENPVC_TIMED(tag, s, cv_split<decltype(npl)::value>(cl, src, nullptr, nullptr, nullptr, w.cl[cl], F, s));
    });
  };
  auto asplit = [&](int cl, const float* src, const float* st, const ConvL* ln, const char* tag) {   // activation the forward pass did not leave
    for_dense_planes([&](auto npl) {
      VAENPVC_TIMED(tag, s, cv_split<decltype(npl)::value>(cl, src, st, ln ? P + ln->gamma_off : nullptr, ln ? P + ln->beta_off : nullptr,
                                                           w.cl[cl], F, s));
    });
  };
  auto vwgrad = [&](int wsite, float* dW, const char* tag) {
    for_dense_planes([&](auto npl) {
      VAENPVC_TIMED(tag, s2, cv_wgrad<decltype(npl)::value>(wsite, w.cl[CWS[wsite].a], w.cl[CWS[wsite].b], dW, F, 512, s2));
    });
  };
  auto fdgrad = [&](int site, const float* grad, float* out, const char* tag) {   // thin site on the fused kernel
    for_dense_planes([&](auto npl) {
      FcArgs fa{grad, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<const unsigned short*>(w.scratch + Pk::cvw + cv_woff(site)), nullptr, out, F};
      VAENPVC_TIMED(tag, s, fconv<decltype(npl)::value>(site, fa, s));
    });
  };
  // thin weight gradient on the fused kernel: A = plain operand (gradient or activation), B = view operand
  auto fwg = [&](int wsite, const float* asrc, const float* ast, const ConvL* aln, const float* bsrc, const float* bst, const ConvL* bln,
                 float* dW, const char* tag) {
    for_dense_planes([&](auto npl) {
      constexpr int NPL = decltype(npl)::value;
      if constexpr (NPL <= 2) {
        FwArgs fa{asrc, ast, aln ? P + aln->gamma_off : nullptr, aln ? P + aln->beta_off : nullptr,
                  bsrc, bst, bln ? P + bln->gamma_off : nullptr, bln ? P + bln->beta_off : nullptr, dW, F};
        VAENPVC_TIMED(tag, s2, fwgrad<NPL>(wsite, fa, s2));
      }
    });
  };
  auto rdgrad = [&](int site, const float* wpl, const float* grad, float* out, const char* tag, const float* planes_in = nullptr,
                    float* planes_out = nullptr, int out_kp = 0) {   // medium site, register-weight fused kernel
    for_dense_planes([&](auto npl) {
      FcArgs fa{grad, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<const unsigned short*>(wpl), nullptr, out, F};
      if (planes_in) {
        fa.cl_in = reinterpret_cast<const unsigned short*>(planes_in);
        fa.cl_plane = cl_plane(CVS[site].x, F);
      }
      if (planes_out) {   // (the result as the operand planes [NPL][F][out_kp] of the GEMMs behind it, no fp32 copy)
        fa.pl_out = us(planes_out);
        fa.pl_plane = (int64_t)F * out_kp;
        fa.pl_kp = out_kp;
      }
      VAENPVC_TIMED(tag, s, fconv_r<decltype(npl)::value>(site, fa, s));
    });
  };
  auto vdgrad = [&](int site, float* out, const char* tag) {
    for_dense_planes([&](auto npl) {
      VAENPVC_TIMED(tag, s, cv_gemm<decltype(npl)::value>(site, w.scratch + Pk::cvw + cv_woff(site), w.cl[CVS[site].x], out, nullptr, F, s));
    });
  };

  // ---- d3: the 1025-tap layer
  if (bwd_on(10)) {
    const ConvL& l2 = m.dec[2];
    if (toep_wgrad_bf16_for(F)) {
      // bf16 planes of both operands exist (forward producer, k_dxh_post above)
      unsigned short* gp = reinterpret_cast<unsigned short*>(w.toep_gp);
      const int zc = (int)cmax(1, cmin_(rt().toep_zc, cdiv(F, 1024)));  // 8 x 8 x zc workgroups (more chunks at small batches measured slower: more epilogues)
      const int fch = rup(cdiv((int)F, zc), WG_KF);
      for_planes([&](auto npl) {
        constexpr int NPL = decltype(npl)::value;
        if constexpr (NPL <= 2) {
          if (rt().toep_wgrad_w4 && bwd_on(17) && F >= 4096) {   // 128 x 128 wave tiles, operands by LDS-DMA: 8 x 4 x zc4 workgroups
            const int zc4 = (int)cmax(1, cmin_(2 * rt().toep_zc, cdiv(F, 512)));
            const int fch4 = rup(cdiv((int)F, zc4), W4_KF);
            rt().ensure_lds(reinterpret_cast<const void*>(&k_toep_wgrad_bf16_w4<NPL>), w4_lds(NPL));
            VAENPVC_TIMED("dec3_wgrad", s2, hipLaunchKernelGGL(k_toep_wgrad_bf16_w4<NPL>, dim3(8 * 4 * (unsigned)cdiv((int)F, fch4)), dim3(256),
                                                              w4_lds(NPL), s2, reinterpret_cast<const unsigned short*>(w.toep_yp), gp,
                                                              G + m.dec[3].w_off, (int)F, fch4));
            return;
          }
          if (!rt().toep_wgrad_k16) {   // 32-frame chunks: two k-steps per barrier
            const int fch32 = rup(cdiv((int)F, zc), W2_KF);
            rt().ensure_lds(reinterpret_cast<const void*>(&k_toep_wgrad_bf16_k32<NPL>), w2_lds(NPL));
            VAENPVC_TIMED("dec3_wgrad", s2, hipLaunchKernelGGL(k_toep_wgrad_bf16_k32<NPL>, dim3(8 * TB_C * (unsigned)cdiv((int)F, fch32)), dim3(512),
                                                              w2_lds(NPL), s2, reinterpret_cast<const unsigned short*>(w.toep_yp), gp,
                                                              G + m.dec[3].w_off, (int)F, fch32));
            return;
          }
        }
        rt().ensure_lds(reinterpret_cast<const void*>(&k_toep_wgrad_bf16<NPL>), wg_lds(NPL));
        VAENPVC_TIMED("dec3_wgrad", s2, hipLaunchKernelGGL(k_toep_wgrad_bf16<NPL>, dim3(8, TB_C, (unsigned)cdiv((int)F, fch)), dim3(512), wg_lds(NPL), s2,
                                                          reinterpret_cast<const unsigned short*>(w.toep_yp), gp, G + m.dec[3].w_off, (int)F, fch));
      });
    } else {
      TnArgs a = tn_args(w.dec_y, 4104, w.d_xh, 513, 4096, 512, F, G + m.dec[3].w_off, 0);
      VAENPVC_TIMED("dec3_wgrad", s2, launch_tngemm(a, true, kchunks_for(F, 32 * 4), s2));
    }
    int ech = cmax(1, cmin_(cdiv(F, 64), 128));
    int efc = rup(cdiv(F, ech), 64);
    if (!post_done)   // (else: the loss kernel left the edge term as parts, added with the bias parts above)
    VAENPVC_TIMED("dec3_row512", s2, hipLaunchKernelGGL(k_toep_wgrad_row512, dim3((unsigned)cdiv(F, efc), 3), dim3(256), 0, s2, w.dec_y, w.d_xh,
                                                       G + m.dec[3].w_off, F, efc));
    if (!toep_planes)  // (k_dxh_post computed it)
      VAENPVC_TIMED("dec3_bias", s2, hipLaunchKernelGGL(k_sum_all_atomic, dim3((unsigned)cmin_(1024, cdiv(F * 513, 1024))), dim3(256), 0, s2, w.d_xh,
                         (int64_t)F * 513, G + m.dec[3].b_off));
    rt().ensure_lds(reinterpret_cast<const void*>(&k_toep_dgrad<8, 8>), TD_LDS);
    rt().ensure_lds(reinterpret_cast<const void*>(&k_toep_dgrad<1, 4>), TD_LDS);
    if (toep_bf16_for(F)) {
      unsigned short* gp = reinterpret_cast<unsigned short*>(w.toep_gp);
      for_planes([&](auto npl) {
        constexpr int NPL = decltype(npl)::value;
        auto launch_dg = [&](auto kern) {
          rt().ensure_lds(reinterpret_cast<const void*>(kern), dg_lds(NPL));
          VAENPVC_TIMED("dec3_dgrad", s, hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(F, DG_M), (unsigned)toep_groups(F)), dim3(256), dg_lds(NPL), s, gp,
                                                            reinterpret_cast<const unsigned short*>(w.scratch + Pk::wdg), (const float*)nullptr,
                                                            w.dy_tmp, (int)F, ToepLna{}));  // (column 512: k_dxh_post)
        };
        if constexpr (NPL == 1) {
          if (abf) { launch_dg(&k_toep_gemm_bf16<false, 1, true>); return; }
        }
        if (dy2_pitch(F) == DY2_PITCH) launch_dg(&k_toep_gemm_bf16<false, NPL, false, DY2_PITCH>);
        else launch_dg(&k_toep_gemm_bf16<false, NPL, false>);
      });
    } else
    if (F >= 8192) {
      VAENPVC_TIMED("dec3_dgrad", s, hipLaunchKernelGGL((k_toep_dgrad<8, 8>), dim3((unsigned)cdiv(F, 32), 1), dim3(512), TD_LDS, s, w.d_xh,
                                                        w.scratch + Pk::wc, w.dy_tmp, F));
    } else {
      VAENPVC_TIMED("dec3_dgrad", s, hipLaunchKernelGGL((k_toep_dgrad<1, 4>), dim3((unsigned)cdiv(F, 32), 8), dim3(256), TD_LDS, s, w.d_xh,
                                                        w.scratch + Pk::wc, w.dy_tmp, F));
    }
    // (with the whole backward step of layer 2 in one kernel, gfx950_fbwd.h, its LayerNorm backward runs there)
    if (!(bwd_on(9) && fb_bwd(FB_D2, F))) {
      VAENPVC_TIMED("lnb_dec2", s, launch_ln_bwd<LnbCfg<8, 513>>(w.dy_tmp, w.dec_a[2], w.dec_st[2], P + l2.gamma_off, P + l2.beta_off, w.d_dec_a[2],
                                       G + l2.gamma_off, G + l2.beta_off, G + l2.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
      dec_bias_done[2] = true;
    }
  } else generic::bwd_dec_layer(m, P, F, w, G, s, 3);
  // one kernel per thin decoder layer: LayerNorm backward + input gradient + weight gradient + the layer's parameter sums
  auto fused_bwd = [&](int layer, int i, const float* dy, float* dx, const char* tag, unsigned short* pl0 = nullptr, int64_t pl0_plane = 0,
                       float* part0 = nullptr) {
    const bool enc = fb_enc(layer);
    const ConvL &l = enc ? m.enc[i] : m.dec[i], &pl = enc ? m.enc[i - 1] : m.dec[i - 1];
    float* const* act = enc ? w.enc_a : w.dec_a;
    float* const* sts = enc ? w.enc_st : w.dec_st;
    for_dense_planes([&](auto npl) {
      constexpr int NPL = decltype(npl)::value;
      FbArgs fa{dy, act[i], sts[i], P + l.gamma_off, P + l.beta_off, act[i - 1], sts[i - 1], P + pl.gamma_off,
                P + pl.beta_off, reinterpret_cast<const unsigned short*>(w.scratch + Pk::cvw + cv_woff(fb_gsite(layer))), dx,
                G + l.w_off, G + l.gamma_off, G + l.beta_off, G + l.b_off, F};
      fa.bf16_act = abf && !enc;
      fa.dy_pitch = (layer == FB_D2 && dy2_pitch(F) == DY2_PITCH) ? DY2_PITCH : 0;
      fa.pl0 = pl0;
      fa.pl0_plane = pl0_plane;
      fa.part0 = part0;
      VAENPVC_TIMED(tag, s, fbwd<NPL>(layer, fa, s));
    });
  };
  // where the gradient at the activated output of the layer being processed lives (the fused kernels ping-pong between
  // dy_tmp and the buffer of the pre-LN gradient they no longer write)
  const float* dy_cur = w.dy_tmp;

  // ---- d2
  if (bwd_on(9) && bwd_on(10) && fb_bwd(FB_D2, F)) {
    const ConvL& pl = m.dec[1];
    fused_bwd(FB_D2, 2, w.dy_tmp, w.d_dec_a[2], "dec2_bwd");
    dy_cur = w.d_dec_a[2];
    if (!(bwd_on(8) && fb_bwd(FB_D1, F))) {
      VAENPVC_TIMED("lnb_dec1", s, launch_ln_bwd<LnbCfg<16, 171>>(dy_cur, w.dec_a[1], w.dec_st[1], P + pl.gamma_off, P + pl.beta_off, w.d_dec_a[1],
                                        G + pl.gamma_off, G + pl.beta_off, G + pl.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
      dec_bias_done[1] = true;
    }
  } else if (bwd_on(9)) {
    const ConvL &l = m.dec[2], &pl = m.dec[1];
    WgArgs a{w.d_dec_a[2], nullptr, nullptr, nullptr, w.dec_a[1], w.dec_st[1], P + pl.gamma_off, P + pl.beta_off,
             G + l.w_off, F, 0};
    const bool fg = fc_bwd(CV_D2G, F), vg = !fg && cv_bwd(CV_D2G, F), fw = fw_bwd(CW_D2, F), vw = !fw && cw_bwd(CW_D2, F);
    if (vg || vw) gsplit(CL_GD2, w.d_dec_a[2], "dec2_gsplit");
    if (vw && !fwd_planes(9, CV_D2F, F)) asplit(CL_YD1, w.dec_a[1], w.dec_st[1], &pl, "dec2_asplit");
    ready();
    if (fw) fwg(CW_D2, w.dec_a[1], w.dec_st[1], &pl, w.d_dec_a[2], nullptr, nullptr, G + l.w_off, "dec2_wgrad");
    else if (vw) vwgrad(CW_D2, G + l.w_off, "dec2_wgrad");
    else VAENPVC_TIMED("dec2_wgrad", s2, launch_convwgrad<WD2>(a, WGS, s2));
    if (!dec_bias_done[2]) generic::bias_grad(w.d_dec_a[2], G + l.b_off, F, l.cout, l.hout, s);
    if (fg) fdgrad(CV_D2G, w.d_dec_a[2], w.dy_tmp, "dec2_dgrad");
    else if (vg) vdgrad(CV_D2G, w.dy_tmp, "dec2_dgrad");
    else
    VAENPVC_TIMED("dec2_dgrad", s, launch_convgemm<GD2>(conv_args(w.d_dec_a[2], nullptr, nullptr, nullptr, w.scratch + Pk::gd2,
                                                                  nullptr, w.dy_tmp, F), nsplit_for<GD2>(F), s));
    if (!(bwd_on(8) && fb_bwd(FB_D1, F))) {   // (layer 1's fused backward kernel does it otherwise)
      VAENPVC_TIMED("lnb_dec1", s, launch_ln_bwd<LnbCfg<16, 171>>(w.dy_tmp, w.dec_a[1], w.dec_st[1], P + pl.gamma_off, P + pl.beta_off, w.d_dec_a[1],
                                        G + pl.gamma_off, G + pl.beta_off, G + pl.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
      dec_bias_done[1] = true;
    }
  } else generic::bwd_dec_layer(m, P, F, w, G, s, 2);

  // LayerNorm backward of decoder layer 0; with the view weight gradient behind it, the channel-last planes that kernel reads
  // leave the same pass (gfx950_lnb_planes.h) and the split pass over d(a0) goes away
  const bool gd0_planes = (VAENPVC_LNB_PLANES & 4) && !lnq && F >= 1024 && bwd_on(7) && bwd_on(8) && cw_bwd(CW_D0, F);
  const bool gd0_only_planes = gd0_planes && (VAENPVC_LNB_PLANES & 8) && fcr_bwd(CV_D0G, F) && dense_planes_now() <= 2;
  auto lnb_dec0 = [&](const float* dy0) {
    const ConvL& pl = m.dec[0];
    if (gd0_only_planes)   // (the input-gradient kernel reads the planes too: no fp32 copy of d(a0) at all)
      for_dense_planes([&](auto npl) {
        constexpr int NPL = decltype(npl)::value;
        VAENPVC_TIMED("lnb_dec0", s, (launch_ln_bwd_planes<LnbCfg<32, 57>, NPL, CL_GD0, false>(dy0, w.dec_a[0], w.dec_st[0], P + pl.gamma_off, P + pl.beta_off,
                                         w.d_dec_a[0], us(w.cl[CL_GD0]), cl_plane(CL_GD0, F), G + pl.gamma_off, G + pl.beta_off, G + pl.b_off,
                                         w.scratch + Pk::lnpart, F, LWGS, s)));
      });
    else if (gd0_planes)
      for_dense_planes([&](auto npl) {
        constexpr int NPL = decltype(npl)::value;
        VAENPVC_TIMED("lnb_dec0", s, (launch_ln_bwd_planes<LnbCfg<32, 57>, NPL, CL_GD0, true>(dy0, w.dec_a[0], w.dec_st[0], P + pl.gamma_off, P + pl.beta_off,
                                         w.d_dec_a[0], us(w.cl[CL_GD0]), cl_plane(CL_GD0, F), G + pl.gamma_off, G + pl.beta_off, G + pl.b_off,
                                         w.scratch + Pk::lnpart, F, LWGS, s)));
      });
    else
      VAENPVC_TIMED("lnb_dec0", s, launch_ln_bwd<LnbCfg<32, 57>>(dy0, w.dec_a[0], w.dec_st[0], P + pl.gamma_off, P + pl.beta_off, w.d_dec_a[0],
                                       G + pl.gamma_off, G + pl.beta_off, G + pl.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
  };
  // ---- d1
  if (bwd_on(8) && bwd_on(9) && fb_bwd(FB_D1, F)) {
    const ConvL& pl = m.dec[0];
    float* dy0 = dy_cur == w.dy_tmp ? w.d_dec_a[1] : w.dy_tmp;
    // (round 5) decoder layer 0's LayerNorm backward inside layer 1's kernel: planes of d(a0) straight from the tile, no d(y0) in HBM
    if (rt().fb_lnb2 && gd0_only_planes && dense_planes_now() <= 2 && !abf) {
      const int nwg = (int)fbwd_grid_d1(dense_planes_now(), F);   // rows of parts = the grid the kernel is launched with
      fused_bwd(FB_D1, 1, dy_cur, dy0, "dec1_bwd", us(w.cl[CL_GD0]), cl_plane(CL_GD0, F), w.scratch + Pk::lnpart);
      VAENPVC_TIMED("lnb_dec0", s, hipLaunchKernelGGL(k_ln_bwd_reduce, dim3(3 * 32), dim3(256), 0, s, w.scratch + Pk::lnpart, nwg, 32,
                                                       G + pl.gamma_off, G + pl.beta_off, G + pl.b_off));
    } else {
      fused_bwd(FB_D1, 1, dy_cur, dy0, "dec1_bwd");
      lnb_dec0(dy0);
    }
    dec_bias_done[0] = true;
  } else if (bwd_on(8)) {
    const ConvL &l = m.dec[1], &pl = m.dec[0];
    WgArgs a{w.d_dec_a[1], nullptr, nullptr, nullptr, w.dec_a[0], w.dec_st[0], P + pl.gamma_off, P + pl.beta_off,
             G + l.w_off, F, 0};
    const bool fg = fc_bwd(CV_D1G, F), vg = !fg && cv_bwd(CV_D1G, F), fw = fw_bwd(CW_D1, F), vw = !fw && cw_bwd(CW_D1, F);
    if (vg || vw) gsplit(CL_GD1, w.d_dec_a[1], "dec1_gsplit");
    if (vw && !fwd_planes(8, CV_D1F, F)) asplit(CL_YD0, w.dec_a[0], w.dec_st[0], &pl, "dec1_asplit");
    ready();
    if (fw) fwg(CW_D1, w.dec_a[0], w.dec_st[0], &pl, w.d_dec_a[1], nullptr, nullptr, G + l.w_off, "dec1_wgrad");
    else if (vw) vwgrad(CW_D1, G + l.w_off, "dec1_wgrad");
    else VAENPVC_TIMED("dec1_wgrad", s2, launch_convwgrad<WD1>(a, WGS, s2));
    if (!dec_bias_done[1]) generic::bias_grad(w.d_dec_a[1], G + l.b_off, F, l.cout, l.hout, s);
    if (fg) fdgrad(CV_D1G, w.d_dec_a[1], w.dy_tmp, "dec1_dgrad");
    else if (vg) vdgrad(CV_D1G, w.dy_tmp, "dec1_dgrad");
    else
    VAENPVC_TIMED("dec1_dgrad", s, launch_convgemm<GD1>(conv_args(w.d_dec_a[1], nullptr, nullptr, nullptr, P + l.w_off, nullptr,
                                                                  w.dy_tmp, F), nsplit_for<GD1>(F), s));
    lnb_dec0(w.dy_tmp);
    dec_bias_done[0] = true;
  } else generic::bwd_dec_layer(m, P, F, w, G, s, 1);

  // d(h) straight to the planes of the two merge GEMMs (the input-gradient kernel owns whole frames and has them in LDS): every consumer
  // of d(h) on this path must be a plane kernel, and the per-speaker sums are then taken from the planes
  const bool dh_planes = rt().d0g_planes && gd0_only_planes && fcr_otl(CV_D0G, dense_planes_now()) && bwd_on(6) && pg_bwd(F) && pg_fwd(F) && fwd_on(6) &&
                         VAENPVC_SPLIT_SEGSUM && F >= 64;
  // ---- d0
  if (bwd_on(7)) {
    const ConvL& l = m.dec[0];
    WgArgs a{w.d_dec_a[0], nullptr, nullptr, nullptr, w.h, nullptr, nullptr, nullptr, G + l.w_off, F, 0};
    const bool rg = fcr_bwd(CV_D0G, F), vg = !rg && cv_bwd(CV_D0G, F), vw = cw_bwd(CW_D0, F);
    if ((vg || vw) && !gd0_planes) gsplit(CL_GD0, w.d_dec_a[0], "dec0_gsplit");
    if (vw && !fwd_planes(7, CV_D0F, F) && !d0f_leaves_planes(F)) asplit(CL_H, w.h, nullptr, nullptr, "dec0_asplit");
    ready();
    if (vw) vwgrad(CW_D0, G + l.w_off, "dec0_wgrad");
    else VAENPVC_TIMED("dec0_wgrad", s2, launch_convwgrad<WD0>(a, WGS, s2));
    if (!dec_bias_done[0]) generic::bias_grad(w.d_dec_a[0], G + l.b_off, F, l.cout, l.hout, s);
    if (rg) rdgrad(CV_D0G, w.scratch + Pk::cvw + cv_woff(CV_D0G), w.d_dec_a[0], w.d_h, "dec0_dgrad", gd0_only_planes ? w.cl[CL_GD0] : nullptr,
                   dh_planes ? w.pl_dh : nullptr, 1600);
    else if (vg) vdgrad(CV_D0G, w.d_h, "dec0_dgrad");
    else
    VAENPVC_TIMED("dec0_dgrad", s, (F < SMALL_BATCH_FRAMES ? launch_convgemm<GD0s>(conv_args(w.d_dec_a[0], nullptr, nullptr, nullptr, w.scratch + Pk::gd0,
                                                                  nullptr, w.d_h, F), nsplit_for<GD0s>(F), s) : launch_convgemm<GD0>(conv_args(w.d_dec_a[0], nullptr, nullptr, nullptr, w.scratch + Pk::gd0,
                                                                  nullptr, w.d_h, F), nsplit_for<GD0>(F), s)));
  } else generic::bwd_dec_layer(m, P, F, w, G, s, 0);
  if (lnq) flush_ln_reduce(*lnq, s);
  bucket(m.dec[0].w_off, m.n_params);  // all decoder conv layers (kernels, biases, LayerNorm parameters)

  // ---- merge + embedding
  if (bwd_on(6)) {
    // (the plane kernels read the z planes the forward pass left behind: both directions must be on them)
    const bool pgm = pg_bwd(F) && pg_fwd(F) && fwd_on(6);
    if (pgm) {
      for_dense_planes([&](auto npl) {
        constexpr int NPL = decltype(npl)::value;
        if (VAENPVC_SPLIT_SEGSUM && F >= 64) {   // the planes of d(h) and the per-speaker column sums S in one pass over d(h)
        // (the chunk partials go to dy_tmp: the decoder's backward pass, its only user on this path, is behind us on this stream)
        int nch = 0;
        if (dh_planes) VAENPVC_TIMED("merge_dsplit", s, (nch = launch_segsum_planes<NPL, MERGE_NY>(us(w.pl_dh), (int64_t)F * 1600, y, 1539, 1600, F, w.dy_tmp, s)));
        else
        VAENPVC_TIMED("merge_dsplit", s, (nch = launch_split_segsum<NPL, MERGE_NY>(w.d_h, y, 1539, 1600, F, us(w.pl_dh), w.dy_tmp, s)));
        VAENPVC_TIMED("merge_segsum", s, launch_sum_parts(w.dy_tmp, nch, MERGE_NY * 1539, w.scratch + Pk::merge_s, s));
        } else {
        SplitArgs sa = split_args(w.d_h, 1539, 1600, F, us(w.pl_dh));
        VAENPVC_TIMED("merge_dsplit", s, launch_split<NPL>(sa, s));
        }
        ready();
        TnpArgs t = tnp_args(w.pl_z, 128, w.pl_dh, 1600, 128, 1539, F, G + m.wz_off, 1539);
        t.tn4 = bwd_on(16);
        VAENPVC_TIMED("merge_wgrad", s2, (launch_gemm_tn<NPL, TN_EPI_PLAIN>(t, TN_WGS_DENSE, s2)));
      });
    } else {
    TnArgs a = tn_args(w.z, 128, w.d_h, 1539, 128, 1539, F, G + m.wz_off, 1539);
    ready();
    VAENPVC_TIMED("merge_wgrad", s2, launch_tngemm(a, false, kchunks_for(F, 13), s2));
    }
    // S[k] = per-speaker column sums of d(h); the bias gradients, dWy = E^T S and dE = S Wy^T follow from it
    float* Sg = w.scratch + Pk::merge_s;
    if (!(pgm && VAENPVC_SPLIT_SEGSUM && F >= 64)) {
      (void)hipMemsetAsync(Sg, 0, (size_t)MERGE_NY * 1539 * 4, s);
      int ch = cmax(1, cmin_(cdiv(F, 64), 128)), fc = cdiv(F, ch);
      VAENPVC_TIMED("merge_segsum", s, hipLaunchKernelGGL(k_segsum_atomic<MERGE_NY>, dim3((unsigned)cdiv(1539, 256), (unsigned)cdiv(F, fc)), dim3(256), 0, s,
                                                          w.d_h, y, 1539, F, fc, Sg));
    }
    const int nb_w = cdiv(128 * 1539, 256), nb_e = cdiv(MERGE_NY * 128, 4), nb_b = cdiv(1539, 256);
    VAENPVC_TIMED("merge_small", s, hipLaunchKernelGGL(k_merge_small<MERGE_NY>, dim3((unsigned)(nb_w + nb_e + nb_b)), dim3(256), 0, s, Sg, P + m.emb_off, P + m.wy_off, 128,
                       1539, G + m.wy_off, G + m.emb_off, G + m.bz_off, G + m.by_off, G + m.bm_off, nb_w, nb_e));
    if (pgm) {
      for_dense_planes([&](auto npl) {
        constexpr int NPL = decltype(npl)::value;
        NtArgs a = nt_args(w.pl_dh, F, 1600, w.scratch + Pk::pg_mergeb, 128, 128, w.d_z, 128);
        VAENPVC_TIMED("merge_dgrad", s, launch_gemm_nt<NPL>(a, s));
      });
    } else {
      DenseArgs d = dense_args(w.d_h, w.scratch + Pk::merge_b, w.d_z, 128, F);
      if (F < SMALL_BATCH_FRAMES) {   // K = 1539 in seven chunks on seven workgroups per tile
        (void)hipMemsetAsync(w.d_z, 0, (size_t)F * 128 * 4, s);
        VAENPVC_TIMED("merge_dgrad", s, launch_densegemm<MergeBs>(d, s, 7));
      } else
      VAENPVC_TIMED("merge_dgrad", s, launch_densegemm<MergeB>(d, s));
    }
  } else generic::bwd_merge(m, P, y, F, w, G, s);
  bucket(m.wz_off, m.dec[0].w_off);  // the two merge FCs and the three merge biases (the embedding goes last)

  const bool heads_tuned = bwd_on(5);
  const bool dz_planes = VAENPVC_DZ_PLANES && F >= 1024 && bwd_on(5) && pg_bwd(F) && pg_fwd(F) && fwd_on(5);
  if (heads_tuned) {  // sampler + KL backward fused with the two head-bias gradients
    const int rch = cmax(1, cmin_(cdiv(F, 32), 1024)), rfc = cdiv(F, rch);
    if (dz_planes)   // straight to the planes [dz_mu | dz_lv] of the two head GEMMs
      for_dense_planes([&](auto npl) {
        // (parts in the LayerNorm-backward scratch: every launch that used it so far on this stream has been reduced)
        VAENPVC_TIMED("reparam_bwd", s, hipLaunchKernelGGL((k_reparam_bwd_planes<decltype(npl)::value>), dim3((unsigned)cdiv(F, rfc)), dim3(256), 0, s, w.d_z, w.z_mu,
                           w.z_lv, eps, us(w.pl_dz), w.scratch + Pk::lnpart, (int)F, rfc, 1.0f / (float)F);
                      hipLaunchKernelGGL(k_colsum_part2, dim3(256), dim3(256), 0, s, w.scratch + Pk::lnpart, cdiv(F, rfc), G + m.bmu_off, G + m.blv_off));
      });
    else
    VAENPVC_TIMED("reparam_bwd", s, hipLaunchKernelGGL(k_reparam_bwd_colsum, dim3((unsigned)cdiv(F, rfc)), dim3(256), 0, s, w.d_z, w.z_mu, w.z_lv, eps, w.d_z_mu,
                       w.d_z_lv, G + m.bmu_off, G + m.blv_off, (int)F, rfc, 1.0f / (float)F));
  } else generic::bwd_reparam(m, eps, F, w, s);

  // ---- heads
  const bool da4_planes = (VAENPVC_LNB_PLANES & 1) && !lnq && F >= 1024 && bwd_on(5) && pg_bwd(F) && pg_fwd(F) && fwd_on(5) && bwd_on(4) && fwd_on(4);
  const bool ge3_planes = (VAENPVC_LNB_PLANES & 2) && !lnq && F >= 1024 && bwd_on(3) && bwd_on(4) && cv_bwd(CV_E3G, F) && cw_bwd(CW_E3, F);
  if (bwd_on(5) && pg_bwd(F) && pg_fwd(F) && fwd_on(5)) {
    const ConvL& l4 = m.enc[4];
    for_dense_planes([&](auto npl) {
      constexpr int NPL = decltype(npl)::value;
      SplitArgs sa = split_args(w.d_z_mu, 256, 256, F, us(w.pl_dz));   // [dz_mu | dz_lv]
      sa.k1 = 128;
      sa.ld1 = 128;
      sa.src2 = w.d_z_lv;
      sa.ld2 = 128;
      if (!dz_planes) VAENPVC_TIMED("heads_dsplit", s, launch_split<NPL>(sa, s));
      ready();
      TnpArgs t = tnp_args(w.pl_y4, 768, w.pl_dz, 256, 768, 256, F, G + m.wmu_off, 128);
      t.C2 = G + m.wlv_off;
      t.split = 128;
      t.tn4 = bwd_on(16);
      VAENPVC_TIMED("heads_wgrad", s2, (launch_gemm_tn<NPL, TN_EPI_PLAIN>(t, TN_WGS_DENSE, s2)));
      NtArgs a = nt_args(w.pl_dz, F, 256, w.scratch + Pk::pg_headsb, 768, 768, w.dy_tmp, 768);
      VAENPVC_TIMED("heads_dgrad", s, launch_gemm_nt<NPL>(a, s));
    });
    if (da4_planes)   // d(a4) leaves as the planes the two dense-shaped GEMMs of layer 4 read; no fp32 copy, no split pass
      for_dense_planes([&](auto npl) {
        constexpr int NPL = decltype(npl)::value;
        VAENPVC_TIMED("lnb_enc4", s, (launch_ln_bwd_planes<LnbCfg<256, 3>, NPL, -1, false>(w.dy_tmp, w.enc_a[4], w.enc_st[4], P + l4.gamma_off, P + l4.beta_off,
                                         w.d_enc_a[4], us(w.pl_da4), (int64_t)F * 768, G + l4.gamma_off, G + l4.beta_off, G + l4.b_off,
                                         w.scratch + Pk::lnpart, F, LWGS, s)));
      });
    else
    VAENPVC_TIMED("lnb_enc4", s, launch_ln_bwd<LnbCfg<256, 3>>(w.dy_tmp, w.enc_a[4], w.enc_st[4], P + l4.gamma_off, P + l4.beta_off, w.d_enc_a[4],
                                     G + l4.gamma_off, G + l4.beta_off, G + l4.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
    enc_bias_done[4] = true;
  } else if (bwd_on(5)) {
    const ConvL& l4 = m.enc[4];
    TnArgs a = tn_args(w.enc_a[4], 768, w.d_z_mu, 128, 768, 128, F, G + m.wmu_off, 128);
    a.st = w.enc_st[4];
    a.gamma = P + l4.gamma_off;
    a.beta = P + l4.beta_off;
    a.lndiv = 3;
    ready();
    VAENPVC_TIMED("heads_wgrad", s2, launch_tngemm(a, false, kchunks_for(F, 6), s2));
    a.Y = w.d_z_lv;
    a.C = G + m.wlv_off;
    launch_tngemm(a, false, kchunks_for(F, 6), s2);
    // (the two head-bias gradients were accumulated by k_reparam_bwd_colsum)
    DenseArgs d = dense_args(w.d_z_mu, w.scratch + Pk::heads_b, w.dy_tmp, 768, F);
    d.in2 = w.d_z_lv;
    if (F < SMALL_BATCH_FRAMES) {
      (void)hipMemsetAsync(w.dy_tmp, 0, (size_t)F * 768 * 4, s);
      VAENPVC_TIMED("heads_dgrad", s, launch_densegemm<HeadsBs>(d, s, 4));
    } else
    VAENPVC_TIMED("heads_dgrad", s, launch_densegemm<HeadsB>(d, s));
    VAENPVC_TIMED("lnb_enc4", s, launch_ln_bwd<LnbCfg<256, 3>>(w.dy_tmp, w.enc_a[4], w.enc_st[4], P + l4.gamma_off, P + l4.beta_off, w.d_enc_a[4],
                                     G + l4.gamma_off, G + l4.beta_off, G + l4.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
    enc_bias_done[4] = true;
  } else generic::bwd_heads(m, P, F, w, G, s);
  bucket(m.wmu_off, m.wz_off);  // the two dense heads

  // ---- encoder convs
  auto wg_enc = [&](int i) {
    const ConvL &l = m.enc[i], &pl = m.enc[i - 1];
    return WgArgs{w.enc_a[i - 1], w.enc_st[i - 1], P + pl.gamma_off, P + pl.beta_off, w.d_enc_a[i], nullptr, nullptr, nullptr,
                  G + l.w_off, F, 0};
  };
  auto lnb_enc3 = [&]() {   // (with both view GEMMs of layer 3 behind it: straight to their channel-last planes, no fp32 copy)
    const ConvL& pl = m.enc[3];
    if (ge3_planes)
      for_dense_planes([&](auto npl) {
        constexpr int NPL = decltype(npl)::value;
        VAENPVC_TIMED("lnb_enc3", s, (launch_ln_bwd_planes<LnbCfg<128, 7>, NPL, CL_GE3, false>(w.dy_tmp, w.enc_a[3], w.enc_st[3], P + pl.gamma_off, P + pl.beta_off,
                                         w.d_enc_a[3], us(w.cl[CL_GE3]), cl_plane(CL_GE3, F), G + pl.gamma_off, G + pl.beta_off, G + pl.b_off,
                                         w.scratch + Pk::lnpart, F, LWGS, s)));
      });
    else
      VAENPVC_TIMED("lnb_enc3", s, launch_ln_bwd<LnbCfg<128, 7>>(w.dy_tmp, w.enc_a[3], w.enc_st[3], P + pl.gamma_off, P + pl.beta_off, w.d_enc_a[3],
                                       G + pl.gamma_off, G + pl.beta_off, G + pl.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
  };
  if (bwd_on(4) && pg_bwd(F) && pg_fwd(F) && fwd_on(4)) {
    // layer 4 as a dense layer: dW from [F,896]^T x [F,768] folded back onto the 7 taps, d(y3) = d(a4) x Wd
    const ConvL &l = m.enc[4], &pl = m.enc[3];
    for_dense_planes([&](auto npl) {
      constexpr int NPL = decltype(npl)::value;
      if (!da4_planes) VAENPVC_TIMED("enc4_dsplit", s, launch_split<NPL>(split_args(w.d_enc_a[4], 768, 768, F, us(w.pl_da4)), s));
      ready();
      TnpArgs t = tnp_args(w.pl_y3, 896, w.pl_da4, 768, 896, 768, F, G + l.w_off, 0);
      t.tn4 = bwd_on(16);
      VAENPVC_TIMED("enc4_wgrad", s2, (launch_gemm_tn<NPL, TN_EPI_ENC4>(t, 512, s2)));
      NtArgs a = nt_args(w.pl_da4, F, 768, w.scratch + Pk::pg_enc4b, 896, 896, w.dy_tmp, 896);
      VAENPVC_TIMED("enc4_dgrad", s, launch_gemm_nt<NPL>(a, s));
    });
    if (!enc_bias_done[4]) generic::bias_grad(w.d_enc_a[4], G + l.b_off, F, l.cout, l.hout, s);
    lnb_enc3();
    enc_bias_done[3] = true;
  } else if (bwd_on(4)) {
    const ConvL &l = m.enc[4], &pl = m.enc[3];
    ready();
    VAENPVC_TIMED("enc4_wgrad", s2, launch_convwgrad<WE4>(wg_enc(4), WGS, s2));
    if (!enc_bias_done[4]) generic::bias_grad(w.d_enc_a[4], G + l.b_off, F, l.cout, l.hout, s);
    VAENPVC_TIMED("enc4_dgrad", s, launch_convgemm<GE4>(conv_args(w.d_enc_a[4], nullptr, nullptr, nullptr, w.scratch + Pk::ge4,
                                                                  nullptr, w.dy_tmp, F), nsplit_for<GE4>(F), s));
    lnb_enc3();
    enc_bias_done[3] = true;
  } else generic::bwd_enc_layer(m, P, x, F, w, G, s, 4);
  if (bwd_on(3)) {
    const ConvL &l = m.enc[3], &pl = m.enc[2];
    const bool vg = cv_bwd(CV_E3G, F), vw = cw_bwd(CW_E3, F);
    if ((vg || vw) && !ge3_planes) gsplit(CL_GE3, w.d_enc_a[3], "enc3_gsplit");
    if (vw && !fwd_planes(3, CV_E3F, F)) asplit(CL_Y2, w.enc_a[2], w.enc_st[2], &pl, "enc3_asplit");
    ready();
    if (vw) vwgrad(CW_E3, G + l.w_off, "enc3_wgrad");
    else VAENPVC_TIMED("enc3_wgrad", s2, launch_convwgrad<WE3>(wg_enc(3), WGS, s2));
    if (!enc_bias_done[3]) generic::bias_grad(w.d_enc_a[3], G + l.b_off, F, l.cout, l.hout, s);
    // the frame-owning tile with layer 2's LayerNorm backward in its epilogue (round 5): d(a2) straight from the GEMM, no d(y2) in HBM
    int lnb_rows = 0;
    if (vg && rt().cg_pf && rt().cg_lnb && !lnq && bwd_on(2))
      for_dense_planes([&](auto npl) {
        VAENPVC_TIMED("enc3_dgrad", s, lnb_rows = cv_gemm_lnb<decltype(npl)::value>(CV_E3G, w.scratch + Pk::cvw + cv_woff(CV_E3G), w.cl[CVS[CV_E3G].x], w.d_enc_a[2],
                                                    w.enc_a[2], w.enc_st[2], P + pl.gamma_off, P + pl.beta_off, w.scratch + Pk::lnpart,
                                                    (int64_t)2048 * 3 * 256, F, s));
      });
    if (lnb_rows > 0) {
      VAENPVC_TIMED("lnb_enc2", s, hipLaunchKernelGGL(k_ln_bwd_reduce, dim3(3 * 64), dim3(256), 0, s, w.scratch + Pk::lnpart, lnb_rows, 64,
                                                       G + pl.gamma_off, G + pl.beta_off, G + pl.b_off));
    } else {
    if (vg) vdgrad(CV_E3G, w.dy_tmp, "enc3_dgrad");
    else
    VAENPVC_TIMED("enc3_dgrad", s, (F < SMALL_BATCH_FRAMES ? launch_convgemm<GE3s>(conv_args(w.d_enc_a[3], nullptr, nullptr, nullptr, w.scratch + Pk::ge3,
                                                                  nullptr, w.dy_tmp, F), nsplit_for<GE3s>(F), s) : launch_convgemm<GE3>(conv_args(w.d_enc_a[3], nullptr, nullptr, nullptr, w.scratch + Pk::ge3,
                                                                  nullptr, w.dy_tmp, F), nsplit_for<GE3>(F), s)));
    VAENPVC_TIMED("lnb_enc2", s, launch_ln_bwd<LnbCfg<64, 19>>(w.dy_tmp, w.enc_a[2], w.enc_st[2], P + pl.gamma_off, P + pl.beta_off, w.d_enc_a[2],
                                     G + pl.gamma_off, G + pl.beta_off, G + pl.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
    }
    enc_bias_done[2] = true;
  } else generic::bwd_enc_layer(m, P, x, F, w, G, s, 3);
  if (bwd_on(2)) {
    const ConvL &l = m.enc[2], &pl = m.enc[1];
    const bool rg = fcr_bwd(CV_E2G, F), fg = !rg && fc_bwd(CV_E2G, F), vg = !rg && !fg && cv_bwd(CV_E2G, F), fw = fw_bwd(CW_E2, F), vw = !fw && cw_bwd(CW_E2, F);
    if (vg || vw) gsplit(CL_GE2, w.d_enc_a[2], "enc2_gsplit");
    if (vw && !fwd_planes(2, CV_E2F, F)) asplit(CL_Y1, w.enc_a[1], w.enc_st[1], &pl, "enc2_asplit");
    ready();
    if (fw) fwg(CW_E2, w.d_enc_a[2], nullptr, nullptr, w.enc_a[1], w.enc_st[1], &pl, G + l.w_off, "enc2_wgrad");
    else if (vw) vwgrad(CW_E2, G + l.w_off, "enc2_wgrad");
    else VAENPVC_TIMED("enc2_wgrad", s2, launch_convwgrad<WE2>(wg_enc(2), WGS, s2));
    if (!enc_bias_done[2]) generic::bias_grad(w.d_enc_a[2], G + l.b_off, F, l.cout, l.hout, s);
    if (rg) rdgrad(CV_E2G, w.scratch + Pk::cvwr_e2g, w.d_enc_a[2], w.dy_tmp, "enc2_dgrad");
    else if (fg) fdgrad(CV_E2G, w.d_enc_a[2], w.dy_tmp, "enc2_dgrad");
    else if (vg) vdgrad(CV_E2G, w.dy_tmp, "enc2_dgrad");
    else
    VAENPVC_TIMED("enc2_dgrad", s, (F < SMALL_BATCH_FRAMES ? launch_convgemm<GE2s>(conv_args(w.d_enc_a[2], nullptr, nullptr, nullptr, w.scratch + Pk::ge2,
                                                                  nullptr, w.dy_tmp, F), nsplit_for<GE2s>(F), s) : launch_convgemm<GE2>(conv_args(w.d_enc_a[2], nullptr, nullptr, nullptr, w.scratch + Pk::ge2,
                                                                  nullptr, w.dy_tmp, F), nsplit_for<GE2>(F), s)));
    if (!(bwd_on(1) && fb_bwd(FB_E1, F))) {   // (layer 1's fused backward kernel does it otherwise, gfx950_fbwd.h)
      VAENPVC_TIMED("lnb_enc1", s, launch_ln_bwd<LnbCfg<32, 57>>(w.dy_tmp, w.enc_a[1], w.enc_st[1], P + pl.gamma_off, P + pl.beta_off, w.d_enc_a[1],
                                       G + pl.gamma_off, G + pl.beta_off, G + pl.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
      enc_bias_done[1] = true;
    }
  } else generic::bwd_enc_layer(m, P, x, F, w, G, s, 2);
  // where encoder layer 0's backward finds the gradient at its activated output
  const float* dy_e0 = w.dy_tmp;
  if (bwd_on(1) && bwd_on(2) && fb_bwd(FB_E1, F)) {
    // LayerNorm backward + input gradient + weight gradient + parameter sums of encoder layer 1 in one kernel; its result goes to
    // the buffer of layer 0's pre-LN gradient (free until layer 0's LayerNorm backward, which may then run in place)
    const ConvL& pl = m.enc[0];
    fused_bwd(FB_E1, 1, w.dy_tmp, w.d_enc_a[0], "enc1_bwd");
    dy_e0 = w.d_enc_a[0];
    if (!enc0_fused)
      VAENPVC_TIMED("lnb_enc0", s, launch_ln_bwd<LnbCfg<16, 171>>(dy_e0, w.enc_a[0], w.enc_st[0], P + pl.gamma_off, P + pl.beta_off, w.d_enc_a[0],
                                        G + pl.gamma_off, G + pl.beta_off, G + pl.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
    enc_bias_done[0] = true;
  } else if (bwd_on(1)) {
    const ConvL &l = m.enc[1], &pl = m.enc[0];
    const bool fg = fc_bwd(CV_E1G, F), vg = !fg && cv_bwd(CV_E1G, F), fw = fw_bwd(CW_E1, F), vw = !fw && cw_bwd(CW_E1, F);
    if (vg || vw) gsplit(CL_GE1, w.d_enc_a[1], "enc1_gsplit");
    if (vw && !fwd_planes(1, CV_E1F, F)) asplit(CL_Y0, w.enc_a[0], w.enc_st[0], &pl, "enc1_asplit");
    ready();
    if (fw) fwg(CW_E1, w.d_enc_a[1], nullptr, nullptr, w.enc_a[0], w.enc_st[0], &pl, G + l.w_off, "enc1_wgrad");
    else if (vw) vwgrad(CW_E1, G + l.w_off, "enc1_wgrad");
    else VAENPVC_TIMED("enc1_wgrad", s2, launch_convwgrad<WE1>(wg_enc(1), WGS, s2));
    if (!enc_bias_done[1]) generic::bias_grad(w.d_enc_a[1], G + l.b_off, F, l.cout, l.hout, s);
    if (fg) fdgrad(CV_E1G, w.d_enc_a[1], w.dy_tmp, "enc1_dgrad");
    else if (vg) vdgrad(CV_E1G, w.dy_tmp, "enc1_dgrad");
    else
    VAENPVC_TIMED("enc1_dgrad", s, launch_convgemm<GE1>(conv_args(w.d_enc_a[1], nullptr, nullptr, nullptr, w.scratch + Pk::ge1,
                                                                  nullptr, w.dy_tmp, F), nsplit_for<GE1>(F), s));
    if (!enc0_fused)
    VAENPVC_TIMED("lnb_enc0", s, launch_ln_bwd<LnbCfg<16, 171>>(w.dy_tmp, w.enc_a[0], w.enc_st[0], P + pl.gamma_off, P + pl.beta_off, w.d_enc_a[0],
                                      G + pl.gamma_off, G + pl.beta_off, G + pl.b_off, w.scratch + Pk::lnpart, F, LWGS, s, lnq));
    enc_bias_done[0] = true;
  } else generic::bwd_enc_layer(m, P, x, F, w, G, s, 1);
  if (enc0_fused) {
    // LayerNorm backward + weight gradient + parameter sums of encoder layer 0 in one pass over d(activated output) and x
    const ConvL& l = m.enc[0];
    const int nwg = cmin_(cdiv(F, 64), 512);
    float* pw = w.scratch + Pk::enc0part;
    float* pc = w.scratch + Pk::lnpart;     // (no other LayerNorm partial rows are alive: the layers before flushed theirs)
    rt().ensure_lds(reinterpret_cast<const void*>(&k_enc0_bwd_wave), Enc0BwdCfg::LDS_BYTES);
    VAENPVC_TIMED("enc0_bwd", s, hipLaunchKernelGGL(k_enc0_bwd_wave, dim3((unsigned)nwg), dim3(256), Enc0BwdCfg::LDS_BYTES, s, x, dy_e0,
                                                    w.enc_st[0], P + l.w_off, P + l.b_off, P + l.gamma_off, P + l.beta_off, pw, pc, F));
    VAENPVC_TIMED("enc0_reduce", s, hipLaunchKernelGGL(k_colsum_part, dim3(7 * 16), dim3(256), 0, s, pw, nwg, 7 * 16, G + l.w_off));
    VAENPVC_TIMED("enc0_reduce", s, hipLaunchKernelGGL(k_ln_bwd_reduce, dim3(3 * 16), dim3(256), 0, s, pc, nwg, 16, G + l.gamma_off, G + l.beta_off, G + l.b_off));
  } else if (bwd_on(0)) {
    const ConvL& l = m.enc[0];
    WgArgs a{x, nullptr, nullptr, nullptr, w.d_enc_a[0], nullptr, nullptr, nullptr, G + l.w_off, F, 0};
    ready();
    if (enc0_wave(rt().bwd_mask, F)) {   // register-resident partial sums, one row of partials per workgroup, one column-sum pass
      const int nwg = cmin_(cdiv(F, 128), 256);   // >= 32 frames per wave: the 112 wave reductions at the end stay below 10 % of a wave's work
      VAENPVC_TIMED("enc0_wgrad", s2, hipLaunchKernelGGL(k_enc0_wgrad_wave, dim3((unsigned)nwg), dim3(256), 0, s2, x, w.d_enc_a[0],
                                                        w.scratch + Pk::enc0part, F));
      VAENPVC_TIMED("enc0_reduce", s2, hipLaunchKernelGGL(k_colsum_part, dim3(7 * 16), dim3(256), 0, s2, w.scratch + Pk::enc0part, nwg, 7 * 16, G + l.w_off));
    } else
    VAENPVC_TIMED("enc0_wgrad", s2, launch_convwgrad<WE0>(a, WGS, s2));
    if (!enc_bias_done[0]) generic::bias_grad(w.d_enc_a[0], G + l.b_off, F, l.cout, l.hout, s);
  } else generic::bwd_enc_layer(m, P, x, F, w, G, s, 0);
  if (lnq) flush_ln_reduce(*lnq, s);
  bucket(0, m.wmu_off);  // speaker embedding + encoder convs
  if (fork) rt().stream_dep(s2, s);  // join: everything after backward (Adam) sees all gradients
}


// ---------------------------------------------------------------- backward of a small batch (gfx950_frame.h)
// The input-gradient chain with every LayerNorm backward runs as ONE launch, one frame per workgroup (frame_backward);
// what is left are the weight gradients, which depend on nothing but tensors that launch wrote: they are dealt to the
// caller's stream and the context's helper stream and run next to each other.
float* frame_zero_region(const Ws& w, int* count) {
  *count = MERGE_NY * 1539;
  return w.scratch + Pk::merge_s;
}
void backward_frame(const Model& m, const float* P, const float* x, const float* target, const int64_t* y, const float* eps,
                    int64_t F64, const Ws& w, float* G, hipStream_t s, bool g_zeroed, float* loss3) {
  const int F = (int)F64;
  if (!g_zeroed) {
    (void)hipMemsetAsync(G, 0, (size_t)m.n_params * 4, s);
    (void)hipMemsetAsync(w.scratch + Pk::merge_s, 0, (size_t)MERGE_NY * 1539 * 4, s);
  }
  // bit 20 of the backward mask (default set): every parameter gradient in ONE launch (gfx950_frame_wgrad.h); cleared = the
  // layered weight-gradient kernels below on two streams (A/B, parity tests)
  const bool one_launch = bwd_on(20) && w.frame_y != nullptr;
  frame_backward(m, P, target ? target : x, eps, F, w, G, s, !one_launch, loss3);
  if (one_launch) {
    frame_wgrad(m, P, x, y, F, w, G, s);
    Runtime& r0 = rt();
    // every gradient comes out of the one launch above: ONE range = one all-reduce of the whole buffer (four ranges handed
    // over at the same moment would be four back-to-back collectives with nothing to overlap: three extra latencies on a
    // 0.3 ms step)
    if (r0.bucket_cb) r0.bucket_cb(r0.bucket_user, r0.bucket_next++, 0, m.n_params, (void*)s);
    return;
  }
  hipStream_t side = bwd_on(30) ? rt().side_stream() : nullptr;
  const bool fork = side != nullptr;
  hipStream_t s2 = fork ? side : s;
  if (fork) rt().stream_dep(s, s2);
  const int WGS = 512;
  // ---- helper stream: the decoder's weight gradients
  {
    TnArgs a = tn_args(w.dec_y, 4104, w.d_xh, 513, 4096, 512, F, G + m.dec[3].w_off, 0);
    VAENPVC_TIMED("dec3_wgrad", s2, launch_tngemm(a, true, kchunks_for(F, 32 * 4), s2));
    int ech = cmax(1, cmin_(cdiv(F, 64), 128));
    int efc = rup(cdiv(F, ech), 64);
    VAENPVC_TIMED("dec3_row512", s2, hipLaunchKernelGGL(k_toep_wgrad_row512, dim3((unsigned)cdiv(F, efc), 3), dim3(256), 0, s2, w.dec_y, w.d_xh, G + m.dec[3].w_off, F, efc));
    VAENPVC_TIMED("dec3_bias", s2, hipLaunchKernelGGL(k_sum_all_atomic, dim3((unsigned)cmin_(1024, cdiv(F * 513, 1024))), dim3(256), 0, s2, w.d_xh,
                       (int64_t)F * 513, G + m.dec[3].b_off));
    for (int i = 2; i >= 0; --i) {
      const ConvL& l = m.dec[i];
      WgArgs a2{w.d_dec_a[i], nullptr, nullptr, nullptr, i ? w.dec_a[i - 1] : w.h, i ? w.dec_st[i - 1] : nullptr,
                i ? P + m.dec[i - 1].gamma_off : nullptr, i ? P + m.dec[i - 1].beta_off : nullptr, G + l.w_off, F, 0};
      if (i == 2) VAENPVC_TIMED("dec2_wgrad", s2, launch_convwgrad<WD2>(a2, WGS, s2));
      else if (i == 1) VAENPVC_TIMED("dec1_wgrad", s2, launch_convwgrad<WD1>(a2, WGS, s2));
      else VAENPVC_TIMED("dec0_wgrad", s2, launch_convwgrad<WD0>(a2, WGS, s2));
    }
  }
  // ---- caller's stream: merge, heads, encoder
  {
    TnArgs a = tn_args(w.z, 128, w.d_h, 1539, 128, 1539, F, G + m.wz_off, 1539);
    VAENPVC_TIMED("merge_wgrad", s, launch_tngemm(a, false, kchunks_for(F, 13), s));
    float* Sg = w.scratch + Pk::merge_s;      // (zeroed by the step's frame_pack launch, or above)
    int ch = cmax(1, cmin_(cdiv(F, 64), 128)), fc = cdiv(F, ch);
    VAENPVC_TIMED("merge_segsum", s, hipLaunchKernelGGL(k_segsum_atomic<MERGE_NY>, dim3((unsigned)cdiv(1539, 256), (unsigned)cdiv(F, fc)), dim3(256), 0, s, w.d_h, y, 1539, F, fc, Sg));
    const int nb_w = cdiv(128 * 1539, 256), nb_e = cdiv(MERGE_NY * 128, 4), nb_b = cdiv(1539, 256);
    VAENPVC_TIMED("merge_small", s, hipLaunchKernelGGL(k_merge_small<MERGE_NY>, dim3((unsigned)(nb_w + nb_e + nb_b)), dim3(256), 0, s, Sg, P + m.emb_off, P + m.wy_off, 128,
                       1539, G + m.wy_off, G + m.emb_off, G + m.bz_off, G + m.by_off, G + m.bm_off, nb_w, nb_e));
    // (recomputes d(z_mu), d(z_lv) from d(z) exactly as the frame kernel did; what is needed here are the two bias sums)
    const int rch = cmax(1, cmin_(cdiv(F, 32), 1024)), rfc = cdiv(F, rch);
    VAENPVC_TIMED("reparam_bwd", s, hipLaunchKernelGGL(k_reparam_bwd_colsum, dim3((unsigned)cdiv(F, rfc)), dim3(256), 0, s, w.d_z, w.z_mu, w.z_lv, eps, w.d_z_mu,
                       w.d_z_lv, G + m.bmu_off, G + m.blv_off, (int)F, rfc, 1.0f / (float)F));
    const ConvL& l4 = m.enc[4];
    TnArgs h = tn_args(w.enc_a[4], 768, w.d_z_mu, 128, 768, 128, F, G + m.wmu_off, 128);
    h.st = w.enc_st[4];
    h.gamma = P + l4.gamma_off;
    h.beta = P + l4.beta_off;
    h.lndiv = 3;
    VAENPVC_TIMED("heads_wgrad", s, launch_tngemm(h, false, kchunks_for(F, 6), s));
    h.Y = w.d_z_lv;
    h.C = G + m.wlv_off;
    launch_tngemm(h, false, kchunks_for(F, 6), s);
    auto wg_enc = [&](int i) {
      const ConvL &l = m.enc[i], &pl = m.enc[i - 1];
      return WgArgs{w.enc_a[i - 1], w.enc_st[i - 1], P + pl.gamma_off, P + pl.beta_off, w.d_enc_a[i], nullptr, nullptr, nullptr,
                    G + l.w_off, F, 0};
    };
    VAENPVC_TIMED("enc4_wgrad", s, launch_convwgrad<WE4>(wg_enc(4), WGS, s));
    VAENPVC_TIMED("enc3_wgrad", s, launch_convwgrad<WE3>(wg_enc(3), WGS, s));
    VAENPVC_TIMED("enc2_wgrad", s, launch_convwgrad<WE2>(wg_enc(2), WGS, s));
    VAENPVC_TIMED("enc1_wgrad", s, launch_convwgrad<WE1>(wg_enc(1), WGS, s));
    WgArgs a0{x, nullptr, nullptr, nullptr, w.d_enc_a[0], nullptr, nullptr, nullptr, G + m.enc[0].w_off, F, 0};
    VAENPVC_TIMED("enc0_wgrad", s, launch_convwgrad<WE0>(a0, WGS, s));
  }
  if (fork) rt().stream_dep(s2, s);
  // gradient ranges for the data-parallel host: everything is enqueued, report the four ranges on the caller's stream
  Runtime& r = rt();
  if (r.bucket_cb) {
    const int64_t cut[5] = {m.n_params, m.dec[0].w_off, m.wz_off, m.wmu_off, 0};
    for (int b = 0; b < 4; ++b) r.bucket_cb(r.bucket_user, r.bucket_next++, cut[b + 1], cut[b] - cut[b + 1], (void*)s);
  }
}

}  // namespace tuned
}  // namespace vaenpvc

#if VAENPVC_PROF
// developer-only entry point of instrumented variant builds (not part of include/vaenpvc.h)
extern "C" int vaenpvc_debug_conv_prof(unsigned long long* out, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(vaenpvc::tuned::g_conv_prof), sizeof(unsigned long long) * 320) != hipSuccess) return -3;
  if (out && hipMemcpyFromSymbol(out + 320, HIP_SYMBOL(vaenpvc::tuned::g_wg_prof), sizeof(unsigned long long) * 128) != hipSuccess) return -3;
  if (out && hipMemcpyFromSymbol(out + 448, HIP_SYMBOL(vaenpvc::tuned::g_tb_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -3;
  if (out && hipMemcpyFromSymbol(out + 456, HIP_SYMBOL(vaenpvc::tuned::g_tw_prof), sizeof(unsigned long long) * 8) != hipSuccess) return -3;
  if (reset) {
    static unsigned long long z[320];
    if (hipMemcpyToSymbol(HIP_SYMBOL(vaenpvc::tuned::g_conv_prof), z, sizeof(z)) != hipSuccess) return -3;
    if (hipMemcpyToSymbol(HIP_SYMBOL(vaenpvc::tuned::g_wg_prof), z, sizeof(unsigned long long) * 128) != hipSuccess) return -3;
    if (hipMemcpyToSymbol(HIP_SYMBOL(vaenpvc::tuned::g_tb_prof), z, sizeof(unsigned long long) * 8) != hipSuccess) return -3;
    if (hipMemcpyToSymbol(HIP_SYMBOL(vaenpvc::tuned::g_tw_prof), z, sizeof(unsigned long long) * 8) != hipSuccess) return -3;
  }
  return 0;
}
#endif
