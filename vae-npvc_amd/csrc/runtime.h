// runtime.h -- mutable per-context state of the library.
//
// include/vaenpvc.h promises "no global mutable state besides the ctx": everything that changes
// after load time lives in a Runtime owned by one vaenpvc_ctx -- the kernel-selection masks,
// the operand precision, the internal weight-gradient stream with its event ring, the
// per-device function-attribute cache and the single-kernel event timer.  An ABI entry point
// locks its context, binds the Runtime to the calling host thread for the duration of the call
// (RtScope) and the launch code reaches it through rt(); two contexts driven by two host
// threads therefore never share anything mutable.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <unordered_set>
#include <utility>
#include <vector>

namespace vaenpvc {

struct Runtime {
  // ---- kernel selection (vaenpvc_set_tuned_masks / vaenpvc_set_precision)
  unsigned fwd_mask = 0xffffffffu, bwd_mask = 0xffffffffu;
  int planes = 2;               // bf16 terms per fp32 operand on the bf16 matrix cores: 3, 2 or 1
  int dense_planes = 0;         // developer override for the dense-shaped layers (VAENPVC_DENSE_PLANES); 0 = by rule
  // conv layers on the view GEMMs (gfx950_viewconv.h): one bit per site (CV_* forward / input-gradient sites 0..11,
  // weight-gradient sites 12..17); -1 = the measured default of the precision (VAENPVC_CV_SITES overrides)
  long cv_sites_env = -1, fc_sites_env = -1;   // (VAENPVC_FC_SITES: thin sites on the fused kernel, gfx950_fconv.h)
  long fcr_sites_env = -1;      // VAENPVC_FCR_SITES: medium sites on the register-weight fused kernel (gfx950_fconv_r.h; bit = CV_* site)
  unsigned fcr_sites() const { return fcr_sites_env >= 0 ? (unsigned)fcr_sites_env : planes == 1 ? FCR_SITES_BF16 : FCR_SITES; }
  static constexpr unsigned FCR_SITES = 0x28au, FCR_SITES_BF16 = 0x28au;   // CV_E2F (1), CV_D0F (3), CV_E2G (7), CV_D0G (9; round 5: with one plane as well -- reading the planes
                                                                            //  decoder layer 1's fused backward now leaves, the LayerNorm pass in between is gone there too: 4.21 -> 4.02 ms)
  long fw_sites_env = -1;       // VAENPVC_FW_SITES: thin weight gradients on the fused kernel (gfx950_fwgrad.h; bit = CW_* site)
  unsigned fw_sites() const { return fw_sites_env >= 0 ? (unsigned)fw_sites_env : planes == 1 ? FW_SITES_BF16 : FW_SITES; }
  static constexpr unsigned FW_SITES = 0x3fu, FW_SITES_BF16 = 0x3fu;   // (round 5: encoder layer 2 as well -- its view GEMM needed two split passes since the forward kernels no
                                                                        //  longer leave its planes: 4.32 -> 4.26 ms per step in the bf16 mode, same box)
  unsigned fc_sites() const { return fc_sites_env >= 0 ? (unsigned)fc_sites_env : planes == 1 ? FC_SITES_BF16 : FC_SITES; }
  static constexpr unsigned FC_SITES = 0xdb1u, FC_SITES_BF16 = 0xdb1u;   // by measurement
  unsigned cv_sites() const { return cv_sites_env >= 0 ? (unsigned)cv_sites_env : planes == 1 ? CV_SITES_BF16 : planes == 2 ? CV_SITES_X2 : CV_SITES_X3; }
  static constexpr unsigned CV_SITES_BF16 = 0xe2ceu, CV_SITES_X2 = 0xc244u, CV_SITES_X3 = 0x4u;  // by measurement
  long fb_layers_env = -1;      // VAENPVC_FB_LAYERS: thin decoder layers whose whole backward step is one kernel (gfx950_fbwd.h; bit = FB_* layer)
  unsigned fb_layers() const { return fb_layers_env >= 0 ? (unsigned)fb_layers_env : 0x7u; }
  bool act_bf16 = false;        // VAENPVC_ACT_BF16=1: bf16 HBM storage of the thin decoder layers' tensors in the bf16 mode (correct, tested;
                                // measured 5.29 -> 5.41 ms: the kernels that touch them are bound by the number of vector-memory operations in
                                // flight, not by bytes, and the phase-stacked epilogue needs two stores where fp32 needs one -- off by default)
  bool tn_k16 = false;          // VAENPVC_TN_K16: the 16-row two-workgroup A^T B kernel instead of the pipelined 32-row one (A/B)
  int tn_w4_tiles = 8;          // VAENPVC_TN_W4_TILES: the four-wave A^T B kernel from this many 256 x 256 tiles per row chunk on (0: every plain site, 99: never)
  int nt_persist = 0;           // VAENPVC_NT_PERSIST=<n>: C = A B^T launches with at least n 128 x 128 tiles run on persistent workgroups (two per CU walk the
                                // tiles, the next tile's first loads ahead of the result stores; 0 = never).  OFF: measured SLOWER (round 5, same box,
                                // two interleaved rounds): encoder layer 4 forward 173.5 -> 190.7 us, its input gradient 171.4 -> 188.3, merge forward
                                // 130.9 -> 146.4, heads input gradient 71.2 -> 84.5
  bool cg_pf = true;            // VAENPVC_CG_PF=0: encoder layer 3's input gradient on the 64 x 256 view-GEMM tiles instead of the frame-owning 192 x 128 tile (A/B)
  bool cg_sf = true;            // VAENPVC_CG_SF=0: encoder layer 3 forward on the one-tile view GEMM + the separate statistics / planes pass (A/B)
  bool cg_lnb = true;           // VAENPVC_CG_LNB=0: encoder layer 2's LayerNorm backward as its own pass behind layer 3's input gradient (A/B)
  int nt_ar = 0;                // VAENPVC_NT_AR: the merge forward GEMM on the A-resident kernel (k_gemm_nt_ar): 1 = from 192 row tiles on (one workgroup
                                // per CU), 0 = never, 2 = whenever the shape is served (parity tests at small batches).  OFF: measured 102 us against
                                // 98 us for the one-tile kernel once both store through LDS, and the decoder layer behind it runs 20 us slower
                                // (279 -> 299 us: the rows of h leave the caches in another order); round 5, three interleaved rounds on one box
  int nt_ring = 1;              // VAENPVC_NT_RING: C = A B^T sites with K >= 256 on the four-wave LDS-DMA ring kernel (gfx950_ntring.h): 1 = from 128 tiles of
                                // 256 x 128 on, 2 = whenever the shape is served (parity tests), 0 = never
  int cg_sf_ring = 1;           // (2: at any batch size -- parity tests; 1: from 9 216 frames on = one 36-frame tile per CU)  VAENPVC_CG_SF_RING=0: encoder layer 3 forward on k_cgemm_sf (two-barrier loop, 18 frames per tile) instead of the ring kernel's main loop
                                // (k_cgemm_sf_ring, 36 whole frames per tile: 166 -> 155 us, round 6; A/B)
  int cg_pf_ring = 0;           // VAENPVC_CG_PF_RING: encoder layer 3's input gradient (+ layer 2's LayerNorm backward) on the (3, 3) ring kernel, 24 whole frames per tile
  bool nt_lep = true;           // VAENPVC_NT_LEP=0: C = A B^T results stored straight from the accumulators (4 bytes per lane) instead of through LDS (A/B)
  bool dxh_skip = true;         // VAENPVC_DXH_SKIP=0: the loss kernel also stores d(xh) as fp32 (nothing reads it when both last-layer GEMMs take its bf16 planes; A/B)
  bool dy2_pad = true;          // VAENPVC_DY2_PAD=0: the 1025-tap layer's input gradient in the tensor's own 513-float rows (unaligned 16-byte stores; A/B)
  bool e2_osp = true;           // VAENPVC_E2_OSP=0: statistics + activated planes of encoder layer 2's output in their own pass (A/B)
  bool tn_d0fit = true;         // VAENPVC_TN_D0FIT=0: decoder layer 0's weight gradient on 128 x 256 tiles (36 % of the MFMA work useful) instead of 96 x 288 (A/B)
  bool fb_lnb2 = true;          // VAENPVC_FB_LNB2=0: decoder layer 0's LayerNorm backward as its own pass behind layer 1's fused backward kernel (A/B)
  bool d0g_planes = true;       // VAENPVC_D0G_PLANES=0: decoder layer 0's input gradient leaves as fp32 d(h) and a split pass makes the merge GEMMs' planes (A/B)
  int d2_lna = 1;               // (2: also below 16 384 frames per step, one channel group per frame tile -- parity tests)  VAENPVC_D2_LNA=0: the separate pass between decoder layer 2 and the 1025-tap layer (k_ln_stats_act_planes: statistics, activation,
                                // operand planes, column 512) instead of statistics in layer 2's epilogue + LayerNorm on load in the 1025-tap forward kernel (A/B)
  bool d2_tail = false;         // VAENPVC_D2_TAIL=1: the pass between decoder layer 2 and the 1025-tap layer (statistics, planes, bin 512, column 512) in the
                                // epilogue of layer 2's forward kernel (k_fconv<TAIL>).  OFF: built, parity-green, NOT faster -- 457 us against 205 + 240 us for
                                // the two kernels (round 5, same box): the epilogue's ~1 100 vector instructions and three barriers per 2-frame group are
                                // serial work in a kernel that then fits two workgroups per CU instead of three (255 registers, 66 KB of LDS), while the
                                // separate pass streams at 4.6 TB/s with 16 waves per CU
  int tn_xcd = -1;              // VAENPVC_TN_XCD=0|1: tile order of the C += A^T B plane GEMM (experiments; -1 = per site)
  int toep_zc = 4;              // VAENPVC_TOEP_ZC: frame chunks of the Toeplitz weight gradient, 64 workgroups each (4: one workgroup per CU, one prologue / epilogue per CU)
  bool toep_f32 = false;        // VAENPVC_TOEP=f32: exact-fp32 MFMA kernels for the 1025-tap layer
  bool toep_wgrad_f32 = false;  // VAENPVC_TOEP_WGRAD_F32
  bool toep_wgrad_w4 = true;    // VAENPVC_TOEP_WGRAD_W4=0: the eight-wave kernel (64 x 64 wave tiles) at every batch size
  bool toep_wgrad_k16 = false;  // VAENPVC_TOEP_WGRAD_K16: 16-frame chunks in the bf16 weight gradient (A/B measurements)
  bool side_enabled = true;     // VAENPVC_SIDE_STREAM=0 disables the internal weight-gradient stream
  bool side_forced = false;     // VAENPVC_SIDE_STREAM=1: the second stream at every batch size
  int frame_max = 512;          // VAENPVC_FRAME_MAX: largest batch on the whole-frame-per-workgroup kernels (gfx950_frame.h); 0 = never.
                                // Bit 21 of a mask cleared = the layered kernels for that pass of this context (A/B, parity tests)
  // ---- what the last train forward of this context left in the workspace (vaenpvc_train_bwd_target re-uses it): batch size, kernel
  //      family (0 generic, 1 layered, 2 frame kernels), the masks / precision it ran under and the workspace it wrote
  int64_t last_F = -1;
  int last_path = -1, last_planes = 0;
  unsigned last_fwd_mask = 0, last_bwd_mask = 0;
  const void* last_ws = nullptr;
  int64_t plz_F = -1;           // batch size whose z planes tuned::reparam_fwd_planes left for the decoder_fwd that follows (-1: none)
  int64_t dxh_post_F = -1;      // batch size whose d(xh) planes / column 512 / bias parts tuned::loss_fwd_post left for the backward pass (-1: none)
  // ---- device binding: created lazily on the device that is current at the first launch
  int device = -1;
  hipStream_t s2 = nullptr;
  hipEvent_t ev[16] = {};
  int ev_next = 0;
  std::unordered_set<const void*> attr_done;  // kernels whose dynamic-LDS limit was raised on `device`
  // ---- gradient-bucket callback (vaenpvc_set_bucket_callback)
  void (*bucket_cb)(void* user, int32_t bucket, int64_t off, int64_t cnt, void* ready_stream) = nullptr;
  void* bucket_user = nullptr;
  int bucket_next = 0;
  // ---- single-kernel event timer (vaenpvc_timer_select / _read)
  std::string tag;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  size_t used = 0;

  void read_env();
  // (re)binds to the current device; drops stream, events and the attribute cache when it changed
  void bind_device();
  void release();
  // raises hipFuncAttributeMaxDynamicSharedMemorySize once per kernel and device
  void ensure_lds(const void* fn, int bytes);
  // internal stream for the weight-gradient kernels; nullptr when disabled or creation failed
  hipStream_t side_stream();
  // make `to` wait for everything enqueued on `from` so far
  void stream_dep(hipStream_t from, hipStream_t to);

  // `tag` is one site name or a comma-separated list of them (a kernel GROUP timed in one pass: bench.py's roofline.sites)
  bool timer_match(const char* t) const {
    if (tag.empty()) return false;
    const size_t n = strlen(t);
    for (size_t p = tag.find(t); p != std::string::npos; p = tag.find(t, p + 1))
      if ((p == 0 || tag[p - 1] == ',') && (p + n == tag.size() || tag[p + n] == ',')) return true;
    return false;
  }
  void timer_begin(hipStream_t s);
  void timer_end(hipStream_t s);
};

// Runtime bound to the calling thread by the ABI entry point that is executing
Runtime& rt();
struct RtScope {
  Runtime* prev;
  explicit RtScope(Runtime* r);
  ~RtScope();
};

}  // namespace vaenpvc
