// runtime.hip -- per-context mutable state (see runtime.h).
#include "runtime.h"

#include <cstdlib>
#include <cstring>

namespace vaenpvc {

static thread_local Runtime* tl_rt = nullptr;
static Runtime g_detached;  // used only if a launcher runs outside an ABI call (never in the product)

Runtime& rt() { return tl_rt ? *tl_rt : g_detached; }
RtScope::RtScope(Runtime* r) : prev(tl_rt) { tl_rt = r; }
RtScope::~RtScope() { tl_rt = prev; }

// Developer knobs, read ONCE when the context is created (not process-global state):
//   VAENPVC_FWD_MASK / VAENPVC_BWD_MASK   initial kernel-selection masks
//   VAENPVC_SIDE_STREAM=0|1               weight gradients on the caller's stream / on the second stream at every batch size
//                                         (default: second stream except for two-plane operands from 16 384 frames per call on)
//   VAENPVC_TOEP=f32                      exact-fp32 MFMA kernels for the 1025-tap layer
//   VAENPVC_TOEP_WGRAD_F32                exact-fp32 weight gradient of that layer only
//   VAENPVC_PLANES=1|2|3                  bf16 terms per fp32 operand (vaenpvc_set_precision)
//   VAENPVC_DENSE_PLANES=1|2|3            terms on the dense-shaped layers regardless of the precision rule (experiments)
//   VAENPVC_CV_SITES=<mask>               conv sites on the view GEMMs (runtime.h: cv_sites; bit = CV_* / 12 + CW_* site)
//   VAENPVC_FC_SITES / _FCR_SITES / _FW_SITES=<mask>   thin / medium conv sites and thin weight gradients on the fused kernels
//   VAENPVC_TOEP_ZC=<n>                   frame chunks of the Toeplitz weight gradient (A/B measurements)
//   VAENPVC_FRAME_MAX=<n>                 largest batch on the whole-frame-per-workgroup kernels (0 = never, <= 1024)
//   VAENPVC_TOEP_WGRAD_K16, VAENPVC_TN_K16, VAENPVC_TN_XCD=0|1   earlier schedules / tile orders of the weight-gradient GEMMs (A/B)
void Runtime::read_env() {
  if (const char* e = getenv("VAENPVC_FWD_MASK")) fwd_mask = (unsigned)strtoul(e, nullptr, 0);
  if (const char* e = getenv("VAENPVC_BWD_MASK")) bwd_mask = (unsigned)strtoul(e, nullptr, 0);
  if (const char* e = getenv("VAENPVC_SIDE_STREAM")) {
    side_enabled = e[0] != '0';
    side_forced = side_enabled;   // explicitly on: at every batch size (default: gfx950_layers.hip, SIDE_STREAM_MAX_FRAMES)
  }
  if (const char* e = getenv("VAENPVC_TOEP")) toep_f32 = !strcmp(e, "f32");
  toep_wgrad_f32 = getenv("VAENPVC_TOEP_WGRAD_F32") != nullptr;
  if (const char* e = getenv("VAENPVC_TOEP_ZC")) toep_zc = atoi(e) > 0 ? atoi(e) : 4;
  if (const char* e = getenv("VAENPVC_TN_XCD")) tn_xcd = atoi(e);
  if (const char* e = getenv("VAENPVC_D2_TAIL")) d2_tail = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_D2_LNA")) d2_lna = atoi(e);
  if (const char* e = getenv("VAENPVC_FB_LNB2")) fb_lnb2 = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_D0G_PLANES")) d0g_planes = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_TN_D0FIT")) tn_d0fit = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_E2_OSP")) e2_osp = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_DY2_PAD")) dy2_pad = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_DXH_SKIP")) dxh_skip = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_NT_LEP")) nt_lep = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_NT_RING")) nt_ring = atoi(e);
  if (const char* e = getenv("VAENPVC_CG_SF_RING")) cg_sf_ring = atoi(e);
  if (const char* e = getenv("VAENPVC_CG_PF_RING")) cg_pf_ring = atoi(e);
  if (const char* e = getenv("VAENPVC_NT_AR")) nt_ar = atoi(e);
  if (const char* e = getenv("VAENPVC_CG_LNB")) cg_lnb = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_CG_SF")) cg_sf = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_CG_PF")) cg_pf = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_NT_PERSIST")) nt_persist = atoi(e);
  if (const char* e = getenv("VAENPVC_TN_W4_TILES")) tn_w4_tiles = atoi(e);
  if (const char* e = getenv("VAENPVC_FRAME_MAX")) frame_max = atoi(e) < 0 ? 0 : (atoi(e) > 1024 ? 1024 : atoi(e));
  tn_k16 = getenv("VAENPVC_TN_K16") != nullptr;
  toep_wgrad_k16 = getenv("VAENPVC_TOEP_WGRAD_K16") != nullptr;
  if (const char* e = getenv("VAENPVC_TOEP_WGRAD_W4")) toep_wgrad_w4 = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_CV_SITES")) cv_sites_env = (long)strtoul(e, nullptr, 0);
  if (const char* e = getenv("VAENPVC_FCR_SITES")) fcr_sites_env = (long)strtoul(e, nullptr, 0);
  if (const char* e = getenv("VAENPVC_FW_SITES")) fw_sites_env = (long)strtoul(e, nullptr, 0);
  if (const char* e = getenv("VAENPVC_FB_LAYERS")) fb_layers_env = (long)strtoul(e, nullptr, 0);
  if (const char* e = getenv("VAENPVC_ACT_BF16")) act_bf16 = atoi(e) != 0;
  if (const char* e = getenv("VAENPVC_FC_SITES")) fc_sites_env = (long)strtoul(e, nullptr, 0);
  if (const char* e = getenv("VAENPVC_DENSE_PLANES")) {
    int p = atoi(e);
    if (p >= 1 && p <= 3) dense_planes = p;
  }
  if (const char* e = getenv("VAENPVC_PLANES")) {
    int p = atoi(e);
    if (p >= 1 && p <= 3) planes = p;
  }
}

void Runtime::release() {
  if (s2) {
    (void)hipStreamSynchronize(s2);
    (void)hipStreamDestroy(s2);
    for (auto& e : ev)
      if (e) (void)hipEventDestroy(e);
  }
  s2 = nullptr;
  for (auto& e : ev) e = nullptr;
  ev_next = 0;
  for (auto& p : pool) {
    (void)hipEventDestroy(p.first);
    (void)hipEventDestroy(p.second);
  }
  pool.clear();
  used = 0;
  attr_done.clear();
}

void Runtime::bind_device() {
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) return;
  if (d == device) return;
  if (device >= 0) {  // the context moved to another device: its stream, events and attributes do not follow
    int cur = d;
    (void)hipSetDevice(device);
    release();
    (void)hipSetDevice(cur);
  }
  device = d;
}

void Runtime::ensure_lds(const void* fn, int bytes) {
  if (attr_done.count(fn)) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  attr_done.insert(fn);
}

hipStream_t Runtime::side_stream() {
  if (!side_enabled) return nullptr;
  if (s2) return s2;
  if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) {
    s2 = nullptr;
    side_enabled = false;
    return nullptr;
  }
  for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  return s2;
}

void Runtime::stream_dep(hipStream_t from, hipStream_t to) {
  hipEvent_t e = ev[ev_next];
  ev_next = (ev_next + 1) % 16;
  (void)hipEventRecord(e, from);
  (void)hipStreamWaitEvent(to, e, 0);
}

static const size_t kPoolMax = 16384;
void Runtime::timer_begin(hipStream_t s) {
  if (used >= kPoolMax) return;
  if (used >= pool.size()) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    pool.emplace_back(a, b);
  }
  (void)hipEventRecord(pool[used].first, s);
}
void Runtime::timer_end(hipStream_t s) {
  if (used >= kPoolMax || used >= pool.size()) return;
  (void)hipEventRecord(pool[used].second, s);
  ++used;
}

}  // namespace vaenpvc
