// gfx950_frame.hip -- launchers of the small-batch frame kernels (gfx950_frame.h): one workgroup = one frame.
//   frame_pack   aligned / transposed weight copies (+ optional zero fill of the gradient buffer), once per step
//   frame_fwd    encoder -> heads -> sampler -> merge -> decoder -> log-density of every frame, then the batch means
//   frame_bwd    input-gradient chain of every frame with the LayerNorm backward in place
//   frame_wgrad  every parameter gradient, one job-list launch (gfx950_frame_wgrad.h)
//   frame_lnp    per-frame channel sums -> gradients of the LayerNorm parameters and conv biases (only behind the LAYERED
//                weight gradients, backward-mask bit 20 cleared: the job list has its own segment for them)
//   toep_fwd / toep_bwd  train steps up to 128 frames: the 1025-tap layer between the two frame kernels (frame_split_on)
// Reference: model/vae.py:72-137 (forward), trainer/vae.py:24 (autodiff).
#include "gfx950_frame.h"
#include "gfx950_frame_wgrad.h"
#include "gfx950_frame_dev.h"

#include <cstdlib>
#include <cstring>

#include "cl_layout.h"
#include "kernels.h"

namespace vaenpvc {
namespace tuned {
using namespace frame;
static_assert(Pk::total == FRAME_PK_FLOATS && LNP_C == FRAME_LNP_C, "workspace constants (cl_layout.h) out of date");
static inline int cmin_i(int a, int b) { return a < b ? a : b; }
// workgroups of a pass: one per frame up to 1024 (VAENPVC_FRAME_GRID: developer override for experiments)
static int frame_grid(int F) {
  static const int env = getenv("VAENPVC_FRAME_GRID") ? atoi(getenv("VAENPVC_FRAME_GRID")) : 0;
  return env > 0 ? cmin_i(env, F) : cmin_i(F, 1024);
}

static long long* g_prof_dev = nullptr;   // [2][512]: forward, backward (debug only; allocated on first use)
static long long* prof_buf() {
  if (!g_prof_dev && getenv("VAENPVC_FRAME_PROF")) {
    if (hipMalloc(&g_prof_dev, 2 * 512 * sizeof(long long)) != hipSuccess) g_prof_dev = nullptr;
    else (void)hipMemset(g_prof_dev, 0, 2 * 512 * sizeof(long long));
  }
  return g_prof_dev;
}

__global__ void __launch_bounds__(256) k_frame_pack(const float* __restrict__ P, POff off, float* __restrict__ pk,
                                                    float* __restrict__ zero, int nzero, float* __restrict__ zero2, int nzero2) {
  const int stride = gridDim.x * blockDim.x;
  if (pk)      // (null: only the zero fills -- a backward pass on weights packed by an earlier call)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Pk::total; i += stride) pk[i] = pack_src(P, off, i);
  // the step's zero fills ride along (gradient buffer, per-speaker sums of the merge backward): no memset launches
  if (zero)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += stride) zero[i] = 0.f;
  if (zero2)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero2; i += stride) zero2[i] = 0.f;
}

template <bool PROF>
__global__ void __launch_bounds__(NT) k_frame_fwd(FwdArgs a, PhiloxKey key, int draw, long long* prof) {
  extern __shared__ __attribute__((aligned(16))) float frame_lds[];
  DevRunner<PROF> run(prof, 0);
  if (PROF && blockIdx.x == 0 && threadIdx.x == 0) prof[0] = clock64();
  const FwdArgs& la = args_to_lds<FwdArgs>(frame_lds);      // (`a` itself is never addressed: see args_to_lds)
  if (draw) key = philox_resolve(key);
  const float* eps = la.eps;
  frame_prologue(run, frame_lds, la.P, la.off);
  for (int f = blockIdx.x; f < la.F; f += gridDim.x)
    frame_fwd(run, frame_lds, la, f, [&](int ff, int d) -> float {
      if (draw) return philox_normal(key, (uint64_t)ff * 128 + (uint64_t)d);
      return eps ? eps[(size_t)ff * 128 + d] : 0.f;
    });
}

template <bool PROF>
__global__ void __launch_bounds__(NT) k_frame_bwd(BwdArgs a, long long* prof) {
  extern __shared__ __attribute__((aligned(16))) float frame_lds[];
  DevRunner<PROF> run(prof, 0);
  if (PROF && blockIdx.x == 0 && threadIdx.x == 0) prof[0] = clock64();
  const BwdArgs& la = args_to_lds<BwdArgs>(frame_lds);
  frame_prologue(run, frame_lds, la.P, la.off);
  for (int f = blockIdx.x; f < la.F; f += gridDim.x) frame_bwd(run, frame_lds, la, f);
}

// every parameter gradient of a small batch (gfx950_frame_wgrad.h): a block looks its job up by block index
__global__ void __launch_bounds__(WT) k_frame_wgrad(WgArgs a, WgPlan pl) {
  extern __shared__ __attribute__((aligned(16))) float wg_lds[];
  WRunner run;
  frame_wgrad_block(run, wg_lds, a, pl, (int)blockIdx.x);
}

// batch means {G, D_KL, logP} (model/vae.py:112-128), one block, fixed summation order
__global__ void __launch_bounds__(256) k_frame_loss(const float* __restrict__ kl_f, const float* __restrict__ nll_f, int F,
                                                    float* __restrict__ loss3) {
  __shared__ float sa[256], sb[256];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < F; i += 256) {
    a += kl_f[i];
    b += nll_f[i];
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    float x = 0.f, y = 0.f;
    for (int i = 0; i < 256; ++i) {
      x += sa[i];
      y += sb[i];
    }
    const float kl = x / (float)F, lp = y / (float)F;
    loss3[0] = -lp + kl;
    loss3[1] = kl;
    loss3[2] = lp;
  }
}

// The 1025-tap layer of a train step, eight workgroups per frame (gfx950_frame.h: toep_split_fwd / toep_split_bwd), between
// the two frame kernels.  The second one also finishes the losses: per-frame log-density from the eight partial sums, and
// (workgroup 0) the batch means in the fixed order of k_frame_loss.
__global__ void __launch_bounds__(TS_T) k_frame_toep_fwd(const float* __restrict__ dec_y, const float* __restrict__ pk,
                                                         const float* __restrict__ P, int b3_off, const float* __restrict__ target,
                                                         float* __restrict__ xh, float* __restrict__ nll8) {
  extern __shared__ __attribute__((aligned(16))) float ts_lds[];
  WRunner run;
  const int f = blockIdx.x >> 3, og = blockIdx.x & 7;
  toep_split_fwd(run, ts_lds, dec_y + (size_t)f * 4104, pk + Pk::w3t, P[b3_off], target + (size_t)f * TP_H, og, xh + (size_t)f * TP_H,
                 nll8 + (size_t)f * 8);
}
__global__ void __launch_bounds__(TS_T) k_frame_toep_bwd(const float* __restrict__ xh, const float* __restrict__ target,
                                                         const float* __restrict__ pk, float invF, float* __restrict__ d_xh,
                                                         float* __restrict__ d_y2, const float* __restrict__ kl_f,
                                                         const float* __restrict__ nll8, float* __restrict__ nll_f, int F,
                                                         float* __restrict__ loss3) {
  extern __shared__ __attribute__((aligned(16))) float ts_lds[];
  WRunner run;
  const int f = blockIdx.x >> 3, c = blockIdx.x & 7;
  toep_split_bwd(run, ts_lds, xh + (size_t)f * TP_H, target + (size_t)f * TP_H, pk + Pk::w3t, c, invF, d_xh + (size_t)f * TP_H,
                 d_y2 + (size_t)f * 4104);
  auto nll_of = [&](int i) {
    float t = 0.f;
    for (int k = 0; k < 8; ++k) t += nll8[(size_t)i * 8 + k];
    return t;
  };
  if (!nll8) return;                   // uniform: backward pass against another target (the losses came from elsewhere)
  if (c == 0 && threadIdx.x == 0) nll_f[f] = nll_of(f);
  if (blockIdx.x == 0 && loss3) {      // uniform
    float* sa = ts_lds;
    float* sb = ts_lds + TS_T;
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < F; i += TS_T) {
      a += kl_f[i];
      b += nll_of(i);
    }
    sa[threadIdx.x] = a;
    sb[threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x == 0) {
      float x = 0.f, y = 0.f;
      for (int i = 0; i < TS_T; ++i) {
        x += sa[i];
        y += sb[i];
      }
      const float kl = x / (float)F, lp = y / (float)F;
      loss3[0] = -lp + kl;
      loss3[1] = kl;
      loss3[2] = lp;
    }
  }
}

// gradients of the LayerNorm offsets / scales and the conv biases of the 8 normalised layers: sums over frames of the
// per-frame channel sums the backward pass left (fixed order: bitwise repeatable)
struct LnpDst {
  int beta[8], gamma[8], bias[8];   // destination offsets in the gradient buffer, layer order of LNP_*
};
__global__ void __launch_bounds__(256) k_frame_lnp(const float* __restrict__ lnp, int F, LnpDst d, float* __restrict__ G) {
  // block = 64 channel slots of one k; the four waves take every fourth frame, partial rows combined through LDS
  __shared__ float sm[4][64];
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;      // (k, channel slot), padded per k
  constexpr int SLOTS = (LNP_C + 63) / 64 * 64;
  const int k = i / SLOTS, cs = i % SLOTS;
  float s = 0.f;
  if (cs < LNP_C)
    for (int f = w; f < F; f += 4) s += lnp[((size_t)f * 3 + k) * LNP_C + cs];
  sm[w][threadIdx.x & 63] = s;
  __syncthreads();
  if (w == 0 && cs < LNP_C) {
    s = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
    constexpr int OFFS[9] = {LNP_DEC2, LNP_DEC1, LNP_DEC0, LNP_ENC4, LNP_ENC3, LNP_ENC2, LNP_ENC1, LNP_ENC0, LNP_C};
    int l = 0;
    while (cs >= OFFS[l + 1]) ++l;
    G[(k == 0 ? d.beta[l] : k == 1 ? d.gamma[l] : d.bias[l]) + cs - OFFS[l]] = s;
  }
}

// developer read-out of the phase stamps (not part of include/vaenpvc.h)
extern "C" int vaenpvc_debug_frame_prof(long long* out /*[2][512]*/) {
  if (!g_prof_dev) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  return hipMemcpy(out, g_prof_dev, 2 * 512 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}

// ---------------------------------------------------------------------------------------------- host side
static POff poff_of(const Model& m) {
  POff o;
  o.emb = (int)m.emb_off;
  for (int i = 0; i < 5; ++i) {
    o.ew[i] = (int)m.enc[i].w_off;
    o.eb[i] = (int)m.enc[i].b_off;
    o.ebeta[i] = (int)m.enc[i].beta_off;
    o.egamma[i] = (int)m.enc[i].gamma_off;
  }
  o.wmu = (int)m.wmu_off;
  o.bmu = (int)m.bmu_off;
  o.wlv = (int)m.wlv_off;
  o.blv = (int)m.blv_off;
  o.wz = (int)m.wz_off;
  o.bz = (int)m.bz_off;
  o.wy = (int)m.wy_off;
  o.by = (int)m.by_off;
  o.bm = (int)m.bm_off;
  for (int i = 0; i < 4; ++i) {
    o.dw[i] = (int)m.dec[i].w_off;
    o.db[i] = (int)m.dec[i].b_off;
    if (i < 3) {
      o.dbeta[i] = (int)m.dec[i].beta_off;
      o.dgamma[i] = (int)m.dec[i].gamma_off;
    }
  }
  return o;
}

int64_t frame_pack_floats() { return Pk::total; }
int64_t frame_lnp_floats(int64_t F) { return F * 3 * LNP_C; }

bool frame_fwd_on(int64_t F) {
  const Runtime& r = rt();
  return r.frame_max > 0 && F <= r.frame_max && ((r.fwd_mask >> 21) & 1u);
}
bool frame_bwd_on(int64_t F) {
  const Runtime& r = rt();
  return r.frame_max > 0 && F <= r.frame_max && ((r.bwd_mask >> 21) & 1u);
}

void frame_pack(const Model& m, const float* P, const Ws& w, float* G, float* zero2, int nzero2, hipStream_t s, bool zero_only) {
  // one packed element per thread: a thread that walks several elements pays an L2 round trip for each
  const int n = zero_only ? (int)m.n_params : Pk::total;
  hipLaunchKernelGGL(k_frame_pack, dim3((n + 255) / 256), dim3(256), 0, s, P, poff_of(m), zero_only ? (float*)nullptr : w.frame_pk, G,
                     G ? (int)m.n_params : 0, zero2, nzero2);
}

// Train steps run the 1025-tap layer outside the frame kernels (bit 18 of the backward mask, default set): needs the
// activated output of decoder layer 2 and the scratch both halves share
// (up to FRAME_SPLIT_MAX frames: the frame kernels leave compute units idle there; at 256 frames every unit holds a frame
//  and the 2 x 2048 extra workgroups cost more than the frame kernels save: 0.362 -> 0.378 ms, against 0.253 -> 0.229 ms at
//  16 frames and 0.273 -> 0.254 at 64; VAENPVC_FRAME_SPLIT_MAX overrides)
bool frame_split_on(const Ws& w, int64_t F) {
  static const int64_t fmax = getenv("VAENPVC_FRAME_SPLIT_MAX") ? atoll(getenv("VAENPVC_FRAME_SPLIT_MAX")) : 128;
  return ((rt().bwd_mask >> 18) & 1u) && F <= fmax && w.dec_y && w.dy_tmp && w.frame_lnp && w.d_xh;
}

// mode: FM_* bits.  x may be null for decode-only, z_in null unless decode-only.
void frame_forward(const Model& m, const float* P, const float* x, const float* target, const int64_t* y, const float* eps,
                   const PhiloxKey* key, const float* z_in, int64_t F, const Ws& w, float* xh_out, int mode, float* loss3,
                   hipStream_t s) {
  FwdArgs a;
  memset(&a, 0, sizeof a);
  a.P = P;
  a.pk = w.frame_pk;
  a.off = poff_of(m);
  a.x = x;
  a.target = target ? target : x;
  a.y = y;
  a.eps = eps;
  a.z_in = z_in;
  a.ny = m.ny;
  a.F = (int)F;
  a.mode = mode;
  a.invF = 1.0f / (float)F;
  for (int i = 0; i < 5; ++i) {
    a.enc_a[i] = w.enc_a[i];
    a.enc_st[i] = w.enc_st[i];
  }
  a.z_mu = w.z_mu;
  a.z_lv = w.z_lv;
  a.z = w.z;
  a.eps_out = w.eps;
  a.h = w.h;
  for (int i = 0; i < 3; ++i) {
    a.dec_a[i] = w.dec_a[i];
    a.dec_st[i] = w.dec_st[i];
  }
  a.xh = xh_out ? xh_out : w.xh;
  a.kl_f = w.kl_f;
  a.nll_f = w.nll_f;
  a.d_xh = w.d_xh;
  a.dec_y = (mode & FM_GRAD) ? w.dec_y : nullptr;      // (train step: operand of the last layer's weight gradient)
  if ((mode & FM_GRAD) && w.frame_y) {                 // activated layer outputs for the weight-gradient launch
    float* yb = w.frame_y;
    const int64_t ne[5] = {2736, 1824, 1216, 896, 768}, nd[2] = {1824, 2736};
    for (int i = 0; i < 5; ++i) {
      a.y_enc[i] = yb;
      yb += F * ne[i];
    }
    for (int i = 0; i < 2; ++i) {
      a.y_dec[i] = yb;
      yb += F * nd[i];
    }
  }
  if (!a.d_xh) a.mode &= ~FM_GRAD;
  // train step: the pass stops behind decoder layer 2; xh and the log-density come from k_frame_toep_fwd below, d(xh) and
  // the batch means from k_frame_toep_bwd (frame_backward)
  const bool split = (a.mode & FM_GRAD) && (a.mode & FM_LOSS) && frame_split_on(w, F);
  if (split) a.mode |= FM_NOD3;
  PhiloxKey k = key ? *key : PhiloxKey{0, 0, 0, 0, nullptr};
  if (long long* pb = prof_buf()) {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_frame_fwd<true>), L_TOTAL * 4);
    hipLaunchKernelGGL(k_frame_fwd<true>, dim3((unsigned)frame_grid((int)F)), dim3(NT), L_TOTAL * 4, s, a, k, key ? 1 : 0, pb);
  } else {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_frame_fwd<false>), L_TOTAL * 4);
    VAENPVC_TIMED("frame_fwd", s, hipLaunchKernelGGL(k_frame_fwd<false>, dim3((unsigned)frame_grid((int)F)), dim3(NT), L_TOTAL * 4, s, a, k, key ? 1 : 0, (long long*)nullptr));
  }
  if (split) {
    VAENPVC_TIMED("frame_toep_fwd", s, hipLaunchKernelGGL(k_frame_toep_fwd, dim3((unsigned)F * 8), dim3(TS_T), TS_FWD_LDS * 4, s, w.dec_y, w.frame_pk, P,
                                                          (int)m.dec[3].b_off, a.target, a.xh, w.frame_lnp));
    return;      // (the losses: k_frame_toep_bwd)
  }
  if (loss3 && (mode & FM_LOSS) && (mode & FM_SAMPLE))
    hipLaunchKernelGGL(k_frame_loss, dim3(1), dim3(256), 0, s, w.kl_f, w.nll_f, (int)F, loss3);
}

void frame_backward(const Model& m, const float* P, const float* target, const float* eps, int64_t F, const Ws& w, float* G,
                    hipStream_t s, bool lnp_sums, float* loss3) {
  BwdArgs a;
  memset(&a, 0, sizeof a);
  a.P = P;
  a.pk = w.frame_pk;
  a.off = poff_of(m);
  a.target = target;
  a.eps = eps;
  a.F = (int)F;
  a.invF = 1.0f / (float)F;
  for (int i = 0; i < 5; ++i) {
    a.enc_a[i] = w.enc_a[i];
    a.enc_st[i] = w.enc_st[i];
    a.d_enc_a[i] = w.d_enc_a[i];
  }
  a.z_mu = w.z_mu;
  a.z_lv = w.z_lv;
  for (int i = 0; i < 3; ++i) {
    a.dec_a[i] = w.dec_a[i];
    a.dec_st[i] = w.dec_st[i];
    a.d_dec_a[i] = w.d_dec_a[i];
  }
  a.xh = w.xh;
  a.d_xh = w.d_xh;
  a.d_h = w.d_h;
  a.d_z = w.d_z;
  a.d_z_mu = w.d_z_mu;
  a.d_z_lv = w.d_z_lv;
  a.lnp = w.frame_lnp;
  if (frame_split_on(w, F)) {
    // d(xh) and the gradient at decoder layer 2's activated output, eight workgroups per frame; with `loss3` (the step's own
    // forward pass came just before) also the per-frame log-density and the batch means from the partial sums that pass left
    // at the head of the frame_lnp region (the backward kernel overwrites it afterwards)
    VAENPVC_TIMED("frame_toep_bwd", s, hipLaunchKernelGGL(k_frame_toep_bwd, dim3((unsigned)F * 8), dim3(TS_T), TS_BWD_LDS * 4, s, w.xh, target, w.frame_pk,
                                                          a.invF, w.d_xh, w.dy_tmp, w.kl_f, loss3 ? w.frame_lnp : (const float*)nullptr, w.nll_f,
                                                          (int)F, loss3));
    a.d_y2 = w.dy_tmp;
  }
  if (long long* pb = prof_buf()) {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_frame_bwd<true>), L_TOTAL * 4);
    hipLaunchKernelGGL(k_frame_bwd<true>, dim3((unsigned)frame_grid((int)F)), dim3(NT), L_TOTAL * 4, s, a, pb + 512);
  } else {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_frame_bwd<false>), L_TOTAL * 4);
    VAENPVC_TIMED("frame_bwd", s, hipLaunchKernelGGL(k_frame_bwd<false>, dim3((unsigned)frame_grid((int)F)), dim3(NT), L_TOTAL * 4, s, a, (long long*)nullptr));
  }
  if (!lnp_sums) return;     // (the one-launch weight gradient reduces them itself)
  // gradients of the LayerNorm parameters and conv biases of the eight normalised layers
  LnpDst d;
  const ConvL* L[8] = {&m.dec[2], &m.dec[1], &m.dec[0], &m.enc[4], &m.enc[3], &m.enc[2], &m.enc[1], &m.enc[0]};
  for (int i = 0; i < 8; ++i) {
    d.beta[i] = (int)L[i]->beta_off;
    d.gamma[i] = (int)L[i]->gamma_off;
    d.bias[i] = (int)L[i]->b_off;
  }
  hipLaunchKernelGGL(k_frame_lnp, dim3((unsigned)(3 * ((LNP_C + 63) / 64))), dim3(256), 0, s, w.frame_lnp, (int)F, d, G);
}

// developer switch (scripts/wgrad_prof.py): bit s cleared = segment s of the job list is left out of the launch
static unsigned g_wg_seg_mask = 0xffffffffu;
extern "C" void vaenpvc_debug_wg_segments(unsigned mask) { g_wg_seg_mask = mask; }
// developer switch: most frame chunks of the nine chunked jobs (null = defaults)
static int g_wg_caps[9];
static bool g_wg_caps_set = false;
extern "C" void vaenpvc_debug_wg_caps(const int* caps) {
  g_wg_caps_set = caps != nullptr;
  if (caps) memcpy(g_wg_caps, caps, sizeof g_wg_caps);
}

void frame_wgrad(const Model& m, const float* P, const float* x, const int64_t* y, int64_t F, const Ws& w, float* G, hipStream_t s) {
  WgArgs a;
  memset(&a, 0, sizeof a);
  a.P = P;
  a.off = poff_of(m);
  a.x = x;
  a.y = y;
  a.ny = m.ny;
  a.F = (int)F;
  const float* yb = w.frame_y;
  const int64_t ne[5] = {2736, 1824, 1216, 896, 768}, nd[2] = {1824, 2736};
  for (int i = 0; i < 5; ++i) {
    a.y_enc[i] = yb;
    yb += F * ne[i];
    a.d_enc_a[i] = w.d_enc_a[i];
  }
  for (int i = 0; i < 2; ++i) {
    a.y_dec[i] = yb;
    yb += F * nd[i];
  }
  a.z = w.z;
  a.h = w.h;
  a.dec_y = w.dec_y;
  a.d_xh = w.d_xh;
  for (int i = 0; i < 3; ++i) a.d_dec_a[i] = w.d_dec_a[i];
  a.d_h = w.d_h;
  a.d_z_mu = w.d_z_mu;
  a.d_z_lv = w.d_z_lv;
  a.lnp = w.frame_lnp;
  a.pk = w.frame_pk;
  a.G = G;
  WgPlan pl = g_wg_caps_set ? make_wgplan((int)F, m.ny, g_wg_caps) : make_wgplan((int)F, m.ny);
  if (g_wg_seg_mask != 0xffffffffu) {      // developer: drop segments (blocks renumbered)
    WgPlan q = pl;
    int n = 0, blk = 0;
    for (int i = 0; i < pl.nseg; ++i)
      if ((g_wg_seg_mask >> i) & 1u) {
        q.start[n] = blk;
        q.kind[n] = pl.kind[i];
        q.layer[n] = pl.layer[i];
        q.tiles[n] = pl.tiles[i];
        q.fc[n] = pl.fc[i];
        blk += pl.start[i + 1] - pl.start[i];
        ++n;
      }
    q.nseg = n;
    q.start[n] = blk;
    pl = q;
    if (n == 0) return;
  }
  VAENPVC_TIMED("frame_wgrad", s, hipLaunchKernelGGL(k_frame_wgrad, dim3((unsigned)pl.start[pl.nseg]), dim3(WT), WG_LDS * 4, s, a, pl));
}

}  // namespace tuned
}  // namespace vaenpvc
