// gfx950_frame.hip -- launchers of the small-batch frame kernels (gfx950_frame.h): one workgroup = one frame.
//   frame_pack   aligned / transposed weight copies (+ optional zero fill of the gradient buffer), once per step
//   frame_fwd    encoder -> heads -> sampler -> merge -> decoder -> log-density of every frame, then the batch means
//   frame_bwd    input-gradient chain of every frame with the LayerNorm backward in place
//   frame_lnp    per-frame channel sums -> gradients of the LayerNorm parameters and conv biases
// Reference: model/vae.py:72-137 (forward), trainer/vae.py:24 (autodiff).
#include "gfx950_frame.h"

#include <cstring>

#include "cl_layout.h"
#include "kernels.h"

namespace vaenpvc {
namespace tuned {
using namespace frame;
static_assert(Pk::total == FRAME_PK_FLOATS && LNP_C == FRAME_LNP_C, "workspace constants (cl_layout.h) out of date");
static inline int cmin_i(int a, int b) { return a < b ? a : b; }

struct DevRunner {
  template <class F>
  __device__ __forceinline__ void phase(F&& f) {
    f((int)threadIdx.x);
    __syncthreads();
  }
};

__global__ void __launch_bounds__(256) k_frame_pack(const float* __restrict__ P, POff off, float* __restrict__ pk,
                                                    float* __restrict__ zero, int nzero) {
  const int stride = gridDim.x * blockDim.x;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Pk::total; i += stride) pk[i] = pack_src(P, off, i);
  if (zero)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nzero; i += stride) zero[i] = 0.f;
}

__global__ void __launch_bounds__(NT) k_frame_fwd(FwdArgs a, PhiloxKey key, int draw) {
  extern __shared__ __attribute__((aligned(16))) float frame_lds[];
  DevRunner run;
  if (draw) key = philox_resolve(key);
  for (int f = blockIdx.x; f < a.F; f += gridDim.x)
    frame_fwd(run, frame_lds, a, f, [&](int ff, int d) -> float {
      if (draw) return philox_normal(key, (uint64_t)ff * 128 + (uint64_t)d);
      return a.eps ? a.eps[(size_t)ff * 128 + d] : 0.f;
    });
}

__global__ void __launch_bounds__(NT) k_frame_bwd(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float frame_lds[];
  DevRunner run;
  for (int f = blockIdx.x; f < a.F; f += gridDim.x) frame_bwd(run, frame_lds, a, f);
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

// gradients of the LayerNorm offsets / scales and the conv biases of the 8 normalised layers: sums over frames of the
// per-frame channel sums the backward pass left (fixed order: bitwise repeatable)
struct LnpDst {
  int beta[8], gamma[8], bias[8];   // destination offsets in the gradient buffer, layer order of LNP_*
};
__global__ void __launch_bounds__(64) k_frame_lnp(const float* __restrict__ lnp, int F, LnpDst d, float* __restrict__ G) {
  const int i = blockIdx.x * 64 + threadIdx.x;      // (k, channel slot)
  if (i >= 3 * LNP_C) return;
  const int k = i / LNP_C, cs = i % LNP_C;
  constexpr int OFFS[9] = {LNP_DEC2, LNP_DEC1, LNP_DEC0, LNP_ENC4, LNP_ENC3, LNP_ENC2, LNP_ENC1, LNP_ENC0, LNP_C};
  int l = 0;
  while (cs >= OFFS[l + 1]) ++l;
  float s = 0.f;
  for (int f = 0; f < F; ++f) s += lnp[((size_t)f * 3 + k) * LNP_C + cs];
  const int c = cs - OFFS[l];
  G[(k == 0 ? d.beta[l] : k == 1 ? d.gamma[l] : d.bias[l]) + c] = s;
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

void frame_pack(const Model& m, const float* P, const Ws& w, float* G, hipStream_t s) {
  hipLaunchKernelGGL(k_frame_pack, dim3(512), dim3(256), 0, s, P, poff_of(m), w.frame_pk, G, G ? (int)m.n_params : 0);
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
  a.dec_y = (mode & FM_GRAD) ? w.dec_y : nullptr;      // (train step: the layered weight gradient of the last layer reads it)
  if (!a.d_xh) a.mode &= ~FM_GRAD;
  PhiloxKey k = key ? *key : PhiloxKey{0, 0, 0, 0, nullptr};
  rt().ensure_lds(reinterpret_cast<const void*>(&k_frame_fwd), L_TOTAL * 4);
  VAENPVC_TIMED("frame_fwd", s, hipLaunchKernelGGL(k_frame_fwd, dim3((unsigned)cmin_i((int)F, 1024)), dim3(NT), L_TOTAL * 4, s, a, k, key ? 1 : 0));
  if (loss3 && (mode & FM_LOSS) && (mode & FM_SAMPLE))
    hipLaunchKernelGGL(k_frame_loss, dim3(1), dim3(256), 0, s, w.kl_f, w.nll_f, (int)F, loss3);
}

void frame_backward(const Model& m, const float* P, const float* target, const float* eps, int64_t F, const Ws& w, float* G,
                    hipStream_t s) {
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
  rt().ensure_lds(reinterpret_cast<const void*>(&k_frame_bwd), L_TOTAL * 4);
  VAENPVC_TIMED("frame_bwd", s, hipLaunchKernelGGL(k_frame_bwd, dim3((unsigned)cmin_i((int)F, 1024)), dim3(NT), L_TOTAL * 4, s, a));
  // gradients of the LayerNorm parameters and conv biases of the eight normalised layers
  LnpDst d;
  const ConvL* L[8] = {&m.dec[2], &m.dec[1], &m.dec[0], &m.enc[4], &m.enc[3], &m.enc[2], &m.enc[1], &m.enc[0]};
  for (int i = 0; i < 8; ++i) {
    d.beta[i] = (int)L[i]->beta_off;
    d.gamma[i] = (int)L[i]->gamma_off;
    d.bias[i] = (int)L[i]->b_off;
  }
  hipLaunchKernelGGL(k_frame_lnp, dim3((unsigned)((3 * LNP_C + 63) / 64)), dim3(64), 0, s, w.frame_lnp, (int)F, d, G);
}

}  // namespace tuned
}  // namespace vaenpvc
