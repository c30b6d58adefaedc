// abi.hip -- the extern "C" surface declared in include/vaenpvc.h (+ the developer hooks of include/vaenpvc_debug.h).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>

#include "../../include/vaenpvc_debug.h"
#include "kernels.h"

using namespace vaenpvc;

typedef std::pair<std::vector<Region>, int64_t> Layout;
struct vaenpvc_ctx {
  Model m;
  int impl = VAENPVC_IMPL_AUTO;
  Runtime rt;                // every piece of mutable state besides the layout cache (runtime.h)
  std::recursive_mutex mu;   // taken by every entry point that receives the context
  std::map<std::pair<int64_t, int>, std::shared_ptr<const Layout>> ws_cache;
};
// lock + bind the context's Runtime to this thread for the duration of an entry point
struct Call {
  std::lock_guard<std::recursive_mutex> lk;
  RtScope sc;
  explicit Call(vaenpvc_ctx* c, bool launches = true) : lk(c->mu), sc(&c->rt) {
    if (launches) c->rt.bind_device();
  }
};

static thread_local char g_err[512] = "";
static int fail(int code, const char* f, ...) {
  va_list ap;
  va_start(ap, f);
  vsnprintf(g_err, sizeof g_err, f, ap);
  va_end(ap);
  return code;
}

namespace vaenpvc {
int abi_error(int code, const char* msg) { return fail(code, "%s", msg); }
}  // namespace vaenpvc

// shared ownership: a layout handed out stays valid even if the cache is trimmed meanwhile
static std::shared_ptr<const Layout> layout_of(vaenpvc_ctx* c, int64_t F, int mode) {
  std::lock_guard<std::recursive_mutex> lk(c->mu);
  auto key = std::make_pair(F, mode);
  auto it = c->ws_cache.find(key);
  if (it == c->ws_cache.end()) {
    if (c->ws_cache.size() > 64) c->ws_cache.clear();
    int64_t total = 0;
    auto regs = workspace_layout(c->m, F, mode, &total);
    it = c->ws_cache.emplace(key, std::make_shared<const Layout>(std::move(regs), total)).first;
  }
  return it->second;
}

static int resolve(vaenpvc_ctx* c, int64_t F, int mode, void* d_ws, size_t ws_bytes, Ws* w) {
  if (F < 1) return fail(VAENPVC_E_ARG, "F must be >= 1 (got %lld)", (long long)F);
  if (F > (1LL << 18)) return fail(VAENPVC_E_ARG, "F too large (%lld > 262144 frames per call)", (long long)F);
  const auto layp = layout_of(c, F, mode);
  const Layout& lay = *layp;
  if (d_ws == nullptr || ws_bytes < (size_t)lay.second * 4)
    return fail(VAENPVC_E_WORKSPACE, "workspace too small: need %lld bytes, got %lld", (long long)lay.second * 4,
                (long long)ws_bytes);
  if (((uintptr_t)d_ws & 15) != 0) return fail(VAENPVC_E_ARG, "workspace must be 16-byte aligned");
  float* base = (float*)d_ws;
  memset(w, 0, sizeof *w);
  for (const Region& r : lay.first) {
    float* p = base + r.offset;
    const std::string& n = r.name;
    auto idx = [&](size_t pre) { return atoi(n.c_str() + pre); };
    if (n.rfind("enc_a", 0) == 0) w->enc_a[idx(5)] = p;
    else if (n.rfind("enc_st", 0) == 0) w->enc_st[idx(6)] = p;
    else if (n.rfind("dec_a", 0) == 0) w->dec_a[idx(5)] = p;
    else if (n.rfind("dec_st", 0) == 0) w->dec_st[idx(6)] = p;
    else if (n.rfind("d_enc_a", 0) == 0) w->d_enc_a[idx(7)] = p;
    else if (n.rfind("d_dec_a", 0) == 0) w->d_dec_a[idx(7)] = p;
    else if (n == "z_mu") w->z_mu = p;
    else if (n == "z_lv") w->z_lv = p;
    else if (n == "z") w->z = p;
    else if (n == "eps") w->eps = p;
    else if (n == "h") w->h = p;
    else if (n == "xh") w->xh = p;
    else if (n == "dec_y") w->dec_y = p;
    else if (n == "kl_f") w->kl_f = p;
    else if (n == "nll_f") w->nll_f = p;
    else if (n == "d_xh") w->d_xh = p;
    else if (n == "d_h") w->d_h = p;
    else if (n == "d_z") w->d_z = p;
    else if (n == "d_e") w->d_e = p;
    else if (n == "d_z_mu") w->d_z_mu = p;
    else if (n == "d_z_lv") w->d_z_lv = p;
    else if (n == "dy_tmp") w->dy_tmp = p;
    else if (n == "toep_gp") w->toep_gp = p;
    else if (n.rfind("cl", 0) == 0) w->cl[idx(2)] = p;
    else if (n == "pl_y3") w->pl_y3 = p;
    else if (n == "pl_y4") w->pl_y4 = p;
    else if (n == "pl_z") w->pl_z = p;
    else if (n == "pl_dz") w->pl_dz = p;
    else if (n == "pl_dh") w->pl_dh = p;
    else if (n == "pl_da4") w->pl_da4 = p;
    else if (n == "toep_yp") w->toep_yp = p;
    else if (n == "scratch") { w->scratch = p; w->scratch_floats = r.count; }
    else if (n == "frame_pk") w->frame_pk = p;
    else if (n == "frame_lnp") w->frame_lnp = p;
    else if (n == "frame_y") w->frame_y = p;
  }
  return 0;
}

static int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(VAENPVC_E_HIP, "%s: %s", what, hipGetErrorString(e));
  return 0;
}

static bool use_tuned(const vaenpvc_ctx* c) {
  return c->impl == VAENPVC_IMPL_AUTO && c->m.is_vcc2016 && tuned::available();
}

extern "C" {

int vaenpvc_set_tuned_masks(vaenpvc_ctx* ctx, uint32_t fwd_mask, uint32_t bwd_mask) {
  if (!ctx) return fail(VAENPVC_E_ARG, "null context");
  Call call(ctx, false);
  ctx->rt.fwd_mask = fwd_mask;
  ctx->rt.bwd_mask = bwd_mask;
  return 0;
}

int vaenpvc_set_precision(vaenpvc_ctx* ctx, int planes) {
  if (!ctx || planes < 1 || planes > 3) return fail(VAENPVC_E_ARG, "precision must be 1, 2 or 3 bf16 terms");
  Call call(ctx, false);
  ctx->rt.planes = planes;
  return 0;
}
int vaenpvc_get_precision(const vaenpvc_ctx* ctx) { return ctx ? ctx->rt.planes : VAENPVC_E_ARG; }

int vaenpvc_set_bucket_callback(vaenpvc_ctx* ctx, vaenpvc_bucket_cb cb, void* user) {
  if (!ctx) return fail(VAENPVC_E_ARG, "null context");
  Call call(ctx, false);
  ctx->rt.bucket_cb = cb;
  ctx->rt.bucket_user = user;
  return 0;
}

int vaenpvc_timer_select(vaenpvc_ctx* ctx, const char* tag) {
  if (!ctx) return fail(VAENPVC_E_ARG, "null context");
  Call call(ctx, false);
  ctx->rt.tag = tag ? tag : "";
  ctx->rt.used = 0;
  return 0;
}

int vaenpvc_timer_read(vaenpvc_ctx* ctx, double* total_ms, int64_t* launches) {
  if (!ctx) return fail(VAENPVC_E_ARG, "null context");
  Call call(ctx, false);
  Runtime& r = ctx->rt;
  double tot = 0.0;
  for (size_t i = 0; i < r.used; ++i) {
    float ms = 0.f;
    if (hipEventSynchronize(r.pool[i].second) != hipSuccess) return fail(VAENPVC_E_HIP, "timer sync");
    if (hipEventElapsedTime(&ms, r.pool[i].first, r.pool[i].second) != hipSuccess) return fail(VAENPVC_E_HIP, "timer elapsed");
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = (int64_t)r.used;
  r.used = 0;
  return 0;
}

int vaenpvc_abi_version(void) { return 3; }
const char* vaenpvc_last_error(void) { return g_err; }

int vaenpvc_ctx_create(const vaenpvc_arch* arch, vaenpvc_ctx** out) {
  if (!arch || !out) return fail(VAENPVC_E_ARG, "null argument");
  vaenpvc_ctx* c = new (std::nothrow) vaenpvc_ctx();
  if (!c) return fail(VAENPVC_E_ARG, "out of host memory");
  std::string err = build_model(*arch, &c->m);
  if (!err.empty()) {
    delete c;
    return fail(VAENPVC_E_ARG, "architecture: %s", err.c_str());
  }
  const char* env = getenv("VAENPVC_IMPL");
  if (env && strcmp(env, "generic") == 0) c->impl = VAENPVC_IMPL_GENERIC;
  c->rt.read_env();
  *out = c;
  return 0;
}

void vaenpvc_ctx_destroy(vaenpvc_ctx* ctx) {
  if (!ctx) return;
  {
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    int cur = -1;
    if (ctx->rt.device >= 0 && hipGetDevice(&cur) == hipSuccess) {
      if (cur != ctx->rt.device) (void)hipSetDevice(ctx->rt.device);
      ctx->rt.release();
      if (cur != ctx->rt.device) (void)hipSetDevice(cur);
    }
  }
  delete ctx;
}

int vaenpvc_set_impl(vaenpvc_ctx* ctx, int impl) {
  if (!ctx || (impl != VAENPVC_IMPL_AUTO && impl != VAENPVC_IMPL_GENERIC)) return fail(VAENPVC_E_ARG, "bad impl");
  Call call(ctx, false);
  ctx->impl = impl;
  return 0;
}

int vaenpvc_param_count(const vaenpvc_ctx* ctx) { return ctx ? (int)ctx->m.table.size() : VAENPVC_E_ARG; }
int64_t vaenpvc_param_floats(const vaenpvc_ctx* ctx) { return ctx ? ctx->m.n_params : VAENPVC_E_ARG; }

int vaenpvc_param_info(const vaenpvc_ctx* ctx, int index, char* name, int name_cap, int64_t* offset_floats,
                       int32_t* ndim, int64_t* shape) {
  if (!ctx || index < 0 || index >= (int)ctx->m.table.size()) return fail(VAENPVC_E_ARG, "bad parameter index");
  const ParamInfo& p = ctx->m.table[index];
  if (name && name_cap > 0) {
    strncpy(name, p.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (offset_floats) *offset_floats = p.offset;
  if (ndim) *ndim = p.ndim;
  if (shape)
    for (int i = 0; i < 4; ++i) shape[i] = p.shape[i];
  return 0;
}

int64_t vaenpvc_workspace_bytes(const vaenpvc_ctx* ctx, int64_t F, int mode) {
  if (!ctx || F < 1 || (mode != VAENPVC_MODE_INFER && mode != VAENPVC_MODE_TRAIN)) return fail(VAENPVC_E_ARG, "bad argument");
  return layout_of(const_cast<vaenpvc_ctx*>(ctx), F, mode)->second * 4;
}

int vaenpvc_ws_find(const vaenpvc_ctx* ctx, int64_t F, int mode, const char* name, int64_t* offset_floats,
                    int64_t* count_floats) {
  if (!ctx || !name || F < 1) return fail(VAENPVC_E_ARG, "bad argument");
  const auto lay = layout_of(const_cast<vaenpvc_ctx*>(ctx), F, mode);
  for (const Region& r : lay->first)
    if (r.name == name) {
      if (offset_floats) *offset_floats = r.offset;
      if (count_floats) *count_floats = r.count;
      return 0;
    }
  return fail(VAENPVC_E_ARG, "no workspace region named '%s'", name);
}

int vaenpvc_encode_fwd(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, int64_t F, float* d_z_mu,
                       float* d_z_lv, void* d_ws, size_t ws_bytes, void* stream) {
  if (!ctx || !d_params || !d_x || !d_z_mu) return fail(VAENPVC_E_ARG, "null argument");
  Call call(ctx);
  Ws w;
  int rc = resolve(ctx, F, VAENPVC_MODE_INFER, d_ws, ws_bytes, &w);
  if (rc) return rc;
  ctx->rt.last_F = -1;   // activations / operand planes of a preceding train step may be overwritten: vaenpvc_train_bwd_target must not re-use them
  hipStream_t s = (hipStream_t)stream;
  if (use_tuned(ctx) && tuned::frame_fwd_on(F)) {
    tuned::frame_pack(ctx->m, d_params, w, nullptr, nullptr, 0, s);
    tuned::frame_forward(ctx->m, d_params, d_x, nullptr, nullptr, nullptr, nullptr, nullptr, F, w, nullptr, tuned::FRAME_ENC, nullptr, s);
  } else if (use_tuned(ctx)) tuned::encoder_fwd(ctx->m, d_params, d_x, F, w, s);
  else generic::encoder_fwd(ctx->m, d_params, d_x, F, w, s);
  size_t nb = (size_t)F * ctx->m.z * 4;
  if (hipMemcpyAsync(d_z_mu, w.z_mu, nb, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(VAENPVC_E_HIP, "copy z_mu");
  if (d_z_lv && hipMemcpyAsync(d_z_lv, w.z_lv, nb, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(VAENPVC_E_HIP, "copy z_lv");
  return check_launch("encode_fwd");
}

int vaenpvc_decode_fwd(vaenpvc_ctx* ctx, const float* d_params, const float* d_z, const int64_t* d_y, int64_t F,
                       float* d_xh, void* d_ws, size_t ws_bytes, void* stream) {
  if (!ctx || !d_params || !d_z || !d_y || !d_xh) return fail(VAENPVC_E_ARG, "null argument");
  Call call(ctx);
  Ws w;
  int rc = resolve(ctx, F, VAENPVC_MODE_INFER, d_ws, ws_bytes, &w);
  if (rc) return rc;
  ctx->rt.last_F = -1;   // activations / operand planes of a preceding train step may be overwritten: vaenpvc_train_bwd_target must not re-use them
  hipStream_t s = (hipStream_t)stream;
  if (use_tuned(ctx) && tuned::frame_fwd_on(F)) {
    tuned::frame_pack(ctx->m, d_params, w, nullptr, nullptr, 0, s);
    tuned::frame_forward(ctx->m, d_params, nullptr, nullptr, d_y, nullptr, nullptr, d_z, F, w, d_xh, tuned::FRAME_DEC, nullptr, s);
  } else if (use_tuned(ctx)) tuned::decoder_fwd(ctx->m, d_params, d_z, d_y, F, w, d_xh, s);
  else generic::decoder_fwd(ctx->m, d_params, d_z, d_y, F, w, d_xh, s);
  return check_launch("decode_fwd");
}

static int fwd_all(vaenpvc_ctx* ctx, const float* P, const float* x, const int64_t* y, const float* eps,
                   const PhiloxKey* key, int64_t F, const Ws& w, bool want_grad, float* loss3, hipStream_t s,
                   const float* target = nullptr, bool allow_frame = true, float* zero_g = nullptr) {
  if (allow_frame && use_tuned(ctx) && tuned::frame_fwd_on(F)) {
    // small batch: one workgroup per frame carries it through the whole forward pass (gfx950_frame.h); the launch that
    // packs the weights also zero-fills what the backward pass accumulates into
    int nz2 = 0;
    float* z2 = zero_g ? tuned::frame_zero_region(w, &nz2) : nullptr;
    tuned::frame_pack(ctx->m, P, w, zero_g, z2, nz2, s);
    tuned::frame_forward(ctx->m, P, x, target, y, eps, key, nullptr, F, w, nullptr,
                         tuned::FRAME_ENC | tuned::FRAME_SAMPLE | tuned::FRAME_DEC | tuned::FRAME_LOSS | (want_grad ? tuned::FRAME_GRAD : 0),
                         loss3, s);
    return 0;
  }
  if (use_tuned(ctx)) tuned::encoder_fwd(ctx->m, P, x, F, w, s);
  else generic::encoder_fwd(ctx->m, P, x, F, w, s);
  rt().plz_F = -1;
  VAENPVC_TIMED("reparam", s, (use_tuned(ctx) && tuned::reparam_fwd_planes(ctx->m, eps, key, F, w, s)) ? (void)0 : generic::reparam_fwd(ctx->m, eps, key, F, w, s));
  if (use_tuned(ctx)) tuned::decoder_fwd(ctx->m, P, w.z, y, F, w, w.xh, s, /*weights_packed=*/true);
  else generic::decoder_fwd(ctx->m, P, w.z, y, F, w, w.xh, s);
  rt().dxh_post_F = -1;
  const bool fused_loss = want_grad && use_tuned(ctx);
  VAENPVC_TIMED("loss", s, (fused_loss && tuned::loss_fwd_post(ctx->m, P, target ? target : x, F, w, loss3, s))
                               ? (void)0 : generic::loss_fwd(ctx->m, target ? target : x, F, w, want_grad, loss3, s));   // (the density's data argument)
  return 0;
}

static PhiloxKey make_key(uint64_t seed, uint64_t offset, const int64_t* d_offset = nullptr) {
  return PhiloxKey{(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)offset, (uint32_t)(offset >> 32), d_offset};
}

static int loss_impl(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, const int64_t* d_y, const float* d_eps,
                     const PhiloxKey* key, int64_t F, float* d_loss3, void* d_ws, size_t ws_bytes, void* stream) {
  if (!ctx || !d_params || !d_x || !d_y || (!d_eps && !key) || !d_loss3) return fail(VAENPVC_E_ARG, "null argument");
  Call call(ctx);
  Ws w;
  int rc = resolve(ctx, F, VAENPVC_MODE_INFER, d_ws, ws_bytes, &w);
  if (rc) return rc;
  ctx->rt.last_F = -1;   // activations / operand planes of a preceding train step may be overwritten: vaenpvc_train_bwd_target must not re-use them
  fwd_all(ctx, d_params, d_x, d_y, d_eps, key, F, w, false, d_loss3, (hipStream_t)stream);
  return check_launch("loss_fwd");
}

int vaenpvc_loss_fwd(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, const int64_t* d_y,
                     const float* d_eps, int64_t F, float* d_loss3, void* d_ws, size_t ws_bytes, void* stream) {
  return loss_impl(ctx, d_params, d_x, d_y, d_eps, nullptr, F, d_loss3, d_ws, ws_bytes, stream);
}
int vaenpvc_loss_fwd_seeded(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, const int64_t* d_y,
                            uint64_t seed, uint64_t offset, int64_t F, float* d_loss3, void* d_ws, size_t ws_bytes,
                            void* stream) {
  PhiloxKey k = make_key(seed, offset);
  return loss_impl(ctx, d_params, d_x, d_y, nullptr, &k, F, d_loss3, d_ws, ws_bytes, stream);
}

static int train_impl(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, const int64_t* d_y,
                      const float* d_eps, const PhiloxKey* key, int64_t F, float* d_grads, float* d_loss3, void* d_ws,
                      size_t ws_bytes, void* stream, const float* d_target = nullptr) {
  if (!ctx || !d_params || !d_x || !d_y || (!d_eps && !key) || !d_grads || !d_loss3) return fail(VAENPVC_E_ARG, "null argument");
  Call call(ctx);
  Ws w;
  int rc = resolve(ctx, F, VAENPVC_MODE_TRAIN, d_ws, ws_bytes, &w);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  // small batches: both passes on the whole-frame kernels, or neither (the layered backward pass reads operand copies
  // only the layered forward pass leaves behind)
  const bool frame = use_tuned(ctx) && tuned::frame_fwd_on(F) && tuned::frame_bwd_on(F);
  fwd_all(ctx, d_params, d_x, d_y, d_eps, key, F, w, true, d_loss3, s, d_target, frame, frame ? d_grads : nullptr);
  {  // what vaenpvc_train_bwd_target may re-use
    Runtime& r = ctx->rt;
    r.last_F = F;
    r.last_path = frame ? 2 : use_tuned(ctx) ? 1 : 0;
    r.last_planes = r.planes;
    r.last_fwd_mask = r.fwd_mask;
    r.last_bwd_mask = r.bwd_mask;
    r.last_ws = d_ws;
  }
  const float* eps_bwd = key ? w.eps : d_eps;   // (the seeded sampler stored its draw in the workspace)
  ctx->rt.bucket_next = 0;
  if (frame) {
    tuned::backward_frame(ctx->m, d_params, d_x, d_target, d_y, eps_bwd, F, w, d_grads, s, /*g_zeroed=*/true, d_loss3);
  } else if (use_tuned(ctx)) {
    tuned::backward(ctx->m, d_params, d_x, d_y, eps_bwd, F, w, d_grads, s);
  } else {
    generic::backward(ctx->m, d_params, d_x, d_y, eps_bwd, F, w, d_grads, s);
    if (ctx->rt.bucket_cb)  // the generic path finishes everything at once: one bucket
      ctx->rt.bucket_cb(ctx->rt.bucket_user, 0, 0, ctx->m.n_params, (void*)s);
  }
  return check_launch("train_fwd_bwd");
}

int vaenpvc_train_fwd_bwd(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, const int64_t* d_y,
                          const float* d_eps, int64_t F, float* d_grads, float* d_loss3, void* d_ws,
                          size_t ws_bytes, void* stream) {
  return train_impl(ctx, d_params, d_x, d_y, d_eps, nullptr, F, d_grads, d_loss3, d_ws, ws_bytes, stream);
}
int vaenpvc_train_fwd_bwd_seeded(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, const int64_t* d_y,
                                 uint64_t seed, uint64_t offset, const int64_t* d_offset, int64_t F, float* d_grads,
                                 float* d_loss3, void* d_ws, size_t ws_bytes, void* stream) {
  PhiloxKey k = make_key(seed, offset, d_offset);
  return train_impl(ctx, d_params, d_x, d_y, nullptr, &k, F, d_grads, d_loss3, d_ws, ws_bytes, stream);
}

int vaenpvc_train_fwd_bwd_target(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, const int64_t* d_y,
                                 const float* d_eps, const float* d_target, int64_t F, float* d_grads,
                                 float* d_loss3, void* d_ws, size_t ws_bytes, void* stream) {
  if (!d_target) return fail(VAENPVC_E_ARG, "null argument");
  return train_impl(ctx, d_params, d_x, d_y, d_eps, nullptr, F, d_grads, d_loss3, d_ws, ws_bytes, stream, d_target);
}

int vaenpvc_train_bwd_target(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, const int64_t* d_y,
                             const float* d_eps, const float* d_target, int64_t F, float* d_grads, float* d_loss3,
                             void* d_ws, size_t ws_bytes, void* stream) {
  if (!ctx || !d_params || !d_x || !d_y || !d_eps || !d_target || !d_grads || !d_loss3) return fail(VAENPVC_E_ARG, "null argument");
  Call call(ctx);
  Ws w;
  int rc = resolve(ctx, F, VAENPVC_MODE_TRAIN, d_ws, ws_bytes, &w);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  {
    // the backward pass consumes what the preceding train step of THIS context left in THIS workspace (activations, packed
    // weights, operand planes): same batch size, same kernel family, same masks and precision -- anything else would read
    // stale or unwritten operands without a sign (round-3 advisor)
    const Runtime& r = ctx->rt;
    const int path = (use_tuned(ctx) && tuned::frame_fwd_on(F) && tuned::frame_bwd_on(F)) ? 2 : use_tuned(ctx) ? 1 : 0;
    if (r.last_F != F || r.last_path != path || r.last_planes != r.planes || r.last_fwd_mask != r.fwd_mask ||
        r.last_bwd_mask != r.bwd_mask || r.last_ws != d_ws)
      return fail(VAENPVC_E_STATE, "train_bwd_target: no matching train step precedes it in this context (batch size, kernel selection, precision or workspace differ)");
  }
  rt().dxh_post_F = -1;
  generic::loss_fwd(ctx->m, d_target, F, w, true, d_loss3, s);   // new d(xh) from the activations already in place
  ctx->rt.bucket_next = 0;
  if (use_tuned(ctx) && tuned::frame_fwd_on(F) && tuned::frame_bwd_on(F)) {
    int nz2 = 0;
    float* z2 = tuned::frame_zero_region(w, &nz2);
    tuned::frame_pack(ctx->m, d_params, w, d_grads, z2, nz2, s, /*zero_only=*/true);   // (the packed weights of the preceding train step are still in the workspace: parameters may not change in between, include/vaenpvc.h)
    tuned::backward_frame(ctx->m, d_params, d_x, d_target, d_y, d_eps, F, w, d_grads, s, /*g_zeroed=*/true);
  } else if (use_tuned(ctx)) {
    tuned::backward(ctx->m, d_params, d_x, d_y, d_eps, F, w, d_grads, s);
  } else {
    generic::backward(ctx->m, d_params, d_x, d_y, d_eps, F, w, d_grads, s);
    if (ctx->rt.bucket_cb) ctx->rt.bucket_cb(ctx->rt.bucket_user, 0, 0, ctx->m.n_params, (void*)s);
  }
  return check_launch("train_bwd_target");
}

int vaenpvc_philox_normal(uint64_t seed, uint64_t offset, float* d_out, int64_t n, void* stream) {
  if (!d_out || n < 1) return fail(VAENPVC_E_ARG, "bad argument");
  launch_philox_normal(d_out, n, make_key(seed, offset), (hipStream_t)stream);
  return check_launch("philox_normal");
}
int vaenpvc_philox_uniform(uint64_t seed, uint64_t offset, float* d_out, int64_t n, void* stream) {
  if (!d_out || n < 1) return fail(VAENPVC_E_ARG, "bad argument");
  launch_philox_uniform(d_out, n, make_key(seed, offset), (hipStream_t)stream);
  return check_launch("philox_uniform");
}

int vaenpvc_adam_step(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n, int64_t step,
                      float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  if (!d_params || !d_grads || !d_m || !d_v || n < 1 || step < 1) return fail(VAENPVC_E_ARG, "bad argument");
  double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)step)) /
                (1.0 - std::pow((double)beta1, (double)step));
  launch_adam(d_params, d_grads, d_m, d_v, n, (float)lr_t, beta1, beta2, eps, grad_scale, (hipStream_t)stream);
  return check_launch("adam_step");
}

int vaenpvc_adam_step_dev(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n, int64_t* d_step,
                          float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  if (!d_params || !d_grads || !d_m || !d_v || !d_step || n < 1) return fail(VAENPVC_E_ARG, "bad argument");
  launch_adam_dev(d_params, d_grads, d_m, d_v, n, d_step, lr, beta1, beta2, eps, grad_scale, (hipStream_t)stream);
  return check_launch("adam_step_dev");
}

int vaenpvc_tanhize_fwd(const float* d_sp, const float* d_xmin, const float* d_xmax, float* d_x, int64_t F,
                        int32_t H, void* stream) {
  if (!d_sp || !d_xmin || !d_xmax || !d_x || F < 1 || H < 1) return fail(VAENPVC_E_ARG, "bad argument");
  launch_tanhize(d_sp, d_xmin, d_xmax, d_x, F, H, true, (hipStream_t)stream);
  return check_launch("tanhize_fwd");
}

int vaenpvc_tanhize_bwd(const float* d_x, const float* d_xmin, const float* d_xmax, float* d_sp, int64_t F,
                        int32_t H, void* stream) {
  if (!d_x || !d_xmin || !d_xmax || !d_sp || F < 1 || H < 1) return fail(VAENPVC_E_ARG, "bad argument");
  launch_tanhize(d_x, d_xmin, d_xmax, d_sp, F, H, false, (hipStream_t)stream);
  return check_launch("tanhize_bwd");
}

int vaenpvc_unpack_records(const float* d_records, int64_t F, int32_t rec_floats, int32_t H, const float* d_xmin,
                           const float* d_xmax, float* d_x, int64_t* d_y, void* stream) {
  if (!d_records || !d_xmin || !d_xmax || !d_x || !d_y || F < 1 || H < 1 || rec_floats < H + 1)
    return fail(VAENPVC_E_ARG, "bad argument");
  launch_unpack(d_records, nullptr, F, rec_floats, H, d_xmin, d_xmax, d_x, d_y, (hipStream_t)stream);
  return check_launch("unpack_records");
}

int vaenpvc_gather_unpack_records(const float* d_records, int64_t n_records, const int64_t* d_index, int64_t F,
                                  int32_t rec_floats, int32_t H, const float* d_xmin, const float* d_xmax, float* d_x,
                                  int64_t* d_y, void* stream) {
  if (!d_records || !d_index || !d_xmin || !d_xmax || !d_x || !d_y || F < 1 || n_records < 1 || H < 1 || rec_floats < H + 1)
    return fail(VAENPVC_E_ARG, "bad argument");
  launch_unpack(d_records, d_index, F, rec_floats, H, d_xmin, d_xmax, d_x, d_y, (hipStream_t)stream);
  return check_launch("gather_unpack_records");
}

int vaenpvc_validate_ids(const vaenpvc_ctx* ctx, const int64_t* d_y, int64_t F, int32_t* d_flag, void* stream) {
  if (!ctx || !d_y || !d_flag || F < 1) return fail(VAENPVC_E_ARG, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  int32_t bad = 0;
  launch_check_ids(d_y, F, ctx->m.ny, d_flag, s);
  if (hipMemcpyAsync(&bad, d_flag, sizeof bad, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return fail(VAENPVC_E_HIP, "validate_ids: copy");
  if (bad) return fail(VAENPVC_E_ARG, "%d speaker id(s) outside [0, %d)", (int)bad, ctx->m.ny);
  return check_launch("validate_ids");
}

int vaenpvc_summary(const float* d_data, int64_t n, const float* d_edges, int32_t n_edges, double* d_stats,
                    uint64_t* d_counts, void* stream) {
  if (!d_data || !d_edges || !d_stats || !d_counts || n < 1 || n_edges < 1 || n_edges > 2048)
    return fail(VAENPVC_E_ARG, "bad argument");
  launch_summary(d_data, n, d_edges, n_edges, d_stats, reinterpret_cast<unsigned long long*>(d_counts), (hipStream_t)stream);
  return check_launch("summary");
}

}  // extern "C"
