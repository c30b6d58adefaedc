// kernels.h -- host-callable launchers shared between the ABI layer and the kernel
// translation units.  Everything enqueues on `stream`; nothing synchronises.
#pragma once
#include <hip/hip_runtime.h>

#include "model.h"
#include "philox.h"
#include "runtime.h"

namespace vaenpvc {

struct Ws {  // resolved workspace pointers (see workspace_layout)
  float* enc_a[VAENPVC_MAX_LAYERS];
  float* enc_st[VAENPVC_MAX_LAYERS];
  float *z_mu, *z_lv, *z, *h;
  float* eps;  // N(0,1) draw of the sampler when it is generated on the device (seeded entry points)
  float* dec_a[VAENPVC_MAX_LAYERS];
  float* dec_st[VAENPVC_MAX_LAYERS];
  float *xh, *kl_f, *nll_f;
  float* dec_y;  // lrelu(LN(dec_a[n_dec-2]))
  // train only
  float* d_xh;
  float* d_dec_a[VAENPVC_MAX_LAYERS];   // d_dec_a[0] is BORROWED from the fused loss kernel to the start of the backward pass (loss_fwd_post parks
                                        // the edge-term parts of the last layer's weight gradient there; k_sum_parts_add consumes them before
                                        // anything writes d(a0)); valid while Runtime::dxh_post_F == F
  float *d_h, *d_z, *d_e, *d_z_mu, *d_z_lv;
  float* d_enc_a[VAENPVC_MAX_LAYERS];
  float* dy_tmp;   // gradient hand-over buffer of the layered backward pass; BORROWED the same way: column 512 of the last layer's
                   // input gradient waits here from the fused loss kernel until dec3_dgrad has written the other 512 columns
  float *pl_y3, *pl_y4, *pl_z, *pl_dz, *pl_dh, *pl_da4;
  float* cl[12];  // channel-last planes of the conv view GEMMs (cl_layout.h: CL_*)
  float* toep_gp;  // bf16 planes of d_xh
  float* toep_yp;  // bf16 planes of dec_y
  float* scratch;
  int64_t scratch_floats;
  float* frame_pk;   // packed weight copies of the small-batch frame kernels (gfx950_frame.h: Pk)
  float* frame_lnp;  // per-frame channel sums of their LayerNorm backward
  float* frame_y;    // activated layer outputs [F][12000] for their weight-gradient launch (train mode)
};

// sets the thread-local message vaenpvc_last_error() returns; returns `code` (abi.hip)
int abi_error(int code, const char* msg);

// ---- single-kernel event timer (state in the context's Runtime, runtime.h) --------
#define VAENPVC_TIMED(tag, stream, ...)               \
  do {                                                \
    ::vaenpvc::Runtime& _rt = ::vaenpvc::rt();        \
    bool _tm = _rt.timer_match(tag);                  \
    if (_tm) _rt.timer_begin(stream);                 \
    __VA_ARGS__;                                      \
    if (_tm) _rt.timer_end(stream);                   \
  } while (0)

// ---- geometry-generic HIP kernels (generic_kernels.hip) -------------------------
namespace generic {
void encoder_fwd(const Model& m, const float* P, const float* x, int64_t F, const Ws& w, hipStream_t s);
// z = z_mu + eps*sqrt(exp(z_lv)) (eps may be null -> z = z_mu); also per-frame KL
// (key != nullptr: eps is drawn on the device with Philox and stored in w.eps for the backward pass)
void reparam_fwd(const Model& m, const float* eps, const PhiloxKey* key, int64_t F, const Ws& w, hipStream_t s);
// decoder from z (w.z or external) and y -> xh_out
void decoder_fwd(const Model& m, const float* P, const float* z, const int64_t* y, int64_t F, const Ws& w,
                 float* xh_out, hipStream_t s);
// per-frame log-density, optional d_xh, then {G, D_KL, logP}
void loss_fwd(const Model& m, const float* x, int64_t F, const Ws& w, bool want_grad, float* loss3, hipStream_t s);
void loss_reduce(int64_t F, const Ws& w, float* loss3, hipStream_t s);   // batch means of kl_f / nll_f -> {G, D_KL, logP}
void backward(const Model& m, const float* P, const float* x, const int64_t* y, const float* eps, int64_t F,
              const Ws& w, float* G, hipStream_t s);
// per-step entry points (the tuned path can fall back to any of them, per layer)
void enc_layer_fwd(const Model& m, const float* P, const float* x, int64_t F, const Ws& w, hipStream_t s, int i);
void heads_fwd(const Model& m, const float* P, int64_t F, const Ws& w, hipStream_t s);
void merge_fwd(const Model& m, const float* P, const float* z, const int64_t* y, int64_t F, const Ws& w, hipStream_t s);
void dec_layer_fwd(const Model& m, const float* P, int64_t F, const Ws& w, float* xh_out, hipStream_t s, int i);
void bias_grad(const float* d, float* db, int64_t F, int C, int H, hipStream_t s);
void bwd_dec_layer(const Model& m, const float* P, int64_t F, const Ws& w, float* G, hipStream_t s, int i);
void bwd_merge(const Model& m, const float* P, const int64_t* y, int64_t F, const Ws& w, float* G, hipStream_t s);
void bwd_reparam(const Model& m, const float* eps, int64_t F, const Ws& w, hipStream_t s);
void bwd_heads(const Model& m, const float* P, int64_t F, const Ws& w, float* G, hipStream_t s);
void bwd_enc_layer(const Model& m, const float* P, const float* x, int64_t F, const Ws& w, float* G, hipStream_t s, int i);
}  // namespace generic

// ---- elementwise / optimiser kernels (misc_kernels.hip) -------------------------
void launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float b1, float b2,
                 float eps, float gscale, hipStream_t s);
void launch_adam_dev(float* p, const float* g, float* m, float* v, int64_t n, int64_t* d_step, float lr, float b1,
                     float b2, float eps, float gscale, hipStream_t s);
void launch_tanhize(const float* in, const float* xmin, const float* xmax, float* out, int64_t F, int H,
                    bool forward, hipStream_t s);
// rows gathered through `idx` (int64 record numbers) when it is not null
void launch_unpack(const float* rec, const int64_t* idx, int64_t F, int rec_floats, int H, const float* xmin,
                   const float* xmax, float* x, int64_t* y, hipStream_t s);
void launch_check_ids(const int64_t* y, int64_t F, int ny, int* flag, hipStream_t s);
void launch_philox_normal(float* out, int64_t n, PhiloxKey key, hipStream_t s);
void launch_philox_uniform(float* out, int64_t n, PhiloxKey key, hipStream_t s);
// min, max, sum, sum of squares (double[4]) and counts per bucket (uint64[n_edges + 1]; bucket b holds
// edges[b-1] <= v < edges[b]) of `n` floats; the outputs must be zeroed by the caller (stats[0..1] = +-inf)
void launch_summary(const float* d, int64_t n, const float* edges, int n_edges, double* stats, unsigned long long* counts,
                    hipStream_t s);

// ---- tuned gfx950 kernels for the VCC2016 geometry (gfx950_*.hip) ----------------
namespace tuned {
// step masks: bit set = use the tuned kernel for that step, clear = generic kernel.
//   forward  bits: 0..4 encoder conv i | 5 heads | 6 merge | 7..10 decoder layer i
//   backward bits: 0..4 encoder conv i | 5 heads | 6 merge | 7..10 decoder layer i
bool available();
void encoder_fwd(const Model& m, const float* P, const float* x, int64_t F, const Ws& w, hipStream_t s);
// `weights_packed`: the packed weight copies in w.scratch are already current (the encoder of the
// same step built them); a stand-alone decode call packs them itself.
void decoder_fwd(const Model& m, const float* P, const float* z, const int64_t* y, int64_t F, const Ws& w,
                 float* xh_out, hipStream_t s, bool weights_packed = false);
void backward(const Model& m, const float* P, const float* x, const int64_t* y, const float* eps, int64_t F,
              const Ws& w, float* G, hipStream_t s);
// the loss of a train step fused with the first pass of the backward over d(xh) (planes of d(xh), column 512 of the last layer's
// input gradient, bias-gradient parts): true = done, the backward pass that follows picks the results up (Runtime::dxh_post_F);
// false = not selected at this batch size / precision / masks, the caller runs generic::loss_fwd
// sampler + KL writing z also as the planes of the merge GEMM (true = done: the decoder_fwd that follows on w.z skips its split pass,
// Runtime::plz_F; false = not selected, the caller runs generic::reparam_fwd)
bool reparam_fwd_planes(const Model& m, const float* eps, const PhiloxKey* key, int64_t F, const Ws& w, hipStream_t s);
bool loss_fwd_post(const Model& m, const float* P, const float* x, int64_t F, const Ws& w, float* loss3, hipStream_t s);
// ---- small-batch path: whole frames per workgroup (gfx950_frame.hip)
constexpr int FRAME_ENC = 1, FRAME_SAMPLE = 2, FRAME_DEC = 4, FRAME_LOSS = 8, FRAME_GRAD = 16;   // = frame::FM_*
constexpr int64_t FRAME_CAP = 1024;          // largest batch the workspace reserves its per-frame sums for
int64_t frame_pack_floats();
int64_t frame_lnp_floats(int64_t F);
bool frame_fwd_on(int64_t F);
bool frame_bwd_on(int64_t F);
// packed weight copies; G_zero / zero2 (nullable): buffers zero-filled by the same launch
void frame_pack(const Model& m, const float* P, const Ws& w, float* G_zero, float* zero2, int nzero2, hipStream_t s, bool zero_only = false);
// scratch region the small-batch backward wants zeroed before it runs (per-speaker sums of the merge backward)
float* frame_zero_region(const Ws& w, int* count);
void frame_forward(const Model& m, const float* P, const float* x, const float* target, const int64_t* y, const float* eps,
                   const PhiloxKey* key, const float* z_in, int64_t F, const Ws& w, float* xh_out, int mode, float* loss3,
                   hipStream_t s);
// input-gradient chain + LayerNorm parameter / conv-bias gradients (the weight gradients are the caller's)
// (lnp_sums: also launch the reduction of the LayerNorm parameter / conv-bias sums; false when frame_wgrad follows)
// (loss3: where the batch means go when the step's forward pass left them to the backward half -- the split train step,
//  gfx950_frame.hip: frame_split_on; null: nobody wants them)
void frame_backward(const Model& m, const float* P, const float* target, const float* eps, int64_t F, const Ws& w, float* G,
                    hipStream_t s, bool lnp_sums, float* loss3 = nullptr);
bool frame_split_on(const Ws& w, int64_t F);
// every parameter gradient in one launch (gfx950_frame_wgrad.h); G must have been zero-filled
void frame_wgrad(const Model& m, const float* P, const float* x, const int64_t* y, int64_t F, const Ws& w, float* G, hipStream_t s);
// the whole backward pass of a small batch: frame_backward + every weight gradient (gfx950_layers.hip)
// (g_zeroed: the gradient buffer and frame_zero_region were zero-filled by this step's frame_pack)
void backward_frame(const Model& m, const float* P, const float* x, const float* target, const int64_t* y, const float* eps,
                    int64_t F, const Ws& w, float* G, hipStream_t s, bool g_zeroed, float* loss3 = nullptr);
}  // namespace tuned

}  // namespace vaenpvc
