/*
 * vaenpvc.h -- C-ABI of the MI355X-native ConvVAE hot path (libvaenpvc_hip.so).
 *
 * The reference (JeremyCCHsu/vae-npvc) has NO foreign-function interface: its hot
 * path is a TensorFlow-1 graph executed by `sess.run`.  This header is therefore the
 * boundary a maintainer would bind instead of building that graph; each entry point
 * cites the reference code whose arithmetic it replaces (paths relative to the
 * reference repository root).  See INTEGRATION.md for the ctypes stub.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer owned by the caller, contiguous,
 *     16-byte aligned; no device memory is allocated or freed behind the ABI;
 *   - all work is enqueued on the caller's `hipStream_t` (passed as void*) and is ordered
 *     with it; nothing synchronises the host (exceptions, stated at the function:
 *     vaenpvc_timer_read, vaenpvc_validate_ids).  ONE internal helper stream per context:
 *     the backward pass forks the weight-gradient kernels onto it (event fork after the
 *     gradient tensor they read is complete, event join before the call returns its work
 *     to the caller's stream), so the caller observes plain stream order.  The stream and
 *     its events are created lazily on the device that is current at the context's first
 *     launch and destroyed by vaenpvc_ctx_destroy; VAENPVC_SIDE_STREAM=0 (read at context
 *     creation) or backward-mask bit 30 cleared keeps everything on the caller's stream
 *     (the library itself does so for two-term operands from 16 384 frames per call on,
 *     where the fork measured slower; VAENPVC_SIDE_STREAM=1 forces the fork everywhere);
 *   - no mutable state outside the context: masks, precision, timer and the helper stream
 *     belong to the `vaenpvc_ctx`; entry points that take a context lock it for the
 *     duration of the call, so different contexts may be driven from different host
 *     threads (and devices) concurrently, and one context from several threads serially;
 *   - return value: 0 = ok, <0 = error (see VAENPVC_E_*); `vaenpvc_last_error()`
 *     returns a thread-local message; no C++ exception crosses the ABI;
 *   - frames are rows: x is [F, H] float32 (the reference's [F,1,H,1] NCHW tensor,
 *     analyzer.py:116-122), y is int64 [F] (analyzer.py:127), H = 513;
 *   - 1 <= F <= 262 144 (2^18) frames per call: element offsets inside the kernels are 32-bit for
 *     the largest per-frame tensor; larger batches return VAENPVC_E_ARG (split them on the host:
 *     frames are independent, gradients of the mean add with weights F_i / F).
 */
#ifndef VAENPVC_H_
#define VAENPVC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VAENPVC_MAX_LAYERS 8

#define VAENPVC_OK 0
#define VAENPVC_E_ARG (-1)       /* bad argument / malformed architecture */
#define VAENPVC_E_WORKSPACE (-2) /* workspace too small */
#define VAENPVC_E_HIP (-3)       /* a HIP runtime call failed */
#define VAENPVC_E_UNSUPPORTED (-4)
#define VAENPVC_E_STATE (-5)     /* the call needs state a preceding call of the same context did not leave (vaenpvc_train_bwd_target) */

/* implementation selector (vaenpvc_set_impl) */
#define VAENPVC_IMPL_AUTO 0 /* tuned gfx950 kernels where the geometry matches, generic otherwise */
#define VAENPVC_IMPL_GENERIC 1 /* geometry-generic HIP kernels only (cross-check path) */

/* workspace modes */
#define VAENPVC_MODE_INFER 0
#define VAENPVC_MODE_TRAIN 1

/* operand precision of the GEMM-shaped kernels on the bf16 matrix cores (vaenpvc_set_precision):
 * every fp32 operand is split into this many bf16 terms; accumulation is always fp32 */
#define VAENPVC_PREC_BF16X3 3 /* 3 terms, 6 products: fp32-exact (1e-6 of max|C|) */
#define VAENPVC_PREC_BF16X2 2 /* default: 2 terms, 3 products, 16 mantissa bits per operand: activations <= 1.4e-5,
                               * gradients <= 3.0e-5 of their tensor's scale against the float64 oracle at 32 768
                               * frames (bars 1e-4 / 2e-4; DESIGN.md section 5 on how lrelu kinks are kept out of
                               * that comparison) */
#define VAENPVC_PREC_AUTO VAENPVC_PREC_BF16X2
#define VAENPVC_PREC_BF16 1   /* plain bf16 operands (BASELINE.json config 2 "bf16"; ~1e-2) */

/* Architecture description = the keys model/vae.py actually reads from
 * architecture-*.json (model/vae.py:22-23,39,74,80-81,86,95).  Kernels are [k,1],
 * strides [s,1] (W = 1 everywhere). */
typedef struct vaenpvc_arch {
  int32_t H;     /* arch["hwc"][0] : bins per frame (513) */
  int32_t z_dim; /* arch["z_dim"] ; also the speaker-embedding width (model/vae.py:21-24) */
  int32_t y_dim; /* arch["y_dim"] : number of speakers */
  int32_t n_enc;
  int32_t enc_kernel[VAENPVC_MAX_LAYERS];
  int32_t enc_stride[VAENPVC_MAX_LAYERS];
  int32_t enc_output[VAENPVC_MAX_LAYERS];
  int32_t gen_h; /* arch["generator"]["hwc"] = [h, w(=1), c] */
  int32_t gen_c;
  int32_t n_dec;
  int32_t dec_kernel[VAENPVC_MAX_LAYERS];
  int32_t dec_stride[VAENPVC_MAX_LAYERS];
  int32_t dec_output[VAENPVC_MAX_LAYERS];
} vaenpvc_arch;

typedef struct vaenpvc_ctx vaenpvc_ctx;

/* ABI version of this header; bumped on any signature change. */
int vaenpvc_abi_version(void);
const char* vaenpvc_last_error(void);

/* Replaces ConvVAE.__init__ / _sanity_check (model/vae.py:9-39): validates the
 * architecture (AssertionError analogue = VAENPVC_E_ARG) and derives the TF 'SAME'
 * shape chain. Holds no device memory. */
int vaenpvc_ctx_create(const vaenpvc_arch* arch, vaenpvc_ctx** out);
void vaenpvc_ctx_destroy(vaenpvc_ctx* ctx);
int vaenpvc_set_impl(vaenpvc_ctx* ctx, int impl);
/* Operand precision of this context (VAENPVC_PREC_*).  The reference computes in fp32
 * (tf.float32 everywhere); BF16X2 / BF16X3 meet its 1e-4 parity bar, BF16 is the reduced
 * precision training mode. */
int vaenpvc_set_precision(vaenpvc_ctx* ctx, int planes);
int vaenpvc_get_precision(const vaenpvc_ctx* ctx);

/* Trainable-tensor table = tf.trainable_variables() of the reference in creation
 * order (model/vae.py:20-24,72-103; util/layers.py:33-64): 44 tensors for the
 * VCC2016 architecture, stored back to back in ONE flat float32 buffer in TF layouts
 * (conv [k,1,Cin,Cout]; conv_transpose [k,1,Cout,Cin]; dense [in,out]). */
int vaenpvc_param_count(const vaenpvc_ctx* ctx);
int64_t vaenpvc_param_floats(const vaenpvc_ctx* ctx);
/* name: caller buffer of name_cap bytes; shape: int64[4]; returns 0 / VAENPVC_E_ARG */
int vaenpvc_param_info(const vaenpvc_ctx* ctx, int index, char* name, int name_cap,
                       int64_t* offset_floats, int32_t* ndim, int64_t* shape);

/* Workspace (activations kept for backward, per-frame statistics, gradient
 * scratch).  `vaenpvc_ws_find` exposes named regions (float offsets into d_ws) so
 * that tests can inspect every intermediate the reference graph would hold:
 *   enc_a<i>, enc_st<i>, z_mu, z_lv, z, eps, h, dec_a<i>, dec_st<i>, xh,
 *   kl_f, nll_f, d_enc_a<i>, d_z_mu, d_z_lv, d_z, d_h, d_dec_a<i>, d_xh
 * The forward regions are written at every batch size.  The gradient regions are the
 * hand-over buffers of the backward pass: on the tuned path at >= 1024 frames several
 * of them are never written because the gradient travels as bf16 operand planes
 * (d_h, d_z_mu / d_z_lv, d_enc_a3 / d_enc_a4, d_dec_a0; regions pl_*, cl<i>) -- select
 * the generic path (vaenpvc_set_impl) to inspect every one of them.               */
int64_t vaenpvc_workspace_bytes(const vaenpvc_ctx* ctx, int64_t F, int mode);
int vaenpvc_ws_find(const vaenpvc_ctx* ctx, int64_t F, int mode, const char* name,
                    int64_t* offset_floats, int64_t* count_floats);

/* ConvVAE.encode (model/vae.py:139-141 -> _encoder 72-82; util/layers.py:47-66):
 * d_z_mu [F,z_dim]; d_z_lv may be NULL (convert.py:85 only uses z_mu). */
int vaenpvc_encode_fwd(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, int64_t F,
                       float* d_z_mu, float* d_z_lv, void* d_ws, size_t ws_bytes, void* stream);

/* ConvVAE.decode / generate (model/vae.py:143-145 -> _generator 84-103, _merge 51-61;
 * util/image.py:4-5 NHWC transpose is a no-op on memory since W = 1):
 * d_z [F,z_dim], d_y int64 [F] -> d_xh [F,H]. */
int vaenpvc_decode_fwd(vaenpvc_ctx* ctx, const float* d_params, const float* d_z,
                       const int64_t* d_y, int64_t F, float* d_xh, void* d_ws, size_t ws_bytes,
                       void* stream);

/* ConvVAE.loss (model/vae.py:106-137; util/layers.py:152-183) + the gradient half of
 * `optimizer.minimize(loss['G'])` (trainer/vae.py:19-24).
 * d_eps [F,z_dim] is the N(0,1) draw of GaussianSampleLayer, injected by the caller
 * (the reference's tf.random_normal is unseeded).
 * d_grads: flat float32 buffer, same layout as d_params, OVERWRITTEN with
 *          d loss['G'] / d params for the LOCAL mean over these F frames.
 * d_loss3: float[3] = { G, D_KL, logP } (model/vae.py:127-130). */
int vaenpvc_train_fwd_bwd(vaenpvc_ctx* ctx, const float* d_params, const float* d_x,
                          const int64_t* d_y, const float* d_eps, int64_t F, float* d_grads,
                          float* d_loss3, void* d_ws, size_t ws_bytes, void* stream);

/* Same step with the sampler's N(0,1) draw generated on the device inside the sampler kernel
 * (util/layers.py:154 tf.random_normal): Philox4x32-10 keyed by `seed`, counter words 2-3 =
 * `offset` (pass the global step so every step draws fresh noise; with data parallelism give
 * every rank its own seed).  Element e of the [F, z_dim] draw is the (e & 3)-th Box-Muller
 * normal of Philox counter (e >> 2, offset); vaenpvc_philox_normal produces the identical
 * tensor stand-alone.  The draw is kept in workspace region "eps" for the backward pass.
 * d_offset (may be NULL): device int64 added to `offset` when the kernel RUNS, so that a
 * captured hipGraph draws fresh noise on every replay (pass the device step counter that
 * vaenpvc_adam_step_dev increments). */
int vaenpvc_train_fwd_bwd_seeded(vaenpvc_ctx* ctx, const float* d_params, const float* d_x,
                                 const int64_t* d_y, uint64_t seed, uint64_t offset,
                                 const int64_t* d_offset, int64_t F, float* d_grads,
                                 float* d_loss3, void* d_ws, size_t ws_bytes, void* stream);
int vaenpvc_philox_normal(uint64_t seed, uint64_t offset, float* d_out, int64_t n, void* stream);
/* U[0,1) from the same generator (element e = top 24 bits of word e & 3 of counter (e >> 2, offset)): the
 * per-frame interpolation coefficients of the gradient penalty (tf.random_uniform in WGAN-GP critics). */
int vaenpvc_philox_uniform(uint64_t seed, uint64_t offset, float* d_out, int64_t n, void* stream);

/* The same step with the log-density evaluated against d_target [F,H] instead of d_x (the encoder still reads
 * d_x): d G / d xh becomes (xh - target) / ((1 + 1e-6) F).  With target = x + alpha (1 + 1e-6) dD(xh)/dxh
 * (vaenpvc_disc_generator_target) the 'Generator' and 'y_emb' ranges of d_grads hold the gradient of
 * l_G = -logP + alpha W_dist of the VAWGAN trainer (trainer/vae.py:141-145); d_loss3 is then meaningless. */
int vaenpvc_train_fwd_bwd_target(vaenpvc_ctx* ctx, const float* d_params, const float* d_x,
                                 const int64_t* d_y, const float* d_eps, const float* d_target, int64_t F,
                                 float* d_grads, float* d_loss3, void* d_ws, size_t ws_bytes, void* stream);
/* Backward pass only, against d_target, on the activations a preceding vaenpvc_train_fwd_bwd / _target call
 * left in d_ws: same context, parameters, x, y, eps, F and workspace, nothing else run on that workspace in
 * between.  The backward pass reads the forward tensors and writes only gradient regions, so the second gradient
 * of the VAWGAN generator step (l_G after l_E) costs one backward instead of a whole step.  The context remembers the
 * batch size, kernel selection, precision and workspace of its last train step and returns VAENPVC_E_STATE when this call
 * does not match them (the backward pass would otherwise consume stale or unwritten operands without a sign). */
int vaenpvc_train_bwd_target(vaenpvc_ctx* ctx, const float* d_params, const float* d_x, const int64_t* d_y,
                             const float* d_eps, const float* d_target, int64_t F, float* d_grads,
                             float* d_loss3, void* d_ws, size_t ws_bytes, void* stream);

/* Gradient buckets for data-parallel overlap (no reference counterpart: the reference is single
 * GPU).  The backward pass finishes the flat gradient buffer back to front -- decoder convs,
 * merge, heads, then embedding + encoder -- and each of those is ONE contiguous range of
 * d_grads.  When a callback is registered it is invoked ON THE CALLING HOST THREAD, during
 * vaenpvc_train_fwd_bwd*, once per range as soon as every kernel that writes it has been
 * enqueued; `ready_stream` is a stream on which those kernels are ordered before anything the
 * callback enqueues on it (record an event there / make the communication stream wait on it and
 * start the all-reduce of d_grads[offset, offset + count) while the rest of the backward pass
 * still runs).  bucket ids count up from 0 in call order; cb = NULL unregisters. */
typedef void (*vaenpvc_bucket_cb)(void* user, int32_t bucket, int64_t offset_floats,
                                  int64_t count_floats, void* ready_stream);
int vaenpvc_set_bucket_callback(vaenpvc_ctx* ctx, vaenpvc_bucket_cb cb, void* user);

/* Forward + losses only (VAETrainer._refresh_status fetch, trainer/vae.py:31-36). */
int vaenpvc_loss_fwd(vaenpvc_ctx* ctx, const float* d_params, const float* d_x,
                     const int64_t* d_y, const float* d_eps, int64_t F, float* d_loss3,
                     void* d_ws, size_t ws_bytes, void* stream);
int vaenpvc_loss_fwd_seeded(vaenpvc_ctx* ctx, const float* d_params, const float* d_x,
                            const int64_t* d_y, uint64_t seed, uint64_t offset, int64_t F,
                            float* d_loss3, void* d_ws, size_t ws_bytes, void* stream);

/* tf.train.AdamOptimizer apply for all trainables as ONE fused pass over the flat
 * buffers (trainer/vae.py:16-24; TF-flavour: p -= lr_t * m / (sqrt(v) + eps),
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t), t = step, 1-based).  grad_scale multiplies the
 * gradient first (1/world_size after an all-reduce SUM). */
int vaenpvc_adam_step(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n,
                      int64_t step, float lr, float beta1, float beta2, float eps,
                      float grad_scale, void* stream);

/* Same update, but the step counter lives in device memory: the kernel sequence first
 * increments *d_step (int64) and derives lr_t from it on the device, so the call can be
 * captured in a hipGraph and replayed (trainer/vae.py:94-99 hot loop without host work). */
int vaenpvc_adam_step_dev(float* d_params, const float* d_grads, float* d_m, float* d_v, int64_t n,
                          int64_t* d_step, float lr, float beta1, float beta2, float eps,
                          float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * VAWGAN branch (trainer/vae.py:115-218, architecture-vawgan-vcc2016.json).  The reference tree holds the
 * trainer and the architecture file but not the model class (README.md:3 points at another git branch), so
 * the critic is SPECIFIED in DESIGN.md section 9 from what the tree pins down: convs of
 * arch["discriminator"] built with the tree's conv + LayerNorm + lrelu block (util/layers.py:47-66,147-149),
 * one dense unit, WGAN-GP losses
 *     W_dist = mean D(x) - mean D(xh),  gp = mean_f (|dD(xi_f)/dxi_f| - 1)^2,  xi = x + t (xh - x),
 *     l_D = -W_dist + lambda gp,  l_E = -logP + D_KL,  l_G = -logP + alpha W_dist.
 * The critic owns its own flat parameter buffer (14 tensors for the VCC2016 file, names
 * Discriminator/Conv2d-<i>/{kernel,bias,layernorm.offset,layernorm.scale}, Discriminator/dense/{kernel,bias});
 * the object holds geometry only, no device memory and no mutable state. */
typedef struct vaenpvc_disc_arch {
  int32_t H; /* arch["hwc"][0] */
  int32_t n_layers;
  int32_t kernel[VAENPVC_MAX_LAYERS]; /* arch["discriminator"]["kernel"][i][0] */
  int32_t stride[VAENPVC_MAX_LAYERS];
  int32_t output[VAENPVC_MAX_LAYERS];
} vaenpvc_disc_arch;
typedef struct vaenpvc_disc vaenpvc_disc;
int vaenpvc_disc_create(const vaenpvc_disc_arch* arch, vaenpvc_disc** out);
void vaenpvc_disc_destroy(vaenpvc_disc* disc);
int vaenpvc_disc_param_count(const vaenpvc_disc* disc);
int64_t vaenpvc_disc_param_floats(const vaenpvc_disc* disc);
int vaenpvc_disc_param_info(const vaenpvc_disc* disc, int index, char* name, int name_cap,
                            int64_t* offset_floats, int32_t* ndim, int64_t* shape);
int64_t vaenpvc_disc_workspace_bytes(const vaenpvc_disc* disc, int64_t F); /* 1 <= F <= 65536 */
/* Critic values of x and xh: d_out (may be NULL) float[2F] = D(x) | D(xh); d_loss2 (may be NULL) = {W_dist, 0}. */
int vaenpvc_disc_fwd(const vaenpvc_disc* disc, const float* d_dparams, const float* d_x, const float* d_xh,
                     int64_t F, float* d_out, float* d_loss2, void* d_ws, size_t ws_bytes, void* stream);
/* `optimizer.minimize(loss['l_D'], var_list=d_vars)` gradient half (trainer/vae.py:141): d_t float[F] are the
 * interpolation coefficients; d_dgrads (critic layout) is OVERWRITTEN with d l_D / d critic parameters
 * (first and second order terms: the penalty differentiates the critic's input gradient);
 * d_loss2 = {W_dist, gp} (the trainer's status line, trainer/vae.py:196-201). */
int vaenpvc_disc_critic_fwd_bwd(const vaenpvc_disc* disc, const float* d_dparams, const float* d_x,
                                const float* d_xh, const float* d_t, int64_t F, float lambda,
                                float* d_dgrads, float* d_loss2, void* d_ws, size_t ws_bytes, void* stream);
/* Generator side of the adversarial term: d_target [F,H] = x + alpha (1 + 1e-6) dD(xh)/dxh, to be passed to
 * vaenpvc_train_fwd_bwd_target; d_loss2 (may be NULL) = {W_dist, 0}. */
int vaenpvc_disc_generator_target(const vaenpvc_disc* disc, const float* d_dparams, const float* d_x,
                                  const float* d_xh, int64_t F, float alpha, float* d_target, float* d_loss2,
                                  void* d_ws, size_t ws_bytes, void* stream);

/* Tanhize.forward_process / backward_process (analyzer.py:82-87), per bin:
 * fwd: clip((x-xmin)/(xmax-xmin),0,1)*2-1 ; bwd: (x*.5+.5)*(xmax-xmin)+xmin.
 * d_xmin/d_xmax: float32 [H]. In-place allowed. */
int vaenpvc_tanhize_fwd(const float* d_sp, const float* d_xmin, const float* d_xmax, float* d_x,
                        int64_t F, int32_t H, void* stream);
int vaenpvc_tanhize_bwd(const float* d_x, const float* d_xmin, const float* d_xmax, float* d_sp,
                        int64_t F, int32_t H, void* stream);

/* analyzer.read record slicing (analyzer.py:113-127): rows of `rec_floats` float32
 * (1029) -> x = Tanhize(row[0:H]) and y = int64(row[rec_floats-1]) (bit-exact cast). */
int vaenpvc_unpack_records(const float* d_records, int64_t F, int32_t rec_floats, int32_t H,
                           const float* d_xmin, const float* d_xmax, float* d_x, int64_t* d_y,
                           void* stream);
/* The shuffle_batch dequeue of analyzer.py:128-135 fused with the slicing: output row f is
 * built from record d_index[f] (int64, 0 <= d_index[f] < n_records; checked on the host side
 * of the binding, not here) of the HBM-resident record store. */
int vaenpvc_gather_unpack_records(const float* d_records, int64_t n_records, const int64_t* d_index,
                                  int64_t F, int32_t rec_floats, int32_t H, const float* d_xmin,
                                  const float* d_xmax, float* d_x, int64_t* d_y, void* stream);

/* Speaker ids index the embedding table (model/vae.py:89): TensorFlow raises on an id outside
 * [0, y_dim) on the CPU.  Here the kernels clamp ids (no out-of-bounds access) and this check
 * reports them: synchronises `stream`, returns VAENPVC_E_ARG if any d_y[f] is out of range.
 * d_flag: int32[1] device scratch. */
int vaenpvc_validate_ids(const vaenpvc_ctx* ctx, const int64_t* d_y, int64_t F, int32_t* d_flag,
                         void* stream);

/* tf.summary.histogram payload (model/vae.py:132-136: histograms of x and xh): d_stats =
 * double[4] {min, max, sum, sum of squares}, d_counts = uint64[n_edges + 1] with bucket b
 * counting d_edges[b-1] <= v < d_edges[b] (d_edges ascending float32, n_edges <= 2048).
 * Both outputs are ACCUMULATED into: the caller initialises them ({+inf, -inf, 0, 0}, zeros). */
int vaenpvc_summary(const float* d_data, int64_t n, const float* d_edges, int32_t n_edges,
                    double* d_stats, uint64_t* d_counts, void* stream);

/* Developer hooks (per-step kernel selection masks, the single-kernel event timer) are declared in
 * vaenpvc_debug.h: they are exported by the same library but are not part of the boundary a maintainer binds. */

#ifdef __cplusplus
}
#endif
#endif /* VAENPVC_H_ */
