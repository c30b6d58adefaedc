/*
 * vaenpvc_debug.h -- developer hooks of libvaenpvc_hip.so (no reference counterpart).
 *
 * NOT part of the drop-in boundary (include/vaenpvc.h): a maintainer binding the ConvVAE path needs none of
 * this.  The parity tests use the selection masks to pin every kernel family against the float64 oracle at
 * small batch sizes, bench.py uses the timer to report the duration of one kernel site.  The bit assignments
 * below are per-round tuning state and may change without an ABI version bump.
 */
#ifndef VAENPVC_DEBUG_H_
#define VAENPVC_DEBUG_H_

#include "vaenpvc.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Debug/validation hook (no reference counterpart): per-step selection between the tuned
 * gfx950 kernel (bit set) and the geometry-generic kernel (bit clear) when the context
 * runs in VAENPVC_IMPL_AUTO on the VCC2016 geometry.  Bits 0..4 = encoder conv i,
 * 5 = heads, 6 = merge, 7..10 = decoder layer i; one mask for forward steps, one for
 * backward steps.  Default: all ones.  State of THIS context.
 * Bit 30 of the forward mask (default set): cleared = use the bf16-split kernels of the last decoder layer at
 * any batch size (they are selected at >= 16 frames otherwise; parity tests).  Bit 30 of the backward mask
 * (default set): cleared = launch the weight-gradient kernels on the caller's stream instead of the context's
 * helper stream (serialised kernels; used by bench.py to time single kernels).
 * Bit 29 of either mask (default set): cleared = keep the dense-shaped layers (heads, merge, encoder layer 4) on the
 * exact-fp32 MFMA kernels instead of the bf16-split plane GEMM kernels, which are selected at >= 1024 frames;
 * bit 28 (default set): cleared = select them at any batch size (parity tests).
 * Bit 27 (default set): cleared = no conv site on the view-GEMM kernels; bit 26 (default set): cleared = EVERY conv
 * site of encoder layers 1-3 / decoder layers 0-2 on the view GEMMs instead of the measured per-precision site set.
 * Bits 25 / 22 (default set): cleared = every thin / medium conv site on the fused kernels (gfx950_fconv.h,
 * gfx950_fconv_r.h) at any batch size; bit 24 of the backward mask: the thin weight gradients on gfx950_fwgrad.h;
 * bit 23: encoder layer 0 on its wave-per-frame kernels.  (The bits 22-28 exist for the parity tests, which pin every
 * kernel family against the float64 restatement at small batch sizes; defaults select by measurement.)
 * Bit 21 of either mask (default set): cleared = never use the small-batch frame kernels (gfx950_frame.h: one workgroup
 * carries one frame through a whole pass; selected up to 512 frames per call, VAENPVC_FRAME_MAX); a train step uses them
 * for both passes or for neither.  Bit 20 of the backward mask (default set): cleared = behind the frame kernels, the
 * weight gradients come from the layered kernels on two streams instead of the one-launch job list
 * (gfx950_frame_wgrad.h).  Bit 18 of the backward mask (default set): cleared = a small-batch train step keeps the 1025-tap
 * layer inside the two frame kernels instead of the two eight-workgroups-per-frame launches between them.  Bit 19 of the backward mask (default set): cleared = encoder layer 0's LayerNorm backward
 * and weight gradient as two passes instead of the fused kernel (k_enc0_bwd_wave, from 1024 frames on).  Bit 17 of the
 * backward mask (default set): cleared = the 1025-tap layer's weight gradient on the eight-wave kernel (64 x 64 wave tiles,
 * k_toep_wgrad_bf16_k32) at every batch size instead of the four-wave kernel (128 x 128 wave tiles, operands by LDS-DMA:
 * k_toep_wgrad_bf16_w4) from 4 096 frames on; bit 16 likewise for the dense-shaped weight gradients with many tiles per row
 * chunk (encoder layer 4: k_gemm_tn4 instead of k_gemm_tn).  Bit 15 of the backward mask (default set): cleared = the thin
 * layers' backward steps (decoder layers 2 and 1, encoder layer 1; environment VAENPVC_FB_LAYERS = bit set of those three) as three
 * kernels each (LayerNorm backward, input gradient, weight gradient) instead of ONE kernel per layer (csrc/gfx950_fbwd.h, from 1 024
 * frames on); bit 14 (default set): cleared = those one-kernel steps at any batch size (parity tests).
 * vaenpvc_timer_select accepts a comma-separated LIST of site tags (a kernel group timed in one pass: bench.py's roofline.sites). */
int vaenpvc_set_tuned_masks(vaenpvc_ctx* ctx, uint32_t fwd_mask, uint32_t bwd_mask);

/* Measurement hook (no reference counterpart): brackets every launch of ONE tagged
 * kernel with a hipEvent pair on the launch stream, so bench.py can report that
 * kernel's average duration over the timed region.  Tags are the kernel-site names
 * listed in DESIGN.md (e.g. "dec3_fwd").  NULL or "" disables.  State of THIS context. */
int vaenpvc_timer_select(vaenpvc_ctx* ctx, const char* tag);
/* Synchronises the recorded events (blocks the host), returns the summed milliseconds and the
 * number of launches since the last read, and resets the accumulator. */
int vaenpvc_timer_read(vaenpvc_ctx* ctx, double* total_ms, int64_t* launches);

/* Process-wide developer switches of the small-batch path (scripts/frame_prof.py, scripts/wgrad_prof.py):
 * per-phase shader clocks of the two frame kernels (environment VAENPVC_FRAME_PROF=1 selects the instrumented
 * instantiations; copies 1024 counters, returns 0 or -1 when profiling is off); a bit set of the job-list segments of
 * the one-launch weight gradient to leave in the launch; the most frame chunks of its nine chunked jobs (NULL = defaults). */
int vaenpvc_debug_frame_prof(long long* out1024);
void vaenpvc_debug_wg_segments(unsigned mask);
void vaenpvc_debug_wg_caps(const int* caps9);

#ifdef __cplusplus
}
#endif
#endif /* VAENPVC_DEBUG_H_ */
