// disc.hip -- the critic ("Discriminator") of the VAWGAN branch: forward, the WGAN-GP critic step with its
// double backward, and the input gradient the generator step needs.  trainer/vae.py:115-218 is the consumer
// in the reference; the model is specified in DESIGN.md section 9 (the reference tree does not hold it).
//
// The branch trains 16-frame batches (architecture-vawgan-vcc2016.json:33), 48 frames through three small
// convolutions.  At that size the device time of a step is (number of kernels) x ~6.5 us plus what the three conv
// kernels take, so: one thread per output with a frame's layer input / output staged in LDS, lanes along the
// contiguous weight dimension, several loads in flight per thread (one wave per SIMD: latency-bound loops); the
// LayerNorm statistics are taken by the consumer while it stages the frame; every per-channel reduction, every
// LayerNorm parameter gradient and every sum of weight-gradient copies of a step runs in ONE launch at the end.
// Deterministic: no atomics, every gradient element is accumulated by exactly one thread, pass after pass on one
// stream.  Conventions follow generic_kernels.hip: frames-major [B, C, H] float32, the PRE-LN conv output `u` plus
// per-frame (mean, rstd) is what is kept, consumers apply lrelu(LN(u)) on load.
//
// Critic step for F frames, B = 3F rows (x | xh | xi = x + t (xh - x)):
//   pass 1  forward of all rows                                  -> d[B]
//   pass 2  backward of sum_f d_f w.r.t. the INPUT, rows xi only -> g_f, keeps abar_l (gradient at the
//           activations) and ubar_l (gradient at the pre-LN outputs)
//           gp_f = (|g_f| - 1)^2 ,  gt_f = (2 lambda / F)(|g_f| - 1) g_f / |g_f|
//   pass 3  adjoint of pass 2, bottom-up, rows xi only: q_l = conv_l(adjoint of abar_{l-1}); the conv input
//           gradient is bilinear in (W_l, ubar_l), so dW_l += wgrad(adjoint of abar_{l-1}, ubar_l); the
//           LayerNorm backward is linear in its upstream (adjoint = the same operator applied to q) and
//           depends on u through xhat and rstd (adjoint `udir_l`, injected into pass 4)
//   pass 4  ordinary backward of all rows with upstream -1/F (x), +1/F (xh), 0 (xi) plus udir_l on rows xi.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"
#include "disc_frame.h"

namespace vaenpvc {
namespace disc {

#define LN_EPS 1e-5f
#define LEAK 0.02f
#define EPSILON 1e-6f

struct G {
  int cin, hin, cout, hout, k, s, pad;
};
struct Act {  // LN-on-load descriptor; st == nullptr -> identity
  const float* st;
  const float* gamma;
  const float* beta;
};
static const Act kNoAct{nullptr, nullptr, nullptr};
// The gradient at the LAST conv layer's activated output is never a tensor: the dense unit d = w . a + c hands down
// coef(row) * w[k] (coef: rows [0, F0) c0, [F0, 2 F0) c1, the rest c2; row = row0 + the kernel's local row).  The kernels
// that would read it as `dy` take this descriptor instead when their dy pointer is null.
struct VDy {
  const float* w;
  float c0, c1, c2;
  int64_t F0, row0;
  __device__ __forceinline__ float at(int64_t f, int i) const {
    const int64_t r = row0 + f;
    return (r < F0 ? c0 : (r < 2 * F0 ? c1 : c2)) * w[i];
  }
};
static const VDy kNoVDy{nullptr, 0.f, 0.f, 0.f, 0, 0};

__device__ __forceinline__ float lnact(float v, const Act& a, int64_t f, int c) {
  if (a.st == nullptr) return v;
  float n = (v - a.st[2 * f]) * a.st[2 * f + 1] * a.gamma[c] + a.beta[c];
  return fmaxf(n, LEAK * n);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// all threads receive the block total; blockDim.x a multiple of 64, <= 1024
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sm[i];
  return r;
}
__device__ __forceinline__ int ch_of(int k, int hlast) { return k / hlast; }
// Cooperative copy global -> (functor) with BATCH independent loads in flight per thread: with one wave per SIMD a plain
// `for (e = tid; e < n; e += 256) dst[e] = src[e]` waits for every load before the next one is issued.
template <int BATCH, class Put>
__device__ __forceinline__ void stage_copy(const float* __restrict__ src, int n, Put&& put) {
  for (int e0 = threadIdx.x; e0 < n; e0 += BATCH * blockDim.x) {
    float v[BATCH];
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      const int e = e0 + b * blockDim.x;
      v[b] = e < n ? src[e] : 0.f;
    }
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      const int e = e0 + b * blockDim.x;
      if (e < n) put(e, v[b]);
    }
  }
}
static inline dim3 grid1(int64_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

// ------------------------------------------------------------------------------------------- forward
// rows [x | xh | xi]: xi_f = x_f + t_f (xh_f - x_f); t == nullptr: two groups only
// (w1t != nullptr: the launch also leaves layer 1's kernel as [t][o][c] for the input-gradient tile of disc_frame.h)
__global__ void k_rows(const float* __restrict__ x, const float* __restrict__ xh, const float* __restrict__ t,
                       float* __restrict__ rows, int64_t F, int H, const float* __restrict__ W1 = nullptr,
                       float* __restrict__ w1t = nullptr, float* __restrict__ zero = nullptr, int64_t nzero = 0) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < nzero) zero[idx] = 0.f;      // (the critic step's gradient buffer: no separate fill launch)
  if (w1t && idx < 7 * 16 * 32) {
    const int c = (int)idx % 16, o = ((int)idx / 16) % 32, tt = (int)idx / 512;
    w1t[idx] = W1[(tt * 16 + c) * 32 + o];
  }
  if (idx >= F * H) return;
  float a = x[idx], b = xh[idx];
  rows[idx] = a;
  rows[F * H + idx] = b;
  if (t) rows[2 * F * H + idx] = a + t[idx / H] * (b - a);
}

// util/layers.py:56-64 : out[f,o,j] = b[o] + sum_c sum_t W[t,c,o] * act(in)[f,c,s*j-pad+t]   (b may be null)
// One workgroup per (frame, block of OB output channels): the frame's activated input is staged in LDS once,
// lanes take consecutive output channels (weight reads coalesced, input reads broadcast).  With four or more input
// channels the four waves take every fourth channel each and their partial sums meet in LDS (the loop is bound by
// load latency with one wave per SIMD, so the channel loop is what is worth spreading).
constexpr int FWD_OB = 8;
__global__ void __launch_bounds__(256) k_conv_fwd(const float* __restrict__ in, Act ai, const float* __restrict__ W,
                                                  const float* __restrict__ b, float* __restrict__ out, G g,
                                                  float* __restrict__ st_new) {
  extern __shared__ float sIn[];  // [cin][hin], then [4 outputs][4 waves][64] partial sums
  __shared__ float red[16];
  float* sPart = sIn + g.cin * g.hin;
  const int64_t f = blockIdx.x;
  const int o0 = blockIdx.y * FWD_OB, ob = min(FWD_OB, g.cout - o0);
  const int nin = g.cin * g.hin;
  if (st_new) {
    // the LayerNorm statistics of the input tensor (the layer in front) are taken HERE, from the staged frame (every
    // workgroup of the frame computes them, the first one stores them for the backward passes): no statistics kernel
    float sm = 0.f;
    stage_copy<12>(in + f * nin, nin, [&](int e, float v) {
      sIn[e] = v;
      sm += v;
    });
    const float mean = block_sum(sm, red) / nin;
    float q = 0.f;
    for (int e = threadIdx.x; e < nin; e += blockDim.x) {
      const float d = sIn[e] - mean;
      q += d * d;
    }
    const float rstd = 1.0f / sqrtf(block_sum(q, red) / nin + LN_EPS);
    if (blockIdx.y == 0 && threadIdx.x == 0) {
      st_new[2 * f] = mean;
      st_new[2 * f + 1] = rstd;
    }
    for (int e = threadIdx.x; e < nin; e += blockDim.x) {
      const int c = e / g.hin;
      const float n = (sIn[e] - mean) * rstd * ai.gamma[c] + ai.beta[c];
      sIn[e] = fmaxf(n, LEAK * n);
    }
  } else {
    stage_copy<12>(in + f * nin, nin, [&](int e, float v) { sIn[e] = lnact(v, ai, f, e / g.hin); });
  }
  __syncthreads();
  const int cs = g.cin >= 4 ? 4 : 1;                      // channel split over the waves
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cpart = cs == 4 ? wave : 0, nloc = cs == 4 ? 64 : 256, loc = cs == 4 ? lane : (int)threadIdx.x;
  const int64_t ws = (int64_t)g.cin * g.cout;
  // OV output channels per item: four when the rows of cout weights can be read 16 bytes at a time (one load then
  // feeds four chains), else one
  const bool v4 = (g.cout & 3) == 0 && (ob & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
  const int OV = v4 ? 4 : 1;
  const int og = ob / OV, items = og * g.hout;
  for (int base = 0; base < items; base += nloc) {       // uniform trip count: barriers inside
    const int idx = base + loc;
    const bool on = idx < items;
    const int o = o0 + OV * (on ? idx % og : 0), j = on ? idx / og : 0;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (on) {
      const int t0 = max(0, g.pad - g.s * j), t1 = min(g.k, g.hin + g.pad - g.s * j);
      for (int c = cpart; c < g.cin; c += cs) {
        const float* row = sIn + c * g.hin + (g.s * j - g.pad);
        const float* wp = W + (int64_t)c * g.cout + o;
        int t = t0;
        if (v4) {
          for (; t + 4 <= t1; t += 4) {                  // four 16-byte weight loads in flight
            float4 wv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wv[q] = *reinterpret_cast<const float4*>(wp + (t + q) * ws);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x = row[t + q];
              acc[0] += x * wv[q].x;
              acc[1] += x * wv[q].y;
              acc[2] += x * wv[q].z;
              acc[3] += x * wv[q].w;
            }
          }
          for (; t < t1; ++t) {
            const float4 w4 = *reinterpret_cast<const float4*>(wp + t * ws);
            const float x = row[t];
            acc[0] += x * w4.x;
            acc[1] += x * w4.y;
            acc[2] += x * w4.z;
            acc[3] += x * w4.w;
          }
        } else {
          float a0 = 0.f, a1 = 0.f;                      // eight weight loads in flight, two chains
          for (; t + 8 <= t1; t += 8) {
            float wv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) wv[q] = wp[(t + q) * ws];
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
              a0 += row[t + q] * wv[q];
              a1 += row[t + q + 1] * wv[q + 1];
            }
          }
          for (; t < t1; ++t) a0 += row[t] * wp[t * ws];
          acc[0] += a0 + a1;
        }
      }
    }
    if (cs == 4) {
#pragma unroll
      for (int v = 0; v < 4; ++v)
        if (v < OV) sPart[(v * 4 + wave) * 64 + lane] = acc[v];
      __syncthreads();
      if (wave == 0 && on) {
        for (int v = 0; v < OV; ++v) {
          const float* sp = sPart + v * 256 + lane;
          out[(f * g.cout + o + v) * g.hout + j] = ((sp[0] + sp[64]) + (sp[128] + sp[192])) + (b ? b[o + v] : 0.f);
        }
      }
      __syncthreads();
    } else if (on) {
      for (int v = 0; v < OV; ++v) out[(f * g.cout + o + v) * g.hout + j] = acc[v] + (b ? b[o + v] : 0.f);
    }
  }
}

// d[f] = c + sum_k act(u)[f,k] w[k]     (flatten is C-major = memory order)
// (dw != nullptr, critic step: also the dense unit's own gradient dw[k] += coef(f) act(u)[f,k], dc += coef(f), coef as in VDy;
//  fp32 atomics into the zeroed gradient buffer -- the separate pass over act(u) this replaces cost a launch)
__global__ void k_dense_fwd(const float* __restrict__ u, Act ai, int hlast, const float* __restrict__ w,
                            const float* __restrict__ c, float* __restrict__ d, int flat, float* __restrict__ st_new,
                            float* __restrict__ dw = nullptr, float* __restrict__ dc = nullptr, int64_t F0 = 0, float c0 = 0.f,
                            float c1 = 0.f, float c2 = 0.f) {
  __shared__ float sm[16];
  int64_t f = blockIdx.x;
  const float coef = f < F0 ? c0 : (f < 2 * F0 ? c1 : c2);
  float mean = 0.f, rstd = 0.f;
  if (st_new) {   // statistics of the last conv layer, taken here (see k_conv_fwd)
    float a = 0.f;
    for (int k = threadIdx.x; k < flat; k += blockDim.x) a += u[f * flat + k];
    mean = block_sum(a, sm) / flat;
    float q = 0.f;
    for (int k = threadIdx.x; k < flat; k += blockDim.x) {
      const float dv = u[f * flat + k] - mean;
      q += dv * dv;
    }
    rstd = 1.0f / sqrtf(block_sum(q, sm) / flat + LN_EPS);
    if (threadIdx.x == 0) {
      st_new[2 * f] = mean;
      st_new[2 * f + 1] = rstd;
    }
  }
  float s = 0.f;
  for (int k = threadIdx.x; k < flat; k += blockDim.x) {
    float v;
    if (st_new) {
      const int ch = k / hlast;
      const float n = (u[f * flat + k] - mean) * rstd * ai.gamma[ch] + ai.beta[ch];
      v = fmaxf(n, LEAK * n);
    } else {
      v = lnact(u[f * flat + k], ai, f, ch_of(k, hlast));
    }
    s += v * w[k];
    if (dw && coef != 0.f) atomicAdd(dw + k, coef * v);
  }
  s = block_sum(s, sm);
  if (threadIdx.x == 0) {
    d[f] = s + c[0];
    // dc = sum_f coef(f): one thread, closed form (the +-1/F terms cancel exactly; per-row atomics left a rounding residue)
    if (dc && f == 0) dc[0] += (float)F0 * c0 + (float)F0 * c1 + (float)((int64_t)gridDim.x - 2 * F0) * c2;
  }
}

// ------------------------------------------------------------------------------------------ backward
// LayerNorm + lrelu backward (autodiff of util/layers.py:32-44,149); one block per frame.
//   n = gamma xhat + beta ; p = dy lrelu'(n) gamma ; du = rstd (p - mean p - xhat mean(p xhat)) (+ add, a tensor whose
//   row 0 belongs to frame add_row0)
__global__ void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ u, const float* __restrict__ st,
                         const float* __restrict__ gamma, const float* __restrict__ beta,
                         const float* __restrict__ add, int64_t add_row0, float* __restrict__ du, int C, int H, VDy vd = kNoVDy) {
  __shared__ float sm[16];
  int64_t f = blockIdx.x;
  int n = C * H;
  float mean = st[2 * f], rstd = st[2 * f + 1];
  add = (add && f >= add_row0) ? add + (f - add_row0) * n - f * n : nullptr;   // rows [add_row0, ..) carry the injected term
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int c = i / H;
    float xh = (u[f * n + i] - mean) * rstd;
    float nn = xh * gamma[c] + beta[c];
    float p = (dy ? dy[f * n + i] : vd.at(f, i)) * (nn >= 0.f ? 1.0f : LEAK) * gamma[c];
    s1 += p;
    s2 += p * xh;
  }
  s1 = block_sum(s1, sm) / n;
  s2 = block_sum(s2, sm) / n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int c = i / H;
    float xh = (u[f * n + i] - mean) * rstd;
    float nn = xh * gamma[c] + beta[c];
    float p = (dy ? dy[f * n + i] : vd.at(f, i)) * (nn >= 0.f ? 1.0f : LEAK) * gamma[c];
    float r = rstd * (p - s1 - xh * s2);
    du[f * n + i] = add ? r + add[f * n + i] : r;
  }
}

// Adjoint of k_ln_bwd (one block per frame).  Inputs: q = adjoint of its result, dy = its upstream (abar),
// u / st / gamma / beta of the layer.  Outputs:
//   at   = adjoint of dy            = P(q) gamma lrelu'(n)         with P(v) = rstd (v - mean v - xhat mean(v xhat))
//   udir = adjoint of u             = rstd (xt - mean xt - xhat mean(q ubar + xt xhat)),
//                                     xt = -rstd (mean(p xhat) q + mean(q xhat) p),  ubar = P(p)
//   pn   = P(q) dy lrelu'(n)        (its per-channel sum is the adjoint of gamma)
// lrelu'' = 0 almost everywhere, so n contributes nothing.
__global__ void k_ln_bwd_bwd(const float* __restrict__ q, const float* __restrict__ dy, const float* __restrict__ u,
                             const float* __restrict__ st, const float* __restrict__ gamma,
                             const float* __restrict__ beta, float* __restrict__ at, float* __restrict__ udir,
                             float* __restrict__ pn, int C, int H, VDy vd = kNoVDy) {
  __shared__ float sm[16];
  int64_t f = blockIdx.x;
  int n = C * H;
  float mean = st[2 * f], rstd = st[2 * f + 1];
  float sp = 0.f, spx = 0.f, sq = 0.f, sqx = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int c = i / H;
    float xh = (u[f * n + i] - mean) * rstd;
    float nn = xh * gamma[c] + beta[c];
    float p = (dy ? dy[f * n + i] : vd.at(f, i)) * (nn >= 0.f ? 1.0f : LEAK) * gamma[c];
    float qq = q[f * n + i];
    sp += p;
    spx += p * xh;
    sq += qq;
    sqx += qq * xh;
  }
  sp = block_sum(sp, sm) / n;
  spx = block_sum(spx, sm) / n;
  sq = block_sum(sq, sm) / n;
  sqx = block_sum(sqx, sm) / n;
  float sx = 0.f, ss = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int c = i / H;
    float xh = (u[f * n + i] - mean) * rstd;
    float nn = xh * gamma[c] + beta[c];
    float p = (dy ? dy[f * n + i] : vd.at(f, i)) * (nn >= 0.f ? 1.0f : LEAK) * gamma[c];
    float qq = q[f * n + i];
    float ub = rstd * (p - sp - xh * spx);
    float xt = -rstd * (spx * qq + sqx * p);
    sx += xt;
    ss += qq * ub + xt * xh;
  }
  sx = block_sum(sx, sm) / n;
  ss = block_sum(ss, sm) / n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int c = i / H;
    float xh = (u[f * n + i] - mean) * rstd;
    float nn = xh * gamma[c] + beta[c];
    float sl = nn >= 0.f ? 1.0f : LEAK;
    float dyv = dy ? dy[f * n + i] : vd.at(f, i);
    float p = dyv * sl * gamma[c];
    float qq = q[f * n + i];
    float xt = -rstd * (spx * qq + sqx * p);
    float pt = rstd * (qq - sq - xh * sqx);
    at[f * n + i] = pt * gamma[c] * sl;
    udir[f * n + i] = rstd * (xt - sx - xh * ss);
    pn[f * n + i] = pt * dyv * sl;
  }
}

// dgamma[c] += sum_{f,h} dn xhat ; dbeta[c] += sum_{f,h} dn ; one block per (layer, channel): blockIdx.y = layer entry
struct ParamGrad {
  const float *dy, *u, *st, *gamma, *beta;      // (dy null: vd)
  float *dgamma, *dbeta;
  int64_t B;
  int C, H;
  VDy vd;
};
struct ParamGrads {
  ParamGrad e[VAENPVC_MAX_LAYERS];
  int count;
};
__global__ void k_ln_param_grad(ParamGrads pg) {
  __shared__ float sm[16];
  const ParamGrad p = pg.e[blockIdx.y];
  const int c = blockIdx.x;
  if (c >= p.C) return;   // (uniform per block)
  float g = p.gamma[c], b = p.beta[c];
  float sg = 0.f, sb = 0.f;
  for (int64_t i = threadIdx.x; i < p.B * p.H; i += blockDim.x) {
    int64_t f = i / p.H;
    int64_t e = (f * p.C + c) * p.H + (int)(i % p.H);
    float xh = (p.u[e] - p.st[2 * f]) * p.st[2 * f + 1];
    float nn = xh * g + b;
    float dn = (p.dy ? p.dy[e] : p.vd.at(f, c * p.H + (int)(i % p.H))) * (nn >= 0.f ? 1.0f : LEAK);
    sg += dn * xh;
    sb += dn;
  }
  sg = block_sum(sg, sm);
  sb = block_sum(sb, sm);
  if (threadIdx.x == 0) {
    p.dgamma[c] += sg;
    p.dbeta[c] += sb;
  }
}

// db[o] += sum_{f,h} d[f,o,h], one block per channel,
// for up to 2 * 8 + 1 (tensor, destination) pairs in one launch: blockIdx.y = pair, blockIdx.x = channel.
// Destinations are distinct tensors (no two pairs share one).
struct ChanSum {
  const float* d;
  float* db;
  int64_t B;
  int C, H;
};
struct ChanSums {
  ChanSum e[2 * VAENPVC_MAX_LAYERS + 1];
  int count;
};
__global__ void k_chan_sum_multi(ChanSums cs) {
  __shared__ float sm[16];
  const ChanSum p = cs.e[blockIdx.y];
  const int o = blockIdx.x;
  if (o >= p.C) return;   // (uniform per block)
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < p.B * p.H; i += blockDim.x) s += p.d[((i / p.H) * p.C + o) * p.H + (int)(i % p.H)];
  s = block_sum(s, sm);
  if (threadIdx.x == 0) p.db[o] += s;
}

// din[f,c,i] = sum_o sum_t W[t,c,o] dout[f,o,j], s*j - pad + t = i       (conv input gradient)
// One workgroup per (frame, block of CB input channels): the frame's dout is staged in LDS, every item walks the
// (at most ceil(k/s)) output positions that reach it and the contiguous cout weights of each.
constexpr int BWD_CB = 4;
__global__ void __launch_bounds__(256) k_conv_bwd_data(const float* __restrict__ dout, const float* __restrict__ W,
                                                       float* __restrict__ din, G g) {
  extern __shared__ float sD[];  // [cout][hout]
  const int64_t f = blockIdx.x;
  const int c0 = blockIdx.y * BWD_CB, cb = min(BWD_CB, g.cin - c0);
  stage_copy<12>(dout + f * g.cout * g.hout, g.cout * g.hout, [&](int e, float v) { sD[e] = v; });
  __syncthreads();
  for (int idx = threadIdx.x; idx < cb * g.hin; idx += blockDim.x) {
    const int c = c0 + idx % cb, i = idx / cb;
    float acc = 0.f;
    // t = i + pad - s j in [0, k)
    const int jlo = max(0, (i + g.pad - g.k + g.s) / g.s), jhi = min(g.hout - 1, (i + g.pad) / g.s);
    for (int j = jlo; j <= jhi; ++j) {
      const int t = i + g.pad - g.s * j;
      const float* wr = W + ((int64_t)t * g.cin + c) * g.cout;
      const float* dc = sD + j;
      if ((g.cout & 3) == 0) {  // rows of cout floats are 16-byte aligned: four weights per load, four chains
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 8
        for (int o = 0; o < g.cout; o += 4) {
          const float4 w4 = *reinterpret_cast<const float4*>(wr + o);
          a0 += w4.x * dc[o * g.hout];
          a1 += w4.y * dc[(o + 1) * g.hout];
          a2 += w4.z * dc[(o + 2) * g.hout];
          a3 += w4.w * dc[(o + 3) * g.hout];
        }
        acc += (a0 + a1) + (a2 + a3);
      } else {
        for (int o = 0; o < g.cout; ++o) acc += wr[o] * dc[o * g.hout];
      }
    }
    din[(f * g.cin + c) * g.hin + i] = acc;
  }
}

// dW[t,c,o] += sum_f sum_j act(in)[f,c,s*j-pad+t] dout[f,o,j]
// One workgroup per (input channel c, chunk of WG_TC taps); thread = (output channel, tap subset), its weights in
// registers over the whole frame loop; per frame the input row of channel c and dout are staged in LDS
// (dout rows have odd length in every layer of the VCC2016 file: conflict-free across output channels).
constexpr int WG_TC = 16, WG_MAXT = 16;
__global__ void __launch_bounds__(256) k_conv_bwd_w(const float* __restrict__ in, Act ai, const float* __restrict__ dout,
                                                    float* __restrict__ dW, int64_t B, G g, int FB, int64_t fper,
                                                    int accumulate) {
  extern __shared__ float sm[];
  const int nD = g.cout * g.hout, nF = nD + g.hin;  // per staged frame: dout [cout][hout], then the input row [hin]
  const int c = blockIdx.x, tbase = blockIdx.y * WG_TC;
  const int o = threadIdx.x % g.cout, tq = threadIdx.x / g.cout, ntq = blockDim.x / g.cout;  // (cout <= 256)
  float acc[WG_MAXT];
#pragma unroll
  for (int m = 0; m < WG_MAXT; ++m) acc[m] = 0.f;
  const bool live = tq < ntq;
  // blockIdx.z owns frames [z fper, (z + 1) fper) and its own copy of dW (summed afterwards in a fixed order)
  const int64_t fbeg = blockIdx.z * fper, fend = min(B, fbeg + fper);
  dW += (int64_t)blockIdx.z * g.k * g.cin * g.cout;
  for (int64_t f0 = fbeg; f0 < fend; f0 += FB) {  // FB frames per barrier pair: the staging latency is paid once for all
    const int nf = (int)min((int64_t)FB, fend - f0);
    __syncthreads();
    // (the nf frames of dout are contiguous in memory: one batched copy; the LDS frame pitch nF differs from nD)
    stage_copy<16>(dout + f0 * nD, nf * nD, [&](int e, float v) {
      const int ff = e / nD;
      sm[ff * nF + (e - ff * nD)] = v;
    });
    for (int ff = 0; ff < nf; ++ff) {
      const float* isrc = in + ((f0 + ff) * g.cin + c) * g.hin;
      stage_copy<4>(isrc, g.hin, [&](int e, float v) { sm[ff * nF + nD + e] = lnact(v, ai, f0 + ff, c); });
    }
    __syncthreads();
    if (live) {
#pragma unroll
      for (int m = 0; m < WG_MAXT; ++m) {
        const int t = tbase + tq + m * ntq;
        if (tq + m * ntq < WG_TC && t < g.k) {
          const int hi = g.hin - 1 + g.pad - t;  // s*j <= hi
          const int j0 = max(0, (g.pad - t + g.s - 1) / g.s), j1 = hi < 0 ? 0 : min(g.hout, hi / g.s + 1);
          float a0 = 0.f, a1 = 0.f;
          for (int ff = 0; ff < nf; ++ff) {
            const float* dr = sm + ff * nF + o * g.hout;
            const float* sRow = sm + ff * nF + nD - g.pad + t;
            int j = j0;
            for (; j + 8 <= j1; j += 8) {  // one wave per SIMD: eight LDS read pairs in flight or the loop is latency
              float x[8], d[8];
#pragma unroll
              for (int q = 0; q < 8; ++q) {
                x[q] = sRow[g.s * (j + q)];
                d[q] = dr[j + q];
              }
#pragma unroll
              for (int q = 0; q < 8; q += 2) {
                a0 += x[q] * d[q];
                a1 += x[q + 1] * d[q + 1];
              }
            }
            for (; j < j1; ++j) a0 += sRow[g.s * j] * dr[j];
          }
          acc[m] += a0 + a1;
        }
      }
    }
  }
  if (live) {
#pragma unroll
    for (int m = 0; m < WG_MAXT; ++m) {
      const int t = tbase + tq + m * ntq;
      if (tq + m * ntq < WG_TC && t < g.k) {
        float* p = dW + ((int64_t)t * g.cin + c) * g.cout + o;
        *p = accumulate ? *p + acc[m] : acc[m];
      }
    }
  }
}

// dW[i] += sum_z part[z][i], z ascending, for up to 8 (weight gradient, copies) pairs in one launch
struct PartSum {
  const float* part;
  float* dW;
  int n, nz;
};
struct PartSums {
  PartSum e[8];
  int count;
};
// (a thread walks the pairs in order: two pairs of the same layer -- passes 3 and 4 -- update element i from the same thread)
__global__ void k_sum_parts(PartSums ps, int nmax) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nmax; i += gridDim.x * blockDim.x) {
    for (int e = 0; e < ps.count; ++e) {
      const PartSum& p = ps.e[e];
      if (i >= p.n) continue;
      float s = 0.f;
      for (int z = 0; z < p.nz; ++z) s += p.part[(int64_t)z * p.n + i];
      p.dW[i] += s;
    }
  }
}

// ---------------------------------------------------------------------------------- dense-like layers
// A conv layer whose every output position sees the whole input (k >= 2 hin - 1 taps around it: the critic's 115-tap
// layer on 57 positions) IS a dense layer [cin hin] -> [cout hout] with the weight matrix
//     Wd[(c, i)][(o, j)] = W[i - s j + pad][c][o]
// expanded once per call (k_expand_dense).  Forward, input gradient and weight gradient are then plain GEMMs on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32; operand maps A: lane l holds A[i = l & 31][k = l >> 5], B: lane l
// holds B[k = l >> 5][j = l & 31], C: 16 registers, row = (reg & 3) + 8 (reg >> 2) + 4 (l >> 5), column = l & 31);
// the weight gradient is computed against Wd and folded back over the positions (k_fold_dense).  The reduction
// dimension is dealt to several workgroups (blockIdx.z) writing private results that k_mm_reduce adds in order.
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct MmArgs {
  const float* A;     // A(m, r) = A[m * a_m + r * a_r]
  int64_t a_m, a_r;
  const float* B;     // B(r, n) = B[r * b_r + n * b_n]
  int64_t b_r, b_n;
  float* C;           // split z writes the row-major [M][N] matrix at C + z * M * N
  int M, N, R, rper;  // rper: reduction elements per split (a multiple of 32)
};
// 64 x 64 tile per workgroup, four waves of 32 x 32, reduction chunks of 32 staged through LDS (pitches 33 / 65: odd).
// A_FAST_R / B_FAST_N: which index is contiguous in memory -- the staging lanes run along it.
template <bool A_FAST_R, bool B_FAST_N>
__global__ void __launch_bounds__(256) k_mm(MmArgs a) {
  __shared__ float sA[64 * 33];
  __shared__ float sB[32 * 65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int r_lo = blockIdx.z * a.rper, r_hi = min(a.R, r_lo + a.rper);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int rc = r_lo; rc < r_hi; rc += 32) {
    float va[8], vb[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q;
      const int ma = A_FAST_R ? e >> 5 : e & 63, ka = A_FAST_R ? e & 31 : e >> 6;
      const int gm = m0 + ma, gra = rc + ka;
      va[q] = (gm < a.M && gra < r_hi) ? a.A[gm * a.a_m + gra * a.a_r] : 0.f;
      const int nb = B_FAST_N ? e & 63 : e >> 5, kb = B_FAST_N ? e >> 6 : e & 31;
      const int gn = n0 + nb, grb = rc + kb;
      vb[q] = (gn < a.N && grb < r_hi) ? a.B[grb * a.b_r + gn * a.b_n] : 0.f;
    }
    __syncthreads();   // the previous chunk is consumed
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int e = tid + 256 * q;
      const int ma = A_FAST_R ? e >> 5 : e & 63, ka = A_FAST_R ? e & 31 : e >> 6;
      sA[ma * 33 + ka] = va[q];
      const int nb = B_FAST_N ? e & 63 : e >> 5, kb = B_FAST_N ? e >> 6 : e & 31;
      sB[kb * 65 + nb] = vb[q];
    }
    __syncthreads();
    const float* pa = sA + (wm * 32 + l31) * 33 + lh;
    const float* pb = sB + lh * 65 + wn * 32 + l31;
#pragma unroll
    for (int st = 0; st < 16; ++st) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * st], pb[2 * st * 65], acc, 0, 0, 0);
  }
  float* C = a.C + (int64_t)blockIdx.z * a.M * a.N;
  const int col = n0 + wn * 32 + l31;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
    if (row < a.M && col < a.N) C[(int64_t)row * a.N + col] = acc[reg];
  }
}
// out[i] = sum_z part[z][i] (+ bias[(i % N) / hdiv]), z ascending
__global__ void k_mm_reduce(const float* __restrict__ part, int nz, int64_t mn, float* __restrict__ out,
                            const float* __restrict__ bias, int N, int hdiv) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= mn) return;
  float s = 0.f;
  for (int z = 0; z < nz; ++z) s += part[z * mn + i];
  if (bias) s += bias[(int)(i % N) / hdiv];
  out[i] = s;
}
// Wd[(c, i)][(o, j)] = W[i - s j + pad][c][o]  (zero where the tap index leaves the kernel)
__global__ void k_expand_dense(const float* __restrict__ W, float* __restrict__ Wd, G g) {
  const int N = g.cout * g.hout;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)g.cin * g.hin * N) return;
  const int n = (int)(idx % N), kk = (int)(idx / N);
  const int o = n / g.hout, j = n - o * g.hout, c = kk / g.hin, i = kk - c * g.hin;
  const int t = i - g.s * j + g.pad;
  Wd[idx] = (t >= 0 && t < g.k) ? W[((int64_t)t * g.cin + c) * g.cout + o] : 0.f;
}
// dW[t][c][o] += sum_j dWd[(c, t + s j - pad)][(o, j)]
__global__ void k_fold_dense(const float* __restrict__ dWd, float* __restrict__ dW, G g) {
  const int N = g.cout * g.hout;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)g.k * g.cin * g.cout) return;
  const int o = (int)(idx % g.cout), c = (int)((idx / g.cout) % g.cin), t = (int)(idx / ((int64_t)g.cout * g.cin));
  float s = 0.f;
  for (int j = 0; j < g.hout; ++j) {
    const int i = t + g.s * j - g.pad;
    if (i >= 0 && i < g.hin) s += dWd[((int64_t)c * g.hin + i) * N + o * g.hout + j];
  }
  dW[idx] += s;
}
// LayerNorm statistics of u (one block per frame, stored) and its activated copy a = lrelu(LN(u)): the A operand of a
// dense-like layer's GEMMs
__global__ void k_act_stats(const float* __restrict__ u, const float* __restrict__ gamma, const float* __restrict__ beta,
                            float* __restrict__ st, float* __restrict__ aout, int C, int H) {
  __shared__ float sm[16];
  const int64_t f = blockIdx.x;
  const int n = C * H;
  const float* p = u + f * n;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
  const float mean = block_sum(s, sm) / n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float d = p[i] - mean;
    q += d * d;
  }
  const float rstd = 1.0f / sqrtf(block_sum(q, sm) / n + LN_EPS);
  if (threadIdx.x == 0) {
    st[2 * f] = mean;
    st[2 * f + 1] = rstd;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int c = i / H;
    const float v = (p[i] - mean) * rstd * gamma[c] + beta[c];
    aout[f * n + i] = fmaxf(v, LEAK * v);
  }
}

__global__ void k_zero(float* __restrict__ p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

// per-frame gradient norm, penalty and the adjoint of g: gt = coef (|g| - 1) g / |g|   (coef = 2 lambda / F)
__global__ void k_gp(const float* __restrict__ g, float* __restrict__ gt, float* __restrict__ gp_f, int H, float coef) {
  __shared__ float sm[16];
  int64_t f = blockIdx.x;
  float s = 0.f;
  for (int i = threadIdx.x; i < H; i += blockDim.x) s += g[f * H + i] * g[f * H + i];
  float nrm = sqrtf(block_sum(s, sm));
  float k = coef * (nrm - 1.0f) / nrm;
  for (int i = threadIdx.x; i < H; i += blockDim.x) gt[f * H + i] = k * g[f * H + i];
  if (threadIdx.x == 0) gp_f[f] = (nrm - 1.0f) * (nrm - 1.0f);
}

// loss2 = { W_dist = mean d[0:F) - mean d[F:2F) , gp = mean gp_f (0 when gp_f is null) }; single block
__global__ void k_losses(const float* __restrict__ d, const float* __restrict__ gp_f, int64_t F, float* __restrict__ loss2) {
  __shared__ float sm[16];
  float a = 0.f, b = 0.f;
  for (int64_t i = threadIdx.x; i < F; i += blockDim.x) {
    a += d[i] - d[F + i];
    if (gp_f) b += gp_f[i];
  }
  a = block_sum(a, sm);
  b = block_sum(b, sm);
  if (threadIdx.x == 0) {
    loss2[0] = a / (float)F;
    loss2[1] = b / (float)F;
  }
}

// target of the generator step: x' = x + alpha (1 + 1e-6) dD(xh)/dxh, so that the reconstruction gradient
// (xh - x') / ((1 + 1e-6) F) of the ConvVAE backward equals d(-logP + alpha W_dist)/dxh
__global__ void k_adv_target(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out, int64_t n,
                             float alpha) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) out[idx] = x[idx] + alpha * (1.0f + EPSILON) * g[idx];
}

}  // namespace disc
}  // namespace vaenpvc

// ============================================================================================ host
using namespace vaenpvc;
using namespace vaenpvc::disc;

// every parameter gradient of the two thin conv layers (disc_frame.h): a workgroup finds its job from its block index
__global__ void __launch_bounds__(frame::WT) k_critic_front_wgrad(front::CwArgs a, front::CwPlan pl) {
  extern __shared__ __attribute__((aligned(16))) float cw_lds[];
  tuned::WRunner run;
  front::critic_front_wgrad_block(run, cw_lds, a, pl, (int)blockIdx.x);
}

// one pass segment of the two thin conv layers, one workgroup per row (disc_frame.h)
template <int MODE>
__global__ void __launch_bounds__(frame::NT) k_critic_front(front::FrontArgs a) {
  extern __shared__ __attribute__((aligned(16))) float cf_lds[];
  tuned::DevRunner<false> run(nullptr, 0);
  const front::FrontArgs& la = tuned::args_to_lds<front::FrontArgs>(cf_lds);   // (`a` itself is never addressed)
  front::critic_front_prologue(run, cf_lds, la);
  for (int r = blockIdx.x; r < la.nrows; r += gridDim.x) front::critic_front_row<MODE>(run, cf_lds, la, r);
}
template <int MODE>
static void launch_front(const front::FrontArgs& a, hipStream_t s) {
  // (the attribute is a property of the function ON A DEVICE: one flag per device, the critic has no context object to keep it in)
  static bool attr[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_critic_front<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              frame::L_TOTAL * 4);
    if (dev >= 0 && dev < 64) attr[dev] = true;
  }
  hipLaunchKernelGGL(k_critic_front<MODE>, dim3((unsigned)std::min(a.nrows, 1024)), dim3(frame::NT), frame::L_TOTAL * 4, s, a);
}

struct DiscL {
  int cin, hin, cout, hout, k, s, pad;
  int64_t w_off, b_off, beta_off, gamma_off;
  bool dense;  // every output position sees the whole input: run as a dense layer on the matrix cores
  int n() const { return cout * hout; }
  int kin() const { return cin * hin; }
};
struct vaenpvc_disc {
  int H, n_layers, flat;
  bool front;   // layers 0-1 have the VCC2016 geometry (1 -> 16 -> 32 channels, 7 taps, stride 3 on 513 bins): disc_frame.h serves them
  DiscL l[VAENPVC_MAX_LAYERS];
  int64_t wd_off, bd_off, n_params;
  std::vector<ParamInfo> table;
};

static G mk(const DiscL& l) { return G{l.cin, l.hin, l.cout, l.hout, l.k, l.s, l.pad}; }
static Act act_of(const DiscL& l, const float* P, const float* st) { return Act{st, P + l.gamma_off, P + l.beta_off}; }

namespace {
// st_new != nullptr: the statistics `ai` would read do not exist yet -- the kernel takes them from the staged input and stores them there
void conv_fwd(const float* in, Act ai, const float* W, const float* b, float* out, int64_t B, const DiscL& l, hipStream_t s,
              float* st_new = nullptr) {
  hipLaunchKernelGGL(k_conv_fwd, dim3((unsigned)B, (unsigned)((l.cout + FWD_OB - 1) / FWD_OB)), dim3(256),
                     ((size_t)l.cin * l.hin + 1024) * sizeof(float), s, in, ai, W, b, out, mk(l), st_new);
}
void conv_bwd_data(const float* dout, const float* W, float* din, int64_t B, const DiscL& l, hipStream_t s) {
  hipLaunchKernelGGL(k_conv_bwd_data, dim3((unsigned)B, (unsigned)((l.cin + BWD_CB - 1) / BWD_CB)), dim3(256),
                     (size_t)l.cout * l.hout * sizeof(float), s, dout, W, din, mk(l));
}
// the frame loop is dealt to several workgroups with private copies of dW: up to 16 for small layers, up to 6 for larger ones
constexpr int WG_SPLIT_MAX_W = 262144, WG_SPLIT_SMALL_W = 16384, WG_SPLIT = 16, WG_SPLIT_BIG = 6;
// `part` / `sums`: where this call may put per-workgroup copies of dW and the list the final k_sum_parts works through
void conv_bwd_w(const float* in, Act ai, const float* dout, float* dW, int64_t B, const DiscL& l, float* part, PartSums* sums,
                hipStream_t s) {
  const int per = l.cout * l.hout + l.hin;
  const int FB = std::max(1, std::min(8, 14000 / per));  // staged frames (<= 56 KB of LDS)
  const int nw = l.k * l.cin * l.cout;
  const dim3 grid((unsigned)l.cin, (unsigned)((l.k + WG_TC - 1) / WG_TC));
  const size_t lds = (size_t)FB * per * sizeof(float);
  if (nw <= WG_SPLIT_MAX_W && B > FB && sums->count < 8) {
    const int nz = (int)std::min<int64_t>(nw <= WG_SPLIT_SMALL_W ? WG_SPLIT : WG_SPLIT_BIG, (B + FB - 1) / FB);
    const int64_t fper = (B + nz - 1) / nz;
    hipLaunchKernelGGL(k_conv_bwd_w, dim3(grid.x, grid.y, (unsigned)nz), dim3(256), lds, s, in, ai, dout, part, B, mk(l), FB, fper, 0);
    sums->e[sums->count++] = PartSum{part, dW, nw, nz};
  } else {
    hipLaunchKernelGGL(k_conv_bwd_w, grid, dim3(256), lds, s, in, ai, dout, dW, B, mk(l), FB, B, 1);
  }
}
constexpr int MM_MAX_SPLIT = 8;   // reduction splits of a dense-like layer's GEMMs
struct DWs {  // resolved workspace of one call; B rows in the forward tensors, R rows in the per-range ones
  float* rows;
  float* u[VAENPVC_MAX_LAYERS];
  float* st[VAENPVC_MAX_LAYERS];
  float* d;
  float* abar[VAENPVC_MAX_LAYERS];
  float* ubar[VAENPVC_MAX_LAYERS];
  float* q[VAENPVC_MAX_LAYERS];
  float* at[VAENPVC_MAX_LAYERS];
  float* udir[VAENPVC_MAX_LAYERS];
  float* pn[VAENPVC_MAX_LAYERS];
  float *g, *gt, *gp_f;
  float* da[VAENPVC_MAX_LAYERS];   // gradient at the activations of layer i (pass 4)
  float* du[VAENPVC_MAX_LAYERS];
  float* part[2][VAENPVC_MAX_LAYERS];  // per-workgroup copies of a layer's weight gradient, per pass (3, 4)
  // dense-like layers: expanded weights, activated input of all rows, split results of the GEMMs, gradient of Wd
  float* Wd[VAENPVC_MAX_LAYERS];
  float* ain[VAENPVC_MAX_LAYERS];
  float* mmpart;
  float* dWd;
  float* w1t;   // layer 1's kernel transposed (disc_frame.h), when the front kernels serve the model
};
int64_t al(int64_t n) { return (n + 63) & ~int64_t(63); }

// carve (base == nullptr: size only).  `critic`: three row groups and the double-backward tensors
int64_t carve(const vaenpvc_disc& m, int64_t F, bool critic, float* base, DWs* w) {
  int64_t off = 0;
  auto take = [&](int64_t n) {
    float* p = base ? base + off : nullptr;
    off += al(n);
    return p;
  };
  const int64_t B = (critic ? 3 : 2) * F;
  DWs t;
  memset(&t, 0, sizeof t);
  t.rows = take(B * m.H);
  if (m.front) t.w1t = take(7 * 16 * 32);
  for (int i = 0; i < m.n_layers; ++i) {
    t.u[i] = take(B * m.l[i].n());
    t.st[i] = take(B * 2);
  }
  t.d = take(B);
  for (int i = 0; i < m.n_layers; ++i) {
    t.abar[i] = take(F * m.l[i].n());
    t.ubar[i] = take(F * m.l[i].n());
  }
  t.g = take(F * m.H);
  {
    int64_t mmax = 0, wmax = 0;
    for (int i = 0; i < m.n_layers; ++i)
      if (m.l[i].dense) {
        const int64_t K = m.l[i].kin(), N = m.l[i].n();
        t.Wd[i] = take(K * N);
        t.ain[i] = take((B + (critic ? F : 0)) * K);   // (critic: + the F rows of pass 3's operand, see the aliases below)
        mmax = std::max(mmax, (int64_t)MM_MAX_SPLIT * B * std::max(K, N));
        wmax = std::max(wmax, K * N);
      }
    if (mmax) t.mmpart = take(mmax);
    if (wmax && critic) t.dWd = take(wmax);
  }
  if (critic) {
    for (int i = 0; i < m.n_layers; ++i) {
      t.q[i] = take(F * m.l[i].n());
      t.at[i] = take(F * m.l[i].n());
      t.udir[i] = take(F * m.l[i].n());
    }
    for (int i = 0; i < m.n_layers; ++i) t.pn[i] = take(F * m.l[i].n());
    t.gt = take(F * m.H);
    t.gp_f = take(F);
    for (int i = 0; i < m.n_layers; ++i) t.da[i] = take(B * m.l[i].n());
    for (int i = 0; i < m.n_layers; ++i) t.du[i] = take((B + (m.l[i].dense ? F : 0)) * m.l[i].n());
    // A dense-like layer's weight gradient is ONE GEMM over the rows of pass 4 and pass 3 stacked: the operands of pass 3
    // (adjoint of the layer's input, ubar of the layer) live directly behind those of pass 4 (activated input, du)
    for (int i = 1; i < m.n_layers; ++i)
      if (m.l[i].dense) {
        t.at[i - 1] = t.ain[i] + B * m.l[i].kin();
        t.ubar[i] = t.du[i] + B * m.l[i].n();
      }
    for (int p = 0; p < 2; ++p)
      for (int i = 0; i < m.n_layers; ++i) {
        const int64_t nw = (int64_t)m.l[i].k * m.l[i].cin * m.l[i].cout;
        t.part[p][i] = nw <= WG_SPLIT_MAX_W ? take(nw * (nw <= WG_SPLIT_SMALL_W ? WG_SPLIT : WG_SPLIT_BIG)) : nullptr;
      }
  }
  if (w) *w = t;
  return off;
}

// ---- dense-like layers (k_mm): host side
struct MmPlan {
  int splits, rper;
};
MmPlan mm_plan(int M, int N, int R) {
  const int tiles = ((M + 63) / 64) * ((N + 63) / 64);
  int splits = std::max(1, std::min(MM_MAX_SPLIT, 256 / std::max(1, tiles)));
  splits = std::min(splits, (R + 31) / 32);
  const int rper = (((R + splits - 1) / splits) + 31) / 32 * 32;
  return MmPlan{(R + rper - 1) / rper, rper};
}
template <bool AF, bool BF>
void mm_launch(const float* A, int64_t a_m, int64_t a_r, const float* Bm, int64_t b_r, int64_t b_n, float* C, int M, int N, int R,
               MmPlan p, hipStream_t s) {
  MmArgs a{A, a_m, a_r, Bm, b_r, b_n, C, M, N, R, p.rper};
  hipLaunchKernelGGL((k_mm<AF, BF>), dim3((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)p.splits), dim3(256), 0, s, a);
}
// out[rows][N] = in[rows][K] Wd (+ bias per output channel)
void dense_fwd(const float* in, const float* Wd, const float* bias, float* out, int64_t rows, const DiscL& l, float* part, hipStream_t s) {
  const int K = l.kin(), N = l.n();
  const MmPlan p = mm_plan((int)rows, N, K);
  mm_launch<true, true>(in, K, 1, Wd, N, 1, part, (int)rows, N, K, p, s);
  hipLaunchKernelGGL(k_mm_reduce, grid1(rows * N), dim3(256), 0, s, part, p.splits, rows * N, out, bias, N, l.hout);
}
// din[rows][K] = dout[rows][N] Wd^T
// (reduce = false: the split-K parts stay in `part` -- the consumer sums them on load, disc_frame.h; returns their number)
int dense_dgrad(const float* dout, const float* Wd, float* din, int64_t rows, const DiscL& l, float* part, hipStream_t s, bool reduce = true) {
  const int K = l.kin(), N = l.n();
  const MmPlan p = mm_plan((int)rows, K, N);
  mm_launch<true, false>(dout, N, 1, Wd, 1, N, part, (int)rows, K, N, p, s);
  if (reduce) hipLaunchKernelGGL(k_mm_reduce, grid1(rows * K), dim3(256), 0, s, part, p.splits, rows * (int64_t)K, din, (const float*)nullptr, K, 1);
  return p.splits;
}
// dW += fold(in^T dout)
void dense_wgrad(const float* in, const float* dout, float* dWd, float* dW, int64_t rows, const DiscL& l, hipStream_t s) {
  const int K = l.kin(), N = l.n();
  mm_launch<false, true>(in, 1, K, dout, N, 1, dWd, K, N, (int)rows, MmPlan{1, (int)((rows + 31) / 32 * 32)}, s);
  hipLaunchKernelGGL(k_fold_dense, grid1((int64_t)l.k * l.cin * l.cout), dim3(256), 0, s, dWd, dW, mk(l));
}
void expand_dense(const vaenpvc_disc& m, const float* P, const DWs& w, hipStream_t s) {
  for (int i = 0; i < m.n_layers; ++i)
    if (m.l[i].dense)
      hipLaunchKernelGGL(k_expand_dense, grid1((int64_t)m.l[i].kin() * m.l[i].n()), dim3(256), 0, s, P + m.l[i].w_off, w.Wd[i], mk(m.l[i]));
}

front::FrontArgs front_args(const vaenpvc_disc& m, const float* P, const DWs& w, int64_t rows0, int64_t nrows) {
  front::FrontArgs a;
  memset(&a, 0, sizeof a);
  const DiscL &l0 = m.l[0], &l1 = m.l[1];
  a.P = P;
  a.w0 = (int)l0.w_off; a.b0 = (int)l0.b_off; a.g0 = (int)l0.gamma_off; a.bt0 = (int)l0.beta_off;
  a.w1 = (int)l1.w_off; a.b1 = (int)l1.b_off; a.g1 = (int)l1.gamma_off; a.bt1 = (int)l1.beta_off;
  a.w1t = w.w1t;
  a.rows0 = (int)rows0;
  a.nrows = (int)nrows;
  a.rows = w.rows;
  a.u0 = w.u[0]; a.st0 = w.st[0]; a.u1 = w.u[1]; a.st1 = w.st[1];
  a.ain2 = w.ain[2];
  a.abar1 = w.abar[1]; a.ubar1 = w.ubar[1]; a.abar0 = w.abar[0]; a.ubar0 = w.ubar[0];
  a.g = w.g; a.gt = w.gt; a.gp_f = w.gp_f;
  a.at0 = w.at[0]; a.udir0 = w.udir[0]; a.pn0 = w.pn[0]; a.at1 = w.at[1]; a.udir1 = w.udir[1]; a.pn1 = w.pn[1];
  a.da1 = w.da[1]; a.du1 = w.du[1]; a.da0 = w.da[0]; a.du0 = w.du[0];
  return a;
}

// (dw / dc non-null, critic step: the dense unit's own gradient rides in its forward kernel, coefficients -1/F | +1/F | 0)
void forward(const vaenpvc_disc& m, const float* P, int64_t B, const DWs& w, hipStream_t s, float* dw = nullptr, float* dc = nullptr,
             int64_t F0 = 0, float c0 = 0.f, float c1 = 0.f) {
  if (m.front) launch_front<front::FP_FWD>(front_args(m, P, w, 0, B), s);   // layers 0-1: outputs, statistics, the 115-tap layer's input
  for (int i = 0; i < m.n_layers; ++i) {
    const DiscL& l = m.l[i];
    Act ai = i == 0 ? kNoAct : act_of(m.l[i - 1], P, w.st[i - 1]);
    if (m.front && i < 2) continue;
    if (l.dense) {  // statistics + activated copy of the input, then one GEMM
      const DiscL& pl = m.l[i - 1];
      if (!(m.front && i == 2))
      hipLaunchKernelGGL(k_act_stats, dim3((unsigned)B), dim3(256), 0, s, w.u[i - 1], P + pl.gamma_off, P + pl.beta_off, w.st[i - 1],
                         w.ain[i], pl.cout, pl.hout);
      dense_fwd(w.ain[i], w.Wd[i], P + l.b_off, w.u[i], B, l, w.mmpart, s);
      continue;
    }
    conv_fwd(i == 0 ? w.rows : w.u[i - 1], ai, P + l.w_off, P + l.b_off, w.u[i], B, l, s, i == 0 ? nullptr : w.st[i - 1]);
  }
  const DiscL& last = m.l[m.n_layers - 1];
  hipLaunchKernelGGL(k_dense_fwd, dim3((unsigned)B), dim3(256), 0, s, w.u[m.n_layers - 1],
                     act_of(last, P, w.st[m.n_layers - 1]), last.hout, P + m.wd_off, P + m.bd_off, w.d, m.flat,
                     w.st[m.n_layers - 1], dw, dc, F0, c0, c1, 0.0f);
}

// pass 2: g = d(sum_f d_f)/d(rows) for R rows starting at row r0; keeps abar_l / ubar_l
// (penalty: the front kernel also leaves the penalty and its adjoint gt = coef (|g| - 1) g / |g| -- the work of k_gp)
void input_gradient(const vaenpvc_disc& m, const float* P, int64_t r0, int64_t R, const DWs& w, hipStream_t s, bool penalty = false,
                    float coef = 0.f) {
  const int L = m.n_layers;
  const VDy ones{P + m.wd_off, 1.0f, 1.0f, 1.0f, R, 0};     // upstream of the last conv layer: w itself (d(sum_f d_f) / d(act))
  int up_parts = 0;
  for (int i = L - 1; i >= 0; --i) {
    const DiscL& l = m.l[i];
    if (m.front && i == 1) {   // layers 1 and 0 (+ the penalty): one launch
      front::FrontArgs fa = front_args(m, P, w, r0, R);
      fa.up = w.mmpart;
      fa.up_parts = up_parts;
      fa.up_stride = (long long)R * m.l[1].n();
      fa.coef = coef;
      fa.penalty = penalty ? 1 : 0;
      launch_front<front::FP_IGRAD>(fa, s);
      break;
    }
    hipLaunchKernelGGL(k_ln_bwd, dim3((unsigned)R), dim3(256), 0, s, i == L - 1 ? (const float*)nullptr : w.abar[i], w.u[i] + r0 * l.n(),
                       w.st[i] + 2 * r0, P + l.gamma_off, P + l.beta_off, (const float*)nullptr, (int64_t)0, w.ubar[i], l.cout, l.hout,
                       i == L - 1 ? ones : kNoVDy);
    if (l.dense) {
      const bool onload = m.front && i == 2;   // the front kernel sums the parts while it loads them
      const int np = dense_dgrad(w.ubar[i], w.Wd[i], w.abar[i - 1], R, l, w.mmpart, s, !onload);
      if (onload) up_parts = np;
    }
    else conv_bwd_data(w.ubar[i], P + l.w_off, i == 0 ? w.g : w.abar[i - 1], R, l, s);
  }
}
}  // namespace

extern "C" {

int vaenpvc_disc_create(const vaenpvc_disc_arch* a, vaenpvc_disc** out) {
  if (!a || !out) return abi_error(VAENPVC_E_ARG, "null argument");
  if (a->n_layers < 1 || a->n_layers > VAENPVC_MAX_LAYERS) return abi_error(VAENPVC_E_ARG, "discriminator: need 1..8 layers");
  if (a->H < 1) return abi_error(VAENPVC_E_ARG, "H must be positive");
  vaenpvc_disc* m = new vaenpvc_disc();
  m->H = a->H;
  m->n_layers = a->n_layers;
  int64_t off = 0;
  auto add = [&](const std::string& name, std::initializer_list<int64_t> shp) {
    ParamInfo p;
    p.name = name;
    p.offset = off;
    p.ndim = (int)shp.size();
    p.count = 1;
    int i = 0;
    for (int64_t s : shp) {
      p.shape[i++] = s;
      p.count *= s;
    }
    for (; i < 4; ++i) p.shape[i] = 1;
    off += p.count;
    m->table.push_back(p);
    return p.offset;
  };
  int c = 1, h = a->H;
  for (int i = 0; i < a->n_layers; ++i) {
    DiscL& l = m->l[i];
    int k = a->kernel[i], s = a->stride[i], o = a->output[i];
    if (k < 1 || s < 1 || o < 1) {
      delete m;
      return abi_error(VAENPVC_E_ARG, "discriminator: kernel/stride/output must be positive");
    }
    l.cin = c;
    l.hin = h;
    l.cout = o;
    l.k = k;
    l.s = s;
    l.hout = (h + s - 1) / s;  // TF SAME
    l.pad = std::max((l.hout - 1) * s + k - h, 0) / 2;
    std::string p = "Discriminator/Conv2d-" + std::to_string(i) + "/";
    l.w_off = add(p + "kernel", {k, 1, c, o});
    l.b_off = add(p + "bias", {o});
    l.beta_off = add(p + "layernorm.offset", {o, 1, 1});
    l.gamma_off = add(p + "layernorm.scale", {o, 1, 1});
    // kernel limits (disc.hip): one thread per output channel in the weight gradient, a frame's layer input /
    // output staged in LDS
    if (o > 256 || (int64_t)l.cin * l.hin > 16000 || (int64_t)l.cout * l.hout + l.hin > 16000) {
      delete m;
      return abi_error(VAENPVC_E_UNSUPPORTED, "discriminator: a layer exceeds 256 channels or 16000 values per frame");
    }
    // dense-like: every (input position, output position) pair has a tap, the layer has an activated input, and the
    // matrices are big enough for the matrix cores to matter (VAENPVC_DISC_DENSE=0: keep the conv kernels, for A/B)
    {
      const char* e = getenv("VAENPVC_DISC_DENSE");
      l.dense = i > 0 && l.pad - s * (l.hout - 1) >= 0 && l.hin - 1 + l.pad < k && (int64_t)l.cin * l.hin >= 512 &&
                (int64_t)l.cout * l.hout >= 512 && !(e && e[0] == '0');
    }
    c = o;
    h = l.hout;
  }
  {
    const char* e = getenv("VAENPVC_DISC_FRONT");   // (=0: keep the per-layer kernels, for A/B)
    const DiscL &l0 = m->l[0], &l1 = m->l[1];
    m->front = a->n_layers >= 3 && a->H == 513 && l0.k == 7 && l0.s == 3 && l0.cout == 16 && l0.pad == 2 && l1.k == 7 && l1.s == 3 &&
               l1.cout == 32 && l1.pad == 2 && !l1.dense && m->l[2].dense && !(e && e[0] == '0');
  }
  m->flat = c * h;
  m->wd_off = add("Discriminator/dense/kernel", {m->flat, 1});
  m->bd_off = add("Discriminator/dense/bias", {1});
  m->n_params = off;
  *out = m;
  return 0;
}

void vaenpvc_disc_destroy(vaenpvc_disc* d) { delete d; }
int vaenpvc_disc_param_count(const vaenpvc_disc* d) { return d ? (int)d->table.size() : VAENPVC_E_ARG; }
int64_t vaenpvc_disc_param_floats(const vaenpvc_disc* d) { return d ? d->n_params : VAENPVC_E_ARG; }

int vaenpvc_disc_param_info(const vaenpvc_disc* d, int index, char* name, int name_cap, int64_t* offset_floats,
                            int32_t* ndim, int64_t* shape) {
  if (!d || index < 0 || index >= (int)d->table.size()) return abi_error(VAENPVC_E_ARG, "bad parameter index");
  const ParamInfo& p = d->table[index];
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

int64_t vaenpvc_disc_workspace_bytes(const vaenpvc_disc* d, int64_t F) {
  if (!d || F < 1 || F > (1LL << 16)) return abi_error(VAENPVC_E_ARG, "bad argument (1 <= F <= 65536)");
  return carve(*d, F, true, nullptr, nullptr) * 4;
}

static int disc_args(const vaenpvc_disc* d, int64_t F, const void* d_ws, size_t ws_bytes) {
  if (F < 1 || F > (1LL << 16)) return abi_error(VAENPVC_E_ARG, "F must be in [1, 65536]");
  if (!d_ws || ws_bytes < (size_t)carve(*d, F, true, nullptr, nullptr) * 4)
    return abi_error(VAENPVC_E_WORKSPACE, "discriminator workspace too small");
  if (((uintptr_t)d_ws & 15) != 0) return abi_error(VAENPVC_E_ARG, "workspace must be 16-byte aligned");
  return 0;
}
static int disc_check(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[256];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    return abi_error(VAENPVC_E_HIP, buf);
  }
  return 0;
}

int vaenpvc_disc_fwd(const vaenpvc_disc* d, const float* d_dparams, const float* d_x, const float* d_xh, int64_t F,
                     float* d_out, float* d_loss2, void* d_ws, size_t ws_bytes, void* stream) {
  if (!d || !d_dparams || !d_x || !d_xh) return abi_error(VAENPVC_E_ARG, "null argument");
  int rc = disc_args(d, F, d_ws, ws_bytes);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  DWs w;
  carve(*d, F, false, (float*)d_ws, &w);
  hipLaunchKernelGGL(k_rows, grid1(std::max<int64_t>(F * d->H, d->front ? 3584 : 0)), dim3(256), 0, s, d_x, d_xh, (const float*)nullptr, w.rows,
                     F, d->H, d->front ? d_dparams + d->l[1].w_off : nullptr, w.w1t);
  expand_dense(*d, d_dparams, w, s);
  forward(*d, d_dparams, 2 * F, w, s);
  if (d_out) (void)hipMemcpyAsync(d_out, w.d, 2 * F * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (d_loss2) hipLaunchKernelGGL(k_losses, dim3(1), dim3(256), 0, s, w.d, (const float*)nullptr, F, d_loss2);
  return disc_check("disc_fwd");
}

int vaenpvc_disc_critic_fwd_bwd(const vaenpvc_disc* d, const float* d_dparams, const float* d_x, const float* d_xh,
                                const float* d_t, int64_t F, float lambda, float* d_dgrads, float* d_loss2,
                                void* d_ws, size_t ws_bytes, void* stream) {
  if (!d || !d_dparams || !d_x || !d_xh || !d_t || !d_dgrads || !d_loss2) return abi_error(VAENPVC_E_ARG, "null argument");
  int rc = disc_args(d, F, d_ws, ws_bytes);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const vaenpvc_disc& m = *d;
  const float* P = d_dparams;
  float* Gd = d_dgrads;
  const int L = m.n_layers;
  const int64_t B = 3 * F;
  DWs w;
  carve(m, F, true, (float*)d_ws, &w);
  // (a kernel, not hipMemsetAsync: the step is replayed from hipGraphs by hipvae/adversarial.py, and a fill node next to
  //  other fill nodes in one graph was observed to misbehave on replay)
  PartSums sums;
  sums.count = 0;
  ChanSums csums;   // per-channel reductions of both passes, one launch at the end
  csums.count = 0;
  // pass 1
  hipLaunchKernelGGL(k_rows, grid1(std::max<int64_t>(F * m.H, m.n_params)), dim3(256), 0, s, d_x, d_xh, d_t, w.rows, F, m.H,
                     m.front ? P + m.l[1].w_off : nullptr, w.w1t, Gd, m.n_params);
  const float cr = -1.0f / (float)F, cf = 1.0f / (float)F;     // pass-4 upstream of d: -1/F (x), +1/F (xh), 0 (xi)
  expand_dense(m, P, w, s);
  forward(m, P, B, w, s, Gd + m.wd_off, Gd + m.bd_off, F, cr, cf);
  // pass 2 (rows xi) and the penalty
  input_gradient(m, P, 2 * F, F, w, s, m.front, 2.0f * lambda / (float)F);
  if (!m.front) hipLaunchKernelGGL(k_gp, dim3((unsigned)F), dim3(256), 0, s, w.g, w.gt, w.gp_f, m.H, 2.0f * lambda / (float)F);
  hipLaunchKernelGGL(k_losses, dim3(1), dim3(256), 0, s, w.d, w.gp_f, F, d_loss2);
  // pass 3, bottom-up over rows xi
  if (m.front) launch_front<front::FP_ADJ>(front_args(m, P, w, 2 * F, F), s);   // layers 0-1 of pass 3: q, at, udir, pn
  for (int i = m.front ? 2 : 0; i < L; ++i) {
    const DiscL& l = m.l[i];
    const float* src = i == 0 ? w.gt : w.at[i - 1];  // adjoint of abar_{i-1} (of g for the first layer)
    if (l.dense) {
      dense_fwd(src, w.Wd[i], nullptr, w.q[i], F, l, w.mmpart, s);   // (its weight gradient: with pass 4's, stacked rows)
    } else {
      conv_fwd(src, kNoAct, P + l.w_off, nullptr, w.q[i], F, l, s);
      if (!(m.front && i < 2)) conv_bwd_w(src, kNoAct, w.ubar[i], Gd + l.w_off, F, l, w.part[0][i], &sums, s);
    }
    hipLaunchKernelGGL(k_ln_bwd_bwd, dim3((unsigned)F), dim3(256), 0, s, w.q[i], i == L - 1 ? (const float*)nullptr : w.abar[i],
                       w.u[i] + 2 * F * l.n(), w.st[i] + 4 * F, P + l.gamma_off, P + l.beta_off, w.at[i], w.udir[i], w.pn[i], l.cout, l.hout,
                       i == L - 1 ? VDy{P + m.wd_off, 1.0f, 1.0f, 1.0f, F, 0} : kNoVDy);
    if (!(m.front && i < 2)) csums.e[csums.count++] = ChanSum{w.pn[i], Gd + l.gamma_off, F, l.cout, l.hout};   // adjoint of gamma
  }
  csums.e[csums.count++] = ChanSum{w.at[L - 1], Gd + m.wd_off, F, m.flat, 1};   // abar_top = w
  // pass 4, all rows: upstream -1/F (x), +1/F (xh), 0 (xi) -- virtual at the last conv layer (VDy)
  const VDy top{P + m.wd_off, cr, cf, 0.0f, F, 0};
  ParamGrads pgs;   // LayerNorm parameter gradients of all layers: one launch at the end
  pgs.count = 0;
  int up4_parts = 0;
  for (int i = L - 1; i >= 0; --i) {
    const DiscL& l = m.l[i];
    if (m.front && i == 1) {   // layers 1 and 0 of pass 4: LayerNorm backward (+ udir on the rows xi), input gradient
      front::FrontArgs fa = front_args(m, P, w, 0, B);
      fa.up = w.mmpart;
      fa.up_parts = up4_parts;
      fa.up_stride = (long long)B * m.l[1].n();
      fa.add1 = w.udir[1];
      fa.add0 = w.udir[0];
      fa.add_row0 = (int)(2 * F);
      launch_front<front::FP_BWD>(fa, s);
      break;   // (their parameter gradients: the job-list launch below)
    }
    const bool fr = false;
    if (!fr)
    pgs.e[pgs.count++] = ParamGrad{i == L - 1 ? (const float*)nullptr : w.da[i], w.u[i], w.st[i], P + l.gamma_off, P + l.beta_off,
                                   Gd + l.gamma_off, Gd + l.beta_off, B, l.cout, l.hout, i == L - 1 ? top : kNoVDy};
    hipLaunchKernelGGL(k_ln_bwd, dim3((unsigned)B), dim3(256), 0, s, i == L - 1 ? (const float*)nullptr : w.da[i], w.u[i], w.st[i],
                       P + l.gamma_off, P + l.beta_off, w.udir[i], 2 * F, w.du[i], l.cout, l.hout,
                       i == L - 1 ? top : kNoVDy);   // udir on the rows xi
    Act ai = i == 0 ? kNoAct : act_of(m.l[i - 1], P, w.st[i - 1]);
    if (l.dense) dense_wgrad(w.ain[i], w.du[i], w.dWd, Gd + l.w_off, B + F, l, s);   // (ain: the activated input kept by pass 1; + pass 3's rows)
    else if (!fr) conv_bwd_w(i == 0 ? w.rows : w.u[i - 1], ai, w.du[i], Gd + l.w_off, B, l, w.part[1][i], &sums, s);
    if (!fr) csums.e[csums.count++] = ChanSum{w.du[i], Gd + l.b_off, B, l.cout, l.hout};   // conv bias
    if (i > 0) {
      if (l.dense) {
        const bool onload = m.front && i == 2;
        const int np = dense_dgrad(w.du[i], w.Wd[i], w.da[i - 1], B, l, w.mmpart, s, !onload);
        if (onload) up4_parts = np;
      }
      else conv_bwd_data(w.du[i], P + l.w_off, w.da[i - 1], B, l, s);
    }
  }
  if (m.front) {   // layers 0-1: weight, bias and LayerNorm-parameter gradients of passes 3 and 4 in one launch (atomics)
    const DiscL &l0 = m.l[0], &l1 = m.l[1];
    front::CwArgs ca{w.rows, w.gt, w.u[0], w.st[0], w.u[1], w.st[1], P + l0.gamma_off, P + l0.beta_off, P + l1.gamma_off, P + l1.beta_off,
                     w.at[0], w.ubar[0], w.ubar[1], w.du[0], w.du[1], w.da[0], w.da[1], w.pn[0], w.pn[1],
                     Gd + l0.w_off, Gd + l1.w_off, Gd + l0.b_off, Gd + l1.b_off, Gd + l0.gamma_off, Gd + l0.beta_off,
                     Gd + l1.gamma_off, Gd + l1.beta_off, (int)F, (int)B};
    const front::CwPlan cp = front::make_cwplan((int)F, (int)B);
    hipLaunchKernelGGL(k_critic_front_wgrad, dim3((unsigned)cp.start[front::CW_SEGS]), dim3(frame::WT), frame::WG_LDS * 4, s, ca, cp);
  }
  if (pgs.count > 0) {
    int cmax = 0;
    for (int e = 0; e < pgs.count; ++e) cmax = std::max(cmax, pgs.e[e].C);
    hipLaunchKernelGGL(k_ln_param_grad, dim3((unsigned)cmax, (unsigned)pgs.count), dim3(B * 16 >= 4096 ? 1024 : 256), 0, s, pgs);   // one workgroup per channel walks all rows
  }
  {
    int cmax = 0;
    for (int e = 0; e < csums.count; ++e) cmax = std::max(cmax, csums.e[e].C);
    hipLaunchKernelGGL(k_chan_sum_multi, dim3((unsigned)cmax, (unsigned)csums.count), dim3(B * 16 >= 4096 ? 1024 : 256), 0, s, csums);
  }
  if (sums.count > 0) {  // the weight-gradient copies of both passes, one launch
    int nmax = 0;
    for (int e = 0; e < sums.count; ++e) nmax = std::max(nmax, sums.e[e].n);
    hipLaunchKernelGGL(k_sum_parts, dim3((unsigned)std::min(256, (nmax + 255) / 256)), dim3(256), 0, s, sums, nmax);
  }
  return disc_check("disc_critic_fwd_bwd");
}

int vaenpvc_disc_generator_target(const vaenpvc_disc* d, const float* d_dparams, const float* d_x, const float* d_xh,
                                  int64_t F, float alpha, float* d_target, float* d_loss2, void* d_ws,
                                  size_t ws_bytes, void* stream) {
  if (!d || !d_dparams || !d_x || !d_xh || !d_target) return abi_error(VAENPVC_E_ARG, "null argument");
  int rc = disc_args(d, F, d_ws, ws_bytes);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  DWs w;
  carve(*d, F, false, (float*)d_ws, &w);
  hipLaunchKernelGGL(k_rows, grid1(std::max<int64_t>(F * d->H, d->front ? 3584 : 0)), dim3(256), 0, s, d_x, d_xh, (const float*)nullptr, w.rows,
                     F, d->H, d->front ? d_dparams + d->l[1].w_off : nullptr, w.w1t);
  expand_dense(*d, d_dparams, w, s);
  forward(*d, d_dparams, 2 * F, w, s);
  input_gradient(*d, d_dparams, F, F, w, s);  // rows xh
  hipLaunchKernelGGL(k_adv_target, grid1(F * d->H), dim3(256), 0, s, d_x, w.g, d_target, F * d->H, alpha);
  if (d_loss2) hipLaunchKernelGGL(k_losses, dim3(1), dim3(256), 0, s, w.d, (const float*)nullptr, F, d_loss2);
  return disc_check("disc_generator_target");
}

}  // extern "C"
