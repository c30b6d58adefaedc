// gfx950_elem.h -- HBM-bound kernels of the tuned path: LayerNorm statistics, the fused
// LayerNorm+lrelu backward (input gradient + d gamma / d beta / d bias in one pass over the
// tensors), column / total reductions, the speaker-embedding gradient and weight packing.
#pragma once
#include "gfx950_common.h"

namespace vaenpvc {
namespace tuned {

// ---------------------------------------------------------------- LayerNorm statistics
// util/layers.py:32 -- one wave per frame, the frame (N floats, N % 4 == 0) is read ONCE
// with 16-byte loads and kept in registers for the two-pass (mean, then centred variance).
template <int N>
__global__ void __launch_bounds__(256) k_ln_stats_fast(const float* __restrict__ a, float* __restrict__ st, int F) {
  static_assert(N % 4 == 0, "frame size must be a multiple of 4 floats");
  constexpr int NV = N / 4, PER = cdiv(NV, 64);
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= F) return;
  const float4* p = reinterpret_cast<const float4*>(a + (int64_t)f * N);
  float4 v[PER];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    int idx = lane + 64 * i;
    v[i] = idx < NV ? p[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (lane + 64 * i < NV) {
      float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float var = wave_sum(q) / N;
  if (lane == 0) {
    st[2 * f] = mean;
    st[2 * f + 1] = 1.0f / sqrtf(var + LN_EPS);
  }
}

// Same, and additionally writes y = lrelu(gamma_c*(a-mean)*rstd + beta_c) ([C][H] per frame).
// Used for the decoder layer in front of the 1025-tap layer, whose activated output is read
// by three GEMM kernels (forward, weight gradient, edges).
template <int N, int H>
__global__ void __launch_bounds__(256) k_ln_stats_act(const float* __restrict__ a, float* __restrict__ st,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ y, int F) {
  static_assert(N % 4 == 0, "frame size must be a multiple of 4 floats");
  constexpr int NV = N / 4, PER = cdiv(NV, 64);
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= F) return;
  const float4* p = reinterpret_cast<const float4*>(a + (int64_t)f * N);
  float4 v[PER];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    int idx = lane + 64 * i;
    v[i] = idx < NV ? p[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / N;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (lane + 64 * i < NV) {
      float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / N + LN_EPS);
  if (lane == 0) {
    st[2 * f] = mean;
    st[2 * f + 1] = rstd;
  }
  float4* py = reinterpret_cast<float4*>(y + (int64_t)f * N);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    int idx = lane + 64 * i;
    if (idx < NV) {
      int e = 4 * idx;
      float4 o;
      o.x = lnact_v(v[i].x, mean, rstd, gamma[e / H], beta[e / H]);
      o.y = lnact_v(v[i].y, mean, rstd, gamma[(e + 1) / H], beta[(e + 1) / H]);
      o.z = lnact_v(v[i].z, mean, rstd, gamma[(e + 2) / H], beta[(e + 2) / H]);
      o.w = lnact_v(v[i].w, mean, rstd, gamma[(e + 3) / H], beta[(e + 3) / H]);
      py[idx] = o;
    }
  }
}

// y = lrelu(LN(a)) from existing statistics (used when the producing step ran the generic kernel)
__global__ void __launch_bounds__(256) k_act_from_stats(const float* __restrict__ a, const float* __restrict__ st,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, int64_t total, int N, int H) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int64_t f = i / N;
  int c = (int)(i - f * N) / H;
  y[i] = lnact_v(a[i], st[2 * f], st[2 * f + 1], gamma[c], beta[c]);
}

// ---------------------------------------------------------------- encoder layer 0
// conv k=7 s=3 pad=2, 1 -> 16 channels, 513 -> 171 bins (util/layers.py:56-64) fused with its
// LayerNorm statistics: K = 7 is far too small for MFMA (0.4 % of the MACs), the layer is
// HBM-bound (reads 2 KB, writes 10.9 KB per frame).  One workgroup walks a chunk of frames;
// the input row sits in LDS, every thread produces the outputs idx = tid + 256k (coalesced
// stores) and the whole frame is reduced in-block for (mean, rstd).
__global__ void __launch_bounds__(256) k_enc0_fwd(const float* __restrict__ x, const float* __restrict__ W,
                                                  const float* __restrict__ bias, float* __restrict__ a,
                                                  float* __restrict__ st, int F, int fchunk) {
  constexpr int H = 513, HO = 171, CO = 16, N = CO * HO, EPT = cdiv(N, 256);
  __shared__ float xs[H + 8];
  __shared__ float wl[7 * CO + CO];
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 7 * CO; i += 256) wl[i] = W[i];
  if (tid < CO) wl[7 * CO + tid] = bias[tid];
  if (tid < 8) xs[tid < 2 ? tid : H + tid] = 0.f;  // zero halos: xs[0..1], xs[H+2..H+7]
  const int fb = blockIdx.x * fchunk, fe = min(F, fb + fchunk);
  for (int f = fb; f < fe; ++f) {
    __syncthreads();
    for (int i = tid; i < H; i += 256) xs[2 + i] = x[(int64_t)f * H + i];
    __syncthreads();
    float v[EPT];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      int idx = tid + 256 * k;
      v[k] = 0.f;
      if (idx < N) {
        int o = idx / HO, j = idx - o * HO;
        float acc = wl[7 * CO + o];
#pragma unroll
        for (int t = 0; t < 7; ++t) acc += wl[t * CO + o] * xs[3 * j + t];
        v[k] = acc;
        a[(int64_t)f * N + idx] = acc;
        s += acc;
      }
    }
    s = wave_sum(s);
    if (lane == 0) red[0][wave] = s;
    __syncthreads();
    const float mean = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) * (1.0f / N);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k)
      if (tid + 256 * k < N) q += (v[k] - mean) * (v[k] - mean);
    q = wave_sum(q);
    if (lane == 0) red[1][wave] = q;
    __syncthreads();
    if (tid == 0) {
      float var = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) * (1.0f / N);
      st[2 * f] = mean;
      st[2 * f + 1] = 1.0f / sqrtf(var + LN_EPS);
    }
  }
}

// ---------------------------------------------------------------- encoder layer 0, one wave per frame
// The same layer without workgroup barriers (the kernel above needs four per frame and divides per output): a lane
// owns the output positions j = lane, lane + 64, lane + 128 and walks the 16 output channels in registers, so every
// store instruction covers 256 contiguous bytes of one channel row; the 7 taps of a position come straight from the
// frame's 2 KB input row (stride-3 loads, L1-resident after the first touch), the weights through uniform (scalar)
// loads; the LayerNorm statistics are two wave reductions over the 48 outputs a lane holds.
__global__ void __launch_bounds__(256) k_enc0_fwd_wave(const float* __restrict__ x, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ a,
                                                       float* __restrict__ st, int F) {
  constexpr int H = 513, HO = 171, CO = 16, N = CO * HO;
  const int lane = threadIdx.x & 63;
  for (int f = blockIdx.x * 4 + (threadIdx.x >> 6); f < F; f += gridDim.x * 4) {
    const float* xf = x + (int64_t)f * H;
    float v[3][CO];
    float s = 0.f;
    float xt[3][7];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = lane + 64 * k;
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const int i = 3 * j + t - 2;
        xt[k][t] = (j < HO && i >= 0 && i < H) ? xf[i] : 0.f;
      }
    }
    // channel blocks of four with the block's 32 parameters as scalar loads (see k_enc0_bwd_wave: all 128 live at once were parked in
    // vector-register lanes, 504 v_readlane / v_writelane per frame)
#pragma unroll
    for (int ob = 0; ob < CO; ob += 4) {
      typedef const __attribute__((address_space(4))) float* cptr;
      unsigned long long w0 = (unsigned long long)W, b0 = (unsigned long long)bias;
      asm volatile("" : "+s"(w0), "+s"(b0));
      const cptr Wp = (cptr)w0, bp = (cptr)b0;
      float Wl[7][4], bl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bl[q] = bp[ob + q];
#pragma unroll
        for (int t = 0; t < 7; ++t) Wl[t][q] = Wp[t * CO + ob + q];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int j = lane + 64 * k;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float acc = bl[q];
#pragma unroll
          for (int t = 0; t < 7; ++t) acc += Wl[t][q] * xt[k][t];
          v[k][ob + q] = j < HO ? acc : 0.f;
          s += v[k][ob + q];
        }
      }
    }
    const float mean = wave_sum(s) * (1.0f / N);
    float q = 0.f;
    float* af = a + (int64_t)f * N;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = lane + 64 * k;
      if (j < HO) {
#pragma unroll
        for (int o = 0; o < CO; ++o) {
          af[o * HO + j] = v[k][o];
          const float d = v[k][o] - mean;
          q += d * d;
        }
      }
    }
    q = wave_sum(q);
    if (lane == 0) {
      st[2 * f] = mean;
      st[2 * f + 1] = 1.0f / sqrtf(q * (1.0f / N) + LN_EPS);
    }
  }
}

// Its weight gradient dW[t][o] = sum_{f,j} x[f][3j + t - 2] * d[f][o][j] the same way: a lane owns positions, keeps the
// 7 x 16 partial sums in registers over all frames of its wave, and the waves' sums are reduced once at the end (wave
// reduction, then one row of partials per workgroup; k_colsum_part adds the rows).
__global__ void __launch_bounds__(256) k_enc0_wgrad_wave(const float* __restrict__ x, const float* __restrict__ d,
                                                         float* __restrict__ part, int F) {
  constexpr int H = 513, HO = 171, CO = 16, N = CO * HO;
  __shared__ float red[4][7 * CO];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc[7][CO];
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[t][o] = 0.f;
  for (int f = blockIdx.x * 4 + wave; f < F; f += gridDim.x * 4) {
    const float* xf = x + (int64_t)f * H;
    const float* df = d + (int64_t)f * N;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = lane + 64 * k;
      float xt[7], dv[CO];
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const int i = 3 * j + t - 2;
        xt[t] = (j < HO && i >= 0 && i < H) ? xf[i] : 0.f;
      }
#pragma unroll
      for (int o = 0; o < CO; ++o) dv[o] = j < HO ? df[o * HO + j] : 0.f;
#pragma unroll
      for (int t = 0; t < 7; ++t)
#pragma unroll
        for (int o = 0; o < CO; ++o) acc[t][o] += xt[t] * dv[o];
    }
  }
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int o = 0; o < CO; ++o) {
      const float r = wave_sum(acc[t][o]);
      if (lane == 0) red[wave][t * CO + o] = r;
    }
  __syncthreads();
  if (threadIdx.x < 7 * CO)
    part[(int64_t)blockIdx.x * (7 * CO) + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// The whole backward of encoder layer 0 in one pass, one wave per frame: LayerNorm + lrelu backward (the autodiff written out
// above k_ln_bwd_fused below), the weight gradient and the four per-channel parameter sums.  The layer's pre-LN output is
// RECOMPUTED from the frame's 2 KB input row (7 FMAs per element) instead of read back (10.9 KB per frame), and the
// gradient of the pre-LN output never goes to memory: HBM traffic = d(activated output) once + x, against two reads +
// one write in the LayerNorm pass and two more reads in the weight-gradient pass.
//   phase 1 (lane = positions j, j + 64, j + 128; 16 channels in registers): a = b + W x ; xhat, dn = dy * lrelu'(n) ;
//            two wave reductions ; du -> the wave's LDS tile ; d gamma / d beta / d bias sums carried per lane;
//   phase 2 (lane = channel o = lane & 15, quarter of the positions): dW[t][o] += x[3j + t - 2] * du[o][j] from LDS,
//            seven accumulators per lane over all frames of the wave.
// No workgroup barrier inside the frame loop (a wave only touches its own LDS tile).
#ifndef VAENPVC_ENC0_BLOCKED
#define VAENPVC_ENC0_BLOCKED 1
#endif
struct Enc0BwdCfg {
  static constexpr int H = 513, HO = 171, CO = 16, N = CO * HO, DP = 172, XS = 528, WAVE_FLOATS = CO * DP + XS;
  static constexpr int LDS_BYTES = 4 * WAVE_FLOATS * 4, NW = 7 * CO, NC = 3 * CO, JQ = 43;
};
__global__ void __launch_bounds__(256, 2) k_enc0_bwd_wave(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ st, const float* __restrict__ W,
                                                          const float* __restrict__ bias, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ part_w,
                                                          float* __restrict__ part_c, int F) {
  using E = Enc0BwdCfg;
  constexpr int H = E::H, HO = E::HO, CO = E::CO, N = E::N, DP = E::DP;
  extern __shared__ __attribute__((aligned(16))) float e0lds[];
  __shared__ float red[4][E::NW + E::NC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* du = e0lds + wave * E::WAVE_FLOATS;   // [16][172]
  float* xs = du + CO * DP;                    // xs[i] = x[i - 2], zero outside
  for (int i = lane; i < E::XS; i += 64) xs[i] = 0.f;
  const int wo = lane & 15, wq = lane >> 4;    // phase 2: channel, quarter of the positions
  float acc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) acc[t] = 0.f;
  float su[CO], sw[CO], sd[CO];
#pragma unroll
  for (int o = 0; o < CO; ++o) su[o] = sw[o] = sd[o] = 0.f;
  for (int f = blockIdx.x * 4 + wave; f < F; f += gridDim.x * 4) {
    const float* xf = x + (int64_t)f * H;
    const float* df = dy + (int64_t)f * N;
    const float mean = st[2 * f], rstd = st[2 * f + 1];
    float dn[3][CO], xh[3][CO];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = lane + 64 * k;
#pragma unroll
      for (int o = 0; o < CO; ++o) dn[k][o] = j < HO ? df[o * HO + j] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int i = lane + 64 * k;
      if (i < H) xs[2 + i] = xf[i];
    }
    float s1 = 0.f, s2 = 0.f;
#if VAENPVC_ENC0_BLOCKED
    // The 160 layer parameters are uniform values.  All live at once they do not fit the scalar file: the compiler parked them in
    // vector-register lanes and fetched them back with v_readlane -- 782 of the frame loop's 2 343 vector instructions.  Walking the
    // channels in blocks of four through LAUNDERED pointers (the compiler cannot hoist the loads out of the frame loop, so only one
    // block's 40 parameters are live, as scalar loads from the constant cache) leaves the vector ALUs to the arithmetic.
    float xt[3][7];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = lane + 64 * k;
#pragma unroll
      for (int t = 0; t < 7; ++t) xt[k][t] = xs[(j < HO ? 3 * j : 0) + t];
    }
#pragma unroll
    for (int ob = 0; ob < CO; ob += 4) {
      typedef const __attribute__((address_space(4))) float* cptr;   // constant address space: uniform loads become scalar loads
      unsigned long long w0 = (unsigned long long)W, b0 = (unsigned long long)bias, g0 = (unsigned long long)gamma, t0 = (unsigned long long)beta;
      asm volatile("" : "+s"(w0), "+s"(b0), "+s"(g0), "+s"(t0));
      const cptr Wp = (cptr)w0, bp = (cptr)b0, gp = (cptr)g0, btp = (cptr)t0;
      float Wl[7][4], bl[4], gl[4], bel[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bl[q] = bp[ob + q];
        gl[q] = gp[ob + q];
        bel[q] = btp[ob + q];
#pragma unroll
        for (int t = 0; t < 7; ++t) Wl[t][q] = Wp[t * CO + ob + q];
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int j = lane + 64 * k;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int o = ob + q;
          float a0 = bl[q];
#pragma unroll
          for (int t = 0; t < 7; ++t) a0 += Wl[t][q] * xt[k][t];
          const float xv = j < HO ? (a0 - mean) * rstd : 0.f;
          const float nn = xv * gl[q] + bel[q];
          const float dv = dn[k][o] * (nn >= 0.f ? 1.0f : LEAK);
          const float dx = dv * gl[q];
          s1 += dx;
          s2 += dx * xv;
          dn[k][o] = dv;
          xh[k][o] = xv;
        }
      }
    }
#else
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = lane + 64 * k;
      float xt[7];
#pragma unroll
      for (int t = 0; t < 7; ++t) xt[t] = xs[(j < HO ? 3 * j : 0) + t];
#pragma unroll
      for (int o = 0; o < CO; ++o) {
        // (the 160 layer parameters are uniform values: the compiler keeps what the scalar file cannot hold in vector-register
        //  lanes; reading them back from LDS as broadcasts instead was 45 % slower)
        float a0 = bias[o];
#pragma unroll
        for (int t = 0; t < 7; ++t) a0 += W[t * CO + o] * xt[t];
        const float xv = j < HO ? (a0 - mean) * rstd : 0.f;
        const float nn = xv * gamma[o] + beta[o];
        const float dv = dn[k][o] * (nn >= 0.f ? 1.0f : LEAK);
        const float dx = dv * gamma[o];
        s1 += dx;
        s2 += dx * xv;
        dn[k][o] = dv;
        xh[k][o] = xv;
      }
    }
#endif
    s1 = wave_sum(s1) * (1.0f / N);
    s2 = wave_sum(s2) * (1.0f / N);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = lane + 64 * k;
      if (j < HO) {
#pragma unroll
        for (int o = 0; o < CO; ++o) {
          const float d = rstd * (dn[k][o] * gamma[o] - s1 - xh[k][o] * s2);
          du[o * DP + j] = d;
          su[o] += dn[k][o] * xh[k][o];
          sw[o] += dn[k][o];
          sd[o] += d;
        }
      }
    }
    // phase 2: the frame's weight-gradient contribution from the wave's own tile (LDS operations of a wave execute in order)
    const int j0 = wq * E::JQ, j1 = j0 + E::JQ < HO ? j0 + E::JQ : HO;
    const float* dr = du + wo * DP;
    for (int j = j0; j < j1; ++j) {
      const float dv = dr[j];
#pragma unroll
      for (int t = 0; t < 7; ++t) acc[t] += xs[3 * j + t] * dv;
    }
  }
  // the quarters of a channel meet (lanes o, o + 16, o + 32, o + 48), then the four waves
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    float r = acc[t];
    r += __shfl_xor(r, 16);
    r += __shfl_xor(r, 32);
    if (lane < CO) red[wave][t * CO + lane] = r;
  }
#pragma unroll
  for (int o = 0; o < CO; ++o) {
    const float u = wave_sum(su[o]), w = wave_sum(sw[o]), d = wave_sum(sd[o]);
    if (lane == 0) {
      red[wave][E::NW + o] = u;
      red[wave][E::NW + CO + o] = w;
      red[wave][E::NW + 2 * CO + o] = d;
    }
  }
  __syncthreads();
  const int tid = threadIdx.x;
  if (tid < E::NW + E::NC) {
    const float r = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    if (tid < E::NW) part_w[(int64_t)blockIdx.x * E::NW + tid] = r;
    else part_c[(int64_t)blockIdx.x * E::NC + (tid - E::NW)] = r;
  }
}
// out[col] += sum over rows of part[row][col]  (one workgroup per column)
__global__ void __launch_bounds__(256) k_colsum_part(const float* __restrict__ part, int rows, int cols, float* __restrict__ out) {
  __shared__ float sm[4];
  const int col = blockIdx.x;
  float s = 0.f;
  for (int r = threadIdx.x; r < rows; r += 256) s += part[(int64_t)r * cols + col];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out + col, (sm[0] + sm[1]) + (sm[2] + sm[3]));
}

// ---------------------------------------------------------------- LayerNorm + lrelu backward
// autodiff of util/layers.py:32-44,149 for one layer with C channels x H positions:
//   n = gamma*xhat+beta ; dn = dy*(n>=0 ? 1 : leak) ; dxh = dn*gamma
//   da = rstd*(dxh - mean(dxh) - xhat*mean(dxh*xhat))
//   d gamma[c] += sum dn*xhat ; d beta[c] += sum dn ; d bias[c] += sum da
// HBM-bound: dy and a are read ONCE, da written ONCE.  A workgroup walks `fchunk` frames; each
// thread owns the element slots i = tid + 256k of every frame (coalesced rows), keeps dn and
// xhat of the current frame in registers across the one block reduction per frame, and carries
// the three per-element sums in registers over the whole chunk; they are reduced per channel
// through LDS once at the end (3*C global atomics per workgroup).
template <int C_, int H_>
struct LnbCfg {
  static constexpr int C = C_, H = H_, N = C * H;
  static constexpr int EPT = cdiv(N, 256);
  static constexpr int LDS_BYTES = 3 * N * 4;
};

// dy and da are NOT __restrict__: encoder layer 0's pass runs in place (dy == da) behind the fused backward of layer 1
// (gfx950_layers.hip).  In place is safe by construction -- a thread reads its elements of frame f before the frame's barrier
// and writes the same elements after it -- and the frame loop's barrier already keeps the next frame's loads behind these stores.
template <class L>
__global__ void __launch_bounds__(256) k_ln_bwd_fused(const float* dy, const float* __restrict__ a,
                                                      const float* __restrict__ st, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* da,
                                                      float* __restrict__ part, int F, int fchunk,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                      float* __restrict__ dbias) {  // non-null: accumulate directly (few workgroups)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[2][4][2];
  constexpr int N = L::N, H = L::H, C = L::C, EPT = L::EPT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fb = blockIdx.x * fchunk, fe = min(F, fb + fchunk);
  float g[EPT], bt[EPT], su[EPT], sw[EPT], sd[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    int i = tid + 256 * k;
    int c = i < N ? i / H : 0;
    g[k] = gamma[c];
    bt[k] = beta[c];
    su[k] = sw[k] = sd[k] = 0.f;
  }
  for (int f = fb; f < fe; ++f) {
    const float mean = st[2 * f], rstd = st[2 * f + 1];
    const float* pd = dy + (int64_t)f * N;
    const float* pa = a + (int64_t)f * N;
    float dn[EPT], xh[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      int i = tid + 256 * k;
      dn[k] = i < N ? ld_nt<VAENPVC_NT_A>(pd + i) : 0.f;
      xh[k] = i < N ? ld_nt<VAENPVC_NT_A>(pa + i) : mean;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      xh[k] = (xh[k] - mean) * rstd;
      float nn = xh[k] * g[k] + bt[k];
      dn[k] = dn[k] * (nn >= 0.f ? 1.0f : LEAK);
      float dx = dn[k] * g[k];
      s1 += dx;
      s2 += dx * xh[k];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const int par = f & 1;
    if (lane == 0) {
      red[par][wave][0] = s1;
      red[par][wave][1] = s2;
    }
    __syncthreads();
    s1 = ((red[par][0][0] + red[par][1][0]) + (red[par][2][0] + red[par][3][0])) * (1.0f / N);
    s2 = ((red[par][0][1] + red[par][1][1]) + (red[par][2][1] + red[par][3][1])) * (1.0f / N);
    float* po = da + (int64_t)f * N;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      int i = tid + 256 * k;
      float d = rstd * (dn[k] * g[k] - s1 - xh[k] * s2);
      if (i < N) {
        st_nt<VAENPVC_NT_A && VAENPVC_NT_AS>(po + i, d);
        su[k] += dn[k] * xh[k];
        sw[k] += dn[k];
        sd[k] += d;
      }
    }
  }
  float* eU = lds;
  float* eW = lds + N;
  float* eD = lds + 2 * N;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    int i = tid + 256 * k;
    if (i < N) {
      eU[i] = su[k];
      eW[i] = sw[k];
      eD[i] = sd[k];
    }
  }
  __syncthreads();
  if constexpr (H >= 32) {
    for (int c = wave; c < C; c += 4) {
      float u = 0.f, w = 0.f, d = 0.f;
      for (int h = lane; h < H; h += 64) {
        u += eU[c * H + h];
        w += eW[c * H + h];
        d += eD[c * H + h];
      }
      u = wave_sum(u);
      w = wave_sum(w);
      d = wave_sum(d);
      if (lane == 0) {
        if (dgamma) {  // uniform
          atomicAdd(dgamma + c, u);
          atomicAdd(dbeta + c, w);
          atomicAdd(dbias + c, d);
        } else {
          float* pp = part + (int64_t)blockIdx.x * (3 * C);
          pp[c] = u;
          pp[C + c] = w;
          pp[2 * C + c] = d;
        }
      }
    }
  } else {
    for (int c = tid; c < C; c += 256) {
      float u = 0.f, w = 0.f, d = 0.f;
      for (int h = 0; h < H; ++h) {
        u += eU[c * H + h];
        w += eW[c * H + h];
        d += eD[c * H + h];
      }
      if (dgamma) {  // uniform
        atomicAdd(dgamma + c, u);
        atomicAdd(dbeta + c, w);
        atomicAdd(dbias + c, d);
      } else {
        float* pp = part + (int64_t)blockIdx.x * (3 * C);
        pp[c] = u;
        pp[C + c] = w;
        pp[2 * C + c] = d;
      }
    }
  }
}

// second stage: out{0,1,2}[c] += sum over workgroups of part[wg][{0,1,2}][c]
__global__ void __launch_bounds__(256) k_ln_bwd_reduce(const float* __restrict__ part, int nwg, int C,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       float* __restrict__ dbias) {
  __shared__ float sm[4];
  const int col = blockIdx.x;  // 0 .. 3C-1
  float s = 0.f;
  for (int w = threadIdx.x; w < nwg; w += 256) s += part[(int64_t)w * (3 * C) + col];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    float* o = col < C ? dgamma + col : (col < 2 * C ? dbeta + (col - C) : dbias + (col - 2 * C));
    atomicAdd(o, t);
  }
}

// Deferred second stages: at small batches the per-layer reduction launch (4.8 us, eight per step) costs more than its
// work; the layers then write their partial rows to DISJOINT ranges of the scratch region and one launch per gradient
// bucket adds them all (k_ln_bwd_reduce_multi).
struct LnReduceEntry {
  const float* part;
  int nwg, C;
  float *dgamma, *dbeta, *dbias;
};
struct LnReduceList {
  static constexpr int MAXN = 8;
  LnReduceEntry e[MAXN];
  int n;
  int64_t used;       // floats of the scratch region handed out so far
  int64_t capacity;   // floats available
};
__global__ void __launch_bounds__(256) k_ln_bwd_reduce_multi(LnReduceList l) {
  __shared__ float sm[4];
  int col = blockIdx.x, k = 0;
  while (k + 1 < l.n && col >= 3 * l.e[k].C) {
    col -= 3 * l.e[k].C;
    ++k;
  }
  const LnReduceEntry en = l.e[k];
  if (col >= 3 * en.C) return;
  float s = 0.f;
  for (int w = threadIdx.x; w < en.nwg; w += 256) s += en.part[(int64_t)w * (3 * en.C) + col];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    float* o = col < en.C ? en.dgamma + col : (col < 2 * en.C ? en.dbeta + (col - en.C) : en.dbias + (col - 2 * en.C));
    atomicAdd(o, t);
  }
}
inline void flush_ln_reduce(LnReduceList& l, hipStream_t s) {
  if (!l.n) return;
  int cols = 0;
  for (int i = 0; i < l.n; ++i) cols += 3 * l.e[i].C;
  hipLaunchKernelGGL(k_ln_bwd_reduce_multi, dim3((unsigned)cols), dim3(256), 0, s, l);
  l.n = 0;
  l.used = 0;
}

// `defer` (may be null): the second stage is queued there instead of launched when its partial rows fit the region
template <class L>
inline void launch_ln_bwd(const float* dy, const float* a, const float* st, const float* gamma, const float* beta,
                          float* da, float* dgamma, float* dbeta, float* dbias, float* part, int F, int target_wgs,
                          hipStream_t s, LnReduceList* defer = nullptr) {
  rt().ensure_lds(reinterpret_cast<const void*>(&k_ln_bwd_fused<L>), L::LDS_BYTES);
  int fchunk = cmax(1, cdiv(F, target_wgs));
  int nwg = cdiv(F, fchunk);
  if (defer && nwg > 8 && defer->n < LnReduceList::MAXN && defer->used + (int64_t)nwg * 3 * L::C <= defer->capacity) {
    float* mine = part + defer->used;
    hipLaunchKernelGGL(k_ln_bwd_fused<L>, dim3((unsigned)nwg), dim3(256), L::LDS_BYTES, s, dy, a, st, gamma, beta, da, mine, F,
                       fchunk, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    defer->e[defer->n++] = LnReduceEntry{mine, nwg, L::C, dgamma, dbeta, dbias};
    defer->used += (int64_t)nwg * 3 * L::C;
    return;
  }
  if (nwg <= 8) {  // tiny batches only: same-address atomics from many workgroups serialise (256 workgroups: 104 us against 7 + 5)
    hipLaunchKernelGGL(k_ln_bwd_fused<L>, dim3((unsigned)nwg), dim3(256), L::LDS_BYTES, s, dy, a, st, gamma, beta, da, part, F,
                       fchunk, dgamma, dbeta, dbias);
    return;
  }
  hipLaunchKernelGGL(k_ln_bwd_fused<L>, dim3((unsigned)nwg), dim3(256), L::LDS_BYTES, s, dy, a, st, gamma, beta, da, part, F,
                     fchunk, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(k_ln_bwd_reduce, dim3(3 * L::C), dim3(256), 0, s, part, nwg, L::C, dgamma, dbeta, dbias);
}

// ---------------------------------------------------------------- reductions
// o[n] += sum_f d[f*ld + n] ; grid (ceil(N/256), frame chunks); up to three destinations
__global__ void __launch_bounds__(256) k_colsum_atomic(const float* __restrict__ d, int ld, int N, int F, int fchunk,
                                                       float* o1, float* o2, float* o3) {
  int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  int fb = blockIdx.y * fchunk, fe = min(F, fb + fchunk);
  float s = 0.f;
  int f = fb;
  for (; f + 8 <= fe; f += 8) {  // eight independent loads in flight
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = d[(int64_t)(f + u) * ld + n];
    s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  }
  for (; f < fe; ++f) s += d[(int64_t)f * ld + n];
  atomicAdd(o1 + n, s);
  if (o2) atomicAdd(o2 + n, s);
  if (o3) atomicAdd(o3 + n, s);
}

// out[0] += sum of all `count` floats
__global__ void __launch_bounds__(256) k_sum_all_atomic(const float* __restrict__ d, int64_t count, float* out) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) s += d[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (sm[0] + sm[1]) + (sm[2] + sm[3]));
}

// dE[y_f][k] += de[f][k] (de = columns [off, off+zd) of a [F][ld] buffer); LDS accumulation
// per frame chunk, then ny*zd global atomics per workgroup.
__global__ void __launch_bounds__(256) k_emb_grad_fast(const float* __restrict__ de, int ld, int off,
                                                       const int64_t* __restrict__ y, float* __restrict__ dE, int F,
                                                       int fchunk, int zd, int ny) {
  extern __shared__ float sm[];
  for (int i = threadIdx.x; i < ny * zd; i += 256) sm[i] = 0.f;
  __syncthreads();
  int fb = blockIdx.x * fchunk, fe = min(F, fb + fchunk);
  for (int64_t e = (int64_t)fb * zd + threadIdx.x; e < (int64_t)fe * zd; e += 256) {
    int f = (int)(e / zd), k = (int)(e - (int64_t)f * zd);
    atomicAdd(&sm[y[f] * zd + k], de[(int64_t)f * ld + off + k]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ny * zd; i += 256) {
    float v = sm[i];
    if (v != 0.f) atomicAdd(dE + i, v);
  }
}

// autodiff of the sampler + KL (see generic k_reparam_bwd); dz read with a row stride
__global__ void __launch_bounds__(256) k_reparam_bwd_ld(const float* __restrict__ dz, int ld,
                                                        const float* __restrict__ zmu, const float* __restrict__ zlv,
                                                        const float* __restrict__ eps, float* __restrict__ dzmu,
                                                        float* __restrict__ dzlv, int64_t N, int zd, float invF) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  int64_t f = i / zd;
  int k = (int)(i - f * zd);
  float mu = zmu[i], lv = zlv[i], v = expf(lv), g = dz[f * ld + k];
  dzmu[i] = g + mu / (1.0f + EPSILON) * invF;
  dzlv[i] = g * (0.5f * eps[i] * sqrtf(v)) + 0.5f * (v / (1.0f + EPSILON) - 1.0f) * invF;
}

// Same arithmetic, fused with the bias gradients of the two heads (column sums of dz_mu / dz_lv): one
// workgroup owns RF frames; thread -> (column k = tid & 127, frame phase tid >> 7); the two phases are
// combined in LDS and every workgroup issues 2 x 128 atomics.  Replaces one element-wise kernel and two
// column-sum kernels.
__global__ void __launch_bounds__(256) k_reparam_bwd_colsum(const float* __restrict__ dz, const float* __restrict__ zmu,
                                                            const float* __restrict__ zlv, const float* __restrict__ eps,
                                                            float* __restrict__ dzmu, float* __restrict__ dzlv,
                                                            float* __restrict__ gbmu, float* __restrict__ gblv, int F,
                                                            int fchunk, float invF) {
  __shared__ float sm[2][128];
  const int k = threadIdx.x & 127, ph = threadIdx.x >> 7;
  const int fb = blockIdx.x * fchunk, fe = min(F, fb + fchunk);
  float smu = 0.f, slv = 0.f;
  for (int f = fb + ph; f < fe; f += 2) {
    const int64_t i = (int64_t)f * 128 + k;
    float mu = zmu[i], lv = zlv[i], v = expf(lv), g = dz[i];
    float a = g + mu / (1.0f + EPSILON) * invF;
    float b = g * (0.5f * eps[i] * sqrtf(v)) + 0.5f * (v / (1.0f + EPSILON) - 1.0f) * invF;
    dzmu[i] = a;
    dzlv[i] = b;
    smu += a;
    slv += b;
  }
  if (ph == 1) {
    sm[0][k] = smu;
    sm[1][k] = slv;
  }
  __syncthreads();
  if (ph == 0) {
    atomicAdd(gbmu + k, smu + sm[0][k]);
    atomicAdd(gblv + k, slv + sm[1][k]);
  }
}

// ---------------------------------------------------------------- merge layer algebra (model/vae.py:51-61,89)
// h = z Wz + bz + E[y] Wy + by + b.  E[y] takes only `ny` (10) distinct values, so
//   forward : h = z Wz + T[y],  T[k] = E[k] Wy + (bz + by + b)                 (a [ny][1539] table, built per step)
//   backward: S[k] = sum_{f: y_f = k} dh[f]   (segmented column sum, [ny][1539])
//             d bz = d by = d b = sum_k S[k] ;  dWy = E^T S ;  dE = S Wy^T ;  dz = dh Wz^T (the only GEMM left)
// instead of a K = 256 forward GEMM, a gathered weight-gradient GEMM and a per-frame d(e) GEMM.
struct PackMergeTable {  // job of k_pack_multi: T[k][n], count = ny * M
  const float *E, *Wy, *bz, *by, *bm;
  int zd, M;
  __device__ float operator()(int i) const {
    const int k = i / M, n = i - k * M;
    float acc = (bz[n] + by[n]) + bm[n];
    const float* e = E + k * zd;
#pragma unroll 16
    for (int j = 0; j < zd; ++j) acc += e[j] * Wy[(int64_t)j * M + n];   // (loads of 16 taps in flight together)
    return acc;
  }
};

// S[y_f][n] += d[f][n]: grid (ceil(N/256), frame chunks); the speaker of a frame is block-uniform, the ny
// partial sums of a thread's column live in registers (uniform switch), ny atomics per thread at the end
template <int NY>
__global__ void __launch_bounds__(256) k_segsum_atomic(const float* __restrict__ d, const int64_t* __restrict__ y, int N, int F,
                                                       int fchunk, float* __restrict__ S) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int fb = blockIdx.y * fchunk, fe = min(F, fb + fchunk);
  float acc[NY];
#pragma unroll
  for (int k = 0; k < NY; ++k) acc[k] = 0.f;
  const int nn = n < N ? n : N - 1;
  for (int f0 = fb; f0 < fe; f0 += 8) {
    float v[8];
    int yk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int f = f0 + u < fe ? f0 + u : fe - 1;
      v[u] = f0 + u < fe ? d[(int64_t)f * N + nn] : 0.f;
      int64_t yy = y[f];
      yk[u] = (int)(yy < 0 ? 0 : (yy >= NY ? NY - 1 : yy));   // ids are clamped (vaenpvc_validate_ids reports them)
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int ku = __builtin_amdgcn_readfirstlane(yk[u]);    // block-uniform
#pragma unroll
      for (int k = 0; k < NY; ++k)
        if (ku == k) acc[k] += v[u];
    }
  }
  if (n < N) {
#pragma unroll
    for (int k = 0; k < NY; ++k)
      if (acc[k] != 0.f) atomicAdd(S + (int64_t)k * N + n, acc[k]);
  }
}

// everything the merge backward derives from S (plain stores: nothing else writes these gradients):
//   blocks [0, nb_w)        : dWy[m][n] = sum_k E[k][m] S[k][n]        (one thread per element)
//   blocks [nb_w, +nb_e)    : dE[k][m]  = sum_n S[k][n] Wy[m][n]       (one wave per element)
//   blocks [.., +nb_b)      : the three bias gradients sum_k S[k][n]
template <int NY>
__global__ void __launch_bounds__(256) k_merge_small(const float* __restrict__ S, const float* __restrict__ E,
                                                     const float* __restrict__ Wy, int zd, int M, float* __restrict__ dWy,
                                                     float* __restrict__ dE, float* __restrict__ db1, float* __restrict__ db2,
                                                     float* __restrict__ db3, int nb_w, int nb_e) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b < nb_w) {
    const int i = b * 256 + tid;
    if (i >= zd * M) return;
    const int m = i / M, n = i - m * M;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NY; ++k) acc += E[k * zd + m] * S[(int64_t)k * M + n];
    dWy[i] = acc;
  } else if (b < nb_w + nb_e) {
    const int o = (b - nb_w) * 4 + (tid >> 6), lane = tid & 63;
    if (o >= NY * zd) return;
    const int k = o / zd, m = o - k * zd;
    float acc = 0.f;
    for (int n = lane; n < M; n += 64) acc += S[(int64_t)k * M + n] * Wy[(int64_t)m * M + n];
    acc = wave_sum(acc);
    if (lane == 0) dE[o] = acc;
  } else {
    const int n = (b - nb_w - nb_e) * 256 + tid;
    if (n >= M) return;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NY; ++k) acc += S[(int64_t)k * M + n];
    db1[n] = acc;
    db2[n] = acc;
    db3[n] = acc;
  }
}

// [a | b] concatenation (two bias vectors of the encoder heads -> one bias row of the fused dense layer)
struct PackCat2 {
  const float *a, *b;
  int n;
  __device__ float operator()(int i) const { return i < n ? a[i] : b[i - n]; }
};

// ---------------------------------------------------------------- weight packing
template <class Fn>
__global__ void __launch_bounds__(256) k_pack(Fn fn, float* __restrict__ dst, int count) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < count) dst[i] = fn(i);
}
template <class Fn>
inline void launch_pack(Fn fn, float* dst, int count, hipStream_t s) {
  hipLaunchKernelGGL(k_pack<Fn>, dim3((unsigned)cdiv(count, 256)), dim3(256), 0, s, fn, dst, count);
}

// All weight-packing jobs of a step in ONE launch (they are ~17 tiny kernels otherwise, each a few
// microseconds of launch gap): job k owns the block range [sum_{j<k} blocks(j), +blocks(k)).
template <class Fn>
struct PackJob {
  Fn fn;
  float* dst;
  int count;
  __device__ void run(int i) const { dst[i] = fn(i); }
};
template <class Fn>
inline PackJob<Fn> pack_job(Fn fn, float* dst, int count) { return PackJob<Fn>{fn, dst, count}; }
template <class J0, class... Js>
__device__ __forceinline__ void run_pack_jobs(int blk, const J0& j0, const Js&... js) {
  const int nb = (j0.count + 255) / 256;
  if (blk < nb) {
    int i = blk * 256 + threadIdx.x;
    if (i < j0.count) j0.run(i);
  } else if constexpr (sizeof...(Js) > 0) {
    run_pack_jobs(blk - nb, js...);
  }
}
template <class... Js>
__global__ void __launch_bounds__(256) k_pack_multi(Js... js) { run_pack_jobs((int)blockIdx.x, js...); }
template <class... Js>
inline void launch_pack_multi(hipStream_t s, Js... js) {
  int blocks = 0;
  ((blocks += (js.count + 255) / 256), ...);
  hipLaunchKernelGGL(k_pack_multi<Js...>, dim3((unsigned)blocks), dim3(256), 0, s, js...);
}

// packed B of a ConvCfg from a TF kernel tensor.
//   transposed == false: src[(t*KC + k)*N + n]     transposed == true: src[(t*N + n)*KC + k]
template <class C>
struct PackConv {
  const float* src;
  bool transposed;
  __device__ float operator()(int i) const {
    int ph = 0;
    if (C::NPH > 1 && i >= C::boff(1)) ph = 1;
    if (C::NPH > 2 && i >= C::boff(2)) ph = 2;
    int r = i - (ph == 0 ? 0 : (ph == 1 ? C::boff(1) : C::boff(2)));
    int row = r / C::NP, n = r - row * C::NP;
    int tau = row / C::KCP, k = row - tau * C::KCP;
    if constexpr (C::PM) {  // column = 4*channel + phase
      ph = n & 3;
      n >>= 2;
      if (ph >= C::S) return 0.f;
    }
    int t = C::TYPEP ? ph + C::S * tau : tau;
    if (t >= C::T || k >= C::KC || n >= C::N) return 0.f;
    return transposed ? src[((int64_t)t * C::N + n) * C::KC + k] : src[((int64_t)t * C::KC + k) * C::N + n];
  }
};

// packed B [KP][NP] of a dense layer built from two row- or column-stacked matrices
//   colcat : B[k][n] = n <  n1 ? s1[k*n1 + n] : s2[k*n2 + (n-n1)]          (k < K)
//   rowcat : B[k][n] = k <  k1 ? s1[k*N + n]  : s2[(k-k1)*N + n]           (n < N)
//   colcatT: B[k][n] = transpose of colcat(K'=N, N'=K): k <  n1 ? s1[n*n1 + k] : s2[n*n2 + (k-n1)]
//   rowcatT: B[k][n] = transpose of rowcat: n < k1 ? s1[n*Nsrc + k] : s2[(n-k1)*Nsrc + k]
struct PackDense {
  const float* s1;
  const float* s2;
  int mode;  // 0 colcat, 1 rowcat, 2 colcatT, 3 rowcatT
  int K, N, NP, split, n2;  // logical K x N of the packed matrix; split = n1 or k1
  __device__ float operator()(int i) const {
    int k = i / NP, n = i - k * NP;
    if (k >= K || n >= N) return 0.f;
    switch (mode) {
      case 0: return n < split ? s1[(int64_t)k * split + n] : s2[(int64_t)k * n2 + (n - split)];
      case 1: return k < split ? s1[(int64_t)k * N + n] : s2[(int64_t)(k - split) * N + n];
      case 2: return k < split ? s1[(int64_t)n * split + k] : s2[(int64_t)n * n2 + (k - split)];
      default: return n < split ? s1[(int64_t)n * K + k] : s2[(int64_t)(n - split) * K + k];
    }
  }
};

// channel-major, zero padded copy of the 1025-tap kernel: Wc[c][8 + t] = W[t][c]  (row = 1040 floats)
struct PackToep {
  const float* src;
  __device__ float operator()(int i) const {
    int c = i / 1040, r = i - c * 1040 - 8;
    return (r >= 0 && r < 1025) ? src[r * 8 + c] : 0.f;
  }
};

}  // namespace tuned
}  // namespace vaenpvc
