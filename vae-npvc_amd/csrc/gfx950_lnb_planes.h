// gfx950_lnb_planes.h -- LayerNorm + lrelu backward (k_ln_bwd_fused, gfx950_elem.h: autodiff of util/layers.py:32-44,149) whose
// result leaves the kernel AS the bf16 operand planes its consumers read, instead of as an fp32 tensor that a separate pass
// (k_split_planes / k_cl_produce) reads again to split:
//   ID <  0  plain rows [NPL][F][N]           (encoder layer 4's gradient: operand of the two dense-shaped GEMMs)
//   ID >= 0  channel-last planes of CLD[ID]   (encoder layer 3 / decoder layer 0: operands of the view GEMMs)
//   F32      the fp32 tensor is written as well (a consumer still reads it: decoder layer 0's input-gradient kernel)
// The terms are the same split_n terms the split kernels produce: bitwise the same planes.
// Channel-last needs a [C][H] -> [H][C] transposition: every thread drops its terms as 2-byte LDS stores into the frame image
// of the current parity; the image of the PREVIOUS frame -- complete once every thread has passed this frame's barrier, the one
// the reduction needs anyway -- is copied out as 16-byte pieces.  No additional barrier per frame.  The images live in the LDS
// region the per-channel reduction uses after the loop.
#pragma once
#include "gfx950_elem.h"
#include "gfx950_viewconv.h"

namespace vaenpvc {
namespace tuned {

template <class L, int NPL, int ID>
struct LnbPlCfg {
  static constexpr bool CLO = ID >= 0;
  static constexpr ClDesc D = CLD[CLO ? ID : 0];
  static constexpr int CPL = D.CP + 8;            // LDS row pitch (elements): rows stay 16-byte aligned, neighbouring positions on different banks
  static constexpr int IMG = D.HP * CPL;          // elements per plane image
  static constexpr int IMG_BYTES = CLO ? 2 * NPL * IMG * 2 : 0;
  static constexpr int LDS_BYTES = L::LDS_BYTES > IMG_BYTES ? L::LDS_BYTES : IMG_BYTES;
  static_assert(!CLO || (D.C == L::C && D.H == L::H && D.C == D.CP), "layer / plane geometry");
};

template <class L, int NPL, int ID, bool F32>
__global__ void __launch_bounds__(256) k_ln_bwd_planes(const float* __restrict__ dy, const float* __restrict__ a, const float* __restrict__ st,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ da,
                                                       unsigned short* __restrict__ pl, int64_t plane, float* __restrict__ part, int F,
                                                       int fchunk) {
  using Q = LnbPlCfg<L, NPL, ID>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float red[2][4][2];
  constexpr int N = L::N, H = L::H, C = L::C, EPT = L::EPT;
  constexpr bool CLO = Q::CLO;
  constexpr int CP = Q::D.CP, HP = Q::D.HP, HLO = Q::D.HLO, CPL = Q::CPL, IMG = Q::IMG;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fb = blockIdx.x * fchunk, fe = min(F, fb + fchunk);
  unsigned short* img = reinterpret_cast<unsigned short*>(lds);   // [2][NPL][IMG]
  float g[EPT], bt[EPT], su[EPT], sw[EPT], sd[EPT];
  int io[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int i = tid + 256 * k;
    const int c = i < N ? i / H : 0, h = i < N ? i - c * H : 0;
    g[k] = gamma[c];
    bt[k] = beta[c];
    su[k] = sw[k] = sd[k] = 0.f;
    io[k] = (HLO + h) * CPL + c;
  }
  if constexpr (CLO) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < 2 * NPL * IMG / 8; i += 256) reinterpret_cast<u32x4*>(img)[i] = z;   // (halo rows stay zero)
    if (blockIdx.x == 0) {   // zero tails behind the planes (as k_cl_produce)
      const int64_t used = (int64_t)F * HP * CP;
      for (int64_t i = used + tid; i < plane; i += 256)
#pragma unroll
        for (int p = 0; p < NPL; ++p) pl[p * plane + i] = 0;
    }
    __syncthreads();
  }
  // the image of frame f (parity f & 1) -> its [HP][CP] frame of every plane, consecutive threads = consecutive 16-byte pieces
  auto copy_out = [&](int f) __attribute__((always_inline)) {
    constexpr int G8 = CP / 8, PPP = HP * G8;
    const unsigned short* im = img + (f & 1) * NPL * IMG;
    unsigned short* df = pl + (int64_t)f * HP * CP;
    for (int it = tid; it < NPL * PPP; it += 256) {
      const int p = it / PPP, r = it - p * PPP, hp = r / G8, g8 = r - hp * G8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(im + p * IMG + hp * CPL + 8 * g8);
      st_nt<VAENPVC_NT_B>(reinterpret_cast<u32x4*>(df + p * plane + (int64_t)r * 8), v);
    }
  };
  for (int f = fb; f < fe; ++f) {
    const float mean = st[2 * f], rstd = st[2 * f + 1];
    const float* pd = dy + (int64_t)f * N;
    const float* pa = a + (int64_t)f * N;
    float dn[EPT], xh[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + 256 * k;
      dn[k] = i < N ? ld_nt<VAENPVC_NT_A>(pd + i) : 0.f;
      xh[k] = i < N ? ld_nt<VAENPVC_NT_A>(pa + i) : mean;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      xh[k] = (xh[k] - mean) * rstd;
      const float nn = xh[k] * g[k] + bt[k];
      dn[k] = dn[k] * (nn >= 0.f ? 1.0f : LEAK);
      const float dx = dn[k] * g[k];
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
    if constexpr (CLO) {
      if (f > fb) copy_out(f - 1);
    }
    float* po = da + (int64_t)f * N;
    unsigned short* im = img + par * NPL * IMG;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int i = tid + 256 * k;
      const float d = rstd * (dn[k] * g[k] - s1 - xh[k] * s2);
      if (i < N) {
        if constexpr (F32) st_nt<VAENPVC_NT_A && VAENPVC_NT_AS>(po + i, d);
        su[k] += dn[k] * xh[k];
        sw[k] += dn[k];
        sd[k] += d;
        unsigned t[NPL];
        split_n<NPL>(d, t);
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          if constexpr (CLO) im[p * IMG + io[k]] = (unsigned short)t[p];
          else pl[p * plane + (int64_t)f * N + i] = (unsigned short)t[p];
        }
      }
    }
  }
  if constexpr (CLO) {
    __syncthreads();
    copy_out(fe - 1);
    __syncthreads();   // the images are read: the region now takes the per-element sums
  }
  float* eU = lds;
  float* eW = lds + N;
  float* eD = lds + 2 * N;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int i = tid + 256 * k;
    if (i < N) {
      eU[i] = su[k];
      eW[i] = sw[k];
      eD[i] = sd[k];
    }
  }
  __syncthreads();
  float* pp = part + (int64_t)blockIdx.x * (3 * C);
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
        pp[c] = u;
        pp[C + c] = w;
        pp[2 * C + c] = d;
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
      pp[c] = u;
      pp[C + c] = w;
      pp[2 * C + c] = d;
    }
  }
}

// (two-stage parameter sums only: the caller keeps k_ln_bwd_fused for the few-workgroup and deferred-reduction cases)
template <class L, int NPL, int ID, bool F32>
inline void launch_ln_bwd_planes(const float* dy, const float* a, const float* st, const float* gamma, const float* beta, float* da,
                                 unsigned short* pl, int64_t plane, float* dgamma, float* dbeta, float* dbias, float* part, int F,
                                 int target_wgs, hipStream_t s) {
  using Q = LnbPlCfg<L, NPL, ID>;
  rt().ensure_lds(reinterpret_cast<const void*>(&k_ln_bwd_planes<L, NPL, ID, F32>), Q::LDS_BYTES);
  const int fchunk = cmax(1, cdiv(F, target_wgs));
  const int nwg = cdiv(F, fchunk);
  hipLaunchKernelGGL((k_ln_bwd_planes<L, NPL, ID, F32>), dim3((unsigned)nwg), dim3(256), Q::LDS_BYTES, s, dy, a, st, gamma, beta, da, pl, plane,
                     part, F, fchunk);
  hipLaunchKernelGGL(k_ln_bwd_reduce, dim3(3 * L::C), dim3(256), 0, s, part, nwg, L::C, dgamma, dbeta, dbias);
}

}  // namespace tuned
}  // namespace vaenpvc
