// gfx950_fwgrad.h -- weight gradients of the THIN conv layers, fused like gfx950_fconv.h: both operands are read from
// their canonical fp32 tensors ONCE, converted to bf16 terms on the way into LDS, and the whole gradient tile
//     dW[(t, c_b)][c_a] = sum over (frame, row j) of  B[f][S j + t][c_b] * A[f][j][c_a]
// (A = plain rows of one tensor, B = the overlapping-row view of the other; gfx950_viewconv.h: CWS) lives in the
// accumulators of ONE persistent workgroup until a single flush of atomics at the end -- no operand planes in HBM, no
// atomics in the loop, no second pass over either tensor.
//   encoder layer i:  A = d(pre-LN output of layer i) [cout][Hout],  B = lrelu(LN(output of layer i-1)) [cin][Hin]
//   decoder layer i:  A = lrelu(LN(output of layer i-1)) [cin][Hin], B = d(pre-LN output of layer i)    [cout][Hout]
// LDS geometry per frame: A as R16 rows of CPLa channels (rows >= R zero: R16 = R rounded up to the MFMA's 16 k), B
// as 3 (R16 - 1) + T rows of CPLb channels with the conv's left pad of zero rows in front: a k-chunk of 16 GEMM rows
// never crosses a frame, and the rows past R multiply zeros.  Fragments through ds_read_b64_tr_b16 (rows are the MFMA's
// k), tile accumulated transposed (rows = (t, c_b), lanes = c_a) so that the flush adds to consecutive addresses.
// Reference: autodiff of util/layers.py:56-64 and model/vae.py:96-99 with respect to the kernels.
#pragma once
#include "gfx950_fconv.h"

namespace vaenpvc {
namespace tuned {

constexpr int fw_tf(int wsite) { return wsite == CW_E1 || wsite == CW_E2 || wsite == CW_D1 || wsite == CW_D2 ? 2 : 0; }

template <int NPL, int WSITE>
struct FwCfg {
  static constexpr CwSite V = CWS[WSITE];
  static constexpr ClDesc A = CLD[V.a], B = CLD[V.b];
  static constexpr int R = V.R, R16 = rup(R, 16), T = V.T, S = 3, PAD = B.HLO;
  static constexpr int CA = A.C, CB = B.C, CPA = A.CP, CPB = B.CP;
  static constexpr int CPLA = (CPA == 32 || CPA == 64 || CPA == 128) ? CPA + 8 : CPA;
  static constexpr int CPLB = (CPB == 32 || CPB == 64 || CPB == 128) ? CPB + 8 : CPB;
  static constexpr int ROWSB = S * (R16 - 1) + T + 1;
  static constexpr int FSA = R16 * CPLA, FSB = ROWSB * CPLB;      // elements per frame
  static constexpr int TF = fw_tf(WSITE);
  static constexpr int APL = TF * FSA, BPL = TF * FSB;            // elements per plane
  static constexpr int N = T * CPB, M = V.M;                      // gradient tile: N rows (tap, channel of B) x M columns
  static constexpr int NT = cdiv(N, 32), MT = cdiv(M, 32);
  static constexpr int KSPLIT = NT <= 2 ? 2 : 1, WN = 4 / KSPLIT; // waves along N; k-chunk parity split when there are few tiles
  static constexpr int NTW = cdiv(NT, WN);                        // n tiles per wave
  static constexpr int HA = A.H, HB = B.H;
  static constexpr int LDS = NPL * (APL + BPL) * 2;
  static_assert(TF > 0 && HA == R && CA <= 64 && CB <= 32, "site not served");
};

struct FwArgs {
  const float* a_src;   // [F][CA][HA] fp32
  const float* a_st;    // LN statistics of A's tensor, or nullptr (A is a gradient)
  const float* a_gamma;
  const float* a_beta;
  const float* b_src;   // [F][CB][HB] fp32
  const float* b_st;
  const float* b_gamma;
  const float* b_beta;
  float* dW;            // [N][M] atomicAdd (the TF kernel tensor)
  int F;
};

#ifndef VAENPVC_FW_THIRDS
#define VAENPVC_FW_THIRDS 1   // 0: lane = position for every tensor (A/B)
#endif
// staging of one operand: items = (frame of the group, 64-position chunk) dealt round-robin to the waves (see k_fconv)
template <int NPL, int C, int CP, int CPL, int H, int TF, int FS, int ROW0, int PLANE, bool BF = false>   // BF: the tensor is stored as bf16 (act_pitch rows)
struct FwStage {
  static constexpr int NCH = cdiv(H, 64), NIT = TF * NCH, IPW = cdiv(NIT, 4);
  // THIRDS (round 5): tensors with few positions and many channels (encoder layer 2's gradient: 64 x 19) are staged with lane = (position,
  // channel third) -- 57 active lanes walking 24 channels -- instead of lane = position: 19 active lanes walking 64 (as gfx950_fconv_r.h)
  static constexpr bool THIRDS = VAENPVC_FW_THIRDS && H < 32 && 3 * H <= 64 && CP >= 24 && !BF;
  static constexpr int CG = THIRDS ? rup(cdiv(CP, 3), 8) : CP;
  float v[IPW][CG];
  float mean[IPW], rstd[IPW];
  bool ok[IPW];   // the item's frame exists (frames past the batch end are stored as zeros)
  __device__ __forceinline__ void load(const float* src, const float* st, int g, int F, int wave, int lane) {
    if constexpr (THIRDS) {
      const int p = lane % H, g3 = lane / H;
#pragma unroll
      for (int u = 0; u < IPW; ++u) {
        const int it = wave + 4 * u, f = g * TF + it;
        const bool fok = it < NIT && f < F;
        ok[u] = fok;
        const int64_t sfo = (int64_t)(fok ? f : 0) * (C * H);
        mean[u] = 0.f;
        rstd[u] = 1.f;
        if (st) {  // uniform
          mean[u] = st[2 * (fok ? f : 0)];
          rstd[u] = st[2 * (fok ? f : 0) + 1];
        }
#pragma unroll
        for (int cc = 0; cc < CG; ++cc) {
          const int c = g3 * CG + cc;
          v[u][cc] = (c < C && g3 < 3 && fok) ? src[sfo + c * H + p] : 0.f;
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const int it = wave + 4 * u, fl = it / NCH, k = it - fl * NCH;
      const int f = g * TF + fl, h = 64 * k + lane;
      const bool fok = it < NIT && f < F;
      ok[u] = fok;
      constexpr int PIN = act_pitch(BF, H);
      const int64_t sfo = (int64_t)(fok ? f : 0) * (C * PIN);
      mean[u] = 0.f;
      rstd[u] = 1.f;
      if (st) {  // uniform
        mean[u] = st[2 * (fok ? f : 0)];
        rstd[u] = st[2 * (fok ? f : 0) + 1];
      }
#pragma unroll
      for (int c = 0; c < CP; ++c) v[u][c] = (c < C && h < H && fok) ? act_ld<BF>(src, sfo + c * PIN + h) : 0.f;
    }
  }
  // frames past the batch end hold zeros in the registers and are stored too (stale rows of an earlier group must not survive)
  // xh_out (TF == 1 only): the normalised values (v - mean) rstd of the frame also go to an fp32 LDS tile [C][H] -- the LayerNorm backward of
  // THIS tensor's layer, fused behind the consumer's input gradient (gfx950_fbwd.h: LNB2), needs them and the branch of lrelu
  __device__ __forceinline__ void store(unsigned short* xs, bool ln, const float* gamma, const float* beta, int wave, int lane,
                                        float* xh_out = nullptr) {
    if constexpr (THIRDS) {
      const int p = lane % H, g3 = lane / H, cbase = g3 * CG;
#pragma unroll
      for (int u = 0; u < IPW; ++u) {
        const int it = wave + 4 * u;
        if (!(it < NIT && g3 < 3)) continue;
        if (ln && ok[u]) {
#pragma unroll
          for (int cc = 0; cc < CG; ++cc)
            if (cbase + cc < C) v[u][cc] = lnact_v(v[u][cc], mean[u], rstd[u], gamma[cbase + cc], beta[cbase + cc]);
        }
        unsigned short* dx = xs + it * FS + (ROW0 + p) * CPL + cbase;
#pragma unroll
        for (int g8 = 0; g8 < CG / 8; ++g8) {
          if (cbase + 8 * g8 >= CPL) continue;   // (the last third's tail beyond the padded row)
          float v8[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v8[j] = v[u][8 * g8 + j];
          u32x4 pk[NPL];
          pack8<NPL>(v8, pk);
#pragma unroll
          for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x4*>(dx + q * PLANE + 8 * g8) = pk[q];
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const int it = wave + 4 * u, fl = it / NCH, k = it - fl * NCH;
      const int h = 64 * k + lane;
      if (!(it < NIT && h < H)) continue;
      if (xh_out) {
#pragma unroll
        for (int c = 0; c < C; ++c) xh_out[c * H + h] = (v[u][c] - mean[u]) * rstd[u];
      }
      if (ln && ok[u]) {
#pragma unroll
        for (int c = 0; c < C; ++c) v[u][c] = lnact_v(v[u][c], mean[u], rstd[u], gamma[c], beta[c]);
      }
      unsigned short* dx = xs + fl * FS + (ROW0 + h) * CPL;
#pragma unroll
      for (int g8 = 0; g8 < CP / 8; ++g8) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = v[u][8 * g8 + j];
        u32x4 pk[NPL];
        pack8<NPL>(v8, pk);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dx + p * PLANE + 8 * g8) = pk[p];
      }
    }
  }
};

__device__ __forceinline__ u32x4 tr_read8_2(const unsigned short* p0, const unsigned short* p1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p0));
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 ab = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(u32x4, ab);
}

template <int NPL, int WSITE>
__global__ void __launch_bounds__(256, 2) k_fwgrad(FwArgs a) {
  using T = FwCfg<NPL, WSITE>;
  extern __shared__ __attribute__((aligned(16))) unsigned short gsm[];
  unsigned short* as = gsm;                   // [NPL][APL]
  unsigned short* bs = gsm + NPL * T::APL;    // [NPL][BPL]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lh = lane >> 5;
  const int ngroups = cdiv(a.F, T::TF);
  FwStage<NPL, T::CA, T::CPA, T::CPLA, T::HA, T::TF, T::FSA, 0, T::APL> sa;
  FwStage<NPL, T::CB, T::CPB, T::CPLB, T::HB, T::TF, T::FSB, T::PAD, T::BPL> sb;
  // LayerNorm parameters of the activation operand, copied once (a fetch per group through the argument pointers otherwise)
  __shared__ float lnpa[2][FwCfg<NPL, WSITE>::CPA], lnpb[2][FwCfg<NPL, WSITE>::CPB];
  if (a.a_st && tid < T::CA) {
    lnpa[0][tid] = a.a_gamma[tid];
    lnpa[1][tid] = a.a_beta[tid];
  }
  if (a.b_st && tid < T::CB) {
    lnpb[0][tid] = a.b_gamma[tid];
    lnpb[1][tid] = a.b_beta[tid];
  }
  int g = blockIdx.x;
  if (g < ngroups) {
    sa.load(a.a_src, a.a_st, g, a.F, wave, lane);
    sb.load(a.b_src, a.b_st, g, a.F, wave, lane);
  }
  {  // once: zero both tiles (pad rows, rows past R, channel padding stay zero for the whole launch)
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < NPL * (T::APL + T::BPL) / 8; i += 256) reinterpret_cast<u32x4*>(gsm)[i] = z;
  }
  __syncthreads();
  // this wave's tiles: n tiles wn, wn + WN, ... (all m tiles), k-chunks of parity kpar when the k range is split
  const int wn = wave % T::WN, kpar = wave / T::WN;
  f32x16 acc[T::NTW][T::MT];
#pragma unroll
  for (int i = 0; i < T::NTW; ++i)
#pragma unroll
    for (int j = 0; j < T::MT; ++j) acc[i][j] = zero16();
  // transpose-read lane map (gfx950_planegemm.h: k_gemm_tn): row (lane & 15) >> 2 (+ 8 lh), then + 4; column quad
  const int trow = ((lane & 15) >> 2) + 8 * lh, tcol = 4 * (lane & 3) + 16 * ((lane >> 4) & 1);
  int bcol[T::NTW], acol[T::MT];
#pragma unroll
  for (int i = 0; i < T::NTW; ++i) {
    int n = 32 * (wn + i * T::WN) + tcol;
    n = n < T::N ? n : 0;   // columns past the K run: duplicates, masked at the flush
    bcol[i] = (n / T::CPB) * T::CPLB + n % T::CPB;
  }
#pragma unroll
  for (int j = 0; j < T::MT; ++j) {
    int m = 32 * j + tcol;
    acol[j] = m < T::CPA ? m : 0;
  }
  for (; g < ngroups; g += gridDim.x) {
    sa.store(as, a.a_st != nullptr, lnpa[0], lnpa[1], wave, lane);
    sb.store(bs, a.b_st != nullptr, lnpb[0], lnpb[1], wave, lane);
    __syncthreads();   // the group's frames are in LDS
    if (g + (int)gridDim.x < ngroups) {
      sa.load(a.a_src, a.a_st, g + gridDim.x, a.F, wave, lane);
      sb.load(a.b_src, a.b_st, g + gridDim.x, a.F, wave, lane);
    }
    // k-chunks: TF frames x R16 / 16 chunks, rows (fl, j = 16 kc' + trow (+4))
    constexpr int CPF = T::R16 / 16;
    for (int kc = kpar; kc < T::TF * CPF; kc += T::KSPLIT) {
      const int fl = kc / CPF, j0 = (kc - fl * CPF) * 16 + trow;
      const unsigned short* pa0 = as + fl * T::FSA + j0 * T::CPLA;
      const unsigned short* pb0 = bs + fl * T::FSB + (T::S * j0) * T::CPLB;
      u32x4 fa[T::MT][NPL], fb[T::NTW][NPL];
#pragma unroll
      for (int j = 0; j < T::MT; ++j)
#pragma unroll
        for (int p = 0; p < NPL; ++p) fa[j][p] = tr_read8_2(pa0 + p * T::APL + acol[j], pa0 + p * T::APL + acol[j] + 4 * T::CPLA);
#pragma unroll
      for (int i = 0; i < T::NTW; ++i)
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          fb[i][p] = tr_read8_2(pb0 + p * T::BPL + bcol[i], pb0 + p * T::BPL + bcol[i] + 4 * T::S * T::CPLB);
      using PR = Prod<NPL>;
#pragma unroll
      for (int t = 0; t < PR::N; ++t)
#pragma unroll
        for (int i = 0; i < T::NTW; ++i)
#pragma unroll
          for (int j = 0; j < T::MT; ++j) acc[i][j] = mfma_bf16(fb[i][PR::B[t]], fa[j][PR::A[t]], acc[i][j]);
    }
    __syncthreads();   // all fragment reads of this group are done before the next one overwrites the tiles
  }
  // flush: rows = n (tap, channel of B), lanes = m: consecutive addresses of dW[n][m]
  const int l31 = lane & 31;
#pragma unroll
  for (int i = 0; i < T::NTW; ++i)
#pragma unroll
    for (int j = 0; j < T::MT; ++j) {
      const int m = 32 * j + l31;
      if (m >= T::M) continue;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int n = 32 * (wn + i * T::WN) + acc_row(reg, lane);
        if (n < T::N && wn + i * T::WN < T::NT) atomicAdd(a.dW + n * T::M + m, acc[i][j][reg]);
      }
    }
}

template <int NPL, int WSITE>
static void launch_fwgrad(const FwArgs& a, hipStream_t s) {
  using T = FwCfg<NPL, WSITE>;
  rt().ensure_lds(reinterpret_cast<const void*>(&k_fwgrad<NPL, WSITE>), T::LDS);
  const unsigned grid = (unsigned)cmin_(cdiv(a.F, T::TF), T::LDS > 80 * 1024 ? 256 : 512);
  hipLaunchKernelGGL((k_fwgrad<NPL, WSITE>), dim3(grid), dim3(256), T::LDS, s, a);
}
constexpr bool fwgrad_serves(int wsite) { return fw_tf(wsite) > 0; }
template <int NPL>
static bool fwgrad(int wsite, const FwArgs& a, hipStream_t s) {
  switch (wsite) {
    case CW_E1: launch_fwgrad<NPL, CW_E1>(a, s); return true;
    case CW_E2: launch_fwgrad<NPL, CW_E2>(a, s); return true;
    case CW_D1: launch_fwgrad<NPL, CW_D1>(a, s); return true;
    case CW_D2: launch_fwgrad<NPL, CW_D2>(a, s); return true;
  }
  return false;
}

}  // namespace tuned
}  // namespace vaenpvc
