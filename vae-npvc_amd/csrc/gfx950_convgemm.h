// gfx950_convgemm.h -- implicit-GEMM engine for every 1-D conv / conv_transpose /
// dense layer of the ConvVAE in forward and input-gradient direction.
//
//   out[f, n, pos_out(r)] = bias[n] + sum_{tau, k} B[tau][k][n] * in'[f, k, pos_in(r, tau)]
//
//   S-type (strided correlation; conv forward, conv_transpose input-gradient, dense):
//        rows r = output positions j ; pos_in = S*j - PAD + tau ; pos_out = j
//   P-type (transposed, stride S, split into S phases; conv_transpose forward, conv
//        input-gradient): phase ph covers outputs p = S*q + ph - PAD, taps t = ph + S*tau,
//        pos_in = q - tau ; pos_out = p
//
// GEMM view per phase: M = (frame, r) rows, N = out channels, K = (tau, k).
//   * one workgroup owns TF whole frames: their input tile is staged ONCE into LDS
//     (coalesced HBM read, LN+lrelu applied on load, zero halos so the k-loop has no
//     bounds checks); A fragments are gathered from LDS with im2col addressing
//     (ds_read_b32, stride-RS rows -> conflict free);
//   * B (weights, pre-packed [K][NP] by gfx950_prep) is streamed from L2 straight into
//     the MFMA operand register: lanes 0..31 read 128 contiguous bytes; software
//     prefetch one U-step chunk ahead;
//   * v_mfma_f32_32x32x2_f32, MB x NB register blocking per wave;
//   * epilogue: accumulators are transposed through a per-wave LDS buffer so that global
//     stores run along the position axis.
#pragma once
#include "gfx950_common.h"

namespace vaenpvc {
namespace tuned {

enum { IN_PLAIN = 0, IN_LN = 1, IN_CONCAT2 = 2 };

template <int KC_, int HIN_, int N_, int HOUT_, int T_, int S_, int PAD_, bool TYPEP_, int TF_, int INKIND_,
          int LNDIV_, int MB_, int NB_>
struct ConvCfg {
  static constexpr int KC = KC_, HIN = HIN_, N = N_, HOUT = HOUT_, T = T_, S = S_, PAD = PAD_, TF = TF_;
  static constexpr bool TYPEP = TYPEP_;
  static constexpr int INKIND = INKIND_, LNDIV = LNDIV_, MB = MB_, NB = NB_;
  static constexpr int NW = 8;  // waves per workgroup (2 per SIMD)
  static constexpr int NTHR = NW * 64;
  static constexpr int TBUF = 32 * 17;  // per-wave transpose buffer (16 rows x 32 columns, padded)
  // K per tap is padded so that KH = KCP/2 k-steps split into uniform chunks of U steps
  static constexpr int KCP = KC % 16 == 0 ? KC : rup(KC, 8), KH = KCP / 2;
  static constexpr int U = KH % 8 == 0 ? 8 : 4;  // k-steps (of 2) per B prefetch chunk
  static constexpr int CPT = KH / U;             // chunks per tap
  static constexpr int NP = rup(N, 32), NT = NP / 32;
  static constexpr int NPH = TYPEP ? S : 1;
  static constexpr int ntaps(int ph) { return TYPEP ? (T - ph + S - 1) / S : T; }
  static constexpr int kt(int ph) { return ntaps(ph) * KH; }
  static constexpr int ktp(int ph) { return kt(ph); }  // already a multiple of U
  static constexpr int q0(int ph) { return TYPEP ? (PAD - ph > 0 ? cdiv(PAD - ph, S) : 0) : 0; }
  static constexpr int q1(int ph) { return TYPEP ? (HOUT - 1 + PAD - ph) / S : HOUT - 1; }
  static constexpr int rows(int ph) { return q1(ph) - q0(ph) + 1; }
  static constexpr int boff(int ph) {
    int o = 0;
    for (int p = 0; p < ph; ++p) o += ktp(p) * 2 * NP;
    return o;
  }
  static constexpr int BTOTAL = boff(NPH);  // floats of packed B
  static constexpr int HLO = TYPEP ? (cdiv(T, S) - 1) : PAD;
  static constexpr int HHI = TYPEP ? cmax(0, (HOUT - 1 + PAD) / S - (HIN - 1)) : cmax(0, S * (HOUT - 1) - PAD + T - 1 - (HIN - 1));
  static constexpr int CSTR = HLO + HIN + HHI;
  // A gathers: lane l of a fragment reads row l = (frame, r) at f*FSTR + r*RS.  Choosing
  // FSTR == rows-per-frame * RS (mod 32) makes the 32 lane addresses one arithmetic progression
  // of step RS (1 or 3, coprime with 32 banks) across frame boundaries: conflict-free.
  static constexpr int FSTR = next_mod32(KCP * CSTR, (rows(0) * (TYPEP ? 1 : S)) % 32);
  static constexpr int TILE = rup(TF * FSTR, 4);
  static constexpr int RS = TYPEP ? 1 : S, TS = TYPEP ? -1 : 1, OFF = TYPEP ? 0 : -PAD;
  static constexpr int LDS_BYTES = (TILE + NW * TBUF) * 4;
  static constexpr int mtiles(int ph) { return cdiv(TF * rows(ph), 32); }
  static constexpr int mblk(int ph) { return cdiv(mtiles(ph), MB); }
  static_assert(!TYPEP || T > S - 1, "every phase needs a tap");
};

struct ConvArgs {
  const float* in;     // [F][KC][HIN]   (IN_CONCAT2: first half  [F][KC/2])
  const float* in2;    // IN_CONCAT2: second half rows [*][KC/2]
  const int64_t* idx;  // IN_CONCAT2: optional row gather for in2 (speaker id)
  const float* st;     // IN_LN: per-frame (mean, rstd)
  const float* gamma;  // IN_LN: per channel (k / LNDIV)
  const float* beta;
  const float* Bp;    // packed weights, see ConvCfg
  const float* bias;  // [N] or nullptr
  float* out;         // [F][N][HOUT]
  int F;
};

template <class C>
__device__ __forceinline__ void conv_stage(const ConvArgs& a, float* tile, int f0) {
  const int tid = threadIdx.x;
  for (int i = tid; i < C::TILE / 4; i += C::NTHR) reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  constexpr int PER = C::KC * C::HIN;
  const int nfr = min(C::TF, a.F - f0);
  const int total = nfr * PER;
  if constexpr (C::INKIND == IN_CONCAT2) {
    constexpr int HALF = C::KC / 2;
    for (int e = tid; e < total; e += C::NTHR) {
      int f = e / PER, k = e - f * PER;
      float v;
      if (k < HALF) {
        v = a.in[(int64_t)(f0 + f) * HALF + k];
      } else {
        int64_t g = a.idx ? a.idx[f0 + f] : (int64_t)(f0 + f);
        v = a.in2[g * HALF + (k - HALF)];
      }
      tile[f * C::FSTR + k * C::CSTR + C::HLO] = v;
    }
  } else if constexpr (C::HIN >= 32) {
    // rows = (frame, channel) x HIN contiguous bins; row index is wave-uniform
    const float* src = a.in + (int64_t)f0 * PER;
    auto rowinfo = [&](int r, int& soff, int& doff, float& sc, float& sh) {
      int f = r / C::KC, k = r - f * C::KC;
      soff = r * C::HIN;
      doff = f * C::FSTR + k * C::CSTR + C::HLO;
      sc = 1.f;
      sh = 0.f;
      if constexpr (C::INKIND == IN_LN) {
        float mean = a.st[2 * (f0 + f)], rstd = a.st[2 * (f0 + f) + 1];
        sc = rstd * a.gamma[k / C::LNDIV];
        sh = a.beta[k / C::LNDIV] - mean * sc;
      }
    };
    stage_rows<C::HIN, C::NW, C::INKIND == IN_LN>(src, tile, nfr * C::KC, rowinfo);
  } else {
    auto put = [&](int e, float v) {
      int f = e / PER;
      int rem = e - f * PER;
      int k = rem / C::HIN;
      int i = rem - k * C::HIN;
      if constexpr (C::INKIND == IN_LN) {
        int ch = k / C::LNDIV;
        v = lnact_v(v, a.st[2 * (f0 + f)], a.st[2 * (f0 + f) + 1], a.gamma[ch], a.beta[ch]);
      }
      tile[f * C::FSTR + k * C::CSTR + C::HLO + i] = v;
    };
    stage_range<(PER % 4 == 0) ? 4 : 1, (PER % 4 == 0) ? 8 : 16, C::NTHR>(a.in + (int64_t)f0 * PER, total, put);
  }
  __syncthreads();
}

// One work item = (phase, block of MB row tiles, block of NB column tiles); the items of ALL
// phases form one list that is dealt round-robin to the 8 waves.
template <class C>
__device__ __forceinline__ void conv_items(const ConvArgs& a, const float* tile, float* tbuf, int f0, int nblk0,
                                           int nblk1) {
  constexpr int U = C::U, MB = C::MB, NB = C::NB, NP = C::NP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nblks = nblk1 - nblk0;
  const int it1 = C::mblk(0) * nblks;
  const int it2 = it1 + (C::NPH > 1 ? C::mblk(1) * nblks : 0);
  const int items = it2 + (C::NPH > 2 ? C::mblk(2) * nblks : 0);
  for (int it = wave; it < items; it += C::NW) {
    // ---- phase parameters (wave-uniform)
    int ph = 0, local = it;
    if (C::NPH > 1 && it >= it1) { ph = 1; local = it - it1; }
    if (C::NPH > 2 && it >= it2) { ph = 2; local = it - it2; }
    const int R = ph == 0 ? C::rows(0) : (ph == 1 ? C::rows(C::NPH > 1 ? 1 : 0) : C::rows(C::NPH > 2 ? 2 : 0));
    const int Q0 = ph == 0 ? C::q0(0) : (ph == 1 ? C::q0(C::NPH > 1 ? 1 : 0) : C::q0(C::NPH > 2 ? 2 : 0));
    const int KT = ph == 0 ? C::kt(0) : (ph == 1 ? C::kt(C::NPH > 1 ? 1 : 0) : C::kt(C::NPH > 2 ? 2 : 0));
    const int BOFF = ph == 0 ? 0 : (ph == 1 ? C::boff(C::NPH > 1 ? 1 : 0) : C::boff(C::NPH > 2 ? 2 : 0));
    const int MTILES = ph == 0 ? C::mtiles(0) : (ph == 1 ? C::mtiles(C::NPH > 1 ? 1 : 0) : C::mtiles(C::NPH > 2 ? 2 : 0));
    const int mblk = local / nblks;
    const int nblk = nblk0 + (local - mblk * nblks);
    const int nbase = nblk * NB * 32;
    int baseA[MB], rowf[MB], rowq[MB];
    bool rowok[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      int rf = (mblk * MB + mb) * 32 + l31;
      bool ok = rf < C::TF * R;
      int rr = ok ? rf : 0;
      int f = rr / R;
      int q = Q0 + (rr - f * R);
      rowf[mb] = f;
      rowq[mb] = q;
      rowok[mb] = ok && (f0 + f) < a.F;
      baseA[mb] = f * C::FSTR + lh * C::CSTR + C::HLO + q * C::RS + C::OFF;
    }
    f32x16 acc[MB][NB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = zero16();

    // K loop over chunks of U k-steps.  B (weights) streams from L2 into two ping-pong register
    // sets, always one chunk ahead of the MFMAs that consume it (no register copies, so the
    // compiler's s_waitcnt for a chunk sits a full chunk of MFMAs after its loads were issued).
    // (loads are unconditional: columns of a partial last block are clamped to the last valid
    //  tile and never stored; the prefetch may run up to two chunks past the phase's rows, which
    //  is still inside the packed-weight / parameter buffer)
    int ncol[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) ncol[nb] = (nbase + nb * 32 < NP ? nbase + nb * 32 : NP - 32) + l31;
    const float* bp0 = a.Bp + BOFF + lh * NP;
    const int nchunks = (KT / C::KH) * C::CPT;
    auto loadB = [&](float (&b)[U][NB], int chunk) {
      const float* p = bp0 + chunk * (U * 2 * NP);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[u][nb] = p[u * 2 * NP + ncol[nb]];
    };
    auto compute = [&](const float (&b)[U][NB], int chunk) {
      const int tau = chunk / C::CPT, c = chunk - tau * C::CPT;
      const int aoff = tau * C::TS + c * (U * 2 * C::CSTR);
      float av[U][MB];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[u][mb] = tile[baseA[mb] + aoff + u * 2 * C::CSTR];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma32(av[u][mb], b[u][nb], acc[mb][nb]);
    };
    float b0[U][NB], b1[U][NB];
    loadB(b0, 0);
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
      loadB(b1, chunk + 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch AHEAD of the MFMAs it overlaps
      compute(b0, chunk);
      __builtin_amdgcn_sched_barrier(0);
      loadB(b0, chunk + 2);
      __builtin_amdgcn_sched_barrier(0);
      if (chunk + 1 < nchunks) compute(b1, chunk + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue: 16-row halves of each 32x32 tile are transposed through LDS so that
    //      16 consecutive positions of one channel are stored by 16 consecutive lanes
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      if ((mblk * MB + mb) >= MTILES) continue;  // wave-uniform
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        // reader lane <-> row (half*16 + l15) of the tile
        int rf = (mblk * MB + mb) * 32 + half * 16 + l15;
        bool ok = rf < C::TF * R;
        int rr = ok ? rf : 0;
        int fr = rr / R;
        int qr = Q0 + (rr - fr * R);
        ok = ok && (f0 + fr) < a.F;
        const int opos = C::TYPEP ? (C::S * qr + ph - C::PAD) : qr;
        const int64_t obase = (int64_t)(f0 + fr) * C::N * C::HOUT + opos;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          if (nbase + nb * 32 >= NP) continue;  // wave-uniform
#pragma unroll
          for (int r8 = 0; r8 < 8; ++r8) {
            int reg = half * 8 + r8;
            tbuf[l31 * 17 + (acc_row(reg, lane) - half * 16)] = acc[mb][nb][reg];
          }
          wave_lds_sync();
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            int nl = 4 * i + lq;
            int n = nbase + nb * 32 + nl;
            float v = tbuf[nl * 17 + l15];
            if (ok && n < C::N) a.out[obase + (int64_t)n * C::HOUT] = v + (a.bias ? a.bias[n] : 0.f);
          }
          wave_lds_sync();
        }
      }
    }
  }
}

template <class C>
__global__ void __launch_bounds__(C::NTHR) k_convgemm(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* tile = lds;
  float* tbuf = lds + C::TILE + (threadIdx.x >> 6) * C::TBUF;
  const int f0 = blockIdx.x * C::TF;
  conv_stage<C>(a, tile, f0);
  constexpr int NBLK = cdiv(C::NT, C::NB);
  const int nblk0 = (int)((int64_t)NBLK * blockIdx.y / gridDim.y);
  const int nblk1 = (int)((int64_t)NBLK * (blockIdx.y + 1) / gridDim.y);
  conv_items<C>(a, tile, tbuf, f0, nblk0, nblk1);
  static_assert(C::NPH <= 3, "stride > 3 not instantiated");
}

template <class C>
inline void launch_convgemm(const ConvArgs& a, int nsplit, hipStream_t s) {
  static bool once = false;
  if (!once) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_convgemm<C>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              C::LDS_BYTES);
    once = true;
  }
  dim3 grid((unsigned)cdiv(a.F, C::TF), (unsigned)nsplit);
  hipLaunchKernelGGL(k_convgemm<C>, grid, dim3(C::NTHR), C::LDS_BYTES, s, a);
}

}  // namespace tuned
}  // namespace vaenpvc
