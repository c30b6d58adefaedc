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
//   * the weights are the MFMA "A" operand and the gathered positions the "B" operand, so that
//     accumulator rows are channels and lanes are positions: results are stored straight from
//     the accumulators along the (contiguous) position axis.
#pragma once
#include "gfx950_common.h"
#include "gfx950_stage.h"

// Compile-time ablation switch for kernel experiments (scripts/build_variant.sh); 0 in the product.
//   1 = staging + barriers only, 2 = MFMA work only (tile staged once), 3 = no output stores
#ifndef VAENPVC_ABL
#define VAENPVC_ABL 0
#endif

#ifndef VAENPVC_UMAX
#define VAENPVC_UMAX 8
#endif
#ifndef VAENPVC_APF
#define VAENPVC_APF 0
#endif

#ifndef VAENPVC_PROF
#define VAENPVC_PROF 0
#endif

namespace vaenpvc {
namespace tuned {

#if VAENPVC_PROF
// developer instrumentation (variant builds only): per-kernel-instance cycle sums of the phases
// of every wave: [slot][0..9] = waves, gload, setup, kloop, epilogue, barrier1, lstore, barrier2, total
__device__ unsigned long long g_conv_prof[32][10];
inline int g_conv_prof_next = 0;
#define PROF_T(var) const long long var = (long long)__builtin_amdgcn_s_memtime()
#else
#define PROF_T(var)
#endif

constexpr int max_div_le(int n, int lim) {
  int best = 1;
  for (int d = 1; d <= lim; ++d)
    if (n % d == 0) best = d;
  return best;
}

enum { IN_PLAIN = 0, IN_LN = 1, IN_CONCAT2 = 2 };

enum { CONV_S = 0, CONV_P = 1, CONV_PM = 2 };
//   CONV_PM ("phase in M", for transposed layers with few output channels): ONE pass with all
//   ceil(T/S) taps; the S phases become extra output columns (column = 4*channel + phase, weights
//   of taps beyond T are zero), rows are q = (p + PAD) div S.  For N = 8 this needs 2.3x fewer
//   MFMAs than three separate phases padded to 32 columns, and every lane ends up holding S
//   consecutive output positions of a channel (contiguous stores instead of stride-S).
template <int KC_, int HIN_, int N_, int HOUT_, int T_, int S_, int PAD_, int KIND_, int TF_, int INKIND_,
          int LNDIV_, int MB_, int NB_, int NW_ = 8>
struct ConvCfg {
  static constexpr int KC = KC_, HIN = HIN_, N = N_, HOUT = HOUT_, T = T_, S = S_, PAD = PAD_, TF = TF_;
  static constexpr int KIND = KIND_;
  static constexpr bool TYPEP = KIND_ != CONV_S;   // transposed addressing (pos_in = q - tau)
  static constexpr bool PM = KIND_ == CONV_PM;
  static_assert(!PM || S_ <= 4, "phase slot is 2 bits");
  static constexpr int INKIND = INKIND_, LNDIV = LNDIV_, MB = MB_, NB = NB_;
  static constexpr int NW = NW_;  // waves per workgroup (8 = 2 per SIMD; 4 for small tiles, more workgroups per CU)
  static constexpr int NTHR = NW * 64;
  static constexpr int TBUF = 0;        // (no transpose buffer: direct epilogue)
  // K per tap is padded so that KH = KCP/2 k-steps split into uniform chunks of U steps
  static constexpr int KCP = KC % 16 == 0 ? KC : rup(KC, 8), KH = KCP / 2;
  static constexpr int U = max_div_le(KH, VAENPVC_UMAX);  // k-steps (of 2) per prefetch chunk
  static constexpr int CPT = KH / U;             // chunks per tap
  static constexpr int NE = PM ? 4 * N : N;  // GEMM columns (PM: 4 phase slots per channel)
  static constexpr int NP = rup(NE, 32), NT = NP / 32;
  static constexpr int NPH = (TYPEP && !PM) ? S : 1;
  static constexpr int ntaps(int ph) { return PM ? cdiv(T, S) : (TYPEP ? (T - ph + S - 1) / S : T); }
  static constexpr int kt(int ph) { return ntaps(ph) * KH; }
  static constexpr int ktp(int ph) { return kt(ph); }  // already a multiple of U
  static constexpr int q0(int ph) { return PM ? PAD / S : (TYPEP ? (PAD - ph > 0 ? cdiv(PAD - ph, S) : 0) : 0); }
  static constexpr int q1(int ph) { return PM ? (HOUT - 1 + PAD) / S : (TYPEP ? (HOUT - 1 + PAD - ph) / S : HOUT - 1); }
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
  // small weight sets stay resident in LDS for the lifetime of the (persistent) workgroup: the
  // k-loop then issues no vector-memory loads at all, so the next tile's prefetch (VMEM, returns
  // in order) is never waited on before the staging point.  + 2 chunks of slack for the
  // over-running prefetch of the fragment ping-pong.  The bias table (NP floats) is always in LDS.
  static constexpr bool BLDS = BTOTAL * 4 <= 28 * 1024;
  static constexpr int BSM = BLDS ? BTOTAL + 2 * U * 2 * NP : 0;
  static constexpr int LDS_BYTES = (TILE + NP + BSM) * 4;
  static_assert(INKIND != IN_LN || LNDIV == 1, "conv layers normalise per channel");
  // LDS allows two resident workgroups -> ask the compiler for <= 128 VGPRs (4 waves per SIMD)
  static constexpr int WPE = cmin_c(4, cmax_c(1, cmin_c(8, (160 * 1024) / LDS_BYTES) * NW / 4));
  static constexpr int mtiles(int ph) { return cdiv(TF * rows(ph), 32); }
  static constexpr int mblk(int ph) { return cdiv(mtiles(ph), MB); }
  static_assert(!TYPEP || T > S - 1, "every phase needs a tap");
};

struct __attribute__((packed, aligned(4))) packed3 {
  float x, y, z;
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
#if VAENPVC_PROF
  int slot;
#endif
};

// One work item = (phase, block of MB row tiles, block of NB column tiles); the items of ALL
// phases form one list that is dealt round-robin to the 8 waves.
template <class C>
__device__ __forceinline__ void conv_items(const ConvArgs& a, const float* tile, const float* lbias, const float* lB,
                                           int f0, int nblk0, int nblk1
#if VAENPVC_PROF
                                           , long long (&pc)[8]
#endif
                                           ) {
  constexpr int U = C::U, MB = C::MB, NB = C::NB, NP = C::NP;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nblks = nblk1 - nblk0;
  const int it1 = C::mblk(0) * nblks;
  const int it2 = it1 + (C::NPH > 1 ? C::mblk(1) * nblks : 0);
  const int items = it2 + (C::NPH > 2 ? C::mblk(2) * nblks : 0);
  // items are dealt to the waves in boustrophedon order (round 0: waves 0..7, round 1: waves 7..0, ...): the
  // phases of a transposed conv have different tap counts (3/2/2), and a plain round-robin gives the waves
  // that drew the heavy first-phase items a second item while others hold one light one
  for (int it0 = 0, rnd = 0; it0 < items; it0 += C::NW, ++rnd) {
    const int it = it0 + ((rnd & 1) ? C::NW - 1 - wave : wave);
    if (it >= items) continue;
    PROF_T(p0);
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
    // accumulators start from the bias of their row (= output column; LDS table, zero where there
    // is no bias): the epilogue is pure stores (a load there would serialise on vmcnt with them)
    const int nbase = nblk * NB * 32;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      f32x16 bv;
      const int cb = (nbase + nb * 32 < NP ? nbase + nb * 32 : NP - 32) + 4 * lh;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) bv[reg] = lbias[cb + (reg & 3) + 8 * (reg >> 2)];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = bv;
    }

    // K loop over chunks of U k-steps.  B (weights) streams from L2 into two ping-pong register
    // sets, always one chunk ahead of the MFMAs that consume it (no register copies, so the
    // compiler's s_waitcnt for a chunk sits a full chunk of MFMAs after its loads were issued).
    // (loads are unconditional: columns of a partial last block are clamped to the last valid
    //  tile and never stored; the prefetch may run up to two chunks past the phase's rows, which
    //  is still inside the packed-weight / parameter buffer)
    int ncol[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) ncol[nb] = (nbase + nb * 32 < NP ? nbase + nb * 32 : NP - 32) + l31;
    const float* bp0 = (C::BLDS ? lB : a.Bp) + BOFF + lh * NP;
    const int nchunks = (KT / C::KH) * C::CPT;
    auto loadB = [&](float (&b)[U][NB], int chunk) {
      const float* p = bp0 + chunk * (U * 2 * NP);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[u][nb] = p[u * 2 * NP + ncol[nb]];
    };
#if VAENPVC_APF
    auto loadA = [&](float (&av)[U][MB], int chunk) {
      const int tau = chunk / C::CPT, c = chunk - tau * C::CPT;
      const int aoff = tau * C::TS + c * (U * 2 * C::CSTR);
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) av[u][mb] = tile[baseA[mb] + aoff + u * 2 * C::CSTR];
    };
    auto mm = [&](const float (&b)[U][NB], const float (&av)[U][MB]) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma32(b[u][nb], av[u][mb], acc[mb][nb]);
    };
    float b0[U][NB], b1[U][NB], a0[U][MB], a1[U][MB];
    loadB(b0, 0);
    loadA(a0, 0);
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
      loadB(b1, chunk + 1);
      loadA(a1, chunk + 1 < nchunks ? chunk + 1 : chunk);
      __builtin_amdgcn_sched_barrier(0);
      mm(b0, a0);
      __builtin_amdgcn_sched_barrier(0);
      loadB(b0, chunk + 2);
      loadA(a0, chunk + 2 < nchunks ? chunk + 2 : chunk);
      __builtin_amdgcn_sched_barrier(0);
      if (chunk + 1 < nchunks) mm(b1, a1);
      __builtin_amdgcn_sched_barrier(0);
    }
#else
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
          for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = mfma32(b[u][nb], av[u][mb], acc[mb][nb]);  // rows = channels, lanes = positions
    };
    float b0[U][NB], b1[U][NB];
    PROF_T(p1);
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
#endif
#if VAENPVC_PROF
    asm volatile("s_nop 0" ::: "memory");
#endif
    PROF_T(p2);
    // ---- epilogue: the weights were fed as the MFMA "A" operand, so accumulator ROWS are output
    //      channels and the 32 LANES of a half-wave are 32 consecutive (frame, position) rows:
    //      every register is stored directly, 128 contiguous bytes per half-wave for S-type layers.
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#if VAENPVC_ABL == 3
      if (a.F > 0) continue;
#endif
      if ((mblk * MB + mb) >= MTILES) continue;  // wave-uniform
      const int opos = C::PM ? (C::S * rowq[mb] - C::PAD) : (C::TYPEP ? (C::S * rowq[mb] + ph - C::PAD) : rowq[mb]);
      float* op = a.out + (int64_t)(f0 + rowf[mb]) * C::N * C::HOUT + opos;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (nbase + nb * 32 >= NP) continue;  // wave-uniform
        if constexpr (C::PM) {
          // registers 4g..4g+2 of a lane = phases 0..2 = three consecutive output positions of
          // channel n: one 12-byte store when the triple is inside the row
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int n = ((nbase + nb * 32) >> 2) + 2 * g + lh;
            float* o3 = op + n * C::HOUT;
            const bool okn = rowok[mb] && n < C::N;
            if (okn && opos >= 0 && opos + C::S <= C::HOUT) {
              if constexpr (C::S == 3) {
                *reinterpret_cast<packed3*>(o3) = packed3{acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1], acc[mb][nb][4 * g + 2]};
              } else {
#pragma unroll
                for (int phs = 0; phs < C::S; ++phs) o3[phs] = acc[mb][nb][4 * g + phs];
              }
            } else if (okn) {
#pragma unroll
              for (int phs = 0; phs < C::S; ++phs)
                if (opos + phs >= 0 && opos + phs < C::HOUT) o3[phs] = acc[mb][nb][4 * g + phs];
            }
          }
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) {
            int cn = nbase + nb * 32 + acc_row(reg, lane);
            if (rowok[mb] && (C::N % 32 == 0 || cn < C::N)) op[cn * C::HOUT] = acc[mb][nb][reg];
          }
        }
      }
    }
#if VAENPVC_PROF
    asm volatile("s_nop 0" ::: "memory");
    PROF_T(p3);
    pc[1] += p1 - p0;
    pc[2] += p2 - p1;
    pc[3] += p3 - p2;
#endif
  }
}

// Persistent kernel: workgroup b walks the frame tiles b, b+gridDim.x, ...  The global loads of
// the next tile are issued before the MFMA work of the current one and land in registers; they
// are written to the (single) LDS tile after the compute phase, two barriers per tile.
template <class C>
__global__ void __launch_bounds__(C::NTHR, C::WPE) k_convgemm(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using St = TileStager<C::KC, C::KC, C::HIN, C::CSTR, C::FSTR, C::HLO, C::INKIND == IN_LN, C::TF, C::NW>;
  float* tile = lds;
  float* lbias = lds + C::TILE;
  float* lB = lbias + C::NP;
  const int tid = threadIdx.x;
  for (int i = tid; i < C::TILE / 4; i += C::NTHR) reinterpret_cast<float4*>(tile)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int c = tid; c < C::NP; c += C::NTHR) {
    int n = C::PM ? (c >> 2) : c;
    lbias[c] = (a.bias && n < C::N) ? a.bias[n] : 0.f;
  }
  if constexpr (C::BLDS)
    for (int i = tid; i < C::BSM; i += C::NTHR) lB[i] = i < C::BTOTAL ? a.Bp[i] : 0.f;
  constexpr int NBLK = cdiv(C::NT, C::NB);
  const int nblk0 = (int)((int64_t)NBLK * blockIdx.y / gridDim.y);
  const int nblk1 = (int)((int64_t)NBLK * (blockIdx.y + 1) / gridDim.y);
  const int tiles = cdiv(a.F, C::TF);
  St st;
  st.init(tile, a.gamma, a.beta, 0, C::KC);
  int t = blockIdx.x;
#if VAENPVC_PROF
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  PROF_T(k0);
#endif
  st.gload(a.in, a.st, t * C::TF, min(C::TF, a.F - t * C::TF), 0, C::KC);
  __syncthreads();  // zero fill (halos, padded channel rows), bias table and resident weights complete
  st.lstore(tile, min(C::TF, a.F - t * C::TF), C::KC);
  __syncthreads();
  for (; t < tiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
#if VAENPVC_PROF
    PROF_T(q0);
    if (tn < tiles) st.gload(a.in, a.st, tn * C::TF, min(C::TF, a.F - tn * C::TF), 0, C::KC);
    __builtin_amdgcn_sched_barrier(0);
    PROF_T(q1);
    conv_items<C>(a, tile, lbias, lB, t * C::TF, nblk0, nblk1, pc);
    PROF_T(q2);
    __syncthreads();
    PROF_T(q3);
    if (tn < tiles) st.lstore(tile, min(C::TF, a.F - tn * C::TF), C::KC);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    PROF_T(q4);
    __syncthreads();
    PROF_T(q5);
    pc[0] += q1 - q0;
    pc[4] += q3 - q2;
    pc[5] += q4 - q3;
    pc[6] += q5 - q4;
#else
#if VAENPVC_ABL != 2
    if (tn < tiles) st.gload(a.in, a.st, tn * C::TF, min(C::TF, a.F - tn * C::TF), 0, C::KC);
#endif
    __builtin_amdgcn_sched_barrier(0);
#if VAENPVC_ABL != 1
    conv_items<C>(a, tile, lbias, lB, t * C::TF, nblk0, nblk1);
#endif
    __syncthreads();
#if VAENPVC_ABL != 2
    if (tn < tiles) st.lstore(tile, min(C::TF, a.F - tn * C::TF), C::KC);
#endif
    __syncthreads();
#endif
  }
#if VAENPVC_PROF
  {
    PROF_T(k1);
    if ((threadIdx.x & 63) == 0) {
      unsigned long long* g = g_conv_prof[a.slot];
      atomicAdd(g + 0, 1ull);
      for (int i = 0; i < 7; ++i) atomicAdd(g + 1 + i, (unsigned long long)pc[i]);
      atomicAdd(g + 8, (unsigned long long)(k1 - k0));
    }
  }
#endif
  static_assert(C::NPH <= 3, "stride > 3 not instantiated");
}

template <class C>
inline void launch_convgemm(const ConvArgs& a, int nsplit, hipStream_t s) {
  rt().ensure_lds(reinterpret_cast<const void*>(&k_convgemm<C>), C::LDS_BYTES);
  // persistent grid: exactly as many workgroups as are resident at once (LDS- or VGPR-bound); a property of
  // the kernel binary, queried once (thread-safe static initialisation)
  static const int per_cu = [] {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&k_convgemm<C>), C::NTHR,
                                                     C::LDS_BYTES) != hipSuccess || n < 1)
      n = 1;
    return n;
  }();
  const int resident = cmax(1, 256 * per_cu / nsplit);
  dim3 grid((unsigned)cmin_(cdiv(a.F, C::TF), resident), (unsigned)nsplit);
#if VAENPVC_PROF
  static int slot = -1;
  if (slot < 0) {
    slot = g_conv_prof_next++;
    fprintf(stderr, "PROF slot %d per_cu %d grid %u : %s\n", slot, per_cu, grid.x, __PRETTY_FUNCTION__);
  }
  ConvArgs ap = a;
  ap.slot = slot;
  hipLaunchKernelGGL(k_convgemm<C>, grid, dim3(C::NTHR), C::LDS_BYTES, s, ap);
  return;
#endif
  hipLaunchKernelGGL(k_convgemm<C>, grid, dim3(C::NTHR), C::LDS_BYTES, s, a);
}

}  // namespace tuned
}  // namespace vaenpvc
