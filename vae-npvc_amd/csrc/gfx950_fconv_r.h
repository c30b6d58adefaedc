// gfx950_fconv_r.h -- the MEDIUM conv sites (64 - 96 GEMM rows, K = 192 - 288) fused like gfx950_fconv.h, with the
// weights in REGISTERS instead of LDS: a wave owns one 32-row tile of the weight matrix for the whole launch (its A
// fragments for every k-step: K/16 x NPL x 4 registers) and multiplies it with every row step of the frames the
// workgroup has staged; LDS holds only the frames (8 per group), so two workgroups fit a CU although the weight matrix
// alone (100 - 130 KB as two planes) would not fit beside them.
//   * transposed-conv sites stack their 3 output phases PER 8 CHANNELS (tile row = phase * 8 + channel % 8, 24 of 32
//     rows used): the three phases of a (channel, position) sit in three registers of one lane and leave as one
//     12-byte store (see k_fconv); every wave has a tile;
//   * tensors with few positions and many channels (19 x 64, 19 x 81) are staged with lanes = (position, channel
//     third) instead of lane = position.
// Sites: decoder layer 0 forward / input gradient, encoder layer 2 forward / input gradient.  Up to two planes.
// Reference: util/layers.py:56-64, model/vae.py:96-99 and their autodiff.
#pragma once
#include "gfx950_fconv.h"

namespace vaenpvc {
namespace tuned {

constexpr bool fcr_serves_site(int site) { return site == CV_D0F || site == CV_D0G || site == CV_E2F || site == CV_E2G; }
// frames per group: 4 where the weight tile takes 136 - 144 registers with two planes (one staging item per wave)
// OTL (round 5): S-type sites park their result frames in an LDS tile ([TF][OC * OH] fp32, the canonical order) and the workgroup copies the
// group out as ONE contiguous run of 16-byte pieces.  Straight from the accumulators these sites stored 4 bytes per lane in runs of OH = 19
// floats (76 bytes: every store instruction touched ~5 partial lines); the same re-ordering took the plane GEMMs' result stores from 64 to 16
// instructions per lane (gfx950_planegemm.h).  Sites: decoder layer 0 input gradient, encoder layer 2 forward (4 frames per group there:
// frames + tile stay under 80 KB, two workgroups per CU).
#ifndef VAENPVC_FCR_CLO_ORDER
#define VAENPVC_FCR_CLO_ORDER 2   // (281 -> 274 us for decoder layer 0 forward; 1: 278)
#endif
#ifndef VAENPVC_FCR_OTL
#define VAENPVC_FCR_OTL 1   // bit 0: the S-type sites (dec0 dgrad 167 -> 155 us, enc2 fwd 119 -> 117), bit 1: decoder layer 0 forward as well (phase-stacked; measured
                            // SLOWER, 276 -> 291 us: its 12-byte stores are not what bounds it, the tile's LDS traffic and the later store issue cost more)
#endif
constexpr bool fcr_otl(int site, int npl) { return VAENPVC_FCR_OTL && npl <= 2 && (site == CV_D0G || site == CV_E2F || (site == CV_D0F && (VAENPVC_FCR_OTL & 2))); }
constexpr int fcr_tf(int site, int npl) {
  return site == CV_D0F ? (npl == 1 ? 8 : 4) : site == CV_D0G ? (npl == 1 && !fcr_otl(site, npl) ? 6 : 4) : site == CV_E2G ? 8 : (fcr_otl(site, npl) ? 4 : 6);
}

template <int NPL, int SITE>
struct FrCfg {
  static constexpr CvSite V = CVS[SITE];
  static constexpr ClDesc X = CLD[V.x];
  static constexpr int C = X.C, H = X.H, CP = X.CP, HLO = X.HLO, HP = X.HP;
  static constexpr bool PERM = V.PH != 0;             // phase-stacked per 8 channels
  static constexpr int MTR = PERM ? V.O / 8 : cdiv(V.M, 32);   // 32-row tiles of the (permuted) weight matrix
  static constexpr int MP = MTR * 32;
  static constexpr int WPT = MTR >= 4 ? 1 : MTR == 3 ? 1 : 4 / MTR;   // waves per tile (3 tiles: the fourth wave only stages)
  static constexpr int CPL = (CP == 32 || CP == 64 || CP == 128) ? CP + 8 : (CP == 88 ? 104 : CP);
  static constexpr int FS = HP * CPL;
  static constexpr int TF = fcr_tf(SITE, NPL);
  static constexpr int XPL = TF * FS + 64;
  static constexpr int K = V.NT * CP, KS = cdiv(K, 16);
  static constexpr int RSTEP = (V.step / CP) * CPL;
  static constexpr bool OTL = fcr_otl(SITE, NPL);
  static constexpr int OFR = V.OC * V.OH;                       // floats per result frame
  static constexpr int OPT4 = rup(OFR, 4);                      // tile row pitch when the rows leave one by one (POUT: 16-byte aligned rows)
  static constexpr int LDS_X = NPL * XPL * 2, LDS = LDS_X + (OTL ? TF * OPT4 * 4 : 0);
  static_assert(!OTL || (LDS_X % 16 == 0 && (TF * OFR) % 4 == 0), "result tile: 16-byte aligned, whole groups are whole pieces");
  // staging: lane = position (H >= 32) or lane = (position, channel third) (H < 32)
  static constexpr bool POS = H >= 32;
  static constexpr int NCH = POS ? cdiv(H, 64) : 1;
  static constexpr int CG = POS ? CP : rup(cdiv(CP, 3), 8);   // channels a lane collects
  static constexpr int NIT = TF * NCH, IPW = cdiv(NIT, 4);
  static_assert(KS * 16 <= V.Kp && (POS || (3 * H <= 64 && 3 * CG <= CPL + 8 && C > 2 * CG)), "site not served");
};

// weight planes with the rows of FrCfg: plain sites as cv_job; PERM: row = tile * 32 + phase * 8 + (channel % 8)
template <int NPL>
struct PackViewPermJob {
  WView w;
  unsigned short* dst;
  int Mp, Kp;
  int count;   // Mp * Kp / 8
  __device__ void run(int i) const {
    const int k8 = Kp >> 3, m = i / k8, k0 = (i - m * k8) << 3;
    const int tile = m >> 5, r = m & 31, pim = r >> 3, o = tile * 8 + (r & 7);
    const int dd = k0 / w.CP, c0 = k0 - dd * w.CP;
    const int tap = w.S * (w.NT - 1 - dd) + pim;
    const bool ok = pim < w.S && o < w.O && dd < w.NT && tap >= 0 && tap < w.T;
    const float* src = w.W + (int64_t)(ok ? tap : 0) * w.s_t + o * w.s_o;
    float v8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v8[j] = (ok && c0 + j < w.C) ? src[(c0 + j) * w.s_c] : 0.f;
    u32x4 pk[NPL];
    pack8<NPL>(v8, pk);
#pragma unroll
    for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dst + (size_t)p * Mp * Kp + (size_t)m * Kp + k0) = pk[p];
  }
};
constexpr int fcr_mp(int site) { return CVS[site].PH ? (CVS[site].O / 8) * 32 : CVS[site].Mp; }
constexpr int fcr_wfloats(int site) { return 3 * fcr_mp(site) * CVS[site].Kp / 2; }
template <int NPL>
static PackViewPermJob<NPL> fcr_perm_job(int site, const float* W, int s_t, int s_o, int s_c, float* dst) {
  const CvSite& v = CVS[site];
  const ClDesc& x = CLD[v.x];
  const int mp = fcr_mp(site);
  return PackViewPermJob<NPL>{WView{W, s_t, s_o, s_c, v.T, v.NT, v.S, v.PH, v.mdiv, v.O, x.C, x.CP},
                              reinterpret_cast<unsigned short*>(dst), mp, v.Kp, mp * v.Kp / 8};
}

// OSP (encoder layer 2 forward with the result tile in LDS): statistics of the result + its activated channel-last planes (FcArgs::st2_out ...)
// POUT (decoder layer 0 input gradient with the result tile in LDS): the result rows leave as bf16 operand planes (FcArgs::pl_out), no fp32 copy
template <int NPL, int SITE, int LN, bool CLO = false, bool PIN = false, bool OSP = false, bool POUT = false>
__global__ void __launch_bounds__(256, 2) k_fconv_r(FcArgs a) {
  using T = FrCfg<NPL, SITE>;
  constexpr CvSite V = T::V;
  static_assert(!OSP || (T::OTL && !T::PERM && T::TF == 4 && V.O <= 64), "result statistics: one wave per frame of the LDS tile");
  static_assert(!POUT || (T::OTL && !T::PERM && !OSP), "operand planes out: from the canonical result tile");
  constexpr int OPT = POUT ? T::OPT4 : T::OFR;   // floats per frame of the result tile
  extern __shared__ __attribute__((aligned(16))) unsigned short rsm[];
  unsigned short* xs = rsm;   // [NPL][XPL]
  __shared__ float lnp[2][FrCfg<NPL, SITE>::C];   // LayerNorm parameters of the input (a fetch per group through the pointers otherwise)
  __shared__ float lno[2][OSP ? 64 : 1];          // ... of the result (OSP)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int ngroups = cdiv(a.F, T::TF);
  // ---- staging registers (one group ahead, as k_fconv)
  float v[T::IPW][T::CG];
  float mean[T::IPW], rstd[T::IPW];
  const int pg_p = lane % (T::POS ? 64 : T::H), pg_g = T::POS ? 0 : lane / T::H;   // (position, channel third) of this lane
  // PIN: the group's frames are one contiguous run of 16-byte pieces in every plane (halo rows included, zero there)
  constexpr int PG8 = T::CP / 8, PPF = T::HP * PG8, PPT = PIN ? cdiv(T::TF * PPF, 256) : 1;
  u32x4 pv[PIN ? NPL : 1][PPT];
  auto fload = [&](int g) __attribute__((always_inline)) {
    if constexpr (PIN) {
      const int last = min(T::TF, a.F - g * T::TF) * PPF - 1;
      const unsigned short* b = a.cl_in + (int64_t)g * (T::TF * PPF * 8);
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int r = min(tid + 256 * u, last);   // (clamped address, masked at the LDS store: a select here would wait for its load)
#pragma unroll
        for (int p = 0; p < NPL; ++p) pv[p][u] = *reinterpret_cast<const u32x4*>(b + p * a.cl_plane + r * 8);
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < T::IPW; ++u) {
      const int it = wave + 4 * u, fl = it / T::NCH, k = it - fl * T::NCH;
      const int f = g * T::TF + fl;
      const bool fok = it < T::NIT && f < a.F;
      const float* sf = a.src + (int64_t)(fok ? f : 0) * (T::C * T::H);
      if constexpr (LN == 1) {
        mean[u] = a.st[2 * (fok ? f : 0)];
        rstd[u] = a.st[2 * (fok ? f : 0) + 1];
      }
      if constexpr (T::POS) {
        const int h = 64 * k + lane;
#pragma unroll
        for (int c = 0; c < T::CG; ++c) v[u][c] = (c < T::C && h < T::H && fok) ? sf[c * T::H + h] : 0.f;
      } else {
        // every lane loads UNCONDITIONALLY, base pointer + compile-time offset, from a valid address of a valid frame and the store below
        // zeroes what is not live: the channel slots past C of the last third are read one third lower (bpB), idle lanes read the first
        // third.  As predicated loads (`c < C && ... ? sf[..] : 0`) every element became a branch of its own with its own 64-bit offset
        // register pair, hoisted out of the group loop; in decoder layer 0's forward kernel twelve of those pairs were spilled and each came
        // back with `s_waitcnt vmcnt(0)` in front of its load -- a full memory round trip per element, eleven per group (round 5)
        constexpr int CL = T::C - 2 * T::CG;   // valid channel slots of the last third
        const int g3 = pg_g < 3 ? pg_g : 0;
        const float* bpA = sf + g3 * (T::CG * T::H) + pg_p;
        const float* bpB = g3 == 2 ? bpA - T::CG * T::H : bpA;
#pragma unroll
        for (int cc = 0; cc < T::CG; ++cc) v[u][cc] = (cc < CL ? bpA : bpB)[cc * T::H];
      }
    }
  };
  auto fstore = [&](int g) __attribute__((always_inline)) {
    if constexpr (PIN) {
      const int nok = min(T::TF, a.F - g * T::TF) * PPF;
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int r = tid + 256 * u;
        if (r >= T::TF * PPF) continue;
        const int fl = r / PPF, q = r - fl * PPF, hp = q / PG8, g8 = q - hp * PG8;
        unsigned short* dx = xs + fl * T::FS + hp * T::CPL + 8 * g8;
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dx + p * T::XPL) = r < nok ? pv[p][u] : u32x4{0u, 0u, 0u, 0u};
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < T::IPW; ++u) {
      const int it = wave + 4 * u, fl = it / T::NCH, k = it - fl * T::NCH;
      const int h = T::POS ? 64 * k + lane : pg_p;
      const int cbase = T::POS ? 0 : pg_g * T::CG;
      const bool live = it < T::NIT && g * T::TF + fl < a.F && h < T::H && (T::POS || pg_g < 3);
      if constexpr (LN == 2) {
        // LayerNorm statistics of the input taken HERE (lane = position, one item = one whole frame: the two sums are wave
        // reductions, no barrier; two-pass in registers as k_ln_stats_fast) and stored for the backward pass: the separate
        // statistics pass over the tensor goes away
        static_assert(LN != 2 || (T::POS && T::NCH == 1 && T::CG >= T::C), "statistics in the staging need one item per frame");
        constexpr float INVN = 1.0f / (T::C * T::H);
        float sm = 0.f;
#pragma unroll
        for (int cc = 0; cc < T::C; ++cc) sm += v[u][cc];        // (invalid lanes / frames hold zeros)
        mean[u] = wave_sum(sm) * INVN;
        float q = 0.f;
#pragma unroll
        for (int cc = 0; cc < T::C; ++cc) {
          const float d = v[u][cc] - mean[u];
          q += d * d;
        }
        q = wave_sum(lane < T::H ? q : 0.f);
        rstd[u] = 1.0f / sqrtf(q * INVN + LN_EPS);
        if (lane == 0 && it < T::NIT && g * T::TF + fl < a.F) {
          a.st_out[2 * (g * T::TF + fl)] = mean[u];
          a.st_out[2 * (g * T::TF + fl) + 1] = rstd[u];
        }
      }
      if (!live) continue;
      if constexpr (LN != 0) {
#pragma unroll
        for (int cc = 0; cc < T::CG; ++cc) {
          const int c = cbase + cc;
          if (c < T::C) v[u][cc] = lnact_v(v[u][cc], mean[u], rstd[u], lnp[0][c], lnp[1][c]);
        }
      }
      if constexpr (!T::POS) {   // (the clamped loads above left copies in the channel slots past C)
#pragma unroll
        for (int cc = 0; cc < T::CG; ++cc)
          if (cbase + cc >= T::C) v[u][cc] = 0.f;
      }
      unsigned short* dx = xs + fl * T::FS + (T::HLO + h) * T::CPL + cbase;
#pragma unroll
      for (int g8 = 0; g8 < T::CG / 8; ++g8) {
        if (cbase + 8 * g8 >= T::CPL) continue;   // (the last third's tail beyond the padded row)
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = v[u][8 * g8 + j];
        u32x4 pk[NPL];
        pack8<NPL>(v8, pk);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *reinterpret_cast<u32x4*>(dx + p * T::XPL + 8 * g8) = pk[p];
      }
    }
  };
  int g = blockIdx.x;
  if (g < ngroups) fload(g);
  // ---- once: zero the frame tile; this wave's weight tile into registers; the input's LayerNorm parameters into LDS
  if (LN != 0 && tid < T::C) {
    lnp[0][tid] = a.gamma[tid];
    lnp[1][tid] = a.beta[tid];
  }
  if constexpr (OSP) {
    if (tid < V.O) {
      lno[0][tid] = a.gamma2[tid];
      lno[1][tid] = a.beta2[tid];
    }
    if (blockIdx.x == 0) {   // zero tails behind the planes (as k_cl_produce)
      constexpr ClDesc D2 = CLD[CL_Y2];
      const int64_t used = (int64_t)a.F * D2.HP * D2.CP;
      for (int64_t i = used + tid; i < a.cl2_plane; i += 256)
#pragma unroll
        for (int p = 0; p < NPL; ++p) a.cl2_out[p * a.cl2_plane + i] = 0;
    }
  }
  {
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int i = tid; i < NPL * T::XPL / 8; i += 256) reinterpret_cast<u32x4*>(xs)[i] = z;
  }
  if constexpr (CLO) {
    if (blockIdx.x == 0) {   // zero tails behind the planes (as k_cl_produce)
      const int64_t used = (int64_t)a.F * T::HP * T::CP;
      for (int64_t i = used + tid; i < a.cl_plane; i += 256)
#pragma unroll
        for (int p = 0; p < NPL; ++p) a.cl_out[p * a.cl_plane + i] = 0;
    }
  }
  const int tile = wave % T::MTR, sub = wave / T::MTR;      // (3 tiles: wave 3 -> tile 0, sub 1: no steps)
  const bool gemm_wave = sub < T::WPT;
  u32x4 wreg[T::KS][NPL];
#pragma unroll
  for (int ks = 0; ks < T::KS; ++ks)
#pragma unroll
    for (int p = 0; p < NPL; ++p)
      wreg[ks][p] = *reinterpret_cast<const u32x4*>(a.W + ((size_t)p * fcr_mp(SITE) + tile * 32 + l31) * V.Kp + ks * 16 + lh * 8);
  // bias of the rows this lane stores, once per kernel: a load inside the epilogue orders every later store behind its round
  // trip (the shared in-order vector-memory counter; gfx950_fconv.h lost 20 - 30 % to exactly that until round 4)
  float bvr[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int ch = T::PERM ? tile * 8 + (k & 3) + 4 * lh : tile * 32 + acc_row(k, lane);
    bvr[k] = (a.bias && ch < V.O) ? a.bias[ch] : 0.f;
  }
  __syncthreads();
  for (; g < ngroups; g += gridDim.x) {
    const int f0 = g * T::TF, nf = min(T::TF, a.F - f0);
    fstore(g);
    __syncthreads();
    auto clo_copy = [&]() __attribute__((always_inline)) {
      // the staged image (bf16 terms, channel-last, zero halo rows) is exactly the group's frames of the planes: copied out as
      // consecutive 16-byte pieces (direct stores from the staging registers -- 16 bytes at a 176-byte stride per lane -- cost
      // as much as the separate split pass they replaced: 217 -> 314 us, same-box)
      constexpr int G8 = T::CP / 8, PPF = T::HP * G8;
      for (int i = tid; i < nf * NPL * PPF; i += 256) {
        const int fl = i / (NPL * PPF), r = i - fl * (NPL * PPF), p = r / PPF, q = r - p * PPF, hp = q / G8, g8 = q - hp * G8;
        const u32x4 vv = *reinterpret_cast<const u32x4*>(xs + p * T::XPL + fl * T::FS + hp * T::CPL + 8 * g8);
        st_nt<VAENPVC_NT_B>(reinterpret_cast<u32x4*>(a.cl_out + p * a.cl_plane + (int64_t)(f0 + fl) * (T::HP * T::CP) + (int64_t)q * 8), vv);
      }
    };
    // VAENPVC_FCR_CLO_ORDER: 0 = copy-out, then the next group's loads, then the GEMM; 1 = loads first; 2 = copy-out behind the GEMM
    if constexpr (CLO && VAENPVC_FCR_CLO_ORDER == 0) clo_copy();
    if (g + (int)gridDim.x < ngroups) fload(g + gridDim.x);
    if constexpr (CLO && VAENPVC_FCR_CLO_ORDER == 1) clo_copy();
    // CHN 32-row steps at a time = CHN independent accumulator chains sharing the wave's weight fragments (a single chain
    // leaves the matrix pipe idle for most of an MFMA's latency).  Two planes: the weight tile leaves no registers for
    // a second chain (it spilled 50 - 118 registers), one chain.
    constexpr int CHN = NPL == 1 ? 2 : 1;
    const int nrows = nf * V.R, npairs = cdiv(nrows, 32 * CHN);
    if (gemm_wave) {
      for (int sp = sub; sp < npairs; sp += T::WPT) {
        int nn[CHN], xoff[CHN];
        bool nok[CHN];
#pragma unroll
        for (int h = 0; h < CHN; ++h) {
          nn[h] = sp * (32 * CHN) + h * 32 + l31;
          nok[h] = nn[h] < nrows;
          const int n = nok[h] ? nn[h] : 0;
          const int fl = n / V.R, q = n - fl * V.R;
          xoff[h] = fl * T::FS + q * T::RSTEP;
        }
        f32x16 acc[CHN];
#pragma unroll
        for (int h = 0; h < CHN; ++h) acc[h] = zero16();
#pragma unroll
        for (int ks = 0; ks < T::KS; ++ks) {
          u32x4 fb[CHN][NPL];
          const int ko = fc_koff<T::CP, T::CPL>(ks, lh);
#pragma unroll
          for (int h = 0; h < CHN; ++h)
#pragma unroll
            for (int p = 0; p < NPL; ++p) fb[h][p] = *reinterpret_cast<const u32x4*>(xs + p * T::XPL + xoff[h] + ko);
          using PR = Prod<NPL>;
          mfma_prio<4>(true);
#pragma unroll
          for (int t = 0; t < PR::N; ++t)
#pragma unroll
            for (int h = 0; h < CHN; ++h) acc[h] = mfma_bf16(wreg[ks][PR::A[t]], fb[h][PR::B[t]], acc[h]);
          mfma_prio<4>(false);
        }
#pragma unroll
        for (int h = 0; h < CHN; ++h) {
          if (!nok[h]) continue;
          const int fl = nn[h] / V.R, q = nn[h] - fl * V.R;
          float* ob = a.out + (int64_t)(f0 + fl) * (V.OC * V.OH);
          const int pbase = q * V.oq + V.o0;
          if constexpr (T::PERM && T::OTL) {
            float* ot = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(rsm) + T::LDS_X) + fl * OPT + pbase;
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) {
              const int ch = tile * 8 + cs + 4 * lh;
              const float bb = bvr[cs];
              float* o = ot + ch * V.OH;
              if (pbase >= 0 && pbase < V.OH) o[0] = acc[h][cs] + bb;
              if (pbase + 1 >= 0 && pbase + 1 < V.OH) o[1] = acc[h][cs + 4] + bb;
              if (pbase + 2 >= 0 && pbase + 2 < V.OH) o[2] = acc[h][cs + 8] + bb;
            }
          } else if constexpr (T::PERM) {
            // tile row = phase * 8 + channel % 8: registers cs, cs + 4, cs + 8 of this lane are the three phases
            struct __attribute__((packed, aligned(4))) f3 { float x, y, z; };
            const bool inner = pbase >= 0 && pbase + 2 < V.OH;
#pragma unroll
            for (int cs = 0; cs < 4; ++cs) {
              const int ch = tile * 8 + cs + 4 * lh;
              const float bb = bvr[cs];
              const float p0 = acc[h][cs] + bb, p1 = acc[h][cs + 4] + bb, p2 = acc[h][cs + 8] + bb;
              float* o = ob + ch * V.OH + pbase;
              if (inner) {
                *reinterpret_cast<f3*>(o) = f3{p0, p1, p2};
              } else {
                if (pbase >= 0 && pbase < V.OH) o[0] = p0;
                if (pbase + 1 >= 0 && pbase + 1 < V.OH) o[1] = p1;
                if (pbase + 2 >= 0 && pbase + 2 < V.OH) o[2] = p2;
              }
            }
          } else if constexpr (T::OTL) {
            float* ot = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(rsm) + T::LDS_X) + fl * OPT + pbase;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int ch = tile * 32 + acc_row(reg, lane);
              if (ch < V.O && pbase >= 0 && pbase < V.OH) ot[ch * V.OH] = acc[h][reg] + bvr[reg];
            }
          } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
              const int ch = tile * 32 + acc_row(reg, lane);
              if (ch < V.O && pbase >= 0 && pbase < V.OH) ob[ch * V.OH + pbase] = acc[h][reg] + bvr[reg];
            }
          }
        }
      }
    }
    if constexpr (CLO && VAENPVC_FCR_CLO_ORDER == 2) clo_copy();
    __syncthreads();
    if constexpr (T::OTL) {
      // the group's result frames: one contiguous run of the output tensor (the next group's results are written behind the barrier that
      // follows its staging, so these reads need no barrier of their own)
      const float* ot = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(rsm) + T::LDS_X);
      if constexpr (POUT) {
        // a row of the tile = a row of the merge GEMMs' operand: split into its bf16 terms eight values at a time, whole padded rows
        // (pl_kp elements: the pieces behind the row's OFR values are zero) as consecutive 16-byte pieces of every plane
        const int p8 = a.pl_kp >> 3;
        constexpr int FULL = T::OFR >> 3, REM = T::OFR & 7;
        for (int i = tid; i < nf * p8; i += 256) {
          const int fl = i / p8, j = i - fl * p8;
          const float* t = ot + fl * OPT + 8 * j;
          float v8[8];
          if (j < FULL) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(t), hi = *reinterpret_cast<const f32x4*>(t + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              v8[k] = lo[k];
              v8[4 + k] = hi[k];
            }
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v8[k] = (j == FULL && k < REM) ? t[k] : 0.f;
          }
          u32x4 pk[NPL];
          pack8<NPL>(v8, pk);
#pragma unroll
          for (int p = 0; p < NPL; ++p)
            st_nt<VAENPVC_NT_B>(reinterpret_cast<u32x4*>(a.pl_out + p * a.pl_plane + (int64_t)(f0 + fl) * a.pl_kp + 8 * j), pk[p]);
        }
      } else {
      float* og = a.out + (int64_t)f0 * T::OFR;
      const int nfl = nf * T::OFR, n4 = nfl >> 2;
      for (int i = tid; i < n4; i += 256) reinterpret_cast<f32x4*>(og)[i] = reinterpret_cast<const f32x4*>(ot)[i];
      if (tid < (nfl & 3)) og[4 * n4 + tid] = ot[4 * n4 + tid];     // (ragged last group)
      }
      if constexpr (OSP) {
        // LayerNorm statistics of the result frames (two-pass, as k_ln_stats_fast) and their activated channel-last planes: wave = frame
        constexpr ClDesc D2 = CLD[CL_Y2];
        static_assert(!OSP || (D2.C == V.OC && D2.H == V.OH && D2.CP == 64), "planes of this site's result");
        constexpr int NE = T::OFR, EPL = cdiv(NE, 64), G8 = D2.CP / 8, ITEMS = D2.HP * G8;
        if (wave < nf) {
          const float* t = ot + wave * NE;
          const int f = f0 + wave;
          float sm = 0.f;
#pragma unroll
          for (int i = 0; i < EPL; ++i) sm += (lane + 64 * i < NE) ? t[lane + 64 * i] : 0.f;
          const float mean = wave_sum(sm) * (1.0f / NE);
          float q = 0.f;
#pragma unroll
          for (int i = 0; i < EPL; ++i) {
            const float d = (lane + 64 * i < NE) ? t[lane + 64 * i] - mean : 0.f;
            q += d * d;
          }
          const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / NE) + LN_EPS);
          if (lane == 0) {
            a.st2_out[2 * f] = mean;
            a.st2_out[2 * f + 1] = rstd;
          }
          unsigned short* df = a.cl2_out + (int64_t)f * (D2.HP * D2.CP);
#pragma unroll
          for (int r = 0; r < cdiv(ITEMS, 64); ++r) {
            const int it = lane + 64 * r;
            if (it >= ITEMS) continue;
            const int hp = it / G8, cg = it - hp * G8, h = hp - D2.HLO;
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int c = cg * 8 + j;
              v8[j] = (h >= 0 && h < D2.H && c < D2.C) ? lnact_v(t[c * D2.H + (h >= 0 && h < D2.H ? h : 0)], mean, rstd, lno[0][c], lno[1][c]) : 0.f;
            }
            u32x4 pk[NPL];
            pack8<NPL>(v8, pk);
#pragma unroll
            for (int p = 0; p < NPL; ++p) st_nt<VAENPVC_NT_B>(reinterpret_cast<u32x4*>(df + p * a.cl2_plane + (int64_t)it * 8), pk[p]);
          }
        }
      }
    }
  }
}

template <int NPL, int SITE>
static void launch_fconv_r(const FcArgs& a, hipStream_t s) {
  using T = FrCfg<NPL, SITE>;
  const unsigned grid = (unsigned)cmin_(cdiv(a.F, T::TF), T::LDS > 80 * 1024 ? 256 : 512);
  if constexpr (SITE == CV_D0F) {
    if (a.cl_out && !a.st) {
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv_r<NPL, SITE, 0, true>), T::LDS);
      hipLaunchKernelGGL((k_fconv_r<NPL, SITE, 0, true>), dim3(grid), dim3(256), T::LDS, s, a);
      return;
    }
  }
  if constexpr (SITE == CV_D0G && T::OTL) {
    if (a.cl_in && a.pl_out) {   // (operand planes in, operand planes of the merge GEMMs out)
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv_r<NPL, SITE, 0, false, true, false, true>), T::LDS);
      hipLaunchKernelGGL((k_fconv_r<NPL, SITE, 0, false, true, false, true>), dim3(grid), dim3(256), T::LDS, s, a);
      return;
    }
  }
  if constexpr (SITE == CV_D0G) {
    if (a.cl_in) {   // (operand planes in: straight copy into the LDS image)
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv_r<NPL, SITE, 0, false, true>), T::LDS);
      hipLaunchKernelGGL((k_fconv_r<NPL, SITE, 0, false, true>), dim3(grid), dim3(256), T::LDS, s, a);
      return;
    }
  }
  if constexpr (SITE == CV_E2F && T::OTL) {
    if (a.st_out && a.cl2_out) {   // (statistics of the input in the staging; statistics + activated planes of the result from the LDS tile)
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv_r<NPL, SITE, 2, false, false, true>), T::LDS);
      hipLaunchKernelGGL((k_fconv_r<NPL, SITE, 2, false, false, true>), dim3(grid), dim3(256), T::LDS, s, a);
      return;
    }
  }
  if constexpr (SITE == CV_E2F) {
    if (a.st_out) {   // (statistics of the input computed in the staging)
      rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv_r<NPL, SITE, 2>), T::LDS);
      hipLaunchKernelGGL((k_fconv_r<NPL, SITE, 2>), dim3(grid), dim3(256), T::LDS, s, a);
      return;
    }
  }
  if (a.st) {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv_r<NPL, SITE, 1>), T::LDS);
    hipLaunchKernelGGL((k_fconv_r<NPL, SITE, 1>), dim3(grid), dim3(256), T::LDS, s, a);
  } else {
    rt().ensure_lds(reinterpret_cast<const void*>(&k_fconv_r<NPL, SITE, 0>), T::LDS);
    hipLaunchKernelGGL((k_fconv_r<NPL, SITE, 0>), dim3(grid), dim3(256), T::LDS, s, a);
  }
}
template <int NPL>
static bool fconv_r(int site, const FcArgs& a, hipStream_t s) {
  if constexpr (NPL <= 2) {
    switch (site) {
      case CV_D0F: launch_fconv_r<NPL, CV_D0F>(a, s); return true;
      case CV_D0G: launch_fconv_r<NPL, CV_D0G>(a, s); return true;
      case CV_E2F: launch_fconv_r<NPL, CV_E2F>(a, s); return true;
      case CV_E2G: launch_fconv_r<NPL, CV_E2G>(a, s); return true;
    }
  }
  return false;
}

}  // namespace tuned
}  // namespace vaenpvc
