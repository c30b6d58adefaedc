// gfx950_viewconv.h -- the conv / conv_transpose layers of the ConvVAE as view GEMMs on the bf16 matrix cores
// (kernels: gfx950_planegemm.h; the idea: section "Convs as GEMMs over overlapping rows" of DESIGN.md).
//
// Every activation / gradient tensor a conv site reads is kept as CHANNEL-LAST planes with zero halo rows
// (unsigned short [NPL][F][HP][CP] + a zero tail), produced by k_cl_produce (below) from the canonical fp32 [F][C][H] tensor.
// With one plane (precision "bf16") these planes ARE bf16 activation storage: the conv sites then read 2 bytes per
// element instead of 4 and run at the bf16 MFMA rate -- the bf16 training mode of BASELINE.json config 2.
// With 2 or 3 planes the same code is the fp32-class variant (used by the parity tests to pin the indexing against the
// float64 restatement of the reference at 1e-4 / 2e-4; in those precisions the exact-fp32 engines stay the default for most sites because
// at the plane kernels' efficiency the view GEMM only wins where the channel count fills a 128-row tile).
//
// Sites (reference: util/layers.py:56-64 conv2d SAME, model/vae.py:96-99 conv2d_transpose SAME, and their autodiff):
//   S-type (strided correlation over the haloed input; HLO = pad):   out[f][o][j] = sum_{t,c} X[f][S j + t][c] W(t,o,c)
//        encoder forward (layers 1..3), decoder input gradient (layers 0..2)
//   P-type (the S output phases r stacked into the GEMM rows m = r*mdiv + o; HLO = NT - 1, NT = ceil(T/S)):
//        out[f][o][S q + r - pad] = sum_{dd<NT,c} X[f][q + dd][c] W(S (NT-1-dd) + r, o, c)      (zero for taps >= T)
//        decoder forward (layers 0..2), encoder input gradient (layers 1..3)
//   weight gradients: C[(t,c)][o] += sum_{(f,j)} A[(f,j)][(t,c)] B[(f,j)][o]  with A = S-type view rows, B = plain rows
#pragma once
#include "cl_layout.h"
#include "gfx950_planegemm.h"

namespace vaenpvc {
namespace tuned {

// ---------------------------------------------------------------- producers of the channel-last planes
// One WAVE per frame: the fp32 frame ([C][H], N = C*H floats) is read once with 16-byte loads into registers, the
// LayerNorm statistics are taken there (LN = 2: two-pass, as k_ln_stats_fast), the activated values go to the wave's
// LDS tile in the source order, and the tile is read back transposed -- 8 channels of one position per lane (row
// stride H is odd: conflict-free) -- split into bf16 terms and stored as 16-byte pieces of the contiguous [HP][CP]
// frame of every plane.  No workgroup barrier; four frames in flight per workgroup.
//   LN: 0 plain copy (gradients, merge output), 1 LayerNorm + lrelu with given statistics, 2 ... computing them here.
struct CpArgs {
  const float* src;
  const float* st;      // LN == 1
  float* st_out;        // LN == 2
  const float* gamma;
  const float* beta;
  unsigned short* dst;
  int64_t plane;
  int F;
};
template <int NPL, int LN, int ID>
__global__ void __launch_bounds__(256) k_cl_produce(CpArgs a) {
  constexpr ClDesc D = CLD[ID];
  constexpr int C = D.C, H = D.H, CP = D.CP, HLO = D.HLO, HP = D.HP, N = C * H, G8 = CP / 8;
  constexpr bool V4 = (N % 4 == 0);
  constexpr int NV = cdiv(N, 4), PER = cdiv(NV, 64);
  extern __shared__ __attribute__((aligned(16))) float tiles[];   // [4 waves][N rounded up to 4]
  constexpr int TS = NV * 4;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* tile = tiles + wv * TS;
  if (blockIdx.x == 0) {  // zero tails behind the planes
    const int64_t used = (int64_t)a.F * HP * CP;
    for (int64_t i = used + threadIdx.x; i < a.plane; i += 256)
#pragma unroll
      for (int p = 0; p < NPL; ++p) a.dst[p * a.plane + i] = 0;
  }
  for (int f = blockIdx.x * 4 + wv; f < a.F; f += gridDim.x * 4) {
    const float* sf = a.src + (int64_t)f * N;
    float v[PER][4];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = lane + 64 * i;
      if constexpr (V4) {
        const f32x4 t = idx < NV ? ld_nt<VAENPVC_NT_B>(reinterpret_cast<const f32x4*>(sf) + idx) : f32x4{0.f, 0.f, 0.f, 0.f};
        v[i][0] = t[0], v[i][1] = t[1], v[i][2] = t[2], v[i][3] = t[3];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[i][j] = 4 * idx + j < N ? sf[4 * idx + j] : 0.f;
      }
    }
    float mean = 0.f, rstd = 1.f;
    if constexpr (LN == 1) {
      mean = a.st[2 * f];
      rstd = a.st[2 * f + 1];
    }
    if constexpr (LN == 2) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
      mean = wave_sum(s) / N;
      float q = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * (lane + 64 * i) + j < N) {
            const float d = v[i][j] - mean;
            q += d * d;
          }
      rstd = 1.0f / sqrtf(wave_sum(q) / N + LN_EPS);
      if (lane == 0) {
        a.st_out[2 * f] = mean;
        a.st_out[2 * f + 1] = rstd;
      }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int idx = lane + 64 * i;
      if (idx < NV) {
        float4 o;
        if constexpr (LN != 0) {
          const int e = 4 * idx;
          o.x = lnact_v(v[i][0], mean, rstd, a.gamma[min(e / H, C - 1)], a.beta[min(e / H, C - 1)]);
          o.y = lnact_v(v[i][1], mean, rstd, a.gamma[min((e + 1) / H, C - 1)], a.beta[min((e + 1) / H, C - 1)]);
          o.z = lnact_v(v[i][2], mean, rstd, a.gamma[min((e + 2) / H, C - 1)], a.beta[min((e + 2) / H, C - 1)]);
          o.w = lnact_v(v[i][3], mean, rstd, a.gamma[min((e + 3) / H, C - 1)], a.beta[min((e + 3) / H, C - 1)]);
        } else {
          o = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
        }
        reinterpret_cast<float4*>(tile)[idx] = o;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // one item = (padded position hp, group of 8 channels); consecutive items are consecutive 16-byte pieces
    unsigned short* df = a.dst + (int64_t)f * HP * CP;
    for (int it = lane; it < HP * G8; it += 64) {
      const int hp = it / G8, cg = it - hp * G8, h = hp - HLO;
      float v8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        v8[j] = (h >= 0 && h < H && c < C) ? tile[c * H + h] : 0.f;
      }
      u32x4 pk[NPL];
      pack8<NPL>(v8, pk);
#pragma unroll
      for (int p = 0; p < NPL; ++p) st_nt<VAENPVC_NT_B>(reinterpret_cast<u32x4*>(df + p * a.plane + (int64_t)it * 8), pk[p]);
    }
    __builtin_amdgcn_wave_barrier();
  }
}
template <int NPL, int LN, int ID>
static void launch_cl_produce_id(const CpArgs& a, hipStream_t s) {
  constexpr int lds = 4 * cdiv(CLD[ID].C * CLD[ID].H, 4) * 4 * 4;
  rt().ensure_lds(reinterpret_cast<const void*>(&k_cl_produce<NPL, LN, ID>), lds);
  const unsigned blocks = (unsigned)cmin_(cdiv(a.F, 4), 2048);
  hipLaunchKernelGGL((k_cl_produce<NPL, LN, ID>), dim3(blocks), dim3(256), lds, s, a);
}
template <int NPL, int LN>
static void launch_cl_produce(int id, const CpArgs& a, hipStream_t s) {
  switch (id) {
#define VAENPVC_CLP(ID) case ID: launch_cl_produce_id<NPL, LN, ID>(a, s); break;
    VAENPVC_CLP(CL_Y0) VAENPVC_CLP(CL_Y1) VAENPVC_CLP(CL_Y2) VAENPVC_CLP(CL_H) VAENPVC_CLP(CL_YD0) VAENPVC_CLP(CL_YD1)
    VAENPVC_CLP(CL_GE1) VAENPVC_CLP(CL_GE2) VAENPVC_CLP(CL_GE3) VAENPVC_CLP(CL_GD0) VAENPVC_CLP(CL_GD1) VAENPVC_CLP(CL_GD2)
#undef VAENPVC_CLP
  }
}

// weight planes of a site: B[m][k], m = pim*mdiv + o, k = dd*CP + c; element W[t*s_t + o*s_o + c*s_c]
struct WView {
  const float* W;
  int s_t, s_o, s_c;
  int T, NT, S, PH;     // PH = 1: P-type (t = S*(NT-1-dd) + pim); 0: S-type (t = dd)
  int mdiv, O, C, CP;
  __device__ float operator()(int m, int k) const {
    const int pim = m / mdiv, o = m - pim * mdiv, dd = k / CP, c = k - dd * CP;
    if (o >= O || c >= C) return 0.f;
    int t;
    if (PH) {
      if (pim >= S || dd >= NT) return 0.f;
      t = S * (NT - 1 - dd) + pim;
    } else {
      if (pim > 0) return 0.f;
      t = dd;
    }
    return t < T ? W[(int64_t)t * s_t + o * s_o + c * s_c] : 0.f;
  }
};

// forward / input-gradient sites
enum { CV_E1F, CV_E2F, CV_E3F, CV_D0F, CV_D1F, CV_D2F, CV_E3G, CV_E2G, CV_E1G, CV_D0G, CV_D1G, CV_D2G, CV_COUNT };
struct CvSite {
  int x;                       // CL id of the operand tensor
  int Mp, Kp, M, mdiv, O;      // weight planes [Mp][Kp]; GEMM rows M; channels per phase; output channels
  int T, NT, S, PH;            // taps, taps per phase, stride, P-type flag
  int R, step;                 // view: rows per frame, elements between rows (x0 = 0)
  int OC, OH, oq, o0;          // output tensor [F][OC][OH]; pos = q*oq + o0 + phase
};
constexpr CvSite CVS[CV_COUNT] = {
    // x      Mp   Kp   M   mdiv O    T NT S PH   R    step     OC  OH   oq  o0
    {CL_Y0, 32, 128, 32, 32, 32, 7, 7, 3, 0, 57, 3 * 16, 32, 57, 1, 0},         // CV_E1F encoder layer 1 forward
    {CL_Y1, 64, 256, 64, 64, 64, 7, 7, 3, 0, 19, 3 * 32, 64, 19, 1, 0},         // CV_E2F
    {CL_Y2, 128, 448, 128, 128, 128, 7, 7, 3, 0, 7, 3 * 64, 128, 7, 1, 0},      // CV_E3F
    {CL_H, 128, 320, 96, 32, 32, 9, 3, 3, 1, 20, 88, 32, 57, 3, -3},            // CV_D0F decoder layer 0 forward (k 9, pad 3)
    {CL_YD0, 64, 128, 48, 16, 16, 7, 3, 3, 1, 58, 32, 16, 171, 3, -2},          // CV_D1F
    {CL_YD1, 32, 64, 24, 8, 8, 7, 3, 3, 1, 172, 16, 8, 513, 3, -2},             // CV_D2F
    {CL_GE3, 256, 384, 192, 64, 64, 7, 3, 3, 1, 8, 128, 64, 19, 3, -3},         // CV_E3G encoder layer 3 input gradient
    {CL_GE2, 128, 192, 96, 32, 32, 7, 3, 3, 1, 20, 64, 32, 57, 3, -2},          // CV_E2G
    {CL_GE1, 64, 128, 48, 16, 16, 7, 3, 3, 1, 58, 32, 16, 171, 3, -2},          // CV_E1G
    {CL_GD0, 128, 320, 81, 81, 81, 9, 9, 3, 0, 19, 3 * 32, 81, 19, 1, 0},       // CV_D0G decoder layer 0 input gradient
    {CL_GD1, 32, 128, 32, 32, 32, 7, 7, 3, 0, 57, 3 * 16, 32, 57, 1, 0},        // CV_D1G
    {CL_GD2, 32, 64, 16, 16, 16, 7, 7, 3, 0, 171, 3 * 8, 16, 171, 1, 0},        // CV_D2G
};
constexpr int cv_wfloats(int site) { return 3 * CVS[site].Mp * CVS[site].Kp / 2; }
constexpr int cv_woff(int site) {
  int o = 0;
  for (int i = 0; i < site; ++i) o += cv_wfloats(i);
  return o;
}
constexpr int CV_WTOTAL = cv_woff(CV_COUNT);

// weight-plane pack job of a site (a job of k_pack_multi); (s_t, s_o, s_c) address the TF kernel tensor by (tap, GEMM
// output channel, contracted channel).  One work item = 8 consecutive k of one row (they share tap and phase: CP is a
// multiple of 8): one 16-byte store per plane, the index arithmetic once per 8 elements.
template <int NPL>
struct PackViewJob {
  WView w;
  unsigned short* dst;
  int Mp, Kp;
  int count;   // Mp * Kp / 8
  __device__ void run(int i) const {
    const int k8 = Kp >> 3, m = i / k8, k0 = (i - m * k8) << 3;
    const int pim = m / w.mdiv, o = m - pim * w.mdiv, dd = k0 / w.CP, c0 = k0 - dd * w.CP;
    const int tap = w.PH ? w.S * (w.NT - 1 - dd) + pim : dd;
    const bool ok = o < w.O && tap >= 0 && tap < w.T && (w.PH ? (pim < w.S && dd < w.NT) : pim == 0);
    const float* src = w.W + (int64_t)(ok ? tap : 0) * w.s_t + o * w.s_o;
    unsigned t[8][NPL];
#pragma unroll
    for (int j = 0; j < 8; ++j) split_n<NPL>((ok && c0 + j < w.C) ? src[(c0 + j) * w.s_c] : 0.f, t[j]);
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
      u32x4 pk;
#pragma unroll
      for (int q = 0; q < 4; ++q) pk[q] = t[2 * q][p] | (t[2 * q + 1][p] << 16);
      *reinterpret_cast<u32x4*>(dst + (size_t)p * Mp * Kp + (size_t)m * Kp + k0) = pk;
    }
  }
};
template <int NPL>
static PackViewJob<NPL> cv_job(int site, const float* W, int s_t, int s_o, int s_c, float* dst) {
  const CvSite& v = CVS[site];
  const ClDesc& x = CLD[v.x];
  return PackViewJob<NPL>{WView{W, s_t, s_o, s_c, v.T, v.NT, v.S, v.PH, v.mdiv, v.O, x.C, x.CP},
                          reinterpret_cast<unsigned short*>(dst), v.Mp, v.Kp, v.Mp * v.Kp / 8};
}

template <int NPL>
static void cv_split(int id, const float* src, const float* st, const float* gamma, const float* beta, float* dst, int F,
                     hipStream_t s) {
  CpArgs a{src, st, nullptr, gamma, beta, reinterpret_cast<unsigned short*>(dst), cl_plane(id, F), F};
  if (st) launch_cl_produce<NPL, 1>(id, a, s);
  else launch_cl_produce<NPL, 0>(id, a, s);
}
// the same, computing the LayerNorm statistics of `src` on the way (stored to st_out)
template <int NPL>
static void cv_stats_split(int id, const float* src, float* st_out, const float* gamma, const float* beta, float* dst, int F,
                           hipStream_t s) {
  CpArgs a{src, nullptr, st_out, gamma, beta, reinterpret_cast<unsigned short*>(dst), cl_plane(id, F), F};
  launch_cl_produce<NPL, 2>(id, a, s);
}

static CgArgs cv_gemm_args(int site, const float* wplanes, const float* xplanes, float* out, const float* bias, int F);
// encoder layer 3 forward on the tile that owns whole frames: conv + bias, the LayerNorm statistics of the result and (planes != nullptr)
// its activated bf16 operand planes in one kernel (k_cgemm_sf).  false: geometry not served, nothing launched
template <int NPL>
static bool cv_gemm_stats_planes(int site, const float* wplanes, const float* xplanes, float* out, const float* bias, float* st,
                                 const float* gamma, const float* beta, float* planes, int F, hipStream_t s) {
  if constexpr (NPL <= 2) {
    CgSfArgs b{cv_gemm_args(site, wplanes, xplanes, out, bias, F), st, gamma, beta, reinterpret_cast<unsigned short*>(planes), F};
    if (!cgemm_sf_serves(b.g)) return false;
    if constexpr (NPL == 2) {
      if (rt().cg_sf_ring && (rt().cg_sf_ring > 1 || F >= 36 * 256) && cgemm_sf_ring_serves(b.g)) {   // the four-wave LDS-DMA ring kernel (gfx950_ntring.h; VAENPVC_CG_SF_RING=0: A/B)
        launch_cgemm_sf_ring(b, s);
        return true;
      }
    }
    launch_cgemm_sf<NPL>(b, s);
    return true;
  }
  return false;
}

// encoder layer 3's input gradient with the LayerNorm + lrelu backward of layer 2 in its epilogue (k_cgemm_pf<LNB>): `out` receives
// d(pre-LN output of layer 2); returns the rows of `part` written (second stage: k_ln_bwd_reduce with C = 64), 0 = not served
template <int NPL>
static int cv_gemm_lnb(int site, const float* wplanes, const float* xplanes, float* out, const float* a2, const float* st, const float* gamma,
                       const float* beta, float* part, int64_t part_capacity, int F, hipStream_t s) {
  if constexpr (NPL <= 2) {
    const CgArgs a = cv_gemm_args(site, wplanes, xplanes, out, nullptr, F);
    if (!cgemm_pf_serves(a) || (int64_t)cdiv(a.N, CgPfTile<NPL>::BN) * 3 * 64 > part_capacity) return 0;
    if constexpr (NPL == 2) {
      // the (3, 3) ring kernel, 24 whole frames per tile (gfx950_ntring.h; VAENPVC_CG_PF_RING: 1 = from 6 144 frames on, 2 = always, 0 = never)
      if (rt().cg_pf_ring && (rt().cg_pf_ring > 1 || F >= 24 * 256) && cgemm_pf_ring_serves(a))
        return launch_cgemm_pf_ring_lnb(a, CgLnbArgs{a2, st, gamma, beta, part}, s);
    }
    return launch_cgemm_pf_lnb<NPL>(a, CgLnbArgs{a2, st, gamma, beta, part}, s);
  }
  return 0;
}

template <int NPL>
static void cv_gemm(int site, const float* wplanes, const float* xplanes, float* out, const float* bias, int F, hipStream_t s) {
  const CgArgs a = cv_gemm_args(site, wplanes, xplanes, out, bias, F);
  if constexpr (NPL <= 2) {
    if (rt().cg_pf && cgemm_pf_serves(a)) {   // encoder layer 3's input gradient: the tile that owns whole frames (VAENPVC_CG_PF=0: A/B)
      launch_cgemm_pf<NPL>(a, s);
      return;
    }
  }
  launch_cgemm_auto<NPL>(a, s);
}
static CgArgs cv_gemm_args(int site, const float* wplanes, const float* xplanes, float* out, const float* bias, int F) {
  const CvSite& v = CVS[site];
  const ClDesc& x = CLD[v.x];
  CgArgs a;
  memset(&a, 0, sizeof a);
  a.W = reinterpret_cast<const unsigned short*>(wplanes);
  a.X = reinterpret_cast<const unsigned short*>(xplanes);
  a.w_plane = (int64_t)v.Mp * v.Kp;
  a.x_plane = cl_plane(v.x, F);
  a.xv = RowView{v.R, x.HP * x.CP, 0, v.step};
  a.Kp = v.Kp;
  a.M = v.M;
  a.N = F * v.R;
  a.out = out;
  a.mdiv = v.mdiv;
  a.C = v.O;
  a.ofs = v.OC * v.OH;
  a.om = v.OH;
  a.oq = v.oq;
  a.o0 = v.o0;
  a.o0s = 1;
  a.OH = v.OH;
  a.bias = bias;
  return a;
}

// weight-gradient sites: dW[n*ldc + m] += sum over rows (f, j) of A[(f,j)][m] * B[(f,j)][n]; A = plain rows of tensor `a`
// (M channels), B = S-type view rows of tensor `b` (the K run (tap, channel) of row j; N = T * CP_b)
enum { CW_E1, CW_E2, CW_E3, CW_D0, CW_D1, CW_D2, CW_COUNT };
struct CwSite {
  int a, b;     // CL ids
  int M, T, R;  // channels of A, taps, rows per frame
};
constexpr CwSite CWS[CW_COUNT] = {
    {CL_GE1, CL_Y0, 32, 7, 57},    // encoder layer 1: dW[t][cin][cout], m = cout, n = (t, cin)
    {CL_GE2, CL_Y1, 64, 7, 19},
    {CL_GE3, CL_Y2, 128, 7, 7},
    {CL_H, CL_GD0, 81, 9, 19},     // decoder layer 0: dW[t][cout][cin], m = cin, n = (t, cout)
    {CL_YD0, CL_GD1, 32, 7, 57},
    {CL_YD1, CL_GD2, 16, 7, 171},
};
template <int NPL>
static void cv_wgrad(int site, const float* aplanes, const float* bplanes, float* dW, int F, int target_wgs, hipStream_t s) {
  const CwSite& v = CWS[site];
  const ClDesc &da = CLD[v.a], &db = CLD[v.b];
  TnpArgs t;
  memset(&t, 0, sizeof t);
  t.A = reinterpret_cast<const unsigned short*>(aplanes);
  t.B = reinterpret_cast<const unsigned short*>(bplanes);
  t.a_plane = cl_plane(v.a, F);
  t.b_plane = cl_plane(v.b, F);
  t.av = RowView{v.R, da.HP * da.CP, da.HLO * da.CP, da.CP};
  t.bv = RowView{v.R, db.HP * db.CP, 0, 3 * db.CP};
  t.lda = da.CP;
  t.ldb = v.T * db.CP;
  t.M = v.M;
  t.N = v.T * db.CP;
  t.F = F * v.R;
  t.C = dW;
  t.ldc = v.M;
  t.xcd = rt().tn_xcd >= 0 ? rt().tn_xcd : (v.M > 64 ? 1 : 0);   // measured: pays with >= 2 tiles of 128 x 256 per row chunk
  if constexpr (NPL <= 2) {
    // decoder layer 0 (M = 81, N = 288): the 96 x 288 tile of the 3 x 3 wave grid instead of two 128 x 256 tiles (VAENPVC_TN_D0FIT=0: A/B)
    if (site == CW_D0 && rt().tn_d0fit && !rt().tn_k16) return launch_gemm_tn32<NPL, TN_EPI_TRANS, 1, 3, 3, 3>(t, target_wgs, s);
  }
  if (v.M > 64) launch_gemm_tn<NPL, TN_EPI_TRANS, 2, 2>(t, target_wgs, s);
  else if (t.N > 128) launch_gemm_tn<NPL, TN_EPI_TRANS, 1, 2>(t, target_wgs, s);
  else launch_gemm_tn<NPL, TN_EPI_TRANS, 1, 1>(t, target_wgs, s);
}

}  // namespace tuned
}  // namespace vaenpvc
