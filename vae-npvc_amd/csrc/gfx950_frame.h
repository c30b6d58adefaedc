// gfx950_frame.h -- the SMALL-BATCH path of the VCC2016 geometry: whole frames per workgroup.
//
// Frames are independent through the whole network (per-sample LayerNorm, util/layers.py:32; batch-mean loss,
// model/vae.py:112-128), so at the reference's own batch sizes (16 frames, architecture-vae-vcc2016.json:23-28; 256 in
// BASELINE config 2's literal reading) a step does not have to be ~67 dependent launches of per-layer kernels
// (0.58 ms at 16 frames = its launch count).  Here ONE workgroup (1024 threads = one CU) carries
// ONE frame through a whole pass with its activations in LDS:
//   k_frame_fwd   x -> e0..e4 -> heads -> sampler + KL -> merge -> d0..d3 -> xh, log-density  (model/vae.py:72-137)
//   k_frame_bwd   d(xh) -> ... -> d(pre-LN output of encoder layer 0): the input-gradient chain with every LayerNorm
//                 backward in place (autodiff of the above, trainer/vae.py:24)
//   k_frame_wgrad every weight / bias / LayerNorm-parameter gradient in ONE launch: a job list of tiles over frame chunks
//                 (gfx950_frame_wgrad.h), tiles meet in LDS and leave as coalesced fp32 atomics into the zeroed buffer
//   k_frame_pack  aligned / transposed weight copies the first two read with 16-byte loads (weights change every step);
//                 the step's zero fills ride along
//   k_frame_toep_fwd / _bwd  (train steps up to 128 frames) the 1025-tap layer as eight 256-thread workgroups per frame
//                 between the two frame kernels, which then stop / start at decoder layer 2 (toep_split_* below)
// Arithmetic: plain fp32 FMAs on the vector ALUs (exact-fp32 class, like the fp32 matrix-core kernels they replace at
// these sizes).  At one frame per CU the pass is bound by streaming the 3.76 MB of weights through one CU's L2 port
// (~50 B/clk) and by LDS operand reads, not by FLOPs: the matrix cores would buy nothing here.
// The tensors written to the workspace are exactly the ones of the layered path (pre-LN outputs, statistics, z, h,
// xh, d(...)), so every stage can be compared with it -- or replaced by it -- in isolation.
//
// The per-thread code is written against two tiny abstractions so that the SAME source also compiles for the host
// (tests/frame_emu: FRAME_EMU defined, g++): a phase runner (device: f(threadIdx.x) + __syncthreads(); host: a loop over
// thread ids) and 16-byte loads.  The CPU suite checks that emulation against its float64 restatement of the reference (tests/test_frame_emu.py).
#pragma once
#include <cstdint>

#ifdef FRAME_EMU
#include <cmath>
#include <cstring>
#define FR_DEV inline
#define FR_RESTRICT
struct fr_f4 {
  float x, y, z, w;
};
FR_DEV fr_f4 fr_load4(const float* base, unsigned off) {
  fr_f4 v;
  std::memcpy(&v, base + off, 16);
  return v;
}
#define FR_G(T) T*
#define FR_UNIFORM(i) (i)
template <class T>
inline T* fr_g(T* p) { return p; }
template <class T>
inline T* fr_l(T* p) { return p; }
#define FR_UNROLL
#define FR_NOUNROLL
#define FR_STAGE inline
#else
#include <hip/hip_runtime.h>
#ifdef FR_NOINLINE
#define FR_DEV __device__ __attribute__((noinline))
#else
#define FR_DEV __device__ __forceinline__
#endif
#define FR_RESTRICT __restrict__
typedef float4 fr_f4;
// Address spaces.  The stages are real function calls, so a pointer read from the argument block is GENERIC to the
// compiler and every access through it becomes a flat_load / flat_store (slower, and it ties the LDS and vector-memory
// wait counters together: weight loads then serialise with LDS traffic).  Neither a cast round trip nor an assumption
// survives to the address-space inference here, so the global pointers are TYPED: FR_G(T) is a pointer to T in global
// memory (address space 1), fr_g() casts a generic pointer to it, fr_l() re-derives an LDS pointer.
#define FR_G(T) T __attribute__((address_space(1)))*
// (every pointer of the argument block is wave-uniform: through readfirstlane it lives in scalar registers instead of a
//  VGPR pair that is spilled and re-read from scratch behind every barrier)
template <class T>
__device__ __forceinline__ FR_G(T) fr_g(T* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return (FR_G(T))(((unsigned long long)hi << 32) | lo);
}
#define FR_UNIFORM(i) ((int)__builtin_amdgcn_readfirstlane((unsigned)(i)))
template <class T>
__device__ __forceinline__ T* fr_l(T* p) {
  return (T*)(__attribute__((address_space(3))) T*)p;
}
typedef float fr_v4 __attribute__((ext_vector_type(4)));
// 16-byte load at a wave-UNIFORM base + a 32-bit per-lane element offset: one global_load_dwordx4 with the base in scalar
// registers and ONE address VGPR (a 64-bit per-lane pointer per load costs two VGPRs each: the conv tiles spilled)
__device__ __forceinline__ fr_f4 fr_load4(FR_G(const float) base, unsigned off) {
  // (the byte offset is formed in 32 bits: a 64-bit scaled index cannot be proven to fit the instruction's 32-bit offset)
  const fr_v4 v = *(FR_G(const fr_v4))((FR_G(const char))base + (off << 2));
  return make_float4(v.x, v.y, v.z, v.w);
}
#define FR_UNROLL _Pragma("unroll")
#define FR_NOUNROLL _Pragma("unroll 1")
// a pass is a chain of STAGES (real function calls): inlined into one body, the compiler hoists every stage's address
// arithmetic to the top of the kernel and spills ~250 registers; a call boundary keeps each stage's values its own
#define FR_STAGE __device__ __attribute__((noinline))
#endif

// FR_PREFETCH=1: the conv tiles keep the next channel's taps in a second register set while the current channel's
// products run.  Measured inside the passes (not in isolation) it does not pay: the second set pushes the stage functions
// over 128 registers and the accumulators spill inside the channel loop (encoder layer 3: 14k -> 40k clocks).
#ifndef FR_PREFETCH
#define FR_PREFETCH 0
#endif

namespace vaenpvc {
namespace frame {

constexpr int NT = 1024;              // threads per workgroup (16 waves: four per SIMD, <= 128 registers each)
constexpr float LN_EPS_F = 1e-5f;     // util/layers.py:44
constexpr float LEAK_F = 0.02f;       // util/layers.py:147
constexpr float EPSILON_F = 1e-6f;    // util/layers.py:7
constexpr float LOG_2PI_F = 1.8378770664093453f;

constexpr int cdiv_(int a, int b) { return (a + b - 1) / b; }
constexpr int imax_(int a, int b) { return a > b ? a : b; }
constexpr int imin_(int a, int b) { return a < b ? a : b; }

// ------------------------------------------------------------------------------------------------ LDS map (floats)
constexpr int BUF = 4864;             // one activation buffer (largest: 8 x 513 outputs; inputs with halos: see *_HP below)
constexpr int PART = 24576;           // partial sums of the K-split products / staged Toeplitz taps
constexpr int L_BUFX = 0, L_BUFY = BUF, L_PART = 2 * BUF, L_RED = L_PART + PART, L_VEC = L_RED + 128, L_ARGS = L_VEC + 1024;
constexpr int ARGS_FLOATS = 176;                      // the pass's argument block, copied once from the kernel arguments
constexpr int L_CH = L_ARGS + ARGS_FLOATS;            // per-channel vectors of the 8 normalised layers: [3][LNP_C] = bias | scale | offset
constexpr int L_TOTAL = L_CH + 3 * 552;               // 37 416 floats = 149 664 bytes: one workgroup per CU
// reduction scratch: per-wave sums of the wave-reduction phases (16 waves), two pairs of slots, and {mean, rstd}
constexpr int NW = NT / 64;
constexpr int R_S1 = 0, R_S2 = 16, R_S3 = 32, R_S4 = 48, R_ST = 64;
FR_DEV float sum16(const float* p) {
  float s = 0.f;
  FR_UNROLL
  for (int i = 0; i < NW; ++i) s += p[i];
  return s;
}

// ------------------------------------------------------------------------------------------------ strided conv
// out[o][j] = sum_{c} sum_{t} W[(t*CC + c)*OO + o] * in[c][S*j - PAD + t]
// (forward of an encoder conv, util/layers.py:56-64 -- or the input gradient of a transposed conv, model/vae.py:96-99,
//  with the roles of the channel axes swapped).  A thread owns 4 consecutive output channels x JT consecutive positions
// and ONE slice of the contracted channels; lanes run along the output-channel quads first, so a wave's weight loads
// fall on consecutive addresses.  Input in LDS as [CC][HP] with the SAME-padding halo materialised as zeros.
template <int CC_, int HIN_, int OO_, int OV_, int HO_, int K_, int S_, int PAD_, int JT_, int KS_>
struct SConv {
  static constexpr int CC = CC_, HIN = HIN_, OO = OO_, OV = OV_, HO = HO_, K = K_, S = S_, PAD = PAD_, JT = JT_, KS = KS_;
  static constexpr int OG = OO / 4, JG = cdiv_(HO, JT), NTH = OG * JG * KS;
  static constexpr int CPS = cdiv_(CC, KS);               // contracted channels per slice
  static constexpr int WIN = S * (JT - 1) + K;            // input window of a thread
  static constexpr int HP = S * (JG * JT - 1) + K;        // row pitch: every window read stays inside the row
  static constexpr int NOUT = OV * HO;
  static_assert(OO % 4 == 0 && NTH <= NT && HP >= PAD + HIN && CC * HP <= BUF && KS * NOUT <= PART, "SConv tiling");
};

template <class T>
FR_DEV void sconv_part(int tid, const float* FR_RESTRICT in, FR_G(const float) W, float* FR_RESTRICT part) {
  if (tid >= T::NTH) return;
  const int og = tid % T::OG, r = tid / T::OG, jg = r % T::JG, ks = r / T::JG;
  const int j0 = jg * T::JT;
  float acc[4][T::JT];
  FR_UNROLL
  for (int q = 0; q < 4; ++q)
    FR_UNROLL
    for (int jj = 0; jj < T::JT; ++jj) acc[q][jj] = 0.f;
  const int c0 = ks * T::CPS, c1 = imin_(T::CC, c0 + T::CPS);
  // one channel's taps are in registers while the previous channel's products run (the loop is otherwise a chain of
  // L2 round trips: measured 2-3x the time of the FMAs at 2-4 waves per SIMD)
  auto loadw = [&](int c, fr_f4 (&w)[T::K]) {
    const unsigned off = (unsigned)(c * T::OO + 4 * og);      // per-lane part; the tap stride goes into the scalar base
    FR_UNROLL
    for (int t = 0; t < T::K; ++t) w[t] = fr_load4(W + (size_t)t * T::CC * T::OO, off);
  };
  auto comp = [&](int c, const fr_f4 (&w)[T::K]) {
    float win[T::WIN];
    const float* ir = in + c * T::HP + T::S * j0;
    FR_UNROLL
    for (int i = 0; i < T::WIN; ++i) win[i] = ir[i];
    FR_UNROLL
    for (int t = 0; t < T::K; ++t) {
      FR_UNROLL
      for (int jj = 0; jj < T::JT; ++jj) {
        const float v = win[T::S * jj + t];
        acc[0][jj] += w[t].x * v;
        acc[1][jj] += w[t].y * v;
        acc[2][jj] += w[t].z * v;
        acc[3][jj] += w[t].w * v;
      }
    }
  };
  if constexpr (FR_PREFETCH && T::K <= 7) {
    fr_f4 wa[T::K], wb[T::K];
    int c = c0;
    if (c < c1) loadw(c, wa);
    FR_NOUNROLL
    for (; c + 2 <= c1; c += 2) {
      loadw(c + 1, wb);
      comp(c, wa);
      if (c + 2 < c1) loadw(c + 2, wa);
      comp(c + 1, wb);
    }
    if (c < c1) comp(c, wa);
  } else {       // nine taps: two sets of them do not fit beside the accumulators (128 registers at 16 waves)
    FR_NOUNROLL
    for (int c = c0; c < c1; ++c) {
      fr_f4 wa[T::K];
      loadw(c, wa);
      comp(c, wa);
    }
  }
  FR_UNROLL
  for (int q = 0; q < 4; ++q) {
    const int o = 4 * og + q;
    if (o >= T::OV) continue;
    FR_UNROLL
    for (int jj = 0; jj < T::JT; ++jj)
      if (j0 + jj < T::HO) part[(ks * T::OV + o) * T::HO + j0 + jj] = acc[q][jj];
  }
}

// ------------------------------------------------------------------------------------------------ transposed conv
// out[o][p] = sum_c sum_j W[((p + PAD - S*j)*CC + c)*OO + o] * in[c][j],  0 <= p + PAD - S*j < K
// (forward of a decoder layer, model/vae.py:96-99 -- or the input gradient of an encoder conv).  With p = S*q + r the
// taps of phase r are t = u + S*m, u = (r + PAD) % S, reading in[c][q + (r + PAD)/S - m]: a thread owns 4 output
// channels x QT input steps x all S phases and one slice of the contracted channels.  Input in LDS as [CC][HP] with
// HL zero positions in front (the taps that reach below position 0) and zeros behind the last one.
template <int S, int K, int PAD>
struct TPhase {   // taps of one output phase (compile-time helpers)
  static constexpr int u(int r) { return (r + PAD) % S; }
  static constexpr int m0(int r) { return (r + PAD) / S; }
  static constexpr int cnt(int r) { return (K - u(r) + S - 1) / S; }
  static constexpr int jlo() {
    int v = 1 << 20;
    for (int r = 0; r < S; ++r) v = imin_(v, m0(r) - (cnt(r) - 1));
    return v;
  }
  static constexpr int jhi() {
    int v = -(1 << 20);
    for (int r = 0; r < S; ++r) v = imax_(v, m0(r));
    return v;
  }
};
template <int CC_, int Q_, int OO_, int OV_, int HOUT_, int K_, int S_, int PAD_, int QT_, int KS_>
struct TConv {
  static constexpr int CC = CC_, Q = Q_, OO = OO_, OV = OV_, HOUT = HOUT_, K = K_, S = S_, PAD = PAD_, QT = QT_, KS = KS_;
  using PH = TPhase<S, K, PAD>;
  static constexpr int OG = OO / 4, QG = cdiv_(Q, QT), NTH = OG * QG * KS;
  static constexpr int CPS = cdiv_(CC, KS);
  static constexpr int HL = -PH::jlo(), HR = PH::jhi();
  static constexpr int WIN = QT + HL + HR;
  static constexpr int HP = HL + QG * QT + HR;
  static constexpr int NOUT = OV * HOUT;
  static_assert(OO % 4 == 0 && NTH <= NT && HL >= 0 && HR >= 0 && CC * HP <= BUF && KS * NOUT <= PART && HOUT <= S * Q,
                "TConv tiling");
};

template <class T>
FR_DEV void tconv_part(int tid, const float* FR_RESTRICT in, FR_G(const float) W, float* FR_RESTRICT part) {
  if (tid >= T::NTH) return;
  using PH = typename T::PH;
  const int og = tid % T::OG, r_ = tid / T::OG, qg = r_ % T::QG, ks = r_ / T::QG;
  const int q0 = qg * T::QT;
  float acc[T::S][T::QT][4];
  FR_UNROLL
  for (int r = 0; r < T::S; ++r)
    FR_UNROLL
    for (int qq = 0; qq < T::QT; ++qq)
      FR_UNROLL
      for (int e = 0; e < 4; ++e) acc[r][qq][e] = 0.f;
  const int c0 = ks * T::CPS, c1 = imin_(T::CC, c0 + T::CPS);
  auto loadw = [&](int c, fr_f4 (&w)[T::K]) {
    const unsigned off = (unsigned)(c * T::OO + 4 * og);      // per-lane part; the tap stride goes into the scalar base
    FR_UNROLL
    for (int t = 0; t < T::K; ++t) w[t] = fr_load4(W + (size_t)t * T::CC * T::OO, off);
  };
  auto comp = [&](int c, const fr_f4 (&w)[T::K]) {
    float win[T::WIN];                       // win[i] = in[c][q0 - HL + i]
    const float* ir = in + c * T::HP + q0;   // (row index of position j is j + HL)
    FR_UNROLL
    for (int i = 0; i < T::WIN; ++i) win[i] = ir[i];
    FR_UNROLL
    for (int r = 0; r < T::S; ++r) {
      FR_UNROLL
      for (int m = 0; m < PH::cnt(r); ++m) {
        const int t = PH::u(r) + T::S * m;
        FR_UNROLL
        for (int qq = 0; qq < T::QT; ++qq) {
          const float v = win[qq + PH::m0(r) - m + T::HL];
          acc[r][qq][0] += w[t].x * v;
          acc[r][qq][1] += w[t].y * v;
          acc[r][qq][2] += w[t].z * v;
          acc[r][qq][3] += w[t].w * v;
        }
      }
    }
  };
  if constexpr (FR_PREFETCH && T::K <= 7) {
    fr_f4 wa[T::K], wb[T::K];
    int c = c0;
    if (c < c1) loadw(c, wa);
    FR_NOUNROLL
    for (; c + 2 <= c1; c += 2) {
      loadw(c + 1, wb);
      comp(c, wa);
      if (c + 2 < c1) loadw(c + 2, wa);
      comp(c + 1, wb);
    }
    if (c < c1) comp(c, wa);
  } else {       // nine taps: two sets of them do not fit beside the accumulators (128 registers at 16 waves)
    FR_NOUNROLL
    for (int c = c0; c < c1; ++c) {
      fr_f4 wa[T::K];
      loadw(c, wa);
      comp(c, wa);
    }
  }
  FR_UNROLL
  for (int e = 0; e < 4; ++e) {
    const int o = 4 * og + e;
    if (o >= T::OV) continue;
    FR_UNROLL
    for (int qq = 0; qq < T::QT; ++qq)
      FR_UNROLL
      for (int r = 0; r < T::S; ++r) {
        const int p = T::S * (q0 + qq) + r;
        if (q0 + qq < T::Q && p < T::HOUT) part[(ks * T::OV + o) * T::HOUT + p] = acc[r][qq][e];
      }
  }
}

// ------------------------------------------------------------------------------------------------ dense (GEMV)
// out[n] = sum_k v[k] * W[k*LDW + n], n in quads (16-byte weight loads), K split over KS slices
template <int KK_, int NN_, int LDW_, int KS_>
struct Dense4 {
  static constexpr int KK = KK_, NN = NN_, LDW = LDW_, KS = KS_;
  static constexpr int NG = NN / 4, NTH = NG * KS, KPS = cdiv_(KK, KS);
  static_assert(NN % 4 == 0 && LDW % 4 == 0 && NTH <= NT && KS * NN <= PART, "Dense4 tiling");
};
template <class T>
FR_DEV void dense4_part(int tid, const float* FR_RESTRICT v, FR_G(const float) W, float* FR_RESTRICT part) {
  if (tid >= T::NTH) return;
  const int ng = tid % T::NG, ks = tid / T::NG;
  const int k0 = ks * T::KPS, k1 = imin_(T::KK, k0 + T::KPS);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int k = k0;
  for (; k + 8 <= k1; k += 8) {          // eight independent 16-byte loads in flight per thread
    fr_f4 w[8];
    FR_UNROLL
    for (int u = 0; u < 8; ++u) w[u] = fr_load4(W, (unsigned)((k + u) * T::LDW + 4 * ng));
    FR_UNROLL
    for (int u = 0; u < 8; ++u) {
      const float x = v[k + u];
      a0 += w[u].x * x;
      a1 += w[u].y * x;
      a2 += w[u].z * x;
      a3 += w[u].w * x;
    }
  }
  for (; k < k1; ++k) {
    const fr_f4 w = fr_load4(W, (unsigned)(k * T::LDW + 4 * ng));
    const float x = v[k];
    a0 += w.x * x;
    a1 += w.y * x;
    a2 += w.z * x;
    a3 += w.w * x;
  }
  float* p = part + ks * T::NN + 4 * ng;
  p[0] = a0;
  p[1] = a1;
  p[2] = a2;
  p[3] = a3;
}

// merge (model/vae.py:51-61): h[n] = sum_k z[k] Wz[k][n] + T[y][n], T = E Wy + (bz + by + b) built once per step by the
// pack launch (the embedding term depends on the speaker only: half of the layer's 1.6 MB of weights never has to be
// streamed per frame).  Rows of 1539 floats are not 16-byte aligned, so lanes run along n with 4-byte loads
// (coalesced); a thread owns n = t, t + 512, t + 1024 (, t + 1536) and one half of k
constexpr int MERGE_N = 1539, MERGE_K = 128, MERGE_NY = 10;
FR_DEV void merge_part(int tid, const float* FR_RESTRICT z /*[128]*/, FR_G(const float) Wz, float* FR_RESTRICT part /*[2][1539]*/) {
  const int t = tid & 511, ks = tid >> 9;
  const unsigned wbase = (unsigned)(ks * 64 * MERGE_N + t);
  const float* v = z + ks * 64;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  const bool ok3 = t + 1536 < MERGE_N;
  FR_NOUNROLL
  for (int k = 0; k < 64; k += 4) {
    float w[4][4];
    FR_UNROLL
    for (int u = 0; u < 4; ++u) {
      const unsigned wo = wbase + (unsigned)((k + u) * MERGE_N);
      w[u][0] = Wz[wo];
      w[u][1] = Wz[wo + 512];
      w[u][2] = Wz[wo + 1024];
      w[u][3] = ok3 ? Wz[wo + 1536] : 0.f;
    }
    FR_UNROLL
    for (int u = 0; u < 4; ++u) {
      const float x = v[k + u];
      FR_UNROLL
      for (int i = 0; i < 4; ++i) a[i] += w[u][i] * x;
    }
  }
  FR_UNROLL
  for (int i = 0; i < 4; ++i)
    if (t + 512 * i < MERGE_N) part[ks * MERGE_N + t + 512 * i] = a[i];
}

// ------------------------------------------------------------------------------------------------ the 1025-tap layer
// model/vae.py:96-99, last decoder layer: out[p] = b + sum_c sum_j W[p + 512 - j][c] y[c][j] (513 outputs, 8 channels).
// Taps staged in LDS as wt[c][TW] (one contiguous row per channel: index p + 512 - j).  A thread owns 9 consecutive
// outputs (513 = 57 x 9; a lane stride of 9 words is conflict free), one channel and one half of the contracted
// positions; the 9 taps it needs slide by one per step, so a step costs ONE new LDS read for nine FMAs (the register
// window rotates through a 9-fold unrolled loop).
constexpr int TP_C = 8, TP_H = 513, TP_K = 1025, TP_W = 1028, TP_G = 57, TP_R = 9, TP_HALF = 257;
constexpr int TP_NTH = TP_G * TP_C * 2;     // 912 threads
static_assert(TP_C * TP_W + 16 * TP_H <= PART, "Toeplitz staging");
// forward: acc[i] += wt[c][p0 + i + 512 - j] * y[c][j]
FR_DEV void toep_fwd_part(int tid, const float* FR_RESTRICT y /*[8][513]*/, const float* FR_RESTRICT wt,
                          float* FR_RESTRICT part /*[16][513]*/) {
  if (tid >= TP_NTH) return;
  const int pg = tid % TP_G, r = tid / TP_G, c = r % TP_C, half = r / TP_C;
  const int p0 = pg * TP_R;
  const int j0 = half * TP_HALF, j1 = imin_(TP_H, j0 + TP_HALF);
  const float* w = wt + c * TP_W + p0 + 512;     // w[i - j]
  const float* yc = y + c * TP_H;
  float acc[TP_R], win[TP_R];
  FR_UNROLL
  for (int i = 0; i < TP_R; ++i) acc[i] = 0.f;
  // window for step j: win[(i + j) % 9] = w[i - j]  (slot of tap index d = i - j is ((d % 9) + 9) % 9 shifted by 2j... kept
  // simple: slot s(j, i) = (i + 8 * j) % 9, i.e. every step overwrites the slot the oldest tap (i = 8) leaves)
  int j = j0;
  // prologue: taps of step j0 for i = 1..8 (i = 0 is loaded inside the step)
  FR_UNROLL
  for (int i = 1; i < TP_R; ++i) win[i] = w[i - j0];
  // slots are addressed relative to the step inside a 9-fold unrolled body: at relative step s (0..8) tap i lives in
  // slot (i - s + 9) % 9 ... with the prologue above written for s = 0
  for (; j + TP_R <= j1; j += TP_R) {
    FR_UNROLL
    for (int s = 0; s < TP_R; ++s) {
      // new tap for i = 0 at this step: w[0 - (j + s)] -> slot (0 - s + 9) % 9
      win[(TP_R - s) % TP_R] = w[-(j + s)];
      const float v = yc[j + s];
      FR_UNROLL
      for (int i = 0; i < TP_R; ++i) acc[i] += win[(i - s + TP_R) % TP_R] * v;
    }
  }
  // tail (fewer than 9 steps left): same body, bounded
  FR_UNROLL
  for (int s = 0; s < TP_R; ++s) {
    if (j + s < j1) {
      win[(TP_R - s) % TP_R] = w[-(j + s)];
      const float v = yc[j + s];
      FR_UNROLL
      for (int i = 0; i < TP_R; ++i) acc[i] += win[(i - s + TP_R) % TP_R] * v;
    }
  }
  float* po = part + (half * TP_C + c) * TP_H + p0;
  FR_UNROLL
  for (int i = 0; i < TP_R; ++i) po[i] = acc[i];
}
// input gradient: dy[c][j0 + i] = sum_p wt[c][p + 512 - j0 - i] * g[p]; partial over one half of p
FR_DEV void toep_dgrad_part(int tid, const float* FR_RESTRICT g /*[513]*/, const float* FR_RESTRICT wt,
                            float* FR_RESTRICT part /*[2][8][513]*/) {
  if (tid >= TP_NTH) return;
  const int jg = tid % TP_G, r = tid / TP_G, c = r % TP_C, half = r / TP_C;
  const int jb = jg * TP_R;
  const int p0 = half * TP_HALF, p1 = imin_(TP_H, p0 + TP_HALF);
  const float* w = wt + c * TP_W + 512 - jb;      // w[p - i]
  float acc[TP_R], win[TP_R];
  FR_UNROLL
  for (int i = 0; i < TP_R; ++i) acc[i] = 0.f;
  // step p: tap of output i is w[p - i]; going to p + 1 the tap of i becomes the old tap of i - 1: the NEW tap enters at
  // i = 0 (w[p + 1]) -- the same rotation as the forward direction with the window walking upwards
  int p = p0;
  FR_UNROLL
  for (int i = 1; i < TP_R; ++i) win[i] = w[p0 - i];
  for (; p + TP_R <= p1; p += TP_R) {
    FR_UNROLL
    for (int s = 0; s < TP_R; ++s) {
      win[(TP_R - s) % TP_R] = w[p + s];
      const float v = g[p + s];
      FR_UNROLL
      for (int i = 0; i < TP_R; ++i) acc[i] += win[(i - s + TP_R) % TP_R] * v;
    }
  }
  FR_UNROLL
  for (int s = 0; s < TP_R; ++s) {
    if (p + s < p1) {
      win[(TP_R - s) % TP_R] = w[p + s];
      const float v = g[p + s];
      FR_UNROLL
      for (int i = 0; i < TP_R; ++i) acc[i] += win[(i - s + TP_R) % TP_R] * v;
    }
  }
  float* po = part + (half * TP_C + c) * TP_H + jb;
  FR_UNROLL
  for (int i = 0; i < TP_R; ++i) po[i] = acc[i];
}

// ------------------------------------------------------------------------------------------------ geometry of the VCC2016 net
//                      CC   HIN  OO   OV   HO   K  S PAD JT  KS
using E0F = SConv<1, 513, 16, 16, 171, 7, 3, 2, 3, 1>;
using E1F = SConv<16, 171, 32, 32, 57, 7, 3, 2, 3, 4>;
using E2F = SConv<32, 57, 64, 64, 19, 7, 3, 2, 3, 8>;
using E3F = SConv<64, 19, 128, 128, 7, 7, 3, 3, 4, 16>;
using E4F = SConv<128, 7, 256, 256, 3, 7, 3, 3, 3, 16>;
// input gradients of the transposed convs (contract over the layer's output channels; weights [t][o][c], c fastest)
using D2G = SConv<8, 513, 16, 16, 171, 7, 3, 2, 3, 2>;
using D1G = SConv<16, 171, 32, 32, 57, 7, 3, 2, 3, 4>;
using D0G = SConv<32, 57, 84, 81, 19, 9, 3, 3, 3, 6>;
//                      CC   Q    OO   OV  HOUT  K  S PAD QT  KS
using D0F = TConv<81, 19, 32, 32, 57, 9, 3, 3, 2, 12>;
using D1F = TConv<32, 57, 16, 16, 171, 7, 3, 2, 2, 8>;
using D2F = TConv<16, 171, 8, 8, 513, 7, 3, 2, 2, 4>;
// input gradients of the encoder convs (contract over the layer's output channels; weights packed [t][o][c])
using E4G = TConv<256, 3, 128, 128, 7, 7, 3, 3, 3, 16>;
using E3G = TConv<128, 7, 64, 64, 19, 7, 3, 3, 2, 16>;
using E2G = TConv<64, 19, 32, 32, 57, 7, 3, 2, 2, 12>;
using E1G = TConv<32, 57, 16, 16, 171, 7, 3, 2, 2, 8>;
using HeadsF = Dense4<768, 256, 256, 16>;     // [y4] x [Wmu | Wlv] (packed side by side: 256 columns)
using HeadsG = Dense4<256, 768, 768, 4>;      // [dz_mu | dz_lv] x [Wmu | Wlv]^T
using MergeG = Dense4<1539, 128, 128, 32>;    // d(h) x Wz^T

// ------------------------------------------------------------------------------------------------ packed weights
// float offsets inside the pack buffer (k_frame_pack; every block 16-byte aligned)
struct Pk {
  static constexpr int heads = 0;                                  // [768][256]   = [Wmu | Wlv]
  static constexpr int headsT = heads + 768 * 256;                 // [256][768]
  static constexpr int wzT = headsT + 256 * 768;                   // [1539][128]
  static constexpr int d0f = wzT + 1539 * 128;                     // [9][81][32]  conv_transpose forward: [t][c][o]
  static constexpr int d1f = d0f + 9 * 81 * 32;                    // [7][32][16]
  static constexpr int d2f = d1f + 7 * 32 * 16;                    // [7][16][8]
  static constexpr int d0g = d2f + 7 * 16 * 8;                     // [9][32][84]  aligned copy of [t][o][c], c padded
  static constexpr int d1g = d0g + 9 * 32 * 84;                    // [7][16][32]
  static constexpr int d2g = d1g + 7 * 16 * 32;                    // [7][8][16]
  static constexpr int e4g = d2g + 7 * 8 * 16;                     // [7][256][128] encoder input gradients: [t][o][c]
  static constexpr int e3g = e4g + 7 * 256 * 128;
  static constexpr int e2g = e3g + 7 * 128 * 64;
  static constexpr int e1g = e2g + 7 * 64 * 32;
  static constexpr int w3t = e1g + 7 * 32 * 16;                    // [8][1028]    taps of the last layer per channel
  static constexpr int mtab = w3t + TP_C * TP_W;                   // [10][1539]   T = E Wy + (bz + by + b) (+ 2 floats of padding)
  static constexpr int wyT = mtab + MERGE_NY * MERGE_N + 2;        // [1539][128]  Wy transposed (speaker-embedding gradient)
  static constexpr int total = wyT + MERGE_N * MERGE_K;
};
static_assert(Pk::heads % 4 == 0 && Pk::headsT % 4 == 0 && Pk::wzT % 4 == 0 && Pk::d0f % 4 == 0 && Pk::d1f % 4 == 0 &&
                  Pk::d2f % 4 == 0 && Pk::d0g % 4 == 0 && Pk::d1g % 4 == 0 && Pk::d2g % 4 == 0 && Pk::e4g % 4 == 0 &&
                  Pk::e3g % 4 == 0 && Pk::e2g % 4 == 0 && Pk::e1g % 4 == 0 && Pk::w3t % 4 == 0 && Pk::mtab % 4 == 0 && Pk::wyT % 4 == 0,
              "packed blocks must be 16-byte aligned");

// parameter offsets of the 44 tensors (flat buffer, TF creation order; model.cpp fills it for the VCC2016 geometry)
struct POff {
  int emb;
  int ew[5], eb[5], ebeta[5], egamma[5];
  int wmu, bmu, wlv, blv;
  int wz, bz, wy, by, bm;
  int dw[4], db[4], dbeta[3], dgamma[3];
};

// source element of packed element i (gather form: one thread per destination element)
FR_DEV float pack_src(const float* FR_RESTRICT P, const POff& o, int i) {
  if (i < Pk::headsT) {                       // heads [k][n]: n < 128 -> Wmu[k][n], else Wlv[k][n - 128]
    const int k = i / 256, n = i % 256;
    return n < 128 ? P[o.wmu + k * 128 + n] : P[o.wlv + k * 128 + n - 128];
  }
  if (i < Pk::wzT) {                          // headsT [n][k]
    const int j = i - Pk::headsT, n = j / 768, k = j % 768;
    return n < 128 ? P[o.wmu + k * 128 + n] : P[o.wlv + k * 128 + n - 128];
  }
  if (i < Pk::d0f) {                          // wzT [n][k] = Wz[k][n]
    const int j = i - Pk::wzT, n = j / 128, k = j % 128;
    return P[o.wz + k * 1539 + n];
  }
  if (i < Pk::d0g) {                          // conv_transpose forward copies [t][c][o] from TF [t][o][c]
    int j, C, O, w;
    if (i < Pk::d1f) { j = i - Pk::d0f; C = 81; O = 32; w = o.dw[0]; }
    else if (i < Pk::d2f) { j = i - Pk::d1f; C = 32; O = 16; w = o.dw[1]; }
    else { j = i - Pk::d2f; C = 16; O = 8; w = o.dw[2]; }
    const int oo = j % O, c = (j / O) % C, t = j / (O * C);
    return P[w + (t * O + oo) * C + c];
  }
  if (i < Pk::e4g) {                          // aligned copies of TF [t][o][c] (c padded to CP)
    int j, C, CP, O, w;
    if (i < Pk::d1g) { j = i - Pk::d0g; C = 81; CP = 84; O = 32; w = o.dw[0]; }
    else if (i < Pk::d2g) { j = i - Pk::d1g; C = 32; CP = 32; O = 16; w = o.dw[1]; }
    else { j = i - Pk::d2g; C = 16; CP = 16; O = 8; w = o.dw[2]; }
    const int c = j % CP, oo = (j / CP) % O, t = j / (CP * O);
    return c < C ? P[w + (t * O + oo) * C + c] : 0.f;
  }
  if (i < Pk::w3t) {                          // encoder input-gradient copies [t][o][c] from TF [t][c][o]
    int j, C, O, w;
    if (i < Pk::e3g) { j = i - Pk::e4g; C = 128; O = 256; w = o.ew[4]; }
    else if (i < Pk::e2g) { j = i - Pk::e3g; C = 64; O = 128; w = o.ew[3]; }
    else if (i < Pk::e1g) { j = i - Pk::e2g; C = 32; O = 64; w = o.ew[2]; }
    else { j = i - Pk::e1g; C = 16; O = 32; w = o.ew[1]; }
    const int c = j % C, oo = (j / C) % O, t = j / (C * O);
    return P[w + (t * C + c) * O + oo];
  }
  if (i < Pk::mtab) {                         // w3t [c][1028]: tap t of channel c (TF [t][1][1][8]); padding zero
    const int j = i - Pk::w3t, c = j / TP_W, t = j % TP_W;
    return t < TP_K ? P[o.dw[3] + t * TP_C + c] : 0.f;
  }
  if (i >= Pk::wyT) {                         // wyT [n][k] = Wy[k][n]
    const int j = i - Pk::wyT, n = j / MERGE_K, k = j % MERGE_K;
    return P[o.wy + (size_t)k * MERGE_N + n];
  }
  {                                           // merge table: T[k][n] = sum_i E[k][i] Wy[i][n] + bz[n] + by[n] + b[n]
    const int j = i - Pk::mtab;
    if (j >= MERGE_NY * MERGE_N) return 0.f;
    const int k = j / MERGE_N, n = j % MERGE_N;
    float s = P[o.bz + n] + P[o.by + n] + P[o.bm + n];
    for (int e0 = 0; e0 < MERGE_K; e0 += 16) {      // 16 + 16 loads in flight, then the sum in the fixed order e = 0, 1, ...
      float ev[16], wv[16];
      FR_UNROLL
      for (int u = 0; u < 16; ++u) {
        ev[u] = P[o.emb + k * MERGE_K + e0 + u];
        wv[u] = P[o.wy + (size_t)(e0 + u) * MERGE_N + n];
      }
      FR_UNROLL
      for (int u = 0; u < 16; ++u) s += ev[u] * wv[u];
    }
    return s;
  }
}

// ------------------------------------------------------------------------------------------------ per-channel vectors
// channel slots of the 8 normalised layers (order of the backward pass: dec2, dec1, dec0, enc4, enc3, enc2, enc1, enc0)
constexpr int LNP_C = 552;
constexpr int LNP_DEC2 = 0, LNP_DEC1 = 8, LNP_DEC0 = 24, LNP_ENC4 = 56, LNP_ENC3 = 312, LNP_ENC2 = 440, LNP_ENC1 = 504, LNP_ENC0 = 536;
// conv bias, LayerNorm scale and offset of every channel, copied to LDS once per workgroup: the short phases between
// the conv tiles then touch no global memory at all (a phase that waits for one L2 round trip costs ~1.5k clocks)
FR_DEV void load_channel_vectors(int tid, FR_G(const float) P, const POff& o, float* FR_RESTRICT ch) {
  for (int i = tid; i < LNP_C; i += NT) {
    int b, g, be, c;
    if (i < LNP_DEC1) { c = i - LNP_DEC2; b = o.db[2]; g = o.dgamma[2]; be = o.dbeta[2]; }
    else if (i < LNP_DEC0) { c = i - LNP_DEC1; b = o.db[1]; g = o.dgamma[1]; be = o.dbeta[1]; }
    else if (i < LNP_ENC4) { c = i - LNP_DEC0; b = o.db[0]; g = o.dgamma[0]; be = o.dbeta[0]; }
    else if (i < LNP_ENC3) { c = i - LNP_ENC4; b = o.eb[4]; g = o.egamma[4]; be = o.ebeta[4]; }
    else if (i < LNP_ENC2) { c = i - LNP_ENC3; b = o.eb[3]; g = o.egamma[3]; be = o.ebeta[3]; }
    else if (i < LNP_ENC1) { c = i - LNP_ENC2; b = o.eb[2]; g = o.egamma[2]; be = o.ebeta[2]; }
    else if (i < LNP_ENC0) { c = i - LNP_ENC1; b = o.eb[1]; g = o.egamma[1]; be = o.ebeta[1]; }
    else { c = i - LNP_ENC0; b = o.eb[0]; g = o.egamma[0]; be = o.ebeta[0]; }
    ch[i] = P[b + c];
    ch[LNP_C + i] = P[g + c];
    ch[2 * LNP_C + i] = P[be + c];
  }
}

// ------------------------------------------------------------------------------------------------ phases shared by both passes
// A runner offers   phase(f)            f(tid) for every thread, then a workgroup barrier
//                   reduce(dst, f)      dst[wave] = sum over the wave's lanes of f(tid), then a barrier
//                   reduce2(dA, dB, f)  the same for a pair of values, f(tid, a, b)
// (device: wave shuffles; host emulation: plain loops).  A workgroup-wide sum is then sum16(dst) in the NEXT phase:
// two barriers per LayerNorm statistic instead of six.

// copy a tensor that sits in LDS to HBM.  The phase barriers do not wait for global stores (nothing written to HBM is read
// again inside a pass), but loads return in order BEHIND older stores: a flush sits at the top of a phase that needs no
// global data, as far ahead of the next weight loads as the data dependences allow
FR_DEV void flush(int tid, const float* FR_RESTRICT src, FR_G(float) dst, int n) {
  if (!dst) return;
  for (int i = tid; i < n; i += NT) dst[i] = src[i];
}

// out[i] = sum over the K slices of part[k][i] (+ bias of the channel) -> `by`; per-wave sums of the outputs -> red[R_S1..]
template <class R, int KS, int NOUT, int H>
FR_DEV void reduce_sum(R& run, const float* part, const float* bias, float* by, float* red) {
  run.reduce(red + R_S1, [&](int tid) {
    float t = 0.f;
    for (int i = tid; i < NOUT; i += NT) {
      float s = bias ? bias[i / H] : 0.f;
      FR_UNROLL
      for (int k = 0; k < KS; ++k) s += part[k * NOUT + i];
      by[i] = s;
      t += s;
    }
    return t;
  });
}
// centred second moment (util/layers.py:32: biased variance over the whole frame), two-pass like the layered kernels
template <class R, int N>
FR_DEV void var_sum(R& run, const float* a, float* red, FR_G(float) a_g) {
  run.reduce(red + R_S2, [&](int tid) {
    flush(tid, a, a_g, N);      // the pre-LN tensor leaves for HBM under two LDS-only phases (nothing waits for the stores)
    const float mean = sum16(red + R_S1) * (1.0f / N);
    float q = 0.f;
    for (int i = tid; i < N; i += NT) {
      const float d = a[i] - mean;
      q += d * d;
    }
    return q;
  });
}
template <int N>
FR_DEV void ln_consts(const float* red, float& mean, float& rstd) {
  mean = sum16(red + R_S1) * (1.0f / N);
  rstd = 1.0f / sqrtf(sum16(red + R_S2) * (1.0f / N) + LN_EPS_F);
}

FR_DEV float lnact(float v, float mean, float rstd, float g, float b) {
  const float n = (v - mean) * rstd * g + b;
  return fmaxf(n, LEAK_F * n);
}

// y = lrelu(LN(a)) written as the next layer's LDS input: rows of HP floats, the valid positions at [HL, HL + H), zeros
// elsewhere (the SAME-padding halo); HP == H and HL == 0 gives the plain [C][H] tensor.  Also stores {mean, rstd}.
template <int C, int H, int HP, int HL>
FR_DEV void ln_apply(int tid, const float* FR_RESTRICT a, const float* FR_RESTRICT red, const float* FR_RESTRICT gamma,
                     const float* FR_RESTRICT beta, float* FR_RESTRICT out, FR_G(float) st_g, FR_G(float) y_g) {
  float mean, rstd;
  ln_consts<C * H>(red, mean, rstd);
  if (tid == 0 && st_g) {
    st_g[0] = mean;
    st_g[1] = rstd;
  }
  for (int i = tid; i < C * HP; i += NT) {
    const int c = i / HP, h = i % HP - HL;
    const bool ok = h >= 0 && h < H;
    const float v = ok ? lnact(a[c * H + h], mean, rstd, gamma[c], beta[c]) : 0.f;
    out[i] = v;
    if (ok && y_g) y_g[c * H + h] = v;      // the activated tensor, plain layout: operand of the weight-gradient launch
  }
}
// plain tensor -> halo layout (no LayerNorm: the merge output, gradients)
template <int C, int H, int HP, int HL, class PT>
FR_DEV void halo_copy(int tid, PT a, float* FR_RESTRICT out) {
  for (int i = tid; i < C * HP; i += NT) {
    const int c = i / HP, h = i % HP - HL;
    out[i] = (h >= 0 && h < H) ? a[c * H + h] : 0.f;
  }
}

// ------------------------------------------------------------------------------------------------ the 1025-tap layer, split out
// In a train step the layer is 40 % of each frame kernel's arithmetic and runs on ONE compute unit per frame there.  The
// two functions below are the bodies of 256-thread workgroups, EIGHT per frame, launched between the frame kernels
// (forward: a group of 65 outputs each; input gradient: a channel each): the same rotating nine-register window, a
// fraction of the length per thread.
constexpr int TS_T = 256;                       // threads per workgroup
constexpr int TS_NO = 65, TS_NG = 8;            // forward: outputs per workgroup (8 x 65 >= 513), computed as 8 groups of 9
constexpr int TS_IL = 129;                      // ... reduction positions per thread (4 quarters)
constexpr int TS_TW = 585;                      // ... taps a workgroup's outputs can meet (584), row pitch
constexpr int TS_FWD_LDS = TP_C * TP_H + TP_C * TS_TW + 32 * 72;
// xh[h] = b3 + sum_c sum_i y[c][i] w[c][h - i + 512] for h in [65 og, 65 og + 65) (conv_transpose, stride 1, SAME: output h meets
// input i through tap h - i + 512; model/vae.py:96-103); log-density terms of those outputs
template <class R>
FR_DEV void toep_split_fwd(R& run, float* lds, const float* y2 /*[8][513]*/, const float* w3t /*[8][1028]*/, float b3,
                           const float* target /*[513]*/, int og, float* xh /*[513]*/, float* nll8 /*this frame's 8 partial sums*/) {
  float* ys = lds;                       // [8][513]
  float* ww = ys + TP_C * TP_H;          // [8][585]: ww[c][k] = w[c][tlo + k], zero outside the kernel
  float* pp = ww + TP_C * TS_TW;         // [32][72] partial sums, then [72] outputs
  const int tlo = TS_NO * og;        // taps h - i + 512 of h in [65 og, 65 og + 72), i in [0, 513): [65 og, 65 og + 583]
  run.phase([&](int tid) {
    for (int i0 = tid; i0 < TP_C * TP_H; i0 += 8 * TS_T) {
      float v[8];
      FR_UNROLL
      for (int u = 0; u < 8; ++u) v[u] = i0 + u * TS_T < TP_C * TP_H ? y2[i0 + u * TS_T] : 0.f;
      FR_UNROLL
      for (int u = 0; u < 8; ++u)
        if (i0 + u * TS_T < TP_C * TP_H) ys[i0 + u * TS_T] = v[u];
    }
    for (int i0 = tid; i0 < TP_C * TS_TW; i0 += 8 * TS_T) {
      float v[8];
      FR_UNROLL
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * TS_T, c = i / TS_TW, t = tlo + i % TS_TW;
        v[u] = (i < TP_C * TS_TW && t >= 0 && t < TP_K) ? w3t[c * TP_W + t] : 0.f;
      }
      FR_UNROLL
      for (int u = 0; u < 8; ++u)
        if (i0 + u * TS_T < TP_C * TS_TW) ww[i0 + u * TS_T] = v[u];
    }
  });
  run.phase([&](int tid) {
    const int g9 = tid % TS_NG, c = (tid / TS_NG) % TP_C, iq = tid / (TS_NG * TP_C);
    const float* yc = ys + c * TP_H;
    const float* wc = ww + c * TS_TW + 9 * g9 + 512;     // tap of output r at position i: wc[r - i]  (local index 9 g9 + r - i + 512 in [0, 583])
    const int ib = iq * TS_IL, ie = imin_(TP_H, ib + TS_IL);
    float acc[TP_R], win[TP_R];
    FR_UNROLL
    for (int r = 0; r < TP_R; ++r) acc[r] = 0.f;
    FR_UNROLL
    for (int r = 1; r < TP_R; ++r) win[r] = wc[r - ib];
    int i = ib;
    for (; i + TP_R <= ie; i += TP_R) {
      FR_UNROLL
      for (int s = 0; s < TP_R; ++s) {
        // rotating window: the tap output 0 meets at this position enters slot (9 - s) % 9; output r meets the tap that
        // entered r steps ago, slot (r + 9 - s) % 9
        win[(TP_R - s) % TP_R] = wc[-(i + s)];
        const float v = yc[i + s];
        FR_UNROLL
        for (int r = 0; r < TP_R; ++r) acc[r] += win[(r + TP_R - s) % TP_R] * v;
      }
    }
    for (; i < ie; ++i) {      // tail (< 9 positions): plain reads
      const float v = yc[i];
      FR_UNROLL
      for (int r = 0; r < TP_R; ++r) acc[r] += wc[r - i] * v;
    }
    FR_UNROLL
    for (int r = 0; r < TP_R; ++r) pp[(c * 4 + iq) * 72 + g9 * TP_R + r] = acc[r];
  });
  run.phase([&](int tid) {
    if (tid < 72) {
      float s = b3;
      for (int k = 0; k < 32; ++k) s += pp[k * 72 + tid];
      const int h = TS_NO * og + tid;
      float t = 0.f;
      if (tid < TS_NO && h < TP_H) {
        xh[h] = s;
        const float d = target[h] - s;
        t = -0.5f * (LOG_2PI_F + (d * d) / (1.0f + EPSILON_F));
      }
      pp[32 * 72 - 72 + tid] = t;      // (row 31 of the partial sums is free once every thread has read it: next phase)
    }
  });
  run.phase([&](int tid) {
    if (tid == 0) {
      float t = 0.f;
      for (int k = 0; k < 72; ++k) t += pp[31 * 72 + k];
      nll8[og] = t;
    }
  });
}

constexpr int TS_HL = 129;                      // input gradient: reduction positions per thread (4 quarters)
constexpr int TS_BWD_LDS = 528 + 1040 + 4 * TP_H;
// d_y2[c][i] = sum_h g[h] w[c][h - i + 512] with g = d(xh) = (xh - target) / ((1 + 1e-6) F)  (model/vae.py:128;
// util/layers.py:159-167); the workgroup of channel 0 also stores g
template <class R>
FR_DEV void toep_split_bwd(R& run, float* lds, const float* xh /*[513]*/, const float* target /*[513]*/, const float* w3t /*[8][1028]*/,
                           int c, float invF, float* d_xh /*[513]*/, float* d_y2 /*[8][513] of the frame*/) {
  float* gs = lds;               // [513]
  float* wr = gs + 528;          // [1025] + zeros
  float* pp = wr + 1040;         // [4][513]
  run.phase([&](int tid) {
    for (int h = tid; h < TP_H; h += TS_T) {
      const float g = (target[h] - xh[h]) * (-invF / (1.0f + EPSILON_F));
      gs[h] = g;
      if (c == 0) d_xh[h] = g;
    }
    for (int i0 = tid; i0 < 1040; i0 += 5 * TS_T) {
      float v[5];
      FR_UNROLL
      for (int u = 0; u < 5; ++u) v[u] = i0 + u * TS_T < TP_K ? w3t[c * TP_W + i0 + u * TS_T] : 0.f;
      FR_UNROLL
      for (int u = 0; u < 5; ++u)
        if (i0 + u * TS_T < 1040) wr[i0 + u * TS_T] = v[u];
    }
  });
  run.phase([&](int tid) {
    if (tid >= TP_G * 4) return;
    const int gi = tid % TP_G, hq = tid / TP_G;
    const int hb = hq * TS_HL, he = imin_(TP_H, hb + TS_HL);
    const float* wc = wr + 512 - 9 * gi;                  // tap of output r at position h: wc[h - r]  (index in [0, 1024])
    float acc[TP_R], win[TP_R];
    FR_UNROLL
    for (int r = 0; r < TP_R; ++r) acc[r] = 0.f;
    FR_UNROLL
    for (int r = 1; r < TP_R; ++r) win[r] = wc[hb - r];
    int h = hb;
    for (; h + TP_R <= he; h += TP_R) {
      FR_UNROLL
      for (int s = 0; s < TP_R; ++s) {
        win[(TP_R - s) % TP_R] = wc[h + s];
        const float v = gs[h + s];
        FR_UNROLL
        for (int r = 0; r < TP_R; ++r) acc[r] += win[(r + TP_R - s) % TP_R] * v;
      }
    }
    for (; h < he; ++h) {
      const float v = gs[h];
      FR_UNROLL
      for (int r = 0; r < TP_R; ++r) acc[r] += wc[h - r] * v;
    }
    FR_UNROLL
    for (int r = 0; r < TP_R; ++r) pp[hq * TP_H + 9 * gi + r] = acc[r];
  });
  run.phase([&](int tid) {
    for (int i = tid; i < TP_H; i += TS_T) d_y2[c * TP_H + i] = (pp[i] + pp[TP_H + i]) + (pp[2 * TP_H + i] + pp[3 * TP_H + i]);
  });
}

// ------------------------------------------------------------------------------------------------ forward pass
struct FwdArgs {
  const float* P;         // flat parameters
  const float* pk;        // packed copies (Pk)
  POff off;
  const float* x;         // [F][513]
  const float* target;    // data argument of the log-density (x, or the shifted target of the VAWGAN generator step)
  const int64_t* y;       // [F]
  const float* eps;       // injected draw [F][128] or nullptr
  const float* z_in;      // decode-only: z [F][128]
  int ny;
  int F;
  int mode;               // FM_* bits
  float invF;
  // workspace tensors of the layered path (any may be null when its stage is off)
  float* enc_a[5];
  float* enc_st[5];
  float *z_mu, *z_lv, *z, *eps_out, *h;
  float* dec_a[3];
  float* dec_st[3];
  float *xh, *kl_f, *nll_f, *d_xh;
  float* dec_y;           // optional: activated output of decoder layer 2 [F][8][513] (operand of the last layer's weight gradient)
  float* y_enc[5];        // optional: activated outputs of the encoder layers / of decoder layers 0-1 (plain [F][C][H]):
  float* y_dec[2];        //           operands of the weight-gradient launch
};
static_assert(sizeof(FwdArgs) <= ARGS_FLOATS * 4, "argument block larger than its LDS slot");
constexpr int FM_ENC = 1, FM_SAMPLE = 2, FM_DEC = 4, FM_LOSS = 8, FM_GRAD = 16;
constexpr int FM_NOD3 = 32;   // the pass stops behind decoder layer 2 (its activated output in dec_y): the 1025-tap layer, the log-density
                              // and d(xh) come from the split launches below (toep_split_fwd / toep_split_bwd)

// optional global output pointer + offset (null stays null)
template <class T>
FR_DEV FR_G(T) ygp(T* p, size_t off) {
  return p ? fr_g(p) + off : (FR_G(T))nullptr;
}
// one conv + LayerNorm + lrelu layer after its `part` phase: outputs + statistics + activated input of the next layer
#define FR_LN_TAIL(CFG, NOUT_H, LOFF, C_, H_, NEXT_HP, NEXT_HL, a_g_, st_, y_)                                                 \
  reduce_sum<R, CFG::KS, CFG::NOUT, NOUT_H>(run, part, ch + LOFF, by, red);                                                   \
  var_sum<R, CFG::NOUT>(run, by, red, a_g_);                                                                                  \
  run.phase([&](int tid) { ln_apply<C_, H_, NEXT_HP, NEXT_HL>(tid, by, red, ch + LNP_C + LOFF, ch + 2 * LNP_C + LOFF, bx, st_, y_); });

template <class R>
FR_STAGE void frame_fwd_enc(R& run, float* lds_, const FwdArgs& a_, int f_) {
  const int f = FR_UNIFORM(f_);
  float* lds = fr_l(lds_);
  const FwdArgs& a = *fr_l(&a_);
  auto pk = fr_g(a.pk);
  (void)pk;
  float* bx = lds + L_BUFX;
  float* by = lds + L_BUFY;
  float* part = lds + L_PART;
  float* red = lds + L_RED;
  float* vec = lds + L_VEC;      // [0,256) z_mu | z_lv, [256,384) z
  const float* ch = lds + L_CH;
  auto P = fr_g(a.P);
  const POff& o = a.off;
  auto xf = fr_g(a.x) + (size_t)f * 513;
  run.phase([&](int tid) { halo_copy<1, 513, E0F::HP, E0F::PAD>(tid, xf, bx); });
  // ---- e0 .. e4: conv + bias -> pre-LN output (kept for the backward pass), statistics, LN + lrelu into the next input
  run.phase([&](int tid) { sconv_part<E0F>(tid, bx, P + FR_UNIFORM(o.ew[0]), part); });
  FR_LN_TAIL(E0F, E0F::HO, LNP_ENC0, 16, 171, E1F::HP, E1F::PAD, fr_g(a.enc_a[0]) + (size_t)f * E0F::NOUT, fr_g(a.enc_st[0]) + 2 * (size_t)f, ygp(a.y_enc[0], (size_t)f * E0F::NOUT))
  run.phase([&](int tid) { sconv_part<E1F>(tid, bx, P + FR_UNIFORM(o.ew[1]), part); });
  FR_LN_TAIL(E1F, E1F::HO, LNP_ENC1, 32, 57, E2F::HP, E2F::PAD, fr_g(a.enc_a[1]) + (size_t)f * E1F::NOUT, fr_g(a.enc_st[1]) + 2 * (size_t)f, ygp(a.y_enc[1], (size_t)f * E1F::NOUT))
  run.phase([&](int tid) { sconv_part<E2F>(tid, bx, P + FR_UNIFORM(o.ew[2]), part); });
  FR_LN_TAIL(E2F, E2F::HO, LNP_ENC2, 64, 19, E3F::HP, E3F::PAD, fr_g(a.enc_a[2]) + (size_t)f * E2F::NOUT, fr_g(a.enc_st[2]) + 2 * (size_t)f, ygp(a.y_enc[2], (size_t)f * E2F::NOUT))
  run.phase([&](int tid) { sconv_part<E3F>(tid, bx, P + FR_UNIFORM(o.ew[3]), part); });
  FR_LN_TAIL(E3F, E3F::HO, LNP_ENC3, 128, 7, E4F::HP, E4F::PAD, fr_g(a.enc_a[3]) + (size_t)f * E3F::NOUT, fr_g(a.enc_st[3]) + 2 * (size_t)f, ygp(a.y_enc[3], (size_t)f * E3F::NOUT))
  run.phase([&](int tid) { sconv_part<E4F>(tid, bx, P + FR_UNIFORM(o.ew[4]), part); });
  // C-major flatten (slim.flatten of the NCHW tensor, model/vae.py:79): index c*3 + h = the plain layout
  FR_LN_TAIL(E4F, E4F::HO, LNP_ENC4, 256, 3, 3, 0, fr_g(a.enc_a[4]) + (size_t)f * E4F::NOUT, fr_g(a.enc_st[4]) + 2 * (size_t)f, ygp(a.y_enc[4], (size_t)f * E4F::NOUT))
  // ---- heads (model/vae.py:80-81)
  run.phase([&](int tid) { dense4_part<HeadsF>(tid, bx, pk + Pk::heads, part); });
  run.phase([&](int tid) {
    if (tid < 256) {
      float s = tid < 128 ? P[FR_UNIFORM(o.bmu) + tid] : P[FR_UNIFORM(o.blv) + tid - 128];
      for (int k = 0; k < HeadsF::KS; ++k) s += part[k * 256 + tid];
      vec[tid] = s;
      if (tid < 128) a.z_mu[(size_t)f * 128 + tid] = s;
      else a.z_lv[(size_t)f * 128 + tid - 128] = s;
    }
  });
}

// eps of element (f, d): injected, or drawn by the caller-supplied functor (device: Philox; host emulation: injected only)
template <class R, class Eps>
FR_STAGE void frame_fwd_mid(R& run, float* lds_, const FwdArgs& a_, int f_, Eps&& draw) {
  const int f = FR_UNIFORM(f_);
  float* lds = fr_l(lds_);
  const FwdArgs& a = *fr_l(&a_);
  auto pk = fr_g(a.pk);
  (void)pk;
  float* bx = lds + L_BUFX;
  float* by = lds + L_BUFY;
  float* part = lds + L_PART;
  float* red = lds + L_RED;
  float* vec = lds + L_VEC;
  auto P = fr_g(a.P);
  const POff& o = a.off;
  if (a.mode & FM_SAMPLE) {
    // ---- sampler + KL (util/layers.py:152-156, 170-183 with mu2 = lv2 = 0)
    run.reduce(red + R_S3, [&](int tid) {
      if (tid >= 128) return 0.f;
      const float mu = vec[tid], lv = vec[128 + tid], v = expf(lv);
      const float e = draw(f, tid);
      if (fr_g(a.eps_out)) a.eps_out[(size_t)f * 128 + tid] = e;
      const float z = mu + e * sqrtf(v);
      vec[256 + tid] = z;
      a.z[(size_t)f * 128 + tid] = z;
      return 0.5f * ((0.f - lv) + (v + mu * mu) / (1.0f + EPSILON_F) - 1.0f);
    });
  } else if (a.mode & FM_DEC) {
    // decode-only / conversion path: z given (model/vae.py:139-145: encode returns z_mu, decode takes any z)
    const bool has_z = a.z_in != nullptr;
    auto zs = fr_g(a.z_in) + (size_t)f * 128;
    run.phase([&](int tid) {
      if (tid < 128) vec[256 + tid] = has_z ? zs[tid] : vec[tid];
    });
  }
  if (a.mode & FM_DEC) {
    // ---- embedding lookup + merge (model/vae.py:51-61, 89): h = z Wz + T[y]
    run.phase([&](int tid) {
      if (tid == 0 && (a.mode & FM_SAMPLE)) a.kl_f[f] = sum16(red + R_S3);
      merge_part(tid, vec + 256, P + FR_UNIFORM(o.wz), part);
    });
    int64_t yid = a.y[f];
    yid = yid < 0 ? 0 : (yid >= a.ny ? a.ny - 1 : yid);
    auto T = pk + Pk::mtab + (size_t)yid * MERGE_N;
    run.phase([&](int tid) {
      for (int n = tid; n < MERGE_N; n += NT) {
        const float s = part[n] + part[MERGE_N + n] + T[n];
        by[n] = s;
        if (fr_g(a.h)) a.h[(size_t)f * MERGE_N + n] = s;
      }
    });
    run.phase([&](int tid) { halo_copy<81, 19, D0F::HP, D0F::HL>(tid, by, bx); });
  } else if (a.mode & FM_SAMPLE) {
    run.phase([&](int tid) {
      if (tid == 0) a.kl_f[f] = sum16(red + R_S3);
    });
  }
  (void)bx;
}

template <class R>
FR_STAGE void frame_fwd_dec(R& run, float* lds_, const FwdArgs& a_, int f_) {
  const int f = FR_UNIFORM(f_);
  float* lds = fr_l(lds_);
  const FwdArgs& a = *fr_l(&a_);
  auto pk = fr_g(a.pk);
  (void)pk;
  float* bx = lds + L_BUFX;
  float* by = lds + L_BUFY;
  float* part = lds + L_PART;
  float* red = lds + L_RED;
  const float* ch = lds + L_CH;
  auto P = fr_g(a.P);
  const POff& o = a.off;
  // ---- d0 .. d2: conv_transpose + bias, LayerNorm, lrelu (model/vae.py:96-102)
  run.phase([&](int tid) { tconv_part<D0F>(tid, bx, pk + Pk::d0f, part); });
  FR_LN_TAIL(D0F, D0F::HOUT, LNP_DEC0, 32, 57, D1F::HP, D1F::HL, fr_g(a.dec_a[0]) + (size_t)f * D0F::NOUT, fr_g(a.dec_st[0]) + 2 * (size_t)f, ygp(a.y_dec[0], (size_t)f * D0F::NOUT))
  run.phase([&](int tid) { tconv_part<D1F>(tid, bx, pk + Pk::d1f, part); });
  FR_LN_TAIL(D1F, D1F::HOUT, LNP_DEC1, 16, 171, D2F::HP, D2F::HL, fr_g(a.dec_a[1]) + (size_t)f * D1F::NOUT, fr_g(a.dec_st[1]) + 2 * (size_t)f, ygp(a.y_dec[1], (size_t)f * D1F::NOUT))
  run.phase([&](int tid) { tconv_part<D2F>(tid, bx, pk + Pk::d2f, part); });
  reduce_sum<R, D2F::KS, D2F::NOUT, D2F::HOUT>(run, part, ch + LNP_DEC2, by, red);
  var_sum<R, D2F::NOUT>(run, by, red, fr_g(a.dec_a[2]) + (size_t)f * D2F::NOUT);
  if (a.mode & FM_NOD3) {     // uniform
    run.phase([&](int tid) {
      ln_apply<8, 513, 513, 0>(tid, by, red, ch + LNP_C + LNP_DEC2, ch + 2 * LNP_C + LNP_DEC2, bx, fr_g(a.dec_st[2]) + 2 * (size_t)f, ygp(a.dec_y, (size_t)f * 4104));
    });
    return;
  }
  run.phase([&](int tid) {
    // taps of the last layer, one contiguous row per channel (loads first: they must not queue behind the stores below)
    for (int i = tid; i < TP_C * TP_W; i += NT) part[i] = pk[Pk::w3t + i];
    ln_apply<8, 513, 513, 0>(tid, by, red, ch + LNP_C + LNP_DEC2, ch + 2 * LNP_C + LNP_DEC2, bx, fr_g(a.dec_st[2]) + 2 * (size_t)f, ygp(a.dec_y, (size_t)f * 4104));
    if (fr_g(a.dec_y))      // (same thread, same elements as ln_apply wrote)
      for (int i = tid; i < 4104; i += NT) a.dec_y[(size_t)f * 4104 + i] = bx[i];
  });
  // ---- d3: the 1025-tap layer, no LayerNorm, no activation (model/vae.py:96-103)
  run.phase([&](int tid) { toep_fwd_part(tid, bx, part, part + TP_C * TP_W); });
  const float b3 = P[FR_UNIFORM(o.db[3])];
  if (a.mode & FM_LOSS) {
    // ---- output + Gaussian log-density with unit variance (util/layers.py:159-167) and its gradient
    auto tf = fr_g(a.target) + (size_t)f * TP_H;
    run.reduce(red + R_S3, [&](int tid) {
      float t = 0.f;
      for (int p = tid; p < TP_H; p += NT) {
        float s = b3;
        for (int k = 0; k < 16; ++k) s += part[TP_C * TP_W + k * TP_H + p];
        const float d = tf[p] - s;
        t += -0.5f * (LOG_2PI_F + (d * d) / (1.0f + EPSILON_F));
        a.xh[(size_t)f * TP_H + p] = s;
        if (a.mode & FM_GRAD) a.d_xh[(size_t)f * TP_H + p] = d * (-a.invF / (1.0f + EPSILON_F));
      }
      return t;
    });
    run.phase([&](int tid) {
      if (tid == 0) a.nll_f[f] = sum16(red + R_S3);
    });
  } else {
    run.phase([&](int tid) {
      for (int p = tid; p < TP_H; p += NT) {
        float s = b3;
        for (int k = 0; k < 16; ++k) s += part[TP_C * TP_W + k * TP_H + p];
        a.xh[(size_t)f * TP_H + p] = s;
      }
    });
  }
}

// once per workgroup, before its first frame (both passes)
template <class R>
FR_DEV void frame_prologue(R& run, float* lds, const float* P, const POff& o) {
  auto Pg = fr_g(P);
  run.phase([&](int tid) { load_channel_vectors(tid, Pg, o, lds + L_CH); });
}

template <class R, class Eps>
FR_DEV void frame_fwd(R& run, float* lds, const FwdArgs& a, int f, Eps&& draw) {
  if (a.mode & FM_ENC) frame_fwd_enc(run, lds, a, f);
  if (a.mode & (FM_SAMPLE | FM_DEC)) frame_fwd_mid(run, lds, a, f, draw);
  if (a.mode & FM_DEC) frame_fwd_dec(run, lds, a, f);
}

// ------------------------------------------------------------------------------------------------ backward pass
// per-frame, per-channel sums the LayerNorm backward leaves for the parameter gradients (reduced over frames by the
// weight-gradient launch): lnp[(f*3 + k)*LNP_C + LNP_OFF[layer] + c], k = 0: d(offset) = sum_h dn, 1: d(scale) =
// sum_h dn * xhat, 2: d(conv bias) = sum_h d(pre-LN output)

struct BwdArgs {
  const float* P;
  const float* pk;
  POff off;
  const float* target;    // data argument of the log-density ([F][513]; x or the shifted target)
  const float* eps;       // the draw the forward pass used [F][128]
  int F;
  float invF;
  // forward tensors
  const float* enc_a[5];
  const float* enc_st[5];
  const float *z_mu, *z_lv;
  const float* dec_a[3];
  const float* dec_st[3];
  const float* xh;
  // gradients written for the weight-gradient launch
  float* d_xh;
  float* d_dec_a[3];
  float* d_h;
  float *d_z, *d_z_mu, *d_z_lv;
  float* d_enc_a[5];
  float* lnp;             // [F][3][LNP_C]
  const float* d_y2;      // non-null: the gradient at decoder layer 2's activated output [F][8][513] comes from toep_split_bwd
                          // (which also wrote d_xh); the pass then starts at that layer's LayerNorm backward
};
static_assert(sizeof(BwdArgs) <= ARGS_FLOATS * 4, "argument block larger than its LDS slot");

// LayerNorm + lrelu backward of one frame (autodiff of util/layers.py:32-44,149):
//   n = gamma xhat + beta, dn = dy lrelu'(n), dx = dn gamma, da = rstd (dx - mean(dx) - xhat mean(dx xhat))
// Entered after the phase that left dy in `by` (plain [C][H]) and the pre-LN tensor a in `bx` with {mean, rstd} in
// red[R_ST..]; leaves da in `by` (the caller flushes it to HBM under its next long phase) and the channel sums in lnp.
FR_DEV float ln_dn(float a, float dy, float mean, float rstd, float g, float b, float& xh) {
  xh = (a - mean) * rstd;
  const float nn = xh * g + b;
  return dy * (nn >= 0.f ? 1.0f : LEAK_F);
}
template <class R, int C, int H>
FR_DEV void ln_bwd(R& run, float* bx, float* by, float* part, float* red, const float* gamma, const float* beta,
                   FR_G(float) lnp_f /* + layer offset, stride LNP_C */) {   // (gamma / beta: the LDS copies)
  constexpr int N = C * H;
  constexpr int SEGS = imin_(NT / C, H), SLEN = cdiv_(H, SEGS);
  run.reduce2(red + R_S1, red + R_S2, [&](int tid, float& s1, float& s2) {
    const float mean = red[R_ST], rstd = red[R_ST + 1];
    s1 = 0.f;
    s2 = 0.f;
    for (int i = tid; i < N; i += NT) {
      const int c = i / H;
      float xh;
      const float dx = ln_dn(bx[i], by[i], mean, rstd, gamma[c], beta[c], xh) * gamma[c];
      s1 += dx;
      s2 += dx * xh;
    }
    // channel sums over position segments: A = sum dn, B = sum dn xhat, X = sum xhat
    if (tid < C * SEGS) {
      const int c = tid / SEGS, sg = tid % SEGS;
      float sa = 0.f, sb = 0.f, sx = 0.f;
      for (int h = sg * SLEN; h < imin_(H, (sg + 1) * SLEN); ++h) {
        float xh;
        const float dn = ln_dn(bx[c * H + h], by[c * H + h], mean, rstd, gamma[c], beta[c], xh);
        sa += dn;
        sb += dn * xh;
        sx += xh;
      }
      part[tid] = sa;
      part[C * SEGS + tid] = sb;
      part[2 * C * SEGS + tid] = sx;
    }
  });
  run.phase([&](int tid) {
    const float mean = red[R_ST], rstd = red[R_ST + 1];
    const float m1 = sum16(red + R_S1) * (1.0f / N), m2 = sum16(red + R_S2) * (1.0f / N);
    if (tid < C) {
      float sa = 0.f, sb = 0.f, sx = 0.f;
      for (int i = 0; i < SEGS; ++i) {
        sa += part[tid * SEGS + i];
        sb += part[C * SEGS + tid * SEGS + i];
        sx += part[2 * C * SEGS + tid * SEGS + i];
      }
      lnp_f[tid] = sa;
      lnp_f[LNP_C + tid] = sb;
      lnp_f[2 * LNP_C + tid] = rstd * (gamma[tid] * sa - (float)H * m1 - m2 * sx);
    }
    for (int i = tid; i < N; i += NT) {
      const int c = i / H;
      float xh;
      const float dx = ln_dn(bx[i], by[i], mean, rstd, gamma[c], beta[c], xh) * gamma[c];
      by[i] = rstd * (dx - m1 - xh * m2);
    }
  });
}
// the phase in front of ln_bwd: dy = sum of the K slices -> `by`, the layer's pre-LN tensor -> `bx`, its statistics
template <int KS, int NOUT>
FR_DEV void reduce_load(int tid, const float* part, float* by, FR_G(const float) a_g, float* bx, FR_G(const float) st_g, float* red) {
  for (int i = tid; i < NOUT; i += NT) {
    float s = 0.f;
    FR_UNROLL
    for (int k = 0; k < KS; ++k) s += part[k * NOUT + i];
    by[i] = s;
    bx[i] = a_g[i];
  }
  if (tid == 0) {
    red[R_ST] = st_g[0];
    red[R_ST + 1] = st_g[1];
  }
}

template <class R>
FR_STAGE void frame_bwd_dec(R& run, float* lds_, const BwdArgs& a_, int f_) {
  const int f = FR_UNIFORM(f_);
  float* lds = fr_l(lds_);
  const BwdArgs& a = *fr_l(&a_);
  auto pk = fr_g(a.pk);
  (void)pk;
  float* bx = lds + L_BUFX;
  float* by = lds + L_BUFY;
  float* part = lds + L_PART;
  float* red = lds + L_RED;
  const float* ch = lds + L_CH;
  float* vec = lds + L_VEC;
  auto P = fr_g(a.P);
  const POff& o = a.off;
  auto lnp_f = fr_g(a.lnp) + (size_t)f * 3 * LNP_C;
  if (a.d_y2) {   // uniform: split step
    auto dy = fr_g(a.d_y2) + (size_t)f * 4104;
    auto ag = fr_g(a.dec_a[2]) + (size_t)f * 4104;
    auto sg = fr_g(a.dec_st[2]) + 2 * (size_t)f;
    run.phase([&](int tid) {
      for (int i = tid; i < 4104; i += NT) {
        by[i] = dy[i];
        bx[i] = ag[i];
      }
      if (tid == 0) {
        red[R_ST] = sg[0];
        red[R_ST + 1] = sg[1];
      }
    });
  } else {
  // ---- d(xh) of G = -logP + D_KL (model/vae.py:128; util/layers.py:159-167): (xh - x) / ((1 + 1e-6) F)
  run.phase([&](int tid) {
    for (int p = tid; p < TP_H; p += NT) {
      const float d = a.target[(size_t)f * TP_H + p] - a.xh[(size_t)f * TP_H + p];
      const float g = d * (-a.invF / (1.0f + EPSILON_F));
      vec[p] = g;
      a.d_xh[(size_t)f * TP_H + p] = g;
    }
    for (int i = tid; i < TP_C * TP_W; i += NT) part[i] = pk[Pk::w3t + i];
  });
  // ---- d3 input gradient, LayerNorm backward of decoder layer 2
  run.phase([&](int tid) { toep_dgrad_part(tid, vec, part, part + TP_C * TP_W); });
  run.phase([&](int tid) { reduce_load<2, 4104>(tid, part + TP_C * TP_W, by, fr_g(a.dec_a[2]) + (size_t)f * 4104, bx, fr_g(a.dec_st[2]) + 2 * (size_t)f, red); });
  }
  ln_bwd<R, 8, 513>(run, bx, by, part, red, ch + LNP_C + LNP_DEC2, ch + 2 * LNP_C + LNP_DEC2, lnp_f + LNP_DEC2);
  run.phase([&](int tid) {
    flush(tid, by, fr_g(a.d_dec_a[2]) + (size_t)f * 4104, 4104);
    halo_copy<8, 513, D2G::HP, D2G::PAD>(tid, by, bx);
  });
  run.phase([&](int tid) {
    sconv_part<D2G>(tid, bx, pk + Pk::d2g, part);
  });
  run.phase([&](int tid) { reduce_load<D2G::KS, D2G::NOUT>(tid, part, by, fr_g(a.dec_a[1]) + (size_t)f * 2736, bx, fr_g(a.dec_st[1]) + 2 * (size_t)f, red); });
  ln_bwd<R, 16, 171>(run, bx, by, part, red, ch + LNP_C + LNP_DEC1, ch + 2 * LNP_C + LNP_DEC1, lnp_f + LNP_DEC1);
  run.phase([&](int tid) {
    flush(tid, by, fr_g(a.d_dec_a[1]) + (size_t)f * 2736, 2736);
    halo_copy<16, 171, D1G::HP, D1G::PAD>(tid, by, bx);
  });
  run.phase([&](int tid) {
    sconv_part<D1G>(tid, bx, pk + Pk::d1g, part);
  });
  run.phase([&](int tid) { reduce_load<D1G::KS, D1G::NOUT>(tid, part, by, fr_g(a.dec_a[0]) + (size_t)f * 1824, bx, fr_g(a.dec_st[0]) + 2 * (size_t)f, red); });
  ln_bwd<R, 32, 57>(run, bx, by, part, red, ch + LNP_C + LNP_DEC0, ch + 2 * LNP_C + LNP_DEC0, lnp_f + LNP_DEC0);
  run.phase([&](int tid) {
    flush(tid, by, fr_g(a.d_dec_a[0]) + (size_t)f * 1824, 1824);
    halo_copy<32, 57, D0G::HP, D0G::PAD>(tid, by, bx);
  });
  run.phase([&](int tid) {
    sconv_part<D0G>(tid, bx, pk + Pk::d0g, part);
  });
  run.phase([&](int tid) {
    for (int i = tid; i < MERGE_N; i += NT) {
      float s = 0.f;
      FR_UNROLL
      for (int k = 0; k < D0G::KS; ++k) s += part[k * MERGE_N + i];
      by[i] = s;
      a.d_h[(size_t)f * MERGE_N + i] = s;
    }
  });
}

template <class R>
FR_STAGE void frame_bwd_mid(R& run, float* lds_, const BwdArgs& a_, int f_) {
  const int f = FR_UNIFORM(f_);
  float* lds = fr_l(lds_);
  const BwdArgs& a = *fr_l(&a_);
  auto pk = fr_g(a.pk);
  (void)pk;
  float* bx = lds + L_BUFX;
  float* by = lds + L_BUFY;
  float* part = lds + L_PART;
  float* red = lds + L_RED;
  const float* ch = lds + L_CH;
  float* vec = lds + L_VEC;
  auto P = fr_g(a.P);
  const POff& o = a.off;
  auto lnp_f = fr_g(a.lnp) + (size_t)f * 3 * LNP_C;
  // ---- merge: d(z) = d(h) Wz^T; sampler + KL backward (util/layers.py:152-156, 170-183)
  run.phase([&](int tid) { dense4_part<MergeG>(tid, by, pk + Pk::wzT, part); });
  run.phase([&](int tid) {
    if (tid < 128) {
      float dz = 0.f;
      for (int k = 0; k < MergeG::KS; ++k) dz += part[k * 128 + tid];
      const size_t e = (size_t)f * 128 + tid;
      const float mu = a.z_mu[e], lv = a.z_lv[e], v = expf(lv);
      const float dmu = dz + mu / (1.0f + EPSILON_F) * a.invF;
      const float dlv = dz * (0.5f * a.eps[e] * sqrtf(v)) + 0.5f * (v / (1.0f + EPSILON_F) - 1.0f) * a.invF;
      vec[256 + tid] = dmu;
      vec[384 + tid] = dlv;
      if (fr_g(a.d_z)) a.d_z[e] = dz;
      a.d_z_mu[e] = dmu;
      a.d_z_lv[e] = dlv;
    }
  });
  // ---- heads: d(y4) = [dz_mu | dz_lv] [Wmu | Wlv]^T, LayerNorm backward of encoder layer 4
  run.phase([&](int tid) { dense4_part<HeadsG>(tid, vec + 256, pk + Pk::headsT, part); });
  run.phase([&](int tid) { reduce_load<HeadsG::KS, 768>(tid, part, by, fr_g(a.enc_a[4]) + (size_t)f * 768, bx, fr_g(a.enc_st[4]) + 2 * (size_t)f, red); });
  ln_bwd<R, 256, 3>(run, bx, by, part, red, ch + LNP_C + LNP_ENC4, ch + 2 * LNP_C + LNP_ENC4, lnp_f + LNP_ENC4);
  run.phase([&](int tid) {
    flush(tid, by, fr_g(a.d_enc_a[4]) + (size_t)f * 768, 768);
    halo_copy<256, 3, E4G::HP, E4G::HL>(tid, by, bx);
  });
}

template <class R>
FR_STAGE void frame_bwd_enc(R& run, float* lds_, const BwdArgs& a_, int f_) {
  const int f = FR_UNIFORM(f_);
  float* lds = fr_l(lds_);
  const BwdArgs& a = *fr_l(&a_);
  auto pk = fr_g(a.pk);
  (void)pk;
  float* bx = lds + L_BUFX;
  float* by = lds + L_BUFY;
  float* part = lds + L_PART;
  float* red = lds + L_RED;
  const float* ch = lds + L_CH;
  auto P = fr_g(a.P);
  const POff& o = a.off;
  auto lnp_f = fr_g(a.lnp) + (size_t)f * 3 * LNP_C;
  run.phase([&](int tid) {
    tconv_part<E4G>(tid, bx, pk + Pk::e4g, part);
  });
  run.phase([&](int tid) { reduce_load<E4G::KS, E4G::NOUT>(tid, part, by, fr_g(a.enc_a[3]) + (size_t)f * 896, bx, fr_g(a.enc_st[3]) + 2 * (size_t)f, red); });
  ln_bwd<R, 128, 7>(run, bx, by, part, red, ch + LNP_C + LNP_ENC3, ch + 2 * LNP_C + LNP_ENC3, lnp_f + LNP_ENC3);
  run.phase([&](int tid) {
    flush(tid, by, fr_g(a.d_enc_a[3]) + (size_t)f * 896, 896);
    halo_copy<128, 7, E3G::HP, E3G::HL>(tid, by, bx);
  });
  run.phase([&](int tid) {
    tconv_part<E3G>(tid, bx, pk + Pk::e3g, part);
  });
  run.phase([&](int tid) { reduce_load<E3G::KS, E3G::NOUT>(tid, part, by, fr_g(a.enc_a[2]) + (size_t)f * 1216, bx, fr_g(a.enc_st[2]) + 2 * (size_t)f, red); });
  ln_bwd<R, 64, 19>(run, bx, by, part, red, ch + LNP_C + LNP_ENC2, ch + 2 * LNP_C + LNP_ENC2, lnp_f + LNP_ENC2);
  run.phase([&](int tid) {
    flush(tid, by, fr_g(a.d_enc_a[2]) + (size_t)f * 1216, 1216);
    halo_copy<64, 19, E2G::HP, E2G::HL>(tid, by, bx);
  });
  run.phase([&](int tid) {
    tconv_part<E2G>(tid, bx, pk + Pk::e2g, part);
  });
  run.phase([&](int tid) { reduce_load<E2G::KS, E2G::NOUT>(tid, part, by, fr_g(a.enc_a[1]) + (size_t)f * 1824, bx, fr_g(a.enc_st[1]) + 2 * (size_t)f, red); });
  ln_bwd<R, 32, 57>(run, bx, by, part, red, ch + LNP_C + LNP_ENC1, ch + 2 * LNP_C + LNP_ENC1, lnp_f + LNP_ENC1);
  run.phase([&](int tid) {
    flush(tid, by, fr_g(a.d_enc_a[1]) + (size_t)f * 1824, 1824);
    halo_copy<32, 57, E1G::HP, E1G::HL>(tid, by, bx);
  });
  run.phase([&](int tid) {
    tconv_part<E1G>(tid, bx, pk + Pk::e1g, part);
  });
  run.phase([&](int tid) { reduce_load<E1G::KS, E1G::NOUT>(tid, part, by, fr_g(a.enc_a[0]) + (size_t)f * 2736, bx, fr_g(a.enc_st[0]) + 2 * (size_t)f, red); });
  ln_bwd<R, 16, 171>(run, bx, by, part, red, ch + LNP_C + LNP_ENC0, ch + 2 * LNP_C + LNP_ENC0, lnp_f + LNP_ENC0);
  run.phase([&](int tid) { flush(tid, by, fr_g(a.d_enc_a[0]) + (size_t)f * 2736, 2736); });
}

template <class R>
FR_DEV void frame_bwd(R& run, float* lds, const BwdArgs& a, int f) {
  frame_bwd_dec(run, lds, a, f);
  frame_bwd_mid(run, lds, a, f);
  frame_bwd_enc(run, lds, a, f);
}

}  // namespace frame
}  // namespace vaenpvc
