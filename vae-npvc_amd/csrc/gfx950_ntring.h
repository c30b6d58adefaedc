// gfx950_ntring.h -- C = A B^T on two operand planes with the machine of the four-wave weight-gradient kernels (round 6): one workgroup
// per CU, ONE wave per SIMD with a (32 TM) x (32 TN) wave tile, operands by LDS-DMA (global_load_lds_dwordx4, inline assembly, counted
// vmcnt) into a ring of per-(operand, plane) K chunks, one bare s_barrier per PHASE, one piece of side work (a fragment read or a request)
// behind each MFMA.  Users:
//   k_gemm_nt_ring   (TM, TN = 4, 2: 256 x 128 tiles) every two-plane C = A B^T launch with K >= 256, no speaker table and >= 128 tiles: encoder
//                    layer 4 as a dense layer and the heads, forward + input gradient, and the merge input gradient (model/vae.py:51-61,79-82,
//                    util/layers.py:56-64); the gains are at the K-long sites (K = 768 / 896), K = 256 and the one-column-tile merge site run equal;
//   k_cgemm_sf_ring  (4, 2) encoder layer 3 forward as a view GEMM whose 252-row tile owns 36 WHOLE frames, with the layer's LayerNorm
//                    statistics and activated operand planes in its epilogue (the successor of k_cgemm_sf; util/layers.py:47-66).
//
// What is different from the two round-3 attempts at this (256 x 128 tiles with 32-k stages of all planes, then 16-k stages in a ring of six
// -- both no faster than the two-barrier 128 x 128 loop, because 64- / 32-byte requests of 128-byte lines carry the L2 -> L1 path twice / four
// times):
//   * a stage is 64 k = ONE FULL 128-byte line per row and plane, and a DMA instruction fetches 8 full lines (8 rows x 128 B).  The LDS
//     image of such a block is 8 rows x 8 pieces of 16 bytes with the piece index XORed by (row >> 1) & 7: the four 16-lane groups a
//     ds_read_b128 is served in ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... of 32 consecutive rows, one logical piece) each hit 16
//     different bank groups (SQ_LDS_BANK_CONFLICT = 0, profiles/r06_ring_pmc.txt);
//   * 64-k stages of all four operand planes (96 KB) do not fit a ring, so the ring turns per (operand, plane): within a K chunk the three
//     products run as three PHASES of 4 TM TN MFMAs -- A0 B1, A0 B0, A1 B0 -- with ALL FOUR k-steps of a plane's fragments held in
//     registers (8 (TM + TN) fragments = 192 VGPRs; a plane chunk is read from LDS exactly once and its slot is free one phase later).
//     Three A slots + three B slots = 144 KB; a request has 3 - 4 phases (>= 3 000 MFMA cycles) to land: the counted waits cost nothing
//     (ablation: 135.5 -> 131.5 us without them).
// Per 64-k chunk and wave (4, 2): 96 MFMAs, 48 fragment reads (ds_read_b128), 24 requests -- LDS bytes per MFMA 25 % below the 64 x 64 wave
// tiles of k_gemm_nt, no LDS staging stores, a third of its barriers.  Measured (encoder layer 4 forward, 32 768 frames): 168 -> 134 us, matrix
// pipes 31 -> 43 % busy.  What bounds it now is the ISSUE of the requests: without them the kernel takes 95 us (one LDS-DMA instruction
// costs ~80 cycles of a wave's issue; 24 per chunk against 96 MFMAs of 32 cycles), without fragment reads 124, without result stores 124.
#pragma once
#include "gfx950_planegemm.h"

namespace vaenpvc {
namespace tuned {

#ifndef VAENPVC_NR_ABL
#define VAENPVC_NR_ABL 0   // developer ablation of the main loop (wrong results): 1 no requests, 2 no fragment reads, 4 no wait / barrier at a phase end, 8 no result stores
#endif
constexpr int NR_BK = 64;
template <int TM, int TN>
struct NrCfg {
  static constexpr int BM = 64 * TM, BN = 64 * TN;                  // workgroup tile: 2 x 2 waves of (32 TM) x (32 TN)
  static constexpr int AS = BM * NR_BK * 2, BS = BN * NR_BK * 2;    // bytes of an A / B slot (one plane, one chunk)
  static constexpr int RING = 3 * AS + 3 * BS;
  static constexpr int NA = 2 * TM, NB = 2 * TN;                    // requests per wave and slot
  static constexpr int NM = 4 * TM * TN;                            // MFMAs per wave and phase
  static_assert(4 * TN + NB + NA <= NM && 4 * TM + NB <= NM && 4 * TM + 4 * TN + NA <= NM, "one piece of side work per MFMA");
  static_assert(RING <= 160 * 1024, "ring fits the LDS");
};

// LDS-DMA with a scalar base + per-lane 32-bit byte offset (the planes of a launch are < 2 GB)
__device__ __forceinline__ void lds_dma16_s(const unsigned char* sbase, unsigned voff, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
// per-lane piece of DMA block j of this wave: row = 8 (wave + 4 j) + (lane >> 3) of the tile, piece (lane & 7) ^ ((row >> 1) & 7)
__device__ __forceinline__ int nr_dma_piece(int wave, int lane) { return (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7); }

// The K loop.  aoffs / boffs: byte offsets of this lane's rows (+ 16 x its piece) from the plane bases A8 / B8; acc: the wave's tile.
template <int TM, int TN>
__device__ __forceinline__ void nr_mainloop(unsigned char* smem, const unsigned char* A8, const unsigned char* B8, size_t a_plane_bytes,
                                            size_t b_plane_bytes, const unsigned (&aoffs)[2 * TM], const unsigned (&boffs)[2 * TN], int nch,
                                            f32x16 (&acc)[TM][TN]) {
  using T = NrCfg<TM, TN>;
  const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;
  // request j of plane `pl`, chunk `kc` (clamped: requests past the last chunk re-read it, nobody reads them) into slot `slot`
  auto dma_a = [&](int j, int pl, int kc, int slot) __attribute__((always_inline)) {
    const int k = kc < nch ? kc : nch - 1;
    lds_dma16_s(A8 + (size_t)pl * a_plane_bytes + (size_t)k * (NR_BK * 2), aoffs[j], lds0 + slot * T::AS + (wave + 4 * j) * 1024);
  };
  auto dma_b = [&](int j, int pl, int kc, int slot) __attribute__((always_inline)) {
    const int k = kc < nch ? kc : nch - 1;
    lds_dma16_s(B8 + (size_t)pl * b_plane_bytes + (size_t)k * (NR_BK * 2), boffs[j], lds0 + 3 * T::AS + slot * T::BS + (wave + 4 * j) * 1024);
  };
  // fragment addresses.  Row r of a slot sits at r * 128; logical piece p = 2 ks + lh of row r at LDS piece p ^ ((r >> 1) & 7)
  // (for the rows of one MFMA tile (r >> 1) & 7 = (l31 >> 1) & 7: per lane a constant)
  const int sw = (l31 >> 1) & 7;
  int aq[4], bq[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int q = (2 * ks + lh) ^ sw;
    aq[ks] = (32 * TM * wm + l31) * 128 + q * 16;      // + 4096 per row tile
    bq[ks] = (32 * TN * wn + l31) * 128 + q * 16;
  }
  u32x4 FA0[TM][4], FA1[TM][4], FB0[TN][4], FB1[TN][4];   // [tile][ks]
  auto rdA = [&](u32x4 (&F)[TM][4], int slot, int idx) __attribute__((always_inline)) {   // idx = ks * TM + t
    if constexpr ((VAENPVC_NR_ABL & 2) != 0) return;
    const int ks = idx / TM, t = idx - ks * TM;
    F[t][ks] = *reinterpret_cast<const u32x4*>(smem + slot * T::AS + aq[ks] + t * 4096);
  };
  auto rdB = [&](u32x4 (&F)[TN][4], int slot, int idx) __attribute__((always_inline)) {   // idx = ks * TN + u
    if constexpr ((VAENPVC_NR_ABL & 2) != 0) return;
    const int ks = idx / TN, u = idx - ks * TN;
    F[u][ks] = *reinterpret_cast<const u32x4*>(smem + 3 * T::AS + slot * T::BS + bq[ks] + u * 4096);
  };
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u) acc[t][u] = zero16();
  // MFMA m of a phase: ks outermost (the TM TN MFMAs of a k-step touch as many different accumulators)
  auto mm = [&](const u32x4 (&FA)[TM][4], const u32x4 (&FB)[TN][4], int m) __attribute__((always_inline)) {
    const int ks = m / (TM * TN), r = m - ks * (TM * TN), t = r / TN, u = r - t * TN;
    acc[t][u] = mfma_bf16(FA[t][ks], FB[u][ks], acc[t][u]);
  };
  auto phase_end = [&](auto n_) __attribute__((always_inline)) {
    if constexpr ((VAENPVC_NR_ABL & 5) != 0) {
      if constexpr ((VAENPVC_NR_ABL & 4) == 0) __builtin_amdgcn_s_barrier();
      return;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of the slot that is requested into next are complete
    wait_vmcnt<decltype(n_)::value>();
    __builtin_amdgcn_s_barrier();
  };
  constexpr int NA = T::NA, NB = T::NB, NM = T::NM;
  // ---- ring.  Streams in the order their chunks are READ:  A: A0(0) A1(0) A0(1) A1(1) ...   B: B1(0) B0(0) B1(1) B0(1) ...   slot = index % 3.
  //      Chunk c:  phase 1 multiplies A0 B1, reads FB0(c),            requests B0(c+1) then A1(c+1)
  //                phase 2 multiplies A0 B0, reads FA1(c),            requests B1(c+2)
  //                phase 3 multiplies A1 B0, reads FA0(c+1) FB1(c+1), requests A0(c+2)
  // prologue: the requests phases 1 .. 3 of chunk -1 would have issued, FA0(0) / FB1(0) read serially
#pragma unroll
  for (int j = 0; j < NA; ++j) dma_a(j, 0, 0, 0);         // A0(0) -> A slot 0
#pragma unroll
  for (int j = 0; j < NB; ++j) dma_b(j, 1, 0, 0);         // B1(0) -> B slot 0
#pragma unroll
  for (int j = 0; j < NB; ++j) dma_b(j, 0, 0, 1);         // B0(0) -> B slot 1
#pragma unroll
  for (int j = 0; j < NA; ++j) dma_a(j, 1, 0, 1);         // A1(0) -> A slot 1
#pragma unroll
  for (int j = 0; j < NB; ++j) dma_b(j, 1, 1, 2);         // B1(1) -> B slot 2
#pragma unroll
  for (int j = 0; j < NA; ++j) dma_a(j, 0, 1, 2);         // A0(1) -> A slot 2
  wait_vmcnt<2 * NA + 2 * NB>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 4 * TM; ++i) rdA(FA0, 0, i);
#pragma unroll
  for (int i = 0; i < 4 * TN; ++i) rdB(FB1, 0, i);
  phase_end(IntC<2 * NA + NB>{});                         // B0(0) has landed everywhere; A slot 0 / B slot 0 are free

  int sa = 0, sb = 0;   // A0(c) = A slot sa, A1(c) = sa + 1, A0(c+1) = sa + 2 (mod 3); B1(c) = sb, B0(c) = sb + 1, B1(c+1) = sb + 2 (mod 3)
  auto m3 = [](int x) { return x >= 3 ? x - 3 : x; };
  for (int c = 0; c < nch; ++c) {
    const int sa1 = m3(sa + 1), sa2 = m3(sa + 2), sb1 = m3(sb + 1), sb2 = m3(sb + 2);
    // ---- phase 1: A0 B1; reads FB0(c) (B slot sb1); requests B0(c+1) -> B slot sb (B1(c) was read a phase ago), A1(c+1) -> A slot sa
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      mm(FA0, FB1, m);
      if (m < 4 * TN) rdB(FB0, sb1, m);
      else if (m < 4 * TN + NB) { if (!(VAENPVC_NR_ABL & 1)) dma_b(m - 4 * TN, 0, c + 1, sb); }
      else if (m < 4 * TN + NB + NA) { if (!(VAENPVC_NR_ABL & 1)) dma_a(m - 4 * TN - NB, 1, c + 1, sa); }
      __builtin_amdgcn_sched_barrier(0);
    }
    phase_end(IntC<2 * NA + 2 * NB>{});                   // A1(c) has landed
    // ---- phase 2: A0 B0; reads FA1(c) (A slot sa1); requests B1(c+2) -> B slot sb1 (B0(c) was read in phase 1)
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      mm(FA0, FB0, m);
      if (m < 4 * TM) rdA(FA1, sa1, m);
      else if (m < 4 * TM + NB) { if (!(VAENPVC_NR_ABL & 1)) dma_b(m - 4 * TM, 1, c + 2, sb1); }
      __builtin_amdgcn_sched_barrier(0);
    }
    phase_end(IntC<NA + 2 * NB>{});                       // A0(c+1), B1(c+1) have landed
    // ---- phase 3: A1 B0; reads FA0(c+1) (A slot sa2), FB1(c+1) (B slot sb2); requests A0(c+2) -> A slot sa1 (A1(c) was read in phase 2)
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      mm(FA1, FB0, m);
      if (m < 4 * TM) rdA(FA0, sa2, m);
      else if (m < 4 * TM + 4 * TN) rdB(FB1, sb2, m - 4 * TM);
      else if (m < 4 * TM + 4 * TN + NA) { if (!(VAENPVC_NR_ABL & 1)) dma_a(m - 4 * TM - 4 * TN, 0, c + 2, sa1); }
      __builtin_amdgcn_sched_barrier(0);
    }
    phase_end(IntC<2 * NA + NB>{});                       // B0(c+1) has landed
    sa = sa2;
    sb = sb2;
  }
  wait_vmcnt<0>();   // requests past the last chunk are still writing into the ring
  __builtin_amdgcn_s_barrier();
}

// ---------------------------------------------------------------- plain C = A B^T (the dense-shaped sites)
constexpr int NR_BM = 256, NR_BN = 128;
constexpr int NR_EP_PITCH = NR_BN + 4;
constexpr int NR_EP_LDS = NR_BM * NR_EP_PITCH * 4;                    // 135 168: the fp32 result tile (after the loop, in the idle ring)
constexpr int NR_LDS = NrCfg<4, 2>::RING > NR_EP_LDS ? NrCfg<4, 2>::RING : NR_EP_LDS;

// serves: two planes, no speaker table, K >= 4 chunks of 64, aligned 16-byte result pieces (rows past M are clamped and never stored; every
// site's B planes are padded to whole column tiles)
inline bool gemm_nt_ring_serves(const NtArgs& a) {
  return !a.rowbias && a.Kp % NR_BK == 0 && a.Kp >= 4 * NR_BK && (!a.C2 || a.split % NR_BN == 0) && (a.ldc % 4) == 0 &&
         (reinterpret_cast<uintptr_t>(a.C) % 16) == 0 && (!a.C2 || reinterpret_cast<uintptr_t>(a.C2) % 16 == 0);
}

// (Persistent workgroups -- one per CU walking its XCD's tile range, the next tile's first requests right behind the result stores -- were
//  built and measured equal: 134.7 vs 135.6 us on encoder layer 4 forward, same box.  Removed.)
__global__ void __launch_bounds__(256) k_gemm_nt_ring(NtArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = cdiv(a.N, NR_BN);
  const int tile = xcd_contiguous(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * NR_BM, n0 = (tile % ntn) * NR_BN;
  const int dpc = nr_dma_piece(wave, lane);
  unsigned aoffs[8], boffs[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int r = m0 + 8 * (wave + 4 * j) + (lane >> 3);
    r = r < a.M ? r : a.M - 1;                        // rows past the end: duplicates, never stored
    aoffs[j] = (unsigned)r * (unsigned)(a.Kp * 2) + dpc * 16;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) boffs[j] = (unsigned)(n0 + 8 * (wave + 4 * j) + (lane >> 3)) * (unsigned)(a.Kp * 2) + dpc * 16;
  f32x16 acc[4][2];
  nr_mainloop<4, 2>(smem, reinterpret_cast<const unsigned char*>(a.A), reinterpret_cast<const unsigned char*>(a.B), (size_t)a.a_plane * 2,
                    (size_t)a.b_plane * 2, aoffs, boffs, a.Kp / NR_BK, acc);

  // ---- epilogue: the 256 x 128 tile through LDS, 16-byte stores of whole 512-byte row runs (as k_gemm_nt's; bias added on the way out)
  float* ot = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        ot[(128 * wm + 32 * t + acc_row(reg, lane)) * NR_EP_PITCH + 64 * wn + 32 * u + l31] = acc[t][u][reg];
  const int pc = tid & 31, r0 = tid >> 5, n = n0 + pc * 4;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (a.bias) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bv[k] = a.bias[min(n + k, a.N - 1)];
  }
  __syncthreads();
  const bool second = a.C2 && n0 >= a.split;
  float* cb = (second ? a.C2 : a.C) + (n - (second ? a.split : 0));
  const bool whole = n + 3 < a.N;
#pragma unroll 4
  for (int i = 0; i < NR_BM / 8; ++i) {
    const int row = r0 + 8 * i, m = m0 + row;
    f32x4 v = *reinterpret_cast<const f32x4*>(ot + row * NR_EP_PITCH + pc * 4);
    v += bv;
    if (m >= a.M || ((VAENPVC_NR_ABL & 8) && v[0] != 12345.678f)) continue;
    float* o = cb + (int64_t)m * a.ldc;
    if (whole) {
      *reinterpret_cast<f32x4*>(o) = v;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (n + k < a.N) o[k] = v[k];
    }
  }
}

inline void launch_gemm_nt_ring(const NtArgs& a, hipStream_t s) {
  const int ntiles = cdiv(a.M, NR_BM) * cdiv(a.N, NR_BN);
  rt().ensure_lds(reinterpret_cast<const void*>(&k_gemm_nt_ring), NR_LDS);
  hipLaunchKernelGGL(k_gemm_nt_ring, dim3((unsigned)ntiles), dim3(256), NR_LDS, s, a);
}

// ---------------------------------------------------------------- encoder layer 3 forward (the successor of k_cgemm_sf)
// The view GEMM out[f][o][j] = sum_{t,c} X[f][3 j + t][c] W[t][c][o] (gfx950_viewconv.h: CV_E3F, K = 7 x 64 = 448 = 7 chunks) with the ROWS of the
// ring kernel = the view rows (f, j) -- 36 whole frames x 7 positions = 252 of the tile's 256 rows, each row's K run one contiguous, line-aligned
// stretch of the channel-last planes -- and its 128 columns = the layer's output channels (the weight planes [128][448]).  The finished tile is
// re-ordered through LDS into [frame][channel][position] fp32 (pitch 904) and the frame pass of k_cgemm_sf runs on it, one wave per frame: bias,
// the layer's LayerNorm statistics (two-pass), the fp32 pre-LN frame out as 16-byte pieces of a contiguous run, and the ACTIVATED frame as the
// bf16 operand planes pl_y3 of encoder layer 4's GEMMs.
constexpr int SFR_TF = 36, SFR_R = 7, SFR_C = 128, SFR_FOUT = SFR_C * SFR_R, SFR_FPITCH = SFR_FOUT + 8;
constexpr int SFR_EP_LDS = SFR_TF * SFR_FPITCH * 4;    // 130 176
constexpr int SFR_LDS = NrCfg<4, 2>::RING > SFR_EP_LDS ? NrCfg<4, 2>::RING : SFR_EP_LDS;
inline bool cgemm_sf_ring_serves(const CgArgs& a) {
  return a.M == 128 && a.mdiv == 128 && a.C == 128 && a.xv.R == 7 && a.OH == 7 && a.om == 7 && a.ofs == 128 * 7 && a.oq == 1 && a.o0 == 0 &&
         a.Kp % NR_BK == 0 && a.Kp >= 4 * NR_BK && a.N % 7 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && (a.xv.step % 64) == 0 &&
         (a.xv.fs % 64) == 0 && (a.xv.x0 % 64) == 0;
}
__global__ void __launch_bounds__(256) k_cgemm_sf_ring(CgSfArgs b) {
  const CgArgs& a = b.g;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_contiguous(blockIdx.x, gridDim.x), f0 = tile * SFR_TF, n0 = f0 * SFR_R;
  const int dpc = nr_dma_piece(wave, lane);
  unsigned aoffs[8], boffs[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int rl = 8 * (wave + 4 * j) + (lane >> 3);
    rl = rl < SFR_TF * SFR_R ? rl : SFR_TF * SFR_R - 1;       // the four idle rows and rows past the end: duplicates, never stored
    int r = n0 + rl;
    r = r < a.N ? r : a.N - 1;
    aoffs[j] = (unsigned)(view_off(a.xv, r) * 2) + dpc * 16;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) boffs[j] = (unsigned)(8 * (wave + 4 * j) + (lane >> 3)) * (unsigned)(a.Kp * 2) + dpc * 16;
  // (fetched BEFORE the K loop: 48 per-lane loads right behind it were an exposed round trip per tile)
  // per-lane constants of the frame pass: lane owns the 8-element pieces lane, lane + 64 of a frame's 896 = 112 x 8 elements;
  // element e = (channel e / 7, position e % 7)
  constexpr int P8 = SFR_FOUT / 8, PPL = cdiv(P8, 64);
  float gm[PPL][8], bt[PPL][8], bs[PPL][8];
#pragma unroll
  for (int u = 0; u < PPL; ++u)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int pc = lane + 64 * u, e = (pc < P8 ? pc : 0) * 8 + k, ch = e / SFR_R;
      gm[u][k] = b.gamma[ch];
      bt[u][k] = b.beta[ch];
      bs[u][k] = a.bias ? a.bias[ch] : 0.f;
    }
  f32x16 acc[4][2];
  nr_mainloop<4, 2>(smem, reinterpret_cast<const unsigned char*>(a.X), reinterpret_cast<const unsigned char*>(a.W), (size_t)a.x_plane * 2,
                    (size_t)a.w_plane * 2, aoffs, boffs, a.Kp / NR_BK, acc);
  // ---- the tile as [frame][channel][position] fp32 in LDS (raw conv sums; the bias is added in the frame pass)
  float* ot = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rl = 128 * wm + 32 * t + acc_row(reg, lane);
      if (rl >= SFR_TF * SFR_R) continue;
      const int fl = rl / SFR_R, q = rl - fl * SFR_R;
#pragma unroll
      for (int u = 0; u < 2; ++u) ot[fl * SFR_FPITCH + (64 * wn + 32 * u + l31) * SFR_R + q] = acc[t][u][reg];
    }
  __syncthreads();
  const int nf = min(SFR_TF, b.F - f0);
  constexpr float INVN = 1.0f / SFR_FOUT;
  // THREE frames of a wave at a time (frames wave, wave + 4, wave + 8 of every dozen): a frame is a dependent chain LDS read -> sum -> wave
  // reduction -> centred squares -> wave reduction -> stores (~2.5 us), and with one wave per SIMD nothing else hides it -- nine frames in a row
  // were most of this kernel's time (7 K chunks per tile: 10 us of MFMAs against ~25 us of frame pass).  The fragment registers are dead here.
  constexpr int NB3 = 3;
  for (int fb = wave; fb < SFR_TF; fb += 4 * NB3) {
    if (fb >= nf) break;                              // (uniform per wave)
    float v[NB3][PPL][8], mean[NB3], rstd[NB3];
    float sm[NB3], q2[NB3];
#pragma unroll
    for (int r = 0; r < NB3; ++r) {
      const int fl = min(fb + 4 * r, SFR_TF - 1);     // (frames past the tile's last: a duplicate, never stored)
      sm[r] = 0.f;
#pragma unroll
      for (int u = 0; u < PPL; ++u) {
        const int pc = lane + 64 * u;
        const bool ok = pc < P8;
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(ot + fl * SFR_FPITCH + (ok ? pc : 0) * 8);
        const f32x4 t1 = *reinterpret_cast<const f32x4*>(ot + fl * SFR_FPITCH + (ok ? pc : 0) * 8 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          v[r][u][k] = ok ? t0[k] + bs[u][k] : 0.f;
          v[r][u][4 + k] = ok ? t1[k] + bs[u][4 + k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) sm[r] += v[r][u][k];
      }
    }
#pragma unroll
    for (int r = 0; r < NB3; ++r) mean[r] = wave_sum(sm[r]) * INVN;
#pragma unroll
    for (int r = 0; r < NB3; ++r) {
      q2[r] = 0.f;
#pragma unroll
      for (int u = 0; u < PPL; ++u) {
        const bool ok = lane + 64 * u < P8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float d = v[r][u][k] - mean[r];
          q2[r] += ok ? d * d : 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < NB3; ++r) rstd[r] = 1.0f / sqrtf(wave_sum(q2[r]) * INVN + LN_EPS);
#pragma unroll
    for (int r = 0; r < NB3; ++r) {
      const int fl = fb + 4 * r;
      if (fl >= nf) continue;                         // (uniform per wave)
      const int f = f0 + fl;
      if (lane == 0) {
        b.st[2 * f] = mean[r];
        b.st[2 * f + 1] = rstd[r];
      }
      float* og = a.out + (int64_t)f * SFR_FOUT;
#pragma unroll
      for (int u = 0; u < PPL; ++u) {
        const int pc = lane + 64 * u;
        if (pc >= P8) continue;
        *reinterpret_cast<f32x4*>(og + pc * 8) = f32x4{v[r][u][0], v[r][u][1], v[r][u][2], v[r][u][3]};
        *reinterpret_cast<f32x4*>(og + pc * 8 + 4) = f32x4{v[r][u][4], v[r][u][5], v[r][u][6], v[r][u][7]};
        if (b.planes) {   // uniform
          float y8[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) y8[k] = lnact_v(v[r][u][k], mean[r], rstd[r], gm[u][k], bt[u][k]);
          u32x4 pk[2];
          pack8<2>(y8, pk);
#pragma unroll
          for (int p = 0; p < 2; ++p) *reinterpret_cast<u32x4*>(b.planes + ((int64_t)p * b.F + f) * SFR_FOUT + pc * 8) = pk[p];
        }
      }
    }
  }
}
inline void launch_cgemm_sf_ring(const CgSfArgs& b, hipStream_t s) {
  rt().ensure_lds(reinterpret_cast<const void*>(&k_cgemm_sf_ring), SFR_LDS);
  hipLaunchKernelGGL(k_cgemm_sf_ring, dim3((unsigned)cdiv(b.F, SFR_TF)), dim3(256), SFR_LDS, s, b);
}

// ---------------------------------------------------------------- encoder layer 3 input gradient + LayerNorm backward of layer 2 (successor of k_cgemm_pf<LNB>)
// The P-type view GEMM of CV_E3G (gfx950_viewconv.h: 3 output phases x 64 channels = 192 weight rows, K = 3 taps x 128 = 384 = 6 chunks, 8 view rows per
// frame) on the (3, 3) instance of the main loop: the ring's ROWS = the view rows of 24 WHOLE frames (192), its COLUMNS = the 192 stacked weight rows,
// wave tiles of 96 x 96.  Epilogue as k_cgemm_pf<LNB>: the tile re-ordered through LDS into [frame][channel][position] (pitch 1 224 floats), then the
// LayerNorm + lrelu backward of encoder layer 2 (autodiff of util/layers.py:32-44,149) on the frames in LDS, one wave per frame: a.out receives d(pre-LN
// output of layer 2), the per-channel sums leave as one row of `part` per workgroup.
constexpr int PFR_TF = 24, PFR_R = 8, PFR_C = 64, PFR_OH = 19, PFR_FOUT = PFR_C * PFR_OH, PFR_FPITCH = PFR_FOUT + 8;
constexpr int PFR_EP_LDS = PFR_TF * PFR_FPITCH * 4;    // 117 504
constexpr int PFR_LDS = NrCfg<3, 3>::RING > PFR_EP_LDS ? NrCfg<3, 3>::RING : PFR_EP_LDS;
static_assert(4 * 3 * PFR_FOUT * 4 <= PFR_LDS, "the per-element sums of four waves fit the LDS");
inline bool cgemm_pf_ring_serves(const CgArgs& a) {
  return a.M == 192 && a.mdiv == 64 && a.C == 64 && a.xv.R == 8 && a.OH == 19 && a.om == 19 && a.ofs == 64 * 19 && a.oq == 3 && a.o0s == 1 && !a.bias &&
         a.Kp % NR_BK == 0 && a.Kp >= 4 * NR_BK && a.N % 8 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 && (a.xv.step % 8) == 0 &&
         (a.xv.fs % 8) == 0 && (a.xv.x0 % 8) == 0;
}
__global__ void __launch_bounds__(256) k_cgemm_pf_ring(CgArgs a, CgLnbArgs lb) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = xcd_contiguous(blockIdx.x, gridDim.x), f0 = tile * PFR_TF, n0 = f0 * PFR_R;
  const int nf = min(PFR_TF, a.N / PFR_R - f0);
  const int dpc = nr_dma_piece(wave, lane);
  unsigned aoffs[6], boffs[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    int r = n0 + 8 * (wave + 4 * j) + (lane >> 3);
    r = r < a.N ? r : a.N - 1;                        // rows past the end: duplicates, never stored
    aoffs[j] = (unsigned)(view_off(a.xv, r) * 2) + dpc * 16;
    boffs[j] = (unsigned)(8 * (wave + 4 * j) + (lane >> 3)) * (unsigned)(a.Kp * 2) + dpc * 16;
  }
  f32x16 acc[3][3];
  nr_mainloop<3, 3>(smem, reinterpret_cast<const unsigned char*>(a.X), reinterpret_cast<const unsigned char*>(a.W), (size_t)a.x_plane * 2,
                    (size_t)a.w_plane * 2, aoffs, boffs, a.Kp / NR_BK, acc);
  // ---- LayerNorm operands of this wave's frames (fl = wave + 4 k): requested before the tile is re-ordered, so the loads fly under the LDS traffic
  float* ot = reinterpret_cast<float*>(smem);
  constexpr int P16 = PFR_FOUT / 4, PPL = cdiv(P16, 64), FPW = PFR_TF / 4;   // 16-byte pieces per frame / per lane; frames per wave
  f32x4 av[FPW][PPL];
  float gm[PPL][4], bt[PPL][4], fmean[FPW], frstd[FPW];
#pragma unroll
  for (int k = 0; k < FPW; ++k) {
    const int f = min(f0 + wave + 4 * k, f0 + nf - 1);
    fmean[k] = lb.st[2 * f];
    frstd[k] = lb.st[2 * f + 1];
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
      const int pc = min(lane + 64 * u, P16 - 1);
      av[k][u] = *reinterpret_cast<const f32x4*>(lb.a2 + (int64_t)f * PFR_FOUT + pc * 4);
    }
  }
#pragma unroll
  for (int u = 0; u < PPL; ++u)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ch = (min(lane + 64 * u, P16 - 1) * 4 + k) / PFR_OH;
      gm[u][k] = lb.gamma[ch];
      bt[u][k] = lb.beta[ch];
    }
  // ---- the tile as [frame][channel][position] fp32: ring row = (frame fl, view row q), ring column = phase * 64 + channel, position = 3 q + o0 + phase
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rl = 96 * wm + 32 * t + acc_row(reg, lane);
      const int fl = rl / PFR_R, q = rl - fl * PFR_R;
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int col = 96 * wn + 32 * u + l31, ph = col >> 6, ch = col & 63;
        const int pos = q * a.oq + a.o0 + ph;
        if (pos >= 0 && pos < PFR_OH) ot[fl * PFR_FPITCH + ch * PFR_OH + pos] = acc[t][u][reg];
      }
    }
  __syncthreads();
  constexpr float INVN = 1.0f / PFR_FOUT;
  float su[PPL][4], sw[PPL][4], sd[PPL][4];
#pragma unroll
  for (int u = 0; u < PPL; ++u)
#pragma unroll
    for (int k = 0; k < 4; ++k) su[u][k] = sw[u][k] = sd[u][k] = 0.f;
#pragma unroll
  for (int kf = 0; kf < FPW; ++kf) {
    const int fl = wave + 4 * kf;
    if (fl >= nf) break;       // (uniform per wave)
    const float mean = fmean[kf], rstd = frstd[kf];
    float dn[PPL][4], xh[PPL][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
      const int pc = lane + 64 * u;
      const bool ok = pc < P16;
      const f32x4 dy = *reinterpret_cast<const f32x4*>(ot + fl * PFR_FPITCH + (ok ? pc : 0) * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xh[u][k] = (av[kf][u][k] - mean) * rstd;
        const float nn = xh[u][k] * gm[u][k] + bt[u][k];
        dn[u][k] = ok ? dy[k] * (nn >= 0.f ? 1.0f : LEAK) : 0.f;
        const float dx = dn[u][k] * gm[u][k];
        s1 += dx;
        s2 += dx * xh[u][k];
      }
    }
    s1 = wave_sum(s1) * INVN;
    s2 = wave_sum(s2) * INVN;
    float* og = a.out + (int64_t)(f0 + fl) * PFR_FOUT;
#pragma unroll
    for (int u = 0; u < PPL; ++u) {
      const int pc = lane + 64 * u;
      if (pc >= P16) continue;
      f32x4 d;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d[k] = rstd * (dn[u][k] * gm[u][k] - s1 - xh[u][k] * s2);
        su[u][k] += dn[u][k] * xh[u][k];
        sw[u][k] += dn[u][k];
        sd[u][k] += d[k];
      }
      *reinterpret_cast<f32x4*>(og + pc * 4) = d;
    }
  }
  __syncthreads();    // every wave is done with the tile: the LDS now carries the per-element sums [wave][3][FOUT]
  float* ps = ot + wave * (3 * PFR_FOUT);
#pragma unroll
  for (int u = 0; u < PPL; ++u) {
    const int pc = lane + 64 * u;
    if (pc >= P16) continue;
    *reinterpret_cast<f32x4*>(ps + pc * 4) = f32x4{su[u][0], su[u][1], su[u][2], su[u][3]};
    *reinterpret_cast<f32x4*>(ps + PFR_FOUT + pc * 4) = f32x4{sw[u][0], sw[u][1], sw[u][2], sw[u][3]};
    *reinterpret_cast<f32x4*>(ps + 2 * PFR_FOUT + pc * 4) = f32x4{sd[u][0], sd[u][1], sd[u][2], sd[u][3]};
  }
  __syncthreads();
  if (tid < 3 * PFR_C) {
    const int which = tid / PFR_C, c = tid - which * PFR_C;
    float v = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4)
      for (int h = 0; h < PFR_OH; ++h) v += ot[w4 * (3 * PFR_FOUT) + which * PFR_FOUT + c * PFR_OH + h];
    lb.part[(int64_t)blockIdx.x * (3 * PFR_C) + tid] = v;
  }
}
// returns the number of rows written to lb.part
inline int launch_cgemm_pf_ring_lnb(const CgArgs& a, const CgLnbArgs& lb, hipStream_t s) {
  const int nwg = cdiv(a.N / PFR_R, PFR_TF);
  rt().ensure_lds(reinterpret_cast<const void*>(&k_cgemm_pf_ring), PFR_LDS);
  hipLaunchKernelGGL(k_cgemm_pf_ring, dim3((unsigned)nwg), dim3(256), PFR_LDS, s, a, lb);
  return nwg;
}

}  // namespace tuned
}  // namespace vaenpvc
