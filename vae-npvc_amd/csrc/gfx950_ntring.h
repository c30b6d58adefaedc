// gfx950_ntring.h -- C = A B^T on two operand planes with the machine of the four-wave weight-gradient kernels (round 6): one workgroup
// per CU, ONE wave per SIMD with a 128 x 64 wave tile (256 x 128 per workgroup, 8 accumulator tiles = 128 AGPRs), operands by LDS-DMA
// (global_load_lds_dwordx4, inline assembly, counted vmcnt) into a ring of per-(operand, plane) K chunks, one bare s_barrier per PHASE,
// one piece of side work (a fragment read or a request) behind each MFMA.  For the K-long dense-shaped sites of k_gemm_nt (encoder layer 4
// as a dense layer, forward + input gradient; the heads' forward GEMM): model/vae.py:79-82, util/layers.py:56-64.
//
// What is different from the two round-3 attempts at this (256 x 128 tiles with 32-k stages of all planes, then
// 16-k stages in a ring of six -- both no faster than the two-barrier 128 x 128 loop, because 64-/32-byte requests of 128-byte lines carry
// the L2 -> L1 path twice / four times):
//   * a stage is 64 k = ONE FULL 128-byte line per row and plane, and a DMA instruction fetches 8 full lines (8 rows x 128 B).  The LDS
//     image of such a block is 8 rows x 8 pieces of 16 bytes with the piece index XORed by (row >> 1) & 7: the 16 lanes a ds_read_b128
//     serves at a time (16 consecutive rows, one logical piece) hit 16 different bank groups;
//   * 64-k stages of all four operand planes (96 KB) do not fit a ring, so the ring turns per (operand, plane): within a K chunk the three
//     products run as three PHASES of 32 MFMAs -- A0 B1, A0 B0, A1 B0 -- with ALL FOUR k-steps of a plane's fragments held in registers
//     (48 fragments = 192 VGPRs; a plane chunk is read from LDS exactly once and its slot is free one phase later).  Three A slots of
//     32 KB + three B slots of 16 KB = 144 KB; a request has 3 - 4 phases (>= 3 000 MFMA cycles) to land.
// Per 64-k chunk and wave: 96 MFMAs, 48 fragment reads (ds_read_b128), 24 requests -- LDS bytes per MFMA 25 % below the 64 x 64 wave tiles
// of k_gemm_nt, no LDS staging stores, a third of its barriers.
#pragma once
#include "gfx950_planegemm.h"

namespace vaenpvc {
namespace tuned {

constexpr int NR_BM = 256, NR_BN = 128, NR_BK = 64;
constexpr int NR_AS = NR_BM * NR_BK * 2, NR_BS = NR_BN * NR_BK * 2;   // bytes of an A / B slot (one plane, one chunk): 32 768 / 16 384
constexpr int NR_RING = 3 * NR_AS + 3 * NR_BS;                        // 147 456
constexpr int NR_EP_PITCH = NR_BN + 4;
constexpr int NR_EP_LDS = NR_BM * NR_EP_PITCH * 4;                    // 135 168: the fp32 result tile (after the loop, in the idle ring)
constexpr int NR_LDS = NR_RING > NR_EP_LDS ? NR_RING : NR_EP_LDS;

// serves: two planes, no speaker table, whole row tiles not required (rows past M are clamped and never stored), N a multiple of 128 in the
// packed weights (every site's B planes are padded to whole column tiles), K padded to 64
inline bool gemm_nt_ring_serves(const NtArgs& a) {
  return !a.rowbias && a.Kp % NR_BK == 0 && a.Kp >= 4 * NR_BK && (!a.C2 || a.split % NR_BN == 0) && (a.ldc % 4) == 0 &&
         (reinterpret_cast<uintptr_t>(a.C) % 16) == 0 && (!a.C2 || reinterpret_cast<uintptr_t>(a.C2) % 16 == 0);
}

// LDS-DMA with a scalar base + per-lane 32-bit byte offset (the planes of this launch are < 2 GB)
__device__ __forceinline__ void lds_dma16_s(const unsigned char* sbase, unsigned voff, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}

__global__ void __launch_bounds__(256) k_gemm_nt_ring(NtArgs a) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = cdiv(a.N, NR_BN);
  const int tile = xcd_contiguous(blockIdx.x, gridDim.x);
  const int m0 = (tile / ntn) * NR_BM, n0 = (tile % ntn) * NR_BN;
  const int nch = a.Kp / NR_BK;
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem;

  // ---- requests.  Block b of a slot = rows 8 b .. 8 b + 7 (1 KB); wave w requests the blocks b = w + 4 j.  Lane i -> row 8 b + (i >> 3),
  //      LDS piece i & 7, which holds the logical piece (i & 7) ^ ((row >> 1) & 7) = (i & 7) ^ ((4 (w & 1) + (i >> 4)) & 7) of the row's chunk
  const int dpc = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  unsigned aoffs[8], boffs[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    int r = m0 + 8 * (wave + 4 * j) + (lane >> 3);
    r = r < a.M ? r : a.M - 1;                        // rows past the end: duplicates, never stored
    aoffs[j] = (unsigned)r * (unsigned)(a.Kp * 2) + dpc * 16;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) boffs[j] = (unsigned)(n0 + 8 * (wave + 4 * j) + (lane >> 3)) * (unsigned)(a.Kp * 2) + dpc * 16;
  const unsigned char* A8 = reinterpret_cast<const unsigned char*>(a.A);
  const unsigned char* B8 = reinterpret_cast<const unsigned char*>(a.B);
  // request j (of 8 / of 4) of plane `pl`, chunk `kc` (clamped: requests past the last chunk re-read it, nobody reads them) into slot `slot`
  auto dma_a = [&](int j, int pl, int kc, int slot) __attribute__((always_inline)) {
    const int k = kc < nch ? kc : nch - 1;
    lds_dma16_s(A8 + (size_t)pl * a.a_plane * 2 + (size_t)k * (NR_BK * 2), aoffs[j], lds0 + slot * NR_AS + (wave + 4 * j) * 1024);
  };
  auto dma_b = [&](int j, int pl, int kc, int slot) __attribute__((always_inline)) {
    const int k = kc < nch ? kc : nch - 1;
    lds_dma16_s(B8 + (size_t)pl * a.b_plane * 2 + (size_t)k * (NR_BK * 2), boffs[j], lds0 + 3 * NR_AS + slot * NR_BS + (wave + 4 * j) * 1024);
  };

  // ---- fragment addresses.  Row r of a slot sits at r * 128; logical piece p = 2 ks + lh of row r at LDS piece p ^ ((r >> 1) & 7)
  //      (for the rows of one MFMA tile (r >> 1) & 7 = (l31 >> 1) & 7: per lane a constant)
  const int sw = (l31 >> 1) & 7;
  int aq[4], bq[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int q = (2 * ks + lh) ^ sw;
    aq[ks] = (128 * wm + l31) * 128 + q * 16;      // + 4096 per row tile
    bq[ks] = (64 * wn + l31) * 128 + q * 16;
  }
  u32x4 FA0[4][4], FA1[4][4], FB0[2][4], FB1[2][4];   // [tile][ks]
  auto rdA = [&](u32x4 (&F)[4][4], int slot, int idx) __attribute__((always_inline)) {   // idx = ks * 4 + t
    const int ks = idx >> 2, t = idx & 3;
    F[t][ks] = *reinterpret_cast<const u32x4*>(smem + slot * NR_AS + aq[ks] + t * 4096);
  };
  auto rdB = [&](u32x4 (&F)[2][4], int slot, int idx) __attribute__((always_inline)) {   // idx = ks * 2 + u
    const int ks = idx >> 1, u = idx & 1;
    F[u][ks] = *reinterpret_cast<const u32x4*>(smem + 3 * NR_AS + slot * NR_BS + bq[ks] + u * 4096);
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int u = 0; u < 2; ++u) acc[t][u] = zero16();
  // MFMA m of a phase: ks outermost (the eight MFMAs of a k-step touch eight different accumulators)
  auto mm = [&](const u32x4 (&FA)[4][4], const u32x4 (&FB)[2][4], int m) __attribute__((always_inline)) {
    const int ks = m >> 3, t = (m >> 1) & 3, u = m & 1;
    acc[t][u] = mfma_bf16(FA[t][ks], FB[u][ks], acc[t][u]);
  };
  auto phase_end = [&](auto n_) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of the slot that is requested into next are complete
    wait_vmcnt<decltype(n_)::value>();
    __builtin_amdgcn_s_barrier();
  };

  // ---- ring.  Streams in the order their chunks are READ:  A: A0(0) A1(0) A0(1) A1(1) ...   B: B1(0) B0(0) B1(1) B0(1) ...   slot = index % 3.
  //      Chunk c:  phase 1 multiplies A0 B1, reads FB0(c),           requests B0(c+1) then A1(c+1)
  //                phase 2 multiplies A0 B0, reads FA1(c),           requests B1(c+2)
  //                phase 3 multiplies A1 B0, reads FA0(c+1) FB1(c+1), requests A0(c+2)
  //      A slot of stream index s: s % 3;  A0(c) = 2c, A1(c) = 2c + 1;  B1(c) = 2c, B0(c) = 2c + 1.
  // prologue: the requests phases 1 .. 3 of chunks -1 would have issued, FA0(0) / FB1(0) read serially
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_a(j, 0, 0, 0);         // A0(0) -> A slot 0
#pragma unroll
  for (int j = 0; j < 4; ++j) dma_b(j, 1, 0, 0);         // B1(0) -> B slot 0
#pragma unroll
  for (int j = 0; j < 4; ++j) dma_b(j, 0, 0, 1);         // B0(0) -> B slot 1
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_a(j, 1, 0, 1);         // A1(0) -> A slot 1
#pragma unroll
  for (int j = 0; j < 4; ++j) dma_b(j, 1, 1, 2);         // B1(1) -> B slot 2
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_a(j, 0, 1, 2);         // A0(1) -> A slot 2
  wait_vmcnt<24>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 16; ++i) rdA(FA0, 0, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) rdB(FB1, 0, i);
  phase_end(IntC<20>{});                                  // B0(0) has landed everywhere; A slot 0 / B slot 0 are free

  int sa = 0, sb = 0;   // A0(c) = A slot sa, A1(c) = sa + 1, A0(c+1) = sa + 2 (mod 3); B1(c) = sb, B0(c) = sb + 1, B1(c+1) = sb + 2 (mod 3)
  auto m3 = [](int x) { return x >= 3 ? x - 3 : x; };
  for (int c = 0; c < nch; ++c) {
    const int sa1 = m3(sa + 1), sa2 = m3(sa + 2), sb1 = m3(sb + 1), sb2 = m3(sb + 2);
    // ---- phase 1: A0 B1; reads FB0(c) (B slot sb1); requests B0(c+1) -> B slot sb (B1(c) was read a phase ago), A1(c+1) -> A slot sa
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      mm(FA0, FB1, m);
      if (m < 8) rdB(FB0, sb1, m);
      else if (m < 12) dma_b(m - 8, 0, c + 1, sb);
      else if (m < 20) dma_a(m - 12, 1, c + 1, sa);
      __builtin_amdgcn_sched_barrier(0);
    }
    phase_end(IntC<24>{});                                // A1(c) has landed
    // ---- phase 2: A0 B0; reads FA1(c) (A slot sa1); requests B1(c+2) -> B slot sb1 (B0(c) was read in phase 1)
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      mm(FA0, FB0, m);
      if (m < 16) rdA(FA1, sa1, m);
      else if (m < 20) dma_b(m - 16, 1, c + 2, sb1);
      __builtin_amdgcn_sched_barrier(0);
    }
    phase_end(IntC<16>{});                                // A0(c+1), B1(c+1) have landed
    // ---- phase 3: A1 B0; reads FA0(c+1) (A slot sa2), FB1(c+1) (B slot sb2); requests A0(c+2) -> A slot sa1 (A1(c) was read in phase 2)
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      mm(FA1, FB0, m);
      if (m < 16) rdA(FA0, sa2, m);
      else if (m < 24) rdB(FB1, sb2, m - 16);
      else dma_a(m - 24, 0, c + 2, sa1);
      __builtin_amdgcn_sched_barrier(0);
    }
    phase_end(IntC<20>{});                                // B0(c+1) has landed
    sa = sa2;
    sb = sb2;
  }
  wait_vmcnt<0>();   // requests past the last chunk are still writing into the ring
  __builtin_amdgcn_s_barrier();

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
    if (m >= a.M) continue;
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

}  // namespace tuned
}  // namespace vaenpvc
