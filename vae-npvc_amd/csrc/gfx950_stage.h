// gfx950_stage.h -- HBM -> registers -> LDS staging of one activation tile, shared by the conv
// engine (gfx950_convgemm.h) and the weight-gradient kernel (gfx950_convwgrad.h).
//
// A tile = CH channels x H bins of TF consecutive frames of a [F][CHTOT][H] tensor (channels
// c0 .. c0+CH-1), laid out in LDS as  f*FSTR + ch*CSTR + LPAD + i  (zero halos around each row are
// written once by the kernel and never touched here).  Staging is split in two halves so that the
// global loads of tile t+1 are in flight while tile t is being multiplied:
//   gload()  : issue every global load of the tile into registers (clamped addresses, no branches);
//   lstore() : apply LN + leaky-relu (optional) and write the LDS tile.
//
// Every instruction here competes with another wave's MFMA stream for the SIMD's issue slots, so the
// code is organised to need as few VALU instructions per element as possible:
//   ROWS (H >= 32): wave w owns rows rr*NWV + w (row = one channel of one frame).  The row index is
//     wave-uniform: addresses, validity and the LayerNorm constants live on the scalar unit, LDS
//     addresses are one per-lane base + immediate offsets; per element: fma, mul, max, ds_write.
//   ELEM (short rows): per frame, element e = tid + NTHR*kk of the contiguous run of the tile's
//     channels; its (channel, bin) split, LDS address and gamma/beta do not depend on the frame and
//     are computed once per thread (init()); mean/rstd of a frame are scalar; per element: sub, mul,
//     fma, mul, max, ds_write.
// Frames past the end (nfr < TF) and channels past the tensor (nch < CH) are clamped on load and
// written as zeros (uniform branches, only taken in the last tile of a chunk).
#pragma once
#include "gfx950_common.h"

namespace vaenpvc {
namespace tuned {

template <int CH, int CHTOT, int H, int CSTR, int FSTR, int LPAD, bool LN, int TF, int NWV>
struct TileStager {
  static constexpr bool ROWS = H >= 32;
  static constexpr int NTHR = NWV * 64;
  static constexpr int NROWS = TF * CH, RPW = cdiv(NROWS, NWV), LPR = cdiv(H, 64);
  static constexpr bool RDIV = CH % NWV == 0;  // (frame, channel) of row rr*NWV + w = compile-time + w
  static constexpr bool RFULL = NROWS % NWV == 0;
  static constexpr int PERF = CH * H, KPF = cdiv(PERF, NTHR);
  static constexpr int NREG = ROWS ? RPW * LPR : TF * KPF;
  static constexpr int NE = ROWS ? 1 : KPF;
  static constexpr bool PARTIAL = CHTOT % CH != 0;  // the last channel tile of the tensor is not full
  float v[NREG];
  float* pk[NE];  // ELEM: LDS address of element kk in frame 0
  float eg[NE], eb[NE];
  // LayerNorm constants travel in registers, never through memory at lstore() time (a scalar or
  // vector load there is a full memory latency in the middle of the LDS stores): lane l of `stv`
  // holds word l of the tile's (mean, rstd) pairs (loaded with the tile in gload()), lane c of
  // `gv`/`bv` holds gamma/beta of channel c (ROWS; loaded once); lstore() broadcasts them with
  // v_readlane.
  float stv, gv, bv;
  // fast path (full tile): per-lane row constants, lane j <-> row j*NWV + w of this wave
  int rst_off;      // ROWS: word offset of row j's (mean, rstd) pair inside the tile's statistics
  float rg, rb;     // ROWS: gamma/beta of row j's channel
  float rmean, rrstd;
  int lane_t;       // lane index of the last 64-bin segment, clamped to H-1 (duplicates are benign)
  int eoff[NE];     // ELEM: clamped element offset inside a frame
  static_assert(!LN || 2 * TF <= 64, "statistics of a tile must fit one register");
  static_assert(!(LN && ROWS) || CH <= 64, "gamma/beta of a ROWS tile must fit one register");

  static __device__ __forceinline__ float bcast(float x, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
  }

  __device__ __forceinline__ void init(float* __restrict__ tile, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, int c0, int nch) {
    if constexpr (ROWS) {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      lane_t = lane + 64 * (LPR - 1);
      lane_t = lane_t < H ? lane_t : H - 1;
      if constexpr (LN) {
        int c = c0 + (lane < nch ? lane : nch - 1);
        gv = gamma[c];
        bv = beta[c];
        int r = (lane < RPW ? lane : RPW - 1) * NWV + wave;
        r = r < NROWS ? r : NROWS - 1;
        int f = r / CH, ch = r - f * CH;
        rst_off = 2 * f;
        rg = gamma[c0 + (ch < nch ? ch : nch - 1)];
        rb = beta[c0 + (ch < nch ? ch : nch - 1)];
      }
    }
    if constexpr (!ROWS) {
#pragma unroll
      for (int kk = 0; kk < KPF; ++kk) {
        int e = threadIdx.x + NTHR * kk;
        int ec = e < PERF ? e : PERF - 1;
        int ch = ec / H, i = ec - ch * H;
        pk[kk] = tile + ch * CSTR + LPAD + i;
        eoff[kk] = ec;
        bool ok = ch < nch;
        eg[kk] = ok ? 1.f : 0.f;
        eb[kk] = 0.f;
        if constexpr (LN) {
          eg[kk] = ok ? gamma[c0 + ch] : 0.f;
          eb[kk] = ok ? beta[c0 + ch] : 0.f;
        }
      }
    }
  }

  static __device__ __forceinline__ void row_of(int rr, int wave, int& f, int& ch) {
    if constexpr (RDIV) {
      f = (rr * NWV) / CH;
      ch = (rr * NWV) % CH + wave;
    } else {
      int r = rr * NWV + wave;
      f = r / CH;
      ch = r - f * CH;
    }
  }

  __device__ __forceinline__ void gload(const float* __restrict__ src, const float* __restrict__ st, int f0, int nfr,
                                        int c0, int nch) {
    const int tid = threadIdx.x, lane = tid & 63;
    if constexpr (LN) stv = st[2 * f0 + (lane < 2 * nfr ? lane : 2 * nfr - 1)];
    if (nfr == TF && nch == CH) {  // uniform: full tile, straight-line code with compile-time offsets
      if constexpr (ROWS) {
        static_assert(!LN || RPW <= 64, "row constants of a wave must fit one register");
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if constexpr (LN) {
          rmean = st[2 * f0 + rst_off];
          rrstd = st[2 * f0 + rst_off + 1];
        }
        const float* wb = src + ((int64_t)f0 * CHTOT + c0) * H;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          if (!RFULL && rr * NWV + wave >= NROWS) break;  // wave-uniform, last row only
          int f, ch;
          row_of(rr, wave, f, ch);
          const float* row = wb + (f * CHTOT + ch) * H;
#pragma unroll
          for (int p = 0; p < LPR; ++p) v[rr * LPR + p] = (p == LPR - 1) ? row[lane_t] : row[lane + 64 * p];
        }
      } else {
#pragma unroll
        for (int f = 0; f < TF; ++f) {
          const float* base = src + ((int64_t)(f0 + f) * CHTOT + c0) * H;
#pragma unroll
          for (int kk = 0; kk < KPF; ++kk) v[f * KPF + kk] = (NTHR * (kk + 1) <= PERF) ? base[tid + NTHR * kk] : base[eoff[kk]];
        }
      }
      return;
    }
    if constexpr (ROWS) {
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
      for (int rr = 0; rr < RPW; ++rr) {
        int f, ch;
        row_of(rr, wave, f, ch);
        bool ok = (RFULL || rr * NWV + wave < NROWS) && f < nfr && ch < nch;
        const float* row = src + ((int64_t)(f0 + (ok ? f : 0)) * CHTOT + c0 + (ok ? ch : 0)) * H;
#pragma unroll
        for (int p = 0; p < LPR; ++p) {
          int i = lane + 64 * p;
          if (64 * (p + 1) > H) i = i < H ? i : H - 1;  // duplicates of the last bin are never stored
          v[rr * LPR + p] = row[i];
        }
      }
    } else {
      const int nvalid = nch * H;
#pragma unroll
      for (int f = 0; f < TF; ++f) {
        const float* base = src + ((int64_t)(f0 + (f < nfr ? f : 0)) * CHTOT + c0) * H;
#pragma unroll
        for (int kk = 0; kk < KPF; ++kk) {
          int e = tid + NTHR * kk;
          v[f * KPF + kk] = base[e < nvalid ? e : nvalid - 1];  // clamped; zeroed through eg/eb
        }
      }
    }
  }

  __device__ __forceinline__ void lstore(float* __restrict__ tile, int nfr, int nch) {
    const int tid = threadIdx.x, lane = tid & 63;
    if (nfr == TF && nch == CH) {  // uniform: full tile
      if constexpr (ROWS) {
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        float scv = 1.f, shv = 0.f;
        if constexpr (LN) {  // constants of all rows of this wave at once (lane j <-> row j)
          scv = rrstd * rg;
          shv = rb - rmean * scv;
        }
        float* tb = tile + wave * CSTR + LPAD + lane;
        float* tt = tile + wave * CSTR + LPAD + lane_t - 64 * (LPR - 1);
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          if (!RFULL && rr * NWV + wave >= NROWS) break;  // wave-uniform
          int f, ch;
          row_of(rr, wave, f, ch);
          const int doff = f * FSTR + (ch - wave) * CSTR;  // compile-time when RDIV
          float sc = 1.f, sh = 0.f;
          if constexpr (LN) {
            sc = bcast(scv, rr);
            sh = bcast(shv, rr);
          }
#pragma unroll
          for (int p = 0; p < LPR; ++p) {
            float x = v[rr * LPR + p];
            if constexpr (LN) {
              x = x * sc + sh;
              x = fmaxf(x, LEAK * x);
            }
            if (p == LPR - 1) tt[doff + 64 * p] = x;
            else tb[doff + 64 * p] = x;
          }
        }
      } else {
#pragma unroll
        for (int f = 0; f < TF; ++f) {
          float rstd = 1.f, mean = 0.f;
          if constexpr (LN) {
            mean = bcast(stv, 2 * f);
            rstd = bcast(stv, 2 * f + 1);
          }
#pragma unroll
          for (int kk = 0; kk < KPF; ++kk) {
            float x = v[f * KPF + kk];
            if constexpr (LN) {  // (x - mean) first: no cancellation between two large products
              x = (x - mean) * (rstd * eg[kk]) + eb[kk];
              x = fmaxf(x, LEAK * x);
            }
            pk[kk][f * FSTR] = x;  // lanes past the row space hold a duplicate of the last element
          }
        }
      }
      return;
    }
    if constexpr (ROWS) {
      const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
      float* tb = tile + lane;
      constexpr int RB = RPW < 8 ? RPW : 8;
#pragma unroll
      for (int rr0 = 0; rr0 < RPW; rr0 += RB) {
        float sc[RB], sh[RB];
        bool okr[RB];
        int doff[RB];
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          int rr = rr0 + j < RPW ? rr0 + j : RPW - 1;
          int f, ch;
          row_of(rr, wave, f, ch);
          okr[j] = (RFULL || rr * NWV + wave < NROWS) && f < nfr && ch < nch;
          doff[j] = f * FSTR + ch * CSTR + LPAD;
          sc[j] = 1.f;
          sh[j] = 0.f;
          if constexpr (LN) {
            int fs = okr[j] ? f : 0, cs = okr[j] ? ch : 0;
            float mean = bcast(stv, 2 * fs), rstd = bcast(stv, 2 * fs + 1);
            sc[j] = rstd * bcast(gv, cs);
            sh[j] = bcast(bv, cs) - mean * sc[j];
          }
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
          if (rr0 + j >= RPW) continue;
          if (!RFULL && (rr0 + j) * NWV + wave >= NROWS) continue;  // wave-uniform
          float* dst = tb + doff[j];
          if (okr[j]) {  // wave-uniform
#pragma unroll
            for (int p = 0; p < LPR; ++p) {
              float x = v[(rr0 + j) * LPR + p];
              if constexpr (LN) {
                x = x * sc[j] + sh[j];
                x = fmaxf(x, LEAK * x);
              }
              if (64 * (p + 1) <= H || lane + 64 * p < H) dst[64 * p] = x;
            }
          } else {
#pragma unroll
            for (int p = 0; p < LPR; ++p)
              if (64 * (p + 1) <= H || lane + 64 * p < H) dst[64 * p] = 0.f;
          }
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < TF; ++f) {
        if (f < nfr) {  // uniform
          float rstd = 1.f, mean = 0.f;
          if constexpr (LN) {
            mean = bcast(stv, 2 * f);
            rstd = bcast(stv, 2 * f + 1);
          }
#pragma unroll
          for (int kk = 0; kk < KPF; ++kk) {
            float x = v[f * KPF + kk];
            if constexpr (LN) {
              x = (x - mean) * (rstd * eg[kk]) + eb[kk];
              x = fmaxf(x, LEAK * x);
            } else if constexpr (PARTIAL) {
              x = x * eg[kk];
            }
            if (NTHR * (kk + 1) <= PERF || tid + NTHR * kk < PERF) pk[kk][f * FSTR] = x;
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < KPF; ++kk)
            if (NTHR * (kk + 1) <= PERF || tid + NTHR * kk < PERF) pk[kk][f * FSTR] = 0.f;
        }
      }
    }
  }
};

}  // namespace tuned
}  // namespace vaenpvc
