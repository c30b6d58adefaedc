// gfx950_frame_wgrad.h -- every parameter gradient of a small batch in ONE launch (k_frame_wgrad).
//
// After the frame passes (gfx950_frame.h) every operand a parameter gradient needs sits in HBM / L2: the activated
// layer inputs y (written by the forward pass), the gradients d(pre-LN output) of every layer, d(h), d(z_mu), d(z_lv),
// d(xh) and the per-frame channel sums of the LayerNorm backward.  The layered path spends ~20 launches on them (two
// streams, ~100 us at 16 frames: a launch each, mostly latency); here the launch is a list of JOBS, a block looks its job
// up by block index:
//   conv      dW[t][a][b] = sum_f sum_j U[f][a][S j - PAD + t] V[f][b][j]   (8 conv / transposed-conv layers)
//   toeplitz  dW[t][c]    = sum_f sum_j y[f][c][j] g[f][j + t - 512]        (the 1025-tap layer)
//   outer     dW[r][c]    = sum_f A[f][r] B[f][c]                            (heads, merge, embedding FC)
//   small     biases of the un-normalised layers, LayerNorm parameter sums, speaker-embedding rows
// The conv and Toeplitz jobs split frames (and positions) over blocks and accumulate with fp32 atomics into the gradient
// buffer, which the step's pack launch zero-filled; everything else is written by exactly one thread.
// Reference: autodiff of model/vae.py:72-137 w.r.t. the 44 trainables (trainer/vae.py:24).
// Written against the phase runner of gfx950_frame.h: tests/frame_emu compiles it for the host.
#pragma once
#include "gfx950_frame.h"

namespace vaenpvc {
namespace frame {

constexpr int WT = 256;           // threads per block of the weight-gradient launch
constexpr int WG_LDS = 9216;      // floats of dynamic LDS (36 KB: several blocks per CU)

struct WgArgs {
  const float* P;
  POff off;
  const float* x;             // [F][513]
  const int64_t* y;           // [F]
  int ny;
  int F;
  const float* y_enc[5];      // activated encoder outputs [F][C][H]
  const float* z;             // [F][128]
  const float* h;             // [F][1539]
  const float* y_dec[2];      // activated decoder outputs 0, 1
  const float* dec_y;         // activated decoder output 2 [F][8][513]
  const float* d_xh;          // [F][513]
  const float* d_dec_a[3];
  const float* d_h;
  const float *d_z_mu, *d_z_lv;
  const float* d_enc_a[5];
  const float* lnp;           // [F][3][LNP_C]
  const float* pk;            // packed copies (Pk: the transposed Wy)
  float* G;                   // flat gradient buffer (zero-filled before the launch)
};

// job list: segment s owns blocks [start[s], start[s + 1]); kind / layer select the code, `tiles` = blocks per frame
// chunk, `fc` = frame chunks
enum { WJ_CONV = 0, WJ_TOEP, WJ_OUTER, WJ_MBIAS, WJ_HBIAS, WJ_B3, WJ_LNP, WJ_EMB };
constexpr int WG_MAXSEG = 24;
constexpr int EMB_SL = 8, EMB_W = 193;      // the embedding job: slices of the 1539 merge columns per speaker
struct WgPlan {
  int nseg;
  int start[WG_MAXSEG + 1];
  int kind[WG_MAXSEG];
  int layer[WG_MAXSEG];
  int tiles[WG_MAXSEG];
  int fc[WG_MAXSEG];
};

#ifdef FRAME_EMU
FR_DEV void fr_atomic_add(float* p, float v) { *p += v; }
#else
FR_DEV void fr_atomic_add(float* p, float v) { atomicAdd(p, v); }
#endif

// ------------------------------------------------------------------------------------------------ conv layers
// dW[(t*CA + a)*CB + b] += sum_{f in chunk} sum_j U[f][a][S j - PAD + t] * V[f][b][j]
//   conv layer (TF kernel [t][c][o]):            a = input channel c,  U = the layer's activated input,  V = d(pre-LN output)
//   transposed conv (TF kernel [t][o][c]):       a = output channel o, U = d(pre-LN output),             V = the layer's input
// A thread owns all K taps x one a x four b x one slice of the positions j; a block stages, frame by frame, the U rows of
// its AT channels (with the SAME-padding halo as zeros) and the whole V tensor in LDS.
// Staging copy with UB loads in flight per thread (the plain strided loop compiles to load / wait / LDS store per element pair:
// one L2 round trip each, which was most of every job's time)
template <int UB, class L>
FR_DEV void wg_stage(int tid, int n, float* dst, L&& ld) {
  for (int b = 0; b < n; b += UB * WT) {
    float v[UB];
    FR_UNROLL
    for (int u = 0; u < UB; ++u) {
      const int i = b + u * WT + tid;
      v[u] = i < n ? ld(i) : 0.f;
    }
    FR_UNROLL
    for (int u = 0; u < UB; ++u) {
      const int i = b + u * WT + tid;
      if (i < n) dst[i] = v[u];
    }
  }
}

// FB frames are staged per trip (a trip costs a global round trip and two barriers whatever it carries); the JS position
// slices of a tile are summed through LDS before the atomics (same-address atomics serialise at ~100 ns each: the first
// version spent 109 us in the 112-element gradient of encoder layer 0).
template <int CA_, int HU_, int CB_, int HV_, int K_, int S_, int PAD_, int AT_, int JS_, int FB_>
struct ConvW {
  static constexpr int CA = CA_, HU = HU_, CB = CB_, HV = HV_, K = K_, S = S_, PAD = PAD_, AT = AT_, JS = JS_, FB = FB_;
  static constexpr int BQ = cdiv_(CB, 4), NTILE = BQ * AT, NTH = NTILE * JS, AB = cdiv_(CA, AT);   // threads used, a-blocks
  static constexpr int HUP = S * (HV - 1) + K;       // staged row: index 0 = position -PAD
  static constexpr int JL = cdiv_(HV, JS);
  static constexpr int FRAME = AT * HUP + CB * HV;   // staged floats per frame
  static constexpr bool VIA_LDS = NTH * K * 4 <= WG_LDS;      // the tile leaves through LDS (coalesced atomics)
  static_assert(NTH <= WT && HUP >= PAD + HU && FB * FRAME <= WG_LDS && (JS == 1 || VIA_LDS), "ConvW tiling");
};
// (`uld(f, a, p)` = element p of channel a of frame f of the strided operand: a plain tensor, or one with a LayerNorm +
//  lrelu applied on load -- the critic's job list, disc_frame.h)
template <class T, class R, class UL>
FR_DEV void convw_job_u(R& run, float* lds, UL&& uld, const float* V, float* dW, int F, int ablk, int fchunk, int nfc) {
  const int a0 = ablk * T::AT;
  const int fper = (F + nfc - 1) / nfc, f0 = fchunk * fper, f1 = imin_(F, f0 + fper);
  float acc[T::K][4];
  run.frames(f0, f1, T::FB,
             // ---- stage: per frame of the trip, the U rows of this block's channels (zero halo) and V
             [&](int tid, int f, int nfb) {
               constexpr int UB = cdiv_(cdiv_(T::FB * T::FRAME, WT), cdiv_(cdiv_(T::FB * T::FRAME, WT), 14));   // <= 14 in flight
               wg_stage<UB>(tid, nfb * T::FRAME, lds, [&](int i) {
                 const int fb = i / T::FRAME, r = i % T::FRAME;
                 if (r >= T::AT * T::HUP) return V[(size_t)(f + fb) * T::CB * T::HV + (r - T::AT * T::HUP)];
                 const int al = r / T::HUP, p = r % T::HUP - T::PAD, a = a0 + al;
                 return (a < T::CA && p >= 0 && p < T::HU) ? uld(f + fb, a, p) : 0.f;
               });
             },
             // ---- zero the accumulators
             [&](int tid, float (&ac)[T::K][4]) {
               FR_UNROLL
               for (int t = 0; t < T::K; ++t)
                 FR_UNROLL
                 for (int q = 0; q < 4; ++q) ac[t][q] = 0.f;
             },
             // ---- accumulate the frames of the trip
             [&](int tid, float (&ac)[T::K][4], int nfb) {
               if (tid >= T::NTH) return;
               const int bq = tid % T::BQ, al = (tid / T::BQ) % T::AT, js = tid / T::NTILE;
               const int j0 = js * T::JL, j1 = imin_(T::HV, j0 + T::JL);
               for (int fb = 0; fb < nfb; ++fb) {
                 const float* ur = lds + fb * T::FRAME + al * T::HUP;
                 const float* vs = lds + fb * T::FRAME + T::AT * T::HUP;
                 for (int j = j0; j < j1; ++j) {
                   float v[4];
                   FR_UNROLL
                   for (int q = 0; q < 4; ++q) v[q] = (4 * bq + q < T::CB) ? vs[(4 * bq + q) * T::HV + j] : 0.f;
                   FR_UNROLL
                   for (int t = 0; t < T::K; ++t) {
                     const float u = ur[T::S * j + t];
                     FR_UNROLL
                     for (int q = 0; q < 4; ++q) ac[t][q] += u * v[q];
                   }
                 }
               }
             },
             // ---- after the last trip: the tile (all position slices) goes to LDS, or straight out where LDS is too small
             [&](int tid, float (&ac)[T::K][4]) {
               if (tid >= T::NTH) return;
               if constexpr (T::VIA_LDS) {
                 FR_UNROLL
                 for (int t = 0; t < T::K; ++t)
                   FR_UNROLL
                   for (int q = 0; q < 4; ++q) lds[tid * (T::K * 4) + t * 4 + q] = ac[t][q];
               } else {
                 const int bq = tid % T::BQ, al = tid / T::BQ, a = a0 + al;
                 if (a >= T::CA) return;
                 FR_UNROLL
                 for (int t = 0; t < T::K; ++t)
                   FR_UNROLL
                   for (int q = 0; q < 4; ++q)
                     if (4 * bq + q < T::CB) fr_atomic_add(dW + (size_t)(t * T::CA + a) * T::CB + 4 * bq + q, ac[t][q]);
               }
             },
             acc);
  if constexpr (T::VIA_LDS) {
    run.phase([&](int tid) {
      // (tile, tap, q) triples dealt to the threads: one atomic per element and block
      // (consecutive lanes: q, then bq, then al = consecutive addresses of dW[t][a][b])
      for (int e = tid; e < T::NTILE * T::K * 4; e += WT) {
        const int t = e / (T::NTILE * 4), tile = (e / 4) % T::NTILE, q = e % 4, r = t * 4 + q;
        const int bq = tile % T::BQ, al = tile / T::BQ, a = a0 + al;
        if (a >= T::CA || 4 * bq + q >= T::CB) continue;
        float sm = 0.f;
        for (int js = 0; js < T::JS; ++js) sm += lds[(js * T::NTILE + tile) * (T::K * 4) + r];
        fr_atomic_add(dW + (size_t)(t * T::CA + a) * T::CB + 4 * bq + q, sm);
      }
    });
  }
}

template <class T, class R>
FR_DEV void convw_job(R& run, float* lds, const float* U, const float* V, float* dW, int F, int ablk, int fchunk, int nfc) {
  convw_job_u<T>(run, lds, [&](int f, int a, int p) { return U[((size_t)f * T::CA + a) * T::HU + p]; }, V, dW, F, ablk, fchunk, nfc);
}

//                 CA   HU   CB   HV   K  S PAD AT  JS  FB
using WE0 = ConvW<1, 513, 16, 171, 7, 3, 2, 1, 57, 2>;
using WE1 = ConvW<16, 171, 32, 57, 7, 3, 2, 8, 4, 2>;
using WE2 = ConvW<32, 57, 64, 19, 7, 3, 2, 8, 2, 4>;
using WE3 = ConvW<64, 19, 128, 7, 7, 3, 3, 8, 1, 4>;
using WE4 = ConvW<128, 7, 256, 3, 7, 3, 3, 4, 1, 8>;
using WD0 = ConvW<32, 57, 81, 19, 9, 3, 3, 12, 1, 3>;
using WD1 = ConvW<16, 171, 32, 57, 7, 3, 2, 8, 4, 2>;
using WD2 = ConvW<8, 513, 16, 171, 7, 3, 2, 8, 8, 1>;

// ------------------------------------------------------------------------------------------------ the 1025-tap layer
// dW[t][c] += sum_f sum_j y[f][c][j] * g[f][j + t - 512]; a thread owns nine consecutive taps of one channel and one half
// of j; the nine values of g it needs slide by one per step (rotating register window, as in the forward direction)
// A workgroup owns 16 tap groups (144 taps) x 8 channels x 2 position slices; the slices meet in LDS and the tile leaves
// as 1152 atomics on consecutive addresses (device-scope atomics are executed at the memory side, one transaction per
// touched line: four position slices with per-lane strided atomics took 45 us at 16 frames, two took 31)
constexpr int TWG_TG = 114, TWG_JS = 2, TWG_TGB = 16, TWG_BLOCKS = cdiv_(TWG_TG, TWG_TGB);
static_assert(TWG_TGB * TP_C * TWG_JS == WT, "Toeplitz weight-gradient tile");
constexpr int TWG_GP = 512 + 513 + 512 + 16;      // zero-padded g
static_assert(TP_C * TP_H + TWG_GP <= WG_LDS, "Toeplitz weight-gradient staging");
template <class R>
FR_DEV void toepw_job(R& run, float* lds, const float* Y, const float* Gx, float* dW, int F, int tblk, int fchunk, int nfc) {
  constexpr int SL = cdiv_(TP_H, TWG_JS), TB = TWG_TGB * TP_R;      // taps per workgroup
  float* ys = lds;                  // [8][513]
  float* gp = lds + TP_C * TP_H;    // gp[i] = g[i - 512]
  const int fper = (F + nfc - 1) / nfc, f0 = fchunk * fper, f1 = imin_(F, f0 + fper);
  float acc[TP_R][1];
  run.frames(f0, f1, 1,
             [&](int tid, int f, int) {
               wg_stage<12>(tid, TP_C * TP_H + TWG_GP, lds, [&](int i) {
                 if (i < TP_C * TP_H) return Y[(size_t)f * TP_C * TP_H + i];
                 const int g = i - TP_C * TP_H - 512;
                 return (g >= 0 && g < TP_H) ? Gx[(size_t)f * TP_H + g] : 0.f;
               });
             },
             [&](int tid, float (&ac)[TP_R][1]) {
               FR_UNROLL
               for (int i = 0; i < TP_R; ++i) ac[i][0] = 0.f;
             },
             [&](int tid, float (&ac)[TP_R][1], int) {
               const int tg = tblk * TWG_TGB + tid % TWG_TGB, c = (tid / TWG_TGB) % TP_C, js = tid / (TWG_TGB * TP_C);
               if (tg >= TWG_TG) return;
               const int t0 = tg * TP_R;
               const int jb = js * SL, je = imin_(TP_H, jb + SL);
               const float* yc = ys + c * TP_H;
               const float* g0 = gp + t0;                 // value of tap i at step j: g0[j + i]
               float win[TP_R];
               FR_UNROLL
               for (int i = 0; i < TP_R - 1; ++i) win[i] = g0[jb + i];
               int j = jb;
               for (; j + TP_R <= je; j += TP_R) {
                 FR_UNROLL
                 for (int s = 0; s < TP_R; ++s) {
                   win[(TP_R - 1 + s) % TP_R] = g0[j + s + TP_R - 1];
                   const float v = yc[j + s];
                   FR_UNROLL
                   for (int i = 0; i < TP_R; ++i) ac[i][0] += win[(i + s) % TP_R] * v;
                 }
               }
               FR_UNROLL
               for (int s = 0; s < TP_R; ++s) {
                 if (j + s < je) {
                   win[(TP_R - 1 + s) % TP_R] = g0[j + s + TP_R - 1];
                   const float v = yc[j + s];
                   FR_UNROLL
                   for (int i = 0; i < TP_R; ++i) ac[i][0] += win[(i + s) % TP_R] * v;
                 }
               }
             },
             [&](int tid, float (&ac)[TP_R][1]) {
               const int tgl = tid % TWG_TGB, r = tid / TWG_TGB;      // r = js * 8 + c
               FR_UNROLL
               for (int i = 0; i < TP_R; ++i) lds[r * TB + tgl * TP_R + i] = ac[i][0];
             },
             acc);
  run.phase([&](int tid) {
    for (int e = tid; e < TB * TP_C; e += WT) {
      const int tl = e / TP_C, c = e % TP_C, t = tblk * TB + tl;
      if (t >= TP_K) continue;
      float sm = 0.f;
      FR_UNROLL
      for (int js = 0; js < TWG_JS; ++js) sm += lds[(js * TP_C + c) * TB + tl];
      fr_atomic_add(dW + t * TP_C + c, sm);
    }
  });
}

// ------------------------------------------------------------------------------------------------ outer products
// dW[(4 rq + q) * ldg + c] = sum_f A(f)[4 rq + q] * B[f * ldb + c]; A(f) = row f of `A` (lda floats apart), or row idx[f]
// of it (the embedding FC: A = the speaker table).  One thread per (row quad, column), lanes along the columns.
FR_DEV void outer_job(int gtid, const float* A, int lda, const int* idx /*LDS: clamped row per frame, or null*/, const float* B,
                      int ldb, int ncol, int nrq, float* dW, int ldg, int F) {
  if (gtid >= ncol * nrq) return;
  const int c = gtid % ncol, rq = gtid / ncol;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  // four frames per trip, every load of the trip issued before the first product (a frame per trip is a chain of L2
  // round trips: 256 frames took 60 us)
  for (int f0 = 0; f0 < F; f0 += 4) {
    float av[4][4], bv[4];
    FR_UNROLL
    for (int u = 0; u < 4; ++u) {
      const int f = imin_(f0 + u, F - 1);
      const int row = idx ? idx[f] : f;
      const float* ar = A + (size_t)row * lda + 4 * rq;
      FR_UNROLL
      for (int q = 0; q < 4; ++q) av[u][q] = ar[q];
      bv[u] = f0 + u < F ? B[(size_t)f * ldb + c] : 0.f;
    }
    FR_UNROLL
    for (int u = 0; u < 4; ++u) {
      a0 += av[u][0] * bv[u];
      a1 += av[u][1] * bv[u];
      a2 += av[u][2] * bv[u];
      a3 += av[u][3] * bv[u];
    }
  }
  float* o = dW + (size_t)(4 * rq) * ldg + c;
  o[0] = a0;
  o[ldg] = a1;
  o[2 * (size_t)ldg] = a2;
  o[3 * (size_t)ldg] = a3;
}

// sum over frames of p[f * stride] with eight loads in flight
FR_DEV float frame_sum(const float* p, size_t stride, int F) {
  float s = 0.f;
  for (int f0 = 0; f0 < F; f0 += 8) {
    float v[8];
    FR_UNROLL
    for (int u = 0; u < 8; ++u) v[u] = f0 + u < F ? p[(size_t)(f0 + u) * stride] : 0.f;
    FR_UNROLL
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  return s;
}

// ------------------------------------------------------------------------------------------------ the launch body
template <class R>
FR_DEV void frame_wgrad_block(R& run, float* lds, const WgArgs& a, const WgPlan& pl, int block) {
  int s = 0;
  while (s + 1 < pl.nseg && block >= pl.start[s + 1]) ++s;
  const int local = block - pl.start[s], kind = pl.kind[s], layer = pl.layer[s], tiles = pl.tiles[s], nfc = pl.fc[s];
  const int tile = local % tiles, fchunk = local / tiles;
  const POff& o = a.off;
  const int F = a.F;
  float* G = a.G;
  if (kind == WJ_CONV) {
    switch (layer) {
      case 0: convw_job<WE0>(run, lds, a.x, a.d_enc_a[0], G + o.ew[0], F, tile, fchunk, nfc); break;
      case 1: convw_job<WE1>(run, lds, a.y_enc[0], a.d_enc_a[1], G + o.ew[1], F, tile, fchunk, nfc); break;
      case 2: convw_job<WE2>(run, lds, a.y_enc[1], a.d_enc_a[2], G + o.ew[2], F, tile, fchunk, nfc); break;
      case 3: convw_job<WE3>(run, lds, a.y_enc[2], a.d_enc_a[3], G + o.ew[3], F, tile, fchunk, nfc); break;
      case 4: convw_job<WE4>(run, lds, a.y_enc[3], a.d_enc_a[4], G + o.ew[4], F, tile, fchunk, nfc); break;
      case 5: convw_job<WD0>(run, lds, a.d_dec_a[0], a.h, G + o.dw[0], F, tile, fchunk, nfc); break;
      case 6: convw_job<WD1>(run, lds, a.d_dec_a[1], a.y_dec[0], G + o.dw[1], F, tile, fchunk, nfc); break;
      default: convw_job<WD2>(run, lds, a.d_dec_a[2], a.y_dec[1], G + o.dw[2], F, tile, fchunk, nfc); break;
    }
  } else if (kind == WJ_TOEP) {
    toepw_job(run, lds, a.dec_y, a.d_xh, G + o.dw[3], F, tile, fchunk, nfc);
  } else if (kind == WJ_OUTER) {
    int* ids = reinterpret_cast<int*>(lds);       // speaker row of every frame (the embedding FC gathers rows of the table)
    if (layer == 1)
      run.phase([&](int tid) {
        for (int f = tid; f < F; f += WT) {
          int64_t yy = a.y[f];
          ids[f] = (int)(yy < 0 ? 0 : (yy >= a.ny ? a.ny - 1 : yy));
        }
      });
    run.phase([&](int tid) {
      const int g = local * WT + tid;
      if (layer == 0) outer_job(g, a.z, 128, nullptr, a.d_h, MERGE_N, MERGE_N, 32, G + o.wz, MERGE_N, F);                    // dWz
      else if (layer == 1) outer_job(g, a.P + o.emb, 128, ids, a.d_h, MERGE_N, MERGE_N, 32, G + o.wy, MERGE_N, F);        // dWy
      else if (layer == 2) outer_job(g, a.y_enc[4], 768, nullptr, a.d_z_mu, 128, 128, 192, G + o.wmu, 128, F);             // dWmu
      else outer_job(g, a.y_enc[4], 768, nullptr, a.d_z_lv, 128, 128, 192, G + o.wlv, 128, F);                                // dWlv
    });
  } else if (kind == WJ_MBIAS) {      // the three merge biases receive the same gradient (model/vae.py:51-61)
    run.phase([&](int tid) {
      const int n = local * WT + tid;
      if (n >= MERGE_N) return;
      const float sm = frame_sum(a.d_h + n, MERGE_N, F);
      G[o.bz + n] = sm;
      G[o.by + n] = sm;
      G[o.bm + n] = sm;
    });
  } else if (kind == WJ_HBIAS) {
    run.phase([&](int tid) {
      const float* d = tid < 128 ? a.d_z_mu : a.d_z_lv;
      const int k = tid & 127;
      G[(tid < 128 ? o.bmu : o.blv) + k] = frame_sum(d + k, 128, F);
    });
  } else if (kind == WJ_B3) {          // bias of the last decoder layer: sum of d(xh); a slice per block, one atomic each
    run.phase([&](int tid) {
      const int n = F * TP_H, per = (n + tiles - 1) / tiles, i0 = local * per, i1 = imin_(n, i0 + per);
      float sm = 0.f;
      for (int ib = i0 + tid; ib < i1; ib += 8 * WT) {
        float v[8];
        FR_UNROLL
        for (int u = 0; u < 8; ++u) v[u] = ib + u * WT < i1 ? a.d_xh[ib + u * WT] : 0.f;
        FR_UNROLL
        for (int u = 0; u < 8; ++u) sm += v[u];
      }
      lds[tid] = sm;
    });
    run.phase([&](int tid) {
      if (tid == 0) {
        float sm = 0.f;
        for (int i = 0; i < WT; ++i) sm += lds[i];
        fr_atomic_add(G + o.db[3], sm);
      }
    });
  } else if (kind == WJ_LNP) {         // LayerNorm offsets / scales and conv biases: sums over frames of the per-frame channel sums
    run.phase([&](int tid) {
      const int i = local * WT + tid;
      if (i >= 3 * LNP_C) return;
      const int k = i / LNP_C, cs = i % LNP_C;
      const float sm = frame_sum(a.lnp + (size_t)k * LNP_C + cs, 3 * LNP_C, F);
      int b, g, be, c;
      if (cs < LNP_DEC1) { c = cs - LNP_DEC2; b = o.db[2]; g = o.dgamma[2]; be = o.dbeta[2]; }
      else if (cs < LNP_DEC0) { c = cs - LNP_DEC1; b = o.db[1]; g = o.dgamma[1]; be = o.dbeta[1]; }
      else if (cs < LNP_ENC4) { c = cs - LNP_DEC0; b = o.db[0]; g = o.dgamma[0]; be = o.dbeta[0]; }
      else if (cs < LNP_ENC3) { c = cs - LNP_ENC4; b = o.eb[4]; g = o.egamma[4]; be = o.ebeta[4]; }
      else if (cs < LNP_ENC2) { c = cs - LNP_ENC3; b = o.eb[3]; g = o.egamma[3]; be = o.ebeta[3]; }
      else if (cs < LNP_ENC1) { c = cs - LNP_ENC2; b = o.eb[2]; g = o.egamma[2]; be = o.ebeta[2]; }
      else if (cs < LNP_ENC0) { c = cs - LNP_ENC1; b = o.eb[1]; g = o.egamma[1]; be = o.ebeta[1]; }
      else { c = cs - LNP_ENC0; b = o.eb[0]; g = o.egamma[0]; be = o.ebeta[0]; }
      G[(k == 0 ? be : k == 1 ? g : b) + c] = sm;      // k: 0 = d(offset), 1 = d(scale), 2 = d(conv bias)
    });
  } else {                             // WJ_EMB: dE[k][i] = sum_n S_k[n] Wy[i][n], S_k = sum of d(h) over the frames of speaker k
    // a block = one speaker x one eighth of n (193 columns); partial rows are added with atomics
    const int k = local / EMB_SL, n0 = (local % EMB_SL) * EMB_W, n1 = imin_(MERGE_N, n0 + EMB_W);
    int* ids = reinterpret_cast<int*>(lds + 1024);
    run.phase([&](int tid) {
      for (int f = tid; f < F; f += WT) {
        int64_t yy = a.y[f];
        ids[f] = (int)(yy < 0 ? 0 : (yy >= a.ny ? a.ny - 1 : yy));
      }
    });
    run.phase([&](int tid) {
      const int n = n0 + tid;
      if (tid < EMB_W && n < n1) {
        float sm = 0.f;
        for (int f0 = 0; f0 < F; f0 += 8) {      // eight loads in flight (the speaker test is on LDS data)
          float v[8];
          FR_UNROLL
          for (int u = 0; u < 8; ++u) v[u] = (f0 + u < F && ids[f0 + u] == k) ? a.d_h[(size_t)(f0 + u) * MERGE_N + n] : 0.f;
          FR_UNROLL
          for (int u = 0; u < 8; ++u) sm += v[u];
        }
        lds[tid] = sm;
      }
    });
    // each thread: one column i of the transposed Wy (lanes along i: coalesced rows of 128 floats) x one half of the slice
    run.phase([&](int tid) {
      const int i = tid & 127, half = tid >> 7;
      const float* wt = a.pk + Pk::wyT + i;
      constexpr int HW = (EMB_W + 1) / 2;
      const int m0 = n0 + half * HW, m1 = imin_(n1, m0 + HW);
      float sm = 0.f;
      for (int nb = m0; nb < m1; nb += 8) {
        float w[8];
        FR_UNROLL
        for (int u = 0; u < 8; ++u) w[u] = nb + u < m1 ? wt[(size_t)(nb + u) * MERGE_K] : 0.f;
        FR_UNROLL
        for (int u = 0; u < 8; ++u) sm += (nb + u < m1 ? lds[nb + u - n0] : 0.f) * w[u];
      }
      fr_atomic_add(G + o.emb + k * 128 + i, sm);
    });
  }
}

// ------------------------------------------------------------------------------------------------ job list (host)
// frame chunks per job: enough blocks to spread a small batch over the chip, few enough that the fp32 atomics of the
// big tensors stay cheap (a chunk adds one atomic per element)
// caps: most frame chunks of the nine chunked jobs, in the order they are added below
constexpr int WG_CAPS_DEFAULT[9] = {64, 16, 64, 16, 16, 32, 32, 32, 32};
inline WgPlan make_wgplan(int F, int ny, const int* caps = WG_CAPS_DEFAULT) {
  WgPlan p;
  int n = 0, blk = 0;
  auto add = [&](int kind, int layer, int tiles, int fc) {
    p.start[n] = blk;
    p.kind[n] = kind;
    p.layer[n] = layer;
    p.tiles[n] = tiles;
    p.fc[n] = fc;
    blk += tiles * fc;
    ++n;
  };
  // frames per block: at least one full staging trip (FB frames), more once the chunk count would pass `cap` (a chunk adds
  // one atomic per element: same-address atomics serialise)
  auto fcs = [&](int fb, int cap) {
    int fpb = (F + cap - 1) / cap;
    if (fpb < fb) fpb = fb;
    return (F + fpb - 1) / fpb;
  };
  // the long jobs first (they decide when the launch ends)
  add(WJ_TOEP, 0, TWG_BLOCKS, fcs(1, caps[0]));
  add(WJ_CONV, 4, WE4::AB, fcs(WE4::FB, caps[1]));
  add(WJ_CONV, 5, WD0::AB, fcs(WD0::FB, caps[2]));
  add(WJ_CONV, 3, WE3::AB, fcs(WE3::FB, caps[3]));
  add(WJ_CONV, 2, WE2::AB, fcs(WE2::FB, caps[4]));
  add(WJ_CONV, 1, WE1::AB, fcs(WE1::FB, caps[5]));
  add(WJ_CONV, 6, WD1::AB, fcs(WD1::FB, caps[6]));
  add(WJ_CONV, 7, WD2::AB, fcs(WD2::FB, caps[7]));
  add(WJ_CONV, 0, WE0::AB, fcs(WE0::FB, caps[8]));
  add(WJ_OUTER, 0, cdiv_(32 * MERGE_N, WT), 1);
  add(WJ_OUTER, 1, cdiv_(32 * MERGE_N, WT), 1);
  add(WJ_OUTER, 2, cdiv_(192 * 128, WT), 1);
  add(WJ_OUTER, 3, cdiv_(192 * 128, WT), 1);
  add(WJ_EMB, 0, ny * EMB_SL, 1);
  add(WJ_MBIAS, 0, cdiv_(MERGE_N, WT), 1);
  add(WJ_HBIAS, 0, 1, 1);
  add(WJ_B3, 0, 8, 1);
  add(WJ_LNP, 0, cdiv_(3 * LNP_C, WT), 1);
  p.nseg = n;
  p.start[n] = blk;
  return p;
}

}  // namespace frame
}  // namespace vaenpvc
