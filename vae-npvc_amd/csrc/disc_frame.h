// disc_frame.h -- the critic's two thin conv layers (1 -> 16 -> 32 channels, 7 taps, stride 3: the ConvVAE encoder's first
// two layers' geometry, architecture-vawgan-vcc2016.json:7-14) on the scheme of the small-batch ConvVAE path
// (gfx950_frame.h: one workgroup carries one row through a whole pass segment; gfx950_frame_wgrad.h: every parameter
// gradient of these layers in one job-list launch).  At the branch's 16-frame batches a critic step is its launch count
// (DESIGN.md section 9); the kernels below replace, per step,
//   k_critic_front_wgrad : 4 x conv weight gradient (passes 3 and 4), and the layers' entries of the LayerNorm-parameter,
//                          channel-sum and partial-sum launches;
//   k_critic_front<FWD>  : pass 1 of both layers (2 x conv forward + statistics / activation pass);
//   k_critic_front<IGRAD>: pass 2 below the 115-tap layer (2 x LayerNorm backward, 2 x conv input gradient, the penalty);
//   k_critic_front<ADJ>  : pass 3 of both layers (2 x conv forward of the adjoint, 2 x adjoint of the LayerNorm backward);
//   k_critic_front<BWD>  : pass 4 below the 115-tap layer (2 x LayerNorm backward, conv input gradient).
// The 115-tap layer stays a dense layer on the matrix cores (disc.hip): as a per-row GEMV it would stream its 8.9 MB of
// expanded weights once per row.
// Reference: trainer/vae.py:117-145 (WGAN-GP critic update), util/layers.py:47-66 (the conv block); DESIGN.md section 9 for
// the pass structure (1 forward, 2 input gradient at xi, 3 adjoint of pass 2, 4 ordinary backward).
#pragma once
#ifdef FRAME_EMU      // host emulation (tests/frame_emu/frame_emu.cpp): runners and headers come from there
#include "gfx950_frame.h"
#include "gfx950_frame_wgrad.h"
#else
#include "gfx950_frame_dev.h"
#endif

namespace vaenpvc {
namespace disc {
namespace front {
using namespace frame;

constexpr int C0 = 16, H0 = 171, N0 = C0 * H0, C1 = 32, H1 = 57, N1 = C1 * H1, HIN = 513;

// ------------------------------------------------------------------------------------------------ weight gradients
struct CwArgs {
  const float* rows;    // [B][513]       input of layer 0 (x | xh | xi)
  const float* gt;      // [F][513]       adjoint input of pass 3
  const float* u0;      // [B][16][171]   pre-LN output of layer 0
  const float* st0;     // [B][2]
  const float* u1;      // [B][32][57]
  const float* st1;
  const float* gamma0;  // LayerNorm parameters (pointers into the flat parameter buffer)
  const float* beta0;
  const float* gamma1;
  const float* beta1;
  const float* at0;     // [F][16][171]   adjoint of abar0 (input of layer 1 in pass 3)
  const float* ubar0;   // [F][16][171]
  const float* ubar1;   // [F][32][57]
  const float* du0;     // [B][16][171]   pass 4: gradient at the pre-LN outputs
  const float* du1;     // [B][32][57]
  const float* da0;     // [B][16][171]   pass 4: gradient at the activated outputs
  const float* da1;
  const float* pn0;     // [F][16][171]   pass 3: per-element adjoint of gamma
  const float* pn1;
  float *dW0, *dW1, *db0, *db1, *dg0, *dbt0, *dg1, *dbt1;   // destinations in the (zeroed) critic gradient buffer
  int F, B;
};
constexpr int CW_SEGS = 6;
struct CwPlan {
  int start[CW_SEGS + 1];
  int tiles[CW_SEGS];
  int fc[CW_SEGS];
};
// per-channel sums of one layer over a chunk of rows: conv bias, LayerNorm scale (pass 4 + the pn term of pass 3), offset
template <int C, int H, class R>
FR_DEV void chan_job(R& run, float* lds, const float* du, const float* da, const float* u, const float* st, const float* gamma,
                     const float* beta, const float* pn, float* db, float* dg, float* dbt, int B, int F, int c, int chunk, int nch) {
  const int per = (B + nch - 1) / nch, r0 = chunk * per, r1 = imin_(B, r0 + per);
  const float g = gamma[c], b = beta[c];
  run.phase([&](int tid) {
    float sd = 0.f, sg = 0.f, sb = 0.f;
    const int n = (r1 - r0) * H;
    for (int i0 = tid; i0 < n; i0 += 4 * WT) {
      float vd[4], va[4], vu[4], vm[4], vr[4], vp[4];
      FR_UNROLL
      for (int k = 0; k < 4; ++k) {
        const int i = i0 + k * WT;
        const bool ok = i < n;
        const int f = r0 + (ok ? i / H : 0), h = ok ? i % H : 0;
        const size_t e = ((size_t)f * C + c) * H + h;
        vd[k] = ok ? du[e] : 0.f;
        va[k] = ok ? da[e] : 0.f;
        vu[k] = ok ? u[e] : 0.f;
        vm[k] = st[2 * f];
        vr[k] = st[2 * f + 1];
        vp[k] = (ok && f < F) ? pn[e] : 0.f;    // (pass 3 runs on the xi rows, stored as rows [0, F) of its tensors)
      }
      FR_UNROLL
      for (int k = 0; k < 4; ++k) {
        if (i0 + k * WT >= n) continue;
        const float xh = (vu[k] - vm[k]) * vr[k];
        const float nn = xh * g + b;
        const float dn = va[k] * (nn >= 0.f ? 1.0f : LEAK_F);
        sd += vd[k];
        sg += dn * xh + vp[k];
        sb += dn;
      }
    }
    lds[tid] = sd;
    lds[WT + tid] = sg;
    lds[2 * WT + tid] = sb;
  });
  run.phase([&](int tid) {
    if (tid < 3) {
      float s = 0.f;
      for (int i = 0; i < WT; ++i) s += lds[tid * WT + i];
      fr_atomic_add(tid == 0 ? db + c : (tid == 1 ? dg + c : dbt + c), s);
    }
  });
}

template <class R>
FR_DEV void critic_front_wgrad_block(R& run, float* lds, const CwArgs& a, const CwPlan& pl, int blk) {
  int s = 0;
  while (s + 1 < CW_SEGS && blk >= pl.start[s + 1]) ++s;
  const int local = blk - pl.start[s], tiles = pl.tiles[s], nfc = pl.fc[s];
  const int tile = local % tiles, chunk = local / tiles;
  if (s == 0) {          // dW0, pass 4: rows (x) du0 over all rows
    convw_job<WE0>(run, lds, a.rows, a.du0, a.dW0, a.B, tile, chunk, nfc);
  } else if (s == 1) {   // dW0, pass 3: gt (x) ubar0 over the xi rows
    convw_job<WE0>(run, lds, a.gt, a.ubar0, a.dW0, a.F, tile, chunk, nfc);
  } else if (s == 2) {   // dW1, pass 4: lrelu(LN(u0)) (x) du1 -- the activated tensor is rebuilt on load
    convw_job_u<WE1>(run, lds, [&](int f, int c, int p) {
      return lnact(a.u0[((size_t)f * C0 + c) * H0 + p], a.st0[2 * f], a.st0[2 * f + 1], a.gamma0[c], a.beta0[c]);
    }, a.du1, a.dW1, a.B, tile, chunk, nfc);
  } else if (s == 3) {   // dW1, pass 3: at0 (x) ubar1
    convw_job<WE1>(run, lds, a.at0, a.ubar1, a.dW1, a.F, tile, chunk, nfc);
  } else if (s == 4) {
    chan_job<C0, H0>(run, lds, a.du0, a.da0, a.u0, a.st0, a.gamma0, a.beta0, a.pn0, a.db0, a.dg0, a.dbt0, a.B, a.F, tile, chunk, nfc);
  } else {
    chan_job<C1, H1>(run, lds, a.du1, a.da1, a.u1, a.st1, a.gamma1, a.beta1, a.pn1, a.db1, a.dg1, a.dbt1, a.B, a.F, tile, chunk, nfc);
  }
}

inline CwPlan make_cwplan(int F, int B) {
  CwPlan p;
  int n = 0, blk = 0;
  auto add = [&](int tiles, int fc) {
    p.start[n] = blk;
    p.tiles[n] = tiles;
    p.fc[n] = fc;
    blk += tiles * fc;
    ++n;
  };
  auto fcs = [](int rows, int fb, int cap) {      // as make_wgplan: at least one staging trip per chunk, at most `cap` chunks
    int per = (rows + cap - 1) / cap;
    if (per < fb) per = fb;
    return (rows + per - 1) / per;
  };
  add(WE0::AB, fcs(B, WE0::FB, 32));
  add(WE0::AB, fcs(F, WE0::FB, 32));
  add(WE1::AB, fcs(B, WE1::FB, 32));
  add(WE1::AB, fcs(F, WE1::FB, 32));
  add(C0, fcs(B, 8, 8));
  add(C1, fcs(B, 8, 8));
  p.start[n] = blk;
  return p;
}

// ------------------------------------------------------------------------------------------------ per-row passes
// One 1024-thread workgroup carries one row through a pass segment; LDS map and tiles of gfx950_frame.h (E0F / E1F: the
// two forward convs, E1G: layer 1's input gradient).  Small per-channel data sits in LDS for the whole launch:
//   ch[0,48) conv biases | [48,96) LayerNorm scales | [96,144) offsets (layer 0: 16, layer 1: 32) | [144,256) layer 0's kernel
constexpr int CH_B = 0, CH_G = 48, CH_BT = 96, CH_W0 = 144;
enum { FP_FWD = 0, FP_IGRAD = 1, FP_ADJ = 2, FP_BWD = 3 };
struct FrontArgs {
  const float* P;         // flat critic parameters
  int w0, b0, g0, bt0, w1, b1, g1, bt1;   // offsets of the two layers' tensors
  const float* w1t;       // layer 1's kernel as [t][o][c] (written by k_rows)
  int rows0, nrows;       // the launch covers rows [rows0, rows0 + nrows) of the B-row tensors
  // B-row tensors
  const float* rows;      // [B][513]
  float* u0;              // [B][16][171] pre-LN outputs (written by FWD, read by the others)
  float* st0;
  float* u1;
  float* st1;
  float* ain2;            // [B][1824] activated output of layer 1 = input of the 115-tap layer's GEMM
  // pass 2 / 3 tensors, row r of the launch = row r of these
  float* abar1;           // IGRAD: in (from the 115-tap layer's input gradient); ADJ: in
  float* ubar1;           // IGRAD: out
  float* abar0;           // IGRAD: out; ADJ: in
  float* ubar0;           // IGRAD: out
  float* g;               // IGRAD: out [.][513]
  float* gt;              // IGRAD: out (coef != 0); ADJ: in
  float* gp_f;            // IGRAD: out
  float coef;             // 2 lambda / F
  int penalty;            // IGRAD: 1 = also leave gt and gp_f (critic step), 0 = only g (generator step)
  float *at0, *udir0, *pn0, *at1, *udir1, *pn1;     // ADJ: out
  // pass 4 tensors (B rows)
  float* da1;             // BWD: in (from the 115-tap layer's input gradient); out when it arrives as split-K parts
  float *du1, *da0, *du0; // BWD: out
  // IGRAD / BWD: the upstream of layer 1 as the `up_parts` split-K partial results of the 115-tap layer's input-gradient GEMM
  // ([part][GEMM row][1824], part stride up_stride floats): summed on load and stored to abar1 / da1 (0: already summed there)
  const float* up;
  long long up_stride;
  int up_parts;
  const float *add1, *add0;   // BWD: the injected adjoints udir1 / udir0 (row r - add_row0 of them), rows >= add_row0 only
  int add_row0;
};
static_assert(sizeof(FrontArgs) <= ARGS_FLOATS * 4, "argument block");

// LayerNorm + lrelu backward of the row in place: u in `bx`, upstream in `by` -> gradient at the pre-LN tensor in `by`
// (+ an injected term).  red[R_ST..] = {mean, rstd}.
template <class R, int C, int H, class AP>
FR_DEV void ln_bwd_row(R& run, const float* bx, float* by, float* red, const float* gamma, const float* beta, AP add) {
  constexpr int N = C * H;
  run.reduce2(red + R_S1, red + R_S2, [&](int tid, float& s1, float& s2) {
    const float mean = red[R_ST], rstd = red[R_ST + 1];
    for (int i = tid; i < N; i += NT) {
      const int c = i / H;
      float xh;
      const float dx = ln_dn(bx[i], by[i], mean, rstd, gamma[c], beta[c], xh) * gamma[c];
      s1 += dx;
      s2 += dx * xh;
    }
  });
  run.phase([&](int tid) {
    const float mean = red[R_ST], rstd = red[R_ST + 1];
    const float m1 = sum16(red + R_S1) * (1.0f / N), m2 = sum16(red + R_S2) * (1.0f / N);
    for (int i = tid; i < N; i += NT) {
      const int c = i / H;
      float xh;
      const float dx = ln_dn(bx[i], by[i], mean, rstd, gamma[c], beta[c], xh) * gamma[c];
      float r = rstd * (dx - m1 - xh * m2);
      if (add) r += add[i];
      by[i] = r;
    }
  });
}

// Adjoint of ln_bwd_row (disc.hip: k_ln_bwd_bwd has the algebra): q in `by`, the upstream of pass 2 (abar) in `dy`, u in
// `bx`.  Leaves at (adjoint of abar) in `by` and stores at / udir / pn.
template <class R, int C, int H, class GP>
FR_DEV void ln_bwd_bwd_row(R& run, const float* bx, float* by, const float* dy, float* red, const float* gamma, const float* beta,
                           GP at_g, GP udir_g, GP pn_g) {
  constexpr int N = C * H;
  constexpr float IN = 1.0f / N;
  auto elem = [&](int i, float mean, float rstd, float& xh, float& sl, float& p) {
    const int c = i / H;
    xh = (bx[i] - mean) * rstd;
    const float nn = xh * gamma[c] + beta[c];
    sl = nn >= 0.f ? 1.0f : LEAK_F;
    p = dy[i] * sl * gamma[c];
  };
  run.reduce2(red + R_S1, red + R_S2, [&](int tid, float& sp, float& spx) {
    const float mean = red[R_ST], rstd = red[R_ST + 1];
    for (int i = tid; i < N; i += NT) {
      float xh, sl, p;
      elem(i, mean, rstd, xh, sl, p);
      sp += p;
      spx += p * xh;
    }
  });
  run.reduce2(red + R_S3, red + R_S4, [&](int tid, float& sq, float& sqx) {
    const float mean = red[R_ST], rstd = red[R_ST + 1];
    for (int i = tid; i < N; i += NT) {
      const float xh = (bx[i] - mean) * rstd;
      sq += by[i];
      sqx += by[i] * xh;
    }
  });
  // (the four means are read from red[R_S1..R_S4]; the next pair of sums goes to red[80..112))
  run.reduce2(red + 80, red + 96, [&](int tid, float& sx, float& ss) {
    const float mean = red[R_ST], rstd = red[R_ST + 1];
    const float sp = sum16(red + R_S1) * IN, spx = sum16(red + R_S2) * IN, sqx = sum16(red + R_S4) * IN;
    for (int i = tid; i < N; i += NT) {
      float xh, sl, p;
      elem(i, mean, rstd, xh, sl, p);
      const float qq = by[i];
      const float ub = rstd * (p - sp - xh * spx);
      const float xt = -rstd * (spx * qq + sqx * p);
      sx += xt;
      ss += qq * ub + xt * xh;
    }
  });
  run.phase([&](int tid) {
    const float mean = red[R_ST], rstd = red[R_ST + 1];
    const float spx = sum16(red + R_S2) * IN, sq = sum16(red + R_S3) * IN, sqx = sum16(red + R_S4) * IN;
    const float sx = sum16(red + 80) * IN, ss = sum16(red + 96) * IN;
    for (int i = tid; i < N; i += NT) {
      const int c = i / H;
      float xh, sl, p;
      elem(i, mean, rstd, xh, sl, p);
      const float qq = by[i];
      const float xt = -rstd * (spx * qq + sqx * p);
      const float pt = rstd * (qq - sq - xh * sqx);
      const float atv = pt * gamma[c] * sl;
      by[i] = atv;
      at_g[i] = atv;
      udir_g[i] = rstd * (xt - sx - xh * ss);
      pn_g[i] = pt * dy[i] * sl;
    }
  });
}

template <int MODE, class R>
FR_DEV void critic_front_row(R& run, float* lds, const FrontArgs& a, int r) {
  float* bx = lds + L_BUFX;
  float* by = lds + L_BUFY;
  float* part = lds + L_PART;
  float* red = lds + L_RED;
  const float* ch = lds + L_CH;
  auto P = fr_g(a.P);
  const size_t row = (size_t)(a.rows0 + r);       // row of the B-row tensors
  const size_t rr = (size_t)r;                     // row of the per-range tensors
  if constexpr (MODE == FP_FWD) {
    auto xf = fr_g(a.rows) + row * HIN;
    run.phase([&](int tid) { halo_copy<1, HIN, E0F::HP, E0F::PAD>(tid, xf, bx); });
    run.phase([&](int tid) { sconv_part<E0F>(tid, bx, P + FR_UNIFORM(a.w0), part); });
    reduce_sum<R, E0F::KS, E0F::NOUT, E0F::HO>(run, part, ch + CH_B, by, red);
    var_sum<R, E0F::NOUT>(run, by, red, fr_g(a.u0) + row * N0);
    run.phase([&](int tid) {
      ln_apply<C0, H0, E1F::HP, E1F::PAD>(tid, by, red, ch + CH_G, ch + CH_BT, bx, fr_g(a.st0) + 2 * row, (FR_G(float))nullptr);
    });
    run.phase([&](int tid) { sconv_part<E1F>(tid, bx, P + FR_UNIFORM(a.w1), part); });
    reduce_sum<R, E1F::KS, E1F::NOUT, E1F::HO>(run, part, ch + CH_B + C0, by, red);
    var_sum<R, E1F::NOUT>(run, by, red, fr_g(a.u1) + row * N1);
    run.phase([&](int tid) {
      ln_apply<C1, H1, H1, 0>(tid, by, red, ch + CH_G + C0, ch + CH_BT + C0, bx, fr_g(a.st1) + 2 * row, fr_g(a.ain2) + row * N1);
    });
  } else if constexpr (MODE == FP_IGRAD || MODE == FP_BWD) {
    constexpr bool IG = MODE == FP_IGRAD;
    // ---- layer 1: LayerNorm backward of the upstream handed down by the 115-tap layer
    auto up1 = IG ? fr_g((const float*)a.abar1) + rr * N1 : fr_g((const float*)a.da1) + row * N1;
    auto u1r = fr_g((const float*)a.u1) + row * N1;
    auto st1r = fr_g((const float*)a.st1) + 2 * row;
    const bool inj = !IG && a.add1 && (int)row >= a.add_row0;   // uniform
    if (a.up_parts > 0) {   // uniform: sum the GEMM's split-K parts on load (z ascending, as k_mm_reduce), keep the sum for the later passes
      auto pp = fr_g(a.up) + (IG ? rr : row) * N1;
      auto keep = IG ? fr_g(a.abar1) + rr * N1 : fr_g(a.da1) + row * N1;
      const int np = FR_UNIFORM(a.up_parts);
      const size_t ps = (size_t)a.up_stride;
      run.phase([&](int tid) {
        for (int i = tid; i < N1; i += NT) {
          float t[8];      // (MM_MAX_SPLIT parts at most, all requested before the first add: z ascending as k_mm_reduce)
          FR_UNROLL
          for (int z = 0; z < 8; ++z) t[z] = z < np ? pp[z * ps + i] : 0.f;
          float v = 0.f;
          FR_UNROLL
          for (int z = 0; z < 8; ++z) v += t[z];
          by[i] = v;
          keep[i] = v;
          bx[i] = u1r[i];
        }
        if (tid == 0) {
          red[R_ST] = st1r[0];
          red[R_ST + 1] = st1r[1];
        }
      });
    } else
    run.phase([&](int tid) {
      for (int i = tid; i < N1; i += NT) {
        by[i] = up1[i];
        bx[i] = u1r[i];
      }
      if (tid == 0) {
        red[R_ST] = st1r[0];
        red[R_ST + 1] = st1r[1];
      }
    });
    {
      auto ad = inj ? fr_g(a.add1) + (row - a.add_row0) * N1 : (FR_G(const float))nullptr;
      ln_bwd_row<R, C1, H1>(run, bx, by, red, ch + CH_G + C0, ch + CH_BT + C0, ad);
    }
    run.phase([&](int tid) {
      flush(tid, by, IG ? fr_g(a.ubar1) + rr * N1 : fr_g(a.du1) + row * N1, N1);
      halo_copy<C1, H1, E1G::HP, E1G::HL>(tid, by, bx);
    });
    // ---- layer 1's input gradient, then layer 0's LayerNorm backward
    run.phase([&](int tid) { tconv_part<E1G>(tid, bx, fr_g(a.w1t), part); });
    run.phase([&](int tid) {
      reduce_load<E1G::KS, E1G::NOUT>(tid, part, by, fr_g((const float*)a.u0) + row * N0, bx, fr_g((const float*)a.st0) + 2 * row, red);
    });
    {
      auto ad = (inj && a.add0) ? fr_g(a.add0) + (row - a.add_row0) * N0 : (FR_G(const float))nullptr;
      auto mid = IG ? fr_g(a.abar0) + rr * N0 : fr_g(a.da0) + row * N0;      // gradient at layer 0's activated output
      run.phase([&](int tid) { flush(tid, by, mid, N0); });
      ln_bwd_row<R, C0, H0>(run, bx, by, red, ch + CH_G, ch + CH_BT, ad);
    }
    if constexpr (!IG) {
      run.phase([&](int tid) { flush(tid, by, fr_g(a.du0) + row * N0, N0); });
    } else {
      // ---- layer 0's input gradient: g[i] = sum_o sum_{t = (i + 2) mod 3, +3, +6} W0[t][o] ubar0[o][(i + 2 - t) / 3], then the penalty
      run.reduce(red + R_S1, [&](int tid) {
        flush(tid, by, fr_g(a.ubar0) + rr * N0, N0);
        float gi = 0.f;
        if (tid < HIN) {
          const int t0 = (tid + 2) % 3;
          for (int t = t0; t < 7; t += 3) {
            const int j = (tid + 2 - t) / 3;
            if (tid + 2 - t < 0 || j >= H0) continue;
            for (int o = 0; o < C0; ++o) gi += ch[CH_W0 + t * C0 + o] * by[o * H0 + j];
          }
          bx[tid] = gi;
          a.g[rr * HIN + tid] = gi;
        }
        return gi * gi;
      });
      run.phase([&](int tid) {
        if (!a.penalty) return;          // uniform: generator step
        const float nrm = sqrtf(sum16(red + R_S1));
        const float k = a.coef * (nrm - 1.0f) / nrm;
        if (tid < HIN) a.gt[rr * HIN + tid] = k * bx[tid];
        if (tid == 0) a.gp_f[rr] = (nrm - 1.0f) * (nrm - 1.0f);
      });
    }
  } else {   // FP_ADJ
    // ---- layer 0: q0 = conv0(gt) (no bias), adjoint of its LayerNorm backward
    auto gtr = fr_g((const float*)a.gt) + rr * HIN;
    run.phase([&](int tid) { halo_copy<1, HIN, E0F::HP, E0F::PAD>(tid, gtr, bx); });
    run.phase([&](int tid) { sconv_part<E0F>(tid, bx, P + FR_UNIFORM(a.w0), part); });
    float* dy = part + E1F::KS * E1F::NOUT;      // the pass-2 upstream of the layer, behind the largest partial-sum area used here
    static_assert(E1F::KS * E1F::NOUT + N0 <= PART && E0F::KS == 1, "LDS areas of the adjoint pass");
    {
      auto u0r = fr_g((const float*)a.u0) + row * N0;
      auto ab = fr_g((const float*)a.abar0) + rr * N0;
      auto st0r = fr_g((const float*)a.st0) + 2 * row;
      run.phase([&](int tid) {
        for (int i = tid; i < N0; i += NT) {
          by[i] = part[i];      // (one K slice: the partial sums are the result)
          bx[i] = u0r[i];
        }
        if (tid == 0) {
          red[R_ST] = st0r[0];
          red[R_ST + 1] = st0r[1];
        }
      });
      run.phase([&](int tid) {
        for (int i = tid; i < N0; i += NT) dy[i] = ab[i];
      });
    }
    ln_bwd_bwd_row<R, C0, H0>(run, bx, by, dy, red, ch + CH_G, ch + CH_BT, fr_g(a.at0) + rr * N0, fr_g(a.udir0) + rr * N0,
                              fr_g(a.pn0) + rr * N0);
    // ---- layer 1: q1 = conv1(at0), adjoint of its LayerNorm backward
    run.phase([&](int tid) { halo_copy<C0, H0, E1F::HP, E1F::PAD>(tid, by, bx); });
    run.phase([&](int tid) { sconv_part<E1F>(tid, bx, P + FR_UNIFORM(a.w1), part); });
    {
      auto u1r = fr_g((const float*)a.u1) + row * N1;
      auto ab = fr_g((const float*)a.abar1) + rr * N1;
      auto st1r = fr_g((const float*)a.st1) + 2 * row;
      run.phase([&](int tid) {
        for (int i = tid; i < N1; i += NT) {
          float q = 0.f;
          FR_UNROLL
          for (int k = 0; k < E1F::KS; ++k) q += part[k * E1F::NOUT + i];
          by[i] = q;
          bx[i] = u1r[i];
          dy[i] = ab[i];
        }
        if (tid == 0) {
          red[R_ST] = st1r[0];
          red[R_ST + 1] = st1r[1];
        }
      });
    }
    ln_bwd_bwd_row<R, C1, H1>(run, bx, by, dy, red, ch + CH_G + C0, ch + CH_BT + C0, fr_g(a.at1) + rr * N1, fr_g(a.udir1) + rr * N1,
                              fr_g(a.pn1) + rr * N1);
  }
}

// per-channel vectors and layer 0's kernel into LDS, once per workgroup
template <class R>
FR_DEV void critic_front_prologue(R& run, float* lds, const FrontArgs& a) {
  float* ch = lds + L_CH;
  auto P = fr_g(a.P);
  run.phase([&](int tid) {
    if (tid < C0) {
      ch[CH_B + tid] = P[a.b0 + tid];
      ch[CH_G + tid] = P[a.g0 + tid];
      ch[CH_BT + tid] = P[a.bt0 + tid];
    } else if (tid < C0 + C1) {
      ch[CH_B + tid] = P[a.b1 + tid - C0];
      ch[CH_G + tid] = P[a.g1 + tid - C0];
      ch[CH_BT + tid] = P[a.bt1 + tid - C0];
    } else if (tid >= 64 && tid < 64 + 7 * C0) {
      ch[CH_W0 + tid - 64] = P[a.w0 + tid - 64];
    }
  });
}

}  // namespace front
}  // namespace disc
}  // namespace vaenpvc
