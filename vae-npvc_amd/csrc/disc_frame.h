// disc_frame.h -- the critic's two thin conv layers (1 -> 16 -> 32 channels, 7 taps, stride 3: the ConvVAE encoder's first
// two layers' geometry, architecture-vawgan-vcc2016.json:7-14) on the scheme of the small-batch ConvVAE path
// (gfx950_frame.h: one workgroup carries one row through a whole pass segment; gfx950_frame_wgrad.h: every parameter
// gradient of these layers in one job-list launch).  At the branch's 16-frame batches a critic step is its launch count
// (DESIGN.md section 9); the kernels below replace, per step,
//   k_critic_front_wgrad : 4 x conv weight gradient (passes 3 and 4), and the layers' entries of the LayerNorm-parameter,
//                          channel-sum and partial-sum launches.
// The 115-tap layer stays a dense layer on the matrix cores (disc.hip): as a per-row GEMV it would stream its 8.9 MB of
// expanded weights once per row.
// Reference: trainer/vae.py:117-145 (WGAN-GP critic update), util/layers.py:47-66 (the conv block); DESIGN.md section 9 for
// the pass structure (1 forward, 2 input gradient at xi, 3 adjoint of pass 2, 4 ordinary backward).
#pragma once
#include "gfx950_frame_dev.h"

namespace vaenpvc {
namespace disc {
namespace front {
using namespace frame;
using tuned::WRunner;

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

}  // namespace front
}  // namespace disc
}  // namespace vaenpvc
