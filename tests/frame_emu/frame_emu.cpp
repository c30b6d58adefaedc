// frame_emu.cpp -- HOST emulation of the small-batch frame kernels (vae-npvc_amd/csrc/gfx950_frame.h), test
// infrastructure only.  The header is written against a phase runner and a 16-byte load, so the very source the GPU
// executes is compiled here with g++ (FRAME_EMU): a phase = a loop over the 1024 thread ids, LDS = a heap array.  The
// CPU test-suite checks the emulated passes against the float64 oracle (tests/test_frame_emu.py): indexing, tilings,
// halos and the algebra of every stage are pinned before a GPU ever runs them.  Nothing in the product links this file.
#define FRAME_EMU 1
#include "../../vae-npvc_amd/csrc/gfx950_frame.h"
#include "../../vae-npvc_amd/csrc/gfx950_frame_wgrad.h"
#include "../../vae-npvc_amd/csrc/disc_frame.h"

#include <cstring>
#include <type_traits>
#include <vector>

using namespace vaenpvc::frame;

struct EmuRunner {
  template <class F>
  void phase(F&& f) {
    for (int t = 0; t < NT; ++t) f(t);
  }
  // dst[wave] = sum over the wave's 64 lanes of f(tid) (the device version does it with wave shuffles)
  template <class F>
  void reduce(float* dst, F&& f) {
    float w[NW];
    for (int wv = 0; wv < NW; ++wv) {
      float s = 0.f;
      for (int l = 0; l < 64; ++l) s += f(wv * 64 + l);
      w[wv] = s;
    }
    for (int wv = 0; wv < NW; ++wv) dst[wv] = w[wv];      // (stored after every thread ran: as behind the device's barrier)
  }
  template <class F>
  void reduce2(float* da, float* db, F&& f) {
    float wa[NW], wb[NW];
    for (int wv = 0; wv < NW; ++wv) {
      float sa = 0.f, sb = 0.f;
      for (int l = 0; l < 64; ++l) {
        float a = 0.f, b = 0.f;
        f(wv * 64 + l, a, b);
        sa += a;
        sb += b;
      }
      wa[wv] = sa;
      wb[wv] = sb;
    }
    for (int wv = 0; wv < NW; ++wv) {
      da[wv] = wa[wv];
      db[wv] = wb[wv];
    }
  }
};

// blocks of the weight-gradient launch (256 threads)
struct EmuWRunner {
  template <class F>
  void phase(F&& f) {
    for (int t = 0; t < WT; ++t) f(t);
  }
  template <class St, class Z, class Ac, class Sp, class AccT>
  void frames(int f0, int f1, int fb, St&& st, Z&& z, Ac&& ac, Sp&& sp, AccT&) {
    struct Box {
      AccT a;
    };
    std::vector<Box> v(WT);
    for (int t = 0; t < WT; ++t) z(t, v[t].a);
    for (int f = f0; f < f1; f += fb) {
      const int n = f1 - f < fb ? f1 - f : fb;
      for (int t = 0; t < WT; ++t) st(t, f, n);
      for (int t = 0; t < WT; ++t) ac(t, v[t].a, n);
    }
    for (int t = 0; t < WT; ++t) sp(t, v[t].a);
  }
};

static POff make_off(const int* p) {
  POff o;
  int i = 0;
  o.emb = p[i++];
  for (int l = 0; l < 5; ++l) {
    o.ew[l] = p[i++];
    o.eb[l] = p[i++];
    o.ebeta[l] = p[i++];
    o.egamma[l] = p[i++];
  }
  o.wmu = p[i++];
  o.bmu = p[i++];
  o.wlv = p[i++];
  o.blv = p[i++];
  o.wz = p[i++];
  o.bz = p[i++];
  o.wy = p[i++];
  o.by = p[i++];
  o.bm = p[i++];
  for (int l = 0; l < 4; ++l) {
    o.dw[l] = p[i++];
    o.db[l] = p[i++];
    if (l < 3) {
      o.dbeta[l] = p[i++];
      o.dgamma[l] = p[i++];
    }
  }
  return o;
}

// tensor order of the offset table `t` (floats into `ws`)
enum { T_Y = 39, T_ENC_A = 0, T_ENC_ST = 5, T_Z_MU = 10, T_Z_LV, T_Z, T_EPS_OUT, T_H, T_DEC_A, T_DEC_ST = T_DEC_A + 3, T_XH = T_DEC_ST + 3,
       T_KL, T_NLL, T_D_XH, T_D_DEC_A, T_D_H = T_D_DEC_A + 3, T_D_Z_MU, T_D_Z_LV, T_D_ENC_A, T_LNP = T_D_ENC_A + 5, T_PK, T_G, T_YY, T_COUNT };

extern "C" {

int frame_emu_tensor_count() { return T_COUNT; }
int frame_emu_pack_floats() { return Pk::total; }
int frame_emu_lnp_c() { return LNP_C; }
int frame_emu_lds_floats() { return L_TOTAL; }

void frame_emu_pack(const float* P, const int* poff, float* pk) {
  POff o = make_off(poff);
  for (int i = 0; i < Pk::total; ++i) pk[i] = pack_src(P, o, i);
}

// mode: FM_* bits of the forward pass; do_bwd: 0 forward only, 1 + backward pass, 2 + weight gradients; + 4: the train step
// with the 1025-tap layer split out (toep_split_fwd / toep_split_bwd between the frame kernels, as the product runs it)
int frame_emu_run(const float* P, const int* poff, const float* x, const float* target, const long long* y, const float* eps,
                  const float* z_in, int ny, int F, int mode, int do_bwd, float* ws, const long long* t) {
  POff o = make_off(poff);
  std::vector<float> lds(L_TOTAL, 0.f);
  EmuRunner run;
  FwdArgs a{};
  a.P = P;
  a.pk = ws + t[T_PK];
  a.off = o;
  a.x = x;
  a.target = target ? target : x;
  a.y = reinterpret_cast<const int64_t*>(y);
  a.eps = eps;
  a.z_in = z_in;
  a.ny = ny;
  a.F = F;
  a.mode = mode;
  a.invF = 1.0f / (float)F;
  for (int i = 0; i < 5; ++i) {
    a.enc_a[i] = ws + t[T_ENC_A + i];
    a.enc_st[i] = ws + t[T_ENC_ST + i];
  }
  a.z_mu = ws + t[T_Z_MU];
  a.z_lv = ws + t[T_Z_LV];
  a.z = ws + t[T_Z];
  a.eps_out = ws + t[T_EPS_OUT];
  a.h = ws + t[T_H];
  for (int i = 0; i < 3; ++i) {
    a.dec_a[i] = ws + t[T_DEC_A + i];
    a.dec_st[i] = ws + t[T_DEC_ST + i];
  }
  a.xh = ws + t[T_XH];
  a.kl_f = ws + t[T_KL];
  a.nll_f = ws + t[T_NLL];
  a.d_xh = ws + t[T_D_XH];
  a.dec_y = ws + t[T_YY] + (size_t)F * 12000;
  {
    float* yb = ws + t[T_YY];
    const int ne[5] = {2736, 1824, 1216, 896, 768}, nd[2] = {1824, 2736};
    for (int i = 0; i < 5; ++i) {
      a.y_enc[i] = yb;
      yb += (size_t)F * ne[i];
    }
    for (int i = 0; i < 2; ++i) {
      a.y_dec[i] = yb;
      yb += (size_t)F * nd[i];
    }
  }
  const bool split = (do_bwd & 4) != 0;
  do_bwd &= 3;
  if (split) a.mode |= FM_NOD3;
  frame_prologue(run, lds.data(), P, o);
  for (int f = 0; f < F; ++f)
    frame_fwd(run, lds.data(), a, f, [&](int ff, int d) { return eps ? eps[(size_t)ff * 128 + d] : 0.f; });
  std::vector<float> d_y2;
  if (split) {
    EmuWRunner sr;
    std::vector<float> sl(TS_FWD_LDS > TS_BWD_LDS ? TS_FWD_LDS : TS_BWD_LDS, 0.f), nll8((size_t)F * 8, 0.f);
    const float* w3t = a.pk + Pk::w3t;
    for (int f = 0; f < F; ++f)
      for (int og = 0; og < 8; ++og)
        toep_split_fwd(sr, sl.data(), a.dec_y + (size_t)f * 4104, w3t, P[o.db[3]], a.target + (size_t)f * 513, og, a.xh + (size_t)f * 513,
                       nll8.data() + (size_t)f * 8);
    for (int f = 0; f < F; ++f) {
      float t = 0.f;
      for (int k = 0; k < 8; ++k) t += nll8[(size_t)f * 8 + k];
      a.nll_f[f] = t;
    }
    if (do_bwd) {
      d_y2.resize((size_t)F * 4104);
      for (int f = 0; f < F; ++f)
        for (int c = 0; c < 8; ++c)
          toep_split_bwd(sr, sl.data(), a.xh + (size_t)f * 513, a.target + (size_t)f * 513, w3t, c, a.invF, a.d_xh + (size_t)f * 513,
                         d_y2.data() + (size_t)f * 4104);
    }
  }
  if (!do_bwd) return 0;
  BwdArgs b{};
  b.P = P;
  b.pk = a.pk;
  b.off = o;
  b.target = a.target;
  b.eps = ws + t[T_EPS_OUT];
  b.F = F;
  b.invF = a.invF;
  for (int i = 0; i < 5; ++i) {
    b.enc_a[i] = a.enc_a[i];
    b.enc_st[i] = a.enc_st[i];
    b.d_enc_a[i] = ws + t[T_D_ENC_A + i];
  }
  b.z_mu = a.z_mu;
  b.z_lv = a.z_lv;
  for (int i = 0; i < 3; ++i) {
    b.dec_a[i] = a.dec_a[i];
    b.dec_st[i] = a.dec_st[i];
    b.d_dec_a[i] = ws + t[T_D_DEC_A + i];
  }
  b.xh = a.xh;
  b.d_xh = a.d_xh;
  b.d_h = ws + t[T_D_H];
  b.d_z = nullptr;
  b.d_z_mu = ws + t[T_D_Z_MU];
  b.d_z_lv = ws + t[T_D_Z_LV];
  b.lnp = ws + t[T_LNP];
  b.d_y2 = split ? d_y2.data() : nullptr;
  for (int f = 0; f < F; ++f) frame_bwd(run, lds.data(), b, f);
  if (do_bwd < 2) return 0;
  // ---- every parameter gradient: the job list of the one-launch kernel, block by block
  WgArgs g{};
  g.P = P;
  g.off = o;
  g.x = x;
  g.y = a.y;
  g.ny = ny;
  g.F = F;
  for (int i = 0; i < 5; ++i) {
    g.y_enc[i] = a.y_enc[i];
    g.d_enc_a[i] = b.d_enc_a[i];
  }
  for (int i = 0; i < 2; ++i) g.y_dec[i] = a.y_dec[i];
  g.z = a.z;
  g.h = a.h;
  g.dec_y = a.dec_y;
  g.d_xh = a.d_xh;
  for (int i = 0; i < 3; ++i) g.d_dec_a[i] = b.d_dec_a[i];
  g.d_h = b.d_h;
  g.d_z_mu = b.d_z_mu;
  g.d_z_lv = b.d_z_lv;
  g.lnp = b.lnp;
  g.pk = a.pk;
  g.G = ws + t[T_G];
  const WgPlan pl = make_wgplan(F, ny);
  std::vector<float> wl(WG_LDS, 0.f);
  EmuWRunner wr;
  for (int blk = 0; blk < pl.start[pl.nseg]; ++blk) frame_wgrad_block(wr, wl.data(), g, pl, blk);
  return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------ critic front path
// The five kernels of csrc/disc_frame.h on the host, chained the way vaenpvc_disc_critic_fwd_bwd chains them around the
// 115-tap layer -- which is cut out here: its input gradients (abar1 for pass 2 on the rows xi, da1 for pass 4 on all rows)
// are INPUTS.  tests/test_frame_emu.py differentiates the same two-layer block with float64 autograd.
extern "C" int critic_front_emu(const float* P, const int* off8 /* w0 b0 g0 bt0 w1 b1 g1 bt1 */, const float* rows, int F, const float* abar1,
                                const float* da1, float coef,
                                float* u0, float* st0, float* u1, float* st1, float* ain2, float* g, float* gp_f, float* at1,
                                float* grads /* dW0[112] dW1[3584] db0[16] db1[32] dg0[16] dbt0[16] dg1[32] dbt1[32] */) {
  using namespace vaenpvc::disc::front;
  const int B = 3 * F;
  std::vector<float> lds(L_TOTAL, 0.f), w1t(7 * 16 * 32), ubar1((size_t)F * N1), abar0((size_t)F * N0), ubar0((size_t)F * N0),
      gt((size_t)F * HIN), at0((size_t)F * N0), udir0((size_t)F * N0), pn0((size_t)F * N0), udir1((size_t)F * N1), pn1((size_t)F * N1),
      du1((size_t)B * N1), da0((size_t)B * N0), du0((size_t)B * N0), ab1(abar1, abar1 + (size_t)F * N1), d1(da1, da1 + (size_t)B * N1);
  for (int i = 0; i < 7 * 16 * 32; ++i) {
    const int c = i % 16, o = (i / 16) % 32, t = i / 512;
    w1t[i] = P[off8[4] + (t * 16 + c) * 32 + o];
  }
  FrontArgs a;
  memset(&a, 0, sizeof a);
  a.P = P;
  a.w0 = off8[0]; a.b0 = off8[1]; a.g0 = off8[2]; a.bt0 = off8[3];
  a.w1 = off8[4]; a.b1 = off8[5]; a.g1 = off8[6]; a.bt1 = off8[7];
  a.w1t = w1t.data();
  a.rows = rows;
  a.u0 = u0; a.st0 = st0; a.u1 = u1; a.st1 = st1; a.ain2 = ain2;
  a.abar1 = ab1.data(); a.ubar1 = ubar1.data(); a.abar0 = abar0.data(); a.ubar0 = ubar0.data();
  a.g = g; a.gt = gt.data(); a.gp_f = gp_f;
  a.at0 = at0.data(); a.udir0 = udir0.data(); a.pn0 = pn0.data(); a.at1 = at1; a.udir1 = udir1.data(); a.pn1 = pn1.data();
  a.da1 = d1.data(); a.du1 = du1.data(); a.da0 = da0.data(); a.du0 = du0.data();
  EmuRunner run;
  auto pass = [&](auto mode, int rows0, int nrows) {
    a.rows0 = rows0;
    a.nrows = nrows;
    critic_front_prologue(run, lds.data(), a);
    for (int r = 0; r < nrows; ++r) critic_front_row<decltype(mode)::value>(run, lds.data(), a, r);
  };
  pass(std::integral_constant<int, FP_FWD>(), 0, B);
  a.coef = coef;
  a.penalty = 1;
  pass(std::integral_constant<int, FP_IGRAD>(), 2 * F, F);
  pass(std::integral_constant<int, FP_ADJ>(), 2 * F, F);
  a.add1 = udir1.data();
  a.add0 = udir0.data();
  a.add_row0 = 2 * F;
  pass(std::integral_constant<int, FP_BWD>(), 0, B);
  // job list
  float* dW0 = grads;
  float* dW1 = dW0 + 112;
  float* db0 = dW1 + 3584;
  float* db1 = db0 + 16;
  float* dg0 = db1 + 32;
  float* dbt0 = dg0 + 16;
  float* dg1 = dbt0 + 16;
  float* dbt1 = dg1 + 32;
  for (int i = 0; i < 112 + 3584 + 48 + 96; ++i) grads[i] = 0.f;
  CwArgs ca{rows, gt.data(), u0, st0, u1, st1, P + off8[2], P + off8[3], P + off8[6], P + off8[7], at0.data(), ubar0.data(), ubar1.data(),
            du0.data(), du1.data(), da0.data(), d1.data(), pn0.data(), pn1.data(), dW0, dW1, db0, db1, dg0, dbt0, dg1, dbt1, F, B};
  const CwPlan pl = make_cwplan(F, B);
  std::vector<float> wl(WG_LDS, 0.f);
  EmuWRunner wrun;
  for (int b = 0; b < pl.start[CW_SEGS]; ++b) critic_front_wgrad_block(wrun, wl.data(), ca, pl, b);
  return 0;
}

