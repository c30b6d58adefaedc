// model.cpp -- shape chain, parameter table, workspace layout (host only).
#include "model.h"
#include "cl_layout.h"

#include <algorithm>
#include <cstdio>

namespace vaenpvc {

static std::string fmt(const char* f, int a = 0, int b = 0) {
  char buf[160];
  snprintf(buf, sizeof buf, f, a, b);
  return buf;
}

std::string build_model(const vaenpvc_arch& a, Model* m) {
  // model/vae.py:37-39 asserts len(output)==len(kernel)==len(stride); the struct form
  // cannot express unequal lengths, so the equivalent check is on counts and values.
  if (a.n_enc < 1 || a.n_enc > VAENPVC_MAX_LAYERS) return "encoder: need 1..8 layers";
  if (a.n_dec < 1 || a.n_dec > VAENPVC_MAX_LAYERS) return "generator: need 1..8 layers";
  if (a.H < 1 || a.z_dim < 1 || a.y_dim < 1) return "H, z_dim, y_dim must be positive";
  if (a.gen_h < 1 || a.gen_c < 1) return "generator.hwc must be positive";
  m->H = a.H;
  m->z = a.z_dim;
  m->ny = a.y_dim;
  m->n_enc = a.n_enc;
  m->n_dec = a.n_dec;
  m->table.clear();
  int64_t off = 0;
  auto add = [&](const std::string& name, std::initializer_list<int64_t> shp) {
    ParamInfo p;
    p.name = name;
    p.offset = off;
    p.ndim = (int)shp.size();
    p.count = 1;
    int i = 0;
    for (int64_t s : shp) {
      p.shape[i++] = s;
      p.count *= s;
    }
    for (; i < 4; ++i) p.shape[i] = 1;
    off += p.count;
    m->table.push_back(p);
    return p.offset;
  };
  // model/vae.py:20-24 (embedding width = z_dim, not y_emb_dim)
  m->emb_off = add("y_embedding/y_emb", {a.y_dim, a.z_dim});
  int c = 1, h = a.H;
  for (int i = 0; i < a.n_enc; ++i) {
    ConvL& l = m->enc[i];
    int k = a.enc_kernel[i], s = a.enc_stride[i], o = a.enc_output[i];
    if (k < 1 || s < 1 || o < 1) return fmt("encoder layer %d: kernel/stride/output must be positive", i);
    l.cin = c;
    l.hin = h;
    l.cout = o;
    l.k = k;
    l.s = s;
    l.hout = (h + s - 1) / s;  // TF SAME
    int total = std::max((l.hout - 1) * s + k - h, 0);
    l.pad = total / 2;
    l.has_ln = true;
    std::string p = "Encoder/Conv2d-" + std::to_string(i) + "/";
    l.w_off = add(p + "kernel", {k, 1, c, o});
    l.b_off = add(p + "bias", {o});
    l.beta_off = add(p + "layernorm.offset", {o, 1, 1});
    l.gamma_off = add(p + "layernorm.scale", {o, 1, 1});
    c = o;
    h = l.hout;
  }
  m->flat = c * h;
  m->wmu_off = add("Encoder/dense/kernel", {m->flat, a.z_dim});
  m->bmu_off = add("Encoder/dense/bias", {a.z_dim});
  m->wlv_off = add("Encoder/dense_1/kernel", {m->flat, a.z_dim});
  m->blv_off = add("Encoder/dense_1/bias", {a.z_dim});
  m->merge = a.gen_h * a.gen_c;
  m->wz_off = add("Generator/fully_connected/weights", {a.z_dim, m->merge});
  m->bz_off = add("Generator/fully_connected/biases", {m->merge});
  m->wy_off = add("Generator/fully_connected_1/weights", {a.z_dim, m->merge});
  m->by_off = add("Generator/fully_connected_1/biases", {m->merge});
  m->bm_off = add("Generator/BiasAdd/biases", {m->merge});
  c = a.gen_c;
  h = a.gen_h;
  for (int i = 0; i < a.n_dec; ++i) {
    ConvL& l = m->dec[i];
    int k = a.dec_kernel[i], s = a.dec_stride[i], o = a.dec_output[i];
    if (k < 1 || s < 1 || o < 1) return fmt("generator layer %d: kernel/stride/output must be positive", i);
    l.cin = c;
    l.hin = h;
    l.cout = o;
    l.k = k;
    l.s = s;
    l.hout = h * s;  // conv2d_transpose SAME
    int total = std::max((h - 1) * s + k - l.hout, 0);
    l.pad = total / 2;
    l.has_ln = i < a.n_dec - 1;  // model/vae.py:100-102
    std::string p = std::string("Generator/conv2d_transpose") + (i == 0 ? "" : "_" + std::to_string(i)) + "/";
    l.w_off = add(p + "kernel", {k, 1, o, c});
    l.b_off = add(p + "bias", {o});
    l.beta_off = l.gamma_off = -1;
    if (l.has_ln) {
      l.beta_off = add("Generator/ConvT-LN" + std::to_string(i) + ".offset", {o, 1, 1});
      l.gamma_off = add("Generator/ConvT-LN" + std::to_string(i) + ".scale", {o, 1, 1});
    }
    c = o;
    h = l.hout;
  }
  if (c * h != a.H) return fmt("generator output %d != input bins %d", c * h, a.H);
  m->n_params = off;

  static const int ek[5] = {7, 7, 7, 7, 7}, eo[5] = {16, 32, 64, 128, 256};
  static const int dk[4] = {9, 7, 7, 1025}, ds[4] = {3, 3, 3, 1}, dout[4] = {32, 16, 8, 1};
  // (10 speakers: the merge table, the per-speaker segment sums and the frame kernels size their scratch for VCC2016's ten)
  bool v = a.H == 513 && a.z_dim == 128 && a.y_dim == 10 && a.n_enc == 5 && a.n_dec == 4 && a.gen_h == 19 && a.gen_c == 81;
  if (v)
    for (int i = 0; i < 5; ++i) v = v && a.enc_kernel[i] == ek[i] && a.enc_stride[i] == 3 && a.enc_output[i] == eo[i];
  if (v)
    for (int i = 0; i < 4; ++i) v = v && a.dec_kernel[i] == dk[i] && a.dec_stride[i] == ds[i] && a.dec_output[i] == dout[i];
  m->is_vcc2016 = v;
  return "";
}

std::vector<Region> workspace_layout(const Model& m, int64_t F, int mode, int64_t* total_floats) {
  std::vector<Region> r;
  int64_t off = 0;
  auto add = [&](const std::string& name, int64_t count) {
    Region g{name, off, count};
    r.push_back(g);
    off += (count + 63) / 64 * 64;  // 256-byte aligned regions
  };
  int64_t maxact = 0;
  for (int i = 0; i < m.n_enc; ++i) {
    add("enc_a" + std::to_string(i), F * m.enc[i].cout * m.enc[i].hout);
    add("enc_st" + std::to_string(i), F * 2);
    maxact = std::max<int64_t>(maxact, (int64_t)m.enc[i].cout * m.enc[i].hout);
  }
  add("z_mu", F * m.z);
  add("z_lv", F * m.z);
  add("z", F * m.z);
  add("eps", F * m.z);  // sampler draw when it is generated on the device (seeded entry points)
  add("h", F * m.merge);
  maxact = std::max<int64_t>(maxact, m.merge);
  for (int i = 0; i < m.n_dec - 1; ++i) {
    add("dec_a" + std::to_string(i), F * m.dec[i].cout * m.dec[i].hout);
    add("dec_st" + std::to_string(i), F * 2);
    maxact = std::max<int64_t>(maxact, (int64_t)m.dec[i].cout * m.dec[i].hout);
  }
  // activated output of the last LN layer of the decoder (operand of the final layer's GEMMs)
  if (m.n_dec >= 2) add("dec_y", F * m.dec[m.n_dec - 2].cout * m.dec[m.n_dec - 2].hout);
  // the same tensor as three bf16 planes, bins padded to a multiple of 16 (bf16 MFMA path of the last layer)
  if (m.n_dec >= 2) add("toep_yp", F * 3 * m.dec[m.n_dec - 2].cout * ((m.dec[m.n_dec - 2].hout + 15) / 16 * 16) / 2);
  // bf16 operand planes of the dense-shaped layers on the bf16 matrix cores (gfx950_planegemm.h), up to 3 planes
  // of [F][Kp] unsigned short each: activated outputs of encoder layers 3 and 4, and z
  if (m.is_vcc2016) {
    // channel-last planes with zero halo rows ([F][HP][CP], cl_layout.h: conv layers as view GEMMs)
    for (int i = 0; i < tuned::CL_FWD_COUNT; ++i) add("cl" + std::to_string(i), tuned::cl_floats(i, F));
    add("pl_y3", F * 896 * 3 / 2);
    add("pl_y4", F * 768 * 3 / 2);
    add("pl_z", F * 128 * 3 / 2);
  }
  add("xh", F * m.H);
  add("kl_f", F);
  add("nll_f", F);
  // packed / transposed weight copies of the tuned kernels (F-independent)
  add("scratch", 8 * m.n_params + 65536);
  if (m.is_vcc2016) {  // small-batch frame kernels (gfx950_frame.h): packed weight copies, per-frame LayerNorm channel sums
    add("frame_pk", tuned::FRAME_PK_FLOATS);
    add("frame_lnp", std::min<int64_t>(F, tuned::FRAME_LNP_CAP) * 3 * tuned::FRAME_LNP_C);
    // activated layer outputs the forward pass leaves for the weight-gradient launch (train mode only)
    if (mode == VAENPVC_MODE_TRAIN) add("frame_y", std::min<int64_t>(F, tuned::FRAME_LNP_CAP) * tuned::FRAME_Y_FLOATS);
  }
  if (mode == VAENPVC_MODE_TRAIN) {
    add("d_xh", F * m.H);
    for (int i = m.n_dec - 2; i >= 0; --i) add("d_dec_a" + std::to_string(i), F * m.dec[i].cout * m.dec[i].hout);
    add("d_h", F * m.merge);
    add("d_z", F * m.z);
    add("d_e", F * m.z);
    add("d_z_mu", F * m.z);
    add("d_z_lv", F * m.z);
    for (int i = m.n_enc - 1; i >= 0; --i) add("d_enc_a" + std::to_string(i), F * m.enc[i].cout * m.enc[i].hout);
    // (VCC2016 geometry: room for the 1025-tap layer's input gradient with rows padded to 16 bytes, [F][8][516]; cl_layout.h)
    add("dy_tmp", F * (m.is_vcc2016 ? std::max<int64_t>(maxact, 8 * tuned::DY2_PITCH) : maxact));
    if (m.is_vcc2016) {  // planes of [dz_mu | dz_lv], d(h) (1539 -> 1600 columns) and d(pre-LN output of encoder layer 4)
      add("pl_dz", F * 256 * 3 / 2);
      add("pl_dh", F * 1600 * 3 / 2);
      add("pl_da4", F * 768 * 3 / 2);
      for (int i = tuned::CL_FWD_COUNT; i < tuned::CL_COUNT; ++i) add("cl" + std::to_string(i), tuned::cl_floats(i, F));
    }
    // three bf16 planes (hi, mid, lo) of d_xh, rows zero padded to a multiple of 16 bins (bf16 MFMA path
    // of the last decoder layer)
    add("toep_gp", F * 3 * ((m.H + 15) / 16 * 16) / 2);
  }
  *total_floats = off;
  return r;
}

}  // namespace vaenpvc
