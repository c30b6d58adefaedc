// model.h -- host-side description of the ConvVAE shape chain, the flat parameter
// table and the workspace layout.  Pure C++ (no HIP), shared by every translation unit.
//
// Shape rules follow TensorFlow 'SAME'  for the layers built by
// model/vae.py:72-103 of the reference.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/vaenpvc.h"

namespace vaenpvc {

struct ConvL {
  int cin, hin, cout, hout, k, s, pad;
  bool has_ln;
  int64_t w_off, b_off, beta_off, gamma_off;  // float offsets into the flat parameter buffer
};

struct ParamInfo {
  std::string name;
  int64_t offset;
  int ndim;
  int64_t shape[4];
  int64_t count;
};

struct Region {
  std::string name;
  int64_t offset;  // in floats
  int64_t count;   // in floats
};

struct Model {
  int H, z, ny;
  int n_enc, n_dec;
  ConvL enc[VAENPVC_MAX_LAYERS];
  ConvL dec[VAENPVC_MAX_LAYERS];
  int flat;   // encoder output C*H (768)
  int merge;  // generator input c*h (1539)
  int64_t emb_off, wmu_off, bmu_off, wlv_off, blv_off;
  int64_t wz_off, bz_off, wy_off, by_off, bm_off;
  int64_t n_params;
  std::vector<ParamInfo> table;
  bool is_vcc2016;  // geometry == architecture-vae-vcc2016.json (tuned kernels apply)
};

// returns empty string on success, else an error message
std::string build_model(const vaenpvc_arch& a, Model* m);

// Workspace regions for F frames.  mode: VAENPVC_MODE_*
std::vector<Region> workspace_layout(const Model& m, int64_t F, int mode, int64_t* total_floats);

}  // namespace vaenpvc
