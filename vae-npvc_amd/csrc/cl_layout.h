// cl_layout.h -- channel-last plane tensors of the conv view GEMMs (gfx950_viewconv.h): geometry shared between the
// workspace layout (model.cpp) and the kernels.  Host-only constants, no device code.
#pragma once
#include <cstdint>

namespace vaenpvc {
namespace tuned {

enum { CL_Y0, CL_Y1, CL_Y2, CL_H, CL_YD0, CL_YD1, CL_GE1, CL_GE2, CL_GE3, CL_GD0, CL_GD1, CL_GD2, CL_COUNT };
constexpr int CL_FWD_COUNT = CL_GE1;  // tensors [0, CL_FWD_COUNT) are written by the forward pass, the rest by backward
struct ClDesc {
  int C, H, CP, HLO, HP;   // channels, positions, padded channels, zero rows before position 0, rows in all
};
// activations: halo of the consumer's view (S-type consumers: HLO = pad; P-type: HLO = 2 and one row behind);
// gradients: halo of their input-gradient site (encoder input gradients are P-type, decoder ones S-type)
constexpr ClDesc CLD[CL_COUNT] = {
    {16, 171, 16, 2, 175},   // CL_Y0  activated output of encoder layer 0   (S-type input of layer 1, pad 2 + 2)
    {32, 57, 32, 2, 61},     // CL_Y1  ... of encoder layer 1                (layer 2, pad 2 + 2)
    {64, 19, 64, 3, 25},     // CL_Y2  ... of encoder layer 2                (layer 3, pad 3 + 3)
    {81, 19, 88, 2, 22},     // CL_H   merge output (81 channels padded to 88), P-type input of decoder layer 0
    {32, 57, 32, 2, 60},     // CL_YD0 activated output of decoder layer 0   (P-type input of decoder layer 1)
    {16, 171, 16, 2, 174},   // CL_YD1 ... of decoder layer 1                (P-type input of decoder layer 2)
    {32, 57, 32, 2, 60},     // CL_GE1 d(pre-LN output of encoder layer 1)   (P-type input of its input gradient)
    {64, 19, 64, 2, 22},     // CL_GE2
    {128, 7, 128, 2, 10},    // CL_GE3
    {32, 57, 32, 3, 63},     // CL_GD0 d(pre-LN output of decoder layer 0)   (S-type, pad 3 + 3)
    {16, 171, 16, 2, 175},   // CL_GD1                                       (pad 2 + 2)
    {8, 513, 8, 2, 517},     // CL_GD2
};
constexpr int DY2_PITCH = 516;   // floats per row of d(activated output of decoder layer 2) when its rows are padded to 16 bytes (8 x 516 per frame)
constexpr int CL_TAIL = 64;  // zero elements behind every plane (K runs padded to the chunk size read into them)
inline int64_t cl_plane(int id, int64_t F) { return F * CLD[id].HP * CLD[id].CP + CL_TAIL; }
// workspace floats of a tensor: three planes of unsigned short
inline int64_t cl_floats(int id, int64_t F) { return cl_plane(id, F) * 3 / 2 + 16; }

// small-batch frame kernels (gfx950_frame.h; asserted equal there): floats of the packed weight copies, channel slots of
// the per-frame LayerNorm sums, largest batch they are reserved for
constexpr int64_t FRAME_PK_FLOATS = 959552 + 15392 + 1539 * 128, FRAME_Y_FLOATS = 12000, FRAME_LNP_C = 552, FRAME_LNP_CAP = 1024;

}  // namespace tuned
}  // namespace vaenpvc
