// placeholder until the tuned kernels land
#include "kernels.h"
namespace vaenpvc { namespace tuned {
bool available() { return false; }
void encoder_fwd(const Model&, const float*, const float*, int64_t, const Ws&, hipStream_t) {}
void decoder_fwd(const Model&, const float*, const float*, const int64_t*, int64_t, const Ws&, float*, hipStream_t) {}
void backward(const Model&, const float*, const float*, const int64_t*, const float*, int64_t, const Ws&, float*, hipStream_t) {}
}}
