// fused 2-D analysis, float, 8 .. 12 taps
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_fwd2_f32_long(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    NDWT_FUSED2_SWITCH_LONG(Fwd2S, float)
}
}  // namespace ndwt
