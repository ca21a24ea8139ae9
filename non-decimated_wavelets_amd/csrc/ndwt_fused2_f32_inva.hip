// fused 2-D synthesis (Inv2S), float, 2 .. 6 taps
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_inv2_f32_short(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    NDWT_FUSED2_SWITCH_SHORT(Inv2S, float)
}
}  // namespace ndwt
