// fused 3-D inv level, float, the lane-shift kernel Inv3S with the x taps stepping over 2 (interleaved complex, a level dilated by 2) or 4
// (a level dilated by 4) scalars
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_inv3_f32_ew(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    NDWT_FUSED_SWITCH_INV_F32_EW(float)
    return -1;
}
}  // namespace ndwt
