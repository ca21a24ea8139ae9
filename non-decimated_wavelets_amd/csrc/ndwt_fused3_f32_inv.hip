// fused 3-D inv level, float
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_inv3_f32(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* taps_dev, hipStream_t s) {
    NDWT_FUSED_SWITCH_INV_F32(float)
}
}  // namespace ndwt
