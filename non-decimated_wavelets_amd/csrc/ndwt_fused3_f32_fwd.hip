// fused 3-D fwd level, float
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_fwd3_f32(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* taps_dev, hipStream_t s) {
    NDWT_FUSED_SWITCH_FWD_F32(float)
}
}  // namespace ndwt
