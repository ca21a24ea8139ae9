// fused 3-D inv level, float, real undilated data: the lane-shift kernel Inv3S (what the pair-packed kernel does not cover) and the LDS kernel
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_inv3_f32_ew(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int ew, const void* taps_dev, hipStream_t s);
int launch_inv3_f32(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* taps_dev, hipStream_t s) {
    if (ew != 1) return launch_inv3_f32_ew(a, t, vec4, ew, taps_dev, s);
    NDWT_FUSED_SWITCH_INV_F32_REAL(float)
}
}  // namespace ndwt
