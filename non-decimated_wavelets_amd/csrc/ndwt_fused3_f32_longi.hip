// fused 3-D synthesis, float, 14 / 16 taps: the lane-shift kernel on the 64x32 tile with 512 threads x 2 items (mixed wavelets with odd tap
// padding, which the pair-packed kernel does not take)
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_long3_f32_inv(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, const void* taps_dev, hipStream_t s) {
    switch (t.Lp) {
        NDWT_FUSED_CASE(Inv3S, true, float, 14, 2)
        NDWT_FUSED_CASE(Inv3S, true, float, 16, 2)
        default: return -1;
    }
}
}  // namespace ndwt
