// fused 2-D levels, float
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_fwd2_f32(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    NDWT_FUSED2_SWITCH(Fwd2S, float)
}
int launch_inv2_f32(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    NDWT_FUSED2_SWITCH(Inv2S, float)
}
}  // namespace ndwt
