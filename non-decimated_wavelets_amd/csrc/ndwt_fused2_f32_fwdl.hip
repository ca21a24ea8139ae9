// fused 2-D analysis, float real data, 14 .. 20 taps (db7 .. db10): Fwd2S with the 256-register budget (2 waves per SIMD), no spills
#include "ndwt_fused_kernels.h"
namespace ndwt {
#define NDWT_LONG2_CASE(LL) \
    case LL: return vec4 ? launch_fused2<Fwd2S<float, LL, true, 2>>(a, taps_dev, s) : launch_fused2<Fwd2S<float, LL, false, 2>>(a, taps_dev, s);
int launch_fwd2_f32_14to20(const Fused2Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s) {
    switch (Lp) {
        NDWT_LONG2_CASE(14)
        NDWT_LONG2_CASE(16)
        NDWT_LONG2_CASE(18)
        NDWT_LONG2_CASE(20)
        default: return -1;
    }
}
}  // namespace ndwt
