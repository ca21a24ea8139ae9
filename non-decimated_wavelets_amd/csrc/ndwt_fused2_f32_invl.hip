// fused 2-D synthesis, float real data, 14 / 16 taps (db7, db8): Inv2S with the 256-register budget (2 waves per SIMD), no spills
// (18 / 20 taps: ndwt_fused2_f32_invm.hip)
#include "ndwt_fused_kernels.h"
namespace ndwt {
#define NDWT_LONG2_CASE(LL) \
    case LL: return vec4 ? launch_fused2<Inv2S<float, LL, true, 2>>(a, taps_dev, s) : launch_fused2<Inv2S<float, LL, false, 2>>(a, taps_dev, s);
int launch_inv2_f32_18to20(const Fused2Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s);
int launch_inv2_f32_14to20(const Fused2Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s) {
    switch (Lp) {
        NDWT_LONG2_CASE(14)
        NDWT_LONG2_CASE(16)
        default: return launch_inv2_f32_18to20(a, Lp, vec4, taps_dev, s);
    }
}
}  // namespace ndwt
