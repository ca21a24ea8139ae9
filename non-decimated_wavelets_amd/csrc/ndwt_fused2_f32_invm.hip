// fused 2-D synthesis, float real data, 18 / 20 taps (db9, db10): Inv2S with the 256-register budget (2 waves per SIMD), no spills
#include "ndwt_fused_kernels.h"
namespace ndwt {
#define NDWT_LONG2_CASE(LL) \
    case LL: return vec4 ? launch_fused2<Inv2S<float, LL, true, 2>>(a, taps_dev, s) : launch_fused2<Inv2S<float, LL, false, 2>>(a, taps_dev, s);
int launch_inv2_f32_18to20(const Fused2Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s) {
    switch (Lp) {
        NDWT_LONG2_CASE(18)
        NDWT_LONG2_CASE(20)
        default: return -1;
    }
}
}  // namespace ndwt
