// fused 2-D synthesis, interleaved complex64 data, 10 .. 16 taps (db5 .. db8): Inv2S with the x taps stepping over (re, im) pairs on the
// 256-register budget (2 waves per SIMD), no spills
#include "ndwt_fused_kernels.h"
namespace ndwt {
#define NDWT_C2_CASE(LL) \
    case LL: return vec4 ? launch_fused2<Inv2S<float, LL, true, 2, 2>>(a, taps_dev, s) : launch_fused2<Inv2S<float, LL, false, 2, 2>>(a, taps_dev, s);
int launch_inv2_c64_10to16(const Fused2Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s) {
    switch (Lp) {
        NDWT_C2_CASE(10)
        NDWT_C2_CASE(12)
        NDWT_C2_CASE(14)
        NDWT_C2_CASE(16)
        default: return -1;
    }
}
}  // namespace ndwt
