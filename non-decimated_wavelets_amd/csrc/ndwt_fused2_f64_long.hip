// fused 2-D levels, double real data, 14 and 16 taps (db7, db8): the analysis fits the 256-register budget, the synthesis spills 22 / 60
// registers (still 3x the per-axis path)
#include "ndwt_fused_kernels.h"
namespace ndwt {
#define NDWT_D2_CASE(KIND, LL) \
    case LL: return vec4 ? launch_fused2<KIND<double, LL, true, 2>>(a, taps_dev, s) : launch_fused2<KIND<double, LL, false, 2>>(a, taps_dev, s);
int launch_long2_f64(bool inverse, const Fused2Args<double>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s) {
    if (!inverse) {
        switch (Lp) {
            NDWT_D2_CASE(Fwd2S, 14)
            NDWT_D2_CASE(Fwd2S, 16)
            default: return -1;
        }
    }
    switch (Lp) {
        NDWT_D2_CASE(Inv2S, 14)
        NDWT_D2_CASE(Inv2S, 16)
        default: return -1;
    }
}
}  // namespace ndwt
