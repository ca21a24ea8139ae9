// fused 2-D synthesis (Inv2S), double
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_inv2_f64(const Fused2Args<double>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    if (Lp > 6) { NDWT_FUSED2_SWITCH_LONG(Inv2S, double) }
    NDWT_FUSED2_SWITCH_SHORT(Inv2S, double)
}
}  // namespace ndwt
