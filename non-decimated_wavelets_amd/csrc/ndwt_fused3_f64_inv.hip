// fused 3-D inv level, double
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_inv3_f64(const Fused3Args<double>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* taps_dev, hipStream_t s) {
    if (ew != 1 && ew != 2) return -1;
    NDWT_FUSED_SWITCH_INV_F64(double)
}
}  // namespace ndwt
