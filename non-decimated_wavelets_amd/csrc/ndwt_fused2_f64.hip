// fused 2-D levels, double
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_fwd2_f64(const Fused2Args<double>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    NDWT_FUSED2_SWITCH(Fwd2S, double)
}
int launch_inv2_f64(const Fused2Args<double>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    NDWT_FUSED2_SWITCH(Inv2S, double)
}
}  // namespace ndwt
