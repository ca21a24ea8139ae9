// fused 2-D levels, float: analysis (Fwd2S) and the dispatch of the synthesis kernels (their instances: ndwt_fused2_f32_{inva,invb,invp}.hip)
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_fwd2_f32_long(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s);
int launch_inv2_f32_short(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s);
int launch_inv2_f32_long(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s);
int launch_fwd2_f32(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    if (Lp > 6) return launch_fwd2_f32_long(a, Lp, vec4, ew, taps_dev, s);
    NDWT_FUSED2_SWITCH_SHORT(Fwd2S, float)
}
int launch_inv2_f32(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    return Lp > 6 ? launch_inv2_f32_long(a, Lp, vec4, ew, taps_dev, s) : launch_inv2_f32_short(a, Lp, vec4, ew, taps_dev, s);
}
}  // namespace ndwt
