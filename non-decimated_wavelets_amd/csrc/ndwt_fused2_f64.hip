// fused 2-D levels, double: analysis (Fwd2S) and the rows-in-flight synthesis (Inv2P); Inv2S: ndwt_fused2_f64_inv.hip
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_fwd2_f64(const Fused2Args<double>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s) {
    if (Lp > 6) { NDWT_FUSED2_SWITCH_LONG(Fwd2S, double) }
    NDWT_FUSED2_SWITCH_SHORT(Fwd2S, double)
}
// double synthesis of real data, rows of whole groups of 4 scalars: rows of band loads in flight, the row loop unrolled in groups of L
// (Inv2P); up to 8 taps fit the 256-register budget without spills (4 rows in flight with 4 taps, 2 otherwise)
int launch_inv2p_f64(const Fused2Args<double>& a, int Lp, const void* taps_dev, hipStream_t s) {
    switch (Lp) {
        case 2: return launch_fused2<Inv2P<double, 2, 2, 2>>(a, taps_dev, s);
        case 4: return launch_fused2<Inv2P<double, 4, 4, 2>>(a, taps_dev, s);
        case 6: return launch_fused2<Inv2P<double, 6, 2, 2>>(a, taps_dev, s);
        case 8: return launch_fused2<Inv2P<double, 8, 2, 2>>(a, taps_dev, s);
        default: return -1;
    }
}
}  // namespace ndwt
