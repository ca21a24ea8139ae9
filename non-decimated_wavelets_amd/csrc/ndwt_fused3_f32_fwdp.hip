// fused 3-D fwd level, float real, 10 / 12 / 14 taps: tall tile with y items of 2 rows, taps pinned in SGPRs (Fwd3 PIN)
#include "ndwt_fused_kernels.h"
namespace ndwt {
#define NDWT_PIN_CASE(LL)                                                                                                 \
    case LL: {                                                                                                            \
        typedef Fused3Tile<float, false, 6> TL;                                                                           \
        return launch_fused3<Fwd3<float, LL, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 1, false, false, true>>(a, t, taps_dev, s); \
    }
int launch_fwd3_pin_f32(const Fused3Args<float>& a, int Lp, const void* taps_dev, hipStream_t s) {
    FusedTapsD t;
    t.Lp = Lp;
    switch (Lp) {
        NDWT_PIN_CASE(10)
        NDWT_PIN_CASE(12)
        NDWT_PIN_CASE(14)
        default: return -1;
    }
}
}  // namespace ndwt
