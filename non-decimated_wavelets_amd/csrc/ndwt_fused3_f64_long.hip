// fused 3-D levels, double real, 14 and 16 taps (db7, db8): 64x8 tiles with 512 threads and the 256-register budget (no spills on rows of
// whole 4-element groups; 10 .. 31 spilled registers in the 16-tap analysis and on ragged rows).  18 and 20 taps spill 400+ registers in
// this form and stay on the per-axis path.
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_long3_f64(bool inverse, const Fused3Args<double>& a, const FusedTapsD& t, bool vec4, const void* taps_dev, hipStream_t s) {
    if (!inverse) {
        if (t.Lp == 16 && vec4) {   // two of the 16 z-window slots in LDS (Fwd3 WLDS): no spills (13 without)
            typedef Fused3Tile<double, false, 5> TL;
            return launch_fused3<Fwd3<double, 16, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 1, false, false, false, 2>>(a, t, taps_dev, s);
        }
        switch (t.Lp) {
            NDWT_FUSED_CASE(Fwd3, false, double, 14, 5)
            NDWT_FUSED_CASE(Fwd3, false, double, 16, 5)
            default: return -1;
        }
    }
    switch (t.Lp) {
        NDWT_FUSED_CASE(Inv3S, true, double, 14, 5)
        NDWT_FUSED_CASE(Inv3S, true, double, 16, 5)
        default: return -1;
    }
}
}  // namespace ndwt
