// fused 3-D fwd level, double
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_fwd3_f64(const Fused3Args<double>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* taps_dev, hipStream_t s) {
    if (ew != 1 && ew != 2) return -1;
    if (ew == 2) {
        if (t.Lp == 12 && vec4) {   // complex128 db6: two of the 12 z-window slots in LDS (Fwd3 WLDS): 2 spilled registers instead of 16
            typedef Fused3Tile<double, false, 5> TL;
            return launch_fused3<Fwd3<double, 12, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 2, false, false, false, 2>>(a, t, taps_dev, s);
        }
        switch (t.Lp) {
            NDWT_FUSED_CASE_C(Fwd3, false, double, 2, 0)
            NDWT_FUSED_CASE_C(Fwd3, false, double, 4, 0)
            NDWT_FUSED_CASE_C(Fwd3, false, double, 6, 0)
            NDWT_FUSED_CASE_C(Fwd3, false, double, 8, 1)   // 64x16 tile, 512 threads: the 256-thread tile spills 16 registers (256^3: 0.97 -> 0.58 ms per launch)
            NDWT_FUSED_CASE_C(Fwd3, false, double, 10, 5)  // 64x8 tile, 512 threads
            NDWT_FUSED_CASE_C(Fwd3, false, double, 12, 5)  // (16 spilled registers)
            default: return -1;
        }
    }
    if (variant == 1) {       // 64x16 tile, 512 threads, one column per thread (A/B)
        switch (t.Lp) {
            NDWT_FUSED_CASE(Fwd3, false, double, 6, 1)
            NDWT_FUSED_CASE(Fwd3, false, double, 8, 1)
            default: break;
        }
    }
    switch (t.Lp) {
        NDWT_FUSED_CASE(Fwd3, false, double, 2, 0)
        NDWT_FUSED_CASE(Fwd3, false, double, 4, 0)
        NDWT_FUSED_CASE(Fwd3, false, double, 6, 0)
        NDWT_FUSED_CASE(Fwd3, false, double, 8, 0)
        NDWT_FUSED_CASE(Fwd3, false, double, 10, 1)
        NDWT_FUSED_CASE(Fwd3, false, double, 12, 5)   // 64x8 tile, 512 threads: no spills (64x16 spills 137 registers)
        default: return -1;
    }
}
}  // namespace ndwt
