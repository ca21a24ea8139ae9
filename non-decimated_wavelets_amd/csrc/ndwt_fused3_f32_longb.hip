// fused 3-D analysis, float, 14 .. 20 taps on the 512-thread 64x16 tile (256-register budget): the kernels of 18 / 20 taps, and of 14 / 16
// taps on request (variant_fwd 1)
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_long3_f32_fwd512(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, const void* taps_dev, hipStream_t s) {
    // 20 taps on rows of whole groups of 4: 4 of the 20 window slots of each of a thread's two columns in LDS (Fwd3 WLDS) -- the tile without
    // its 18 spilled registers: 512^3 db10 analysis 2.79 -> 1.95 ms per launch, bit-identical; 18 taps do not spill and gain nothing from it
    // (variant_fwd 3: the spilling form)
    if (t.Lp == 20 && vec4 && variant != 3) {
        typedef Fused3Tile<float, false, 1> TL;
        return launch_fused3<Fwd3<float, 20, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 1, false, false, false, 4>>(a, t, taps_dev, s);
    }
    if (!vec4 && variant != 3 && (t.Lp == 18 || t.Lp == 20)) {   // ragged rows: 2 / 6 slots in LDS, no spills (8 / 33 without)
        typedef Fused3Tile<float, false, 1> TL;
        if (t.Lp == 18) return launch_fused3<Fwd3<float, 18, TL::TX, TL::TY, TL::NT, TL::RY, false, TL::WPE, 1, false, false, false, 2>>(a, t, taps_dev, s);
        return launch_fused3<Fwd3<float, 20, TL::TX, TL::TY, TL::NT, TL::RY, false, TL::WPE, 1, false, false, false, 6>>(a, t, taps_dev, s);
    }
    switch (t.Lp) {
        NDWT_FUSED_CASE(Fwd3, false, float, 14, 1)
        NDWT_FUSED_CASE(Fwd3, false, float, 16, 1)
        NDWT_FUSED_CASE(Fwd3, false, float, 18, 1)
        NDWT_FUSED_CASE(Fwd3, false, float, 20, 1)   // (variant_fwd 3 / ragged-free reference form: spills 18 of its 256 registers)
        default: return -1;
    }
}
}  // namespace ndwt
