// fused 3-D levels, float, tap lengths 14 .. 20 (db7 .. db10): the 512-thread tiles with the 256-register budget
// (analysis 64x16, one column per thread; synthesis 64x32, two items per thread: 14 and 16 taps only -- the fallback of the
// pair-packed kernel for mixed wavelets with odd tap padding).
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_long3_f32(bool inverse, const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, const void* taps_dev, hipStream_t s) {
    if (!inverse) {
        // 16 taps on rows of whole groups of 4: two of the 16 slots of the z window in LDS (Fwd3 WLDS) keep the tall tile free of spills --
        // 512^3 db8 analysis 1.53 -> 1.16 ms per launch, bit-identical (pinned taps on top: 1.20, not used; variant_fwd 3: the spilling form)
        if (t.Lp == 16 && vec4 && variant != 1 && variant != 3) {
            typedef Fused3Tile<float, false, 6> TL;
            return launch_fused3<Fwd3<float, 16, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 1, false, false, false, 2>>(a, t, taps_dev, s);
        }
        // 20 taps on rows of whole groups of 4: 4 of the 20 window slots of each of a thread's two columns in LDS -- the 512-thread tile
        // without its 18 spilled registers: 512^3 db10 analysis 2.79 -> 1.95 ms per launch, bit-identical; 18 taps (no spills as they are) gain nothing from it (variant_fwd 3: the spilling form)
        if (t.Lp == 20 && vec4 && variant != 3) {
            typedef Fused3Tile<float, false, 1> TL;
            return launch_fused3<Fwd3<float, 20, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 1, false, false, false, 4>>(a, t, taps_dev, s);
        }
        if (!vec4 && variant != 3 && (t.Lp == 18 || t.Lp == 20)) {   // ragged rows: 2 / 6 slots in LDS, no spills (8 / 33 without)
            typedef Fused3Tile<float, false, 1> TL;
            if (t.Lp == 18) return launch_fused3<Fwd3<float, 18, TL::TX, TL::TY, TL::NT, TL::RY, false, TL::WPE, 1, false, false, false, 2>>(a, t, taps_dev, s);
            return launch_fused3<Fwd3<float, 20, TL::TX, TL::TY, TL::NT, TL::RY, false, TL::WPE, 1, false, false, false, 6>>(a, t, taps_dev, s);
        }
        // 16 taps on rows of whole groups of 4: two of the 16 slots of the z window in LDS (Fwd3 WLDS) keep the tall tile free of spills --
        // 512^3 db8 analysis 1.53 -> 1.16 ms per launch, bit-identical (pinned taps on top: 1.20, not used; variant_fwd 3: the spilling form)
        if (t.Lp == 16 && vec4 && variant != 1 && variant != 3) {
            typedef Fused3Tile<float, false, 6> TL;
            return launch_fused3<Fwd3<float, 16, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 1, false, false, false, 2>>(a, t, taps_dev, s);
        }
        // 20 taps on rows of whole groups of 4: 4 of the 20 window slots of each of a thread's two columns in LDS -- the 512-thread tile
        // without its 18 spilled registers: 512^3 db10 analysis 2.79 -> 1.95 ms per launch, bit-identical; 18 taps (no spills as they are) gain nothing from it (variant_fwd 3: the spilling form)
        if (t.Lp == 20 && vec4 && variant != 3) {
            typedef Fused3Tile<float, false, 1> TL;
            return launch_fused3<Fwd3<float, 20, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 1, false, false, false, 4>>(a, t, taps_dev, s);
        }
        if (t.Lp == 18 && vec4 && variant == 10) {   // A/B
            typedef Fused3Tile<float, false, 1> TL;
            return launch_fused3<Fwd3<float, 18, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 1, false, false, false, 4>>(a, t, taps_dev, s);
        }
        if (variant != 1) {       // default: tall 64x32 tile, 1024 threads (db7 analysis 1.45 -> 1.15 ms per launch; 16 taps on ragged rows spill 8 registers)
            switch (t.Lp) {
                NDWT_FUSED_CASE(Fwd3, false, float, 14, 6)   // (y items of 2 rows: ndwt_fused_tile.h)
                NDWT_FUSED_CASE(Fwd3, false, float, 16, 6)
                default: break;
            }
        }
        switch (t.Lp) {
            NDWT_FUSED_CASE(Fwd3, false, float, 14, 1)
            NDWT_FUSED_CASE(Fwd3, false, float, 16, 1)
            NDWT_FUSED_CASE(Fwd3, false, float, 18, 1)
            NDWT_FUSED_CASE(Fwd3, false, float, 20, 1)   // spills 18 of its 256 registers; still 5x the per-axis path
            default: return -1;
        }
    }
    switch (t.Lp) {
        NDWT_FUSED_CASE(Inv3S, true, float, 14, 2)
        NDWT_FUSED_CASE(Inv3S, true, float, 16, 2)
        default: return -1;
    }
}
}  // namespace ndwt
