// fused 3-D levels, float, tap lengths 14 .. 20 (db7 .. db10): dispatch, and the analysis on the tall 64x32 tile (14 / 16 taps).
// The 512-thread analysis instances are in ndwt_fused3_f32_longb.hip, the lane-shift synthesis (14 / 16 taps: the fallback of the pair-packed
// kernel for mixed wavelets with odd tap padding) in ndwt_fused3_f32_longi.hip.
#include "ndwt_fused_kernels.h"
namespace ndwt {
int launch_long3_f32_fwd512(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, const void* taps_dev, hipStream_t s);
int launch_long3_f32_inv(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, const void* taps_dev, hipStream_t s);
int launch_long3_f32(bool inverse, const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, const void* taps_dev, hipStream_t s) {
    if (inverse) return launch_long3_f32_inv(a, t, vec4, taps_dev, s);
    // 16 taps on rows of whole groups of 4: two of the 16 slots of the z window in LDS (Fwd3 WLDS) keep the tall tile free of spills --
    // 512^3 db8 analysis 1.53 -> 1.16 ms per launch, bit-identical (pinned taps on top: 1.20, not used; variant_fwd 3: the spilling form)
    if (t.Lp == 16 && vec4 && variant != 1 && variant != 3) {
        typedef Fused3Tile<float, false, 6> TL;
        return launch_fused3<Fwd3<float, 16, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 1, false, false, false, 2>>(a, t, taps_dev, s);
    }
    if (variant != 1) {       // default: tall 64x32 tile, 1024 threads (db7 analysis 1.45 -> 1.15 ms per launch; 16 taps on ragged rows spill 8 registers)
        switch (t.Lp) {
            NDWT_FUSED_CASE(Fwd3, false, float, 14, 6)   // (y items of 2 rows: ndwt_fused_tile.h)
            NDWT_FUSED_CASE(Fwd3, false, float, 16, 6)
            default: break;
        }
    }
    return launch_long3_f32_fwd512(a, t, vec4, variant, taps_dev, s);
}
}  // namespace ndwt
