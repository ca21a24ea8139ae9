// Level 1 of a denoising step (dec -> shrink -> rec) without its detail bands in memory, float, real data, tap lengths 2 .. 8:
// Den3 (analysis of the haloed tile + thresholding + synthesis in one launch) and the approximation-only analysis Fwd3<.., LOWONLY>
// that feeds the deeper levels.  Reference use case: README.md:2 ("iterative algorithm").
#include "ndwt_fused_kernels.h"
namespace ndwt {

template <int LL> static int go_den(const Fused3Args<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Den3<float, LL, 1024, 4, (LL == 8 ? 6 : 0)> K;   // 8 taps: 6 of the 8 pending z sums in LDS (no spills)
    FusedTapsD unused;
    unused.Lp = LL;
    return launch_fused3<K>(a, unused, taps_dev, s);
}

int launch_den3_f32(const Fused3Args<float>& a, int Lp, const void* taps_dev, hipStream_t s) {
    switch (Lp) {
        case 2: return go_den<2>(a, taps_dev, s);
        case 4: return go_den<4>(a, taps_dev, s);
        case 6: return go_den<6>(a, taps_dev, s);
        case 8: return go_den<8>(a, taps_dev, s);
        default: return -1;
    }
}

// band 0 only: TILE 2 = the tall 64 x 32 tile of the float analysis (1024 threads, one workgroup per CU), 0 = 64 x 16 (256 threads)
template <int LL, bool V, int TILE = 2> static int go_low(const Fused3Args<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Fused3Tile<float, false, TILE> TL;
    typedef Fwd3<float, LL, TL::TX, TL::TY, TL::NT, TL::RY, V, TL::WPE, 1, true> K;
    FusedTapsD unused;
    unused.Lp = LL;
    return launch_fused3<K>(a, unused, taps_dev, s);
}

int launch_fwd3_low_f32(const Fused3Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s) {
    if (a.nty == (a.n2 + 15) / 16 && a.nty != (a.n2 + 31) / 32) {     // the caller laid out 64 x 16 tiles
        switch (Lp) {
            case 2: return vec4 ? go_low<2, true, 0>(a, taps_dev, s) : go_low<2, false, 0>(a, taps_dev, s);
            case 4: return vec4 ? go_low<4, true, 0>(a, taps_dev, s) : go_low<4, false, 0>(a, taps_dev, s);
            case 6: return vec4 ? go_low<6, true, 0>(a, taps_dev, s) : go_low<6, false, 0>(a, taps_dev, s);
            case 8: return vec4 ? go_low<8, true, 0>(a, taps_dev, s) : go_low<8, false, 0>(a, taps_dev, s);
            default: return -1;
        }
    }
    switch (Lp) {
        case 2: return vec4 ? go_low<2, true>(a, taps_dev, s) : go_low<2, false>(a, taps_dev, s);
        case 4: return vec4 ? go_low<4, true>(a, taps_dev, s) : go_low<4, false>(a, taps_dev, s);
        case 6: return vec4 ? go_low<6, true>(a, taps_dev, s) : go_low<6, false>(a, taps_dev, s);
        case 8: return vec4 ? go_low<8, true>(a, taps_dev, s) : go_low<8, false>(a, taps_dev, s);
        default: return -1;
    }
}

// 4-D analysis with the t axis folded in (Fwd3<.., TPRE>): the tall tile with y items of 2 rows (Fused3Tile<float, false, 6>: the 8
// prefetched frames fit its register budget), rows of whole groups of 4 scalars
template <int LL> static int go_tpre(const Fused3Args<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Fused3Tile<float, false, 6> TL;
    typedef Fwd3<float, LL, TL::TX, TL::TY, TL::NT, TL::RY, true, TL::WPE, 1, false, true> K;
    FusedTapsD unused;
    unused.Lp = LL;
    return launch_fused3<K>(a, unused, taps_dev, s);
}

int launch_fwd3_tpre_f32(const Fused3Args<float>& a, int Lp, const void* taps_dev, hipStream_t s) {
    switch (Lp) {
        case 2: return go_tpre<2>(a, taps_dev, s);
        case 4: return go_tpre<4>(a, taps_dev, s);
        case 6: return go_tpre<6>(a, taps_dev, s);
        case 8: return go_tpre<8>(a, taps_dev, s);
        default: return -1;
    }
}
}  // namespace ndwt
