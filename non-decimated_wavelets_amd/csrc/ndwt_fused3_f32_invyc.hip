// fused 3-D inv level, float, interleaved complex data, stride 1: the pair-packed lane-shift kernel (Inv3Y, EW = 2), tap lengths 2..16
#include "ndwt_fused_kernels.h"
namespace ndwt {

template <int LL, bool V, int DEPTH> static int go(const Fused3Args<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Inv3Y<float, LL, inv3y_tx(LL, 2), inv3y_ty(LL, 2), 1024, V, 4, DEPTH, 2> K;
    FusedTapsD unused;
    unused.Lp = LL;
    return launch_fused3<K>(a, unused, taps_dev, s);
}

// depth 2 (two register sets of band loads) where it fits 128 registers without spills, as for real data
#define NDWT_INVYC_CASE(LL, D2OK) \
    case LL:                      \
        if constexpr (D2OK) {     \
            if (vec4 && depth == 2) return go<LL, true, 2>(a, taps_dev, s); \
        }                         \
        return vec4 ? go<LL, true, 1>(a, taps_dev, s) : go<LL, false, 1>(a, taps_dev, s);

// the x stage in scatter form (Inv3Y::xsyn_scatter on (re, im) pairs): rows of whole groups of 4 scalars
template <int LL, int DEPTH> static int gos(const Fused3Args<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Inv3Y<float, LL, inv3y_tx(LL, 2), inv3y_ty(LL, 2), 1024, true, 4, DEPTH, 2, 0, 0, false, true> K;
    FusedTapsD unused;
    unused.Lp = LL;
    return launch_fused3<K>(a, unused, taps_dev, s);
}
int launch_inv3yc_f32(const Fused3Args<float>& a, int Lp, bool vec4, int depth, const void* taps_dev, hipStream_t s, int scatter) {
    if (scatter && vec4) {
        switch (Lp) {
            case 8: return depth == 2 ? gos<8, 2>(a, taps_dev, s) : gos<8, 1>(a, taps_dev, s);
            case 10: return gos<10, 1>(a, taps_dev, s);
            case 12: return gos<12, 1>(a, taps_dev, s);
            case 14: return gos<14, 1>(a, taps_dev, s);
            case 16: return gos<16, 1>(a, taps_dev, s);
            default: break;
        }
    }
    switch (Lp) {
        NDWT_INVYC_CASE(2, true)
        NDWT_INVYC_CASE(4, false)
        NDWT_INVYC_CASE(6, false)
        NDWT_INVYC_CASE(8, true)
        NDWT_INVYC_CASE(10, false)   // (two register sets: 1 spilled register)
        NDWT_INVYC_CASE(12, false)
        NDWT_INVYC_CASE(14, false)   // 48-wide tiles (ndwt_fused_tile.h)
        NDWT_INVYC_CASE(16, false)
        default: return -1;
    }
}

// a level dilated by 4 on real data (EW = 4): rows of whole groups of 4 scalars only
template <int LL, int DEPTH> static int go4(const Fused3Args<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Inv3Y<float, LL, inv3y_tx(LL, 4), inv3y_ty(LL, 4), 1024, true, 4, DEPTH, 4> K;
    FusedTapsD unused;
    unused.Lp = LL;
    return launch_fused3<K>(a, unused, taps_dev, s);
}
// the same with the x stage in scatter form (the sums walk from lane to lane instead of the samples: Inv3Y::xsyn_scatter4)
template <int LL, int DEPTH> static int go4s(const Fused3Args<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Inv3Y<float, LL, inv3y_tx(LL, 4), inv3y_ty(LL, 4), 1024, true, 4, DEPTH, 4, 0, 0, false, true> K;
    FusedTapsD unused;
    unused.Lp = LL;
    return launch_fused3<K>(a, unused, taps_dev, s);
}
int launch_inv3y4_f32(const Fused3Args<float>& a, int Lp, int depth, const void* taps_dev, hipStream_t s, int scatter) {
    if (scatter) {
        switch (Lp) {
            case 4: return go4s<4, 1>(a, taps_dev, s);
            case 6: return go4s<6, 1>(a, taps_dev, s);
            case 8: return depth == 2 ? go4s<8, 2>(a, taps_dev, s) : go4s<8, 1>(a, taps_dev, s);
            default: break;
        }
    }
    switch (Lp) {
        case 2: return depth == 2 ? go4<2, 2>(a, taps_dev, s) : go4<2, 1>(a, taps_dev, s);
        case 4: return go4<4, 1>(a, taps_dev, s);
        case 6: return go4<6, 1>(a, taps_dev, s);
        case 8: return depth == 2 ? go4<8, 2>(a, taps_dev, s) : go4<8, 1>(a, taps_dev, s);
        default: return -1;
    }
}
}  // namespace ndwt
