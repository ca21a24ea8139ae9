// fused 3-D inv level, float, real data, stride 1: the pair-packed kernel with its x stage in SCATTER form (Inv3Y<..., XSC>):
// partial sums travel between lanes (v_add_f32_dpp) instead of samples (v_mov_b32_dpp)
#include "ndwt_fused_kernels.h"
namespace ndwt {

template <int LL, int DEPTH, bool UNI> static int go(const Fused3Args<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Inv3Y<float, LL, inv3y_tx(LL), inv3y_ty(LL), 1024, true, 4, DEPTH, 1, inv3y_zlds(LL, DEPTH), 0, UNI, true> K;
    FusedTapsD unused;
    unused.Lp = LL;
    return launch_fused3<K>(a, unused, taps_dev, s);
}

// rows of whole groups of 4 scalars; depth and the shared y / z tap pairs as in launch_inv3y_f32.  -1: no instance (the caller runs the
// gather form).
int launch_inv3ys_f32(const Fused3Args<float>& a, int Lp, int depth, const void* taps_dev, hipStream_t s, int uniform_yz) {
    switch (Lp) {
        case 8: return depth == 2 ? go<8, 2, false>(a, taps_dev, s) : -1;
        case 10: return depth == 2 ? go<10, 2, false>(a, taps_dev, s) : -1;
        case 12: return depth != 2 ? -1 : uniform_yz ? go<12, 2, true>(a, taps_dev, s) : go<12, 2, false>(a, taps_dev, s);
        case 14: return uniform_yz ? go<14, 1, true>(a, taps_dev, s) : go<14, 1, false>(a, taps_dev, s);
        case 16: return uniform_yz ? go<16, 1, true>(a, taps_dev, s) : go<16, 1, false>(a, taps_dev, s);
        case 18: return uniform_yz ? go<18, 1, true>(a, taps_dev, s) : go<18, 1, false>(a, taps_dev, s);
        case 20: return uniform_yz ? go<20, 1, true>(a, taps_dev, s) : go<20, 1, false>(a, taps_dev, s);
        default: return -1;
    }
}
}  // namespace ndwt
