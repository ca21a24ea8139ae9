// fused 2-D synthesis with rows of band loads in flight (Inv2P), float
#include "ndwt_fused_kernels.h"
namespace ndwt {
// float synthesis of real data, rows of whole groups of 4 scalars: PD rows of band loads in flight per wave, the row loop unrolled in
// groups of L (Inv2P); 2 waves per SIMD (the 256-register budget)
template <int LL, int PD, bool PK = false> static int go_p(const Fused2Args<float>& a, const void* taps_dev, hipStream_t s) {
    return launch_fused2<Inv2P<float, LL, PD, 2, PK>>(a, taps_dev, s);
}
int launch_inv2p_f32(const Fused2Args<float>& a, int Lp, int depth, const void* taps_dev, hipStream_t s, int packed) {
    if (packed && depth == 4) {      // packed FMAs on pairs of adjacent x outputs, tap pairs pinned in SGPRs
        switch (Lp) {
            case 4: return go_p<4, 4, true>(a, taps_dev, s);
            case 8: return go_p<8, 4, true>(a, taps_dev, s);
            case 12: return go_p<12, 4, true>(a, taps_dev, s);
            default: break;
        }
    }
    switch (Lp) {
        case 2: return go_p<2, 2>(a, taps_dev, s);
        case 4: return depth == 4 ? go_p<4, 4>(a, taps_dev, s) : go_p<4, 2>(a, taps_dev, s);
        case 6: return go_p<6, 2>(a, taps_dev, s);
        case 8: return depth == 4 ? go_p<8, 4>(a, taps_dev, s) : go_p<8, 2>(a, taps_dev, s);
        case 10: return go_p<10, 2>(a, taps_dev, s);
        case 12: return depth == 4 ? go_p<12, 4>(a, taps_dev, s) : go_p<12, 2>(a, taps_dev, s);
        default: return -1;
    }
}
}  // namespace ndwt
