// fused 3-D inv level, float, real data, stride 1: the pair-packed lane-shift kernel (Inv3Y), tap lengths 2..20
#include "ndwt_fused_kernels.h"
namespace ndwt {

template <int LL, bool V, int DEPTH, bool UNI = false> static int go(const Fused3Args<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Inv3Y<float, LL, inv3y_tx(LL), inv3y_ty(LL), 1024, V, 4, DEPTH, 1, inv3y_zlds(LL, DEPTH), 0, UNI> K;
    FusedTapsD unused;
    unused.Lp = LL;
    return launch_fused3<K>(a, unused, taps_dev, s);
}

// depth 2 (two register sets of band loads, staggered refill) is the default where it fits the 128 registers of a 1024-thread
// workgroup without spills: tap lengths 2, 8 and 10 as they are; 12 with 6 of the pending z sums in LDS
// (inv3y_zlds; 4 and 6 taps fit that way too and run slower than depth 1; a spill reload in the plane loop would wait vmcnt(0), i.e. for every load in flight); depth 1 serves the longer
// filters, unaligned volumes and A/B runs
// D2RAG: depth 2 also for rows that are not whole groups of 4 scalars (the VEC4 = false instance keeps 4 offsets per lane: 2 and 8
// taps fit, 10 and 12 would spill)
#define NDWT_INVY_CASE(LL, D2OK, D2RAG) \
    case LL:                     \
        if constexpr (D2OK) {    \
            if (vec4 && depth == 2) return go<LL, true, 2>(a, taps_dev, s); \
        }                        \
        if constexpr (D2RAG) {   \
            if (!vec4 && depth == 2) return go<LL, false, 2>(a, taps_dev, s); \
        }                        \
        return vec4 ? go<LL, true, 1>(a, taps_dev, s) : go<LL, false, 1>(a, taps_dev, s);

int launch_inv3y_f32(const Fused3Args<float>& a, int Lp, bool vec4, int depth, const void* taps_dev, hipStream_t s, int uniform_yz) {
    // the same taps on the y and z axes, rows of whole groups of 4, the default depth, 12 .. 20 taps: the instance whose z stage reads the y
    // tap pairs (L fewer SGPRs held: 512^3 synthesis db6 -1.8 %, db10 -2.4 % per launch, identical results).  8 taps: -0.6 % on cfg3 and
    // +1.2 % on cfg5's batched volumes in interleaved A/B runs -- within noise of each other, so the 8- and 10-tap kernels stay as they were.
    if (uniform_yz && vec4) {
        switch (Lp) {
#ifndef NDWT_INVY_DB4_ONLY
            case 12: if (depth == 2) return go<12, true, 2, true>(a, taps_dev, s); break;
            case 14: return go<14, true, 1, true>(a, taps_dev, s);
            case 16: return go<16, true, 1, true>(a, taps_dev, s);
            case 18: return go<18, true, 1, true>(a, taps_dev, s);
            case 20: return go<20, true, 1, true>(a, taps_dev, s);
#endif
            default: break;
        }
    }
    switch (Lp) {
        NDWT_INVY_CASE(8, true, true)
#ifndef NDWT_INVY_DB4_ONLY
        NDWT_INVY_CASE(2, true, true)
        NDWT_INVY_CASE(4, false, false)
        NDWT_INVY_CASE(6, false, false)
        NDWT_INVY_CASE(10, true, false)
        NDWT_INVY_CASE(12, true, false)
        NDWT_INVY_CASE(14, false, false)
        NDWT_INVY_CASE(16, false, false)
        NDWT_INVY_CASE(18, false, false)     // 64 x 24 tile: 41 haloed rows on 14 waves
        NDWT_INVY_CASE(20, false, false)     // 48 x 28 tile: 47 haloed rows on 16 waves (3 spilled registers, reloaded once per plane)
#endif
        default: return -1;
    }
}
}  // namespace ndwt
