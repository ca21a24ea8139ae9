// Host emulation of the fused HIP kernels -- TEST INFRASTRUCTURE, not a product path.
// Compiles non-decimated_wavelets_amd/csrc/ndwt_device.h with NDWT_HOST_EMU (plain C++, clang) and
// runs every workgroup's threads sequentially between barriers, under AddressSanitizer/UBSan, so
// out-of-bounds global/LDS indexing is caught on the CPU before a kernel ever reaches a GPU.
// (GPU AddressSanitizer is not available on the pool.)  The library never links this file.
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#define NDWT_HOST_EMU 1
#include "ndwt_device.h"
#include "ndwt_geom.h"
#include "ndwt_fused_tile.h"

// The pair-packed synthesis instantiations (10 tap lengths x 2 access widths x 2 depths x tiles) live in translation units of their
// own (EMU_PART 9 .. 12): as part of emu3<float, true> they made that one unit compile for 6.5 minutes.
int emu_y_small(int depth, int Lp, int vec4, ndwt::Fused3Args<float>& a, const double* lo, const double* hi);
int emu_y_small2(int Lp, int vec4, ndwt::Fused3Args<float>& a, const double* lo, const double* hi);
int emu_y_prod(int depth, int Lp, int vec4, ndwt::Fused3Args<float>& a, const double* lo, const double* hi);
int emu_yc_small(int depth, int Lp, int vec4, ndwt::Fused3Args<float>& a, const double* lo, const double* hi);
int emu_y_dilated(int ew, int depth, int Lp, ndwt::Fused3Args<float>& a, const double* lo, const double* hi);
int emu_y_scatter(int small, int Lp, ndwt::Fused3Args<float>& a, const double* lo, const double* hi);
int emu_y_dilated4s(int Lp, ndwt::Fused3Args<float>& a, const double* lo, const double* hi);
int emu_yc_scatter(int small, int Lp, ndwt::Fused3Args<float>& a, const double* lo, const double* hi);

namespace {

template <class State, int NT> struct EmuExec {
    std::vector<State> st;
    EmuExec() : st(NT) {}
    // value an expression takes in lane (tid + D) of the same 64-lane wave (0 outside it): what the DPP wave
    // shifts deliver on the GPU
    template <int D, class G> auto peer_value(int tid, G&& getter) -> decltype(getter(st[0])) {
        const int lane = tid % 64 + D;
        if (lane < 0 || lane > 63) return decltype(getter(st[0]))(0);
        return getter(st[tid + D]);
    }
    template <class F> void each(F&& f) {
        for (int tid = 0; tid < NT; ++tid) f(tid, st[tid]);
    }
    void barrier() {}
};

template <class K, typename T>
int run(ndwt::Fused3Args<T>& a, const double* lo, const double* hi) {
    typename K::Taps tp;
    for (int ax = 0; ax < 3; ++ax)
        for (int j = 0; j < K::L; ++j) {
            tp.lo[ax][j] = (T)lo[ax * ndwt::kMaxTaps + j];
            tp.hi[ax][j] = (T)hi[ax * ndwt::kMaxTaps + j];
        }
    const int nblocks = a.ntx * a.nty * a.nzc * a.nbatch;
    // LDS image on the heap with exact size so ASan sees overruns
    for (int b = 0; b < nblocks; ++b) {
        std::unique_ptr<typename K::Shared> sh(new typename K::Shared);
        EmuExec<typename K::State, K::NT> ex;
        K::block(ex, *sh, a, tp, b);
    }
    return 0;
}

// the pair-packed float synthesis kernel (Inv3Y): extended tap table
template <class K, typename T>
int runY(ndwt::Fused3Args<T>& a, const double* lo, const double* hi) {
    typename K::Taps tp;
    for (int ax = 0; ax < 3; ++ax)
        for (int j = 0; j < K::L; ++j) {
            tp.lo[ax][j] = (T)lo[ax * ndwt::kMaxTaps + j];
            tp.hi[ax][j] = (T)hi[ax * ndwt::kMaxTaps + j];
        }
    for (int k = 0; k <= K::L; ++k)
        for (int h = 0; h < 2; ++h) {
            const int j = k - h;
            tp.xplo[k][h] = (j >= 0 && j < K::L) ? (T)lo[j] : T(0);
            tp.xphi[k][h] = (j >= 0 && j < K::L) ? (T)hi[j] : T(0);
        }
    const int nblocks = a.ntx * a.nty * a.nzc * a.nbatch;
    for (int b = 0; b < nblocks; ++b) {
        std::unique_ptr<typename K::Shared> sh(new typename K::Shared);
        EmuExec<typename K::State, K::NT> ex;
        K::block(ex, *sh, a, tp, b);
    }
    return 0;
}

template <typename T, int TX, int TY, int NT, bool ALL, int DEPTH, int EW = 1>
int dispatchY(int Lp, int vec4, ndwt::Fused3Args<T>& a, const double* lo, const double* hi) {
    // pending z sums in LDS: the production choice on the production tile, half of them on the small test tile (every tap length)
#define ZL(LL) (ALL ? ((DEPTH == 2 && LL >= 4) ? LL / 2 : 0) : ndwt::inv3y_zlds(LL, DEPTH, EW))
#define CASEY(LL)                                                                   \
    case LL:                                                                        \
        return vec4 ? runY<ndwt::Inv3Y<T, LL, TX, TY, NT, true, 2, DEPTH, EW, ZL(LL)>, T>(a, lo, hi)  \
                    : runY<ndwt::Inv3Y<T, LL, TX, TY, NT, false, 2, DEPTH, EW, ZL(LL)>, T>(a, lo, hi);
    if constexpr (ALL) {
        switch (Lp) {
            CASEY(2) CASEY(4) CASEY(6) CASEY(8) CASEY(10) CASEY(12) CASEY(14) CASEY(16) CASEY(18) CASEY(20)
            default: return -1;
        }
    } else {
        switch (Lp) {
            CASEY(8) CASEY(12)
            default: return -1;
        }
    }
#undef CASEY
#undef ZL
}

template <typename T, template <typename, int, int, int, int, int, bool, int, int> class KIND, int TX, int TY, int NT, int RY, bool ALL, int EW = 1>
int dispatch(int Lp, int vec4, ndwt::Fused3Args<T>& a, const double* lo, const double* hi) {
#define CASE(LL)                                                              \
    case LL:                                                                  \
        return vec4 ? run<KIND<T, LL, TX, TY, NT, RY, true, 2, EW>, T>(a, lo, hi)    \
                    : run<KIND<T, LL, TX, TY, NT, RY, false, 2, EW>, T>(a, lo, hi);
    if constexpr (ALL) {
        switch (Lp) {
            CASE(2) CASE(4) CASE(6) CASE(8) CASE(10) CASE(12)
            default: return -1;
        }
    } else {   // the production tile shape is emulated for db4 only (build time)
        switch (Lp) {
            CASE(8)
            default: return -1;
        }
    }
#undef CASE
}

// INV is a compile-time parameter so that the analysis and synthesis kernels land in separate translation units
// (tests/emu/Makefile compiles this file once per EMU_PART, in parallel)
template <typename T, bool INV>
int emu3(int Lp, int vec4, const T* in, T* out, int n1, int n2, int n3, int nbatch, int zchunk,
         const double* lo, const double* hi, int z_wrap, int small_tile, int variant, int ew, double shrink_thr, int shrink_mask,
         int shrink_hard, int dil) {
    ndwt::Fused3Args<T> a;
    std::memset(&a, 0, sizeof(a));
    a.n1 = n1; a.n2 = n2; a.n3 = n3; a.nbatch = nbatch;
    const long long vol = (long long)n1 * n2 * n3;
    if (dil > 1) {              // a level dilated by `dil` (n2, n3 are the full sizes): one (y, z) sub-lattice per batch item, x by EW = dil
        a.n2 = n2 / dil; a.n3 = n3 / dil; a.nbatch = dil * dil;
        ew = dil;
    }
    const int halo = Lp - 1;
    const long long vol_in = z_wrap ? vol : (long long)n1 * n2 * (n3 + halo);
    a.z_wrap = z_wrap;
    a.zlo = 0;
    a.zhi = n3 - (Lp - 1);
    if (INV) { a.shrink_thr = (T)shrink_thr; a.shrink_mask = shrink_mask; a.shrink_hard = shrink_hard; }
    a.in_bstride = vol_in; a.out_bstride = vol;
    if constexpr (!INV) {
        a.in[0] = in;
        for (int b = 0; b < 8; ++b) a.out[b] = out + b * vol * nbatch;
    } else {
        for (int b = 0; b < 8; ++b) a.in[b] = in + b * vol_in * nbatch;
        a.out[0] = out;
    }
    auto geometry = [&](int TX, int TY) {
        ndwt::fused3_geometry(a, TX, TY, Lp, 2048, zchunk);
        if (dil > 1) {
            a.rs = dil * n1;
            a.plane = (long long)dil * n1 * n2;
            a.bsplit = dil;
            a.in_bstride = a.out_bstride = n1;
            a.in_bstride2 = a.out_bstride2 = (long long)n1 * n2;
        }
    };
    // the tile shapes the library launches
    typedef ndwt::Fused3Tile<T, false, 0> PF;
    typedef ndwt::Fused3Tile<T, true, 0> PI;
    typedef ndwt::Fused3Tile<T, false, 1> PF1;
    typedef ndwt::Fused3Tile<T, true, 1> PI1;   // the default lane-shift synthesis configuration (float and double)
    typedef ndwt::Fused3Tile<T, true, 2> PI2;   // lane-shift synthesis, smaller tile
    if (ew == 4) {      // level dilated by 4: the library's 512-thread tiles (float only)
        if constexpr (sizeof(T) == 4) {
            typedef ndwt::Fused3Tile<T, false, 1> DF;
            typedef ndwt::Fused3Tile<T, true, 4> DI;
            if constexpr (INV) {
                if (variant == 10 && vec4 && Lp >= 4 && Lp <= 8) {         // ... with the x stage in scatter form (the sums walk from lane to lane)
                    geometry(ndwt::inv3y_tx(Lp, 4), ndwt::inv3y_ty(Lp, 4));
                    return emu_y_dilated4s(Lp, a, lo, hi);
                }
                if ((variant == 5 || variant == 8) && vec4 && Lp <= 8) {   // the pair-packed kernel on whole-lane x shifts (EW = 4), production tiles
                    geometry(ndwt::inv3y_tx(Lp, 4), ndwt::inv3y_ty(Lp, 4));
                    return emu_y_dilated(4, variant == 5 ? 1 : 2, Lp, a, lo, hi);
                }
                geometry(DI::TX, DI::TY);
                return dispatch<T, ndwt::Inv3S, DI::TX, DI::TY, DI::NT, DI::RY, false, 4>(Lp, vec4, a, lo, hi);
            } else {
                geometry(DF::TX, DF::TY);
                return dispatch<T, ndwt::Fwd3, DF::TX, DF::TY, DF::NT, DF::RY, false, 4>(Lp, vec4, a, lo, hi);
            }
        }
        return -1;
    }
    if (ew == 2) {      // interleaved complex: n1 counts scalars; lane-shift synthesis and LDS analysis kernels
        if constexpr (INV && sizeof(T) == 4) {   // ... with the x stage in scatter form: small tile (every tap length), production tiles 10 .. 16 taps
            if (variant == 10 && vec4 && Lp <= 16 && (small_tile || Lp >= 10)) {
                if (small_tile) geometry(16, 8);
                else geometry(ndwt::inv3y_tx(Lp, 2), ndwt::inv3y_ty(Lp, 2));
                return emu_yc_scatter(small_tile ? 1 : 0, Lp, a, lo, hi);
            }
        }
        if constexpr (INV && sizeof(T) == 4) {   // pair-packed synthesis on (re, im) pairs: small tile, every tap length up to 16
            if ((variant == 5 || variant == 8) && small_tile && Lp <= 16) {
                geometry(16, 8);
                return emu_yc_small(variant == 5 ? 1 : 2, Lp, vec4, a, lo, hi);
            }
        }
        if constexpr (INV && sizeof(T) == 4) {   // a level dilated by 2 on the production tile of the pair-packed kernel's interleaved form
            if ((variant == 5 || variant == 8) && !small_tile && vec4 && dil == 2 && (Lp == 4 || Lp == 8)) {
                geometry(ndwt::inv3y_tx(Lp, 2), ndwt::inv3y_ty(Lp, 2));
                return emu_y_dilated(2, variant == 5 ? 1 : 2, Lp, a, lo, hi);
            }
        }
        if (small_tile) {
            geometry(16, 8);
            if constexpr (INV) return dispatch<T, ndwt::Inv3S, 16, 8, 128, 2, true, 2>(Lp, vec4, a, lo, hi);
            else return dispatch<T, ndwt::Fwd3, 16, 8, 64, 2, true, 2>(Lp, vec4, a, lo, hi);
        }
        if constexpr (INV) {
            geometry(PI1::TX, PI1::TY);
            return dispatch<T, ndwt::Inv3S, PI1::TX, PI1::TY, PI1::NT, PI1::RY, false, 2>(Lp, vec4, a, lo, hi);
        } else {
            geometry(PF::TX, PF::TY);
            return dispatch<T, ndwt::Fwd3, PF::TX, PF::TY, PF::NT, PF::RY, false, 2>(Lp, vec4, a, lo, hi);
        }
    }
    if constexpr (INV && sizeof(T) == 4) {   // pair-packed synthesis (Inv3Y): variant 5 one register set of band loads, 8 two (staggered refill)
        if (variant == 10 && vec4) {            // its x stage in scatter form (XSC): the library's default for 10 .. 20 taps
            if (small_tile) geometry(16, 8);
            else geometry(ndwt::inv3y_tx(Lp), ndwt::inv3y_ty(Lp));
            return emu_y_scatter(small_tile ? 1 : 0, Lp, a, lo, hi);
        }
        if (variant == 5 || variant == 8) {
            if (small_tile) {
                geometry(16, 8);
                return variant == 5 ? emu_y_small(1, Lp, vec4, a, lo, hi) : emu_y_small2(Lp, vec4, a, lo, hi);   // depth 2 on 8 waves: both refill schedules run
            }
            geometry(ndwt::kInv3YTX, ndwt::kInv3YTY);
            return emu_y_prod(variant == 5 ? 1 : 2, Lp, vec4, a, lo, hi);
        }
    }
    if (small_tile) {   // a second tile shape exercises different item/lane mappings
        geometry(16, 8);
        if constexpr (INV) {
            if (variant == 2)   // lane-shift synthesis on the small tile, every tap length
                return dispatch<T, ndwt::Inv3S, 16, 8, 128, 2, true>(Lp, vec4, a, lo, hi);
            return dispatch<T, ndwt::Inv3, 16, 8, 64, 2, true>(Lp, vec4, a, lo, hi);
        } else {
            return dispatch<T, ndwt::Fwd3, 16, 8, 64, 2, true>(Lp, vec4, a, lo, hi);
        }
    }
    if constexpr (INV) {
        if (variant == 3) {
            geometry(PI2::TX, PI2::TY);
            return dispatch<T, ndwt::Inv3S, PI2::TX, PI2::TY, PI2::NT, PI2::RY, false>(Lp, vec4, a, lo, hi);
        }
        if (variant == 1) {
            geometry(PI1::TX, PI1::TY);
            return dispatch<T, ndwt::Inv3S, PI1::TX, PI1::TY, PI1::NT, PI1::RY, false>(Lp, vec4, a, lo, hi);
        }
        geometry(PI::TX, PI::TY);
        return dispatch<T, ndwt::Inv3, PI::TX, PI::TY, PI::NT, PI::RY, false>(Lp, vec4, a, lo, hi);
    } else {
        if (variant == 1) {
            geometry(PF1::TX, PF1::TY);
            return dispatch<T, ndwt::Fwd3, PF1::TX, PF1::TY, PF1::NT, PF1::RY, false>(Lp, vec4, a, lo, hi);
        }
        geometry(PF::TX, PF::TY);
        return dispatch<T, ndwt::Fwd3, PF::TX, PF::TY, PF::NT, PF::RY, false>(Lp, vec4, a, lo, hi);
    }
}

template <class TP> static auto fill_x_pairs(TP& tp, const double* lo, const double* hi, int L, int) -> decltype((void)tp.xplo) {
    for (int k = 0; k <= L; ++k)
        for (int h = 0; h < 2; ++h) {
            const int j = k - h;
            typedef typename std::remove_reference<decltype(tp.xplo[0][0])>::type E;
            tp.xplo[k][h] = (j >= 0 && j < L) ? (E)lo[j] : E(0);
            tp.xphi[k][h] = (j >= 0 && j < L) ? (E)hi[j] : E(0);
        }
}
template <class TP> static void fill_x_pairs(TP&, const double*, const double*, int, long) {}
template <class K, typename T> int run2(ndwt::Fused2Args<T>& a, const double* lo, const double* hi) {
    typename K::Taps tp;
    for (int ax = 0; ax < 3; ++ax)
        for (int j = 0; j < K::L; ++j) {
            tp.lo[ax][j] = (T)lo[ax * ndwt::kMaxTaps + j];
            tp.hi[ax][j] = (T)hi[ax * ndwt::kMaxTaps + j];
        }
    fill_x_pairs(tp, lo, hi, K::L, 0);
    const int nblocks = a.ntx * a.nyc * a.nbatch;
    for (int b = 0; b < nblocks; ++b) {
        typename K::Shared sh;
        EmuExec<typename K::State, K::NT> ex;
        K::block(ex, sh, a, tp, b);
    }
    return 0;
}

#define EMU2_GEOM(a, ...)                    \
    do {                                     \
        ndwt::fused2_geometry(a, __VA_ARGS__); \
        if (dil > 1) a.rs = dil * n1;        \
    } while (0)

template <typename T>
int emu2(int inverse, int Lp, int vec4, const T* in, T* out, int n1, int n2, int ychunk, const double* lo, const double* hi,
         int y_wrap, int ew, double shrink_thr, int shrink_mask, int shrink_hard, int dil) {
    ndwt::Fused2Args<T> a;
    std::memset(&a, 0, sizeof(a));
    a.n1 = n1; a.n2 = n2; a.nbatch = 1;
    if (dil > 1) { a.n2 = n2 / dil; a.nbatch = dil; ew = dil; }     // a level dilated by dil: row sub-lattices as batch items, x by EW
    const long long vol = (long long)n1 * n2;
    const long long vol_in = y_wrap ? vol : (long long)n1 * (n2 + Lp - 1);
    a.y_wrap = y_wrap;
    if (inverse) { a.shrink_thr = (T)shrink_thr; a.shrink_mask = shrink_mask; a.shrink_hard = shrink_hard; }
    a.in_bstride = vol_in; a.out_bstride = vol;
    if (dil > 1) a.in_bstride = a.out_bstride = n1;
    if (!inverse) {
        a.in[0] = in;
        for (int b = 0; b < 4; ++b) a.out[b] = out + b * vol;
    } else {
        for (int b = 0; b < 4; ++b) a.in[b] = in + b * vol_in;
        a.out[0] = out;
    }
#define CASE2C(LL)                                                                                                            \
    case LL:                                                                                                                  \
        if (inverse) {                                                                                                        \
            EMU2_GEOM(a, ndwt::Inv2S<T, LL, true, 4, 2>::WX, Lp, 64, ychunk);                                      \
            return vec4 ? run2<ndwt::Inv2S<T, LL, true, 4, 2>, T>(a, lo, hi) : run2<ndwt::Inv2S<T, LL, false, 4, 2>, T>(a, lo, hi); \
        } else {                                                                                                              \
            EMU2_GEOM(a, ndwt::Fwd2S<T, LL, true, 4, 2>::WX, Lp, 64, ychunk);                                      \
            return vec4 ? run2<ndwt::Fwd2S<T, LL, true, 4, 2>, T>(a, lo, hi) : run2<ndwt::Fwd2S<T, LL, false, 4, 2>, T>(a, lo, hi); \
        }
    if (ew == 2) {
        switch (Lp) {
            CASE2C(2) CASE2C(4) CASE2C(8)
            default: return -1;
        }
    }
#undef CASE2C
#define CASE2(LL)                                                                                                       \
    case LL:                                                                                                            \
        if (inverse) {                                                                                                  \
            EMU2_GEOM(a, ndwt::Inv2S<T, LL, true>::WX, Lp, 64, ychunk);                                      \
            return vec4 ? run2<ndwt::Inv2S<T, LL, true>, T>(a, lo, hi) : run2<ndwt::Inv2S<T, LL, false>, T>(a, lo, hi); \
        } else {                                                                                                        \
            EMU2_GEOM(a, ndwt::Fwd2S<T, LL, true>::WX, Lp, 64, ychunk);                                      \
            return vec4 ? run2<ndwt::Fwd2S<T, LL, true>, T>(a, lo, hi) : run2<ndwt::Fwd2S<T, LL, false>, T>(a, lo, hi); \
        }
    switch (Lp) {
        CASE2(2) CASE2(4) CASE2(6) CASE2(8) CASE2(10) CASE2(12)
        default: return -1;
    }
#undef CASE2
}

template <typename T, int L, bool SYN>
int run_march(ndwt::MarchArgs<T>& a, const double* lo, const double* hi) {
    typedef ndwt::AxisMarch<T, L, SYN> K;
    typename K::Taps tp;
    for (int j = 0; j < L; ++j) { tp.lo[j] = (T)lo[j]; tp.hi[j] = (T)hi[j]; }
    const long long iblocks = (a.ngroups * a.outer + K::NT - 1) / K::NT;
    const long long nblocks = iblocks * a.nchunks;
    for (long long b = 0; b < nblocks; ++b) {
        typename K::Shared sh;
        EmuExec<typename K::State, K::NT> ex;
        K::block(ex, sh, a, tp, (int)b);
    }
    return 0;
}

template <typename T>
int emu_march(int syn, int L, const T* in0, const T* in1, T* out0, T* out1, long long inner, long long n, long long outer, int chunk,
              int wrap, const double* lo, const double* hi) {
    ndwt::MarchArgs<T> a;
    std::memset(&a, 0, sizeof(a));
    a.in0 = in0; a.in1 = in1; a.out0 = out0; a.out1 = out1;
    a.inner = inner; a.n = n; a.n_in = wrap ? n : n + L - 1; a.outer = outer; a.wrap = wrap;
    a.ngroups = inner / 4;
    a.chunk = chunk > 0 && chunk < n ? chunk : (int)n;
    a.nchunks = (int)((n + a.chunk - 1) / a.chunk);
#define CASEM(LL) case LL: return syn ? run_march<T, LL, true>(a, lo, hi) : run_march<T, LL, false>(a, lo, hi);
    switch (L) {
        CASEM(2) CASEM(4) CASEM(8) CASEM(12) CASEM(20)
        default: return -1;
    }
#undef CASEM
}

template <typename T, int L, bool SYN, int EW>
int run_axisx(ndwt::AxisXArgs<T> a, int vec4, const double* lo, const double* hi) {
    typedef ndwt::AxisX<T, L, SYN, EW, true> K;
    ndwt::MarchTaps<T, L> tp;
    for (int j = 0; j < L; ++j) { tp.lo[j] = (T)lo[j]; tp.hi[j] = (T)hi[j]; }
    a.nseg = (a.row + K::WX - 1) / K::WX;
    const long long nblocks = (a.outer * a.nseg + 3) / 4;
    for (long long b = 0; b < nblocks; ++b) {
        if (vec4) {
            typename K::Shared sh;
            EmuExec<typename K::State, K::NT> ex;
            K::block(ex, sh, a, tp, (int)b);
        } else {
            typedef ndwt::AxisX<T, L, SYN, EW, false> K0;
            typename K0::Shared sh;
            EmuExec<typename K0::State, K0::NT> ex;
            K0::block(ex, sh, a, tp, (int)b);
        }
    }
    return 0;
}

template <typename T>
int emu_axisx(int syn, int L, int ew, int vec4, const T* in0, const T* in1, T* out0, T* out1, long long row, long long outer,
              const double* lo, const double* hi) {
    ndwt::AxisXArgs<T> a;
    std::memset(&a, 0, sizeof(a));
    a.in0 = in0; a.in1 = in1; a.out0 = out0; a.out1 = out1; a.row = row; a.outer = outer;
#define CASEX(LL)                                                                                                        \
    case LL:                                                                                                             \
        if (ew == 1) return syn ? run_axisx<T, LL, true, 1>(a, vec4, lo, hi) : run_axisx<T, LL, false, 1>(a, vec4, lo, hi); \
        return syn ? run_axisx<T, LL, true, 2>(a, vec4, lo, hi) : run_axisx<T, LL, false, 2>(a, vec4, lo, hi);
    switch (L) {
        CASEX(2) CASEX(8) CASEX(12)
        default: return -1;
    }
#undef CASEX
}

}  // namespace

// Each EMU_PART is one translation unit of libndwt_emu*.so (parallel build); without EMU_PART everything is compiled.
#ifndef EMU_PART
#define EMU_ALL 1
#else
#define EMU_ALL 0
#endif
#define EMU_IN(part) (EMU_ALL || EMU_PART == part)

#if EMU_IN(9)
int emu_y_small(int depth, int Lp, int vec4, ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    return depth == 1 ? dispatchY<float, 16, 8, 128, true, 1>(Lp, vec4, a, lo, hi) : -2;
}
#endif
#if EMU_IN(10)
int emu_y_small2(int Lp, int vec4, ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    return dispatchY<float, 16, 8, 512, true, 2>(Lp, vec4, a, lo, hi);
}
#endif
#if EMU_IN(11)
int emu_y_prod(int depth, int Lp, int vec4, ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    // the same taps on y and z, two register sets, rows of whole groups of 4: the instance whose z stage reads the y tap pairs (UNIYZ), as the library
    bool uni = depth == 2 && vec4;
    for (int j = 0; j < Lp; ++j) uni = uni && lo[1 * ndwt::kMaxTaps + j] == lo[2 * ndwt::kMaxTaps + j];
    if (uni && Lp == 12) return runY<ndwt::Inv3Y<float, 12, ndwt::kInv3YTX, ndwt::kInv3YTY, 1024, true, 2, 2, 1, ndwt::inv3y_zlds(12, 2), 0, true>, float>(a, lo, hi);
    return depth == 1 ? dispatchY<float, ndwt::kInv3YTX, ndwt::kInv3YTY, 1024, false, 1>(Lp, vec4, a, lo, hi)
                      : dispatchY<float, ndwt::kInv3YTX, ndwt::kInv3YTY, 1024, false, 2>(Lp, vec4, a, lo, hi);
}
#endif
#if EMU_IN(17)
// Inv3Y with the x stage in scatter form: the small test tile for every tap length (partial sums that travel 1, 2 and 3 lanes), the
// production instances of the library (ndwt_fused3_f32_invys.hip) for 8 / 12 / 20 taps
template <int LL, int TX, int TY, int NT, int DEPTH, int ZLV, bool UNI> static int run_ys(ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    return runY<ndwt::Inv3Y<float, LL, TX, TY, NT, true, 2, DEPTH, 1, ZLV, 0, UNI, true>, float>(a, lo, hi);
}
int emu_y_scatter(int small, int Lp, ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    bool uni = true;
    for (int j = 0; j < Lp; ++j) uni = uni && lo[1 * ndwt::kMaxTaps + j] == lo[2 * ndwt::kMaxTaps + j];
    if (small) {
#define CASES(LL) case LL: return run_ys<LL, 16, 8, 512, 2, (LL >= 4 ? LL / 2 : 0), false>(a, lo, hi);
        switch (Lp) {
            CASES(2) CASES(4) CASES(6) CASES(8) CASES(10) CASES(12) CASES(14) CASES(16) CASES(18) CASES(20)
            default: return -1;
        }
#undef CASES
    }
    switch (Lp) {
        case 8: return run_ys<8, ndwt::inv3y_tx(8), ndwt::inv3y_ty(8), 1024, 2, ndwt::inv3y_zlds(8, 2), false>(a, lo, hi);
        case 12: return uni ? run_ys<12, ndwt::inv3y_tx(12), ndwt::inv3y_ty(12), 1024, 2, ndwt::inv3y_zlds(12, 2), true>(a, lo, hi)
                            : run_ys<12, ndwt::inv3y_tx(12), ndwt::inv3y_ty(12), 1024, 2, ndwt::inv3y_zlds(12, 2), false>(a, lo, hi);
        case 20: return run_ys<20, ndwt::inv3y_tx(20), ndwt::inv3y_ty(20), 1024, 1, ndwt::inv3y_zlds(20, 1), true>(a, lo, hi);
        default: return -1;
    }
}
#endif
#if EMU_IN(17)
// a level dilated by 4 through Inv3Y<.., EW = 4, XSC>: the library's instances (8 taps: two register sets)
template <int LL, int D> static int run_y4s(ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    return runY<ndwt::Inv3Y<float, LL, ndwt::inv3y_tx(LL, 4), ndwt::inv3y_ty(LL, 4), 1024, true, 2, D, 4, 0, 0, false, true>, float>(a, lo, hi);
}
int emu_y_dilated4s(int Lp, ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    switch (Lp) {
        case 4: return run_y4s<4, 1>(a, lo, hi);
        case 6: return run_y4s<6, 1>(a, lo, hi);
        case 8: return run_y4s<8, 2>(a, lo, hi);
        default: return -1;
    }
}
#endif
#if EMU_IN(18)
// Inv3Y on (re, im) pairs with the x stage in scatter form (EW = 2, XSC)
template <int LL, int TX, int TY, int NT, int D> static int run_ycs(ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    return runY<ndwt::Inv3Y<float, LL, TX, TY, NT, true, 2, D, 2, 0, 0, false, true>, float>(a, lo, hi);
}
int emu_yc_scatter(int small, int Lp, ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    if (small) {
#define CASES(LL) case LL: return run_ycs<LL, 16, 8, 512, 2>(a, lo, hi);
        switch (Lp) {
            CASES(2) CASES(4) CASES(6) CASES(8) CASES(10) CASES(12) CASES(14) CASES(16)
            default: return -1;
        }
#undef CASES
    }
#define CASEP(LL) case LL: return run_ycs<LL, ndwt::inv3y_tx(LL, 2), ndwt::inv3y_ty(LL, 2), 1024, 1>(a, lo, hi);
    switch (Lp) {
        CASEP(10) CASEP(12) CASEP(16)
        default: return -1;
    }
#undef CASEP
}
#endif
#if EMU_IN(12)
int emu_yc_small(int depth, int Lp, int vec4, ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    return depth == 1 ? dispatchY<float, 16, 8, 128, true, 1, 2>(Lp, vec4, a, lo, hi)
                      : dispatchY<float, 16, 8, 512, true, 2, 2>(Lp, vec4, a, lo, hi);
}
#endif

#if EMU_IN(15)
// dilated levels through the pair-packed synthesis kernel: EW = 2 (its interleaved-pair form) and EW = 4 (whole-lane x shifts), production tiles
template <int LL, int EWV, int D> static int run_yd(ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    return runY<ndwt::Inv3Y<float, LL, ndwt::inv3y_tx(LL, EWV), ndwt::inv3y_ty(LL, EWV), 1024, true, 2, D, EWV>, float>(a, lo, hi);
}
int emu_y_dilated(int ew, int depth, int Lp, ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    if (ew == 2) {
        switch (Lp) {
            case 4: return run_yd<4, 2, 1>(a, lo, hi);
            case 8: return depth == 2 ? run_yd<8, 2, 2>(a, lo, hi) : run_yd<8, 2, 1>(a, lo, hi);
            default: return -1;
        }
    }
    switch (Lp) {
        case 2: return depth == 2 ? run_yd<2, 4, 2>(a, lo, hi) : run_yd<2, 4, 1>(a, lo, hi);
        case 4: return run_yd<4, 4, 1>(a, lo, hi);
        case 6: return run_yd<6, 4, 1>(a, lo, hi);
        case 8: return depth == 2 ? run_yd<8, 4, 2>(a, lo, hi) : run_yd<8, 4, 1>(a, lo, hi);
        default: return -1;
    }
}
#endif

#if EMU_IN(13)
// level 1 of a denoising step in one launch (Den3) and the approximation-only analysis (Fwd3<.., LOWONLY>), production tile
template <int LL>
static int run_den3(ndwt::Fused3Args<float>& a, const double* slo, const double* shi, const double* alo, const double* ahi) {
    typedef ndwt::Den3<float, LL, 1024, 4, (LL == 8 ? 6 : 0)> K;
    std::unique_ptr<typename K::Taps> tp(new typename K::Taps);
    for (int ax = 0; ax < 3; ++ax)
        for (int j = 0; j < LL; ++j) {
            tp->syn.lo[ax][j] = (float)slo[ax * ndwt::kMaxTaps + j];
            tp->syn.hi[ax][j] = (float)shi[ax * ndwt::kMaxTaps + j];
            tp->alo[ax][j] = (float)alo[ax * ndwt::kMaxTaps + j];
        }
    for (int k = 0; k <= LL; ++k)
        for (int h = 0; h < 2; ++h) {
            const int j = k - h;
            tp->syn.xplo[k][h] = (j >= 0 && j < LL) ? (float)slo[j] : 0.0f;
            tp->syn.xphi[k][h] = (j >= 0 && j < LL) ? (float)shi[j] : 0.0f;
        }
    for (int j = 0; j < LL; ++j) {
        tp->azp[j][0] = (float)alo[2 * ndwt::kMaxTaps + j];
        tp->azp[j][1] = (float)ahi[2 * ndwt::kMaxTaps + j];
    }
    for (int k = 0; k <= LL; ++k)
        for (int h = 0; h < 2; ++h) {
            const int j = k - h;
            tp->axp[k][h] = (j >= 0 && j < LL) ? (float)alo[j] : 0.0f;
        }
    const int nblocks = a.ntx * a.nty * a.nzc * a.nbatch;
    for (int b = 0; b < nblocks; ++b) {
        std::unique_ptr<typename K::Shared> sh(new typename K::Shared);
        EmuExec<typename K::State, K::NT> ex;
        K::block(ex, *sh, a, *tp, b);
    }
    return 0;
}
template <int LL, bool V> static int run_low3(ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    typedef ndwt::Fused3Tile<float, false, 2> TL;
    return run<ndwt::Fwd3<float, LL, TL::TX, TL::TY, TL::NT, TL::RY, V, 2, 1, true>, float>(a, lo, hi);
}
extern "C" int ndwt_emu_den3_f32(int Lp, const float* x, const float* apx, float* out, int n1, int n2, int n3, int zchunk,
                                 const double* slo, const double* shi, const double* alo, const double* ahi, double thr, int hard) {
    ndwt::Fused3Args<float> a;
    std::memset(&a, 0, sizeof(a));
    a.n1 = n1; a.n2 = n2; a.n3 = n3; a.nbatch = 1; a.z_wrap = 1;
    a.in[0] = x; a.in[1] = apx; a.out[0] = out;
    a.shrink_thr = (float)thr; a.shrink_mask = 0xFE; a.shrink_hard = hard;
    ndwt::fused3_geometry(a, 64, 32, 2 * Lp - 1, 4, zchunk);
    switch (Lp) {
        case 2: return run_den3<2>(a, slo, shi, alo, ahi);
        case 4: return run_den3<4>(a, slo, shi, alo, ahi);
        case 6: return run_den3<6>(a, slo, shi, alo, ahi);
        case 8: return run_den3<8>(a, slo, shi, alo, ahi);
        default: return -1;
    }
}
// one 4-D analysis level with the t axis folded into the fused launches (Fwd3<.., TPRE>): x (n4 frames) -> 16 bands
template <int LL> static int run_tpre(ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    typedef ndwt::Fused3Tile<float, false, 6> TL;
    return run<ndwt::Fwd3<float, LL, TL::TX, TL::TY, TL::NT, TL::RY, true, 2, 1, false, true>, float>(a, lo, hi);
}
extern "C" int ndwt_emu_tpre_f32(int Lp, const float* x, float* out, int n1, int n2, int n3, int n4, int zchunk, const double* alo,
                                 const double* ahi, const double* tlo, const double* thi) {
    const long long vol3 = (long long)n1 * n2 * n3;
    for (int tb = 0; tb < 2; ++tb) {
        ndwt::Fused3Args<float> a;
        std::memset(&a, 0, sizeof(a));
        a.n1 = n1; a.n2 = n2; a.n3 = n3; a.nbatch = n4; a.z_wrap = 1;
        a.in[0] = x;
        a.in_bstride = a.out_bstride = vol3;
        for (int b = 0; b < 8; ++b) a.out[b] = out + (long long)(8 * tb + b) * vol3 * n4;
        for (int j = 0; j < Lp; ++j) a.tt[j] = (float)(tb ? thi[j] : tlo[j]);
        ndwt::fused3_geometry(a, 64, 32, Lp, 4, zchunk);
        int rc = -1;
        switch (Lp) {
            case 2: rc = run_tpre<2>(a, alo, ahi); break;
            case 4: rc = run_tpre<4>(a, alo, ahi); break;
            case 6: rc = run_tpre<6>(a, alo, ahi); break;
            case 8: rc = run_tpre<8>(a, alo, ahi); break;
        }
        if (rc) return rc;
    }
    return 0;
}
// 2-D float synthesis with PD rows of band loads in flight per wave, row loop unrolled in groups of L (Inv2P)
template <int LL, int PD, bool PK = false> static int run_inv2p(ndwt::Fused2Args<float>& a, const double* lo, const double* hi, int ychunk) {
    typedef ndwt::Inv2P<float, LL, PD, 2, PK> K;
    ndwt::fused2_geometry(a, K::WX, LL, 64, ychunk);
    return run2<K, float>(a, lo, hi);
}
extern "C" int ndwt_emu2_inv2p_f32(int Lp, int depth, const float* in, float* out, int n1, int n2, int ychunk, const double* lo, const double* hi,
                                   double shrink_thr, int shrink_mask, int shrink_hard) {
    ndwt::Fused2Args<float> a;
    std::memset(&a, 0, sizeof(a));
    a.n1 = n1; a.n2 = n2; a.nbatch = 1; a.y_wrap = 1;
    a.shrink_thr = (float)shrink_thr; a.shrink_mask = shrink_mask; a.shrink_hard = shrink_hard;
    const long long vol = (long long)n1 * n2;
    a.in_bstride = a.out_bstride = vol;
    for (int b = 0; b < 4; ++b) a.in[b] = in + b * vol;
    a.out[0] = out;
    if (depth == 14) switch (Lp) {   // packed form, 4 rows in flight
        case 4: return run_inv2p<4, 4, true>(a, lo, hi, ychunk);
        case 8: return run_inv2p<8, 4, true>(a, lo, hi, ychunk);
        case 12: return run_inv2p<12, 4, true>(a, lo, hi, ychunk);
        default: return -1;
    }
    switch (Lp) {
        case 2: return run_inv2p<2, 2>(a, lo, hi, ychunk);
        case 4: return depth == 4 ? run_inv2p<4, 4>(a, lo, hi, ychunk) : run_inv2p<4, 2>(a, lo, hi, ychunk);
        case 6: return run_inv2p<6, 2>(a, lo, hi, ychunk);
        case 8: return depth == 4 ? run_inv2p<8, 4>(a, lo, hi, ychunk) : run_inv2p<8, 2>(a, lo, hi, ychunk);
        case 12: return depth == 4 ? run_inv2p<12, 4>(a, lo, hi, ychunk) : run_inv2p<12, 2>(a, lo, hi, ychunk);
        default: return -1;
    }
}
extern "C" int ndwt_emu_low3_f32(int Lp, int vec4, const float* x, float* out, int n1, int n2, int n3, int zchunk, const double* alo,
                                 const double* ahi) {
    ndwt::Fused3Args<float> a;
    std::memset(&a, 0, sizeof(a));
    a.n1 = n1; a.n2 = n2; a.n3 = n3; a.nbatch = 1; a.z_wrap = 1;
    a.in[0] = x; a.out[0] = out;
    ndwt::fused3_geometry(a, 64, 32, Lp, 4, zchunk);
    switch (Lp) {
        case 2: return vec4 ? run_low3<2, true>(a, alo, ahi) : run_low3<2, false>(a, alo, ahi);
        case 4: return vec4 ? run_low3<4, true>(a, alo, ahi) : run_low3<4, false>(a, alo, ahi);
        case 6: return vec4 ? run_low3<6, true>(a, alo, ahi) : run_low3<6, false>(a, alo, ahi);
        case 8: return vec4 ? run_low3<8, true>(a, alo, ahi) : run_low3<8, false>(a, alo, ahi);
        default: return -1;
    }
}
#endif

#if EMU_IN(14)
// the analysis kernel with its taps pinned in SGPRs and the high-pass taps derived from the low-pass ones (Fwd3<.., PIN>), production tile
template <int LL> static int run_pin3(ndwt::Fused3Args<float>& a, const double* lo, const double* hi) {
    typedef ndwt::Fused3Tile<float, false, 6> TL;
    return run<ndwt::Fwd3<float, LL, TL::TX, TL::TY, TL::NT, TL::RY, true, 2, 1, false, false, true>, float>(a, lo, hi);
}
extern "C" int ndwt_emu_pin3_f32(int Lp, const float* x, float* out, int n1, int n2, int n3, int zchunk, const double* alo, const double* ahi) {
    ndwt::Fused3Args<float> a;
    std::memset(&a, 0, sizeof(a));
    a.n1 = n1; a.n2 = n2; a.n3 = n3; a.nbatch = 1; a.z_wrap = 1;
    a.in[0] = x;
    for (int b = 0; b < 8; ++b) a.out[b] = out + (long long)b * n1 * n2 * n3;
    ndwt::fused3_geometry(a, 64, 32, Lp, 4, zchunk);
    switch (Lp) {
        case 10: return run_pin3<10>(a, alo, ahi);
        case 12: return run_pin3<12>(a, alo, ahi);
        case 14: return run_pin3<14>(a, alo, ahi);
        case 16: {   // 16 taps: plain taps, 2 of the 16 z-window slots in LDS (Fwd3 WLDS)
            typedef ndwt::Fused3Tile<float, false, 6> TL;
            return run<ndwt::Fwd3<float, 16, TL::TX, TL::TY, TL::NT, TL::RY, true, 2, 1, false, false, false, 2>, float>(a, alo, ahi);
        }
        case 20: {   // 20 taps: the 512-thread 64x16 tile, two columns per thread, 4 of the 20 window slots of each in LDS
            typedef ndwt::Fused3Tile<float, false, 1> TL;
            ndwt::fused3_geometry(a, TL::TX, TL::TY, Lp, 4, zchunk);
            return run<ndwt::Fwd3<float, 20, TL::TX, TL::TY, TL::NT, TL::RY, true, 2, 1, false, false, false, 4>, float>(a, alo, ahi);
        }
        default: return -1;
    }
}
#endif

#if EMU_IN(16)
// cascaded 2-D synthesis (Inv2C): in = 1 + 3 nlev bands in the reference's order
template <int LL, int NLEV, int PD> static int run_inv2c(ndwt::Fused2CIArgs<float>& a, const double* lo, const double* hi, int ychunk) {
    typedef ndwt::Inv2C<float, LL, NLEV, PD, 2> K;
    a.ntx = (a.n1 + K::WX - 1) / K::WX;
    a.ychunk = ychunk > 0 ? (ychunk < a.n2 ? ychunk : a.n2) : a.n2;
    a.nyc = (a.n2 + a.ychunk - 1) / a.ychunk;
    typename K::Taps tp;
    for (int ax = 0; ax < 3; ++ax)
        for (int j = 0; j < LL; ++j) {
            tp.lo[ax][j] = (float)lo[ax * ndwt::kMaxTaps + j];
            tp.hi[ax][j] = (float)hi[ax * ndwt::kMaxTaps + j];
        }
    fill_x_pairs(tp, lo, hi, LL, 0);
    for (int b = 0; b < a.ntx * a.nyc; ++b) {
        typename K::Shared sh;
        EmuExec<typename K::State, K::NT> ex;
        K::block(ex, sh, a, tp, b);
    }
    return 0;
}
// cascaded 2-D analysis (Fwd2C): out = 1 + 3 nlev bands in the reference's order for an nlev-level transform
template <int LL, int NLEV> static int run_fwd2c(ndwt::Fused2CArgs<float>& a, const double* lo, const double* hi, int ychunk) {
    typedef ndwt::Fwd2C<float, LL, NLEV, 2> K;
    a.ntx = (a.n1 + K::WX - 1) / K::WX;
    a.ychunk = ychunk > 0 ? (ychunk < a.n2 ? ychunk : a.n2) : a.n2;
    a.nyc = (a.n2 + a.ychunk - 1) / a.ychunk;
    typename K::Taps tp;
    for (int ax = 0; ax < 3; ++ax)
        for (int j = 0; j < LL; ++j) {
            tp.lo[ax][j] = (float)lo[ax * ndwt::kMaxTaps + j];
            tp.hi[ax][j] = (float)hi[ax * ndwt::kMaxTaps + j];
        }
    for (int b = 0; b < a.ntx * a.nyc; ++b) {
        typename K::Shared sh;
        EmuExec<typename K::State, K::NT> ex;
        K::block(ex, sh, a, tp, b);
    }
    return 0;
}
#endif
extern "C" {
#if EMU_IN(1)
int ndwt_emu_axisx_f32(int syn, int L, int ew, int vec4, const float* in0, const float* in1, float* out0, float* out1, long long row,
                       long long outer, const double* lo, const double* hi) {
    return emu_axisx<float>(syn, L, ew, vec4, in0, in1, out0, out1, row, outer, lo, hi);
}
int ndwt_emu_axisx_f64(int syn, int L, int ew, int vec4, const double* in0, const double* in1, double* out0, double* out1, long long row,
                       long long outer, const double* lo, const double* hi) {
    return emu_axisx<double>(syn, L, ew, vec4, in0, in1, out0, out1, row, outer, lo, hi);
}
#endif
#if EMU_IN(2)
int ndwt_emu_march_f32(int syn, int L, const float* in0, const float* in1, float* out0, float* out1, long long inner, long long n,
                       long long outer, int chunk, int wrap, const double* lo, const double* hi) {
    return emu_march<float>(syn, L, in0, in1, out0, out1, inner, n, outer, chunk, wrap, lo, hi);
}
int ndwt_emu_march_f64(int syn, int L, const double* in0, const double* in1, double* out0, double* out1, long long inner, long long n,
                       long long outer, int chunk, int wrap, const double* lo, const double* hi) {
    return emu_march<double>(syn, L, in0, in1, out0, out1, inner, n, outer, chunk, wrap, lo, hi);
}
#endif
#if EMU_IN(3)
int ndwt_emu2_f32(int inverse, int Lp, int vec4, const float* in, float* out, int n1, int n2, int ychunk, const double* lo,
                  const double* hi, int y_wrap, int ew, double shrink_thr, int shrink_mask, int shrink_hard, int dil) {
    return emu2<float>(inverse, Lp, vec4, in, out, n1, n2, ychunk, lo, hi, y_wrap, ew, shrink_thr, shrink_mask, shrink_hard, dil);
}
#endif
#if EMU_IN(16)
int ndwt_emu2_cascade_inv_f32(int Lp, int nlev, int depth, const float* in, float* out, int n1, int n2, int ychunk, const double* lo, const double* hi) {
    ndwt::Fused2CIArgs<float> a;
    std::memset(&a, 0, sizeof(a));
    a.out = out; a.n1 = n1; a.n2 = n2; a.rs = n1;
    for (int b = 0; b < 1 + 3 * nlev; ++b) a.in[b] = in + (long long)b * n1 * n2;
#define CASEC(LL) case LL: return nlev == 3 ? (depth == 2 ? run_inv2c<LL, 3, 2>(a, lo, hi, ychunk) : run_inv2c<LL, 3, 1>(a, lo, hi, ychunk)) \
                                            : (depth == 2 ? run_inv2c<LL, 2, 2>(a, lo, hi, ychunk) : run_inv2c<LL, 2, 1>(a, lo, hi, ychunk));
    if (nlev != 2 && nlev != 3) return -1;
    switch (Lp) {
        CASEC(2) CASEC(4) CASEC(6) CASEC(8)
        default: return -1;
    }
#undef CASEC
}
int ndwt_emu2_cascade_f32(int Lp, int nlev, const float* in, float* out, int n1, int n2, int ychunk, const double* lo, const double* hi) {
    ndwt::Fused2CArgs<float> a;
    std::memset(&a, 0, sizeof(a));
    a.in = in; a.n1 = n1; a.n2 = n2; a.rs = n1;
    for (int b = 0; b < 1 + 3 * nlev; ++b) a.out[b] = out + (long long)b * n1 * n2;
#define CASEC(LL) case LL: return nlev == 3 ? run_fwd2c<LL, 3>(a, lo, hi, ychunk) : run_fwd2c<LL, 2>(a, lo, hi, ychunk);
    if (nlev != 2 && nlev != 3) return -1;
    switch (Lp) {
        CASEC(2) CASEC(4) CASEC(6) CASEC(8)
        case 12: return nlev == 2 ? run_fwd2c<12, 2>(a, lo, hi, ychunk) : -1;
        default: return -1;
    }
#undef CASEC
}
#endif
#if EMU_IN(4)
int ndwt_emu2_f64(int inverse, int Lp, int vec4, const double* in, double* out, int n1, int n2, int ychunk, const double* lo,
                  const double* hi, int y_wrap, int ew, double shrink_thr, int shrink_mask, int shrink_hard, int dil) {
    return emu2<double>(inverse, Lp, vec4, in, out, n1, n2, ychunk, lo, hi, y_wrap, ew, shrink_thr, shrink_mask, shrink_hard, dil);
}
#endif
// in/out: band-planar, batch inside band: [band][batch][n3(+halo)][n2][n1]; lo/hi: [3][20] padded kernel-form taps
#define EMU3_ARGS(T) int Lp, int vec4, const T* in, T* out, int n1, int n2, int n3, int nbatch, int zchunk, const double* lo, \
                     const double* hi, int z_wrap, int small_tile, int variant, int ew, double shrink_thr, int shrink_mask, \
                     int shrink_hard, int dil
#define EMU3_PASS Lp, vec4, in, out, n1, n2, n3, nbatch, zchunk, lo, hi, z_wrap, small_tile, variant, ew, shrink_thr, shrink_mask, shrink_hard, dil
#if EMU_IN(5)
int ndwt_emu3_f32_fwd(EMU3_ARGS(float)) { return emu3<float, false>(EMU3_PASS); }
#endif
#if EMU_IN(6)
int ndwt_emu3_f32_inv(EMU3_ARGS(float)) { return emu3<float, true>(EMU3_PASS); }
#endif
#if EMU_IN(7)
int ndwt_emu3_f64_fwd(EMU3_ARGS(double)) { return emu3<double, false>(EMU3_PASS); }
#endif
#if EMU_IN(8)
int ndwt_emu3_f64_inv(EMU3_ARGS(double)) { return emu3<double, true>(EMU3_PASS); }
#endif
#if EMU_IN(1)
int ndwt_emu3_f32_fwd(EMU3_ARGS(float));
int ndwt_emu3_f32_inv(EMU3_ARGS(float));
int ndwt_emu3_f64_fwd(EMU3_ARGS(double));
int ndwt_emu3_f64_inv(EMU3_ARGS(double));
int ndwt_emu3_f32(int inverse, EMU3_ARGS(float)) { return inverse ? ndwt_emu3_f32_inv(EMU3_PASS) : ndwt_emu3_f32_fwd(EMU3_PASS); }
int ndwt_emu3_f64(int inverse, EMU3_ARGS(double)) { return inverse ? ndwt_emu3_f64_inv(EMU3_PASS) : ndwt_emu3_f64_fwd(EMU3_PASS); }
#endif
}
