// ndwt_fused_kernels.h -- __global__ wrappers + launch switch for the fused kernels (HIP only).
#pragma once
#include "ndwt_fused.h"

namespace ndwt {

template <class State> struct GpuExec {
    typedef State state_type;
    State st;
    template <class F> __device__ __forceinline__ void each(F&& f) { f((int)threadIdx.x, st); }   // (callers pass always-inline lambdas)
    __device__ __forceinline__ void barrier() { __syncthreads(); }
};

// amdgpu_waves_per_eu pins the register budget: without it hipcc aims at the LDS-limited occupancy and spills the
// filter windows to scratch (512-thread workgroups: 65 VGPRs + 176 B/lane of scratch).
// The taps live in a small device buffer owned by the plan and are read through the CONSTANT address space, so
// every tap is a scalar load into SGPRs (a by-value struct argument ends up partly in scratch once the stage
// functions nest a few lambdas deep).
template <class K>
__global__ __launch_bounds__(K::NT) __attribute__((amdgpu_waves_per_eu(K::WPE, K::WPE))) void fused3_kernel(
    const typename K::Args a, const typename K::Taps* __restrict__ taps_global) {
    __shared__ typename K::Shared sh;
    GpuExec<typename K::State> ex;
    typedef const __attribute__((address_space(4))) typename K::Taps* ctaps_ptr;
    const typename K::Taps& tp = *(const typename K::Taps*)(ctaps_ptr)taps_global;
    K::block(ex, sh, a, tp, (int)blockIdx.x);
}

// taps_dev: device buffer holding Taps3<T, Lp> (lo[3][Lp] then hi[3][Lp]) for this direction
template <class K> int launch_fused3(const typename K::Args& a, const FusedTapsD& t, const void* taps_dev, hipStream_t s) {
    (void)t;
    // the host computed the tiling for a tile shape (fused3_tile_shape); this kernel was compiled for one: they must be the same,
    // or workgroups would run off their tiles -- a status code here instead of a memory fault there
    if (a.ntx != (a.n1 + K::TX - 1) / K::TX || a.nty != (a.n2 + K::TY - 1) / K::TY || a.zchunk < 1 ||
        (long long)a.nzc * a.zchunk < a.n3 || (long long)(a.nzc - 1) * a.zchunk >= a.n3)
        return -2;
    const int nblocks = a.ntx * a.nty * a.nzc * a.nbatch;
    hipLaunchKernelGGL(fused3_kernel<K>, dim3(nblocks), dim3(K::NT), 0, s, a, (const typename K::Taps*)taps_dev);
    return (int)hipGetLastError();
}

template <class K> int launch_fused2(const typename K::Args& a, const void* taps_dev, hipStream_t s) {
    const int nblocks = a.ntx * a.nyc * a.nbatch;
    hipLaunchKernelGGL(fused3_kernel<K>, dim3(nblocks), dim3(K::NT), 0, s, a, (const typename K::Taps*)taps_dev);
    return (int)hipGetLastError();
}

#define NDWT_FUSED2_CASE(KIND, T, LL)                                                        \
    case LL:                                                                                 \
        if constexpr (sizeof(T) == 4 && LL <= 8) {   /* a level dilated by 4: x taps over 4 scalars (float, db1..db4) */ \
            if (ew == 4)                                                                     \
                return vec4 ? launch_fused2<KIND<T, LL, true, 4, 4>>(a, taps_dev, s) : launch_fused2<KIND<T, LL, false, 4, 4>>(a, taps_dev, s); \
        }                                                                                    \
        if (ew == 4) return -1;                                                              \
        if constexpr (LL <= 8) {   /* interleaved complex: up to 8 taps here (complex64 10 .. 16: ndwt_fused2_f32_{fwdc,invc}.hip) */ \
            if (ew == 2)                                                                     \
                return vec4 ? launch_fused2<KIND<T, LL, true, (sizeof(T) == 8 ? 2 : 4), 2>>(a, taps_dev, s)   \
                            : launch_fused2<KIND<T, LL, false, (sizeof(T) == 8 ? 2 : 4), 2>>(a, taps_dev, s); \
        }                                                                                    \
        if (ew == 2) return -1;                                                              \
        return vec4 ? launch_fused2<KIND<T, LL, true, (sizeof(T) == 8 ? 2 : 4)>>(a, taps_dev, s) : launch_fused2<KIND<T, LL, false, (sizeof(T) == 8 ? 2 : 4)>>(a, taps_dev, s);
// (the tap lengths are split over two translation units per kernel family: one unit with all of them is the long pole of the build)
#define NDWT_FUSED2_SWITCH_SHORT(KIND, T) \
    switch (Lp) {                     \
        NDWT_FUSED2_CASE(KIND, T, 2)  \
        NDWT_FUSED2_CASE(KIND, T, 4)  \
        NDWT_FUSED2_CASE(KIND, T, 6)  \
        default: return -1;           \
    }
#define NDWT_FUSED2_SWITCH_LONG(KIND, T) \
    switch (Lp) {                     \
        NDWT_FUSED2_CASE(KIND, T, 8)  \
        NDWT_FUSED2_CASE(KIND, T, 10) \
        NDWT_FUSED2_CASE(KIND, T, 12) \
        default: return -1;           \
    }

// dispatch on variant, padded tap length and vector path
#define NDWT_FUSED_K(KIND, INV, T, LL, V, VEC)                                                               \
    KIND<T, LL, Fused3Tile<T, INV, V>::TX, Fused3Tile<T, INV, V>::TY, Fused3Tile<T, INV, V>::NT,             \
         Fused3Tile<T, INV, V>::RY, VEC, Fused3Tile<T, INV, V>::WPE>
// interleaved complex (2 scalars per x element)
#define NDWT_FUSED_KC(KIND, INV, T, LL, V, VEC)                                                              \
    KIND<T, LL, Fused3Tile<T, INV, V>::TX, Fused3Tile<T, INV, V>::TY, Fused3Tile<T, INV, V>::NT,             \
         Fused3Tile<T, INV, V>::RY, VEC, Fused3Tile<T, INV, V>::WPE, 2>
// x taps stepping over EWV interleaved scalars (EWV = 4: a level dilated by 4)
#define NDWT_FUSED_KE(KIND, INV, T, LL, V, VEC, EWV)                                                         \
    KIND<T, LL, Fused3Tile<T, INV, V>::TX, Fused3Tile<T, INV, V>::TY, Fused3Tile<T, INV, V>::NT,             \
         Fused3Tile<T, INV, V>::RY, VEC, Fused3Tile<T, INV, V>::WPE, EWV>
#define NDWT_FUSED_CASE_E(KIND, INV, T, LL, V, EWV)                                                          \
    case LL:                                                                                                 \
        return vec4 ? launch_fused3<NDWT_FUSED_KE(KIND, INV, T, LL, V, true, EWV)>(a, t, taps_dev, s)        \
                    : launch_fused3<NDWT_FUSED_KE(KIND, INV, T, LL, V, false, EWV)>(a, t, taps_dev, s);
#define NDWT_FUSED_CASE_C(KIND, INV, T, LL, V)                                                                \
    case LL:                                                                                                 \
        return vec4 ? launch_fused3<NDWT_FUSED_KC(KIND, INV, T, LL, V, true)>(a, t, taps_dev, s)             \
                    : launch_fused3<NDWT_FUSED_KC(KIND, INV, T, LL, V, false)>(a, t, taps_dev, s);

#define NDWT_FUSED_CASE(KIND, INV, T, LL, V)                                                                  \
    case LL:                                                                                                 \
        return vec4 ? launch_fused3<NDWT_FUSED_K(KIND, INV, T, LL, V, true)>(a, t, taps_dev, s)                        \
                    : launch_fused3<NDWT_FUSED_K(KIND, INV, T, LL, V, false)>(a, t, taps_dev, s);

// float synthesis other than the pair-packed kernel (ndwt_fused3_f32_invy*.hip, the default wherever it applies): the lane-shift
// kernel Inv3S on a tall 64x32 tile (1024 threads, one workgroup per CU; db6: 512 threads with two items each -- the 1024-thread
// form spills there) for mixed wavelets with odd tap padding, dilated levels and NDWT_VARIANT_INV=4; variant 3 = the LDS kernel
// (A/B, db4 only)
#define NDWT_FUSED_SWITCH_INV_F32_EW(T)                                      \
    if (ew == 4) {                                                        \
        switch (t.Lp) {                                                   \
            NDWT_FUSED_CASE_E(Inv3S, true, T, 2, 4, 4)                    \
            NDWT_FUSED_CASE_E(Inv3S, true, T, 4, 4, 4)                    \
            NDWT_FUSED_CASE_E(Inv3S, true, T, 6, 4, 4)                    \
            NDWT_FUSED_CASE_E(Inv3S, true, T, 8, 4, 4)                    \
            default: return -1;                                           \
        }                                                                 \
    }                                                                     \
    if (ew == 2) {                                                        \
        switch (t.Lp) {                                                   \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 2, 1)                       \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 4, 1)                       \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 6, 1)                       \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 8, 1)                       \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 10, 2)   /* complex db5 / db6: 512 threads x 2 items (db6 spills 42 of 256 registers) */ \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 12, 2)                      \
            default: return -1;                                           \
        }                                                                 \
    }
#define NDWT_FUSED_SWITCH_INV_F32_REAL(T)                                 \
    if (variant == 3 && t.Lp == 8) { switch (t.Lp) { NDWT_FUSED_CASE(Inv3, true, T, 8, 3) } }   \
    switch (t.Lp) {                                                       \
        NDWT_FUSED_CASE(Inv3S, true, T, 2, 1)                             \
        NDWT_FUSED_CASE(Inv3S, true, T, 4, 1)                             \
        NDWT_FUSED_CASE(Inv3S, true, T, 6, 1)                             \
        NDWT_FUSED_CASE(Inv3S, true, T, 8, 1)                             \
        NDWT_FUSED_CASE(Inv3S, true, T, 10, 1)                            \
        NDWT_FUSED_CASE(Inv3S, true, T, 12, 2)   /* db6: 512 threads x 2 items, no spills (1.78 vs 2.05 ms) */ \
        default: return -1;                                               \
    }

// double synthesis: the lane-shift kernel on a 64x16 tile with 512 threads; variant 3 = the LDS kernel (A/B runs, db4 only)
#define NDWT_FUSED_SWITCH_INV_F64(T)                                      \
    if (ew == 2) {                                                        \
        switch (t.Lp) {                                                   \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 2, 1)                       \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 4, 1)                       \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 6, 1)                       \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 8, 1)                       \
            NDWT_FUSED_CASE_C(Inv3S, true, T, 10, 5)   /* complex128 db5: 64x8 tile, 512 threads (6 spilled registers) */ \
            default: return -1;                                           \
        }                                                                 \
    }                                                                     \
    if (variant == 3 && t.Lp == 8) { switch (t.Lp) { NDWT_FUSED_CASE(Inv3, true, T, 8, 3) } }   \
    switch (t.Lp) {                                                       \
        NDWT_FUSED_CASE(Inv3S, true, T, 2, 1)                             \
        NDWT_FUSED_CASE(Inv3S, true, T, 4, 1)                             \
        NDWT_FUSED_CASE(Inv3S, true, T, 6, 1)                             \
        NDWT_FUSED_CASE(Inv3S, true, T, 8, 1)                             \
        NDWT_FUSED_CASE(Inv3S, true, T, 10, 5)   /* 64x8 tile, 512 threads: no spills (64x16: 32 / 71 spilled registers) */ \
        NDWT_FUSED_CASE(Inv3S, true, T, 12, 5)                            \
        default: return -1;                                               \
    }

// float analysis: 256-thread kernel for tap lengths <= 8, the tall 64x32 tile with 1024 threads for 10 and 12 (and 14, 16:
// ndwt_fused3_f32_long.hip); variant 1 = 512 threads, one column per thread (A/B; the kernel of interleaved complex data with 10 / 12 taps)
#define NDWT_FUSED_SWITCH_FWD_F32(T)                                      \
    if (ew == 4) {                                                        \
        switch (t.Lp) {                                                   \
            NDWT_FUSED_CASE_E(Fwd3, false, T, 2, 1, 4)                    \
            NDWT_FUSED_CASE_E(Fwd3, false, T, 4, 1, 4)                    \
            NDWT_FUSED_CASE_E(Fwd3, false, T, 6, 1, 4)                    \
            NDWT_FUSED_CASE_E(Fwd3, false, T, 8, 1, 4)                    \
            default: return -1;                                           \
        }                                                                 \
    }                                                                     \
    if (ew == 2 && variant == 2) {   /* interleaved complex on the tall tile */ \
        switch (t.Lp) {                                                   \
            NDWT_FUSED_CASE_C(Fwd3, false, T, 6, 2)                       \
            NDWT_FUSED_CASE_C(Fwd3, false, T, 8, 2)                       \
            default: break;                                               \
        }                                                                 \
    }                                                                     \
    if (ew == 2) {                                                        \
        switch (t.Lp) {                                                   \
            NDWT_FUSED_CASE_C(Fwd3, false, T, 2, 0)                       \
            NDWT_FUSED_CASE_C(Fwd3, false, T, 4, 0)                       \
            NDWT_FUSED_CASE_C(Fwd3, false, T, 6, 0)                       \
            NDWT_FUSED_CASE_C(Fwd3, false, T, 8, 0)                       \
            NDWT_FUSED_CASE_C(Fwd3, false, T, 10, 1)                      \
            NDWT_FUSED_CASE_C(Fwd3, false, T, 12, 1)                      \
            NDWT_FUSED_CASE_C(Fwd3, false, T, 14, 1)                      \
            NDWT_FUSED_CASE_C(Fwd3, false, T, 16, 1)                      \
            default: return -1;                                           \
        }                                                                 \
    }                                                                     \
    if (variant == 2) { switch (t.Lp) { NDWT_FUSED_CASE(Fwd3, false, T, 12, 2) NDWT_FUSED_CASE(Fwd3, false, T, 2, 2) NDWT_FUSED_CASE(Fwd3, false, T, 4, 2) NDWT_FUSED_CASE(Fwd3, false, T, 6, 2) NDWT_FUSED_CASE(Fwd3, false, T, 8, 2) } }  \
    if (variant == 1) { switch (t.Lp) { NDWT_FUSED_CASE(Fwd3, false, T, 10, 1) NDWT_FUSED_CASE(Fwd3, false, T, 12, 1) } }  \
    if (variant == 6) { switch (t.Lp) { NDWT_FUSED_CASE(Fwd3, false, T, 8, 6) NDWT_FUSED_CASE(Fwd3, false, T, 10, 6) } }  \
    switch (t.Lp) {                                                       \
        NDWT_FUSED_CASE(Fwd3, false, T, 2, 0)                             \
        NDWT_FUSED_CASE(Fwd3, false, T, 4, 0)                             \
        NDWT_FUSED_CASE(Fwd3, false, T, 6, 0)                             \
        NDWT_FUSED_CASE(Fwd3, false, T, 8, 0)                             \
        NDWT_FUSED_CASE(Fwd3, false, T, 10, 2)   /* 10 .. 16 taps: the tall tile (db6 analysis 1.33 -> 1.02 ms per launch) */ \
        NDWT_FUSED_CASE(Fwd3, false, T, 12, 6)   /* 12 .. 16 taps: y items of 2 rows (10 of the 16 waves in the y stage instead of 5: db6 -6 %) */ \
        default: return -1;                                               \
    }

}  // namespace ndwt
