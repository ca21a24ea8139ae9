// ndwt_fused_kernels.h -- __global__ wrappers + launch switch for the fused kernels (HIP only).
#pragma once
#include "ndwt_fused.h"

namespace ndwt {

template <class State> struct GpuExec {
    State st;
    template <class F> __device__ __forceinline__ void each(F&& f) { f((int)threadIdx.x, st); }
    __device__ __forceinline__ void barrier() { __syncthreads(); }
};

template <class K>
__global__ __launch_bounds__(K::NT) void fused3_kernel(const typename K::Args a, const typename K::Taps tp) {
    __shared__ typename K::Shared sh;
    GpuExec<typename K::State> ex;
    K::block(ex, sh, a, tp, (int)blockIdx.x);
}

template <class K> int launch_fused3(const typename K::Args& a, const FusedTapsD& t, hipStream_t s) {
    typename K::Taps tp;
    for (int ax = 0; ax < 3; ++ax)
        for (int j = 0; j < K::L; ++j) {
            tp.lo[ax][j] = (decltype(tp.lo[0][0] + 0))t.lo[ax][j];
            tp.hi[ax][j] = (decltype(tp.hi[0][0] + 0))t.hi[ax][j];
        }
    const int nblocks = a.ntx * a.nty * a.nzc * a.nbatch;
    hipLaunchKernelGGL(fused3_kernel<K>, dim3(nblocks), dim3(K::NT), 0, s, a, tp);
    return (int)hipGetLastError();
}

// NDWT_FUSED_SWITCH(KIND, T): dispatch on padded length and vector path
#define NDWT_FUSED_CASE(KIND, T, LL)                                                                         \
    case LL:                                                                                                 \
        return vec4 ? launch_fused3<KIND<T, LL, Fused3Tile<T>::TX, Fused3Tile<T>::TY, Fused3Tile<T>::NT,      \
                                         Fused3Tile<T>::RY, true>>(a, t, s)                                  \
                    : launch_fused3<KIND<T, LL, Fused3Tile<T>::TX, Fused3Tile<T>::TY, Fused3Tile<T>::NT,      \
                                         Fused3Tile<T>::RY, false>>(a, t, s);

#define NDWT_FUSED_SWITCH(KIND, T)        \
    switch (t.Lp) {                       \
        NDWT_FUSED_CASE(KIND, T, 2)       \
        NDWT_FUSED_CASE(KIND, T, 4)       \
        NDWT_FUSED_CASE(KIND, T, 6)       \
        NDWT_FUSED_CASE(KIND, T, 8)       \
        NDWT_FUSED_CASE(KIND, T, 10)      \
        NDWT_FUSED_CASE(KIND, T, 12)      \
        default: return -1;               \
    }

}  // namespace ndwt
