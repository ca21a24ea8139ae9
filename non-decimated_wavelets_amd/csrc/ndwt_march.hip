// register-window march kernels for one non-contiguous axis (ndwt_device.h: AxisMarch)
#include "ndwt_fused_kernels.h"

namespace ndwt {

template <class K>
__global__ __launch_bounds__(K::NT) void march_kernel(const typename K::Args a, const typename K::Taps tp) {
    typename K::Shared sh;
    GpuExec<typename K::State> ex;
    K::block(ex, sh, a, tp, (int)blockIdx.x);
}

template <typename T, int L, bool SYN>
static int launch_march_L(const MarchArgs<T>& a, const double* lo, const double* hi, hipStream_t s) {
    typedef AxisMarch<T, L, SYN> K;
    typename K::Taps tp;
    for (int j = 0; j < L; ++j) { tp.lo[j] = (T)lo[j]; tp.hi[j] = (T)hi[j]; }
    const long long iblocks = (a.ngroups * a.outer + K::NT - 1) / K::NT;
    const long long nblocks = iblocks * a.nchunks;
    if (nblocks <= 0 || nblocks > 0x7fffffffLL) return -2;
    hipLaunchKernelGGL(march_kernel<K>, dim3((unsigned)nblocks), dim3(K::NT), 0, s, a, tp);
    return (int)hipGetLastError();
}

#define NDWT_MARCH_CASE(LL) case LL: return syn ? launch_march_L<T, LL, true>(a, lo, hi, s) : launch_march_L<T, LL, false>(a, lo, hi, s);
template <typename T> static int launch_march_T(bool syn, int L, const MarchArgs<T>& a, const double* lo, const double* hi, hipStream_t s) {
    switch (L) {
        NDWT_MARCH_CASE(2) NDWT_MARCH_CASE(4) NDWT_MARCH_CASE(6) NDWT_MARCH_CASE(8) NDWT_MARCH_CASE(10)
        NDWT_MARCH_CASE(12) NDWT_MARCH_CASE(14) NDWT_MARCH_CASE(16) NDWT_MARCH_CASE(18) NDWT_MARCH_CASE(20)
        default: return -1;
    }
}

int launch_march_f32(bool syn, int L, const MarchArgs<float>& a, const double* lo, const double* hi, hipStream_t s) {
    return launch_march_T<float>(syn, L, a, lo, hi, s);
}
int launch_march_f64(bool syn, int L, const MarchArgs<double>& a, const double* lo, const double* hi, hipStream_t s) {
    return launch_march_T<double>(syn, L, a, lo, hi, s);
}

}  // namespace ndwt
