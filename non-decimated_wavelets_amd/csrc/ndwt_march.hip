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

// ---- contiguous axis (AxisX): 1-D signals and interleaved complex arrays
namespace ndwt {

template <typename T, int L, bool SYN, int EW>
static int launch_axisx_L(const AxisXArgs<T>& a0, bool vec4, const double* lo, const double* hi, hipStream_t s) {
    AxisXArgs<T> a = a0;
    typedef AxisX<T, L, SYN, EW, true> K;
    typename K::Taps tp;
    for (int j = 0; j < L; ++j) { tp.lo[j] = (T)lo[j]; tp.hi[j] = (T)hi[j]; }
    a.nseg = (a.row + K::WX - 1) / K::WX;
    const long long nblocks = (a.outer * a.nseg + 3) / 4;
    if (nblocks <= 0 || nblocks > 0x7fffffffLL || a.row >= (1LL << 30)) return -2;
    if (vec4) hipLaunchKernelGGL(march_kernel<K>, dim3((unsigned)nblocks), dim3(K::NT), 0, s, a, tp);
    else hipLaunchKernelGGL((march_kernel<AxisX<T, L, SYN, EW, false>>), dim3((unsigned)nblocks), dim3(K::NT), 0, s, a, tp);
    return (int)hipGetLastError();
}

#define NDWT_AXISX_CASE(LL)                                                                                   \
    case LL:                                                                                                  \
        if (ew == 1) return syn ? launch_axisx_L<T, LL, true, 1>(a, vec4, lo, hi, s) : launch_axisx_L<T, LL, false, 1>(a, vec4, lo, hi, s); \
        return syn ? launch_axisx_L<T, LL, true, 2>(a, vec4, lo, hi, s) : launch_axisx_L<T, LL, false, 2>(a, vec4, lo, hi, s);
template <typename T>
static int launch_axisx_T(bool syn, int L, int ew, const AxisXArgs<T>& a, bool vec4, const double* lo, const double* hi, hipStream_t s) {
    if (ew != 1 && ew != 2) return -1;
    switch (L) {
        NDWT_AXISX_CASE(2) NDWT_AXISX_CASE(4) NDWT_AXISX_CASE(6) NDWT_AXISX_CASE(8) NDWT_AXISX_CASE(10) NDWT_AXISX_CASE(12)
        default: return -1;
    }
}
int launch_axisx_f32(bool syn, int L, int ew, const AxisXArgs<float>& a, bool vec4, const double* lo, const double* hi, hipStream_t s) {
    return launch_axisx_T<float>(syn, L, ew, a, vec4, lo, hi, s);
}
int launch_axisx_f64(bool syn, int L, int ew, const AxisXArgs<double>& a, bool vec4, const double* lo, const double* hi, hipStream_t s) {
    return launch_axisx_T<double>(syn, L, ew, a, vec4, lo, hi, s);
}

}  // namespace ndwt
