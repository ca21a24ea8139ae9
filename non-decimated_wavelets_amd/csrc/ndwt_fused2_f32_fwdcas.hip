// cascaded 2-D analysis, float real data: two or three levels of an image in one launch (Fwd2C), tap lengths 2 .. 8 and 12
#include "ndwt_fused_kernels.h"
namespace ndwt {
template <int LL, int NLEV> static int go(const Fused2CArgs<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Fwd2C<float, LL, NLEV, 2> K;
    if (a.ntx != (a.n1 + K::WX - 1) / K::WX || a.ychunk < 1 || (long long)a.nyc * a.ychunk < a.n2) return -2;
    hipLaunchKernelGGL(fused3_kernel<K>, dim3(a.ntx * a.nyc), dim3(K::NT), 0, s, a, (const typename K::Taps*)taps_dev);
    return (int)hipGetLastError();
}
int fwd2c_tile_width(int Lp, int nlev) {
    const int LH = Lp / 2 - 1, RH = Lp / 2;
    return 4 * ((64 - nlev * ((LH + 3) / 4 + (RH + 3) / 4)) / 8 * 8);
}
int launch_fwd2c_f32(const Fused2CArgs<float>& a, int Lp, int nlev, const void* taps_dev, hipStream_t s) {
#define NDWT_CAS(LL) case LL: return nlev == 3 ? go<LL, 3>(a, taps_dev, s) : go<LL, 2>(a, taps_dev, s);
    if (nlev != 2 && nlev != 3) return -1;
    switch (Lp) {
        NDWT_CAS(2) NDWT_CAS(4) NDWT_CAS(6) NDWT_CAS(8)
        case 12: return nlev == 2 ? go<12, 2>(a, taps_dev, s) : -1;   // (three levels of 12 taps: 132 spilled registers)
        default: return -1;
    }
#undef NDWT_CAS
}
}  // namespace ndwt
