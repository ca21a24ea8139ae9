// cascaded 2-D synthesis, float real data: two or three levels of an image in one launch (Inv2C), tap lengths 2 .. 8
#include "ndwt_fused_kernels.h"
namespace ndwt {
template <int LL, int NLEV, int PD> static int go(const Fused2CIArgs<float>& a, const void* taps_dev, hipStream_t s) {
    typedef Inv2C<float, LL, NLEV, PD, 2> K;
    if (a.ntx != (a.n1 + K::WX - 1) / K::WX || a.ychunk < 1 || (long long)a.nyc * a.ychunk < a.n2) return -2;
    hipLaunchKernelGGL(fused3_kernel<K>, dim3(a.ntx * a.nyc), dim3(K::NT), 0, s, a, (const typename K::Taps*)taps_dev);
    return (int)hipGetLastError();
}
int inv2c_tile_width(int Lp, int nlev) {
    const int LH = Lp / 2, RH = Lp / 2 - 1;
    return 4 * ((64 - nlev * ((LH + 3) / 4 + (RH + 3) / 4)) / 8 * 8);
}
int launch_inv2c_f32(const Fused2CIArgs<float>& a, int Lp, int nlev, int depth, const void* taps_dev, hipStream_t s) {
#define NDWT_CAS(LL) case LL: return nlev == 3 ? (depth == 2 ? go<LL, 3, 2>(a, taps_dev, s) : go<LL, 3, 1>(a, taps_dev, s)) \
                                               : (depth == 2 ? go<LL, 2, 2>(a, taps_dev, s) : go<LL, 2, 1>(a, taps_dev, s));
    if (nlev != 2 && nlev != 3) return -1;
    switch (Lp) {
        NDWT_CAS(2) NDWT_CAS(4) NDWT_CAS(6) NDWT_CAS(8)
        default: return -1;
    }
#undef NDWT_CAS
}
}  // namespace ndwt
