// ndwt_filters.h -- host-side filter construction (replaces Functions/wave_filters.m and the
// get_filters() of the classes, nd_dwt_3D.m:263-342: no N-D kernel volumes, only per-axis taps).
#pragma once
#include <cctype>
#include <cmath>
#include <cstring>
#include <string>

#include "ndwt_taps.inc"

namespace ndwt {

// "dbK" (case-insensitive, K = 1..10) -> K, or 0 if unknown (wave_filters.m:158-160)
inline int parse_wavelet(const char* wname) {
    if (!wname) return 0;
    std::string w(wname);
    for (auto& c : w) c = (char)std::tolower((unsigned char)c);
    if (w.size() < 3 || w[0] != 'd' || w[1] != 'b') return 0;
    int k = 0;
    for (size_t i = 2; i < w.size(); ++i) {
        if (!std::isdigit((unsigned char)w[i])) return 0;
        k = k * 10 + (w[i] - '0');
        if (k > 1000) return 0;
    }
    return (k >= 1 && k <= NDWT_MAX_ORDER) ? k : 0;
}

// h[0..2K-1], large taps first (the literal order of wave_filters.m:21-156)
inline void scaling_filter(int K, double* h) {
    for (int j = 0; j < 2 * K; ++j) h[j] = NDWT_DB_H[K - 1][j];
    if (K == 1) h[0] = h[1] = 1.0 / std::sqrt(2.0);   // wave_filters.m:21
}

// LO_D[m] = h[L-1-m], HI_D[m] = (-1)^(m+1) h[m]   (wave_filters.m:164-172)
inline void wave_filters(int K, double* lo_d, double* hi_d) {
    double h[20];
    scaling_filter(K, h);
    const int L = 2 * K;
    for (int m = 0; m < L; ++m) {
        lo_d[m] = h[L - 1 - m];
        hi_d[m] = ((m + 1) % 2 ? -1.0 : 1.0) * h[m];
    }
}

// Kernel-form taps of one axis, scales folded in:
//   analysis  out[n] = sum_j a[j] x[n - (L/2-1) s + j s]   a_lo[j] = c LO_D[L-1-j], a_hi[j] = c HI_D[L-1-j]
//   synthesis r[n]   = sum_j s_lo[j] a[n - (L/2) s + j s] + s_hi[j] d[...]   s_lo[j] = c' LO_D[j], s_hi[j] = c' HI_D[j]
// c = 1/sqrt(2) with pres_l2_norm else 1 (nd_dwt_1D.m:278-282); c' = 1/sqrt(2) with pres_l2_norm,
// else 1/2 per axis = the 1/2^d of nd_dwt_3D.m:233-235 / nddwt.c:176-182.
struct AxisFilter {
    int len;
    double ana_lo[20], ana_hi[20], syn_lo[20], syn_hi[20];
};

inline AxisFilter make_axis_filter(int K, bool pres_l2_norm) {
    AxisFilter f;
    double lo_d[20], hi_d[20];
    wave_filters(K, lo_d, hi_d);
    const int L = 2 * K;
    f.len = L;
    const double c = pres_l2_norm ? 1.0 / std::sqrt(2.0) : 1.0;
    const double cs = pres_l2_norm ? 1.0 / std::sqrt(2.0) : 0.5;
    for (int j = 0; j < L; ++j) {
        f.ana_lo[j] = c * lo_d[L - 1 - j];
        f.ana_hi[j] = c * hi_d[L - 1 - j];
        f.syn_lo[j] = cs * lo_d[j];
        f.syn_hi[j] = cs * hi_d[j];
    }
    return f;
}

// zero-pad symmetrically to the even length Lp >= len: the centring conventions above are preserved
inline void pad_taps(const double* src, int len, int Lp, double* dst) {
    const int pad = (Lp - len) / 2;
    for (int j = 0; j < Lp; ++j) dst[j] = 0.0;
    for (int j = 0; j < len; ++j) dst[j + pad] = src[j];
}

}  // namespace ndwt
