/* ndwt_spatial.c -- CPU oracle #2 for the non-decimated wavelet transform hot path: TEST INFRASTRUCTURE ONLY.
 *
 * A plain C / OpenMP restatement of the signal-domain formula of SURVEY.md 3.4 -- the same numbers the reference computes in the DFT
 * domain (Functions/nd_dwt_3D.m:345-374, mex/nddwt.c:98-186), written as periodic correlations:
 *
 *   analysis along axis a :  lo[n] = s * sum_m LO_D[m] * x[(n - (m - L/2) * stride) mod N]      (hi likewise with HI_D)
 *   synthesis along axis a:  r[n]  = s * sum_m (LO_D[m] * a[(n + (m - L/2) * stride) mod N] + HI_D[m] * d[(n + (m - L/2) * stride) mod N])
 *
 * with s = 1/sqrt(2) when pres_l2_norm is set and 1 otherwise, a final 1 / 2^d per reconstructed level otherwise
 * (nd_dwt_3D.m:233-235), stride = 1 at every level (what the reference computes, nd_dwt_3D.m:178-186) or 2^(level-1) ("a trous").
 * Level loop and band order: nd_dwt_3D.m:178-186,229-244 / mex/nddwt.c:189-292 -- band 0 = coarsest approximation, the 2^d - 1 detail
 * bands of level l at 1 + (2^d - 1)(level - l) .., band bit a = high-pass on axis a (nd_dwt_3D.m:334-341).
 *
 * It is the checker and the CPU baseline, never the product: only tests/, __graft_entry__ (build / smoke) and bench.py's cpu_baseline
 * leg may build, load or call it.  PARITY STATUS: "parity unpinned", as oracle/ndwt_oracle.py (the reference stores no vectors); this
 * file is pinned against that restatement (tests/test_oracle_c.py), with which it shares the tap tables only.
 *
 * Arrays: C order, shape[0] outermost; a volume has prod(shape) * ncomp scalars (ncomp = 2: interleaved complex, filtered as two real
 * signals); coefficients are band-planar: y[band][volume].  Axis k of `shape` takes the taps lo[k], hi[k] (LO_D / HI_D of
 * wave_filters.m, length len[k], each row of the tables `maxlen` long).
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC oracle/ndwt_spatial.c -o oracle/_build/libndwt_spatial.so   (oracle/Makefile)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef long long i64;

#define NDWT_C_MAXDIM 4

static i64 wrap(i64 i, i64 n) {
    i %= n;
    return i < 0 ? i + n : i;
}

/* ---- one axis, generic over the scalar type ---- */
#define DEFINE_AXIS(T, SUF)                                                                                                           \
    /* x: [outer][N][inner] -> lo, hi of the same shape */                                                                           \
    static void analysis_axis_##SUF(const T* x, T* lo, T* hi, i64 outer, i64 N, i64 inner, const double* lo_d, const double* hi_d,   \
                                    int L, double s, i64 stride) {                                                                   \
        if (inner == 1) {                       /* the contiguous axis: one periodically padded copy of the row per thread */        \
            const i64 P = (i64)(L / 2) * stride;                                                                                     \
            _Pragma("omp parallel")                                                                                                  \
            {                                                                                                                        \
                T* xp = (T*)malloc((size_t)(N + 2 * P) * sizeof(T));                                                                 \
                _Pragma("omp for schedule(static)") for (i64 o = 0; o < outer; ++o) {                                                \
                    if (!xp) continue;                                                                                               \
                    for (i64 j = 0; j < N + 2 * P; ++j) xp[j] = x[o * N + wrap(j - P, N)];                                           \
                    T* pl = lo + o * N;                                                                                              \
                    T* ph = hi + o * N;                                                                                              \
                    for (i64 n = 0; n < N; ++n) pl[n] = ph[n] = 0;                                                                   \
                    for (int m = 0; m < L; ++m) {                                                                                    \
                        const T tl = (T)(s * lo_d[m]), th = (T)(s * hi_d[m]);                                                        \
                        const T* px = xp + P - (i64)(m - L / 2) * stride;                                                            \
                        for (i64 n = 0; n < N; ++n) {                                                                                \
                            pl[n] += tl * px[n];                                                                                     \
                            ph[n] += th * px[n];                                                                                     \
                        }                                                                                                            \
                    }                                                                                                                \
                }                                                                                                                    \
                free(xp);                                                                                                            \
            }                                                                                                                        \
            return;                                                                                                                  \
        }                                                                                                                            \
        const i64 CH = 2048, nch = (inner + CH - 1) / CH, items = outer * N * nch;                                                   \
        _Pragma("omp parallel for schedule(static)") for (i64 it = 0; it < items; ++it) {                                            \
            const i64 r = it / nch, i0 = (it % nch) * CH, i1 = i0 + CH < inner ? i0 + CH : inner;                                    \
            const i64 o = r / N, n = r % N;                                                                                          \
            T* pl = lo + r * inner;                                                                                                  \
            T* ph = hi + r * inner;                                                                                                  \
            for (i64 i = i0; i < i1; ++i) pl[i] = ph[i] = 0;                                                                         \
            for (int m = 0; m < L; ++m) {                                                                                            \
                const T* px = x + (o * N + wrap(n - (i64)(m - L / 2) * stride, N)) * inner;                                          \
                const T tl = (T)(s * lo_d[m]), th = (T)(s * hi_d[m]);                                                                \
                for (i64 i = i0; i < i1; ++i) {                                                                                      \
                    pl[i] += tl * px[i];                                                                                             \
                    ph[i] += th * px[i];                                                                                             \
                }                                                                                                                    \
            }                                                                                                                        \
        }                                                                                                                            \
    }                                                                                                                                \
    /* a, d: [outer][N][inner] -> r (post = the factor applied to the finished sum: 1 or 1 / 2^d on the last axis of a level) */      \
    static void synthesis_axis_##SUF(const T* a, const T* d, T* r_, i64 outer, i64 N, i64 inner, const double* lo_d,                 \
                                     const double* hi_d, int L, double s, i64 stride, double post) {                                 \
        if (inner == 1) {                                                                                                            \
            const i64 P = (i64)(L / 2) * stride;                                                                                     \
            _Pragma("omp parallel")                                                                                                  \
            {                                                                                                                        \
                T* ap = (T*)malloc((size_t)(N + 2 * P) * 2 * sizeof(T));                                                             \
                T* dp = ap ? ap + (N + 2 * P) : 0;                                                                                   \
                _Pragma("omp for schedule(static)") for (i64 o = 0; o < outer; ++o) {                                                \
                    if (!ap) continue;                                                                                               \
                    for (i64 j = 0; j < N + 2 * P; ++j) {                                                                            \
                        const i64 q = o * N + wrap(j - P, N);                                                                        \
                        ap[j] = a[q];                                                                                                \
                        dp[j] = d[q];                                                                                                \
                    }                                                                                                                \
                    T* pr = r_ + o * N;                                                                                              \
                    for (i64 n = 0; n < N; ++n) pr[n] = 0;                                                                           \
                    for (int m = 0; m < L; ++m) {                                                                                    \
                        const T tl = (T)(s * lo_d[m]), th = (T)(s * hi_d[m]);                                                        \
                        const T* pa = ap + P + (i64)(m - L / 2) * stride;                                                            \
                        const T* pd = dp + P + (i64)(m - L / 2) * stride;                                                            \
                        for (i64 n = 0; n < N; ++n) pr[n] += tl * pa[n] + th * pd[n];                                                \
                    }                                                                                                                \
                    if (post != 1.0)                                                                                                 \
                        for (i64 n = 0; n < N; ++n) pr[n] *= (T)post;                                                                \
                }                                                                                                                    \
                free(ap);                                                                                                            \
            }                                                                                                                        \
            return;                                                                                                                  \
        }                                                                                                                            \
        const i64 CH = 2048, nch = (inner + CH - 1) / CH, items = outer * N * nch;                                                   \
        _Pragma("omp parallel for schedule(static)") for (i64 it = 0; it < items; ++it) {                                            \
            const i64 r = it / nch, i0 = (it % nch) * CH, i1 = i0 + CH < inner ? i0 + CH : inner;                                    \
            const i64 o = r / N, n = r % N;                                                                                          \
            T* pr = r_ + r * inner;                                                                                                  \
            for (i64 i = i0; i < i1; ++i) pr[i] = 0;                                                                                 \
            for (int m = 0; m < L; ++m) {                                                                                            \
                const i64 q = (o * N + wrap(n + (i64)(m - L / 2) * stride, N)) * inner;                                              \
                const T tl = (T)(s * lo_d[m]), th = (T)(s * hi_d[m]);                                                                \
                const T* pa = a + q;                                                                                                 \
                const T* pd = d + q;                                                                                                 \
                for (i64 i = i0; i < i1; ++i) pr[i] += tl * pa[i] + th * pd[i];                                                      \
            }                                                                                                                        \
            if (post != 1.0)                                                                                                         \
                for (i64 i = i0; i < i1; ++i) pr[i] *= (T)post;                                                                      \
        }                                                                                                                            \
    }                                                                                                                                \
    /* one level: x (volume) -> out[b], b = 0 .. 2^d - 1 (band bit a = high-pass on axis a).  Returns 0, or -1 when out of memory. */  \
    static int level_dec_##SUF(const T* x, T* const* out, int ndim, const i64* shape, int ncomp, const double* lo, const double* hi, \
                               const int* len, int maxlen, double s, i64 stride) {                                                   \
        i64 vol = ncomp;                                                                                                             \
        for (int k = 0; k < ndim; ++k) vol *= shape[k];                                                                              \
        const int nb = 1 << ndim;                                                                                                    \
        /* stage k filters axis k of the 2^k arrays of the stage before; the last stage writes the output bands */                     \
        const T* cur[1 << NDWT_C_MAXDIM];                                                                                            \
        T* tmp[2][1 << NDWT_C_MAXDIM];                                                                                               \
        memset(tmp, 0, sizeof tmp);                                                                                                  \
        cur[0] = x;                                                                                                                  \
        int rc = 0;                                                                                                                  \
        for (int k = 0; k < ndim && !rc; ++k) {                                                                                      \
            const int have = 1 << k;                                                                                                 \
            T** dst = tmp[k & 1];                                                                                                    \
            for (int i = 0; i < 2 * have; ++i) {                                                                                     \
                if (k == ndim - 1) dst[i] = out[i];                                                                                  \
                else if (!(dst[i] = (T*)malloc((size_t)vol * sizeof(T)))) rc = -1;                                                   \
            }                                                                                                                        \
            if (rc) break;                                                                                                           \
            i64 outer = 1, inner = ncomp;                                                                                            \
            for (int j = 0; j < k; ++j) outer *= shape[j];                                                                           \
            for (int j = k + 1; j < ndim; ++j) inner *= shape[j];                                                                    \
            for (int i = 0; i < have; ++i)                                                                                           \
                analysis_axis_##SUF(cur[i], dst[i], dst[i + have], outer, shape[k], inner, lo + (size_t)k * maxlen,                  \
                                    hi + (size_t)k * maxlen, len[k], s, stride);                                                     \
            if (k > 0)                                                                                                               \
                for (int i = 0; i < have; ++i) { free(tmp[(k - 1) & 1][i]); tmp[(k - 1) & 1][i] = 0; }                               \
            for (int i = 0; i < 2 * have; ++i) cur[i] = dst[i];                                                                      \
        }                                                                                                                            \
        for (int p = 0; p < 2; ++p)                                                                                                  \
            for (int i = 0; i < nb; ++i) {                                                                                           \
                int is_out = 0;                                                                                                      \
                for (int b = 0; b < nb; ++b) is_out |= tmp[p][i] == out[b];                                                          \
                if (tmp[p][i] && !is_out) free(tmp[p][i]);                                                                           \
            }                                                                                                                        \
        return rc;                                                                                                                   \
    }                                                                                                                                \
    /* one level: in[b], b = 0 .. 2^d - 1 -> r (volume) */                                                                           \
    static int level_rec_##SUF(const T* const* in, T* r, int ndim, const i64* shape, int ncomp, const double* lo, const double* hi,  \
                               const int* len, int maxlen, double s, i64 stride, int l2) {                                           \
        i64 vol = ncomp;                                                                                                             \
        for (int k = 0; k < ndim; ++k) vol *= shape[k];                                                                              \
        const T* cur[1 << NDWT_C_MAXDIM];                                                                                            \
        T* mine[1 << NDWT_C_MAXDIM];                                                                                                 \
        memset(mine, 0, sizeof mine);                                                                                                \
        for (int b = 0; b < (1 << ndim); ++b) cur[b] = in[b];                                                                        \
        int rc = 0;                                                                                                                  \
        for (int k = ndim - 1; k >= 0 && !rc; --k) {       /* the last axis first, as the analysis applied it last */                \
            const int half = 1 << k;                                                                                                 \
            T* dst[1 << NDWT_C_MAXDIM];                                                                                              \
            for (int i = 0; i < half; ++i) {                                                                                         \
                if (k == 0) dst[i] = r;                                                                                              \
                else if (!(dst[i] = (T*)malloc((size_t)vol * sizeof(T)))) rc = -1;                                                   \
            }                                                                                                                        \
            if (rc && k > 0)                                                                                                         \
                for (int i = 0; i < half; ++i) { free(dst[i]); dst[i] = 0; }                                                         \
            if (!rc) {                                                                                                               \
                i64 outer = 1, inner = ncomp;                                                                                        \
                for (int j = 0; j < k; ++j) outer *= shape[j];                                                                       \
                for (int j = k + 1; j < ndim; ++j) inner *= shape[j];                                                                \
                for (int i = 0; i < half; ++i)                                                                                       \
                    synthesis_axis_##SUF(cur[i], cur[i + half], dst[i], outer, shape[k], inner, lo + (size_t)k * maxlen,             \
                                         hi + (size_t)k * maxlen, len[k], s, stride, (k == 0 && !l2) ? 1.0 / (double)(1 << ndim) : 1.0); \
            }                                                                                                                        \
            for (int i = 0; i < 2 * half; ++i) { free(mine[i]); mine[i] = 0; }                                                       \
            for (int i = 0; i < half; ++i) {                                                                                         \
                cur[i] = dst[i];                                                                                                     \
                mine[i] = k == 0 ? 0 : dst[i];                                                                                       \
            }                                                                                                                        \
        }                                                                                                                            \
        for (int i = 0; i < (1 << NDWT_C_MAXDIM); ++i) free(mine[i]);                                                                \
        return rc;                                                                                                                   \
    }                                                                                                                                \
    /* the multi-level drivers: y is band-planar, bands = 2^d + (2^d - 1)(level - 1) */                                               \
    int ndwt_c_dec_##SUF(const T* x, T* y, int ndim, const i64* shape, int ncomp, const double* lo, const double* hi, const int* len, \
                         int maxlen, int level, int l2, int atrous) {                                                                \
        if (ndim < 1 || ndim > NDWT_C_MAXDIM || level < 1 || (ncomp != 1 && ncomp != 2)) return -2;                                  \
        i64 vol = ncomp;                                                                                                             \
        for (int k = 0; k < ndim; ++k) vol *= shape[k];                                                                              \
        const int nb = 1 << ndim;                                                                                                    \
        const double s = l2 ? 1.0 / sqrt(2.0) : 1.0;                                                                                 \
        T* approx[2] = {0, 0};                                                                                                       \
        const T* src = x;                                                                                                            \
        for (int lev = 1; lev <= level; ++lev) {                                                                                     \
            T* out[1 << NDWT_C_MAXDIM];                                                                                              \
            /* band 0 of a level that is not the last is the next level's input: a scratch volume */                                 \
            if (lev < level) {                                                                                                       \
                if (!approx[lev & 1] && !(approx[lev & 1] = (T*)malloc((size_t)vol * sizeof(T)))) { free(approx[0]); free(approx[1]); return -1; } \
                out[0] = approx[lev & 1];                                                                                            \
            } else {                                                                                                                 \
                out[0] = y;                                                                                                          \
            }                                                                                                                        \
            for (int b = 1; b < nb; ++b) out[b] = y + (size_t)(1 + (nb - 1) * (level - lev) + (b - 1)) * vol;                        \
            if (level_dec_##SUF(src, out, ndim, shape, ncomp, lo, hi, len, maxlen, s, atrous ? (i64)1 << (lev - 1) : 1)) {           \
                free(approx[0]); free(approx[1]);                                                                                    \
                return -1;                                                                                                           \
            }                                                                                                                        \
            src = out[0];                                                                                                            \
        }                                                                                                                            \
        free(approx[0]);                                                                                                             \
        free(approx[1]);                                                                                                             \
        return 0;                                                                                                                    \
    }                                                                                                                                \
    int ndwt_c_rec_##SUF(const T* y, T* x, int ndim, const i64* shape, int ncomp, const double* lo, const double* hi, const int* len, \
                         int maxlen, int level, int l2, int atrous) {                                                                \
        if (ndim < 1 || ndim > NDWT_C_MAXDIM || level < 1 || (ncomp != 1 && ncomp != 2)) return -2;                                  \
        i64 vol = ncomp;                                                                                                             \
        for (int k = 0; k < ndim; ++k) vol *= shape[k];                                                                              \
        const int nb = 1 << ndim;                                                                                                    \
        const double s = l2 ? 1.0 / sqrt(2.0) : 1.0;                                                                                 \
        T* approx[2] = {0, 0};                                                                                                       \
        const T* prev = y;                                  /* band 0: the coarsest approximation */                                 \
        for (int lev = level; lev >= 1; --lev) {            /* the coarsest level is reconstructed first */                          \
            const T* in[1 << NDWT_C_MAXDIM];                                                                                         \
            in[0] = prev;                                                                                                            \
            for (int b = 1; b < nb; ++b) in[b] = y + (size_t)(1 + (nb - 1) * (level - lev) + (b - 1)) * vol;                         \
            T* dst = x;                                                                                                              \
            if (lev > 1) {                                                                                                           \
                if (!approx[lev & 1] && !(approx[lev & 1] = (T*)malloc((size_t)vol * sizeof(T)))) { free(approx[0]); free(approx[1]); return -1; } \
                dst = approx[lev & 1];                                                                                               \
            }                                                                                                                        \
            if (level_rec_##SUF(in, dst, ndim, shape, ncomp, lo, hi, len, maxlen, s, atrous ? (i64)1 << (lev - 1) : 1, l2)) {        \
                free(approx[0]); free(approx[1]);                                                                                    \
                return -1;                                                                                                           \
            }                                                                                                                        \
            prev = dst;                                                                                                              \
        }                                                                                                                            \
        free(approx[0]);                                                                                                             \
        free(approx[1]);                                                                                                             \
        return 0;                                                                                                                    \
    }

DEFINE_AXIS(float, f32)
DEFINE_AXIS(double, f64)

#ifdef _OPENMP
#include <omp.h>
int ndwt_c_max_threads(void) { return omp_get_max_threads(); }
void ndwt_c_set_threads(int n) { omp_set_num_threads(n); }
#else
int ndwt_c_max_threads(void) { return 1; }
void ndwt_c_set_threads(int n) { (void)n; }
#endif
