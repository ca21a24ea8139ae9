// ndwt_device.h -- device code of the non-decimated wavelet engine (gfx950 / CDNA4, wave64).
//
// What it computes (reference semantics; reference: mex/nddwt.c:98-186 and
// Functions/nd_dwt_3D.m:345-374 do the same in the DFT domain): per axis of length N, periodic,
//   analysis   lo[n] = sum_j alo[j] x[n - (L/2-1)s + j s],  hi[n] = sum_j ahi[j] x[n - (L/2-1)s + j s]
//   synthesis  r[n]  = sum_j slo[j] a[n - (L/2)s + j s] + shi[j] d[n - (L/2)s + j s]
// with alo[j] = c h[j], ahi[j] = c (-1)^j h[L-1-j], slo[j] = c' h[L-1-j], shi[j] = c' (-1)^(j+1) h[j]
// (host folds the l2 / 1/2 scales into the taps, ndwt_filters.h), s = tap stride of the level.
//
// Two families of kernels:
//   * axis kernels  -- one axis per launch, any stride/size/dtype; the general path.
//   * fused kernels -- all axes of a 3-D (or 2-D) level in ONE launch, 1 read -> 2^d writes
//     (analysis) / 2^d reads -> 1 write (synthesis): the compulsory HBM traffic.  A workgroup owns
//     an (x,y) tile and marches along the outer axis keeping the outer-axis filter window in
//     REGISTERS; the inner axes go through LDS tiles.  Arithmetic is on (lo,hi) pairs so the
//     compiler can emit packed v_pk_fma_f32.
//
// The fused code is written as per-thread "stage" functions driven through an executor so the
// very same source runs under a host emulator (tests/emu, clang + ASan) for index checking.
#pragma once
#include <stdint.h>
#include <type_traits>

#ifdef NDWT_HOST_EMU
#define NDWT_DEV inline
#define NDWT_UNROLL _Pragma("unroll")
#else
#include <hip/hip_runtime.h>
#define NDWT_DEV __device__ __forceinline__
#define NDWT_UNROLL _Pragma("unroll")
#endif

namespace ndwt {

constexpr int kMaxTaps = 20;

template <typename T> struct VecT {
    typedef T v2 __attribute__((ext_vector_type(2)));
    typedef T v4 __attribute__((ext_vector_type(4)));
};

NDWT_DEV int modn(int v, int n) {
    int m = v % n;
    return m < 0 ? m + n : m;
}
NDWT_DEV long long modn64(long long v, long long n) {
    long long m = v % n;
    return m < 0 ? m + n : m;
}

// Compile-time loop: the per-thread register arrays (filter windows, prefetch buffers) must only ever be
// indexed by constants that are visible BEFORE loop unrolling, otherwise hipcc merges the rotated
// switch cases and the arrays fall into scratch memory.
template <int I, int N, class F> NDWT_DEV void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
#define NDWT_SFOR(var, N) static_for<0, N>([&](auto var##_c) { constexpr int var = decltype(var##_c)::value;
#define NDWT_SEND });

// ------------------------------------------------------------------------------------------------
// Axis kernels (general path).  Array viewed as [outer][N][inner], inner contiguous.
// ------------------------------------------------------------------------------------------------
template <typename T> struct AxisTaps {
    T lo[kMaxTaps];
    T hi[kMaxTaps];
    int len;
};

template <typename T> struct AxisArgs {
    long long inner;      // elements faster than the axis (contiguous run)
    long long n;          // axis length of the OUTPUT (local planes in slab mode)
    long long outer;      // product of slower dims
    long long total;      // outer*n*inner
    long long stride;     // tap stride s
    long long left;       // left extent in samples: (L/2-1)*s analysis, (L/2)*s synthesis
    int wrap;             // 1: periodic over n; 0: input holds `left` + n + right planes (slab mode)
    long long n_in;       // axis length of the input (n if wrap, n + (L-1)*s otherwise)
};

// lo/hi[o,n,i] = sum_j taps[j] * in[o, n - left + j*stride, i]
template <typename T>
NDWT_DEV void axis_analysis_elem(long long idx, const T* __restrict__ in, T* __restrict__ lo, T* __restrict__ hi,
                                 const AxisTaps<T>& tp, const AxisArgs<T>& a) {
    long long i = idx % a.inner;
    long long t = idx / a.inner;
    long long n = t % a.n;
    long long o = t / a.n;
    const T* base = in + o * a.n_in * a.inner + i;
    long long pos = a.wrap ? modn64(n - a.left, a.n) : n;   // slab mode: input plane 0 == output plane -left
    T accl = 0, acch = 0;
    for (int j = 0; j < tp.len; ++j) {
        T v = base[pos * a.inner];
        accl += tp.lo[j] * v;
        acch += tp.hi[j] * v;
        pos += a.stride;
        if (a.wrap) { while (pos >= a.n) pos -= a.n; }
    }
    lo[idx] = accl;
    hi[idx] = acch;
}

// r[o,n,i] = sum_j lo[j]*a[o, n-left+j*stride, i] + hi[j]*d[...]
template <typename T>
NDWT_DEV void axis_synthesis_elem(long long idx, const T* __restrict__ ain, const T* __restrict__ din, T* __restrict__ out,
                                  const AxisTaps<T>& tp, const AxisArgs<T>& a) {
    long long i = idx % a.inner;
    long long t = idx / a.inner;
    long long n = t % a.n;
    long long o = t / a.n;
    long long boff = o * a.n_in * a.inner + i;
    long long pos = a.wrap ? modn64(n - a.left, a.n) : n;
    T acc = 0;
    for (int j = 0; j < tp.len; ++j) {
        acc += tp.lo[j] * ain[boff + pos * a.inner];
        acc += tp.hi[j] * din[boff + pos * a.inner];
        pos += a.stride;
        if (a.wrap) { while (pos >= a.n) pos -= a.n; }
    }
    out[idx] = acc;
}

// ------------------------------------------------------------------------------------------------
// Fused 3-D level.  Axis 0 = x (contiguous, n1), 1 = y (n2), 2 = z (n3, marched).  A leading batch
// dimension lets 4-D volumes reuse it.  Tap arrays are zero-padded to the common even length L.
// ------------------------------------------------------------------------------------------------
template <typename T, int L> struct Taps3 {
    T lo[3][L];
    T hi[3][L];
};

template <typename T> struct Fused3Args {
    const T* in[8];        // analysis: in[0] only; synthesis: the 2^3 bands
    T* out[8];             // analysis: the 2^3 bands; synthesis: out[0] only
    int n1, n2, n3;        // local sizes (outputs)
    int nbatch;
    long long in_bstride;  // elements between batch items, inputs
    long long out_bstride; // elements between batch items, outputs
    long long plane;       // n1*n2
    int zchunk;            // output planes per workgroup
    int ntx, nty, nzc;     // tiles per axis
    int z_wrap;            // 1: periodic in z; 0: inputs start `left` planes before local plane 0 (slab mode)
};

// XCD-aware block order: hardware deals workgroups round-robin over the 8 XCDs (each with its own
// L2), so give every XCD a contiguous run of logical tiles -- x/y neighbours then share halo rows
// through one L2.  Speed only; any placement is correct.
NDWT_DEV int xcd_remap(int bid, int nblocks) {
    const int nx = 8;
    int q = nblocks / nx, r = nblocks % nx;
    int k = bid % nx, i = bid / nx;
    int start = (k < r) ? k * (q + 1) : r * (q + 1) + (k - r) * q;
    return start + i;
}

struct TileCoord {
    int x0, y0, zbeg, zend, batch;
};

template <typename T> NDWT_DEV TileCoord decode_tile(const Fused3Args<T>& a, int bid, int TX, int TY) {
    int nblocks = a.ntx * a.nty * a.nzc * a.nbatch;
    int lb = xcd_remap(bid, nblocks);
    TileCoord tc;
    int tx = lb % a.ntx;
    lb /= a.ntx;
    int ty = lb % a.nty;
    lb /= a.nty;
    int zc = lb % a.nzc;
    tc.batch = lb / a.nzc;
    tc.x0 = tx * TX;
    tc.y0 = ty * TY;
    tc.zbeg = zc * a.zchunk;
    tc.zend = tc.zbeg + a.zchunk < a.n3 ? tc.zbeg + a.zchunk : a.n3;
    return tc;
}

// ---------------------------------------------------------------------------------- analysis ----
template <typename T, int L_, int TX_, int TY_, int NT_, int RY_, bool VEC4_> struct Fwd3 {
    static constexpr int L = L_, TX = TX_, TY = TY_, NT = NT_, RY = RY_;
    static constexpr bool VEC4 = VEC4_;
    static constexpr int NE = VEC4 ? 1 : 4;              // offsets kept per column
    static constexpr int LH = L / 2 - 1;                 // samples left of the output index
    static constexpr int RH = L / 2;                     // samples right of it
    static constexpr int GL = (LH + 3) / 4, GR = (RH + 3) / 4;   // halo in groups of 4 x
    static constexpr int W = TX + 4 * (GL + GR);         // haloed tile width
    static constexpr int NG = W / 4;
    static constexpr int NR = TY + L - 1;                // haloed tile rows
    static constexpr int NCOLS = NG * NR;                // z-stage columns (4 x each)
    static constexpr int NCOL = (NCOLS + NT - 1) / NT;   // columns per thread
    static constexpr int YITEMS = W * (TY / RY);
    static constexpr int NYI = (YITEMS + NT - 1) / NT;
    static constexpr int XITEMS = (TX / 4) * TY * 2;
    static constexpr int NXI = (XITEMS + NT - 1) / NT;
    static constexpr int XV = 4 * (1 + GL + GR);         // values an x item reads
    static_assert(TX % 4 == 0 && TY % RY == 0 && L % 2 == 0, "tile shape");
    typedef typename VecT<T>::v2 v2;
    typedef typename VecT<T>::v4 v4;
    typedef Taps3<T, L> Taps;
    typedef Fused3Args<T> Args;

    struct Shared {
        v2 zs[NR][W];        // (lo3, hi3) of the raw tile
        v2 ys[2][TY][W];     // [y-bit][row][x] of (z-bit 0, z-bit 1)
    };
    struct State {
        v4 win[NCOL][L];     // raw samples of the last L planes, rotating
        v4 nxt[NCOL];        // prefetched plane
        int off[NCOL][NE];
    };

    static NDWT_DEV void setup(State& st, const Args& a, const TileCoord& tc, int tid) {
        NDWT_SFOR(k, NCOL)
            int c = tid + k * NT;
            if (c >= NCOLS) c = NCOLS - 1;               // surplus lanes load a valid column, never store
            int ug = c % NG, r = c / NG;
            int y = modn(tc.y0 - LH + r, a.n2);
            int xb = tc.x0 - 4 * GL + 4 * ug;
            NDWT_SFOR(e, NE)
                st.off[k][e] = y * a.n1 + modn(xb + e, a.n1);
            NDWT_SEND
        NDWT_SEND
    }

    static NDWT_DEV void load_plane(State& st, const Args& a, const T* inb, int zraw) {
        long long zm = a.z_wrap ? (long long)modn(zraw, a.n3) : (long long)(zraw + LH);
        const T* p = inb + zm * a.plane;
        NDWT_SFOR(k, NCOL)
            if constexpr (VEC4) {
                st.nxt[k] = *reinterpret_cast<const v4*>(p + st.off[k][0]);
            } else {
                NDWT_SFOR(e, NE)
                    st.nxt[k][e] = p[st.off[k][e]];
                NDWT_SEND
            }
        NDWT_SEND
    }

    // rotation R: the newest plane lands in slot (R+L-1)%L; tap j reads slot (R+j)%L
    template <int R> static NDWT_DEV void zstage(State& st, Shared& sh, const Taps& tp, int tid) {
        NDWT_SFOR(k, NCOL)
            st.win[k][(R + L - 1) % L] = st.nxt[k];
            int c = tid + k * NT;
            if (c < NCOLS) {
                v2 a0 = (v2)(T(0)), a1 = (v2)(T(0)), a2 = (v2)(T(0)), a3 = (v2)(T(0));
                NDWT_SFOR(j, L)
                    v4 w = st.win[k][(R + j) % L];
                    v2 t = {tp.lo[2][j], tp.hi[2][j]};
                    a0 += t * w[0]; a1 += t * w[1]; a2 += t * w[2]; a3 += t * w[3];
                NDWT_SEND
                int ug = c % NG, r = c / NG;
                v2* dst = &sh.zs[r][4 * ug];
                dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
            }
        NDWT_SEND
    }
    template <int R> static NDWT_DEV void zdispatch(int r, State& st, Shared& sh, const Taps& tp, int tid) {
        if constexpr (R < L) {
            if (r == R) zstage<R>(st, sh, tp, tid);
            else zdispatch<R + 1>(r, st, sh, tp, tid);
        }
    }
    static NDWT_DEV void prologue(State& st, const Args& a, const T* inb, int zbeg) {
        NDWT_SFOR(j, L - 1)
            load_plane(st, a, inb, zbeg - LH + j);
            NDWT_SFOR(k, NCOL)
                st.win[k][j] = st.nxt[k];
            NDWT_SEND
        NDWT_SEND
    }

    static NDWT_DEV void ystage(Shared& sh, const Taps& tp, int tid) {
        NDWT_UNROLL
        for (int k = 0; k < NYI; ++k) {
            int it = tid + k * NT;
            if (it >= YITEMS) continue;
            int u = it % W, yg = it / W;
            v2 zin[RY + L - 1];
            NDWT_UNROLL
            for (int r = 0; r < RY + L - 1; ++r) zin[r] = sh.zs[yg * RY + r][u];
            NDWT_UNROLL
            for (int i = 0; i < RY; ++i) {
                v2 lo = (v2)(T(0)), hi = (v2)(T(0));
                NDWT_UNROLL
                for (int j = 0; j < L; ++j) {
                    lo += tp.lo[1][j] * zin[i + j];
                    hi += tp.hi[1][j] * zin[i + j];
                }
                sh.ys[0][yg * RY + i][u] = lo;
                sh.ys[1][yg * RY + i][u] = hi;
            }
        }
    }

    static NDWT_DEV void xstage(Shared& sh, const Taps& tp, const Args& a, const TileCoord& tc, long long obase, int z,
                                int tid) {
        NDWT_UNROLL
        for (int k = 0; k < NXI; ++k) {
            int it = tid + k * NT;
            if (it >= XITEMS) continue;
            int xg = it % (TX / 4);
            int y = (it / (TX / 4)) % TY;
            int q = it / ((TX / 4) * TY);
            int gy = tc.y0 + y, gx = tc.x0 + 4 * xg;
            if (gy >= a.n2 || gx >= a.n1) continue;
            v2 v[XV];
            NDWT_UNROLL
            for (int t = 0; t < XV; ++t) v[t] = sh.ys[q][y][4 * xg + t];
            v4 o00, o01, o10, o11;                        // [x-bit][z-bit]
            NDWT_UNROLL
            for (int e = 0; e < 4; ++e) {
                v2 lo = (v2)(T(0)), hi = (v2)(T(0));
                NDWT_UNROLL
                for (int j = 0; j < L; ++j) {
                    lo += tp.lo[0][j] * v[4 * GL + e - LH + j];
                    hi += tp.hi[0][j] * v[4 * GL + e - LH + j];
                }
                o00[e] = lo.x; o01[e] = lo.y; o10[e] = hi.x; o11[e] = hi.y;
            }
            long long off = obase + (long long)z * a.plane + (long long)gy * a.n1 + gx;
            T* b00 = a.out[2 * q] + off;
            T* b10 = a.out[2 * q + 1] + off;
            T* b01 = a.out[2 * q + 4] + off;
            T* b11 = a.out[2 * q + 5] + off;
            if constexpr (VEC4) {
                *reinterpret_cast<v4*>(b00) = o00;
                *reinterpret_cast<v4*>(b10) = o10;
                *reinterpret_cast<v4*>(b01) = o01;
                *reinterpret_cast<v4*>(b11) = o11;
            } else {
                NDWT_UNROLL
                for (int e = 0; e < 4; ++e) {
                    if (gx + e < a.n1) { b00[e] = o00[e]; b10[e] = o10[e]; b01[e] = o01[e]; b11[e] = o11[e]; }
                }
            }
        }
    }

    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared& sh, const Args& a, const Taps& tp, int bid) {
        const TileCoord tc = decode_tile(a, bid, TX, TY);
        const T* inb = a.in[0] + (long long)tc.batch * a.in_bstride;
        const long long obase = (long long)tc.batch * a.out_bstride;
        // planes zbeg-LH .. zbeg-LH+L-2 into slots 0..L-2, then prefetch the plane of step 0
        ex.each([&](int tid, State& st) {
            setup(st, a, tc, tid);
            prologue(st, a, inb, tc.zbeg);
            load_plane(st, a, inb, tc.zbeg + RH);
        });
        const int nsteps = tc.zend - tc.zbeg;
        for (int s = 0; s < nsteps; ++s) {
            const int z = tc.zbeg + s;
            ex.each([&](int tid, State& st) {
                zdispatch<0>(s % L, st, sh, tp, tid);                     // consumes st.nxt
                if (s + 1 < nsteps) load_plane(st, a, inb, z + 1 + RH);   // prefetch for the next step
            });
            ex.barrier();
            ex.each([&](int tid, State&) { ystage(sh, tp, tid); });
            ex.barrier();
            ex.each([&](int tid, State&) { xstage(sh, tp, a, tc, obase, z, tid); });
        }
    }
};

// --------------------------------------------------------------------------------- synthesis ----
// x-synthesis and y-synthesis go through LDS on the haloed tile (the 2^3 bands are read with an
// x/y halo, mostly from L2), the z-synthesis window (L planes of (a,d) pairs) stays in registers.
// Per new plane: for y-bit 0,1 { raw 4 bands -> LDS; x-synth -> xs[y-bit] } ; y-synth -> P ; z-synth.
template <typename T, int L_, int TX_, int TY_, int NT_, int RY_, bool VEC4_> struct Inv3 {
    static constexpr int L = L_, TX = TX_, TY = TY_, NT = NT_, RY = RY_;
    static constexpr bool VEC4 = VEC4_;
    static constexpr int NE = VEC4 ? 1 : 4;
    static constexpr int LH = L / 2;                     // synthesis: samples left of the output index
    static constexpr int RH = L / 2 - 1;                 // samples right of it
    static constexpr int GL = (LH + 3) / 4, GR = (RH + 3) / 4;
    static constexpr int W = TX + 4 * (GL + GR);
    static constexpr int NG = W / 4;
    static constexpr int NR = TY + L - 1;
    static constexpr int LITEMS = 2 * NG * NR;           // load items: (x-bit, row, group of 4 x)
    static constexpr int NLI = (LITEMS + NT - 1) / NT;
    static constexpr int XITEMS = (TX / 4) * NR;         // x-synthesis items per y-bit
    static constexpr int NXI = (XITEMS + NT - 1) / NT;
    static constexpr int YITEMS = TX * (TY / RY);        // y/z-synthesis items: one x, RY rows
    static constexpr int NYI = (YITEMS + NT - 1) / NT;
    static constexpr int XV = 4 * (1 + GL + GR);
    static_assert(TX % 4 == 0 && TY % RY == 0 && L % 2 == 0, "tile shape");
    typedef typename VecT<T>::v2 v2;
    typedef typename VecT<T>::v4 v4;
    typedef Taps3<T, L> Taps;
    typedef Fused3Args<T> Args;

    struct Shared {
        v2 raw[2][NR][W];    // [x-bit][row][x] of (z-bit 0, z-bit 1), one y-bit at a time
        v2 xs[2][NR][TX];    // [y-bit][row][x] after x-synthesis
    };
    struct State {
        v2 win[NYI][RY][L];  // (a,d) pairs for the z-synthesis, rotating
        v4 pre[NLI][2];      // prefetched raw values: [item][z-bit]
        int off[NLI][NE];
    };

    static NDWT_DEV void setup(State& st, const Args& a, const TileCoord& tc, int tid) {
        NDWT_SFOR(k, NLI)
            int it = tid + k * NT;
            if (it >= LITEMS) it = LITEMS - 1;
            int ug = it % NG, r = (it / NG) % NR;
            int y = modn(tc.y0 - LH + r, a.n2);
            int xb = tc.x0 - 4 * GL + 4 * ug;
            NDWT_SFOR(e, NE)
                st.off[k][e] = y * a.n1 + modn(xb + e, a.n1);
            NDWT_SEND
        NDWT_SEND
    }

    // issue the global loads of (plane zraw, y-bit yb) into st.pre
    static NDWT_DEV void load_raw(State& st, const Args& a, long long ibase, int zraw, int yb, int tid) {
        long long zm = a.z_wrap ? (long long)modn(zraw, a.n3) : (long long)(zraw + LH);
        long long pb = ibase + zm * a.plane;
        NDWT_SFOR(k, NLI)
            int it = tid + k * NT;
            if (it >= LITEMS) it = LITEMS - 1;
            int xb = it / (NG * NR);
            const T* p0 = a.in[xb + 2 * yb] + pb;       // z-bit 0
            const T* p1 = a.in[xb + 2 * yb + 4] + pb;   // z-bit 1
            if constexpr (VEC4) {
                st.pre[k][0] = *reinterpret_cast<const v4*>(p0 + st.off[k][0]);
                st.pre[k][1] = *reinterpret_cast<const v4*>(p1 + st.off[k][0]);
            } else {
                NDWT_SFOR(e, NE)
                    st.pre[k][0][e] = p0[st.off[k][e]];
                    st.pre[k][1][e] = p1[st.off[k][e]];
                NDWT_SEND
            }
        NDWT_SEND
    }

    static NDWT_DEV void stash_raw(State& st, Shared& sh, int tid) {
        NDWT_SFOR(k, NLI)
            int it = tid + k * NT;
            if (it < LITEMS) {
                int ug = it % NG, r = (it / NG) % NR, xb = it / (NG * NR);
                v2* dst = &sh.raw[xb][r][4 * ug];
                v4 p0 = st.pre[k][0], p1 = st.pre[k][1];
                dst[0] = v2{p0[0], p1[0]};
                dst[1] = v2{p0[1], p1[1]};
                dst[2] = v2{p0[2], p1[2]};
                dst[3] = v2{p0[3], p1[3]};
            }
        NDWT_SEND
    }

    static NDWT_DEV void xsyn(Shared& sh, const Taps& tp, int yb, int tid) {
        NDWT_UNROLL
        for (int k = 0; k < NXI; ++k) {
            int it = tid + k * NT;
            if (it >= XITEMS) continue;
            int xg = it % (TX / 4), r = it / (TX / 4);
            v2 av[XV], dv[XV];
            NDWT_UNROLL
            for (int t = 0; t < XV; ++t) {
                av[t] = sh.raw[0][r][4 * xg + t];
                dv[t] = sh.raw[1][r][4 * xg + t];
            }
            NDWT_UNROLL
            for (int e = 0; e < 4; ++e) {
                v2 acc = (v2)(T(0));
                NDWT_UNROLL
                for (int j = 0; j < L; ++j) {
                    acc += tp.lo[0][j] * av[4 * GL + e - LH + j];
                    acc += tp.hi[0][j] * dv[4 * GL + e - LH + j];
                }
                sh.xs[yb][r][4 * xg + e] = acc;
            }
        }
    }

    // y-synthesis of the new plane into window slot (R+L-1)%L; if `emit`, z-synthesis of plane z and store
    template <int R>
    static NDWT_DEV void yzsyn(State& st, Shared& sh, const Taps& tp, const Args& a, const TileCoord& tc, long long obase,
                               int z, bool emit, int tid) {
        NDWT_SFOR(k, NYI)
            int it = tid + k * NT;
            if (it < YITEMS) {
                int x = it % TX, yg = it / TX;
                v2 av[RY + L - 1], dv[RY + L - 1];
                NDWT_UNROLL
                for (int t = 0; t < RY + L - 1; ++t) {
                    av[t] = sh.xs[0][yg * RY + t][x];
                    dv[t] = sh.xs[1][yg * RY + t][x];
                }
                NDWT_SFOR(i, RY)
                    v2 acc = (v2)(T(0));
                    NDWT_UNROLL
                    for (int j = 0; j < L; ++j) {
                        acc += tp.lo[1][j] * av[i + j];
                        acc += tp.hi[1][j] * dv[i + j];
                    }
                    st.win[k][i][(R + L - 1) % L] = acc;
                NDWT_SEND
                if (emit) {
                    int gx = tc.x0 + x;
                    NDWT_SFOR(i, RY)
                        v2 acc = (v2)(T(0));
                        NDWT_SFOR(j, L)
                            v2 t = {tp.lo[2][j], tp.hi[2][j]};
                            acc += t * st.win[k][i][(R + j) % L];
                        NDWT_SEND
                        int gy = tc.y0 + yg * RY + i;
                        if (gx < a.n1 && gy < a.n2)
                            a.out[0][obase + (long long)z * a.plane + (long long)gy * a.n1 + gx] = acc.x + acc.y;
                    NDWT_SEND
                }
            }
        NDWT_SEND
    }
    template <int R>
    static NDWT_DEV void yzdispatch(int r, State& st, Shared& sh, const Taps& tp, const Args& a, const TileCoord& tc,
                                    long long obase, int z, bool emit, int tid) {
        if constexpr (R < L) {
            if (r == R) yzsyn<R>(st, sh, tp, a, tc, obase, z, emit, tid);
            else yzdispatch<R + 1>(r, st, sh, tp, a, tc, obase, z, emit, tid);
        }
    }

    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared& sh, const Args& a, const Taps& tp, int bid) {
        const TileCoord tc = decode_tile(a, bid, TX, TY);
        const long long ibase = (long long)tc.batch * a.in_bstride;
        const long long obase = (long long)tc.batch * a.out_bstride;
        const int nsteps = tc.zend - tc.zbeg;
        const int nplanes = nsteps + L - 1;              // planes zbeg-LH .. zend-1+RH
        ex.each([&](int tid, State& st) {
            setup(st, a, tc, tid);
            load_raw(st, a, ibase, tc.zbeg - LH, 0, tid);
        });
        for (int p = 0; p < nplanes; ++p) {
            const int zraw = tc.zbeg - LH + p;
            const int s = p - (L - 1);                   // output step this plane completes (if >= 0)
            for (int yb = 0; yb < 2; ++yb) {
                ex.each([&](int tid, State& st) {
                    stash_raw(st, sh, tid);
                    if (yb == 0) load_raw(st, a, ibase, zraw, 1, tid);
                    else if (p + 1 < nplanes) load_raw(st, a, ibase, zraw + 1, 0, tid);
                });
                ex.barrier();
                ex.each([&](int tid, State&) { xsyn(sh, tp, yb, tid); });
                ex.barrier();
            }
            ex.each([&](int tid, State& st) {
                // plane p lives in slot p%L; rotation R puts the newest into (R+L-1)%L -> R = (p+1)%L
                yzdispatch<0>((p + 1) % L, st, sh, tp, a, tc, obase, tc.zbeg + s, s >= 0, tid);
            });
        }
    }
};

}  // namespace ndwt
