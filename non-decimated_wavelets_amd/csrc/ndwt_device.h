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
    // the same vectors at element alignment: global dwordx2 / dwordx4 (x2 for double) accesses need no more than that.  What the
    // kernels' VEC4 = false instances use for every lane whose 4 x are contiguous in memory -- all but the one lane per row that
    // straddles the periodic wrap when the row length is not a multiple of 4 (reference shapes: mex/mex_test.m:48,84, 129 x 131 and
    // 131 x 128 x 30; Test/nddwt1D_test.m:5, N = 54321); that lane, and rows shorter than 4, go element by element.
    typedef T v2u __attribute__((ext_vector_type(2), aligned(sizeof(T))));
    typedef T v4u __attribute__((ext_vector_type(4), aligned(sizeof(T))));
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
#define NDWT_SFOR(var, N) static_for<0, N>([&](auto var##_c) __attribute__((always_inline)) { constexpr int var = decltype(var##_c)::value;
#define NDWT_SEND });

#ifdef NDWT_HOST_EMU
#define NDWT_SCHED_FENCE() ((void)0)
#define NDWT_SETPRIO(n) ((void)(n))
#else
#define NDWT_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)   // the instruction scheduler moves nothing across this point
#define NDWT_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif

// store of data this kernel writes once and never reads back: nontemporal for float data, so that the output streams do not
// displace the input lines neighbouring tiles are about to share in L2 (tools/micro/stream_pattern: 8 -> 1 copy 0.92 -> 0.86 ms,
// 1 -> 8 0.98 -> 0.96; cfg3 step -2 %, 2-D -5 %).  Double keeps plain stores: at 512^3 the nontemporal form takes the analysis
// kernel (32-byte stores per lane) from 2.11 to 2.81 ms per launch and the synthesis kernel from 2.26 to 2.32 ms.
// nt: set by the host for float data whose rows are whole 128-byte lines (300-float rows: a nontemporal store of a partly
// covered line costs a read-modify-write in memory -- 300x400x500 analysis 0.89 ms against 0.51 ms with plain stores, which L2 merges)
template <class V> NDWT_DEV void stream_store(V* p, V v, int nt) {
#if !defined(NDWT_HOST_EMU) && !defined(NDWT_NO_NT_STORE)
    if constexpr (sizeof(v[0]) == 4) {
        if (nt) { __builtin_nontemporal_store(v, p); return; }
    }
#endif
    (void)nt;
    *p = v;
}
NDWT_DEV float ndwt_sqrt(float v) { return __builtin_sqrtf(v); }
NDWT_DEV double ndwt_sqrt(double v) { return __builtin_sqrt(v); }

// soft / hard shrinkage of the 4 scalars one lane holds; EW = 2: two interleaved complex values, the magnitude shrinks
template <typename T, class V4> NDWT_DEV void shrink4_flat(V4& c, T thr, int hard);
template <typename T, int EW, class V4> NDWT_DEV void shrink4(V4& c, T thr, int hard) {
    if constexpr (EW == 1) {
        // real data: m > thr ? (hard ? v : (v < 0 ? v + thr : v - thr)) : 0 without divergent branches (the same operations, the same bits)
        shrink4_flat<T>(c, thr, hard);
    } else {
        NDWT_SFOR(h, 2)
            const T re = c[2 * h], im = c[2 * h + 1];
            const T m = ndwt_sqrt(re * re + im * im);
            const T g = m > thr ? (hard ? T(1) : (m - thr) / m) : T(0);
            c[2 * h] = re * g;
            c[2 * h + 1] = im * g;
        NDWT_SEND
    }
}

// the same thresholding of real data without divergent branches (Den3 thresholds 28 values per lane and plane): soft
// v - clamp(v, -t, t) (the same subtraction / addition shrink4 performs, so the same bits), hard |v| > t ? v : 0; `hard` is uniform
template <typename T, class V4> NDWT_DEV void shrink4_flat(V4& c, T thr, int hard) {
    if (hard) {
        NDWT_SFOR(e, 4)
            const T v = c[e];
            c[e] = (v < T(0) ? -v : v) > thr ? v : T(0);
        NDWT_SEND
    } else {
        NDWT_SFOR(e, 4)
            const T v = c[e];
#if !defined(NDWT_HOST_EMU)
            if constexpr (sizeof(T) == 4) c[e] = v - __builtin_amdgcn_fmed3f(v, -thr, thr);
            else c[e] = v - (v < -thr ? -thr : (v > thr ? thr : v));
#else
            c[e] = v - (v < -thr ? -thr : (v > thr ? thr : v));
#endif
        NDWT_SEND
    }
}

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
    long long plane;       // elements between planes (n1*n2; dilated levels: stride * n1 * n2)
    int rs;                // elements between rows (n1; dilated levels: stride * n1)
    int bsplit;            // > 0: batch item b starts at (b % bsplit) * bstride + (b / bsplit) * bstride2 (the stride^2 (y, z)
    long long in_bstride2, out_bstride2;   // sub-lattices of a dilated level; x is handled by EW = stride)
    int zchunk;            // output planes per workgroup
    int ntx, nty, nzc;     // tiles per axis
    int z_wrap;            // outer axis: 1 periodic; 0 inputs start `left` planes before local plane 0 (haloed slab);
                           // 2 analysis with separate halo buffers in[1] (before) / in[2] (after); 3 synthesis of the
                           // zero-extended slab: output plane k sums input planes k-(L-1)..k, planes outside
                           // [zlo, zhi) read as 0 (whole slab: zlo = 0, zhi = n_in, n3 = n_in + L-1)
    int zlo, zhi;          // z_wrap == 3 only
    int zbs;               // z_wrap == 3: batch item i tests plane + i*zbs against [zlo, zhi) (runs of one slab)
    // lane-shift synthesis only: shrink the input bands whose bit is set in shrink_mask as they are loaded (soft, or
    // hard with shrink_hard) -- the thresholding of the detail bands fused into the reconstruction
    T shrink_thr;
    int shrink_mask, shrink_hard;
    int nt;                // nontemporal output stores (float data whose rows are whole 128-byte lines)
    long long* stamps;     // diagnostic builds (-DNDWT_STAMPS) only: per wave, cycles spent in each phase of the plane loop
    // 4-D analysis with the t axis folded into the launch (Fwd3<.., TPRE>): the batch items are the frames of a periodic t axis and
    // every raw plane is the t-filtered combination, with these taps (the low- or the high-pass ones, zero-padded to L like the
    // others), of the same plane of the frames t - (L/2-1) .. t + L/2 -- read where the neighbouring frames' workgroups have just
    // read them (the frame index runs fastest in that kernel's block order, so those workgroups share an XCD's L2)
    T tt[kMaxTaps];
};

// XCD-aware block order: hardware deals workgroups round-robin over the 8 XCDs (each with its own
// L2), so give every XCD a contiguous run of logical tiles -- x/y neighbours then share halo rows
// through one L2.  Speed only; any placement is correct.
NDWT_DEV long long batch_base(int b, int bsplit, long long bs1, long long bs2) {
    return bsplit > 0 ? (long long)(b % bsplit) * bs1 + (long long)(b / bsplit) * bs2 : (long long)b * bs1;
}

// A wave-uniform 64-bit value the compiler holds in VGPRs (batch_base: the divisions of its index arithmetic run on the vector ALU), moved
// to SGPRs: what is derived from it per plane -- base pointer + plane offset of every band -- is then scalar arithmetic instead of one
// v_lshl_add_u64 and two v_readfirstlane_b32 per band and plane (27 of the ~380 vector instructions per wave and plane of Inv3Y<12>)
NDWT_DEV long long uniform_ll(long long v) {
#ifndef NDWT_HOST_EMU
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)v >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
#else
    return v;
#endif
}

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

// Tiles of rows that are not whole groups of 4 scalars (the VEC4 = false instances; reference shapes: mex/mex_test.m:48,84,
// Test/nddwt1D_test.m:5): a lane owns 4 consecutive scalars starting at tile origin + a multiple of 4, and takes them in ONE access
// when they are contiguous in memory.  With origins at multiples of the tile width the one lane per row whose 4 scalars straddle the
// periodic wrap goes element by element -- 4 accesses per band in every wave of the last tile column, 2.1x on the whole launch.
// So the tiles whose haloed extent reaches the end of the row are anchored THERE (origin = n - k * tile width): their lanes are
// aligned to the wrap point and none straddles it; they overlap the tile before them, whose outputs they write again with the same
// values.  (Rows shorter than such a tile and its left halo keep the plain origins and the element-wise lane.)
NDWT_DEV int tile_origin(int t, int ntiles, int width, int n, int halo_l, int halo_r) {
    const int x0 = t * width;
    const int xr = n - (ntiles - t) * width;              // origin of tile t counted from the end of the row
    return (n % 4 != 0 && x0 + width + halo_r > n && xr >= halo_l) ? xr : x0;
}

// BFAST (compile time: the extra division changes the register allocation of kernels that are a few registers from spilling): the batch
// item runs fastest in the block order, so the workgroups of one tile over all batch items are neighbours on one XCD
template <typename T, bool BFAST = false>
NDWT_DEV TileCoord decode_tile(const Fused3Args<T>& a, int bid, int TX, int TY, int halo_l = -1, int halo_r = 0) {
    int nblocks = a.ntx * a.nty * a.nzc * a.nbatch;
    int lb = xcd_remap(bid, nblocks);
    TileCoord tc;
    int fast_batch = 0;
    if constexpr (BFAST) {
        fast_batch = lb % a.nbatch;
        lb /= a.nbatch;
    }
    int tx = lb % a.ntx;
    lb /= a.ntx;
    int ty = lb % a.nty;
    lb /= a.nty;
    int zc = lb % a.nzc;
    tc.batch = BFAST ? fast_batch : lb / a.nzc;
    tc.x0 = halo_l >= 0 ? tile_origin(tx, a.ntx, TX, a.n1, halo_l, halo_r) : tx * TX;
    tc.y0 = ty * TY;
    tc.zbeg = zc * a.zchunk;
    tc.zend = tc.zbeg + a.zchunk < a.n3 ? tc.zbeg + a.zchunk : a.n3;
    return tc;
}

// ------------------------------------------------------------------------------------ LDS rows ----
// LDS tiles are rows of 16-byte chunks (float: 2 (lo,hi) pairs per chunk; double: 1 pair).  Every access is a
// whole chunk (ds_read_b128 / ds_write_b128: 256 B/clk, twice the rate of the ds_read2_b64 hipcc emits for 8-byte
// aligned pairs).  Chunk c of a row is stored at position S(c): lanes that each own 4 consecutive pairs step
// through chunks 2g+k (float) / 4g+k (double), i.e. every 2nd / 4th chunk, which without the swizzle puts lanes
// g and g+8 (g+4) on the same banks.  S permutes chunks inside aligned groups of 2 (float) / 4 (double), so
// rows must hold a multiple of 2 / 4 chunks.  Checked with tools/lds_bank_sim.py.
template <typename T> struct Lds;
template <> struct Lds<float> {
    typedef VecT<float>::v2 v2;
    typedef VecT<float>::v4 chunk;
    static constexpr int CH = 2;                                           // pairs per chunk
    static NDWT_DEV int S(int c) { return c ^ (((c >> 3) ^ (c >> 4)) & 1); }
    static NDWT_DEV v2 get(const chunk& c, int sub) { return sub ? v2{c[2], c[3]} : v2{c[0], c[1]}; }
    static NDWT_DEV void set(chunk& c, int sub, v2 v) {
        if (sub) { c[2] = v[0]; c[3] = v[1]; } else { c[0] = v[0]; c[1] = v[1]; }
    }
};
template <> struct Lds<double> {
    typedef VecT<double>::v2 v2;
    typedef VecT<double>::v2 chunk;
    static constexpr int CH = 1;
    static NDWT_DEV int S(int c) { return c ^ ((c >> 4) & 3); }
    static NDWT_DEV v2 get(const chunk& c, int) { return c; }
    static NDWT_DEV void set(chunk& c, int, v2 v) { c = v; }
};

// N consecutive pairs starting at pair index u0 (a multiple of 4) of a swizzled row
template <typename T, int N> NDWT_DEV void lds_load_run(const typename Lds<T>::chunk* row, int u0, typename Lds<T>::v2 (&v)[N]) {
    typedef Lds<T> LD;
    static_assert(N % LD::CH == 0, "whole chunks");
    const int c0 = u0 / LD::CH;
    NDWT_UNROLL
    for (int k = 0; k < N / LD::CH; ++k) {
        typename LD::chunk ch = row[LD::S(c0 + k)];
        NDWT_UNROLL
        for (int sub = 0; sub < LD::CH; ++sub) v[k * LD::CH + sub] = LD::get(ch, sub);
    }
}
template <typename T, int N> NDWT_DEV void lds_store_run(typename Lds<T>::chunk* row, int u0, const typename Lds<T>::v2 (&v)[N]) {
    typedef Lds<T> LD;
    static_assert(N % LD::CH == 0, "whole chunks");
    const int c0 = u0 / LD::CH;
    NDWT_UNROLL
    for (int k = 0; k < N / LD::CH; ++k) {
        typename LD::chunk ch;
        NDWT_UNROLL
        for (int sub = 0; sub < LD::CH; ++sub) LD::set(ch, sub, v[k * LD::CH + sub]);
        row[LD::S(c0 + k)] = ch;
    }
}

// rows r and r+m of a tile with row stride RSB bytes start on the same bank when m*RSB is a multiple of 256 B.
// Items that put 16 lanes on each row (a 64-wide tile = 16 groups of 4 x) pair such rows inside one 32-lane
// half-wave so a ds_read_b128's 16-lane groups, which mix both rows, stay conflict free: slot s -> row.
template <int RSB> struct RowPair {
    static constexpr int gcd256 = (RSB % 256 == 0) ? 256 : (RSB % 128 == 0) ? 128 : (RSB % 64 == 0) ? 64 : (RSB % 32 == 0) ? 32 : 16;
    static constexpr int M = 256 / gcd256;               // 1, 2, 4, 8 or 16
    static NDWT_DEV int row(int s) { return (s / (2 * M)) * (2 * M) + (s % (2 * M)) / 2 + M * (s % 2); }
    static constexpr int slots(int nrows) { return ((nrows + 2 * M - 1) / (2 * M)) * (2 * M); }
};

// ---- packed FMAs with explicit operand selection (v_pk_fma_f32 op_sel / neg modifiers), float only ----
// Why by hand: (1) hipcc keeps a scalar multiplier of a packed FMA as a DUPLICATED SGPR pair (t, t), which for the taps of three
// axes overflows the SGPR file, so the compiler re-loads them from constant memory (s_load + lgkmcnt(0), which also drains the LDS
// queue) inside the plane loop; (2) the high-pass taps are the low-pass taps mirrored with alternating signs, which the hardware
// applies for free through op_sel (swap the halves of a tap pair) and neg_lo / neg_hi.  Tap pairs pinned in SGPRs ("+s" through an
// empty asm) stay there across the plane loop.  The statements are not volatile: the compiler still schedules them.
struct PkF32 {
    typedef VecT<float>::v2 v2;
    static NDWT_DEV v2 pinned(v2 t) {
#ifndef NDWT_HOST_EMU
        asm volatile("" : "+s"(t));
#endif
        return t;
    }
    // acc += (a[SELA], a[SELA]) * (t0, t1) with (t0, t1) = tp, or (tp[1], tp[0]) if SWAP; NEGLO / NEGHI negate t0 / t1.
    template <int SELA, bool SWAP, bool NEGLO, bool NEGHI> static NDWT_DEV void fma_bt(v2& acc, const v2 a, const v2 tp) {
#ifndef NDWT_HOST_EMU
#define NDWT_PKF(A, S, NS, NL, NH)                                                                                                  \
        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[" #A "," #S ",0] op_sel_hi:[" #A "," #NS ",1] neg_lo:[0," #NL ",0] neg_hi:[0," #NH ",0]" \
            : "+v"(acc) : "v"(a), "s"(tp))
        if constexpr (SELA == 0 && !SWAP) NDWT_PKF(0, 0, 1, 0, 0);
        else if constexpr (SELA == 1 && !SWAP) NDWT_PKF(1, 0, 1, 0, 0);
        else if constexpr (SELA == 0 && NEGLO) NDWT_PKF(0, 1, 0, 1, 0);
        else if constexpr (SELA == 1 && NEGLO) NDWT_PKF(1, 1, 0, 1, 0);
        else if constexpr (SELA == 0) NDWT_PKF(0, 1, 0, 0, 1);
        else NDWT_PKF(1, 1, 0, 0, 1);
#undef NDWT_PKF
        static_assert(SWAP ? (NEGLO != NEGHI) : (!NEGLO && !NEGHI), "forms used by the x stages");
#else
        acc.x += a[SELA] * (NEGLO ? -tp[SWAP ? 1 : 0] : tp[SWAP ? 1 : 0]);
        acc.y += a[SELA] * (NEGHI ? -tp[SWAP ? 0 : 1] : tp[SWAP ? 0 : 1]);
#endif
    }
    // acc = (a[SELA], a[SELA]) * (t0, t1): the first term of a sum (fma_bt without the zeroed accumulator it would need)
    template <int SELA, bool SWAP, bool NEGLO, bool NEGHI> static NDWT_DEV void mul_bt(v2& acc, const v2 a, const v2 tp) {
#ifndef NDWT_HOST_EMU
#define NDWT_PKM(A, S, NS, NL, NH)                                                                                          \
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[" #A "," #S "] op_sel_hi:[" #A "," #NS "] neg_lo:[0," #NL "] neg_hi:[0," #NH "]" \
            : "=v"(acc) : "v"(a), "s"(tp))
        if constexpr (SELA == 0 && !SWAP) NDWT_PKM(0, 0, 1, 0, 0);
        else if constexpr (SELA == 1 && !SWAP) NDWT_PKM(1, 0, 1, 0, 0);
        else if constexpr (SELA == 0 && NEGLO) NDWT_PKM(0, 1, 0, 1, 0);
        else if constexpr (SELA == 1 && NEGLO) NDWT_PKM(1, 1, 0, 1, 0);
        else if constexpr (SELA == 0) NDWT_PKM(0, 1, 0, 0, 1);
        else NDWT_PKM(1, 1, 0, 0, 1);
#undef NDWT_PKM
        static_assert(SWAP ? (NEGLO != NEGHI) : (!NEGLO && !NEGHI), "forms used by the x stages");
#else
        acc.x = a[SELA] * (NEGLO ? -tp[SWAP ? 1 : 0] : tp[SWAP ? 1 : 0]);
        acc.y = a[SELA] * (NEGHI ? -tp[SWAP ? 0 : 1] : tp[SWAP ? 0 : 1]);
#endif
    }
    // acc += x * (t, t), t = tp[HI], negated if NEG
    template <int HI, bool NEG> static NDWT_DEV void fma_s(v2& acc, const v2 x, const v2 tp) {
#ifndef NDWT_HOST_EMU
        if constexpr (HI == 0 && !NEG) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "s"(tp));
        else if constexpr (HI == 1 && !NEG) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(x), "s"(tp));
        else if constexpr (HI == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(acc) : "v"(x), "s"(tp));
        else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,1,0] neg_hi:[0,1,0]" : "+v"(acc) : "v"(x), "s"(tp));
#else
        acc += x * (NEG ? -tp[HI] : tp[HI]);
#endif
    }
    // acc = x * (t, t), t = tp[HI], negated if NEG: the first term of a sum
    template <int HI, bool NEG> static NDWT_DEV void mul_s(v2& acc, const v2 x, const v2 tp) {
#ifndef NDWT_HOST_EMU
        if constexpr (HI == 0 && !NEG) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(acc) : "v"(x), "s"(tp));
        else if constexpr (HI == 1 && !NEG) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(acc) : "v"(x), "s"(tp));
        else if constexpr (HI == 0) asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(acc) : "v"(x), "s"(tp));
        else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(acc) : "v"(x), "s"(tp));
#else
        acc = x * (NEG ? -tp[HI] : tp[HI]);
#endif
    }
    // acc += x * ANALYSIS low-pass tap J (HIGH = false) or high-pass tap J = (-1)^J low-pass tap L-1-J, from the pairs lo[m] = (t[2m], t[2m+1])
    template <int L, int J, bool HIGH> static NDWT_DEV void tap_ana(v2& acc, const v2 x, const v2 (&lo)[L / 2]) {
        constexpr int jj = HIGH ? L - 1 - J : J;
        fma_s<jj & 1, HIGH && (J % 2 == 1)>(acc, x, lo[jj / 2]);
    }
};

// ---------------------------------------------------------------------------------- analysis ----
// EW = scalars per element along x: 1 real, 2 interleaved complex (n1 then counts scalars and the x taps step over
// (re, im) pairs; the y and z stages are component-wise and do not change)
// LOWONLY_: only band 0 (the approximation) is computed through and stored -- the analysis that feeds the deeper levels of a
// denoising step whose finest level never materialises its detail bands (Den3)
// TPRE_: the t axis of a 4-D level folded in (Fused3Args::tt): the raw plane of frame t is the t-filtered combination of L frames,
// so a 4-D level needs no pass of its own over the data for the t axis (17 volume transfers instead of 21: the t pass wrote two
// volumes and the fused launches read them back)
// PIN_ (float): the taps as pairs pinned in SGPRs for the whole plane loop, the high-pass ones derived from the low-pass ones by the operand
// modifiers of the packed FMA (PkF32) -- 4 L SGPRs instead of 12 L, so nothing is re-loaded from constant memory inside the loop (the
// plain form issues 27 scalar loads per plane with 8 taps, 40 with 12).  Needs even zero padding of every axis' taps (the host checks).
template <typename T, int L_, int TX_, int TY_, int NT_, int RY_, bool VEC4_, int WPE_ = 2, int EW_ = 1, bool LOWONLY_ = false, bool TPRE_ = false,
          bool PIN_ = false, int WLDS_ = 0> struct Fwd3 {
    static constexpr int L = L_, TX = TX_, TY = TY_, NT = NT_, RY = RY_, EW = EW_;
    static constexpr bool LOWONLY = LOWONLY_, TPRE = TPRE_, PIN = PIN_;
    // WLDS_ > 0: that many of the L slots of a thread's z window live in LDS instead of registers (each is written and read by its own
    // thread only: no barrier) -- what keeps 16 .. 20 taps inside the 128 registers of the 1024-thread tile without spills
    static constexpr int WLDS = WLDS_;
    static_assert(WLDS_ >= 0 && WLDS_ < L_, "at least one window slot stays in registers");
    static_assert(!PIN_ || sizeof(T) == 4, "pinned / derived taps: float only (v_pk_fma_f32)");
    static_assert(!TPRE_ || VEC4_, "the folded t axis exists for rows of whole groups of 4 scalars");
    static constexpr bool VEC4 = VEC4_;
    static constexpr int NE = VEC4 ? 1 : 4;              // offsets kept per column
    static constexpr int WPE = WPE_;                     // waves per SIMD the register budget is sized for
    static constexpr int LH = L / 2 - 1;                 // samples left of the output index
    static constexpr int RH = L / 2;                     // samples right of it
    static constexpr int GL = (LH * EW + 3) / 4, GR = (RH * EW + 3) / 4;   // halo in groups of 4 x
    static constexpr int W = TX + 4 * (GL + GR);         // haloed tile width (pairs per LDS row)
    static constexpr int NG = W / 4;
    static constexpr int NR = TY + L - 1;                // haloed tile rows
    static constexpr int NCOLS = NG * NR;                // z-stage columns (4 x each)
    static constexpr int NCOL = (NCOLS + NT - 1) / NT;   // columns per thread
    typedef Lds<T> LD;
    static constexpr int CH = LD::CH;
    static constexpr int WC = W / CH;                    // chunks per LDS row
    static constexpr int YITEMS = WC * (TY / RY);        // y-stage item: one chunk column, RY output rows
    static constexpr int NYI = (YITEMS + NT - 1) / NT;
    static constexpr int XITEMS = (TX / 4) * TY * 2;     // x-stage item: 4 x of one row of one y-bit plane
    static constexpr int NXI = (XITEMS + NT - 1) / NT;
    static constexpr int XV = 4 * (1 + GL + GR);         // pairs an x item reads
    static_assert(TX % 4 == 0 && TY % RY == 0 && L % 2 == 0 && W % 4 == 0, "tile shape");
    typedef typename VecT<T>::v2 v2;
    typedef typename VecT<T>::v4 v4;
    typedef typename LD::chunk chunk;
    typedef Taps3<T, L> Taps;
    typedef Fused3Args<T> Args;

    struct Shared {
        chunk zs[NR][WC];        // (lo3, hi3) of the raw tile
        chunk ys[TY][2][WC];     // [row][y-bit][x] of (z-bit 0, z-bit 1); y-bit planes of a row are a multiple of 256 B apart
        v4 zw[WLDS ? WLDS : 1][WLDS ? NCOL * NT : 1];   // [slot][column of the thread]: the window slots that do not live in registers
    };
    struct State {
        v4 win[NCOL][L - WLDS];   // raw samples of the last L planes, rotating (slots WLDS .. L-1; slots 0 .. WLDS-1: Shared::zw)
        v4 nxt[NCOL];        // prefetched plane
        v4 tfr[TPRE ? L : 1][NCOL];   // TPRE: the prefetched plane of the L frames under the t filter
        int off[NCOL][NE];
    };
    struct TFrames { long long off[L]; };                // TPRE: element offsets of the L frames under the t filter of this workgroup's frame
    struct RegT {                                        // PIN: tap pairs in SGPRs
        v2 ax[PIN ? L / 2 : 1], ay[PIN ? L / 2 : 1];     // (lo[2m], lo[2m+1]) of the x and y axes
        v2 az[PIN ? L : 1];                              // (lo_z[j], hi_z[j])
    };
    template <int SLOT> static NDWT_DEV v4 win_get(const State& st, const Shared& sh, int k, int tid) {
        if constexpr (SLOT < WLDS) return sh.zw[SLOT][k * NT + tid];
        else return st.win[k][SLOT - WLDS];
    }
    template <int SLOT> static NDWT_DEV void win_put(State& st, Shared& sh, int k, int tid, v4 v) {
        if constexpr (SLOT < WLDS) sh.zw[SLOT][k * NT + tid] = v;
        else st.win[k][SLOT - WLDS] = v;
    }
    static NDWT_DEV void load_regt(RegT& rt, const Taps& tp) {
        if constexpr (PIN) {
            NDWT_SFOR(m, L / 2)
                rt.ax[m] = PkF32::pinned(v2{tp.lo[0][2 * m], tp.lo[0][2 * m + 1]});
                rt.ay[m] = PkF32::pinned(v2{tp.lo[1][2 * m], tp.lo[1][2 * m + 1]});
            NDWT_SEND
            NDWT_SFOR(j, L)
                rt.az[j] = PkF32::pinned(v2{tp.lo[2][j], tp.hi[2][j]});
            NDWT_SEND
        }
    }

    static NDWT_DEV void setup(State& st, const Args& a, const TileCoord& tc, int tid) {
        NDWT_SFOR(k, NCOL)
            int c = tid + k * NT;
            if (c >= NCOLS) c = NCOLS - 1;               // surplus lanes load a valid column, never store
            int ug = c % NG, r = c / NG;
            int y = modn(tc.y0 - LH + r, a.n2);
            int xb = tc.x0 - 4 * GL + 4 * ug;
            NDWT_SFOR(e, NE)
                st.off[k][e] = y * a.rs + modn(xb + e, a.n1);
            NDWT_SEND
        NDWT_SEND
    }

    // TPRE: issue the loads of plane zraw of the L frames; tcombine() turns them into the t-filtered raw plane when it is consumed
    static NDWT_DEV void load_frames(State& st, const Args& a, const TFrames& tf, int zraw) {
        const long long zm = (long long)modn(zraw, a.n3);
        NDWT_SFOR(j, (TPRE ? L : 0))
            const T* p = a.in[0] + tf.off[j] + zm * a.plane;
            NDWT_SFOR(k, NCOL)
                st.tfr[j][k] = *reinterpret_cast<const v4*>(p + st.off[k][0]);
            NDWT_SEND
        NDWT_SEND
    }
    static NDWT_DEV void tcombine(State& st, const Args& a) {
        NDWT_SFOR(k, NCOL)
            v4 acc = (v4)(T(0));
            NDWT_SFOR(j, (TPRE ? L : 0))
                acc += a.tt[j] * st.tfr[j][k];
            NDWT_SEND
            st.nxt[k] = acc;
        NDWT_SEND
    }

    static NDWT_DEV void load_plane(State& st, const Args& a, const T* inb, int zraw) {
        const T* p;
        if (a.z_wrap == 2) {                             // halo planes live in their own (received) buffers
            if (zraw < 0) p = a.in[1] + (long long)(zraw + LH) * a.plane;
            else if (zraw >= a.n3) p = a.in[2] + (long long)(zraw - a.n3) * a.plane;
            else p = inb + (long long)zraw * a.plane;
        } else {
            long long zm = a.z_wrap ? (long long)modn(zraw, a.n3) : (long long)(zraw + LH);
            p = inb + zm * a.plane;
        }
        NDWT_SFOR(k, NCOL)
            if constexpr (VEC4) {
                st.nxt[k] = *reinterpret_cast<const v4*>(p + st.off[k][0]);
            } else {
                if (st.off[k][NE - 1] == st.off[k][0] + 3) {          // 4 contiguous x: one access (VecT::v4u)
                    st.nxt[k] = *reinterpret_cast<const typename VecT<T>::v4u*>(p + st.off[k][0]);
                } else {
                    NDWT_SFOR(e, NE)
                        st.nxt[k][e] = p[st.off[k][e]];
                    NDWT_SEND
                }
            }
        NDWT_SEND
    }

    // rotation R: the newest plane lands in slot (R+L-1)%L; tap j reads slot (R+j)%L
    template <int R> static NDWT_DEV void zstage(State& st, Shared& sh, const Taps& tp, const RegT& rt, int tid) {
        NDWT_SFOR(k, NCOL)
            win_put<(R + L - 1) % L>(st, sh, k, tid, st.nxt[k]);
            int c = tid + k * NT;
            if (c < NCOLS) {
                v2 acc[4];
                acc[0] = acc[1] = acc[2] = acc[3] = (v2)(T(0));
                NDWT_SFOR(j, L)
                    v4 w = (j == L - 1) ? st.nxt[k] : win_get<(R + j) % L>(st, sh, k, tid);   // (the newest plane is still in registers)
                    if constexpr (PIN) {                  // (lo, hi) += w[e] * (lo_z[j], hi_z[j]): the sample broadcast from a half of its register pair
                        const v2 w01 = {w[0], w[1]}, w23 = {w[2], w[3]};
                        PkF32::fma_bt<0, false, false, false>(acc[0], w01, rt.az[j]);
                        PkF32::fma_bt<1, false, false, false>(acc[1], w01, rt.az[j]);
                        PkF32::fma_bt<0, false, false, false>(acc[2], w23, rt.az[j]);
                        PkF32::fma_bt<1, false, false, false>(acc[3], w23, rt.az[j]);
                    } else {
                        v2 t = {tp.lo[2][j], tp.hi[2][j]};
                        acc[0] += t * w[0]; acc[1] += t * w[1]; acc[2] += t * w[2]; acc[3] += t * w[3];
                    }
                NDWT_SEND
                int ug = c % NG, r = c / NG;
                lds_store_run<T, 4>(sh.zs[r], 4 * ug, acc);
            }
        NDWT_SEND
    }
    template <int R> static NDWT_DEV void zdispatch(int r, State& st, Shared& sh, const Taps& tp, const RegT& rt, int tid) {
        if constexpr (R < L) {
            if (r == R) zstage<R>(st, sh, tp, rt, tid);
            else zdispatch<R + 1>(r, st, sh, tp, rt, tid);
        }
    }
    static NDWT_DEV void prologue(State& st, Shared& sh, const Args& a, const T* inb, int zbeg, int tid) {
        NDWT_SFOR(j, L - 1)
            load_plane(st, a, inb, zbeg - LH + j);
            NDWT_SFOR(k, NCOL)
                win_put<j>(st, sh, k, tid, st.nxt[k]);
            NDWT_SEND
        NDWT_SEND
    }

    // y filter: an item owns one chunk column (CH adjacent x) and RY output rows.  The RY + L - 1 input chunks are read YGRP at a
    // time and scattered into the RY outputs they feed (the register footprint no longer grows with the tap length: what kept
    // tap lengths above 12 off the fused kernels)
    static constexpr int YGRP = 4;
    static NDWT_DEV void ystage(Shared& sh, const Taps& tp, const RegT& rt, int tid) {
        NDWT_UNROLL
        for (int k = 0; k < NYI; ++k) {
            int it = tid + k * NT;
            if (it >= YITEMS) continue;
            int cc = it % WC, yg = it / WC;
            const int pc = LD::S(cc);
            v2 l[RY][CH], h[RY][CH];
            NDWT_SFOR(i, RY)
                NDWT_SFOR(sub, CH)
                    l[i][sub] = (v2)(T(0));
                    h[i][sub] = (v2)(T(0));
                NDWT_SEND
            NDWT_SEND
            NDWT_SFOR(g, (RY + L - 1 + YGRP - 1) / YGRP)
                chunk zin[YGRP];
                NDWT_SFOR(t, YGRP)
                    if constexpr (g * YGRP + t < RY + L - 1) zin[t] = sh.zs[yg * RY + g * YGRP + t][pc];
                NDWT_SEND
                NDWT_SFOR(t, YGRP)
                    constexpr int r = g * YGRP + t;
                    if constexpr (r < RY + L - 1) {
                        NDWT_SFOR(i, RY)
                            constexpr int j = r - i;
                            if constexpr (j >= 0 && j < L) {
                                NDWT_SFOR(sub, CH)
                                    const v2 z = LD::get(zin[t], sub);
                                    if constexpr (PIN) {
                                        PkF32::tap_ana<L, j, false>(l[i][sub], z, rt.ay);
                                        if constexpr (!LOWONLY) PkF32::tap_ana<L, j, true>(h[i][sub], z, rt.ay);
                                    } else {
                                        l[i][sub] += tp.lo[1][j] * z;
                                        if constexpr (!LOWONLY) h[i][sub] += tp.hi[1][j] * z;
                                    }
                                NDWT_SEND
                            }
                        NDWT_SEND
                    }
                NDWT_SEND
                if constexpr (L > 12) NDWT_SCHED_FENCE();  // long filters: keep the groups apart (hipcc hoists every read otherwise)
            NDWT_SEND
            NDWT_SFOR(i, RY)
                chunk lo, hi;
                NDWT_SFOR(sub, CH)
                    LD::set(lo, sub, l[i][sub]);
                    LD::set(hi, sub, h[i][sub]);
                NDWT_SEND
                sh.ys[yg * RY + i][0][pc] = lo;
                if constexpr (!LOWONLY) sh.ys[yg * RY + i][1][pc] = hi;
            NDWT_SEND
        }
    }

    // x filter + stores: item order (4-x group fastest, then y-bit, then row): the two 16-lane halves of a
    // 32-lane group read the two y-bit planes of one row
    static NDWT_DEV void xstage(Shared& sh, const Taps& tp, const RegT& rt, const Args& a, const TileCoord& tc, long long obase, int z,
                                int tid) {
        NDWT_UNROLL
        for (int k = 0; k < NXI; ++k) {
            int it = tid + k * NT;
            if (it >= XITEMS) continue;
            int xg = it % (TX / 4);
            int q = (it / (TX / 4)) % 2;
            int y = it / ((TX / 4) * 2);
            int gy = tc.y0 + y, gx = tc.x0 + 4 * xg;
            if constexpr (LOWONLY) {
                if (q != 0) continue;                     // the y-high plane feeds detail bands only
            }
            v2 v[XV];
            lds_load_run<T, XV>(sh.ys[y][q], 4 * xg, v);
            if (gy >= a.n2 || gx >= a.n1) continue;
            v4 o00, o01, o10, o11;                        // [x-bit][z-bit]
            NDWT_UNROLL
            for (int e = 0; e < 4; ++e) {
                v2 lo = (v2)(T(0)), hi = (v2)(T(0));
                if constexpr (PIN) {
                    NDWT_SFOR(j, L)
                        PkF32::tap_ana<L, j, false>(lo, v[4 * GL + e + (j - LH) * EW], rt.ax);
                        if constexpr (!LOWONLY) PkF32::tap_ana<L, j, true>(hi, v[4 * GL + e + (j - LH) * EW], rt.ax);
                    NDWT_SEND
                } else {
                    NDWT_UNROLL
                    for (int j = 0; j < L; ++j) {
                        lo += tp.lo[0][j] * v[4 * GL + e + (j - LH) * EW];
                        if constexpr (!LOWONLY) hi += tp.hi[0][j] * v[4 * GL + e + (j - LH) * EW];
                    }
                }
                o00[e] = lo.x; o01[e] = lo.y; o10[e] = hi.x; o11[e] = hi.y;
            }
            long long off = obase + (long long)z * a.plane + (long long)gy * a.rs + gx;
            if constexpr (LOWONLY) {
                T* b0 = a.out[0] + off;
                if constexpr (VEC4) {
                    stream_store(reinterpret_cast<v4*>(b0), o00, a.nt);
                } else if (gx + 3 < a.n1) {
                    *reinterpret_cast<typename VecT<T>::v4u*>(b0) = o00;
                } else {
                    NDWT_UNROLL
                    for (int e = 0; e < 4; ++e)
                        if (gx + e < a.n1) b0[e] = o00[e];
                }
                continue;
            }
            // Indexing the kernel-argument array with the per-lane q makes hipcc FETCH the four pointers with vector loads in every plane
            // and wait vmcnt(0) for them -- i.e. also for the next plane's prefetch -- before the stores.  Selects avoid that: double
            // -5 % (512^3) .. -10 % (256^3) per launch.  Float runs 1.7 % SLOWER without that wait (interleaved A/B, 512^3: 0.902 vs
            // 0.917 ms; the wait keeps the waves of a workgroup in step, as DESIGN_HISTORY.md notes for the synthesis kernel) and keeps it.
            T *b00, *b10, *b01, *b11;
            if constexpr (sizeof(T) == 8) {
                b00 = (q ? a.out[2] : a.out[0]) + off;
                b10 = (q ? a.out[3] : a.out[1]) + off;
                b01 = (q ? a.out[6] : a.out[4]) + off;
                b11 = (q ? a.out[7] : a.out[5]) + off;
            } else {
                b00 = a.out[2 * q] + off;
                b10 = a.out[2 * q + 1] + off;
                b01 = a.out[2 * q + 4] + off;
                b11 = a.out[2 * q + 5] + off;
            }
            if constexpr (VEC4) {
                stream_store(reinterpret_cast<v4*>(b00), o00, a.nt);
                stream_store(reinterpret_cast<v4*>(b10), o10, a.nt);
                stream_store(reinterpret_cast<v4*>(b01), o01, a.nt);
                stream_store(reinterpret_cast<v4*>(b11), o11, a.nt);
            } else if (gx + 3 < a.n1) {
                typedef typename VecT<T>::v4u v4u;
                *reinterpret_cast<v4u*>(b00) = o00;
                *reinterpret_cast<v4u*>(b10) = o10;
                *reinterpret_cast<v4u*>(b01) = o01;
                *reinterpret_cast<v4u*>(b11) = o11;
            } else {
                NDWT_UNROLL
                for (int e = 0; e < 4; ++e) {
                    if (gx + e < a.n1) { b00[e] = o00[e]; b10[e] = o10[e]; b01[e] = o01[e]; b11[e] = o11[e]; }
                }
            }
        }
    }

    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared& sh, const Args& a, const Taps& tp, int bid) {
        const TileCoord tc = decode_tile<T, TPRE>(a, bid, TX, TY, VEC4 ? -1 : 4 * GL, 4 * GR);
        const T* inb = a.in[0] + batch_base(tc.batch, a.bsplit, a.in_bstride, a.in_bstride2);
        const long long obase = batch_base(tc.batch, a.bsplit, a.out_bstride, a.out_bstride2);
        RegT rt;
        load_regt(rt, tp);
        TFrames tf;
        if constexpr (TPRE) {                             // frames t - LH .. t + RH of the periodic t axis (the batch index)
            NDWT_SFOR(j, L)
                tf.off[j] = (long long)modn(tc.batch - LH + j, a.nbatch) * a.in_bstride;
            NDWT_SEND
        }
        auto fetch = [&](State& st, int zraw) __attribute__((always_inline)) {
            if constexpr (TPRE) load_frames(st, a, tf, zraw);
            else load_plane(st, a, inb, zraw);
        };
        // planes zbeg-LH .. zbeg-LH+L-2 into slots 0..L-2, then prefetch the plane of step 0
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            setup(st, a, tc, tid);
            if constexpr (TPRE) {
                NDWT_SFOR(j, L - 1)
                    load_frames(st, a, tf, tc.zbeg - LH + j);
                    tcombine(st, a);
                    NDWT_SFOR(k, NCOL)
                        win_put<j>(st, sh, k, tid, st.nxt[k]);
                    NDWT_SEND
                NDWT_SEND
            } else {
                prologue(st, sh, a, inb, tc.zbeg, tid);
            }
            fetch(st, tc.zbeg + RH);
        });
        const int nsteps = tc.zend - tc.zbeg;
        for (int s = 0; s < nsteps; ++s) {
            const int z = tc.zbeg + s;
            ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                if constexpr (TPRE) tcombine(st, a);                      // the prefetched frames -> the t-filtered raw plane
                zdispatch<0>(s % L, st, sh, tp, rt, tid);                 // consumes st.nxt
                if (s + 1 < nsteps) fetch(st, z + 1 + RH);                // prefetch for the next step
            });
            ex.barrier();
            NDWT_SETPRIO(1);                              // the stages that end in this plane's stores go ahead of the other
            ex.each([&](int tid, State&) __attribute__((always_inline)) { ystage(sh, tp, rt, tid); });   // waves' loads (-2 %)
            ex.barrier();
            ex.each([&](int tid, State&) __attribute__((always_inline)) { xstage(sh, tp, rt, a, tc, obase, z, tid); });
            NDWT_SETPRIO(0);
        }
    }
};

// --------------------------------------------------------------------------------- synthesis ----
// x-synthesis and y-synthesis go through LDS on the haloed tile (the 2^3 bands are read with an
// x/y halo, mostly from L2), the z-synthesis window (L planes of (a,d) pairs) stays in registers.
// Per new plane: for y-bit 0,1 { raw 4 bands -> LDS; x-synth -> xs[y-bit] } ; y-synth -> P ; z-synth.
template <typename T, int L_, int TX_, int TY_, int NT_, int RY_, bool VEC4_, int WPE_ = 2, int EW_ = 1> struct Inv3 {
    static_assert(EW_ == 1, "the LDS synthesis kernel handles real data only");
    static constexpr int L = L_, TX = TX_, TY = TY_, NT = NT_, RY = RY_;
    static constexpr bool VEC4 = VEC4_;
    static constexpr int NE = VEC4 ? 1 : 4;
    static constexpr int WPE = WPE_;                     // waves per SIMD the register budget is sized for
    static constexpr int LH = L / 2;                     // synthesis: samples left of the output index
    static constexpr int RH = L / 2 - 1;                 // samples right of it
    static constexpr int GL = (LH + 3) / 4, GR = (RH + 3) / 4;
    static constexpr int W = TX + 4 * (GL + GR);
    static constexpr int NG = W / 4;
    static constexpr int NR = TY + L - 1;
    typedef Lds<T> LD;
    static constexpr int CH = LD::CH;
    static constexpr int WC = W / CH, TXC = TX / CH;     // chunks per row of the raw / x-synthesised tiles
    typedef RowPair<WC * 16> RP;                         // row pairing of the x-synthesis items
    static constexpr int LITEMS = 2 * NG * NR;           // load items: (x-bit, row, group of 4 x)
    static constexpr int NLI = (LITEMS + NT - 1) / NT;
    static constexpr int XITEMS = (TX / 4) * RP::slots(NR);   // x-synthesis items per y-bit: 4 x of one row
    static constexpr int NXI = (XITEMS + NT - 1) / NT;
    static constexpr int YITEMS = TXC * (TY / RY);       // y/z-synthesis item: one chunk column (CH x), RY rows
    static constexpr int NYI = (YITEMS + NT - 1) / NT;
    static constexpr int NP = CH * RY;                   // output positions (window columns) per item
    static constexpr int XV = 4 * (1 + GL + GR);
    static_assert(TX % 4 == 0 && TY % RY == 0 && L % 2 == 0 && W % 4 == 0, "tile shape");
    typedef typename VecT<T>::v2 v2;
    typedef typename VecT<T>::v4 v4;
    typedef typename LD::chunk chunk;
    typedef Taps3<T, L> Taps;
    typedef Fused3Args<T> Args;

    struct Shared {
        chunk raw[2][NR][WC];    // [x-bit][row][x] of (z-bit 0, z-bit 1), one y-bit at a time
        chunk xs[2][NR][TXC];    // [y-bit][row][x] after x-synthesis
    };
    struct State {
        v2 win[NYI][NP][L];  // (a,d) pairs for the z-synthesis, rotating
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
                st.off[k][e] = y * a.rs + modn(xb + e, a.n1);
            NDWT_SEND
        NDWT_SEND
    }

    // issue the global loads of (plane zraw, y-bit yb) into st.pre
    static NDWT_DEV void load_raw(State& st, const Args& a, long long ibase, int zraw, int yb, int tid, int zsh) {
        long long zm = a.z_wrap == 1 ? (long long)modn(zraw, a.n3) : (long long)(zraw + LH);
        if (a.z_wrap == 3) {                             // zero-extended slab: output plane k needs inputs k-(L-1)+j
            zm = zraw - RH;
            if (zm + zsh < a.zlo || zm + zsh >= a.zhi) {
                NDWT_SFOR(k, NLI)
                    st.pre[k][0] = (v4)(T(0));
                    st.pre[k][1] = (v4)(T(0));
                NDWT_SEND
                return;
            }
        }
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
                v4 p0 = st.pre[k][0], p1 = st.pre[k][1];
                v2 pr[4] = {v2{p0[0], p1[0]}, v2{p0[1], p1[1]}, v2{p0[2], p1[2]}, v2{p0[3], p1[3]}};
                lds_store_run<T, 4>(sh.raw[xb][r], 4 * ug, pr);
            }
        NDWT_SEND
    }

    static NDWT_DEV void xsyn(Shared& sh, const Taps& tp, int yb, int tid) {
        NDWT_UNROLL
        for (int k = 0; k < NXI; ++k) {
            int it = tid + k * NT;
            if (it >= XITEMS) continue;
            int xg = it % (TX / 4), r = RP::row(it / (TX / 4));
            if (r >= NR) continue;
            v2 av[XV], dv[XV];
            lds_load_run<T, XV>(sh.raw[0][r], 4 * xg, av);
            lds_load_run<T, XV>(sh.raw[1][r], 4 * xg, dv);
            v2 acc[4];
            NDWT_UNROLL
            for (int e = 0; e < 4; ++e) {
                acc[e] = (v2)(T(0));
                NDWT_UNROLL
                for (int j = 0; j < L; ++j) {
                    acc[e] += tp.lo[0][j] * av[4 * GL + e - LH + j];
                    acc[e] += tp.hi[0][j] * dv[4 * GL + e - LH + j];
                }
            }
            lds_store_run<T, 4>(sh.xs[yb][r], 4 * xg, acc);
        }
    }

    // y-synthesis of the new plane into window slot (R+L-1)%L; if `emit`, z-synthesis of plane z and store
    template <int R>
    static NDWT_DEV void yzsyn(State& st, Shared& sh, const Taps& tp, const Args& a, const TileCoord& tc, long long obase,
                               int z, bool emit, int tid) {
        NDWT_SFOR(k, NYI)
            int it = tid + k * NT;
            if (it < YITEMS) {
                int cx = it % TXC, yg = it / TXC;
                const int pc = LD::S(cx);
                chunk av[RY + L - 1], dv[RY + L - 1];
                NDWT_UNROLL
                for (int t = 0; t < RY + L - 1; ++t) {
                    av[t] = sh.xs[0][yg * RY + t][pc];
                    dv[t] = sh.xs[1][yg * RY + t][pc];
                }
                NDWT_SFOR(i, RY)
                    NDWT_SFOR(sub, CH)
                        v2 acc = (v2)(T(0));
                        NDWT_UNROLL
                        for (int j = 0; j < L; ++j) {
                            acc += tp.lo[1][j] * LD::get(av[i + j], sub);
                            acc += tp.hi[1][j] * LD::get(dv[i + j], sub);
                        }
                        st.win[k][i * CH + sub][(R + L - 1) % L] = acc;
                    NDWT_SEND
                NDWT_SEND
                if (emit) {
                    int gx = tc.x0 + cx * CH;
                    NDWT_SFOR(i, RY)
                        T o[CH];
                        NDWT_SFOR(sub, CH)
                            v2 acc = (v2)(T(0));
                            NDWT_SFOR(j, L)
                                v2 t = {tp.lo[2][j], tp.hi[2][j]};
                                acc += t * st.win[k][i * CH + sub][(R + j) % L];
                            NDWT_SEND
                            o[sub] = acc.x + acc.y;
                        NDWT_SEND
                        int gy = tc.y0 + yg * RY + i;
                        if (gy < a.n2) {
                            T* dst = a.out[0] + obase + (long long)z * a.plane + (long long)gy * a.rs + gx;
                            if constexpr (VEC4 && CH == 2) {
                                if (gx < a.n1) stream_store(reinterpret_cast<v2*>(dst), v2{o[0], o[CH - 1]}, a.nt);
                            } else {
                                NDWT_SFOR(sub, CH)
                                    if (gx + sub < a.n1) dst[sub] = o[sub];
                                NDWT_SEND
                            }
                        }
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
        const TileCoord tc = decode_tile(a, bid, TX, TY, VEC4 ? -1 : 4 * GL, 4 * GR);
        const long long ibase = batch_base(tc.batch, a.bsplit, a.in_bstride, a.in_bstride2);
        const long long obase = batch_base(tc.batch, a.bsplit, a.out_bstride, a.out_bstride2);
        const int zsh = tc.batch * a.zbs;
        const int nsteps = tc.zend - tc.zbeg;
        const int nplanes = nsteps + L - 1;              // planes zbeg-LH .. zend-1+RH
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            setup(st, a, tc, tid);
            load_raw(st, a, ibase, tc.zbeg - LH, 0, tid, zsh);
        });
        for (int p = 0; p < nplanes; ++p) {
            const int zraw = tc.zbeg - LH + p;
            const int s = p - (L - 1);                   // output step this plane completes (if >= 0)
            for (int yb = 0; yb < 2; ++yb) {
                ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                    stash_raw(st, sh, tid);
                    if (yb == 0) load_raw(st, a, ibase, zraw, 1, tid, zsh);
                    else if (p + 1 < nplanes) load_raw(st, a, ibase, zraw + 1, 0, tid, zsh);
                });
                ex.barrier();
                ex.each([&](int tid, State&) __attribute__((always_inline)) { xsyn(sh, tp, yb, tid); });
                ex.barrier();
            }
            ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                // plane p lives in slot p%L; rotation R puts the newest into (R+L-1)%L -> R = (p+1)%L
                yzdispatch<0>((p + 1) % L, st, sh, tp, a, tc, obase, tc.zbeg + s, s >= 0, tid);
            });
        }
    }
};

// ------------------------------------------------------------------- synthesis, lane-shift form ----
// Same arithmetic as Inv3 with a different data path: the x-synthesis takes its x neighbours straight from the
// adjacent lanes' registers (DPP wave shifts) instead of a raw tile in LDS.  A wave holds whole haloed rows
// (NG = TX/4 + halo groups lanes per row, RPW rows per wave), every lane loads 4 x of all 2^3 bands of one row.
// LDS only carries the x-synthesised tile, double buffered: ONE barrier per plane and no y-bit phases.
//   per plane:  x-synth(p) from registers -> xs[p&1] ; prefetch raw(p+1) ; barrier ; y-synth + z-synth(p) from xs[p&1]
#ifdef NDWT_HOST_EMU
#define NDWT_LANE_SHIFT(ex, tid, D, expr_of_s) ((ex).template peer_value<D>((tid), [&](const State& s) { return (expr_of_s); }))
#else
template <typename T> __device__ __forceinline__ T dpp_shr1(T v);   // lane i <- lane i-1 (0 into lane 0)
template <typename T> __device__ __forceinline__ T dpp_shl1(T v);   // lane i <- lane i+1
template <> __device__ __forceinline__ float dpp_shr1<float>(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
template <> __device__ __forceinline__ float dpp_shl1<float>(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}
template <> __device__ __forceinline__ double dpp_shr1<double>(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
template <> __device__ __forceinline__ double dpp_shl1<double>(double v) {
    long long b = __builtin_bit_cast(long long, v);
    int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xF, 0xF, true);
    int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// value of `v` in lane (i + D)
template <int D, typename T> __device__ __forceinline__ T lane_shift(T v) {
    if constexpr (D == 0) return v;
    else if constexpr (D < 0) return lane_shift<D + 1>(dpp_shr1<T>(v));
    else return lane_shift<D - 1>(dpp_shl1<T>(v));
}
#define NDWT_LANE_SHIFT(ex, tid, D, expr_of_s) (lane_shift<D>([&](const State& s) { return (expr_of_s); }(st)))
#endif

template <typename T, int L_, int TX_, int TY_, int NT_, int RY_, bool VEC4_, int WPE_ = 2, int EW_ = 1> struct Inv3S {
    static constexpr int L = L_, TX = TX_, TY = TY_, NT = NT_, RY = RY_, EW = EW_;
    static constexpr bool VEC4 = VEC4_;
    static constexpr int NE = VEC4 ? 1 : 4;
    static constexpr int WPE = WPE_;
    static constexpr int LH = L / 2, RH = L / 2 - 1;
    static constexpr int GL = (LH * EW + 3) / 4, GR = (RH * EW + 3) / 4;
    static constexpr int NG = TX / 4 + GL + GR;          // lanes per haloed row
    static constexpr int NR = TY + L - 1;
    static constexpr int RPW = 64 / NG;                  // rows per wave
    static constexpr int NW = NT / 64;
    static constexpr int RPR = RPW * NW;                 // rows per round
    static constexpr int NRND = (NR + RPR - 1) / RPR;
    typedef Lds<T> LD;
    static constexpr int CH = LD::CH;
    static constexpr int TXC = TX / CH;
    static constexpr int YITEMS = TXC * (TY / RY);
    static constexpr int NYI = (YITEMS + NT - 1) / NT;
    static constexpr int NP = CH * RY;
    static constexpr int XV = 4 * (1 + GL + GR);
    static constexpr int YGRP = 4;                       // LDS chunks the y stage holds at a time
    static_assert(TX % 4 == 0 && TY % RY == 0 && L % 2 == 0 && NT % 64 == 0 && RPW >= 1, "tile shape");
    typedef typename VecT<T>::v2 v2;
    typedef typename VecT<T>::v4 v4;
    typedef typename LD::chunk chunk;
    typedef Taps3<T, L> Taps;
    typedef Fused3Args<T> Args;

    struct Shared {
        chunk xs[2][2][NR][TXC];   // [buffer][y-bit][row][x] of (z-bit 0, z-bit 1)
    };
    struct State {
        T zacc[NYI][L][NP];        // z-synthesis in scatter form: partial sums of the next L output planes, rotating
        v4 raw[NRND][8];           // 4 x of every band of this lane's row(s); refilled as soon as consumed
        int off[NRND][NE];
    };

    static NDWT_DEV void lane_item(int tid, int rnd, int& ug, int& r, bool& valid) {
        const int lane = tid % 64, wv = tid / 64;
        int rs = lane / NG;
        ug = lane % NG;
        r = rnd * RPR + wv * RPW + rs;
        valid = rs < RPW && r < NR;
        if (rs >= RPW) rs = RPW - 1;
        if (r >= NR) r = NR - 1;
    }

    static NDWT_DEV void setup(State& st, const Args& a, const TileCoord& tc, int tid) {
        NDWT_SFOR(k, NRND)
            int ug, r;
            bool valid;
            lane_item(tid, k, ug, r, valid);
            int y = modn(tc.y0 - LH + r, a.n2);
            int xb = tc.x0 - 4 * GL + 4 * ug;
            NDWT_SFOR(e, NE)
                st.off[k][e] = valid ? y * a.rs + modn(xb + e, a.n1) : -1;   // -1: this lane holds no row and loads nothing
            NDWT_SEND
            NDWT_SFOR(b, 8)
                st.raw[k][b] = (v4)(T(0));               // lanes without a row keep zeros (their values reach no output)
            NDWT_SEND
        NDWT_SEND
    }

    static NDWT_DEV void load_raw(State& st, const Args& a, long long ibase, int zraw, int zsh) {
        long long zm = a.z_wrap == 1 ? (long long)modn(zraw, a.n3) : (long long)(zraw + LH);
        if (a.z_wrap == 3) {                             // zero-extended slab
            zm = zraw - RH;
            if (zm + zsh < a.zlo || zm + zsh >= a.zhi) {
                NDWT_SFOR(k, NRND)
                    NDWT_SFOR(b, 8)
                        st.raw[k][b] = (v4)(T(0));
                    NDWT_SEND
                NDWT_SEND
                return;
            }
        }
        long long pb = ibase + zm * a.plane;
        NDWT_SFOR(k, NRND)
            // lanes (and whole waves) without a row issue no loads: the per-CU vector-memory pipe bounds these kernels, and the
            // waves past the last haloed row used to re-load a clamped row (3 of 16 waves on the 64x32 tile)
            if constexpr (VEC4) {
                if (st.off[k][0] >= 0) {
                    NDWT_SFOR(b, 8)
                        st.raw[k][b] = *reinterpret_cast<const v4*>(a.in[b] + pb + st.off[k][0]);
                    NDWT_SEND
                }
            } else {
                if (st.off[k][0] >= 0 && st.off[k][NE - 1] == st.off[k][0] + 3) {      // 4 contiguous x: one access per band (VecT::v4u)
                    NDWT_SFOR(b, 8)
                        st.raw[k][b] = *reinterpret_cast<const typename VecT<T>::v4u*>(a.in[b] + pb + st.off[k][0]);
                    NDWT_SEND
                } else if (st.off[k][0] >= 0) {
                    NDWT_SFOR(b, 8)
                        NDWT_SFOR(e, NE)
                            st.raw[k][b][e] = (a.in[b] + pb)[st.off[k][e]];
                        NDWT_SEND
                    NDWT_SEND
                }
            }
        NDWT_SEND
    }

    // thresholding fused into the reconstruction: every lane shrinks the band values it holds before the x-stage reads
    // them (its own and, through the lane shifts, its neighbours')
    static NDWT_DEV void shrink_raw(State& st, const Args& a) {
        if (a.shrink_mask == 0) return;
        NDWT_SFOR(k, NRND)
            NDWT_SFOR(b, 8)
                if ((a.shrink_mask >> b) & 1) shrink4<T, EW>(st.raw[k][b], a.shrink_thr, a.shrink_hard);
            NDWT_SEND
        NDWT_SEND
    }

    // all 64 lanes of every wave execute the shifts (no divergence before them); only the LDS store is predicated
    template <class Exec> static NDWT_DEV void xsyn(Exec& ex, State& st, Shared& sh, const Taps& tp, int buf, int tid) {
        NDWT_SFOR(k, NRND)
            int ug, r;
            bool valid;
            lane_item(tid, k, ug, r, valid);
            NDWT_SFOR(yb, 2)
                // Scalar accumulators per z-bit: the loaded registers feed the FMAs as they are.  (Accumulating (z-bit 0,
                // z-bit 1) pairs makes hipcc build the pairs with v_mov right behind the loads -- i.e. wait for the
                // next plane's loads before the barrier instead of leaving them in flight across the y/z stage;
                // packed FMAs have no rate advantage on CDNA4.  The translation unit is built with -fno-slp-vectorize
                // for the same reason: the SLP vectorizer would re-pack these.)
                T a0[4], a1[4];
                NDWT_SFOR(e, 4)
                    a0[e] = T(0);
                    a1[e] = T(0);
                NDWT_SEND
                NDWT_SFOR(i, XV)                          // window element i <-> x offset i - 4*GL from this lane's first x
                    constexpr int D = i / 4 - GL;         // lane distance
                    constexpr int c = i % 4;
                    // low-pass (x-bit 0) and high-pass (x-bit 1) inputs of z-bit 0 and z-bit 1
                    const T wa0 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[k][0 + 2 * yb][c]);
                    const T wd0 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[k][1 + 2 * yb][c]);
                    const T wa1 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[k][4 + 2 * yb][c]);
                    const T wd1 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[k][5 + 2 * yb][c]);
                    NDWT_SFOR(e, 4)
                        constexpr int dj = i - 4 * GL - e;                 // = (j - LH) * EW
                        if constexpr (dj % EW == 0) {
                            constexpr int j = dj / EW + LH;
                            if constexpr (j >= 0 && j < L) {
                                a0[e] += tp.lo[0][j] * wa0;
                                a0[e] += tp.hi[0][j] * wd0;
                                a1[e] += tp.lo[0][j] * wa1;
                                a1[e] += tp.hi[0][j] * wd1;
                            }
                        }
                    NDWT_SEND
                NDWT_SEND
                v2 acc[4];
                NDWT_SFOR(e, 4)
                    acc[e] = v2{a0[e], a1[e]};
                NDWT_SEND
                if (valid && ug >= GL && ug < GL + TX / 4) lds_store_run<T, 4>(sh.xs[buf][yb][r], 4 * (ug - GL), acc);
            NDWT_SEND
        NDWT_SEND
    }

    // y-synthesis of the newest plane (pairs P = (a, d)), then z-synthesis in scatter form: P adds tap j into the
    // partial sum of output plane (newest - j); rotation R keeps that sum in slot (R-1-j) mod L, the j = 0 slot is
    // (re)initialised and the j = L-1 slot completes output plane z.  (L-1 floats of state per position instead
    // of the L pairs a gather window needs.)
    template <int R>
    static NDWT_DEV void yzsyn(State& st, Shared& sh, const Taps& tp, const Args& a, const TileCoord& tc, long long obase,
                               int z, bool emit, int buf, int tid) {
        NDWT_SFOR(k, NYI)
            int it = tid + k * NT;
            if (it < YITEMS) {
                int cx = it % TXC, yg = it / TXC;
                const int pc = LD::S(cx);
                v2 P[NP];
                NDWT_SFOR(q, NP)
                    P[q] = (v2)(T(0));
                NDWT_SEND
                NDWT_SFOR(yb, 2)
                    // the RY + L - 1 input chunks, YGRP at a time, scattered into the RY outputs they feed (the register
                    // footprint does not grow with the tap length)
                    NDWT_SFOR(g, (RY + L - 1 + YGRP - 1) / YGRP)
                        chunk cv[YGRP];
                        NDWT_SFOR(t, YGRP)
                            if constexpr (g * YGRP + t < RY + L - 1) cv[t] = sh.xs[buf][yb][yg * RY + g * YGRP + t][pc];
                        NDWT_SEND
                        NDWT_SFOR(t, YGRP)
                            constexpr int r = g * YGRP + t;
                            if constexpr (r < RY + L - 1) {
                                NDWT_SFOR(i, RY)
                                    constexpr int j = r - i;
                                    if constexpr (j >= 0 && j < L) {
                                        NDWT_SFOR(sub, CH)
                                            if constexpr (yb == 0) P[i * CH + sub] += tp.lo[1][j] * LD::get(cv[t], sub);
                                            else P[i * CH + sub] += tp.hi[1][j] * LD::get(cv[t], sub);
                                        NDWT_SEND
                                    }
                                NDWT_SEND
                            }
                        NDWT_SEND
                        if constexpr (L > 12) NDWT_SCHED_FENCE();
                    NDWT_SEND
                NDWT_SEND
                NDWT_SFOR(q, NP)
                    NDWT_SFOR(j, L)
                        constexpr int slot = ((R - 1 - j) % L + L) % L;
                        const T c = tp.lo[2][j] * P[q].x + tp.hi[2][j] * P[q].y;
                        if constexpr (j == 0) st.zacc[k][slot][q] = c;
                        else st.zacc[k][slot][q] += c;
                    NDWT_SEND
                NDWT_SEND
                if (emit) {
                    constexpr int done = ((R - L) % L + L) % L;
                    int gx = tc.x0 + cx * CH;
                    NDWT_SFOR(i, RY)
                        int gy = tc.y0 + yg * RY + i;
                        if (gy < a.n2) {
                            T* dst = a.out[0] + obase + (long long)z * a.plane + (long long)gy * a.rs + gx;
                            if constexpr (VEC4 && CH == 2) {
                                if (gx < a.n1) stream_store(reinterpret_cast<v2*>(dst), v2{st.zacc[k][done][i * CH], st.zacc[k][done][i * CH + CH - 1]}, a.nt);
                            } else if (CH == 2 && gx + 1 < a.n1) {
                                *reinterpret_cast<typename VecT<T>::v2u*>(dst) = v2{st.zacc[k][done][i * CH], st.zacc[k][done][i * CH + CH - 1]};
                            } else {
                                NDWT_SFOR(sub, CH)
                                    if (gx + sub < a.n1) dst[sub] = st.zacc[k][done][i * CH + sub];
                                NDWT_SEND
                            }
                        }
                    NDWT_SEND
                }
            }
        NDWT_SEND
    }
    template <int R>
    static NDWT_DEV void yzdispatch(int r, State& st, Shared& sh, const Taps& tp, const Args& a, const TileCoord& tc,
                                    long long obase, int z, bool emit, int buf, int tid) {
        if constexpr (R < L) {
            if (r == R) yzsyn<R>(st, sh, tp, a, tc, obase, z, emit, buf, tid);
            else yzdispatch<R + 1>(r, st, sh, tp, a, tc, obase, z, emit, buf, tid);
        }
    }

    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared& sh, const Args& a, const Taps& tp, int bid) {
        const TileCoord tc = decode_tile(a, bid, TX, TY, VEC4 ? -1 : 4 * GL, 4 * GR);
        const long long ibase = batch_base(tc.batch, a.bsplit, a.in_bstride, a.in_bstride2);
        const long long obase = batch_base(tc.batch, a.bsplit, a.out_bstride, a.out_bstride2);
        const int zsh = tc.batch * a.zbs;
        const int nsteps = tc.zend - tc.zbeg;
        const int nplanes = nsteps + L - 1;              // planes zbeg-LH .. zend-1+RH
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            setup(st, a, tc, tid);
            load_raw(st, a, ibase, tc.zbeg - LH, zsh);
        });
#if defined(NDWT_STAMPS) && !defined(NDWT_HOST_EMU)
        unsigned long long acc_wait = 0, acc_x = 0, acc_bar = 0, acc_yz = 0, t0;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#define NDWT_STAMP(accu) { unsigned long long t1_; __builtin_amdgcn_sched_barrier(0); \
                           asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_) :: "memory"); \
                           __builtin_amdgcn_sched_barrier(0); accu += t1_ - t0; t0 = t1_; }
#else
#define NDWT_STAMP(accu)
#endif
        for (int p = 0; p < nplanes; ++p) {
            const int s = p - (L - 1);
#if defined(NDWT_STAMPS) && !defined(NDWT_HOST_EMU)
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the wait for this plane's loads, timed on its own
            NDWT_STAMP(acc_wait)
#endif
            ex.each([&](int, State& st) __attribute__((always_inline)) { shrink_raw(st, a); });
            ex.each([&](int tid, State& st) __attribute__((always_inline)) { xsyn(ex, st, sh, tp, p & 1, tid); });
            NDWT_STAMP(acc_x)
            ex.each([&](int, State& st) __attribute__((always_inline)) {                 // (separate pass only matters to the host emulator)
                if (p + 1 < nplanes) load_raw(st, a, ibase, tc.zbeg - LH + p + 1, zsh);
            });
            ex.barrier();
            NDWT_STAMP(acc_bar)
            NDWT_SETPRIO(1);                              // the short y / z stage and its store ahead of every wave's x stage and loads
            ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                yzdispatch<0>((p + 1) % L, st, sh, tp, a, tc, obase, tc.zbeg + s, s >= 0, p & 1, tid);
            });
            NDWT_SETPRIO(0);
            NDWT_STAMP(acc_yz)
        }
#if defined(NDWT_STAMPS) && !defined(NDWT_HOST_EMU)
        if (a.stamps && threadIdx.x % 64 == 0) {
            long long* d = a.stamps + ((long long)bid * (NT / 64) + threadIdx.x / 64) * 4;
            d[0] = (long long)acc_wait; d[1] = (long long)acc_x; d[2] = (long long)acc_bar; d[3] = (long long)acc_yz;
        }
#endif
    }
};

// ---------------------------------------------------------- synthesis, pair-packed lane-shift form ----
// The float default (real data, stride 1, tap lengths <= 8).  Same data path as Inv3S -- x neighbours by DPP wave shifts, the
// x-synthesised tile through LDS with one barrier per plane, z in scatter form -- rebuilt around what the round-2
// measurements showed (tools/timeline_inv.py, tools/micro/valu_rate.hip):
//  * The kernel is bound by the per-CU VECTOR-MEMORY PIPE (about 70 cycles per 1-KiB load instruction), not by arithmetic:
//    lanes and waves that hold no row issue no loads (Inv3S let them re-load a clamped row: 24 of 128 load instructions per
//    plane), and band pointers are wave-uniform SGPR bases + a 32-bit lane offset instead of eight 64-bit VGPR pairs.
//  * Waves stall on their own loads until the pipe accepts them, so WHEN loads are issued matters more than how far ahead:
//    with two register sets (DEPTH 2) half of the waves refill at the start of a plane and half after their x stage, and
//    loads are being issued through the whole plane.  The y / z stages run at raised priority (s_setprio): under oldest-first
//    arbitration the youngest waves' store waited behind the older waves' queued loads and the workgroup waited for them.
//  * All arithmetic is on PAIRS OF ADJACENT x (v_pk_fma_f32: 4.3 cycles per wave-instruction for two FMAs per lane, against
//    4.2 for one v_fmac_f32 with a scalar tap).  x stage: outputs (e, e+1) of one lane share the window element w and take
//    the adjacent taps (t[k], t[k-1]), a pair the host stores in the tap table (Taps3Y::xplo).  An LDS chunk holds, for
//    two adjacent x, (z-bit 0: x0, x1 | z-bit 1: x0, x1), so the y and z stages run on (x0, x1) pairs as well.
//  * Only the low-pass taps live in SGPRs (34 of them): the high-pass taps are the low-pass taps mirrored with alternating
//    signs, applied by the op_sel / neg modifiers of the packed FMA (the host checks the padding parity this relies on).
template <typename T, int L> struct Taps3Y {             // the first two members are Taps3<T, L>: one table serves every kernel
    T lo[3][L];
    T hi[3][L];
    T xplo[L + 1][2];        // (lo[0][k], lo[0][k-1]), k = 0..L, taps outside [0, L) = 0
    T xphi[L + 1][2];        // the same for the high-pass taps (not read by Inv3Y, which derives them)
};

// EW = 2: interleaved complex data.  A pair of adjacent scalars is then the (re, im) of one element and takes ONE tap, so the x
// stage has the form of the y and z stages (a broadcast tap per packed FMA) and the (t[k], t[k-1]) tap pairs are not used.
// ZLDS_ > 0: that many of the L pending z sums of a thread live in LDS instead of registers (each is read, updated and written
// back once per plane by its own thread: no barrier) -- what lets 12 taps keep two register sets of band loads inside the 128
// registers of a 1024-thread workgroup without spills.
// XH_ > 0: every haloed row carries that many more lanes on each side than the synthesis needs (Den3: the lanes whose coefficients
// exist only as inputs of the neighbouring lanes' x analysis).
// UNIYZ_: the y and z axes carry the same taps (the same wavelet, the usual case): the z stage reads the y tap pairs, which frees L SGPRs
// (the kernel holds (L + 1) + L tap pairs, 8 band pointers and its loop scalars in ~100 SGPRs and spills the rest to VGPR lanes).
// XSC_: the x stage in SCATTER form (real data).  The gather form moves every neighbouring sample a lane needs into the lane (v_mov_b32_dpp:
// 8 per band with 8 taps, 12 with 12 -- as many again as a third of its packed FMAs), once per BAND.  In scatter form a lane multiplies only
// its own 4 samples -- into partial sums of the output pairs they reach, its neighbours' included -- and the partial sums travel instead:
// one v_add_f32_dpp per float and hop, once per output STREAM (the x-low and x-high band of a stream add up before they move), and the first
// term of every sum is a v_pk_mul_f32 (no zeroed accumulators): 48 shifted adds instead of 96 moves + 32 zeroings per lane and plane with
// 12 taps, the same 208 packed multiply-adds.  The order of summation differs from the gather form's (results agree to rounding).
template <typename T, int L_, int TX_, int TY_, int NT_, bool VEC4_, int WPE_ = 4, int DEPTH_ = 1, int EW_ = 1, int ZLDS_ = 0, int XH_ = 0,
          bool UNIYZ_ = false, bool XSC_ = false> struct Inv3Y {
    static_assert(sizeof(T) == 4, "pair-packed synthesis: float only (v_pk_fma_f32)");
    static_assert(EW_ == 1 || EW_ == 2 || EW_ == 4, "real data, interleaved complex data / a level dilated by 2, a level dilated by 4");
    static constexpr int L = L_, TX = TX_, TY = TY_, NT = NT_, WPE = WPE_, EW = EW_;
    static constexpr bool VEC4 = VEC4_;
    static constexpr int DEPTH = DEPTH_;                 // register sets of band loads per lane (planes in flight)
    static_assert(DEPTH == 1 || DEPTH == 2, "one or two register sets");
    static constexpr int ZLDS = ZLDS_;                   // pending z sums kept in LDS (slots L - ZLDS .. L - 1)
    static_assert(ZLDS >= 0 && ZLDS < L_, "at least one pending z sum stays in registers");
    static constexpr int NE = VEC4 ? 1 : 4;
    static constexpr int LH = L / 2, RH = L / 2 - 1;
    static constexpr int GL = (LH * EW + 3) / 4, GR = (RH * EW + 3) / 4;
    static constexpr int XH = XH_;
    static constexpr bool UNIYZ = UNIYZ_;
    static constexpr bool XSC = XSC_;
    static_assert(!XSC || XH_ == 0, "scatter x stage: rows without extra lanes");
    // scatter x stage: own sample c (0..3) reaches the output pair P = (2P, 2P+1), counted from the lane's first x, through the tap pair
    // K = c + LH - 2P (0 <= K <= L); pair P belongs to the lane floor(P / 2) away, as its pair P mod 2
    static constexpr int PMIN = -(LH / 2), PMAX = (3 + LH) / 2, NPQ = PMAX - PMIN + 1;
    static constexpr int fdiv2(int p) { return p >= 0 ? p / 2 : -((-p + 1) / 2); }
    static constexpr int DFAR = fdiv2(PMAX) > -fdiv2(PMIN) ? fdiv2(PMAX) : -fdiv2(PMIN);   // farthest lane a partial sum travels
    // EW = 2 (interleaved complex data / a level dilated by 2): the lane's two ELEMENTS c = 0, 1 (a (re, im) pair each) reach the output
    // elements n = c + LH - j, j = 0 .. L-1, i.e. -RH .. LH + 1; element n belongs to the lane floor(n / 2) away, as its element n mod 2
    static constexpr int SMIN = EW_ == 2 ? -RH : PMIN, SMAX = EW_ == 2 ? LH + 1 : PMAX, NSQ = SMAX - SMIN + 1;
    static constexpr int SFAR = fdiv2(SMAX) > -fdiv2(SMIN) ? fdiv2(SMAX) : -fdiv2(SMIN);
    static_assert(!XSC || EW_ == 4 || (fdiv2(SMAX) <= GL && -fdiv2(SMIN) <= GR), "halo lanes cover the reach of the partial sums");
    static constexpr int NG = TX / 4 + GL + GR + 2 * XH; // lanes per haloed row
    static constexpr int NR = TY + L - 1;                // haloed rows: loaded, x-synthesised, kept in LDS
    static constexpr int RPW = 64 / NG;                  // rows per wave
    static constexpr int NW = NT / 64;
    static constexpr int RPR = RPW * NW;                 // rows per round
    static constexpr int NRND = (NR + RPR - 1) / RPR;
    static constexpr int TXC = TX / 2;                   // chunks per row: one 16-byte chunk = two adjacent x (EW = 2: one element)
    static constexpr int YITEMS = TXC * TY;              // y/z item: one chunk column of one output row
    static constexpr int NYI = (YITEMS + NT - 1) / NT;
    static constexpr int XV = 4 * (1 + GL + GR);
    static constexpr int KB = LH - 4 * GL;               // window element i feeds output e through tap j = i + KB - e
    static constexpr int YG = DEPTH == 2 ? 4 : 8;        // LDS chunks the y stage holds at a time
    static constexpr unsigned kNoRow = 0xFFFFFFFFu;      // State::off of a lane that holds no row
    static_assert(TX % 4 == 0 && L % 2 == 0 && NT % 64 == 0 && RPW >= 1, "tile shape");
    typedef typename VecT<T>::v2 v2;
    typedef typename VecT<T>::v4 v4;
    typedef Lds<T> LD;
    typedef v4 chunk;
    typedef Taps3Y<T, L> Taps;
    typedef Fused3Args<T> Args;

    struct Shared {
        chunk xs[2][2][NR][TXC];   // [buffer][y-bit][row][x pair] of (z-bit 0: x0, x1 | z-bit 1: x0, x1)
        v2 zl[ZLDS ? ZLDS : 1][ZLDS ? NYI * NT : 1];   // [slot][item]: the pending z sums that do not live in registers
    };
    struct State {
        v2 zacc[NYI][L - ZLDS];    // z-synthesis in scatter form: partial sums of the next L output planes (x0, x1), rotating
        v4 raw[DEPTH][NRND][8];    // 4 x of every band of this lane's row(s); DEPTH 2: the set index is the plane's parity
        unsigned off[NRND][NE];    // BYTE offsets inside a plane (kNoRow: this lane holds no row and loads nothing)
        v2 P[NYI][2];              // y-synthesised (x0, x1) pairs of the newest plane: z-low / z-high inputs of the z stage
        v2 xq[XSC ? (EW_ == 4 ? 8 : NSQ) : 1];   // scatter x stage: partial sums of the output pairs PMIN .. PMAX of one stream (short-lived);
                                   // tap stride 4: the sums on their way to the right / left, [generation][direction][pair of the lane]
        v2 xo[XSC ? 2 : 1][2];     // scatter x stage: the lane's two output pairs of the z-low / z-high stream of a y-bit
        unsigned ooff[NYI];        // byte offset of this thread's output pair inside a plane
        int ostore[NYI];           // outputs this thread stores: 0 none (outside the volume), 1 the first x only, 2 the pair
    };

    // the taps the kernel uses, read once per workgroup into SGPR pairs (low-pass only: the high-pass ones are derived by the
    // operand modifiers of the packed FMAs).  Passed through an empty asm so that the compiler treats them as values to keep
    // in registers: as loads from constant memory it re-issues them (s_load + lgkmcnt(0)) inside every band of the x stage.
    struct RegTaps {
        v2 xp[EW == 1 ? L + 1 : 1];   // (lo_x[k], lo_x[k-1]); real data only
        v2 xl[EW != 1 ? L / 2 : 1];   // (lo_x[2m], lo_x[2m+1]); EW = 2, 4 (one tap per pair of scalars)
        v2 yl[L / 2], zl[L / 2];      // (lo[2m], lo[2m+1]) of the y and z axes
    };
    static NDWT_DEV v2 pinned(v2 t) {
#ifndef NDWT_HOST_EMU
        asm volatile("" : "+s"(t));
#endif
        return t;
    }
    static NDWT_DEV v2 pinned_v(v2 v) {
#ifndef NDWT_HOST_EMU
        asm volatile("" : "+v"(v));
#endif
        return v;
    }
    static NDWT_DEV void load_taps(RegTaps& rt, const Taps& tp) {
        if constexpr (EW == 1) {
            NDWT_SFOR(k, L + 1)
                rt.xp[k] = pinned(v2{tp.xplo[k][0], tp.xplo[k][1]});
            NDWT_SEND
        } else {
            NDWT_SFOR(m, L / 2)
                rt.xl[m] = pinned(v2{tp.lo[0][2 * m], tp.lo[0][2 * m + 1]});
            NDWT_SEND
        }
        NDWT_SFOR(m, L / 2)
            rt.yl[m] = pinned(v2{tp.lo[1][2 * m], tp.lo[1][2 * m + 1]});
            if constexpr (!UNIYZ) rt.zl[m] = pinned(v2{tp.lo[2][2 * m], tp.lo[2][2 * m + 1]});
        NDWT_SEND
    }
    static NDWT_DEV const v2 (&ztaps(const RegTaps& rt))[L / 2] {
        if constexpr (UNIYZ) return rt.yl;
        else return rt.zl;
    }

    // Global memory through a WAVE-UNIFORM base (kernel arguments and tile coordinates only) plus a 32-bit per-lane byte
    // offset: the scalar-base addressing form.  readfirstlane pins the base to SGPRs -- without it hipcc hoists
    // `band pointer + lane offset` out of the plane loop as eight 64-bit VGPR pairs -- and the address_space(1) casts keep the
    // accesses global_* (a pointer rebuilt from integers would otherwise be a flat one).
#ifndef NDWT_HOST_EMU
    typedef const __attribute__((address_space(1))) char* gcptr;
    typedef __attribute__((address_space(1))) char* gptr;
    static NDWT_DEV unsigned long long uniform_bits(const void* p) {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return ((unsigned long long)hi << 32) | lo;
    }
    template <class V> static NDWT_DEV V gload(const void* base, unsigned off) {
        return *reinterpret_cast<const __attribute__((address_space(1))) V*>((gcptr)uniform_bits(base) + off);
    }
    // the output plane is written once and never read back by this kernel: a nontemporal store keeps it from displacing the
    // band lines that neighbouring tiles are about to share in L2 (tools/micro/stream_pattern: 8 -> 1 copy 0.92 -> 0.86 ms)
    template <class V> static NDWT_DEV void gstore(void* base, unsigned off, V v, int nt) {
#ifndef NDWT_NO_NT_STORE
        if (nt) { __builtin_nontemporal_store(v, reinterpret_cast<__attribute__((address_space(1))) V*>((gptr)uniform_bits(base) + off)); return; }
#endif
        *reinterpret_cast<__attribute__((address_space(1))) V*>((gptr)uniform_bits(base) + off) = v;
    }
#else
    // (memcpy: a template argument drops the element alignment of VecT::v4u / v2u, and the host would use aligned vector moves)
    template <class V> static NDWT_DEV V gload(const void* base, unsigned off) { V v; __builtin_memcpy(&v, (const char*)base + off, sizeof(V)); return v; }
    template <class V> static NDWT_DEV void gstore(void* base, unsigned off, V v, int) { __builtin_memcpy((char*)base + off, &v, sizeof(V)); }
#endif

    // ---- packed FMAs with explicit operand selection (v_pk_fma_f32 op_sel / neg modifiers) ----
    // Why by hand: (1) hipcc keeps a scalar multiplier of a packed FMA as a DUPLICATED SGPR pair (t, t); (2) the high-pass
    // taps are the low-pass taps mirrored with alternating signs (s_hi[j] = (-1)^(j+1) s_lo[L-1-j], ndwt_filters.h; the
    // symmetric zero padding to L keeps this when the padding on each side is even), which the hardware applies for free
    // through op_sel (swap the halves of a tap pair) and neg_lo / neg_hi.  Together: 34 SGPRs of taps instead of 132, so the
    // tap table stays in SGPRs across the plane loop instead of being reloaded (s_load + lgkmcnt(0)) or spilled to VGPR
    // lanes (v_readlane) inside it.  The statements are not volatile: the compiler still schedules them.
    // acc += (a[SELA], a[SELA]) * (t0, t1) with (t0, t1) = tp, or (tp[1], tp[0]) if SWAP; NEGLO / NEGHI negate t0 / t1.
    template <int SELA, bool SWAP, bool NEGLO, bool NEGHI> static NDWT_DEV void pk_fma_bt(v2& acc, const v2 a, const v2 tp) {
        PkF32::fma_bt<SELA, SWAP, NEGLO, NEGHI>(acc, a, tp);
    }
    // acc += x * (t, t), t = tp[HI], negated if NEG
    template <int HI, bool NEG> static NDWT_DEV void pk_fma_s(v2& acc, const v2 x, const v2 tp) { PkF32::fma_s<HI, NEG>(acc, x, tp); }
    // acc += x * low-pass tap J (HIGH = false) or high-pass tap J = (-1)^(J+1) low-pass tap L-1-J, from the low-pass tap pairs `lo`
    template <int J, bool HIGH> static NDWT_DEV void tap_fma(v2& acc, const v2 x, const v2 (&lo)[L / 2]) {
        constexpr int jj = HIGH ? L - 1 - J : J;
        pk_fma_s<jj & 1, HIGH && (J % 2 == 0)>(acc, x, lo[jj / 2]);
    }
    // acc(e, e+1) += (t[K], t[K-1]) * w[H]: the x taps of two adjacent outputs for one window entry.  Low-pass: the pair
    // xp[K].  High-pass: (hi[K], hi[K-1]) = ((-1)^(K+1) lo[L-1-K], (-1)^K lo[L-K]) = the halves of xp[L-K] swapped, one negated.
    template <int H, int K, bool HIGH> static NDWT_DEV void xtap(v2& acc, const v2 w, const RegTaps& tp) {
        if constexpr (!HIGH) pk_fma_bt<H, false, false, false>(acc, w, tp.xp[K]);
        else pk_fma_bt<H, true, K % 2 == 0, K % 2 != 0>(acc, w, tp.xp[L - K]);
    }

    static NDWT_DEV void lane_item(int tid, int rnd, int& ug, int& r, bool& valid) {
        const int lane = tid % 64, wv = tid / 64;
        const int rs = lane / NG;
        ug = lane % NG;
        r = rnd * RPR + wv * RPW + rs;
        valid = rs < RPW && r < NR;
        if (r >= NR) r = NR - 1;
    }
    // DEPTH 2: the waves that refill a register set at the START of a plane rather than after their x stage.  Waves w, w+4,
    // w+8, w+12 share a SIMD: every SIMD gets two of each kind.
    static NDWT_DEV bool early_refill(int tid) { return (((tid >> 6) >> 2) & 1) == 1; }

    static NDWT_DEV void setup(State& st, const Args& a, const TileCoord& tc, int tid) {
        NDWT_SFOR(k, NRND)
            int ug, r;
            bool valid;
            lane_item(tid, k, ug, r, valid);
            const int y = modn(tc.y0 - LH + r, a.n2);
            const int xb = tc.x0 - 4 * (GL + XH) + 4 * ug;
            NDWT_SFOR(e, NE)
                st.off[k][e] = valid ? (unsigned)(y * a.rs + modn(xb + e, a.n1)) * (unsigned)sizeof(T) : kNoRow;
            NDWT_SEND
        NDWT_SEND
        NDWT_SFOR(d, DEPTH)                               // lanes without a row keep zeros (their values reach no output)
            NDWT_SFOR(k, NRND)
                NDWT_SFOR(b, 8)
                    st.raw[d][k][b] = (v4)(T(0));
                NDWT_SEND
            NDWT_SEND
        NDWT_SEND
        NDWT_SFOR(k, NYI)                                 // where this thread's output pair goes
            const int it = tid + k * NT;
            const int cx = it % TXC, q = it / TXC;
            const int gx = tc.x0 + 2 * cx, gy = tc.y0 + q;
            st.ostore[k] = (it < YITEMS && gy < a.n2 && gx < a.n1) ? (gx + 1 < a.n1 ? 2 : 1) : 0;
            st.ooff[k] = st.ostore[k] ? (unsigned)(gy * a.rs + gx) * (unsigned)sizeof(T) : 0u;
        NDWT_SEND
    }

    // issue the loads of plane zraw into register set SET (lanes, and whole waves, without a row issue none)
    template <int SET> static NDWT_DEV void load_raw(State& st, const Args& a, long long ibase, int zraw, int zsh) {
        long long zm = a.z_wrap == 1 ? (long long)modn(zraw, a.n3) : (long long)(zraw + LH);
        if (a.z_wrap == 3) {                             // zero-extended slab
            zm = zraw - RH;
            if (zm + zsh < a.zlo || zm + zsh >= a.zhi) {
                NDWT_SFOR(k, NRND)
                    NDWT_SFOR(b, 8)
                        st.raw[SET][k][b] = (v4)(T(0));
                    NDWT_SEND
                NDWT_SEND
                return;
            }
        }
        const long long pb = ibase + zm * a.plane;
        NDWT_SFOR(k, NRND)
            if constexpr (VEC4) {
                if (st.off[k][0] != kNoRow) {
                    NDWT_SFOR(b, 8)
                        st.raw[SET][k][b] = gload<v4>(a.in[b] + pb, st.off[k][0]);          // (a.in[b] + pb: wave-uniform)
                    NDWT_SEND
                }
            } else {
                // rows that are not whole groups of 4: every lane whose 4 x are contiguous in memory still takes them in one access
                // (VecT::v4u); the one lane per row that straddles the periodic wrap (and rows shorter than 4) go element by element
                if (st.off[k][0] != kNoRow && st.off[k][NE - 1] == st.off[k][0] + 3 * (unsigned)sizeof(T)) {
                    NDWT_SFOR(b, 8)
                        st.raw[SET][k][b] = gload<typename VecT<T>::v4u>(a.in[b] + pb, st.off[k][0]);
                    NDWT_SEND
                } else if (st.off[k][0] != kNoRow) {
                    NDWT_SFOR(b, 8)
                        NDWT_SFOR(e, NE)
                            st.raw[SET][k][b][e] = gload<T>(a.in[b] + pb, st.off[k][e]);
                        NDWT_SEND
                    NDWT_SEND
                }
            }
        NDWT_SEND
    }

    template <int SET> static NDWT_DEV void shrink_raw(State& st, const Args& a) {
        if (a.shrink_mask == 0) return;
        NDWT_SFOR(k, NRND)
            NDWT_SFOR(b, 8)
                if ((a.shrink_mask >> b) & 1) shrink4<T, (EW == 2 ? 2 : 1)>(st.raw[SET][k][b], a.shrink_thr, a.shrink_hard);
            NDWT_SEND
        NDWT_SEND
    }

    // x-synthesis of this lane's 4 x, as the two pairs (0,1) and (2,3): pair01 = sum_i (t[k], t[k-1]) w_i with k = i + KB,
    // pair23 the same with k = i + KB - 2
    template <int SET, class Exec>
    static NDWT_DEV void xsyn(Exec& ex, State& st, Shared& sh, const RegTaps& tp, int buf, int tid) {
        NDWT_SFOR(k, NRND)
            int ug, r;
            bool valid;
            lane_item(tid, k, ug, r, valid);
            NDWT_SFOR(yb, 2)
                v2 acc[2][2];                            // [z-bit][pair 01 / 23]
                NDWT_SFOR(zb, 2)
                    acc[zb][0] = (v2)(T(0));
                    acc[zb][1] = (v2)(T(0));
                    // one input band at a time (x-low then x-high of this z-bit).  The window is held as PAIRS of adjacent
                    // entries -- the halves of the loaded v4 of this lane, and of the neighbours' shifted into this lane --
                    // so that every packed-FMA operand is an aligned register pair with nothing wasted.
                    NDWT_SFOR(xb, 2)
                        NDWT_SFOR(m, XV / 2)
                            constexpr int D = (2 * m) / 4 - GL;      // lane distance of window entries 2m, 2m+1
                            constexpr int c = (2 * m) % 4;
                            constexpr int ka = 2 * m + KB, kb = 2 * m + 1 + KB;        // tap-pair index of entry 2m / 2m+1 for outputs (0,1)
                            constexpr bool ua01 = ka >= 0 && ka <= L, ub01 = kb >= 0 && kb <= L;
                            constexpr bool ua23 = ka - 2 >= 0 && ka - 2 <= L, ub23 = kb - 2 >= 0 && kb - 2 <= L;
                            constexpr int jc0 = m - 2 * GL + LH, jc1 = jc0 - 1;             // EW = 2: taps of the lane's two elements
                            constexpr bool c0ok = jc0 >= 0 && jc0 < L, c1ok = jc1 >= 0 && jc1 < L;
                            constexpr int j4 = D + LH;                                      // EW = 4: the lane D away is x element D away
                            constexpr bool ok4 = j4 >= 0 && j4 < L;
                            if constexpr (EW == 1 ? (ua01 || ub01 || ua23 || ub23) : EW == 2 ? (c0ok || c1ok) : ok4) {
                                const v2 w = {NDWT_LANE_SHIFT(ex, tid, D, s.raw[SET][k][xb + 2 * yb + 4 * zb][c]),
                                              NDWT_LANE_SHIFT(ex, tid, D, s.raw[SET][k][xb + 2 * yb + 4 * zb][c + 1])};
                                if constexpr (EW == 1) {
                                    if constexpr (ua01) xtap<0, ka, xb == 1>(acc[zb][0], w, tp);
                                    if constexpr (ub01) xtap<1, kb, xb == 1>(acc[zb][0], w, tp);
                                    if constexpr (ua23) xtap<0, ka - 2, xb == 1>(acc[zb][1], w, tp);
                                    if constexpr (ub23) xtap<1, kb - 2, xb == 1>(acc[zb][1], w, tp);
                                } else if constexpr (EW == 2) {
                                    // w = window element m - 2 GL (re, im); it feeds the lane's element e through tap m - 2 GL - e + LH
                                    if constexpr (c0ok) tap_fma<jc0 < 0 ? 0 : (jc0 >= L ? 0 : jc0), xb == 1>(acc[zb][0], w, tp.xl);
                                    if constexpr (c1ok) tap_fma<jc1 < 0 ? 0 : (jc1 >= L ? 0 : jc1), xb == 1>(acc[zb][1], w, tp.xl);
                                } else {
                                    // a level dilated by 4: the lane's 4 scalars are the same x of 4 sub-lattices, the lane D away holds the
                                    // element D steps away on each of them -- whole-lane shifts, one tap for both pairs
                                    tap_fma<j4 < 0 ? 0 : (j4 >= L ? 0 : j4), xb == 1>(acc[zb][c / 2], w, tp.xl);
                                }
                            }
                        NDWT_SEND
                        NDWT_SCHED_FENCE();               // hipcc otherwise hoists every DPP move of a y-bit ahead of the FMAs
                    NDWT_SEND
                NDWT_SEND
                if (valid && ug >= GL + XH && ug < GL + XH + TX / 4) {
                    const chunk c0 = {acc[0][0].x, acc[0][0].y, acc[1][0].x, acc[1][0].y};
                    const chunk c1 = {acc[0][1].x, acc[0][1].y, acc[1][1].x, acc[1][1].y};
                    chunk* row = sh.xs[buf][yb][r];
                    row[LD::S(2 * (ug - GL - XH))] = c0;
                    row[LD::S(2 * (ug - GL - XH) + 1)] = c1;
                }
            NDWT_SEND
        NDWT_SEND
    }

    // first = true: acc = ..., else acc += ... (the x tap pair K of sample half H, low- or high-pass)
    template <int H, int K, bool HIGH, bool FIRST> static NDWT_DEV void xtap_first(v2& acc, const v2 w, const RegTaps& tp) {
        if constexpr (!FIRST) xtap<H, K, HIGH>(acc, w, tp);
        else if constexpr (!HIGH) PkF32::mul_bt<H, false, false, false>(acc, w, tp.xp[K]);
        else PkF32::mul_bt<H, true, K % 2 == 0, K % 2 != 0>(acc, w, tp.xp[L - K]);
    }
    static constexpr bool xsc_has(int P) { return P >= SMIN && P <= SMAX; }
    // acc = / += w * (synthesis tap J, low- or high-pass) from the pairs xl[m] = (lo[2m], lo[2m+1]); FIRST: the first term of a sum
    template <int J, bool HIGH, bool FIRST> static NDWT_DEV void tap_first(v2& acc, const v2 w, const v2 (&lo)[L / 2]) {
        if constexpr (!FIRST) tap_fma<J, HIGH>(acc, w, lo);
        else {
            constexpr int jj = HIGH ? L - 1 - J : J;
            PkF32::mul_s<jj & 1, HIGH && (J % 2 == 0)>(acc, w, lo[jj / 2]);
        }
    }
    // x stage in scatter form (XSC; EW = 1 real data, EW = 2 interleaved complex data / a level dilated by 2).  Runs at workgroup level: the
    // partial sums of the neighbouring lanes are read between its passes (one straight line of code on the GPU; the host emulation finishes a
    // pass for every lane before the next one reads them).
    template <int SET, class Exec>
    static NDWT_DEV void xsyn_scatter(Exec& ex, Shared& sh, const RegTaps& tp, int buf) {
        NDWT_SFOR(k, NRND)
            NDWT_SFOR(yb, 2)
                NDWT_SFOR(zb, 2)
                    // pass 1: the lane's 4 scalars of the x-low and the x-high band times every tap (pair) they reach
                    ex.each([&](int, State& st) __attribute__((always_inline)) {
                        NDWT_SFOR(pi, NSQ)
                            constexpr int P = SMIN + pi;
                            NDWT_SFOR(xb, 2)
                                const v4 r = st.raw[SET][k][xb + 2 * yb + 4 * zb];
                                const v2 w01 = {r[0], r[1]}, w23 = {r[2], r[3]};
                                if constexpr (EW == 1) {
                                    NDWT_SFOR(c, 4)
                                        constexpr int K = c + LH - 2 * P;
                                        if constexpr (K >= 0 && K <= L) {
                                            // the first valid (xb, c) of this pair starts the sum
                                            constexpr int c_first = (2 * P - LH) > 0 ? (2 * P - LH) : 0;
                                            constexpr bool first = xb == 0 && c == c_first;
                                            xtap_first<c % 2, K, xb == 1, first>(st.xq[pi], c < 2 ? w01 : w23, tp);
                                        }
                                    NDWT_SEND
                                } else {
                                    NDWT_SFOR(c, 2)                  // element c = (re, im) -> output element P through tap c - P + LH
                                        constexpr int J = c - P + LH;
                                        if constexpr (J >= 0 && J < L) {
                                            constexpr int c_first = (P - LH) > 0 ? (P - LH) : 0;
                                            constexpr bool first = xb == 0 && c == c_first;
                                            tap_first<J, xb == 1, first>(st.xq[pi], c == 0 ? w01 : w23, tp.xl);
                                        }
                                    NDWT_SEND
                                }
                            NDWT_SEND
                        NDWT_SEND
                    });
                    // passes 2 ..: sums that travel more than one lane are added to the next-nearer lane's sum for the same destination
                    NDWT_SFOR(hh, (SFAR > 1 ? SFAR - 1 : 0))
                        constexpr int h = SFAR - hh;     // SFAR .. 2
                        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                            (void)tid;
                            NDWT_SFOR(q, 2)
                                if constexpr (xsc_has(2 * h + q)) {          // to the right: lane i takes lane i-1's
                                    constexpr int far = 2 * h + q - SMIN, near = 2 * (h - 1) + q - SMIN;
                                    st.xq[near].x += NDWT_LANE_SHIFT(ex, tid, -1, s.xq[far].x);
                                    st.xq[near].y += NDWT_LANE_SHIFT(ex, tid, -1, s.xq[far].y);
                                }
                                if constexpr (xsc_has(-2 * h + q)) {         // to the left: lane i takes lane i+1's
                                    constexpr int far = -2 * h + q - SMIN, near = -2 * (h - 1) + q - SMIN;
                                    st.xq[near].x += NDWT_LANE_SHIFT(ex, tid, 1, s.xq[far].x);
                                    st.xq[near].y += NDWT_LANE_SHIFT(ex, tid, 1, s.xq[far].y);
                                }
                            NDWT_SEND
                        });
                    NDWT_SEND
                    // last pass: own sums + the left neighbour's sums for this lane + the right neighbour's
                    ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                        (void)tid;
                        NDWT_SFOR(q, 2)
                            v2 o = st.xq[q - SMIN];
                            if constexpr (xsc_has(2 + q)) {
                                o.x += NDWT_LANE_SHIFT(ex, tid, -1, s.xq[2 + q - SMIN].x);
                                o.y += NDWT_LANE_SHIFT(ex, tid, -1, s.xq[2 + q - SMIN].y);
                            }
                            if constexpr (xsc_has(-2 + q)) {
                                o.x += NDWT_LANE_SHIFT(ex, tid, 1, s.xq[-2 + q - SMIN].x);
                                o.y += NDWT_LANE_SHIFT(ex, tid, 1, s.xq[-2 + q - SMIN].y);
                            }
                            // (the sums are only stored under the lane's `valid` test below: without this the compiler sinks the two adds into
                            // that branch, away from their wave shifts, and the shifts stay v_mov_b32_dpp + v_add_f32 instead of v_add_f32_dpp)
                            st.xo[zb][q] = pinned_v(o);
                        NDWT_SEND
                    });
                NDWT_SEND
                ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                    int ug, r;
                    bool valid;
                    lane_item(tid, k, ug, r, valid);
                    if (valid && ug >= GL + XH && ug < GL + XH + TX / 4) {
                        const chunk c0 = {st.xo[0][0].x, st.xo[0][0].y, st.xo[1][0].x, st.xo[1][0].y};
                        const chunk c1 = {st.xo[0][1].x, st.xo[0][1].y, st.xo[1][1].x, st.xo[1][1].y};
                        chunk* row = sh.xs[buf][yb][r];
                        row[LD::S(2 * (ug - GL - XH))] = c0;
                        row[LD::S(2 * (ug - GL - XH) + 1)] = c1;
                    }
                });
            NDWT_SEND
        NDWT_SEND
    }
    // Scatter form of the x stage at tap stride 4 (EW = 4: a lane's 4 scalars are one x of 4 sub-lattices, the lane D away holds the
    // element D steps away on each).  The gather form shifts every band's 4 scalars past L - 1 lanes (28 v_mov_b32_dpp per band with 8
    // taps: 224 per lane and plane against 128 packed FMAs).  Here a lane multiplies its own scalars by every tap and the SUMS walk: the
    // sum on its way to the right takes, at every lane it passes, that lane's term for the same destination (S_k = own * t[LH - k] +
    // S_{k+1} of the lane to the left), likewise to the left -- one v_add_f32_dpp per scalar and hop, once per output stream instead
    // of once per band: 112 instead of 224, no zeroed accumulators.
    template <int SET, class Exec>
    static NDWT_DEV void xsyn_scatter4(Exec& ex, Shared& sh, const RegTaps& tp, int buf) {
        NDWT_SFOR(k, NRND)
            NDWT_SFOR(yb, 2)
                NDWT_SFOR(zb, 2)
                    // generation g of the walking sums lives in xq[4 (g & 1) + 2 dir + q]: a pass reads its neighbour's previous generation
                    NDWT_SFOR(g, (LH > RH ? LH : RH))
                        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                            (void)tid;
                            const v4 rl = st.raw[SET][k][0 + 2 * yb + 4 * zb], rh = st.raw[SET][k][1 + 2 * yb + 4 * zb];
                            NDWT_SFOR(q, 2)
                                const v2 wl = {rl[2 * q], rl[2 * q + 1]}, wh = {rh[2 * q], rh[2 * q + 1]};
                                if constexpr (g < LH) {                       // to the right: destination LH - g lanes away, tap g
                                    v2 a;
                                    tap_first<g, false, true>(a, wl, tp.xl);
                                    tap_first<g, true, false>(a, wh, tp.xl);
                                    if constexpr (g > 0) {
                                        a.x += NDWT_LANE_SHIFT(ex, tid, -1, s.xq[4 * ((g - 1) & 1) + q].x);
                                        a.y += NDWT_LANE_SHIFT(ex, tid, -1, s.xq[4 * ((g - 1) & 1) + q].y);
                                    }
                                    st.xq[4 * (g & 1) + q] = a;
                                }
                                if constexpr (g < RH) {                       // to the left: destination RH - g lanes away, tap L - 1 - g
                                    v2 a;
                                    tap_first<L - 1 - g, false, true>(a, wl, tp.xl);
                                    tap_first<L - 1 - g, true, false>(a, wh, tp.xl);
                                    if constexpr (g > 0) {
                                        a.x += NDWT_LANE_SHIFT(ex, tid, 1, s.xq[4 * ((g - 1) & 1) + 2 + q].x);
                                        a.y += NDWT_LANE_SHIFT(ex, tid, 1, s.xq[4 * ((g - 1) & 1) + 2 + q].y);
                                    }
                                    st.xq[4 * (g & 1) + 2 + q] = a;
                                }
                            NDWT_SEND
                        });
                    NDWT_SEND
                    // the lane's own term (tap LH) + what has arrived from the left (generation LH - 1) and from the right (generation RH - 1)
                    ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                        (void)tid;
                        const v4 rl = st.raw[SET][k][0 + 2 * yb + 4 * zb], rh = st.raw[SET][k][1 + 2 * yb + 4 * zb];
                        NDWT_SFOR(q, 2)
                            const v2 wl = {rl[2 * q], rl[2 * q + 1]}, wh = {rh[2 * q], rh[2 * q + 1]};
                            v2 o;
                            tap_first<LH, false, true>(o, wl, tp.xl);
                            tap_first<LH, true, false>(o, wh, tp.xl);
                            if constexpr (LH > 0) {
                                o.x += NDWT_LANE_SHIFT(ex, tid, -1, s.xq[4 * ((LH - 1) & 1) + q].x);
                                o.y += NDWT_LANE_SHIFT(ex, tid, -1, s.xq[4 * ((LH - 1) & 1) + q].y);
                            }
                            if constexpr (RH > 0) {
                                o.x += NDWT_LANE_SHIFT(ex, tid, 1, s.xq[4 * ((RH - 1) & 1) + 2 + q].x);
                                o.y += NDWT_LANE_SHIFT(ex, tid, 1, s.xq[4 * ((RH - 1) & 1) + 2 + q].y);
                            }
                            st.xo[zb][q] = pinned_v(o);
                        NDWT_SEND
                    });
                NDWT_SEND
                ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                    int ug, r;
                    bool valid;
                    lane_item(tid, k, ug, r, valid);
                    if (valid && ug >= GL + XH && ug < GL + XH + TX / 4) {
                        const chunk c0 = {st.xo[0][0].x, st.xo[0][0].y, st.xo[1][0].x, st.xo[1][0].y};
                        const chunk c1 = {st.xo[0][1].x, st.xo[0][1].y, st.xo[1][1].x, st.xo[1][1].y};
                        chunk* row = sh.xs[buf][yb][r];
                        row[LD::S(2 * (ug - GL - XH))] = c0;
                        row[LD::S(2 * (ug - GL - XH) + 1)] = c1;
                    }
                });
            NDWT_SEND
        NDWT_SEND
    }
    // the x stage of register set SET into xs[buf], in the form this instance uses
    template <int SET, class Exec>
    static NDWT_DEV void xstage(Exec& ex, Shared& sh, const RegTaps& tp, int buf) {
        if constexpr (XSC && EW == 4) xsyn_scatter4<SET>(ex, sh, tp, buf);
        else if constexpr (XSC) xsyn_scatter<SET>(ex, sh, tp, buf);
        else ex.each([&](int tid, State& st) __attribute__((always_inline)) { xsyn<SET>(ex, st, sh, tp, buf, tid); });
    }

    // y-synthesis of output row q from LDS rows q .. q+L-1: the (x0, x1) pairs of the z-low / z-high inputs of the z-synthesis.
    // Kept apart from the z stage so that it exists once, not once per rotation of the z window.
    static NDWT_DEV void ysyn(State& st, Shared& sh, const RegTaps& tp, int buf, int tid) {
        NDWT_SFOR(k, NYI)
            const int it = tid + k * NT;
            v2 P0 = (v2)(T(0)), P1 = (v2)(T(0));
            if (it < YITEMS) {
                const int cx = it % TXC, q = it / TXC;
                const int pc = LD::S(cx);
                NDWT_SFOR(yb, 2)
                    NDWT_SFOR(h, (L + YG - 1) / YG)            // YG chunks at a time: the register footprint of the LDS reads
                        chunk cv[YG];
                        NDWT_SFOR(t, YG)
                            if constexpr (h * YG + t < L) cv[t] = sh.xs[buf][yb][q + h * YG + t][pc];
                        NDWT_SEND
                        NDWT_SFOR(t, YG)
                            if constexpr (h * YG + t < L) {
                                tap_fma<h * YG + t, yb == 1>(P0, v2{cv[t][0], cv[t][1]}, tp.yl);
                                tap_fma<h * YG + t, yb == 1>(P1, v2{cv[t][2], cv[t][3]}, tp.yl);
                            }
                        NDWT_SEND
                        NDWT_SCHED_FENCE();               // the next group's chunks are read after this one's are consumed
                    NDWT_SEND
                NDWT_SEND
            }
            st.P[k][0] = P0;
            st.P[k][1] = P1;
        NDWT_SEND
    }

    // z-synthesis in scatter form (rotation R as in Inv3S) and the store of the plane it completes
    template <int R>
    static NDWT_DEV void zsyn(State& st, Shared& sh, const RegTaps& tp, const Args& a, long long obase, int z, bool emit, int tid) {
        NDWT_SFOR(k, NYI)
            const int it = tid + k * NT;
            if (it < YITEMS) {
                const v2 P0 = st.P[k][0], P1 = st.P[k][1];
                v2 o = (v2)(T(0));                       // the sum that tap L-1 completes: output plane z
                NDWT_SFOR(j, L)
                    constexpr int slot = ((R - 1 - j) % L + L) % L;
                    if constexpr (slot < L - ZLDS) {
                        if constexpr (j == 0) st.zacc[k][slot] = (v2)(T(0));
                        tap_fma<j, false>(st.zacc[k][slot], P0, ztaps(tp));
                        tap_fma<j, true>(st.zacc[k][slot], P1, ztaps(tp));
                        if constexpr (j == L - 1) o = st.zacc[k][slot];
                    } else {
                        v2 acc = (v2)(T(0));
                        if constexpr (j != 0) acc = sh.zl[slot - (L - ZLDS)][it];
                        tap_fma<j, false>(acc, P0, ztaps(tp));
                        tap_fma<j, true>(acc, P1, ztaps(tp));
                        if constexpr (j == L - 1) o = acc;
                        else sh.zl[slot - (L - ZLDS)][it] = acc;
                    }
                NDWT_SEND
                if (emit && st.ostore[k]) {
                    T* dst = a.out[0] + obase + (long long)z * a.plane;   // wave-uniform
                    if constexpr (VEC4) {
                        gstore<v2>(dst, st.ooff[k], o, a.nt);
                    } else if (st.ostore[k] == 2) {
                        gstore<typename VecT<T>::v2u>(dst, st.ooff[k], o, 0);
                    } else {
                        gstore<T>(dst, st.ooff[k], o.x, 0);
                    }
                }
            }
        NDWT_SEND
    }
    // the store of the pending sum in slot R as it is (the z stage of an iteration whose inputs are all zero)
    template <int R> static NDWT_DEV void zemit(State& st, Shared& sh, const Args& a, long long obase, int z, int tid) {
        NDWT_SFOR(k, NYI)
            const int it = tid + k * NT;
            if (it < YITEMS && st.ostore[k]) {
                v2 o;
                if constexpr (R < L - ZLDS) o = st.zacc[k][R];
                else o = sh.zl[R - (L - ZLDS)][it];
                T* dst = a.out[0] + obase + (long long)z * a.plane;   // wave-uniform
                if constexpr (VEC4) {
                    gstore<v2>(dst, st.ooff[k], o, a.nt);
                } else if (st.ostore[k] == 2) {
                    gstore<typename VecT<T>::v2u>(dst, st.ooff[k], o, 0);
                } else {
                    gstore<T>(dst, st.ooff[k], o.x, 0);
                }
            }
        NDWT_SEND
    }
    template <int R> static NDWT_DEV void zemit_dispatch(int r, State& st, Shared& sh, const Args& a, long long obase, int z, int tid) {
        if constexpr (R < L) {
            if (r == R) zemit<R>(st, sh, a, obase, z, tid);
            else zemit_dispatch<R + 1>(r, st, sh, a, obase, z, tid);
        }
    }
    template <int R>
    static NDWT_DEV void zdispatch(int r, State& st, Shared& sh, const RegTaps& tp, const Args& a, long long obase, int z, bool emit, int tid) {
        if constexpr (R < L) {
            if (r == R) zsyn<R>(st, sh, tp, a, obase, z, emit, tid);
            else zdispatch<R + 1>(r, st, sh, tp, a, obase, z, emit, tid);
        }
    }

#if defined(NDWT_STAMPS) && !defined(NDWT_HOST_EMU)
    // diagnostic build: the clock at six points of planes 100 and 101, every wave of every workgroup (a timeline, not sums;
    // no scheduling fences: the product kernel's overlaps stay) -- tools/timeline_inv.py
#define NDWT_TL(slot) if (a.stamps && (p == 100 || p == 101)) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)); \
        if (threadIdx.x % 64 == 0) a.stamps[(((long long)bid * (NT / 64) + threadIdx.x / 64) * 2 + (p - 100)) * 8 + (slot)] = (long long)t_; }
#else
#define NDWT_TL(slot)
#endif

    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared& sh, const Args& a, const Taps& tpm, int bid) {
        RegTaps tp;
        load_taps(tp, tpm);
        const TileCoord tc = decode_tile(a, bid, TX, TY, VEC4 ? -1 : 4 * (GL + XH), 4 * (GR + XH));
        const long long ibase = uniform_ll(batch_base(tc.batch, a.bsplit, a.in_bstride, a.in_bstride2));
        const long long obase = uniform_ll(batch_base(tc.batch, a.bsplit, a.out_bstride, a.out_bstride2));
        const int zsh = tc.batch * a.zbs;
        const int nsteps = tc.zend - tc.zbeg;
        const int nplanes = nsteps + L - 1;              // planes zbeg-LH .. zend-1+RH
        // Zero-extended slab (z_wrap 3): input plane zm = zbeg + p - (L-1) of iteration p exists for zlo <= zm + zsh < zhi only.  The
        // iterations before the first such plane are skipped (at most L-1 of them: those complete no output plane), the ones after the
        // last run the z stage alone (they only add zeros to the pending sums and emit them) -- a run of m planes at the end of a slab
        // costs m full iterations instead of m + L-1 (the partial sums a rank owes its neighbours: 45 -> see DESIGN.md section 5).
        int p0 = 0, pend = nplanes;                      // iterations [p0, pend): x, y and z stages; [pend, nplanes): z stage only
        if (a.z_wrap == 3) {
            const int first = a.zlo - zsh - tc.zbeg + (L - 1), last = a.zhi - zsh - tc.zbeg + (L - 1);
            p0 = first < 0 ? 0 : (first > L - 1 ? L - 1 : first);
            pend = last < p0 ? p0 : (last > nplanes ? nplanes : last);
        }
        const int zb0 = tc.zbeg - LH + p0;               // the plane iteration p0 reads; iteration p0 + i reads zb0 + i
        const int np = pend - p0;
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            setup(st, a, tc, tid);
            NDWT_SFOR(k, NYI)                             // (a march that starts at p0 > 0 never ran the first taps of the sums pending then)
                NDWT_SFOR(j, L - ZLDS)
                    st.zacc[k][j] = (v2)(T(0));
                NDWT_SEND
                NDWT_SFOR(j, ZLDS)
                    if (tid + k * NT < YITEMS) sh.zl[j][tid + k * NT] = (v2)(T(0));
                NDWT_SEND
            NDWT_SEND
            if (np > 0) load_raw<0>(st, a, ibase, zb0, zsh);
            if constexpr (DEPTH == 2) {
                if (np > 1) load_raw<1>(st, a, ibase, zb0 + 1, zsh);
            }
        });
        // the y and z stages of iteration p (x-synthesised tile in xs[buf]) at raised priority: they are short and end in the
        // plane's store, and under oldest-first arbitration the youngest waves' store waits behind the older waves' queued loads
        auto yz = [&](int p, int buf) __attribute__((always_inline)) {
            const int s = p - (L - 1);                   // output plane this input plane completes (if >= 0)
            NDWT_SETPRIO(1);
            ex.each([&](int tid, State& st) __attribute__((always_inline)) { ysyn(st, sh, tp, buf, tid); });
            NDWT_TL(3)
            ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                zdispatch<0>((p + 1) % L, st, sh, tp, a, obase, tc.zbeg + s, s >= 0, tid);
            });
            NDWT_SETPRIO(0);
            NDWT_TL(4)
        };
        if constexpr (DEPTH == 1) {
            // per plane:  x-synth(i) from registers -> xs[i&1] ; refill with plane i+1 ; barrier ; y/z(i) from xs[i&1]
            for (int i = 0; i < np; ++i) {
                const int p = p0 + i;
                NDWT_TL(0)
                ex.each([&](int, State& st) __attribute__((always_inline)) { shrink_raw<0>(st, a); });
                xstage<0>(ex, sh, tp, i & 1);
                NDWT_TL(1)
                ex.each([&](int, State& st) __attribute__((always_inline)) {
                    if (i + 1 < np) load_raw<0>(st, a, ibase, zb0 + i + 1, zsh);
                });
                NDWT_TL(2)
                ex.barrier();
                yz(p, i & 1);
            }
        } else if (np > 0) {
            // Two register sets, refilled at TWO DIFFERENT POINTS of the plane.  A wave stalls on its own loads until the pipe
            // has accepted them; when every wave refills after its x stage the pipe idles from the barrier until the first
            // x stage ends, and the workgroup then waits for the last wave's loads to be accepted.  Half of the waves refill
            // at the START of an iteration (the set their previous x stage freed), the others right after their x stage.
            //   iteration i (parity Q): early waves: set Q <- plane i+2 ; y/z(i) ; x(i+1) from set 1-Q ; late waves: set 1-Q <- plane i+3 ; barrier
            auto iter = [&](auto q_c, int i) __attribute__((always_inline)) {
                constexpr int Q = decltype(q_c)::value;
                const int p = p0 + i;
                NDWT_TL(0)
                ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                    if (early_refill(tid) && i + 2 < np) load_raw<Q>(st, a, ibase, zb0 + i + 2, zsh);
                });
                NDWT_TL(1)
                yz(p, Q);
                if (i + 1 < np) {
                    ex.each([&](int, State& st) __attribute__((always_inline)) { shrink_raw<1 - Q>(st, a); });
                    xstage<1 - Q>(ex, sh, tp, 1 - Q);
                    NDWT_TL(2)
                    ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                        if (!early_refill(tid) && i + 3 < np) load_raw<1 - Q>(st, a, ibase, zb0 + i + 3, zsh);
                    });
                }
                NDWT_TL(5)
                ex.barrier();
            };
            // prologue: x stage of plane 0; the late waves refill its set with plane 2 (the early ones at the start of iteration 0)
            ex.each([&](int, State& st) __attribute__((always_inline)) { shrink_raw<0>(st, a); });
            xstage<0>(ex, sh, tp, 0);
            ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                if (!early_refill(tid) && 2 < np) load_raw<0>(st, a, ibase, zb0 + 2, zsh);
            });
            ex.barrier();
            for (int i = 0; i < np; i += 2) {
                iter(std::integral_constant<int, 0>{}, i);
                if (i + 1 < np) iter(std::integral_constant<int, 1>{}, i + 1);
            }
        }
        // the iterations past the last plane of a zero-extended slab: nothing to synthesise in x and y, and the z stage would add zeros
        // to the pending sums -- what is left of it is the store of the sum each iteration completes (slot (p + 1) % L, zsyn)
        for (int p = pend; p < nplanes; ++p) {
            const int s = p - (L - 1);
            if (s >= 0) ex.each([&](int tid, State& st) __attribute__((always_inline)) { zemit_dispatch<0>((p + 1) % L, st, sh, a, obase, tc.zbeg + s, tid); });
        }
    }
};

// ------------------------------------------------ level 1 of dec -> shrink -> rec in ONE launch ----
// The consumer of SURVEY 8f-3 (reference README.md:2, "iterative algorithm"): the finest level of a denoising step without its
// seven detail bands ever existing in memory.  The kernel reads the signal x and the reconstructed level-1 approximation,
// RECOMPUTES the level-1 detail coefficients of its haloed tile from x (the analysis of Fwd3: z in a rotating register
// window, y from an LDS tile, x with the neighbouring lanes' values by DPP wave shifts), thresholds them in registers, and
// hands all eight bands to the synthesis stages of Inv3Y -- whose lanes hold exactly these values: 4 x of one haloed row.
// Level 1 then moves x (read twice: once here, once by the approximation-only analysis that feeds the deeper levels), the
// approximation (written and read once each) and the output: 5 volumes instead of 18.  Arithmetic is 2.3x the synthesis
// kernel's (the analysis runs on a tile haloed by L - 1 more samples per axis), which the removed loads pay for: the
// synthesis kernel is bound by its 8 band loads per lane and plane, this one issues 2.
//   per coefficient plane p:  A: z analysis of raw plane -> zs (LDS)  | barrier |  B: y analysis (zs) + x analysis (DPP) +
//   shrink + x synthesis (DPP) -> xs[p & 1]  | barrier |  C: y / z synthesis (Inv3Y::ysyn / zsyn), then A of plane p + 1
// Float, real data, tap stride 1, every axis with the same tap length L <= 8 (one extra lane per side of a haloed row).
template <typename T, int L> struct TapsDen {
    Taps3Y<T, L> syn;        // synthesis taps, the table of Inv3Y
    T alo[3][L];             // analysis low-pass taps (x, y, z); the high-pass ones are derived: ahi[j] = (-1)^j alo[L-1-j]
    T azp[L][2];             // (alo_z[j], ahi_z[j]): the z stage produces (lo, hi) pairs from one broadcast sample
    T axp[L + 1][2];         // (alo_x[k], alo_x[k-1]), k = 0..L, taps outside [0, L) = 0: the x stage works on pairs of adjacent x
};

template <typename T, int L_, int NT_ = 1024, int WPE_ = 4, int ZLDS_ = 0> struct Den3 {
    static_assert(sizeof(T) == 4, "float only");
    static constexpr int L = L_, TX = 64, TY = 32, NT = NT_, WPE = WPE_;
    static constexpr int ALH = L / 2 - 1, ARH = L / 2;   // analysis: samples left / right of the output index
    static constexpr int XH = (ARH + 3) / 4;             // extra lanes per side of a haloed row
    typedef Inv3Y<T, L, TX, TY, NT, true, WPE, 1, 1, ZLDS_, XH> Y;   // (ZLDS_ of the L pending z sums in LDS: Inv3Y)
    static constexpr int NG = Y::NG, RPW = Y::RPW, NW = NT / 64;
    static constexpr int NRS = Y::NR;                    // rows of coefficients under the tile (TY + L - 1)
    static constexpr int NRA = NRS + L - 1;              // rows of raw samples under those
    static constexpr int WCA = 2 * NG;                   // chunks per row of the z-analysed tile
    static_assert(XH == 1 && Y::NRND == 1 && RPW * NW >= NRA, "tile shape: one round of rows, x neighbours one lane away");
    typedef typename VecT<T>::v2 v2;
    typedef typename VecT<T>::v4 v4;
    typedef Lds<T> LD;
    typedef v4 chunk;
    typedef TapsDen<T, L> Taps;
    typedef Fused3Args<T> Args;                          // in[0] = x, in[1] = approximation band, out[0]; shrink_thr / shrink_hard

    struct Shared : Y::Shared {
        chunk zs[NRA][WCA];        // (lo_z, hi_z) pairs of the raw tile: [row][x pair-of-pairs], swizzled like every LDS row here
    };
    struct State : Y::State {
        v4 win[L];                 // raw planes of this lane's z-stage row (4 x), rotating
        v4 nxt;                    // prefetched raw plane
        v4 apx;                    // prefetched approximation values of this lane's coefficient row
        v2 ya[2][4];               // y-analysed (z-bit 0, z-bit 1) pairs of this lane's 4 x: [y-bit][x]; the neighbouring lanes read them
        unsigned aoff;             // byte offset of the z-stage row inside a plane (kNoRow: this lane holds none)
    };
    struct RegTapsA {
        v2 xp[L + 1];              // (alo_x[k], alo_x[k-1])
        v2 ay[L / 2];              // (alo_y[2m], alo_y[2m+1])
        v2 az[L];                  // (alo_z[j], ahi_z[j])
    };
    static NDWT_DEV void load_taps_a(RegTapsA& ta, const Taps& tp) {
        NDWT_SFOR(k, L + 1)
            ta.xp[k] = Y::pinned(v2{tp.axp[k][0], tp.axp[k][1]});
        NDWT_SEND
        NDWT_SFOR(m, L / 2)
            ta.ay[m] = Y::pinned(v2{tp.alo[1][2 * m], tp.alo[1][2 * m + 1]});
        NDWT_SEND
        NDWT_SFOR(j, L)
            ta.az[j] = Y::pinned(v2{tp.azp[j][0], tp.azp[j][1]});
        NDWT_SEND
    }
    // acc(e, e+1) += (t[K], t[K-1]) * w[H]: the x taps of two adjacent outputs for one window entry.  Low-pass: the pair xp[K].
    // High-pass: (ahi[K], ahi[K-1]) = ((-1)^K alo[L-1-K], (-1)^(K-1) alo[L-K]) = the halves of xp[L-K] swapped, one negated.
    template <int H, int K, bool HIGH> static NDWT_DEV void xtap_a(v2& acc, const v2 w, const RegTapsA& ta) {
        if constexpr (!HIGH) Y::template pk_fma_bt<H, false, false, false>(acc, w, ta.xp[K]);
        else Y::template pk_fma_bt<H, true, K % 2 != 0, K % 2 == 0>(acc, w, ta.xp[L - K]);
    }
    // acc += x * analysis low-pass tap J (HIGH = false) or high-pass tap J = (-1)^J low-pass tap L-1-J
    template <int J, bool HIGH> static NDWT_DEV void tap_a(v2& acc, const v2 x, const v2 (&lo)[L / 2]) {
        constexpr int jj = HIGH ? L - 1 - J : J;
        Y::template pk_fma_s<jj & 1, HIGH && (J % 2 == 1)>(acc, x, lo[jj / 2]);
    }

    // z-stage rows: NRA of them, RPW per wave, the lane layout of Inv3Y (lane = (row of the wave, group of 4 x))
    static NDWT_DEV void lane_item_a(int tid, int& ug, int& r, bool& valid) {
        const int lane = tid % 64, wv = tid / 64;
        const int rs = lane / NG;
        ug = lane % NG;
        r = wv * RPW + rs;
        valid = rs < RPW && r < NRA;
        if (r >= NRA) r = NRA - 1;
    }
    static NDWT_DEV void setup_a(State& st, const Args& a, const TileCoord& tc, int tid) {
        int ug, r;
        bool valid;
        lane_item_a(tid, ug, r, valid);
        const int y = modn(tc.y0 - Y::LH - ALH + r, a.n2);
        const int xb = tc.x0 - 4 * (Y::GL + XH) + 4 * ug;
        st.aoff = valid ? (unsigned)(y * a.rs + modn(xb, a.n1)) * (unsigned)sizeof(T) : Y::kNoRow;
        st.nxt = (v4)(T(0));
        st.apx = (v4)(T(0));
        NDWT_SFOR(j, L)
            st.win[j] = (v4)(T(0));
        NDWT_SEND
    }
    // zm: plane index inside the volume (the caller wraps it: one compare per plane instead of a division)
    static NDWT_DEV void load_x(State& st, const Args& a, int zm) {
        if (st.aoff != Y::kNoRow) st.nxt = Y::template gload<v4>(a.in[0] + (long long)zm * a.plane, st.aoff);
    }
    static NDWT_DEV void load_apx(State& st, const Args& a, int zm) {
        if (st.off[0][0] != Y::kNoRow) st.apx = Y::template gload<v4>(a.in[1] + (long long)zm * a.plane, st.off[0][0]);
    }
    static NDWT_DEV int next_plane(int zm, int n3) { return zm + 1 == n3 ? 0 : zm + 1; }

    // z analysis of the newest raw plane: rotation R puts it into slot (R+L-1)%L, tap j reads slot (R+j)%L (Fwd3::zstage)
    template <int R> static NDWT_DEV void zana(State& st, Shared& sh, const RegTapsA& ta, int tid) {
        st.win[(R + L - 1) % L] = st.nxt;
        v2 acc[4];
        acc[0] = acc[1] = acc[2] = acc[3] = (v2)(T(0));
        NDWT_SFOR(j, L)
            const v4 w = st.win[(R + j) % L];
            const v2 w01 = {w[0], w[1]}, w23 = {w[2], w[3]};
            Y::template pk_fma_bt<0, false, false, false>(acc[0], w01, ta.az[j]);    // (lo, hi) += w0 * (alo_z[j], ahi_z[j])
            Y::template pk_fma_bt<1, false, false, false>(acc[1], w01, ta.az[j]);
            Y::template pk_fma_bt<0, false, false, false>(acc[2], w23, ta.az[j]);
            Y::template pk_fma_bt<1, false, false, false>(acc[3], w23, ta.az[j]);
        NDWT_SEND
        int ug, r;
        bool valid;
        lane_item_a(tid, ug, r, valid);
        if (valid) lds_store_run<T, 4>(sh.zs[r], 4 * ug, acc);
    }
    template <int R> static NDWT_DEV void zdispatch_a(int r, State& st, Shared& sh, const RegTapsA& ta, int tid) {
        if constexpr (R < L) {
            if (r == R) zana<R>(st, sh, ta, tid);
            else zdispatch_a<R + 1>(r, st, sh, ta, tid);
        }
    }

    // y analysis of this lane's 4 x of coefficient row r: rows r .. r+L-1 of the z-analysed tile
    static NDWT_DEV void yana(State& st, Shared& sh, const RegTapsA& ta, int tid) {
        int ug, r;
        bool valid;
        Y::lane_item(tid, 0, ug, r, valid);               // (r is clamped to a row of the tile: every lane reads inside zs)
        v2 lo[4], hi[4];
        NDWT_SFOR(e, 4)
            lo[e] = (v2)(T(0));
            hi[e] = (v2)(T(0));
        NDWT_SEND
        const int p0 = LD::S(2 * ug), p1 = LD::S(2 * ug + 1);
        NDWT_SFOR(j, L)
            const chunk c0 = sh.zs[r + j][p0], c1 = sh.zs[r + j][p1];
            const v2 z0 = {c0[0], c0[1]}, z1 = {c0[2], c0[3]}, z2 = {c1[0], c1[1]}, z3 = {c1[2], c1[3]};
            tap_a<j, false>(lo[0], z0, ta.ay); tap_a<j, true>(hi[0], z0, ta.ay);
            tap_a<j, false>(lo[1], z1, ta.ay); tap_a<j, true>(hi[1], z1, ta.ay);
            tap_a<j, false>(lo[2], z2, ta.ay); tap_a<j, true>(hi[2], z2, ta.ay);
            tap_a<j, false>(lo[3], z3, ta.ay); tap_a<j, true>(hi[3], z3, ta.ay);
        NDWT_SEND
        NDWT_SFOR(e, 4)
            st.ya[0][e] = lo[e];
            st.ya[1][e] = hi[e];
        NDWT_SEND
    }

    // x analysis: the window of this lane's 4 outputs reaches ALH samples into the lane before and ARH into the lane after;
    // the results are the eight bands of the lane's 4 x, in the registers the synthesis x stage reads (band = x-bit + 2 y-bit + 4 z-bit)
    template <class Exec> static NDWT_DEV void xana(Exec& ex, State& st, const RegTapsA& ta, int tid) {
        NDWT_SFOR(yb, 2)
            v2 acc[2][2][2];                              // [x-bit][z-bit][outputs (0,1) / (2,3)]: pairs of adjacent x, like Inv3Y::xsyn
            NDWT_SFOR(q, 8)
                acc[q / 4][(q / 2) % 2][q % 2] = (v2)(T(0));
            NDWT_SEND
            NDWT_SFOR(ii, 4 + L - 1)
                constexpr int i = ii - ALH;                // x offset of the window entry from the lane's first x
                constexpr int D = i < 0 ? -1 : (i >= 4 ? 1 : 0);
                constexpr int c = (i + 4) % 4;
                constexpr int K01 = i + ALH, K23 = i - 2 + ALH;      // tap-pair index of this entry for outputs (0,1) / (2,3)
                constexpr bool u01 = K01 >= 0 && K01 <= L, u23 = K23 >= 0 && K23 <= L;
                const v2 w = {NDWT_LANE_SHIFT(ex, tid, D, s.ya[yb][c].x), NDWT_LANE_SHIFT(ex, tid, D, s.ya[yb][c].y)};   // (z-bit 0, z-bit 1)
                NDWT_SFOR(zb, 2)
                    if constexpr (u01) {
                        if constexpr (yb + zb != 0) xtap_a<zb, u01 ? K01 : 0, false>(acc[0][zb][0], w, ta);   // (band 0 is not used: the
                        xtap_a<zb, u01 ? K01 : 0, true>(acc[1][zb][0], w, ta);                                  //  approximation replaces it)
                    }
                    if constexpr (u23) {
                        if constexpr (yb + zb != 0) xtap_a<zb, u23 ? K23 : 0, false>(acc[0][zb][1], w, ta);
                        xtap_a<zb, u23 ? K23 : 0, true>(acc[1][zb][1], w, ta);
                    }
                NDWT_SEND
            NDWT_SEND
            NDWT_SFOR(xb, 2)
                NDWT_SFOR(zb, 2)
                    st.raw[0][0][xb + 2 * yb + 4 * zb] = v4{acc[xb][zb][0].x, acc[xb][zb][0].y, acc[xb][zb][1].x, acc[xb][zb][1].y};
                NDWT_SEND
            NDWT_SEND
            NDWT_SCHED_FENCE();
        NDWT_SEND
    }

    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared& sh, const Args& a, const Taps& tpm, int bid) {
        typename Y::RegTaps tp;
        Y::load_taps(tp, tpm.syn);
        RegTapsA ta;
        load_taps_a(ta, tpm);
        const TileCoord tc = decode_tile(a, bid, TX, TY);
        const int nsteps = tc.zend - tc.zbeg;
        const int nplanes = nsteps + L - 1;              // coefficient planes zbeg - LH .. zend - 1 + RH of the synthesis
        const int zc0 = tc.zbeg - Y::LH;
        int zx = modn(zc0 - ALH, a.n3);                  // next raw plane / next approximation plane, wrapped into the volume
        int za = modn(zc0, a.n3);
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            Y::setup(st, a, tc, tid);
            setup_a(st, a, tc, tid);
        });
        NDWT_SFOR(j, L - 1)                               // raw planes zc0 - ALH .. zc0 + ARH - 1 into window slots 0 .. L-2
            ex.each([&](int, State& st) __attribute__((always_inline)) {
                load_x(st, a, zx);
                st.win[j] = st.nxt;
            });
            zx = next_plane(zx, a.n3);
        NDWT_SEND
        ex.each([&](int, State& st) __attribute__((always_inline)) {
            load_x(st, a, zx);
            load_apx(st, a, za);
        });
        zx = next_plane(zx, a.n3);
        za = next_plane(za, a.n3);
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            zdispatch_a<0>(0, st, sh, ta, tid);
            if (nplanes > 1) load_x(st, a, zx);
        });
        zx = next_plane(zx, a.n3);
        ex.barrier();
        for (int p = 0; p < nplanes; ++p) {
            ex.each([&](int tid, State& st) __attribute__((always_inline)) { yana(st, sh, ta, tid); });
            ex.each([&](int tid, State& st) __attribute__((always_inline)) { xana(ex, st, ta, tid); });
            ex.each([&](int, State& st) __attribute__((always_inline)) {
                NDWT_SFOR(b, 7)                           // the seven detail bands, thresholded where they are
                    shrink4_flat<T>(st.raw[0][0][b + 1], a.shrink_thr, a.shrink_hard);
                NDWT_SEND
                st.raw[0][0][0] = st.apx;                 // band 0: the approximation the deeper levels reconstructed
                if (p + 1 < nplanes) load_apx(st, a, za);
            });
            za = next_plane(za, a.n3);
            ex.each([&](int tid, State& st) __attribute__((always_inline)) { Y::template xsyn<0>(ex, st, sh, tp, p & 1, tid); });
            ex.barrier();
            const int s = p - (L - 1);
            NDWT_SETPRIO(1);
            ex.each([&](int tid, State& st) __attribute__((always_inline)) { Y::ysyn(st, sh, tp, p & 1, tid); });
            ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                Y::template zdispatch<0>((p + 1) % L, st, sh, tp, a, 0LL, tc.zbeg + s, s >= 0, tid);
            });
            NDWT_SETPRIO(0);
            if (p + 1 < nplanes) {
                ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                    zdispatch_a<0>((p + 1) % L, st, sh, ta, tid);
                    if (p + 2 < nplanes) load_x(st, a, zx);
                });
                zx = next_plane(zx, a.n3);
                ex.barrier();
            }
        }
    }
};

// ------------------------------------------------------------------------------------------------
// Fused 2-D level, register-only.  Axis 0 = x (contiguous, n1), axis 1 = y (n2, marched).  One WAVE is one tile:
// lane l owns 4 consecutive x (64 lanes = 256 columns, GL+GR of them halo groups), the y filter runs on a rotating
// register window while the wave marches down a chunk of rows, and the x filter takes its neighbours from the
// adjacent lanes (DPP wave shifts).  No LDS, no barriers; 1 read -> 4 writes (analysis) / 4 reads -> 1 write.
// ------------------------------------------------------------------------------------------------
template <typename T> struct Fused2Args {
    const T* in[4];        // analysis: in[0]; synthesis: the 4 bands
    T* out[4];             // analysis: the 4 bands; synthesis: out[0]
    int n1, n2;
    int nbatch;
    long long in_bstride, out_bstride;
    int ychunk;            // output rows per wave
    int ntx, nyc;          // wave tiles along x, chunks along y
    int y_wrap;            // 1: periodic in y; 0: inputs start `left` rows before local row 0 (slab mode)
    int rs;                // elements between rows (n1; a level dilated by s: s * n1, the s row sub-lattices are the batch items)
    T shrink_thr;          // synthesis: shrink input band b on load when bit b of shrink_mask is set
    int shrink_mask, shrink_hard;
    int nt;                // nontemporal output stores (see Fused3Args)
};

struct Tile2Coord {
    int x0, ybeg, yend, batch;
};
template <typename T> NDWT_DEV Tile2Coord decode_tile2(const Fused2Args<T>& a, int bid, int WX, int halo_l = -1, int halo_r = 0) {
    Tile2Coord tc;
    int tx = bid % a.ntx;
    bid /= a.ntx;
    int yc = bid % a.nyc;
    tc.batch = bid / a.nyc;
    tc.x0 = halo_l >= 0 ? tile_origin(tx, a.ntx, WX, a.n1, halo_l, halo_r) : tx * WX;
    tc.ybeg = yc * a.ychunk;
    tc.yend = tc.ybeg + a.ychunk < a.n2 ? tc.ybeg + a.ychunk : a.n2;
    return tc;
}

template <typename T, int L_, bool VEC4_, int WPE_ = 4, int EW_ = 1> struct Fwd2S {
    static constexpr int L = L_, NT = 64, WPE = WPE_, EW = EW_;
    static constexpr bool VEC4 = VEC4_;
    static constexpr int NE = VEC4 ? 1 : 4;
    static constexpr int LH = L / 2 - 1, RH = L / 2;
    static constexpr int GL = (LH * EW + 3) / 4, GR = (RH * EW + 3) / 4;
    static constexpr int WX = 4 * (64 - GL - GR);        // output columns per wave
    static constexpr int XV = 4 * (1 + GL + GR);
    typedef typename VecT<T>::v2 v2;
    typedef typename VecT<T>::v4 v4;
    typedef Taps3<T, L> Taps;                            // axes 0 (x) and 1 (y) are used
    typedef Fused2Args<T> Args;
    struct Shared { int unused; };
    struct State {
        v4 win[L];         // raw rows, rotating
        v4 nxt;            // prefetched row
        v2 yz[4];          // (lo2, hi2) of this lane's 4 x, read by the neighbouring lanes
        int off[NE];
    };

    static NDWT_DEV void setup(State& st, const Args& a, const Tile2Coord& tc, int tid) {
        int xb = tc.x0 - 4 * GL + 4 * tid;
        NDWT_SFOR(e, NE)
            st.off[e] = modn(xb + e, a.n1);
        NDWT_SEND
    }
    static NDWT_DEV void load_row(State& st, const Args& a, const T* inb, int yraw) {
        long long ym = a.y_wrap ? (long long)modn(yraw, a.n2) : (long long)(yraw + LH);
        const T* p = inb + ym * a.rs;
        if constexpr (VEC4) {
            st.nxt = *reinterpret_cast<const v4*>(p + st.off[0]);
        } else if (st.off[NE - 1] == st.off[0] + 3) {           // 4 contiguous x: one access (VecT::v4u)
            st.nxt = *reinterpret_cast<const typename VecT<T>::v4u*>(p + st.off[0]);
        } else {
            NDWT_SFOR(e, NE)
                st.nxt[e] = p[st.off[e]];
            NDWT_SEND
        }
    }
    template <int R> static NDWT_DEV void ystage(State& st, const Taps& tp) {
        st.win[(R + L - 1) % L] = st.nxt;
        v2 acc[4];
        acc[0] = acc[1] = acc[2] = acc[3] = (v2)(T(0));
        NDWT_SFOR(j, L)
            v4 w = st.win[(R + j) % L];
            v2 t = {tp.lo[1][j], tp.hi[1][j]};
            acc[0] += t * w[0]; acc[1] += t * w[1]; acc[2] += t * w[2]; acc[3] += t * w[3];
        NDWT_SEND
        st.yz[0] = acc[0]; st.yz[1] = acc[1]; st.yz[2] = acc[2]; st.yz[3] = acc[3];
    }
    template <int R> static NDWT_DEV void ydispatch(int r, State& st, const Taps& tp) {
        if constexpr (R < L) {
            if (r == R) ystage<R>(st, tp);
            else ydispatch<R + 1>(r, st, tp);
        }
    }
    static NDWT_DEV void prologue(State& st, const Args& a, const T* inb, int ybeg) {
        NDWT_SFOR(j, L - 1)
            load_row(st, a, inb, ybeg - LH + j);
            st.win[j] = st.nxt;
        NDWT_SEND
    }
    // every lane executes the shifts; stores are predicated
    template <class Exec> static NDWT_DEV void xstage(Exec& ex, State& st, const Taps& tp, const Args& a, const Tile2Coord& tc,
                                                      long long obase, int y, int tid) {
        v2 xlo[4], xhi[4];
        NDWT_SFOR(e, 4)
            xlo[e] = (v2)(T(0));
            xhi[e] = (v2)(T(0));
        NDWT_SEND
        NDWT_SFOR(i, XV)
            constexpr int D = i / 4 - GL;
            constexpr int c = i % 4;
            {
                v2 v = {NDWT_LANE_SHIFT(ex, tid, D, s.yz[c].x), NDWT_LANE_SHIFT(ex, tid, D, s.yz[c].y)};
                NDWT_SFOR(e, 4)
                    constexpr int dj = i - 4 * GL - e;                     // = (j - LH) * EW
                    if constexpr (dj % EW == 0) {
                        constexpr int j = dj / EW + LH;
                        if constexpr (j >= 0 && j < L) {
                            xlo[e] += tp.lo[0][j] * v;
                            xhi[e] += tp.hi[0][j] * v;
                        }
                    }
                NDWT_SEND
            }
        NDWT_SEND
        int gx = tc.x0 + 4 * (tid - GL);
        if (tid < GL || tid >= 64 - GR || gx >= a.n1) return;
        long long off = obase + (long long)y * a.rs + gx;
        v4 o0 = {xlo[0].x, xlo[1].x, xlo[2].x, xlo[3].x}, o1 = {xhi[0].x, xhi[1].x, xhi[2].x, xhi[3].x};
        v4 o2 = {xlo[0].y, xlo[1].y, xlo[2].y, xlo[3].y}, o3 = {xhi[0].y, xhi[1].y, xhi[2].y, xhi[3].y};
        if constexpr (VEC4) {
            stream_store(reinterpret_cast<v4*>(a.out[0] + off), o0, a.nt);
            stream_store(reinterpret_cast<v4*>(a.out[1] + off), o1, a.nt);
            stream_store(reinterpret_cast<v4*>(a.out[2] + off), o2, a.nt);
            stream_store(reinterpret_cast<v4*>(a.out[3] + off), o3, a.nt);
        } else if (gx + 3 < a.n1) {
            typedef typename VecT<T>::v4u v4u;
            *reinterpret_cast<v4u*>(a.out[0] + off) = o0;
            *reinterpret_cast<v4u*>(a.out[1] + off) = o1;
            *reinterpret_cast<v4u*>(a.out[2] + off) = o2;
            *reinterpret_cast<v4u*>(a.out[3] + off) = o3;
        } else {
            NDWT_SFOR(e, 4)
                if (gx + e < a.n1) { a.out[0][off + e] = o0[e]; a.out[1][off + e] = o1[e]; a.out[2][off + e] = o2[e]; a.out[3][off + e] = o3[e]; }
            NDWT_SEND
        }
    }

    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared&, const Args& a, const Taps& tp, int bid) {
        const Tile2Coord tc = decode_tile2(a, bid, WX, VEC4 ? -1 : 4 * GL, 4 * GR);
        const T* inb = a.in[0] + (long long)tc.batch * a.in_bstride;
        const long long obase = (long long)tc.batch * a.out_bstride;
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            setup(st, a, tc, tid);
            prologue(st, a, inb, tc.ybeg);
            load_row(st, a, inb, tc.ybeg + RH);
        });
        const int nsteps = tc.yend - tc.ybeg;
        for (int s = 0; s < nsteps; ++s) {
            const int y = tc.ybeg + s;
            ex.each([&](int, State& st) __attribute__((always_inline)) {
                ydispatch<0>(s % L, st, tp);
                if (s + 1 < nsteps) load_row(st, a, inb, y + 1 + RH);
            });
            ex.each([&](int tid, State& st) __attribute__((always_inline)) { xstage(ex, st, tp, a, tc, obase, y, tid); });
        }
    }
};

// ---- NLEV analysis levels of an image in ONE launch (the reference's level loop, nd_dwt_2D.m:141-197, cascaded inside the march) ----
// The reference applies the same un-dilated filters at every level, so level l + 1 is the same one-level analysis applied to the
// approximation of level l.  A wave marches down its chunk of rows once: the raw row it reads enters the y window of level 1, the
// approximation row level 1 completes enters the y window of level 2 in the same step, and so on -- the approximations between the
// levels live in registers only (a per-level launch writes each to memory and the next launch reads it back: 1 + 3 NLEV + 1 volumes
// move instead of 5 NLEV).  Level l runs l RH rows behind the raw row and is valid l GL / l GR lanes inside the wave, so a wave stores
// 4 (64 - NLEV (GL + GR)) columns of every band and starts NLEV (L/2 - 1) rows above its chunk.  Same FMAs in the same order as NLEV
// launches of Fwd2S.  Float / double real data, rows of whole groups of 4 scalars, periodic in y.
template <typename T> struct Fused2CArgs {
    const T* in;           // the image (or the approximation the cascade starts from)
    T* out[10];            // [0] the approximation of the LAST level of the cascade; level l = 1 .. NLEV (1 = first / finest of the launch):
                           // its detail bands b = 1 .. 3 at [1 + 3 (NLEV - l) + (b - 1)] -- the reference's band order for NLEV levels
    int n1, n2;
    int ychunk, ntx, nyc;
    int rs;                // elements between rows
    int nt;                // nontemporal stores
    int mode;              // A/B (tools/bench2d_cascade.py): bit 0 = every level computes from its first input row on (no take-in-only steps)
};

template <typename T, int L_, int NLEV_, int WPE_ = 2> struct Fwd2C {
    static constexpr int L = L_, NLEV = NLEV_, NT = 64, WPE = WPE_;
    static constexpr int LH = L / 2 - 1, RH = L / 2;
    static constexpr int GL = (LH + 3) / 4, GR = (RH + 3) / 4;
    // output columns per wave: the lanes whose values are valid at every level, rounded down to whole 128-byte lines (8 lanes) so that no
    // line of any band is shared by two waves (a partly written line of a nontemporal stream is a read-modify-write in memory)
    static constexpr int WX = 4 * ((64 - NLEV * (GL + GR)) / 8 * 8);
    static constexpr int XV = 4 * (1 + GL + GR);
    static_assert(NLEV >= 2 && NLEV <= 3 && WX > 0, "two or three levels per launch");
    typedef typename VecT<T>::v2 v2;
    typedef typename VecT<T>::v4 v4;
    typedef Taps3<T, L> Taps;                            // axes 0 (x) and 1 (y) are used
    typedef Fused2CArgs<T> Args;
    struct Shared { int unused; };
    struct State {
        v4 win[NLEV][L];   // the last L input rows of every level, rotating
        v4 nxt;            // prefetched raw row
        v4 cur;            // the row entering the level being processed (raw row, then the approximation rows)
        v2 yz[4];          // (lo2, hi2) of this lane's 4 x, read by the neighbouring lanes
        int off;
    };

    static NDWT_DEV void load_row(State& st, const Args& a, int yraw) {
        st.nxt = *reinterpret_cast<const v4*>(a.in + (long long)modn(yraw, a.n2) * a.rs + st.off);
    }
    template <int LEV, int R> static NDWT_DEV void ystage(State& st, const Taps& tp) {
        st.win[LEV][(R + L - 1) % L] = st.cur;
        v2 acc[4];
        acc[0] = acc[1] = acc[2] = acc[3] = (v2)(T(0));
        NDWT_SFOR(j, L)
            v4 w = st.win[LEV][(R + j) % L];
            v2 t = {tp.lo[1][j], tp.hi[1][j]};
            acc[0] += t * w[0]; acc[1] += t * w[1]; acc[2] += t * w[2]; acc[3] += t * w[3];
        NDWT_SEND
        st.yz[0] = acc[0]; st.yz[1] = acc[1]; st.yz[2] = acc[2]; st.yz[3] = acc[3];
    }
    // x filter of level LEV (every lane executes the shifts), stores of its bands for output row y if `emit`; its approximation row -> st.cur
    template <int LEV, class Exec> static NDWT_DEV void xstage(Exec& ex, State& st, const Taps& tp, const Args& a, int x0, int y, bool emit,
                                                               int tid) {
        v2 xlo[4], xhi[4];
        NDWT_SFOR(e, 4)
            xlo[e] = (v2)(T(0));
            xhi[e] = (v2)(T(0));
        NDWT_SEND
        NDWT_SFOR(i, XV)
            constexpr int D = i / 4 - GL;
            constexpr int c = i % 4;
            {
                v2 v = {NDWT_LANE_SHIFT(ex, tid, D, s.yz[c].x), NDWT_LANE_SHIFT(ex, tid, D, s.yz[c].y)};
                NDWT_SFOR(e, 4)
                    constexpr int j = i - 4 * GL - e + LH;
                    if constexpr (j >= 0 && j < L) {
                        xlo[e] += tp.lo[0][j] * v;
                        xhi[e] += tp.hi[0][j] * v;
                    }
                NDWT_SEND
            }
        NDWT_SEND
        const v4 o0 = {xlo[0].x, xlo[1].x, xlo[2].x, xlo[3].x}, o1 = {xhi[0].x, xhi[1].x, xhi[2].x, xhi[3].x};
        const v4 o2 = {xlo[0].y, xlo[1].y, xlo[2].y, xlo[3].y}, o3 = {xhi[0].y, xhi[1].y, xhi[2].y, xhi[3].y};
        st.cur = o0;
        const int gx = x0 + 4 * (tid - NLEV * GL);
        if (!emit || tid < NLEV * GL || tid >= NLEV * GL + WX / 4 || gx >= a.n1) return;
        const long long off = (long long)y * a.rs + gx;
        constexpr int d0 = 1 + 3 * (NLEV - 1 - LEV);      // LEV counts from 0 here
        if constexpr (LEV == NLEV - 1) stream_store(reinterpret_cast<v4*>(a.out[0] + off), o0, a.nt);
        stream_store(reinterpret_cast<v4*>(a.out[d0] + off), o1, a.nt);
        stream_store(reinterpret_cast<v4*>(a.out[d0 + 1] + off), o2, a.nt);
        stream_store(reinterpret_cast<v4*>(a.out[d0 + 2] + off), o3, a.nt);
    }
    // one step of the march with rotation R: raw row r has arrived; level LEV completes its output row r - (LEV + 1) RH
    template <int R, class Exec> static NDWT_DEV void step(Exec& ex, const Args& a, const Taps& tp, int x0, int ybeg, int yend, int r, bool more) {
        ex.each([&](int, State& st) __attribute__((always_inline)) {
            st.cur = st.nxt;
            if (more) load_row(st, a, r + 1);             // prefetch for the next step
        });
        const int s = r - (ybeg - NLEV * LH);             // step of the march
        NDWT_SFOR(lev, NLEV)
            // Level lev's input rows are valid from step lev (L - 1) on and its window is full L - 1 steps later: before the first it
            // has nothing to do, between the two it only takes the row in (what it would compute is march-in garbage; wave-uniform tests)
            if (lev == 0 || s >= (lev + 1) * (L - 1) || ((a.mode & 1) && s >= lev * (L - 1))) {
                ex.each([&](int, State& st) __attribute__((always_inline)) { ystage<lev, R>(st, tp); });
                const int y = r - (lev + 1) * RH;
                const bool emit = y >= ybeg && y < yend;
                ex.each([&](int tid, State& st) __attribute__((always_inline)) { xstage<lev>(ex, st, tp, a, x0, y, emit, tid); });
            } else if (s >= lev * (L - 1)) {
                ex.each([&](int, State& st) __attribute__((always_inline)) { st.win[lev][(R + L - 1) % L] = st.cur; });
            }
        NDWT_SEND
    }
    template <int R, class Exec> static NDWT_DEV void dispatch(int rot, Exec& ex, const Args& a, const Taps& tp, int x0, int ybeg, int yend, int r,
                                                               bool more) {
        if constexpr (R < L) {
            if (rot == R) step<R>(ex, a, tp, x0, ybeg, yend, r, more);
            else dispatch<R + 1>(rot, ex, a, tp, x0, ybeg, yend, r, more);
        }
    }

    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared&, const Args& a, const Taps& tp, int bid) {
        const int tx = bid % a.ntx, yc = bid / a.ntx;
        const int x0 = tx * WX;
        const int ybeg = yc * a.ychunk;
        const int yend = ybeg + a.ychunk < a.n2 ? ybeg + a.ychunk : a.n2;
        const int r0 = ybeg - NLEV * LH;                  // first raw row: level NLEV's first output row needs it
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            st.off = modn(x0 - 4 * NLEV * GL + 4 * tid, a.n1);
            NDWT_SFOR(lev, NLEV)
                NDWT_SFOR(j, L)
                    st.win[lev][j] = (v4)(T(0));
                NDWT_SEND
            NDWT_SEND
            if (!(a.mode & 1)) {
                NDWT_SFOR(j, L - 1)                      // steps 0 .. L-2: level 0 only takes rows in -- all of them in flight at once
                    load_row(st, a, r0 + j);
                    st.win[0][j] = st.nxt;
                NDWT_SEND
                load_row(st, a, r0 + L - 1);
            } else {
                load_row(st, a, r0);
            }
        });
        const int nsteps = (yend - ybeg) + NLEV * (L - 1);
        // step s >= L-1 runs with rotation (s + 1) % L: its row lands in slot L-1 at s = L-1, behind the rows of the steps before it
        if (!(a.mode & 1)) {
            for (int s = L - 1; s < nsteps; ++s) dispatch<0>((s + 1) % L, ex, a, tp, x0, ybeg, yend, r0 + s, s + 1 < nsteps);
        } else {
            for (int s = 0; s < nsteps; ++s) dispatch<0>(s % L, ex, a, tp, x0, ybeg, yend, r0 + s, s + 1 < nsteps);
        }
    }
};

template <typename T, int L_, bool VEC4_, int WPE_ = 4, int EW_ = 1> struct Inv2S {
    static constexpr int L = L_, NT = 64, WPE = WPE_, EW = EW_;
    static constexpr bool VEC4 = VEC4_;
    static constexpr int NE = VEC4 ? 1 : 4;
    static constexpr int LH = L / 2, RH = L / 2 - 1;
    static constexpr int GL = (LH * EW + 3) / 4, GR = (RH * EW + 3) / 4;
    static constexpr int WX = 4 * (64 - GL - GR);
    static constexpr int XV = 4 * (1 + GL + GR);
    typedef typename VecT<T>::v2 v2;
    typedef typename VecT<T>::v4 v4;
    typedef Taps3<T, L> Taps;
    typedef Fused2Args<T> Args;
    struct Shared { int unused; };
    struct State {
        T yacc[L][4];      // y-synthesis in scatter form: partial sums of the next L output rows (4 x each)
        v4 raw[4];         // 4 x of every band of the newest row
        int off[NE];
    };
    static NDWT_DEV void setup(State& st, const Args& a, const Tile2Coord& tc, int tid) {
        int xb = tc.x0 - 4 * GL + 4 * tid;
        NDWT_SFOR(e, NE)
            st.off[e] = modn(xb + e, a.n1);
        NDWT_SEND
    }
    static NDWT_DEV void load_row(State& st, const Args& a, long long ibase, int yraw) {
        long long ym = a.y_wrap ? (long long)modn(yraw, a.n2) : (long long)(yraw + LH);
        if constexpr (VEC4) {
            NDWT_SFOR(b, 4)
                st.raw[b] = *reinterpret_cast<const v4*>(a.in[b] + ibase + ym * a.rs + st.off[0]);
            NDWT_SEND
        } else if (st.off[NE - 1] == st.off[0] + 3) {           // 4 contiguous x: one access per band (VecT::v4u)
            NDWT_SFOR(b, 4)
                st.raw[b] = *reinterpret_cast<const typename VecT<T>::v4u*>(a.in[b] + ibase + ym * a.rs + st.off[0]);
            NDWT_SEND
        } else {
            NDWT_SFOR(b, 4)
                NDWT_SFOR(e, NE)
                    st.raw[b][e] = (a.in[b] + ibase + ym * a.rs)[st.off[e]];
                NDWT_SEND
            NDWT_SEND
        }
    }
    // x-synthesis of the newest row via lane shifts, then y-synthesis (scatter); rotation R as in Inv3S
    template <int R, class Exec>
    static NDWT_DEV void step(Exec& ex, State& st, const Taps& tp, const Args& a, const Tile2Coord& tc, long long obase, int y,
                              bool emit, int tid) {
        T p0[4], p1[4];    // x-synthesis of y-bit 0 / y-bit 1 = the (a, d) inputs of the y-synthesis; scalars, not pairs:
        NDWT_SFOR(e, 4)    // see Inv3S::xsyn (pairs would be built right behind the prefetch loads)
            p0[e] = T(0);
            p1[e] = T(0);
        NDWT_SEND
        NDWT_SFOR(i, XV)
            constexpr int D = i / 4 - GL;
            constexpr int c = i % 4;
            {
                const T wa0 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[0][c]);   // x-bit 0, y-bit 0
                const T wd0 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[1][c]);   // x-bit 1, y-bit 0
                const T wa1 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[2][c]);   // x-bit 0, y-bit 1
                const T wd1 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[3][c]);   // x-bit 1, y-bit 1
                NDWT_SFOR(e, 4)
                    constexpr int dj = i - 4 * GL - e;
                    if constexpr (dj % EW == 0) {
                        constexpr int j = dj / EW + LH;
                        if constexpr (j >= 0 && j < L) {
                            p0[e] += tp.lo[0][j] * wa0;
                            p0[e] += tp.hi[0][j] * wd0;
                            p1[e] += tp.lo[0][j] * wa1;
                            p1[e] += tp.hi[0][j] * wd1;
                        }
                    }
                NDWT_SEND
            }
        NDWT_SEND
        NDWT_SFOR(j, L)
            constexpr int slot = ((R - 1 - j) % L + L) % L;
            NDWT_SFOR(e, 4)
                const T c = tp.lo[1][j] * p0[e] + tp.hi[1][j] * p1[e];
                if constexpr (j == 0) st.yacc[slot][e] = c;
                else st.yacc[slot][e] += c;
            NDWT_SEND
        NDWT_SEND
        if (!emit) return;
        constexpr int done = ((R - L) % L + L) % L;
        int gx = tc.x0 + 4 * (tid - GL);
        if (tid < GL || tid >= 64 - GR || gx >= a.n1) return;
        long long off = obase + (long long)y * a.rs + gx;
        if constexpr (VEC4) {
            stream_store(reinterpret_cast<v4*>(a.out[0] + off), v4{st.yacc[done][0], st.yacc[done][1], st.yacc[done][2], st.yacc[done][3]}, a.nt);
        } else if (gx + 3 < a.n1) {
            *reinterpret_cast<typename VecT<T>::v4u*>(a.out[0] + off) = v4{st.yacc[done][0], st.yacc[done][1], st.yacc[done][2], st.yacc[done][3]};
        } else {
            NDWT_SFOR(e, 4)
                if (gx + e < a.n1) a.out[0][off + e] = st.yacc[done][e];
            NDWT_SEND
        }
    }
    template <int R, class Exec>
    static NDWT_DEV void dispatch(int r, Exec& ex, State& st, const Taps& tp, const Args& a, const Tile2Coord& tc, long long obase,
                                  int y, bool emit, int tid) {
        if constexpr (R < L) {
            if (r == R) step<R>(ex, st, tp, a, tc, obase, y, emit, tid);
            else dispatch<R + 1>(r, ex, st, tp, a, tc, obase, y, emit, tid);
        }
    }
    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared&, const Args& a, const Taps& tp, int bid) {
        const Tile2Coord tc = decode_tile2(a, bid, WX, VEC4 ? -1 : 4 * GL, 4 * GR);
        const long long ibase = (long long)tc.batch * a.in_bstride;
        const long long obase = (long long)tc.batch * a.out_bstride;
        const int nsteps = tc.yend - tc.ybeg;
        const int nrows = nsteps + L - 1;
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            setup(st, a, tc, tid);
            load_row(st, a, ibase, tc.ybeg - LH);
        });
        for (int p = 0; p < nrows; ++p) {
            const int s = p - (L - 1);
            ex.each([&](int, State& st) __attribute__((always_inline)) {               // fused thresholding of the detail bands
                if (a.shrink_mask) {
                    NDWT_SFOR(b, 4)
                        if ((a.shrink_mask >> b) & 1) shrink4<T, EW>(st.raw[b], a.shrink_thr, a.shrink_hard);
                    NDWT_SEND
                }
            });
            ex.each([&](int tid, State& st) __attribute__((always_inline)) {
                dispatch<0>((p + 1) % L, ex, st, tp, a, tc, obase, tc.ybeg + s, s >= 0, tid);
            });
            ex.each([&](int, State& st) __attribute__((always_inline)) {
                if (p + 1 < nrows) load_row(st, a, ibase, tc.ybeg - LH + p + 1);
            });
        }
    }
};

// The same 2-D synthesis with PD rows of band loads in flight per wave and the row loop UNROLLED IN GROUPS OF L, so that the rotation of
// the y sums and the slot of every row are compile-time constants in straight-line code.  Inv2S issues the loads of row p+1 after it has
// consumed row p and dispatches on p % L at run time: hipcc's wait-count insertion cannot follow loads across that dispatch and waits
// vmcnt(0) at the first use, so a wave exposes one full memory latency per row and the kernel needs 8 waves per CU on chunks so short that
// the L-1 prologue rows re-read 20 % of every chunk (FETCH_SIZE 1.40x the bands at 4096^2).  Here the compiler sees which load feeds which
// row and waits for exactly that one.  Float / double real data in rows of whole groups of 4 scalars; everything else keeps Inv2S.
// PK_ (float): the arithmetic as packed FMAs on pairs of adjacent x outputs -- x stage: (out[e], out[e+1]) += w * (t[k], t[k-1]) with the
// neighbour's sample w broadcast from one half of its register pair and the tap pairs of Taps3Y pinned in SGPRs; y stage: a tap broadcast
// from an SGPR pair.  104 packed FMAs per row and lane instead of 192 scalar ones (L = 8).
template <typename T, int L_, int PD_ = 2, int WPE_ = 2, bool PK_ = false> struct Inv2P {
    static constexpr int L = L_, NT = 64, WPE = WPE_, PD = PD_;
    static constexpr bool PK = PK_;
    static_assert(L % PD == 0, "the depth divides the tap length: a row's slot is p % PD with p = group base + k");
    static_assert(!PK_ || sizeof(T) == 4, "packed form: float only (v_pk_fma_f32)");
    static constexpr int LH = L / 2, RH = L / 2 - 1;
    static constexpr int GL = (LH + 3) / 4, GR = (RH + 3) / 4;
    static constexpr int WX = 4 * (64 - GL - GR);
    static constexpr int XV = 4 * (1 + GL + GR);
    typedef typename VecT<T>::v4 v4;
    typedef typename VecT<T>::v2 v2;
    typedef Taps3Y<T, L> Taps;      // (its first two members are Taps3<T, L>; the x tap pairs are read by the packed form only)
    typedef Fused2Args<T> Args;
    struct Shared { int unused; };
    struct State {
        T yacc[L][4];      // y-synthesis in scatter form: partial sums of the next L output rows (4 x each)
        v4 raw[PD][4];     // 4 x of every band of the rows in flight: row p in slot p % PD
        int off;
    };
    struct RegT {          // PK: tap pairs pinned in SGPRs
        v2 xl[PK ? L + 1 : 1], xh[PK ? L + 1 : 1];    // (t[k], t[k-1]) of the x low-pass / high-pass taps
        v2 yl[PK ? L / 2 : 1], yh[PK ? L / 2 : 1];    // (t[2m], t[2m+1]) of the y taps
    };
    static NDWT_DEV void load_regt(RegT& rt, const Taps& tp) {
        if constexpr (PK) {
            NDWT_SFOR(k, L + 1)
                rt.xl[k] = PkF32::pinned(v2{tp.xplo[k][0], tp.xplo[k][1]});
                rt.xh[k] = PkF32::pinned(v2{tp.xphi[k][0], tp.xphi[k][1]});
            NDWT_SEND
            NDWT_SFOR(m, L / 2)
                rt.yl[m] = PkF32::pinned(v2{tp.lo[1][2 * m], tp.lo[1][2 * m + 1]});
                rt.yh[m] = PkF32::pinned(v2{tp.hi[1][2 * m], tp.hi[1][2 * m + 1]});
            NDWT_SEND
        }
    }
    template <int S> static NDWT_DEV void load_row(State& st, const Args& a, long long ibase, int yraw) {
        const long long ym = a.y_wrap ? (long long)modn(yraw, a.n2) : (long long)(yraw + LH);
        NDWT_SFOR(b, 4)
            st.raw[S][b] = *reinterpret_cast<const v4*>(a.in[b] + ibase + ym * a.rs + st.off);
        NDWT_SEND
    }
    template <int S> static NDWT_DEV void pre(State& st, const Args& a) {
        if (a.shrink_mask) {
            NDWT_SFOR(b, 4)
                if ((a.shrink_mask >> b) & 1) shrink4<T, 1>(st.raw[S][b], a.shrink_thr, a.shrink_hard);
            NDWT_SEND
        }
    }
    // row p = group base + K (rotation R = (K + 1) % L, slot K % PD): x-synthesis via lane shifts, y-synthesis in scatter form
    // the packed form of step<K>
    template <int K, class Exec>
    static NDWT_DEV void step_pk(Exec& ex, State& st, const RegT& rt, const Args& a, const Tile2Coord& tc, long long obase, int y, bool emit, int tid) {
        constexpr int R = (K + 1) % L, S = K % PD;
        v2 P[2][2];                                       // [y band][x outputs (0, 1) / (2, 3)]
        P[0][0] = P[0][1] = P[1][0] = P[1][1] = (v2)(T(0));
        NDWT_SFOR(ii, XV / 2)
            constexpr int i0 = 2 * ii;
            constexpr int D = i0 / 4 - GL;
            constexpr int c = i0 % 4;                     // 0 or 2: the register pair (c, c + 1) of the lane D away
            v2 w[4];
            NDWT_SFOR(b, 4)
                w[b] = v2{NDWT_LANE_SHIFT(ex, tid, D, s.raw[S][b][c]), NDWT_LANE_SHIFT(ex, tid, D, s.raw[S][b][c + 1])};
            NDWT_SEND
            NDWT_SFOR(h, 2)
                constexpr int k0 = i0 + h - 4 * GL + LH;  // tap pair of the outputs (0, 1); (2, 3): two taps earlier
                NDWT_SFOR(q, 2)
                    constexpr int k = k0 - 2 * q;
                    if constexpr (k >= 0 && k <= L) {
                        PkF32::fma_bt<h, false, false, false>(P[0][q], w[0], rt.xl[k]);
                        PkF32::fma_bt<h, false, false, false>(P[0][q], w[1], rt.xh[k]);
                        PkF32::fma_bt<h, false, false, false>(P[1][q], w[2], rt.xl[k]);
                        PkF32::fma_bt<h, false, false, false>(P[1][q], w[3], rt.xh[k]);
                    }
                NDWT_SEND
            NDWT_SEND
        NDWT_SEND
        NDWT_SFOR(j, L)
            constexpr int slot = ((R - 1 - j) % L + L) % L;
            NDWT_SFOR(q, 2)
                v2 acc;
                if constexpr (j == 0) acc = (v2)(T(0));
                else acc = v2{st.yacc[slot][2 * q], st.yacc[slot][2 * q + 1]};
                PkF32::fma_s<j % 2, false>(acc, P[0][q], rt.yl[j / 2]);
                PkF32::fma_s<j % 2, false>(acc, P[1][q], rt.yh[j / 2]);
                st.yacc[slot][2 * q] = acc.x;
                st.yacc[slot][2 * q + 1] = acc.y;
            NDWT_SEND
        NDWT_SEND
        if (!emit) return;
        constexpr int done = ((R - L) % L + L) % L;
        const int gx = tc.x0 + 4 * (tid - GL);
        if (tid < GL || tid >= 64 - GR || gx >= a.n1) return;
        stream_store(reinterpret_cast<v4*>(a.out[0] + obase + (long long)y * a.rs + gx),
                     v4{st.yacc[done][0], st.yacc[done][1], st.yacc[done][2], st.yacc[done][3]}, a.nt);
    }
    template <int K, class Exec>
    static NDWT_DEV void step(Exec& ex, State& st, const Taps& tp, const Args& a, const Tile2Coord& tc, long long obase, int y, bool emit, int tid) {
        constexpr int R = (K + 1) % L, S = K % PD;
        T p0[4], p1[4];
        NDWT_SFOR(e, 4)
            p0[e] = T(0);
            p1[e] = T(0);
        NDWT_SEND
        NDWT_SFOR(i, XV)
            constexpr int D = i / 4 - GL;
            constexpr int c = i % 4;
            {
                const T wa0 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[S][0][c]);
                const T wd0 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[S][1][c]);
                const T wa1 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[S][2][c]);
                const T wd1 = NDWT_LANE_SHIFT(ex, tid, D, s.raw[S][3][c]);
                NDWT_SFOR(e, 4)
                    constexpr int j = i - 4 * GL - e + LH;
                    if constexpr (j >= 0 && j < L) {
                        p0[e] += tp.lo[0][j] * wa0;
                        p0[e] += tp.hi[0][j] * wd0;
                        p1[e] += tp.lo[0][j] * wa1;
                        p1[e] += tp.hi[0][j] * wd1;
                    }
                NDWT_SEND
            }
        NDWT_SEND
        NDWT_SFOR(j, L)
            constexpr int slot = ((R - 1 - j) % L + L) % L;
            NDWT_SFOR(e, 4)
                const T c = tp.lo[1][j] * p0[e] + tp.hi[1][j] * p1[e];
                if constexpr (j == 0) st.yacc[slot][e] = c;
                else st.yacc[slot][e] += c;
            NDWT_SEND
        NDWT_SEND
        if (!emit) return;
        constexpr int done = ((R - L) % L + L) % L;
        const int gx = tc.x0 + 4 * (tid - GL);
        if (tid < GL || tid >= 64 - GR || gx >= a.n1) return;
        stream_store(reinterpret_cast<v4*>(a.out[0] + obase + (long long)y * a.rs + gx),
                     v4{st.yacc[done][0], st.yacc[done][1], st.yacc[done][2], st.yacc[done][3]}, a.nt);
    }
    // one row of a group: threshold, consume, refill the slot with row p + PD (separate passes: neighbouring lanes read this lane's registers)
    template <int K, class Exec>
    static NDWT_DEV void row(Exec& ex, const Taps& tp, const RegT& rt, const Args& a, const Tile2Coord& tc, long long ibase, long long obase, int p, int nrows) {
        const int s = p - (L - 1);
        ex.each([&](int, State& st) __attribute__((always_inline)) { pre<K % PD>(st, a); });
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            if constexpr (PK) step_pk<K>(ex, st, rt, a, tc, obase, tc.ybeg + s, s >= 0, tid);
            else step<K>(ex, st, tp, a, tc, obase, tc.ybeg + s, s >= 0, tid);
        });
        ex.each([&](int, State& st) __attribute__((always_inline)) {
            if (p + PD < nrows) load_row<K % PD>(st, a, ibase, tc.ybeg - LH + p + PD);
        });
    }
    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared&, const Args& a, const Taps& tp, int bid) {
        const Tile2Coord tc = decode_tile2(a, bid, WX);
        const long long ibase = (long long)tc.batch * a.in_bstride;
        const long long obase = (long long)tc.batch * a.out_bstride;
        const int nrows = tc.yend - tc.ybeg + L - 1;
        RegT rt;
        load_regt(rt, tp);
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            st.off = modn(tc.x0 - 4 * GL + 4 * tid, a.n1);
            NDWT_SFOR(q, PD)
                if (q < nrows) load_row<q>(st, a, ibase, tc.ybeg - LH + q);
            NDWT_SEND
        });
        int p = 0;
        for (; p + L <= nrows; p += L) {                  // whole groups: straight-line code, no test per row
            NDWT_SFOR(k, L)
                row<k>(ex, tp, rt, a, tc, ibase, obase, p + k, nrows);
            NDWT_SEND
        }
        NDWT_SFOR(k, L - 1)                               // the last, partial group
            if (p + k < nrows) row<k>(ex, tp, rt, a, tc, ibase, obase, p + k, nrows);
        NDWT_SEND
    }
};

// ---- NLEV synthesis levels of an image in ONE launch (the reference's level loop of rec, nd_dwt_2D.m:200-252, cascaded inside the march) ----
// The mirror image of Fwd2C: a wave marches down its chunk once; the row the coarsest level of the cascade completes is, in the same
// step, the approximation row of the level below it, whose detail rows are read next to it -- 1 + 3 NLEV volumes are read and one is
// written (one launch per level reads 4 and writes 1 each).  Arithmetic of Inv2P's packed form (pairs of adjacent x outputs per
// v_pk_fma_f32, tap pairs pinned in SGPRs), rows in groups of L so that rotations and load slots are compile-time constants; every level
// computes in every step (what it makes of march-in rows never reaches an output row of the chunk), so the code is straight-line and the
// compiler counts the loads in flight exactly.  Float real data, rows of whole groups of 4 scalars, periodic in y.
template <typename T> struct Fused2CIArgs {
    const T* in[10];       // [0] the approximation of the coarsest level; level l (1 = finest of the launch) detail bands b = 1 .. 3 at
                           // [1 + 3 (NLEV - l) + (b - 1)] (the order of Fused2CArgs::out)
    T* out;
    int n1, n2;
    int ychunk, ntx, nyc;
    int rs;
    int nt;
    T shrink_thr;          // ndwt_denoise: soft (or, shrink_hard, hard) thresholding of every detail band as its rows are loaded; 0 = off
    int shrink_on, shrink_hard;
};

template <typename T, int L_, int NLEV_, int PD_ = 1, int WPE_ = 2> struct Inv2C {
    static_assert(sizeof(T) == 4, "packed form: float only (v_pk_fma_f32)");
    static constexpr int L = L_, NLEV = NLEV_, NT = 64, WPE = WPE_, PD = PD_;
    static_assert(L % PD == 0 && NLEV >= 2 && NLEV <= 3, "the depth divides the tap length; two or three levels per launch");
    static constexpr int LH = L / 2, RH = L / 2 - 1;
    static constexpr int GL = (LH + 3) / 4, GR = (RH + 3) / 4;
    static constexpr int WX = 4 * ((64 - NLEV * (GL + GR)) / 8 * 8);   // whole 128-byte lines per wave and row (see Fwd2C)
    static constexpr int XV = 4 * (1 + GL + GR);
    typedef typename VecT<T>::v4 v4;
    typedef typename VecT<T>::v2 v2;
    typedef Taps3Y<T, L> Taps;
    typedef Fused2CIArgs<T> Args;
    struct Shared { int unused; };
    struct State {
        T yacc[NLEV][L][4];          // per level: partial sums of its next L output rows (4 x each), rotating
        v4 raw[NLEV][PD][4];         // rows in flight: level 0 (the coarsest) all 4 bands; finer levels [1 .. 3] = their detail bands
        v4 cur[NLEV];                // cur[c]: the row level c - 1 has just completed = the approximation row of level c (c >= 1)
        int off;
    };
    struct RegT {
        v2 xl[L + 1], xh[L + 1];     // (t[k], t[k-1]) of the x low-pass / high-pass taps
        v2 yl[L / 2], yh[L / 2];     // (t[2m], t[2m+1]) of the y taps
    };
    static NDWT_DEV void load_regt(RegT& rt, const Taps& tp) {
        NDWT_SFOR(k, L + 1)
            rt.xl[k] = PkF32::pinned(v2{tp.xplo[k][0], tp.xplo[k][1]});
            rt.xh[k] = PkF32::pinned(v2{tp.xphi[k][0], tp.xphi[k][1]});
        NDWT_SEND
        NDWT_SFOR(m, L / 2)
            rt.yl[m] = PkF32::pinned(v2{tp.lo[1][2 * m], tp.lo[1][2 * m + 1]});
            rt.yh[m] = PkF32::pinned(v2{tp.hi[1][2 * m], tp.hi[1][2 * m + 1]});
        NDWT_SEND
    }
    // band rows of level C (0 = coarsest) for march step p: level C consumes row rr0 + p - C RH (rows are periodic: any index loads)
    template <int C, int S> static NDWT_DEV void load_rows(State& st, const Args& a, int row) {
        const long long o = (long long)modn(row, a.n2) * a.rs + st.off;
        if constexpr (C == 0) {
            NDWT_SFOR(b, 4)
                st.raw[0][S][b] = *reinterpret_cast<const v4*>(a.in[b] + o);
            NDWT_SEND
        } else {
            NDWT_SFOR(b, 3)
                st.raw[C][S][1 + b] = *reinterpret_cast<const v4*>(a.in[1 + 3 * C + b] + o);
            NDWT_SEND
        }
    }
    // level C of step K of a group (rotation R = (K + 1) % L, slot K % PD): x-synthesis via lane shifts, y-synthesis in scatter form;
    // the row it completes -> cur[C + 1], or -- the finest level -- the output row y if `emit`
    template <int C, int K, class Exec>
    static NDWT_DEV void level(Exec& ex, State& st, const RegT& rt, const Args& a, int x0, int y, bool emit, int tid) {
        constexpr int R = (K + 1) % L, S = K % PD;
        v2 P[2][2];                                       // [y band][x outputs (0, 1) / (2, 3)]
        P[0][0] = P[0][1] = P[1][0] = P[1][1] = (v2)(T(0));
        NDWT_SFOR(ii, XV / 2)
            constexpr int i0 = 2 * ii;
            constexpr int D = i0 / 4 - GL;
            constexpr int c = i0 % 4;
            v2 w[4];
            if constexpr (C == 0) w[0] = v2{NDWT_LANE_SHIFT(ex, tid, D, s.raw[0][S][0][c]), NDWT_LANE_SHIFT(ex, tid, D, s.raw[0][S][0][c + 1])};
            else w[0] = v2{NDWT_LANE_SHIFT(ex, tid, D, s.cur[C][c]), NDWT_LANE_SHIFT(ex, tid, D, s.cur[C][c + 1])};
            NDWT_SFOR(b, 3)
                w[1 + b] = v2{NDWT_LANE_SHIFT(ex, tid, D, s.raw[C][S][1 + b][c]), NDWT_LANE_SHIFT(ex, tid, D, s.raw[C][S][1 + b][c + 1])};
            NDWT_SEND
            NDWT_SFOR(h, 2)
                constexpr int k0 = i0 + h - 4 * GL + LH;
                NDWT_SFOR(q, 2)
                    constexpr int k = k0 - 2 * q;
                    if constexpr (k >= 0 && k <= L) {
                        PkF32::fma_bt<h, false, false, false>(P[0][q], w[0], rt.xl[k]);
                        PkF32::fma_bt<h, false, false, false>(P[0][q], w[1], rt.xh[k]);
                        PkF32::fma_bt<h, false, false, false>(P[1][q], w[2], rt.xl[k]);
                        PkF32::fma_bt<h, false, false, false>(P[1][q], w[3], rt.xh[k]);
                    }
                NDWT_SEND
            NDWT_SEND
        NDWT_SEND
        NDWT_SFOR(j, L)
            constexpr int slot = ((R - 1 - j) % L + L) % L;
            NDWT_SFOR(q, 2)
                v2 acc;
                if constexpr (j == 0) acc = (v2)(T(0));
                else acc = v2{st.yacc[C][slot][2 * q], st.yacc[C][slot][2 * q + 1]};
                PkF32::fma_s<j % 2, false>(acc, P[0][q], rt.yl[j / 2]);
                PkF32::fma_s<j % 2, false>(acc, P[1][q], rt.yh[j / 2]);
                st.yacc[C][slot][2 * q] = acc.x;
                st.yacc[C][slot][2 * q + 1] = acc.y;
            NDWT_SEND
        NDWT_SEND
        constexpr int done = ((R - L) % L + L) % L;
        const v4 o = {st.yacc[C][done][0], st.yacc[C][done][1], st.yacc[C][done][2], st.yacc[C][done][3]};
        if constexpr (C + 1 < NLEV) {
            st.cur[C + 1] = o;
        } else {
            const int gx = x0 + 4 * (tid - NLEV * GL);
            if (!emit || tid < NLEV * GL || tid >= NLEV * GL + WX / 4 || gx >= a.n1) return;
            stream_store(reinterpret_cast<v4*>(a.out + (long long)y * a.rs + gx), o, a.nt);
        }
    }
    // step p of the march (K = p % L): every level consumes its row, then the consumed load slots are refilled with the rows of step p + PD
    template <int K, class Exec>
    static NDWT_DEV void row(Exec& ex, const RegT& rt, const Args& a, int x0, int ybeg, int yend, int rr0, int p, int nrows) {
        const int y = rr0 + p - NLEV * RH;
        if (a.shrink_on) {                                // thresholding fused into the reconstruction (the detail rows this step consumes)
            ex.each([&](int, State& st) __attribute__((always_inline)) {
                NDWT_SFOR(c, NLEV)
                    NDWT_SFOR(b, 3)
                        shrink4_flat<T>(st.raw[c][K % PD][1 + b], a.shrink_thr, a.shrink_hard);
                    NDWT_SEND
                NDWT_SEND
            });
        }
        NDWT_SFOR(c, NLEV)
            ex.each([&](int tid, State& st) __attribute__((always_inline)) { level<c, K>(ex, st, rt, a, x0, y, y >= ybeg && y < yend, tid); });
        NDWT_SEND
        ex.each([&](int, State& st) __attribute__((always_inline)) {
            if (p + PD < nrows) {
                NDWT_SFOR(c, NLEV)
                    load_rows<c, K % PD>(st, a, rr0 + p + PD - c * RH);
                NDWT_SEND
            }
        });
    }
    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared&, const Args& a, const Taps& tp, int bid) {
        const int tx = bid % a.ntx, yc = bid / a.ntx;
        const int x0 = tx * WX;
        const int ybeg = yc * a.ychunk;
        const int yend = ybeg + a.ychunk < a.n2 ? ybeg + a.ychunk : a.n2;
        const int rr0 = ybeg - NLEV * LH;                 // the coarsest level's first row: the chunk's first output row needs it
        const int nrows = (yend - ybeg) + NLEV * (L - 1);
        RegT rt;
        load_regt(rt, tp);
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            st.off = modn(x0 - 4 * NLEV * GL + 4 * tid, a.n1);
            NDWT_SFOR(c, NLEV)                            // (sums and rows of levels whose march-in has not reached them yet: finite values)
                st.cur[c] = (v4)(T(0));
                NDWT_SFOR(j, L)
                    NDWT_SFOR(e, 4)
                        st.yacc[c][j][e] = T(0);
                    NDWT_SEND
                NDWT_SEND
            NDWT_SEND
            NDWT_SFOR(q, PD)
                NDWT_SFOR(c, NLEV)
                    load_rows<c, q>(st, a, rr0 + q - c * RH);
                NDWT_SEND
            NDWT_SEND
        });
        int p = 0;
        for (; p + L <= nrows; p += L) {                  // whole groups: straight-line code, no test per row
            NDWT_SFOR(k, L)
                row<k>(ex, rt, a, x0, ybeg, yend, rr0, p + k, nrows);
            NDWT_SEND
        }
        NDWT_SFOR(k, L - 1)                               // the last, partial group
            if (p + k < nrows) row<k>(ex, rt, a, x0, ybeg, yend, rr0, p + k, nrows);
        NDWT_SEND
    }
};

// ------------------------------------------------------------------------------------------------
// One non-contiguous axis per launch, marched with the filter window in registers (no LDS, no halo re-reads):
// 1 read -> 2 writes (analysis) / 2 reads -> 1 write (synthesis, scatter form).  The array is [outer][N][inner];
// a thread owns 4 consecutive `inner` elements and marches a chunk of the axis.  Used for the outer axis of 4-D
// volumes and for the outer axes of the per-axis path (complex data included: inner then carries the factor 2).
// ------------------------------------------------------------------------------------------------
template <typename T, int L> struct MarchTaps {
    T lo[L];
    T hi[L];
};
template <typename T> struct MarchArgs {
    const T* in0;          // analysis: input; synthesis: low band
    const T* in1;          // synthesis: high band
    T* out0;               // analysis: low band; synthesis: output
    T* out1;               // analysis: high band
    long long inner;       // multiple of 4
    long long n;           // output length of the axis
    long long n_in;        // input length (n, or n + L - 1 in slab mode)
    long long outer;
    int chunk, nchunks;    // output planes per thread, chunks along the axis
    int wrap;              // 1 periodic, 0 input starts `left` planes before output plane 0
    long long ngroups;     // inner / 4
};

template <typename T, int L_, bool SYN> struct AxisMarch {
    static constexpr int L = L_, NT = 256, WPE = 4;
    static constexpr int LH = SYN ? L / 2 : L / 2 - 1, RH = SYN ? L / 2 - 1 : L / 2;
    typedef typename VecT<T>::v4 v4;
    typedef MarchTaps<T, L> Taps;
    typedef MarchArgs<T> Args;
    struct Shared { int unused; };
    struct State {
        v4 win[L];         // analysis: raw planes; synthesis: partial sums of the pending outputs
        v4 nxt[2];
        long long base_in, base_out;
        int active;
    };
    static NDWT_DEV long long in_plane(const Args& a, long long zraw) {
        return a.wrap ? modn64(zraw, a.n) : zraw + LH;
    }
    static NDWT_DEV void load(State& st, const Args& a, long long zraw) {
        long long off = st.base_in + in_plane(a, zraw) * a.inner;
        st.nxt[0] = *reinterpret_cast<const v4*>(a.in0 + off);
        if constexpr (SYN) st.nxt[1] = *reinterpret_cast<const v4*>(a.in1 + off);
    }
    template <int R> static NDWT_DEV void step(State& st, const Args& a, const Taps& tp, long long z, bool emit) {
        if constexpr (!SYN) {
            st.win[(R + L - 1) % L] = st.nxt[0];
            v4 lo = (v4)(T(0)), hi = (v4)(T(0));
            NDWT_SFOR(j, L)
                lo += tp.lo[j] * st.win[(R + j) % L];
                hi += tp.hi[j] * st.win[(R + j) % L];
            NDWT_SEND
            if (emit) {
                long long off = st.base_out + z * a.inner;
                *reinterpret_cast<v4*>(a.out0 + off) = lo;
                *reinterpret_cast<v4*>(a.out1 + off) = hi;
            }
        } else {
            NDWT_SFOR(j, L)
                constexpr int slot = ((R - 1 - j) % L + L) % L;
                const v4 c = tp.lo[j] * st.nxt[0] + tp.hi[j] * st.nxt[1];
                if constexpr (j == 0) st.win[slot] = c;
                else st.win[slot] += c;
            NDWT_SEND
            if (emit) *reinterpret_cast<v4*>(a.out0 + st.base_out + z * a.inner) = st.win[((R - L) % L + L) % L];
        }
    }
    template <int R> static NDWT_DEV void dispatch(int r, State& st, const Args& a, const Taps& tp, long long z, bool emit) {
        if constexpr (R < L) {
            if (r == R) step<R>(st, a, tp, z, emit);
            else dispatch<R + 1>(r, st, a, tp, z, emit);
        }
    }
    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared&, const Args& a, const Taps& tp, int bid) {
        // bid -> (block of 256 items, chunk); an item = (outer index, group of 4 contiguous elements), so short
        // contiguous runs (the dilated contiguous axis has inner = stride) still fill the workgroup
        const long long items = a.ngroups * a.outer;
        const long long iblocks = (items + NT - 1) / NT;
        const long long ib = bid % iblocks;
        const int ck = (int)(bid / iblocks);
        const long long zbeg = (long long)ck * a.chunk;
        const long long zend = zbeg + a.chunk < a.n ? zbeg + a.chunk : a.n;
        const int nsteps = (int)(zend - zbeg);
        // analysis: output z needs planes z-LH .. z+RH; the window is primed with the first L-1 of them.
        // synthesis: input plane q contributes to outputs q-RH .. q+LH; output z is complete after plane z+RH.
        const int nplanes = nsteps + L - 1;
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            const long long it = ib * NT + tid;
            st.active = it < items;
            const long long iv = st.active ? it : 0;
            const long long o = iv / a.ngroups, gi = iv % a.ngroups;
            st.base_in = o * a.n_in * a.inner + gi * 4;
            st.base_out = o * a.n * a.inner + gi * 4;
            load(st, a, zbeg - LH);
        });
        for (int p = 0; p < nplanes; ++p) {
            const int s = p - (L - 1);                    // output step completed by plane p
            ex.each([&](int, State& st) __attribute__((always_inline)) {
                dispatch<0>((p + 1) % L, st, a, tp, zbeg + s, s >= 0 && st.active);
                load(st, a, zbeg - LH + (p + 1 < nplanes ? p + 1 : p));
            });
        }
    }
};

// ------------------------------------------------------------------------------------------------
// The CONTIGUOUS axis, one launch (1 read -> 2 writes / 2 reads -> 1 write), for data the fused kernels do not cover:
// 1-D signals and interleaved complex arrays (EW = 2 floats per element: taps step over the (re, im) pairs).
// A wave owns a segment of one row; lane l holds 4 consecutive scalars and takes its neighbours from the adjacent
// lanes with DPP wave shifts (several hops for long filters / complex data).  Rows are [outer][n*EW] scalars.
// ------------------------------------------------------------------------------------------------
template <typename T> struct AxisXArgs {
    const T* in0;
    const T* in1;          // synthesis: high band
    T* out0;
    T* out1;               // analysis: high band
    long long row;         // scalars per row = n * EW
    long long outer;       // rows
    long long nseg;        // wave segments per row
};

template <typename T, int L_, bool SYN, int EW_, bool VEC4_> struct AxisX {
    static constexpr int L = L_, EW = EW_, NT = 256, WPE = 4;
    static constexpr bool VEC4 = VEC4_;
    static constexpr int LH = SYN ? L / 2 : L / 2 - 1, RH = SYN ? L / 2 - 1 : L / 2;
    static constexpr int GL = (LH * EW + 3) / 4, GR = (RH * EW + 3) / 4;
    static constexpr int WX = 4 * (64 - GL - GR);        // output scalars per wave segment
    static_assert(GL + GR < 32, "filter too long for one wave");
    typedef typename VecT<T>::v4 v4;
    typedef MarchTaps<T, L> Taps;
    typedef AxisXArgs<T> Args;
    struct Shared { int unused; };
    struct State {
        v4 raw[2];         // this lane's 4 scalars of the input band(s)
        long long ibase, obase;
        int xg;            // first scalar of this lane inside the row (may be outside [0,row) for halo lanes)
        int valid;
    };
    static NDWT_DEV void load(State& st, const Args& a) {
        NDWT_SFOR(b, (SYN ? 2 : 1))
            const T* p = (b == 0 ? a.in0 : a.in1) + st.ibase;
            if constexpr (VEC4) {
                st.raw[b] = *reinterpret_cast<const v4*>(p + modn64(st.xg, a.row));
            } else if (st.xg >= 0 && (long long)st.xg + 3 < a.row) {      // 4 contiguous scalars: one access (VecT::v4u)
                st.raw[b] = *reinterpret_cast<const typename VecT<T>::v4u*>(p + st.xg);
            } else {
                NDWT_SFOR(e, 4)
                    st.raw[b][e] = p[modn64((long long)st.xg + e, a.row)];
                NDWT_SEND
            }
        NDWT_SEND
    }
    template <class Exec> static NDWT_DEV void compute(Exec& ex, State& st, const Args& a, const Taps& tp, int tid) {
        v4 o0 = (v4)(T(0)), o1 = (v4)(T(0));
        NDWT_SFOR(e, 4)
            NDWT_SFOR(j, L)
                constexpr int idx = e + (j - LH) * EW + 4 * GL;      // scalar index in the wave-local window, >= 0
                constexpr int D = idx / 4 - GL;
                constexpr int c = idx % 4;
                if constexpr (!SYN) {
                    const T v = NDWT_LANE_SHIFT(ex, tid, D, s.raw[0][c]);
                    o0[e] += tp.lo[j] * v;
                    o1[e] += tp.hi[j] * v;
                } else {
                    o0[e] += tp.lo[j] * NDWT_LANE_SHIFT(ex, tid, D, s.raw[0][c]);
                    o0[e] += tp.hi[j] * NDWT_LANE_SHIFT(ex, tid, D, s.raw[1][c]);
                }
            NDWT_SEND
        NDWT_SEND
        const int lane = tid % 64;
        if (!st.valid || lane < GL || lane >= 64 - GR || st.xg >= a.row) return;
        if constexpr (VEC4) {
            *reinterpret_cast<v4*>(a.out0 + st.obase + st.xg) = o0;
            if constexpr (!SYN) *reinterpret_cast<v4*>(a.out1 + st.obase + st.xg) = o1;
        } else if ((long long)st.xg + 3 < a.row) {
            *reinterpret_cast<typename VecT<T>::v4u*>(a.out0 + st.obase + st.xg) = o0;
            if constexpr (!SYN) *reinterpret_cast<typename VecT<T>::v4u*>(a.out1 + st.obase + st.xg) = o1;
        } else {
            NDWT_SFOR(e, 4)
                if (st.xg + e < a.row) {
                    a.out0[st.obase + st.xg + e] = o0[e];
                    if constexpr (!SYN) a.out1[st.obase + st.xg + e] = o1[e];
                }
            NDWT_SEND
        }
    }
    template <class Exec> static NDWT_DEV void block(Exec& ex, Shared&, const Args& a, const Taps& tp, int bid) {
        ex.each([&](int tid, State& st) __attribute__((always_inline)) {
            const long long item = (long long)bid * (NT / 64) + tid / 64;       // one wave per (row, segment)
            const long long nitems = a.outer * a.nseg;
            st.valid = item < nitems;
            const long long it = st.valid ? item : nitems - 1;
            const long long o = it / a.nseg, sg = it % a.nseg;
            st.ibase = o * a.row;
            st.obase = o * a.row;
            // (row segments like tiles: the ones that reach the end of a row whose length is not a multiple of 4 are anchored there)
            const int seg0 = (VEC4 || a.row >= (1LL << 30)) ? (int)(sg * WX) : tile_origin((int)sg, (int)a.nseg, WX, (int)a.row, 4 * GL, 4 * GR);
            st.xg = seg0 + 4 * (tid % 64 - GL);
            load(st, a);
        });
        ex.each([&](int tid, State& st) __attribute__((always_inline)) { compute(ex, st, a, tp, tid); });
    }
};

}  // namespace ndwt
