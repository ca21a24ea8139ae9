// ndwt_geom.h -- launch geometry of the fused kernels (shared by the library and the host emulator).
#pragma once
#include <cstdint>

#include "ndwt_device.h"

namespace ndwt {

// fills the tiling fields of `a` (n1,n2,n3,nbatch must be set).  Workgroups march `zchunk` output
// planes each; the chunk is sized so the grid has at most ~target_blocks workgroups (the number that is
// resident on the chip at once: one full round, no partial second round) while the (L-1)-plane march
// prologue stays a small fraction of the chunk.
// With more tiles than resident slots the chunk count is the one with the fewest plane steps over all rounds (a partial last
// round costs a whole one: 4-D analysis, 1024 tiles, 768 resident: 2.19 ms with 1024 workgroups, 1.88 ms with 2048; 768^3
// synthesis, 288 tiles on 256 CUs: 5.6 ms in one chunk, 8 chunks = 9 full rounds).
template <typename T>
inline void fused3_geometry(Fused3Args<T>& a, int TX, int TY, int Lp, int target_blocks = 2048, int force_zchunk = 0) {
    a.ntx = (a.n1 + TX - 1) / TX;
    a.nty = (a.n2 + TY - 1) / TY;
    a.plane = (long long)a.n1 * a.n2;             // dense volume; a dilated launch overrides rs / plane afterwards
    a.rs = a.n1;
    long long per_chunk = (long long)a.ntx * a.nty * a.nbatch;
    int want = (int)(target_blocks / per_chunk);
    int min_chunk = 4 * (Lp - 1);                 // prologue <= 25 % of the chunk
    if (min_chunk < 8) min_chunk = 8;
    if (want < 1 || per_chunk * want * 4 < (long long)target_blocks * 3) {
        // More tiles than resident slots, or a single round that would leave more than a quarter of them empty (384^3 double
        // synthesis: 144 tiles on 256 CUs): the launch runs in rounds of `target_blocks` workgroups and a partial last round
        // costs a whole one (768^3 synthesis: 288 tiles on 256 CUs = 2 rounds of 775 planes, 5.6 ms; cut into 8 chunks = 9 full
        // rounds of 103 planes, 3.6 ms).  Pick the chunk count with the fewest plane steps, a little in favour of fewer rounds.
        double best = 0.0;
        want = 1;
        const int max_chunks = a.n3 / min_chunk > 0 ? a.n3 / min_chunk : 1;
        for (int c = 1; c <= max_chunks && c <= 64; ++c) {
            const long long rounds = (per_chunk * c + target_blocks - 1) / target_blocks;
            const double cost = (double)rounds * ((a.n3 + c - 1) / c + Lp - 1) * (1.0 + 0.01 * (double)(rounds - 1));
            if (c == 1 || cost < best) { best = cost; want = c; }
        }
    }
    int zc = (a.n3 + want - 1) / want;
    // small problems leave most CUs idle at that chunk size and every workgroup is a serial march of ~2 us plane steps:
    // fill the chip instead (one round of workgroups, chunks down to 2 planes; the prologue re-reads come from L2 at
    // these sizes).  64^3 db4 L3: 433 -> 129 us per dec+rec, 128^3: 503 -> 181 us.
    if (per_chunk * ((a.n3 + min_chunk - 1) / min_chunk) < target_blocks / 2) min_chunk = 2;
    if (zc < min_chunk) zc = min_chunk;
    if (zc > a.n3) zc = a.n3;
    if (force_zchunk > 0) zc = force_zchunk < a.n3 ? force_zchunk : a.n3;
    a.nzc = (a.n3 + zc - 1) / zc;
    if (force_zchunk <= 0) zc = (a.n3 + a.nzc - 1) / a.nzc;      // equal chunks: the slowest workgroup sets the time
    a.zchunk = zc;
    a.nzc = (a.n3 + zc - 1) / zc;
}

// the fused kernels keep intra-plane offsets in 32-bit ints
inline bool fused3_fits(long long n1, long long n2, long long n3, long long nbatch) {
    return n1 * n2 < (1LL << 31) && n3 < (1LL << 30) && nbatch < (1LL << 20) && n1 >= 1 && n2 >= 1 && n3 >= 1;
}

// fused 2-D kernels: one wave (64 lanes x 4 columns, minus halo groups) marches `ychunk` rows
template <typename T>
inline void fused2_geometry(Fused2Args<T>& a, int WX, int Lp, int target_waves = 4096, int force_ychunk = 0) {
    a.ntx = (a.n1 + WX - 1) / WX;
    a.rs = a.n1;                                  // dense image; a dilated launch overrides it afterwards
    long long per_chunk = (long long)a.ntx * a.nbatch;
    int want = (int)((target_waves + per_chunk - 1) / per_chunk);
    if (want < 1) want = 1;
    int yc = (a.n2 + want - 1) / want;
    int min_chunk = 4 * (Lp - 1);
    if (min_chunk < 8) min_chunk = 8;
    if (per_chunk * ((a.n2 + min_chunk - 1) / min_chunk) < target_waves / 2) min_chunk = 2;   // small images: see above (256^2: 153 -> 45 us)
    if (yc < min_chunk) yc = min_chunk;
    if (yc > a.n2) yc = a.n2;
    if (force_ychunk > 0) yc = force_ychunk < a.n2 ? force_ychunk : a.n2;
    a.ychunk = yc;
    a.nyc = (a.n2 + yc - 1) / yc;
}

}  // namespace ndwt
