// ndwt_fused_tile.h -- tile shapes of the fused 3-D kernels (shared by launch geometry, kernels and the host emulator)
#pragma once
namespace ndwt {
// TX x TY output tile per workgroup of NT threads; RY = output rows per y-stage item; WPE = waves per SIMD the
// register budget is sized for (amdgpu_waves_per_eu).  V selects a variant (NDWT_VARIANT / ndwt_plan_set_variant):
// every variant computes the same values, they differ in occupancy and instruction mix only.
constexpr int kFused3Variants = 4;
template <typename T, bool INVERSE, int V> struct Fused3Tile;
// float, analysis
template <> struct Fused3Tile<float, false, 0> { static constexpr int TX = 64, TY = 16, NT = 256, RY = 4, WPE = 3; };   // db1, db2 and small volumes
template <> struct Fused3Tile<float, false, 1> { static constexpr int TX = 64, TY = 16, NT = 512, RY = 4, WPE = 2; };   // one column per thread: 18 / 20 taps, complex data with 10 / 12 taps
template <> struct Fused3Tile<float, false, 2> { static constexpr int TX = 64, TY = 32, NT = 1024, RY = 4, WPE = 4; };  // tall tile, one workgroup per CU: the default for 6 .. 16 taps on volumes with >= 32 such tiles
template <> struct Fused3Tile<float, false, 3> { static constexpr int TX = 64, TY = 16, NT = 256, RY = 4, WPE = 3; };
template <> struct Fused3Tile<float, false, 6> { static constexpr int TX = 64, TY = 32, NT = 1024, RY = 2, WPE = 4; };  // tall tile, y items of 2 rows (twice the waves in the y stage)
// float, synthesis
template <> struct Fused3Tile<float, true, 0>  { static constexpr int TX = 64, TY = 16, NT = 256, RY = 2, WPE = 2; };
template <> struct Fused3Tile<float, true, 1>  { static constexpr int TX = 64, TY = 32, NT = 1024, RY = 1, WPE = 4; };  // lane-shift kernel (Inv3S), tall tile: the float default
template <> struct Fused3Tile<float, true, 2>  { static constexpr int TX = 64, TY = 32, NT = 512, RY = 1, WPE = 2; };   // lane-shift kernel, tall tile, 512 threads x 2 items: long filters (no spills)
template <> struct Fused3Tile<float, true, 3>  { static constexpr int TX = 64, TY = 16, NT = 256, RY = 2, WPE = 2; };   // LDS kernel (Inv3), for A/B runs
// float synthesis of a level dilated by 4 (x taps step over 4 scalars: 23 lanes per haloed row, 2 rows per wave, a 32-value
// x window): 512 threads with the 256-register budget, two rounds of rows
template <> struct Fused3Tile<float, true, 4>  { static constexpr int TX = 64, TY = 16, NT = 512, RY = 1, WPE = 2; };
// double
template <int V> struct Fused3Tile<double, false, V> { static constexpr int TX = 64, TY = 8, NT = 256, RY = 4, WPE = 2; };
template <> struct Fused3Tile<double, false, 1>      { static constexpr int TX = 64, TY = 16, NT = 512, RY = 4, WPE = 2; };   // long filters (db5, db6)
template <int V> struct Fused3Tile<double, true, V>  { static constexpr int TX = 64, TY = 8, NT = 256, RY = 4, WPE = 2; };
template <> struct Fused3Tile<double, true, 1>       { static constexpr int TX = 64, TY = 16, NT = 512, RY = 2, WPE = 2; };   // lane-shift kernel (Inv3S): the double default
// double, 10 / 12 taps: 64 x 8 tiles with 512 threads -- the only shapes that keep these windows in 256 registers without spills
template <> struct Fused3Tile<double, false, 5>      { static constexpr int TX = 64, TY = 8, NT = 512, RY = 4, WPE = 2; };
template <> struct Fused3Tile<double, true, 5>       { static constexpr int TX = 64, TY = 8, NT = 512, RY = 1, WPE = 2; };
// float synthesis default (pair-packed kernel Inv3Y): 64 x 32 tile, 1024 threads, one workgroup per CU
constexpr int kInv3YTX = 64, kInv3YTY = 32;
// The pair-packed synthesis tile: the haloed rows (TY + L - 1) must fit the 16 waves of the workgroup at three rows per wave,
// i.e. at most 21 lanes per haloed row.  Real data: 64 wide up to 18 taps (64x32, 18 taps 64x24), 20 taps 48x28 (22 lanes per
// haloed 64-wide row would leave two rows per wave).  Interleaved complex (ew = 2, TX in scalars): 64 wide up to 10 taps, 12 taps 48.
// two register sets of band loads (staggered refill) fit the 128 registers of the 1024-thread workgroup without spills for 2, 8 and
// 10 taps as they are, and for 12 taps with 6 of the 12 pending z sums in LDS (Inv3Y::ZLDS): 512^3 db6 1.37 -> 1.30 ms per launch.
// (4 and 6 taps fit the same way -- 2 / 4 sums in LDS -- and run SLOWER than one register set: 1.05 vs 1.02, 1.18 vs 1.08 ms.)
// 18 and 20 taps (one register set): 8 of the pending z sums in LDS -- without them the instances spill 7 .. 19 registers.
constexpr int inv3y_zlds(int L, int depth, int ew = 1) { return ew != 1 ? 0 : (depth == 2 && L == 12) ? 6 : (depth == 2 && L == 10) ? 2 : (L >= 18 ? 8 : 0); }
// A level dilated by 4 (ew = 4, TX in scalars, up to 8 taps): the x halo is L - 1 whole lanes; 8 taps: 23 lanes per haloed row, two rows per
// wave, 32 haloed rows -> 64 x 24.
constexpr int inv3y_ty(int L, int ew = 1) { return ew == 4 ? (L <= 6 ? kInv3YTY : 24) : ew == 2 ? kInv3YTY : (L <= 16 ? kInv3YTY : (L <= 18 ? 24 : 28)); }
constexpr int inv3y_tx(int L, int ew = 1) { return ew == 4 ? kInv3YTX : ew == 2 ? (L <= 10 ? kInv3YTX : 48) : (L <= 18 ? kInv3YTX : 48); }
}  // namespace ndwt
