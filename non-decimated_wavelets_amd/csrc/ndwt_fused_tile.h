// ndwt_fused_tile.h -- tile shape of the fused 3-D kernels (shared by launch geometry, kernels and the host emulator)
#pragma once
namespace ndwt {
template <typename T> struct Fused3Tile;
template <> struct Fused3Tile<float>  { static constexpr int TX = 64, TY = 16, NT = 256, RY = 4; };
template <> struct Fused3Tile<double> { static constexpr int TX = 64, TY = 8,  NT = 256, RY = 4; };
}  // namespace ndwt
