// ndwt_fused.h -- host-callable launchers of the fused level kernels (one TU per dtype x direction
// so the build parallelises).  Returns 0 on success, -1 if no instantiation covers (Lp), else a
// hipError_t.
#pragma once
#include <hip/hip_runtime.h>

#include "ndwt_device.h"
#include "ndwt_fused_tile.h"

namespace ndwt {

struct FusedTapsD {       // per axis (0 = x, 1 = y, 2 = z), zero-padded to Lp, double precision
    int Lp;
    double lo[3][kMaxTaps];
    double hi[3][kMaxTaps];
};

// tile shape a variant uses (for the launch geometry)
void fused3_tile_shape(bool f64, bool inverse, int variant, int Lp, int* TX, int* TY, int ew);

int launch_fwd3_f32(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* taps_dev, hipStream_t s);
int launch_inv3_f32(const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* taps_dev, hipStream_t s);
int launch_fwd3_pin_f32(const Fused3Args<float>& a, int Lp, const void* taps_dev, hipStream_t s);   // 10 / 12 / 14 taps, tall tile, taps pinned in SGPRs (vec4 data)
int launch_fwd3_f64(const Fused3Args<double>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* taps_dev, hipStream_t s);
int launch_inv3_f64(const Fused3Args<double>& a, const FusedTapsD& t, bool vec4, int variant, int ew, const void* taps_dev, hipStream_t s);

// float synthesis of real data with tap stride 1, tap lengths <= 8: the pair-packed kernel (Inv3Y) on a 64 x 32 tile with 1024
// threads; depth = register sets of band loads (2: staggered refill, aligned volumes only)
int launch_inv3y_f32(const Fused3Args<float>& a, int Lp, bool vec4, int depth, const void* taps_dev, hipStream_t s, int uniform_yz = 0);
// the same kernel with its x stage in scatter form (rows of whole groups of 4, two register sets); -1: no instance for this tap length
int launch_inv3ys_f32(const Fused3Args<float>& a, int Lp, int depth, const void* taps_dev, hipStream_t s, int uniform_yz);
int launch_inv3yc_f32(const Fused3Args<float>& a, int Lp, bool vec4, int depth, const void* taps_dev, hipStream_t s, int scatter = 0);   // interleaved complex
int launch_inv3y4_f32(const Fused3Args<float>& a, int Lp, int depth, const void* taps_dev, hipStream_t s, int scatter = 0);   // a level dilated by 4 (EW = 4), vec4 rows

// level 1 of a denoising step in one launch (Den3: in[0] = x, in[1] = approximation band) and the approximation-only analysis
// that goes with it (tall 64 x 32 tile); float, real data, tap lengths 2 .. 8: ndwt_fused3_f32_den.hip
int launch_den3_f32(const Fused3Args<float>& a, int Lp, const void* taps_dev, hipStream_t s);
int launch_fwd3_low_f32(const Fused3Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s);
// one t-band of a 4-D analysis level with the t axis folded into the launch (a.tt = its t taps, batch items = frames, frame index fastest in the block order)
int launch_fwd3_tpre_f32(const Fused3Args<float>& a, int Lp, const void* taps_dev, hipStream_t s);

// float, tap lengths 14..18 (analysis) / 14..16 (synthesis): ndwt_fused3_f32_long.hip
int launch_long3_f32(bool inverse, const Fused3Args<float>& a, const FusedTapsD& t, bool vec4, int variant, const void* taps_dev, hipStream_t s);
int launch_long3_f64(bool inverse, const Fused3Args<double>& a, const FusedTapsD& t, bool vec4, const void* taps_dev, hipStream_t s);   // 14 / 16 taps

// fused 2-D kernels (register-only, one wave per tile)
int fused2_tile_width(bool inverse, int Lp, int ew);
int launch_fwd2_f32(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s);
int launch_inv2_f32(const Fused2Args<float>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s);
int launch_inv2p_f32(const Fused2Args<float>& a, int Lp, int depth, const void* taps_dev, hipStream_t s, int packed = 0);
int launch_inv2p_f64(const Fused2Args<double>& a, int Lp, const void* taps_dev, hipStream_t s);   // up to 8 taps
int launch_fwd2_f32_14to20(const Fused2Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s);   // float real, db7 .. db10
int launch_inv2_f32_14to20(const Fused2Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s);
int launch_fwd2_c64_10to16(const Fused2Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s);    // interleaved complex64, db5 .. db8
int launch_inv2_c64_10to16(const Fused2Args<float>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s);
int launch_long2_f64(bool inverse, const Fused2Args<double>& a, int Lp, bool vec4, const void* taps_dev, hipStream_t s);   // double real, db7 / db8   // Inv2P: 2 or 4 rows in flight per wave
// two or three analysis levels of an image in one launch (Fwd2C; float real data, rows of whole groups of 4, 2 .. 8 and 12 taps)
int fwd2c_tile_width(int Lp, int nlev);
int launch_fwd2c_f32(const Fused2CArgs<float>& a, int Lp, int nlev, const void* taps_dev, hipStream_t s);
// ... and the synthesis levels (Inv2C; 2 .. 8 taps; depth = rows of band loads in flight per level)
int inv2c_tile_width(int Lp, int nlev);
int launch_inv2c_f32(const Fused2CIArgs<float>& a, int Lp, int nlev, int depth, const void* taps_dev, hipStream_t s);
int launch_fwd2_f64(const Fused2Args<double>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s);
int launch_inv2_f64(const Fused2Args<double>& a, int Lp, bool vec4, int ew, const void* taps_dev, hipStream_t s);

// one non-contiguous axis with the window in registers (taps: kernel-form lo/hi of length L)
int launch_march_f32(bool syn, int L, const MarchArgs<float>& a, const double* lo, const double* hi, hipStream_t s);
int launch_march_f64(bool syn, int L, const MarchArgs<double>& a, const double* lo, const double* hi, hipStream_t s);

// the contiguous axis with lane shifts (ew = scalars per element: 1 real, 2 interleaved complex)
int launch_axisx_f32(bool syn, int L, int ew, const AxisXArgs<float>& a, bool vec4, const double* lo, const double* hi, hipStream_t s);
int launch_axisx_f64(bool syn, int L, int ew, const AxisXArgs<double>& a, bool vec4, const double* lo, const double* hi, hipStream_t s);

}  // namespace ndwt
