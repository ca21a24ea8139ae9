// Pattern ceilings for the fused 3-D kernels: what plain copy kernels reach on the same stream patterns, without halos or
// arithmetic.  512^3 floats per stream; "fan-in" = 8 band streams -> 1 (synthesis), "fan-out" = 1 -> 8 (analysis).
//   linear: every wave reads / writes 1 KiB of consecutive addresses (grid-stride): the best case for any kernel
//   tiled : one workgroup per CU owns a 64 x 32 tile of the 512 x 512 plane and marches 256 planes with three planes of loads
//           in flight -- the synthesis kernel's footprint (a CU only ever touches 256 B of every 2-KiB row)
// band pitch: packed (2^27 elements: every band at the same address modulo 2^29) or +64 elements (ndwt_band_pitch)
//   hipcc -O3 --offload-arch=gfx950 tools/micro/stream_pattern.hip -o tools/micro/stream_pattern && tools/micro/stream_pattern
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 512;
constexpr long long VOL = (long long)N * N * N;
typedef float v4 __attribute__((ext_vector_type(4)));

struct Bands { float* p[8]; };
template <bool NT> __device__ __forceinline__ v4 ld(const float* p) {
    if constexpr (NT) return __builtin_nontemporal_load((const v4*)p);
    else return *(const v4*)p;
}
template <bool NT> __device__ __forceinline__ void st(float* p, v4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, (v4*)p);
    else *(v4*)p = v;
}

__global__ void __launch_bounds__(256) copy_linear(const float* in, float* out, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) ((v4*)out)[i] = ((const v4*)in)[i];
}
__global__ void __launch_bounds__(256) read_linear(const float* in, float* out, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    v4 s = {0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) s += ((const v4*)in)[i];
    if (s.x == 12345.f) out[0] = s.y + s.z + s.w;
}
__global__ void __launch_bounds__(256) write_linear(float* out, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) ((v4*)out)[i] = v4{1, 2, 3, 4};
}

__global__ void __launch_bounds__(256) fan_in_linear(Bands in, float* out) {
    const long long n4 = VOL / 4, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        v4 s = ((const v4*)in.p[0])[i];
#pragma unroll
        for (int b = 1; b < 8; ++b) s += ((const v4*)in.p[b])[i];
        ((v4*)out)[i] = s;
    }
}

__global__ void __launch_bounds__(256) fan_out_linear(const float* in, Bands out) {
    const long long n4 = VOL / 4, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const v4 s = ((const v4*)in)[i];
#pragma unroll
        for (int b = 0; b < 8; ++b) ((v4*)out.p[b])[i] = s * (float)(b + 1);
    }
}

// tile (tx, ty) of 64 x TY, z chunk zc of `zchunk` planes; thread t: row t / 16, 16-byte chunk t % 16
template <int TY, int NT>
__device__ __forceinline__ long long tile_base(int& zbeg, int zchunk) {
    const int ntx = N / 64, nty = N / TY;
    int lb = blockIdx.x;
    const int nblocks = gridDim.x, q = nblocks / 8, k = lb % 8, i = lb / 8;   // contiguous runs of tiles per XCD (as decode_tile)
    lb = k * q + i;
    const int tx = lb % ntx, ty = (lb / ntx) % nty, zc = lb / (ntx * nty);
    zbeg = zc * zchunk;
    const int row = threadIdx.x / 16, ch = threadIdx.x % 16;
    return (long long)(ty * TY + row) * N + tx * 64 + ch * 4;
}

// 512 threads = the 512 16-byte chunks of a 64 x 32 tile; `lds_bytes` of dynamic LDS keep it at one workgroup per CU
template <bool NT_LD, bool NT_ST> __global__ void __launch_bounds__(512) fan_in_tiled(Bands in, float* out, int zchunk) {
    extern __shared__ float unused_lds[];
    int zbeg;
    const long long off = tile_base<32, 512>(zbeg, zchunk);
    const long long P = (long long)N * N;
    v4 r0[8], r1[8], r2[8];                               // three planes of loads in flight
    const int zend = zbeg + zchunk;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        r0[b] = ld<NT_LD>(in.p[b] + off + zbeg * P);
        r1[b] = ld<NT_LD>(in.p[b] + off + (zbeg + 1) * P);
    }
    for (int z = zbeg; z < zend; ++z) {
        if (z + 2 < zend) {
#pragma unroll
            for (int b = 0; b < 8; ++b) r2[b] = ld<NT_LD>(in.p[b] + off + (z + 2) * P);
        }
        v4 s = r0[0];
#pragma unroll
        for (int b = 1; b < 8; ++b) s += r0[b];
        st<NT_ST>(out + off + z * P, s);
#pragma unroll
        for (int b = 0; b < 8; ++b) { r0[b] = r1[b]; r1[b] = r2[b]; }
    }
    if (threadIdx.x == 100000) unused_lds[0] = 0.f;
}

template <bool NT_LD, bool NT_ST, int TY = 16> __global__ void __launch_bounds__(TY * 16) fan_out_tiled(const float* in, Bands out, int zchunk) {
    int zbeg;
    const long long off = tile_base<TY, TY * 16>(zbeg, zchunk);
    v4 cur = ld<NT_LD>(in + off + (long long)zbeg * N * N), nxt = cur;
    for (int z = zbeg; z < zbeg + zchunk; ++z) {
        if (z + 1 < zbeg + zchunk) nxt = ld<NT_LD>(in + off + (long long)(z + 1) * N * N);
#pragma unroll
        for (int b = 0; b < 8; ++b) st<NT_ST>(out.p[b] + off + (long long)z * N * N, cur * (float)(b + 1));
        cur = nxt;
    }
}


// the same 8 -> 1 tile pattern fed by LDS-DMA (global_load_lds_dwordx4: no VGPR destination): two planes of all 8 bands in an LDS
// ring (2 x 64 KiB), every wave loads its rows of the plane after next while the workgroup sums the current one from LDS.  The
// bounded experiment of VERDICT r2 item 6: does a loader path that keeps the compute waves out of the vector-memory pipe's return
// path beat the register path (fan_in_tiled) on this footprint?
__global__ void __launch_bounds__(512) fan_in_tiled_dma(Bands in, float* out, int zchunk) {
    extern __shared__ float ring[];                       // [2 planes][8 bands][512 chunks of 16 B]
    int zbeg;
    const long long off = tile_base<32, 512>(zbeg, zchunk);
    const long long P = (long long)N * N;
    const int zend = zbeg + zchunk;
    const int wave = threadIdx.x / 64;
    auto issue = [&](int z, int slot) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            // LDS destination: wave-uniform base + lane * 16: chunk index = thread index, band-major
            float* dst = ring + ((slot * 8 + b) * 512 + wave * 64) * 4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in.p[b] + off + z * P),
                                             (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
    };
    issue(zbeg, 0);
    if (zbeg + 1 < zend) issue(zbeg + 1, 1);
    for (int z = zbeg; z < zend; ++z) {
        const int slot = (z - zbeg) & 1;
        // the loads of plane z are the older group of the two in flight (this wave's own 8; every wave waits for its own rows only:
        // each thread reads back exactly the chunks its own wave loaded, so no barrier is needed)
        if (z + 1 < zend) __builtin_amdgcn_s_waitcnt(0x0F70 | 8 | ((8 >> 4) << 14));   // vmcnt(8)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                                        // vmcnt(0)
        v4 s = *(const v4*)(ring + ((slot * 8 + 0) * 512 + threadIdx.x) * 4);
#pragma unroll
        for (int b = 1; b < 8; ++b) s += *(const v4*)(ring + ((slot * 8 + b) * 512 + threadIdx.x) * 4);
        st<true>(out + off + z * P, s);
        __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the LDS reads are done before the slot is refilled
        if (z + 2 < zend) issue(z + 2, slot);
    }
}

// planes per workgroup so that `blocks` workgroups of 64 x TY tiles cover the volume exactly (anything else would run off the arrays)
static int checked_zchunk(int blocks, int ty) {
    const int tiles = (N / 64) * (N / ty);
    if (blocks % tiles != 0 || N % (blocks / tiles) != 0) { printf("bad launch geometry\n"); exit(1); }
    return N / (blocks / tiles);
}

template <class F> static float time_ms(F&& launch) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main() {
    CK(hipFuncSetAttribute((const void*)fan_in_tiled<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)fan_in_tiled<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)fan_in_tiled<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipFuncSetAttribute((const void*)fan_in_tiled_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    float *coef = nullptr, *vol = nullptr;
    const long long pad = 64;
    CK(hipMalloc(&coef, (size_t)(8 * (VOL + pad)) * 4));
    CK(hipMalloc(&vol, (size_t)VOL * 4));
    CK(hipMemset(coef, 0, (size_t)(8 * (VOL + pad)) * 4));
    CK(hipMemset(vol, 0, (size_t)VOL * 4));
    const double gb = 9.0 * VOL * 4 / 1e9;
    {   // calibration: one stream in, one out, 4 GiB each way (the 8 bands as one array)
        const long long n4 = 8 * VOL / 4 / 2;
        float t = time_ms([&] { hipLaunchKernelGGL(copy_linear, dim3(256 * 8), dim3(256), 0, 0, coef, coef + 4 * n4, n4); });
        printf("copy    1->1 linear  2 GiB -> 2 GiB   %.3f ms  %.0f GB/s\n", t, 2.0 * n4 * 16 / 1e9 / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL(read_linear, dim3(256 * 8), dim3(256), 0, 0, coef, vol, 2 * n4); });
        printf("read         linear  4 GiB            %.3f ms  %.0f GB/s\n", t, 2.0 * n4 * 16 / 1e9 / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL(write_linear, dim3(256 * 8), dim3(256), 0, 0, coef, 2 * n4); });
        printf("write        linear  4 GiB            %.3f ms  %.0f GB/s\n", t, 2.0 * n4 * 16 / 1e9 / t * 1e3);
    }
    for (int pitched = 0; pitched < 2; ++pitched) {
        Bands b;
        for (int k = 0; k < 8; ++k) b.p[k] = coef + k * (VOL + (pitched ? pad : 0));
        const char* lay = pitched ? "pitched (+256 B)" : "packed          ";
        float t;
        t = time_ms([&] { hipLaunchKernelGGL(fan_in_linear, dim3(256 * 8), dim3(256), 0, 0, b, vol); });
        printf("fan-in  8->1 linear  %s %.3f ms  %.0f GB/s\n", lay, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((fan_in_tiled<false, false>), dim3(256), dim3(512), 96 * 1024, 0, b, vol, checked_zchunk(256, 32)); });
        printf("fan-in  8->1 tiled   %s %.3f ms  %.0f GB/s\n", lay, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((fan_in_tiled<true, false>), dim3(256), dim3(512), 96 * 1024, 0, b, vol, checked_zchunk(256, 32)); });
        printf("fan-in  8->1 tiled   %s %.3f ms  %.0f GB/s  (nontemporal loads)\n", lay, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((fan_in_tiled<true, true>), dim3(256), dim3(512), 96 * 1024, 0, b, vol, checked_zchunk(256, 32)); });
        printf("fan-in  8->1 tiled   %s %.3f ms  %.0f GB/s  (nontemporal loads and stores)\n", lay, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL(fan_in_tiled_dma, dim3(256), dim3(512), 128 * 1024, 0, b, vol, checked_zchunk(256, 32)); });
        printf("fan-in  8->1 tiled   %s %.3f ms  %.0f GB/s  (LDS-DMA ring of 2 planes, nontemporal stores)\n", lay, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL(fan_out_linear, dim3(256 * 8), dim3(256), 0, 0, vol, b); });
        printf("fan-out 1->8 linear  %s %.3f ms  %.0f GB/s\n", lay, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((fan_out_tiled<false, false>), dim3(512), dim3(256), 0, 0, vol, b, checked_zchunk(512, 16)); });
        printf("fan-out 1->8 tiled   %s %.3f ms  %.0f GB/s  (512 workgroups)\n", lay, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((fan_out_tiled<false, true>), dim3(512), dim3(256), 0, 0, vol, b, checked_zchunk(512, 16)); });
        printf("fan-out 1->8 tiled   %s %.3f ms  %.0f GB/s  (512 workgroups, nontemporal stores)\n", lay, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((fan_out_tiled<false, true, 32>), dim3(256), dim3(512), 0, 0, vol, b, checked_zchunk(256, 32)); });
        printf("fan-out 1->8 tiled   %s %.3f ms  %.0f GB/s  (64x32 tiles, 256 workgroups of 512 threads, nontemporal stores)\n", lay, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((fan_out_tiled<false, true, 64>), dim3(256), dim3(1024), 0, 0, vol, b, checked_zchunk(256, 64)); });
        printf("fan-out 1->8 tiled   %s %.3f ms  %.0f GB/s  (64x64 tiles, 256 workgroups of 1024 threads, nontemporal stores)\n", lay, t, gb / t * 1e3);
        t = time_ms([&] { hipLaunchKernelGGL((fan_out_tiled<false, false>), dim3(2048), dim3(256), 0, 0, vol, b, checked_zchunk(2048, 16)); });
        printf("fan-out 1->8 tiled   %s %.3f ms  %.0f GB/s  (2048 workgroups)\n", lay, t, gb / t * 1e3);
    }
    return 0;
}
