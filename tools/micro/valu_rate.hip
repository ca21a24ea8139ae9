// VALU issue-rate microbenchmark (gfx950): cycles per wave64 instruction on one SIMD with 1, 2, 4 waves per SIMD, for
// v_fmac_f32 (VGPR and SGPR operands), v_pk_fma_f32 and v_mov_b32_dpp wave_shr:1.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE> __global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float s0, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = out[threadIdx.x];
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {b + 1, b + 2};
    v2 sp = {s0, s0 * 0.5f};
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 64 independent-ish FMAs (8 chains x 8)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fmac_f32 %0, %8, %0\n v_fmac_f32 %1, %8, %1\n v_fmac_f32 %2, %8, %2\n v_fmac_f32 %3, %8, %3\n"
                             "v_fmac_f32 %4, %8, %4\n v_fmac_f32 %5, %8, %5\n v_fmac_f32 %6, %8, %6\n v_fmac_f32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            }
        } else if (MODE == 1) {   // SGPR multiplier
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fmac_f32 %0, %8, %0\n v_fmac_f32 %1, %8, %1\n v_fmac_f32 %2, %8, %2\n v_fmac_f32 %3, %8, %3\n"
                             "v_fmac_f32 %4, %8, %4\n v_fmac_f32 %5, %8, %5\n v_fmac_f32 %6, %8, %6\n v_fmac_f32 %7, %8, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s0));
            }
        } else if (MODE == 2 || MODE == 6 || MODE == 7 || MODE == 8 || MODE == 9) {   // packed: 32 v_pk_fma = 64 FMAs
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_fma_f32 %0, %4, %0, %0\n v_pk_fma_f32 %1, %4, %1, %1\n v_pk_fma_f32 %2, %4, %2, %2\n v_pk_fma_f32 %3, %4, %3, %3\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
            }
        } else if (MODE == 3) {   // 64 DPP wave shifts
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (MODE == 4) {   // 64 DPP row shifts (within 16 lanes)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else if (MODE == 6) {   // packed, multiplier = SGPR pair, multiplicand broadcast from the low half of a VGPR pair
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1]\n"
                             "v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[0,1,1]\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "s"(sp));
            }
        } else if (MODE == 7) {   // packed, both multiplicands VGPR pairs, one broadcast
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_fma_f32 %0, %4, %5, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1]\n"
                             "v_pk_fma_f32 %2, %4, %5, %2 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %4, %5, %3 op_sel_hi:[0,1,1]\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            }
        } else if (MODE == 8) {   // ONE dependent chain of packed FMAs
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n"
                             : "+v"(p0) : "v"(pb), "v"(pc));
            }
        } else if (MODE == 9) {   // TWO interleaved dependent chains
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n"
                             : "+v"(p0), "+v"(p1) : "v"(pb), "v"(pc));
            }
        } else if (MODE == 10) {   // ONE dependent chain of scalar FMAs
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n"
                             "v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n v_fmac_f32 %0, %1, %0\n"
                             : "+v"(a0) : "v"(b));
            }
        } else {   // fmac with a DPP-shifted operand folded in (v_fmac_f32_dpp)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                asm volatile("v_fmac_f32_dpp %0, %8, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_fmac_f32_dpp %1, %8, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_fmac_f32_dpp %2, %8, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_fmac_f32_dpp %3, %8, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_fmac_f32_dpp %4, %8, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_fmac_f32_dpp %5, %8, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_fmac_f32_dpp %6, %8, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_fmac_f32_dpp %7, %8, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (MODE == 2 || MODE == 6 || MODE == 7 || MODE == 8 || MODE == 9) { a0 = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE> void run(const char* name, int per_iter) {
    float* out;
    long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMemset(out, 0, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 16 * 8);
    const int iters = 2000;
    for (int nt : {256, 512, 1024}) {   // 1, 2, 4 waves per SIMD (one workgroup per CU)
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(nt), 0, 0, out, cyc, 1.0001f, iters);
        hipDeviceSynchronize();
        std::vector<long long> h(nt / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        long long mx = 0;
        for (auto v : h) mx = v > mx ? v : mx;
        // per SIMD: (nt/256) waves, each iters*per_iter instructions, in mx cycles
        printf("%-28s waves/SIMD %d: %.2f cycles per wave-instruction (per SIMD), wave elapsed %.2f per instr\n", name, nt / 256,
               (double)mx / ((double)iters * per_iter * (nt / 256)), (double)mx / ((double)iters * per_iter));
    }
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("v_fmac_f32 vgpr", 64);
    run<1>("v_fmac_f32 sgpr", 64);
    run<2>("v_pk_fma_f32 (2 FMA/lane)", 32);
    run<3>("v_mov_b32_dpp wave_shr:1", 64);
    run<4>("v_mov_b32_dpp row_shr:1", 64);
    run<5>("v_fmac_f32_dpp wave_shr:1", 64);
    run<6>("v_pk_fma sgpr-pair x bcast", 32);
    run<7>("v_pk_fma vgpr-pair x bcast", 32);
    run<8>("v_pk_fma 1 dependent chain", 32);
    run<9>("v_pk_fma 2 dependent chains", 32);
    run<10>("v_fmac 1 dependent chain", 64);
    return 0;
}
