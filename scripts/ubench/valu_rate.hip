// VALU issue-rate microbenchmark for gfx950: v_fma_f32 vs v_pk_fma_f32 vs v_mov_b32_dpp at 1..8 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void rate_kernel(float* out, int iters, float w) {
    __shared__ float4 s_w[64];
    if (MODE == 11 || MODE == 12) {
        if (threadIdx.x < 64) s_w[threadIdx.x] = make_float4(w, w + 1e-7f, w - 1e-7f, w);
        __syncthreads();
    }
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long mask = 0x5555555555555555ull + (unsigned long long)iters;
    asm volatile("" : "+s"(mask));
    float b0 = 1, b1 = 2, b2 = 3, b3 = 4, b4 = 5, b5 = 6, b6 = 7, b7 = 8;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 independent v_fma_f32 (SGPR weight)
            REP8(asm volatile("v_fmac_f32 %0, %8, %0\n v_fmac_f32 %1, %8, %1\n v_fmac_f32 %2, %8, %2\n v_fmac_f32 %3, %8, %3\n"
                              "v_fmac_f32 %4, %8, %4\n v_fmac_f32 %5, %8, %5\n v_fmac_f32 %6, %8, %6\n v_fmac_f32 %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(w));)
        } else if (MODE == 1) {  // 4 independent v_pk_fma_f32 = 8 fmas
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                              "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                              : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));)
        } else if (MODE == 2) {  // 8 v_mov_b32_dpp wave_shr
            REP8(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %6 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 3) {  // 8 v_cndmask (vcc)
            REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                              "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
        } else if (MODE == 5) {  // v_cndmask with an SGPR-pair mask (e64)
            REP8(asm volatile("v_cndmask_b32 %0, %0, %1, %8\n v_cndmask_b32 %1, %1, %2, %8\n v_cndmask_b32 %2, %2, %3, %8\n v_cndmask_b32 %3, %3, %4, %8\n"
                              "v_cndmask_b32 %4, %4, %5, %8\n v_cndmask_b32 %5, %5, %6, %8\n v_cndmask_b32 %6, %6, %7, %8\n v_cndmask_b32 %7, %7, %0, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(mask));)
        } else if (MODE == 6) {  // v_fma_f32 VGPR operands only
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                              "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
        } else if (MODE == 7) {  // v_med3_f32
            REP8(asm volatile("v_med3_f32 %0, %0, 0, %8\n v_med3_f32 %1, %1, 0, %8\n v_med3_f32 %2, %2, 0, %8\n v_med3_f32 %3, %3, 0, %8\n"
                              "v_med3_f32 %4, %4, 0, %8\n v_med3_f32 %5, %5, 0, %8\n v_med3_f32 %6, %6, 0, %8\n v_med3_f32 %7, %7, 0, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(w));)
        } else if (MODE == 8) {  // v_cmp + v_cndmask pairs (4 pairs = 8 instructions), explicit nop for the vcc hazard
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n s_nop 1\n v_cndmask_b32 %2, %2, %3, vcc\n"
                              "v_cmp_lt_f32 vcc, %4, %5\n s_nop 1\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f32 vcc, %6, %7\n s_nop 1\n v_cndmask_b32 %6, %6, %7, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");)
        } else if (MODE == 9) {  // v_max_f32 + v_min_f32
            REP8(asm volatile("v_max_f32 %0, 0, %0\n v_min_f32 %0, %8, %0\n v_max_f32 %1, 0, %1\n v_min_f32 %1, %8, %1\n"
                              "v_max_f32 %2, 0, %2\n v_min_f32 %2, %8, %2\n v_max_f32 %3, 0, %3\n v_min_f32 %3, %8, %3\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(w));)
        } else if (MODE == 10) {  // v_lshl_add_u64
            REP8(asm volatile("v_lshl_add_u64 %0, %0, 2, %4\n v_lshl_add_u64 %1, %1, 2, %4\n v_lshl_add_u64 %2, %2, 2, %4\n v_lshl_add_u64 %3, %3, 2, %4\n"
                              "v_lshl_add_u64 %0, %0, 2, %4\n v_lshl_add_u64 %1, %1, 2, %4\n v_lshl_add_u64 %2, %2, 2, %4\n v_lshl_add_u64 %3, %3, 2, %4\n"
                              : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6) : "v"(*(double*)&b0));)
        } else if (MODE == 11) {  // 64 fmas whose weights arrive by broadcast ds_read_b128 (4 weights per read)
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float4 wa = s_w[(i + 2 * g) & 63], wb = s_w[(i + 2 * g + 1) & 63];
                a0 = __builtin_fmaf(a0, wa.x, a0); a1 = __builtin_fmaf(a1, wa.y, a1);
                a2 = __builtin_fmaf(a2, wa.z, a2); a3 = __builtin_fmaf(a3, wa.w, a3);
                a4 = __builtin_fmaf(a4, wb.x, a4); a5 = __builtin_fmaf(a5, wb.y, a5);
                a6 = __builtin_fmaf(a6, wb.z, a6); a7 = __builtin_fmaf(a7, wb.w, a7);
            }
        } else if (MODE == 12) {  // same with one weight per ds_read_b32
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const float* sw = (const float*)s_w;
                const int b = (i + 8 * g) & 127;
                a0 = __builtin_fmaf(a0, sw[b], a0); a1 = __builtin_fmaf(a1, sw[b + 1], a1);
                a2 = __builtin_fmaf(a2, sw[b + 2], a2); a3 = __builtin_fmaf(a3, sw[b + 3], a3);
                a4 = __builtin_fmaf(a4, sw[b + 4], a4); a5 = __builtin_fmaf(a5, sw[b + 5], a5);
                a6 = __builtin_fmaf(a6, sw[b + 6], a6); a7 = __builtin_fmaf(a7, sw[b + 7], a7);
            }
        } else if (MODE == 4) {  // fmac with dpp source (VGPR weight)
            REP8(asm volatile("v_fmac_f32_dpp %0, %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %1, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_fmac_f32_dpp %2, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %3, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_fmac_f32_dpp %4, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %5, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_fmac_f32_dpp %6, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fmac_f32_dpp %7, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
}

template <int MODE>
double run(int waves_per_simd, float* out, const char* name, double per_instr_work) {
    // 256 CUs x 4 SIMDs; one block of 64*4*waves_per_simd threads per CU
    const int threads = 64 * 4 * waves_per_simd;
    const int blocks = 256;
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 64;  // 8 x 8 instructions per iteration
    const double cyc = ms * 1e-3 * 2.4e9;             // at the nominal 2.4 GHz
    const double cyc_per_instr_per_simd = cyc / (instr_per_wave * waves_per_simd);
    printf("%-18s waves/SIMD %d  %.3f ms  %.2f cyc/instr/SIMD (@2.4GHz)  x%.0f work\n", name, waves_per_simd, ms,
           cyc_per_instr_per_simd, per_instr_work);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4 * 8);
    for (int w : {1, 2, 4}) {
        run<0>(w, out, "v_fmac_f32", 1);
        run<1>(w, out, "v_pk_fma_f32", 2);
        run<2>(w, out, "v_mov_b32_dpp", 1);
        run<3>(w, out, "v_cndmask_b32", 1);
        run<4>(w, out, "v_fmac_f32_dpp", 1);
        run<5>(w, out, "v_cndmask e64 sgpr", 1);
        run<6>(w, out, "v_fma_f32 vgpr", 1);
        run<7>(w, out, "v_med3_f32", 1);
        run<8>(w, out, "cmp+nop+cndmask", 1);
        run<9>(w, out, "v_max+v_min", 1);
        run<10>(w, out, "v_lshl_add_u64", 1);
        run<11>(w, out, "fma + lds b128 w", 1);
        run<12>(w, out, "fma + lds b32 w", 1);
    }
    return 0;
}
