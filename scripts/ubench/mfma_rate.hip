// Issue-rate probe: v_mfma_f32_4x4x1_16b_f32 alone (independent / dependent chain) and mixed with VGPR-only v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x

template <int MODE>
__global__ void k(float* out, int iters) {
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    v4f c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float f0 = 1, f1 = 2, f2 = 3, f3 = 4, f4 = 5, f5 = 6, f6 = 7, f7 = 8;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // 8 MFMAs on 4 independent accumulators
            REP4(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %4, %5, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %4, %5, %1\n"
                              "v_mfma_f32_4x4x1_16b_f32 %2, %4, %5, %2\n v_mfma_f32_4x4x1_16b_f32 %3, %4, %5, %3\n"
                              "v_mfma_f32_4x4x1_16b_f32 %0, %4, %5, %0\n v_mfma_f32_4x4x1_16b_f32 %1, %4, %5, %1\n"
                              "v_mfma_f32_4x4x1_16b_f32 %2, %4, %5, %2\n v_mfma_f32_4x4x1_16b_f32 %3, %4, %5, %3\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));)
        } else if (MODE == 1) {  // 8 dependent MFMAs (one accumulator)
            REP4(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n"
                              "v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n"
                              "v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n"
                              "v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0\n"
                              : "+v"(c0) : "v"(a), "v"(b));)
        } else if (MODE == 2) {  // dependent MFMA chain interleaved 1:1 with VGPR-only fmas (8 + 8)
            REP4(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %9, %10, %0\n v_fma_f32 %1, %1, %10, %1\n v_mfma_f32_4x4x1_16b_f32 %0, %9, %10, %0\n v_fma_f32 %2, %2, %10, %2\n"
                              "v_mfma_f32_4x4x1_16b_f32 %0, %9, %10, %0\n v_fma_f32 %3, %3, %10, %3\n v_mfma_f32_4x4x1_16b_f32 %0, %9, %10, %0\n v_fma_f32 %4, %4, %10, %4\n"
                              "v_mfma_f32_4x4x1_16b_f32 %0, %9, %10, %0\n v_fma_f32 %5, %5, %10, %5\n v_mfma_f32_4x4x1_16b_f32 %0, %9, %10, %0\n v_fma_f32 %6, %6, %10, %6\n"
                              "v_mfma_f32_4x4x1_16b_f32 %0, %9, %10, %0\n v_fma_f32 %7, %7, %10, %7\n v_mfma_f32_4x4x1_16b_f32 %0, %9, %10, %0\n v_fma_f32 %8, %8, %10, %8\n"
                              : "+v"(c0), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a), "v"(b));)
        } else if (MODE == 3) {  // 1 dependent MFMA per 4 VGPR-only fmas (2 + 8 per block) ~ the ratio 9 : 40 of a row
            REP4(asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %9, %10, %0\n v_fma_f32 %1, %1, %10, %1\n v_fma_f32 %2, %2, %10, %2\n v_fma_f32 %3, %3, %10, %3\n v_fma_f32 %4, %4, %10, %4\n"
                              "v_mfma_f32_4x4x1_16b_f32 %0, %9, %10, %0\n v_fma_f32 %5, %5, %10, %5\n v_fma_f32 %6, %6, %10, %6\n v_fma_f32 %7, %7, %10, %7\n v_fma_f32 %8, %8, %10, %8\n"
                              : "+v"(c0), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a), "v"(b));)
        } else if (MODE == 4) {  // only the 8 VGPR fmas of mode 3 (reference)
            REP4(asm volatile("v_fma_f32 %1, %1, %10, %1\n v_fma_f32 %2, %2, %10, %2\n v_fma_f32 %3, %3, %10, %3\n v_fma_f32 %4, %4, %10, %4\n"
                              "v_fma_f32 %5, %5, %10, %5\n v_fma_f32 %6, %6, %10, %6\n v_fma_f32 %7, %7, %10, %7\n v_fma_f32 %8, %8, %10, %8\n"
                              : "+v"(c0), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a), "v"(b));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int MODE>
void run(int wps, float* out, const char* name, int instr_per_block) {
    const int threads = 256 * wps, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double blocks = (double)iters * 4;
    printf("%-28s waves/SIMD %d  %.3f ms  %.1f cyc per block of %d instr per SIMD (@2.4GHz)\n", name, wps, ms,
           ms * 1e-3 * 2.4e9 / (blocks * wps), instr_per_block);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4 * 4);
    for (int w : {1, 2, 4}) {
        run<0>(w, out, "8 mfma independent", 8);
        run<1>(w, out, "8 mfma dependent", 8);
        run<2>(w, out, "8 mfma dep + 8 fma", 16);
        run<3>(w, out, "2 mfma dep + 8 fma", 10);
        run<4>(w, out, "8 fma (vgpr)", 8);
    }
    return 0;
}
