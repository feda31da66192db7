// Packed-f32 VALU on gfx950: (1) which half of an SGPR-pair source v_pk_fma_f32 reads under op_sel / op_sel_hi, (2) issue rate
// of v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 / v_pk_mov_b32 with SGPR and VGPR sources, beside the scalar forms.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void semantics_kernel(float* out, unsigned long long wpair) {
    // wpair = {lo = 3.0f, hi = 5.0f} in an SGPR pair; x = (1, 2); acc = (10, 20)
    f2 x = {1.0f, 2.0f}, r;
    asm volatile("" : "+s"(wpair));
    f2 a = {10.0f, 20.0f};
    r = a; asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r) : "v"(x), "s"(wpair));                                   // default
    out[0] = r.x; out[1] = r.y;
    r = a; asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(r) : "v"(x), "s"(wpair));                 // lo for both
    out[2] = r.x; out[3] = r.y;
    r = a; asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(r) : "v"(x), "s"(wpair));  // hi for both
    out[4] = r.x; out[5] = r.y;
    r = a; asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(r) : "v"(x), "s"(wpair));  // swapped
    out[6] = r.x; out[7] = r.y;
    f2 wv; wv.x = 3.0f; wv.y = 5.0f; asm volatile("" : "+v"(wv));
    r = a; asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(r) : "v"(x), "v"(wv));     // VGPR pair, hi for both
    out[8] = r.x; out[9] = r.y;
    r = a; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "s"(wpair));
    out[10] = r.x; out[11] = r.y;
    r = a; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "s"(wpair));
    out[12] = r.x; out[13] = r.y;
}

#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void rate_kernel(float* out, int iters, unsigned long long wpair, float w) {
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    f2 b = {1.0001f, 0.9999f};
    float c0 = threadIdx.x, c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3, c4 = c0 + 4, c5 = c0 + 5, c6 = c0 + 6, c7 = c0 + 7;
    f2 x0 = {0.5f, 0.25f}, x1 = {0.75f, 0.5f}, x2 = {0.3f, 0.2f}, x3 = {0.1f, 0.9f};
    asm volatile("" : "+s"(wpair), "+s"(w), "+v"(b), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {  // v_pk_fma_f32, SGPR pair, lo broadcast
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %4, %1 op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 %2, %2, %4, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %4, %3 op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 %0, %0, %4, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %1, %4, %1 op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 %2, %2, %4, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %3, %4, %3 op_sel_hi:[1,0,1]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(wpair));)
        } else if (MODE == 1) {  // v_pk_fma_f32, VGPR pair
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                              "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if (MODE == 2) {  // v_pk_mul_f32 SGPR
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[1,0]\n"
                              "v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[1,0]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(wpair));)
        } else if (MODE == 3) {  // v_pk_add_f32 VGPR
            REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));)
        } else if (MODE == 4) {  // v_pk_mov_b32
            REP8(asm volatile("v_pk_mov_b32 %0, %1, %1\n v_pk_mov_b32 %1, %2, %2\n v_pk_mov_b32 %2, %3, %3\n v_pk_mov_b32 %3, %0, %0\n"
                              "v_pk_mov_b32 %0, %1, %1\n v_pk_mov_b32 %1, %2, %2\n v_pk_mov_b32 %2, %3, %3\n v_pk_mov_b32 %3, %0, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (MODE == 5) {  // v_fmac_f32 SGPR (reference)
            REP8(asm volatile("v_fmac_f32 %0, %8, %0\n v_fmac_f32 %1, %8, %1\n v_fmac_f32 %2, %8, %2\n v_fmac_f32 %3, %8, %3\n"
                              "v_fmac_f32 %4, %8, %4\n v_fmac_f32 %5, %8, %5\n v_fmac_f32 %6, %8, %6\n v_fmac_f32 %7, %8, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "s"(w));)
        } else if (MODE == 6) {  // v_max_f32 with inline constant 0
            REP8(asm volatile("v_max_f32 %0, 0, %0\n v_max_f32 %1, 0, %1\n v_max_f32 %2, 0, %2\n v_max_f32 %3, 0, %3\n"
                              "v_max_f32 %4, 0, %4\n v_max_f32 %5, 0, %5\n v_max_f32 %6, 0, %6\n v_max_f32 %7, 0, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));)
        } else if (MODE == 7) {  // v_mov_b32 vgpr
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));)
        } else if (MODE == 8) {  // mixed: pk_fma SGPR + independent ds_write/ds_read traffic is not modelled; pk_fma SGPR hi broadcast
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %1, %1, %4, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                              "v_pk_fma_f32 %2, %2, %4, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %3, %4, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                              "v_pk_fma_f32 %0, %0, %4, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %1, %1, %4, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                              "v_pk_fma_f32 %2, %2, %4, %2 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n v_pk_fma_f32 %3, %3, %4, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(wpair));)
        } else if (MODE == 9) {  // v_fma_f32 3-address with SGPR
            REP8(asm volatile("v_fma_f32 %0, %1, %8, %0\n v_fma_f32 %1, %2, %8, %1\n v_fma_f32 %2, %3, %8, %2\n v_fma_f32 %3, %4, %8, %3\n"
                              "v_fma_f32 %4, %5, %8, %4\n v_fma_f32 %5, %6, %8, %5\n v_fma_f32 %6, %7, %8, %6\n v_fma_f32 %7, %0, %8, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "s"(w));)
        } else if (MODE == 10) {  // v_pk_fma_f32 sgpr, x in its own VGPR pair (acc += x * w)
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %8, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %5, %8, %1 op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 %2, %6, %8, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %7, %8, %3 op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 %0, %5, %8, %0 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %6, %8, %1 op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 %2, %7, %8, %2 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %4, %8, %3 op_sel_hi:[1,0,1]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(wpair));)
        } else if (MODE == 11) {  // v_fmac_f32 sgpr, x in its own VGPR
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %10\n v_fmac_f32 %2, %8, %11\n v_fmac_f32 %3, %8, %12\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %10\n v_fmac_f32 %6, %8, %11\n v_fmac_f32 %7, %8, %12\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "s"(w), "v"(x0.x), "v"(x1.x), "v"(x2.x), "v"(x3.x));)
        } else if (MODE == 12) {  // 3-address v_pk_fma_f32 sgpr: dst pair differs from both sources (role rotation)
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %8, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %5, %8, %2 op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 %2, %6, %8, %3 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %7, %8, %0 op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 %0, %5, %8, %1 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %6, %8, %2 op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 %2, %7, %8, %3 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %4, %8, %0 op_sel_hi:[1,0,1]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "s"(wpair));)
        } else if (MODE == 13) {  // v_pk_fma_f32 all VGPR pairs distinct
            REP8(asm volatile("v_pk_fma_f32 %0, %4, %8, %0\n v_pk_fma_f32 %1, %5, %8, %1\n"
                              "v_pk_fma_f32 %2, %6, %8, %2\n v_pk_fma_f32 %3, %7, %8, %3\n"
                              "v_pk_fma_f32 %0, %5, %8, %0\n v_pk_fma_f32 %1, %6, %8, %1\n"
                              "v_pk_fma_f32 %2, %7, %8, %2\n v_pk_fma_f32 %3, %4, %8, %3\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(b));)
        } else if (MODE == 14) {  // v_fma_f32 all VGPR distinct
            REP8(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %10\n v_fmac_f32 %2, %8, %11\n v_fmac_f32 %3, %8, %12\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %10\n v_fmac_f32 %6, %8, %11\n v_fmac_f32 %7, %8, %12\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(b.x), "v"(x0.x), "v"(x1.x), "v"(x2.x), "v"(x3.x));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0.x + a0.y + a1.x + a1.y + a2.x + a2.y + a3.x + a3.y + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

template <int MODE>
void run(int waves_per_simd, float* out, const char* name, int work) {
    const int threads = 64 * 4 * waves_per_simd, blocks = 256, iters = 2000;
    union { float f[2]; unsigned long long u; } wp;
    wp.f[0] = 1.0001f; wp.f[1] = 0.9999f;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, wp.u, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, wp.u, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 64 * waves_per_simd);
    printf("%-28s waves/SIMD %d  %.3f ms  %.2f cyc/instr/SIMD (@2.4GHz)  x%d work -> %.2f cyc per f32 op\n", name, waves_per_simd, ms, cyc, work,
           cyc / work);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4 * 8);
    union { float f[2]; unsigned long long u; } wp;
    wp.f[0] = 3.0f; wp.f[1] = 5.0f;
    hipLaunchKernelGGL(semantics_kernel, dim3(1), dim3(64), 0, 0, out, wp.u);
    float h[14];
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    printf("x=(1,2) acc=(10,20) sgpr pair (lo=3, hi=5)\n");
    printf("default                         -> (%g, %g)   [13,30 = lo,hi]\n", h[0], h[1]);
    printf("op_sel_hi:[1,0,1]               -> (%g, %g)   [13,26 = lo,lo]\n", h[2], h[3]);
    printf("op_sel:[0,1,0] op_sel_hi:[1,1,1]-> (%g, %g)   [15,30 = hi,hi]\n", h[4], h[5]);
    printf("op_sel:[0,1,0] op_sel_hi:[1,0,1]-> (%g, %g)   [15,26 = hi,lo]\n", h[6], h[7]);
    printf("VGPR pair hi,hi                 -> (%g, %g)   [15,30]\n", h[8], h[9]);
    printf("pk_mul sgpr lo,lo               -> (%g, %g)   [3,6]\n", h[10], h[11]);
    printf("pk_mul sgpr hi,hi               -> (%g, %g)   [5,10]\n", h[12], h[13]);
    for (int w : {1, 2, 3, 4}) {
        run<0>(w, out, "v_pk_fma_f32 sgpr lo,lo", 2);
        run<8>(w, out, "v_pk_fma_f32 sgpr hi,hi", 2);
        run<1>(w, out, "v_pk_fma_f32 vgpr", 2);
        run<2>(w, out, "v_pk_mul_f32 sgpr", 2);
        run<3>(w, out, "v_pk_add_f32 vgpr", 2);
        run<4>(w, out, "v_pk_mov_b32", 2);
        run<5>(w, out, "v_fmac_f32 sgpr", 1);
        run<9>(w, out, "v_fma_f32 3-addr sgpr", 1);
        run<6>(w, out, "v_max_f32 0", 1);
        run<7>(w, out, "v_mov_b32 vgpr", 1);
        run<10>(w, out, "v_pk_fma sgpr, x distinct", 2);
        run<12>(w, out, "v_pk_fma sgpr, 3-address", 2);
        run<13>(w, out, "v_pk_fma vgpr, all distinct", 2);
        run<11>(w, out, "v_fmac sgpr, x distinct", 1);
        run<14>(w, out, "v_fmac vgpr, all distinct", 1);
    }
    return 0;
}
