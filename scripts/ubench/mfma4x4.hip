// Layout and exactness probe for v_mfma_f32_4x4x1_16b_f32 on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma4x4 mfma4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(float* out) {
    const int l = threadIdx.x;
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 100.0f * (l + 1), c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

// 9-tap chain: acc[r] = fma chain over t of a_t[l] * b_t[l'] ; compare with VALU fmaf chain bit for bit
__global__ void chain_kernel(const float* __restrict__ cw /*[9][64]*/, const float* __restrict__ w /*[9][4]*/,
                             float* __restrict__ out_mfma, float* __restrict__ out_valu) {
    const int l = threadIdx.x;
    v4f c = {0, 0, 0, 0};
    float acc[4] = {0, 0, 0, 0};
    for (int t = 0; t < 9; ++t) {
        const float x = cw[t * 64 + l];
        c = __builtin_amdgcn_mfma_f32_4x4x1f32(w[t * 4 + (l & 3)], x, c, 0, 0, 0);
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_fmaf(x, w[t * 4 + k], acc[k]);
    }
    for (int r = 0; r < 4; ++r) {
        out_mfma[l * 4 + r] = c[r];
        out_valu[l * 4 + r] = acc[r];
    }
}

int main() {
    float *d, *dcw, *dw, *o1, *o2;
    hipMalloc(&d, 256 * 4);
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    int ok_layout = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int b = l / 4;
            const float want = (float)(4 * b + r + 1) * 100.0f * (l + 1);   // D[b][i=r][j=l%4] = A[b][r] * B[b][l%4]
            if (h[l * 4 + r] != want) ok_layout = 0;
        }
    printf("layout D[vgpr r][lane l] = A[lane 4*(l/4)+r] * B[lane l]: %s\n", ok_layout ? "YES" : "NO");
    if (!ok_layout) for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);

    std::mt19937 rng(1);
    std::vector<float> cw(9 * 64), w(36);
    int bad = 0, total = 0;
    hipMalloc(&dcw, cw.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&o1, 1024); hipMalloc(&o2, 1024);
    for (int trial = 0; trial < 200; ++trial) {
        std::uniform_real_distribution<float> u(-300.f, 300.f), uw(-1.f, 1.f);
        for (auto& x : cw) x = u(rng);
        for (auto& x : w) x = uw(rng);
        if (trial == 5) { cw[3] = NAN; cw[70] = INFINITY; cw[9 * 64 - 1] = -INFINITY; cw[10] = 1e-41f; w[2] = 1e-3f; }
        if (trial == 6) for (auto& x : cw) x *= 1e-38f;   // denormal products
        hipMemcpy(dcw, cw.data(), cw.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, dcw, dw, o1, o2);
        std::vector<float> a(256), b(256);
        hipMemcpy(a.data(), o1, 1024, hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), o2, 1024, hipMemcpyDeviceToHost);
        int tb = 0;
        for (int i = 0; i < 256; ++i) { ++total; if (memcmp(&a[i], &b[i], 4) != 0 && !(std::isnan(a[i]) && std::isnan(b[i]))) { ++bad; ++tb; } }
        if (tb && (trial == 5 || trial == 6 || bad < 10)) {
            for (int i = 0; i < 256 && tb; ++i) if (memcmp(&a[i], &b[i], 4) != 0 && !(std::isnan(a[i]) && std::isnan(b[i]))) { printf("trial %d idx %d mfma %.9g valu %.9g\n", trial, i, a[i], b[i]); --tb; if (tb > 3) tb = 3; }
        }
    }
    printf("chain of 9 mfma_4x4x1 vs fmaf chain: %d / %d values differ bitwise\n", bad, total);
    return 0;
}
