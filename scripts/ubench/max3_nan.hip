// gfx950's v_maximum3_f32 / v_minimum3_f32 (IEEE-754-2019 maximum / minimum: NaN-propagating) as the relu / clip of the RGB chain,
// and the operand swaps of packed f32 instructions on VGPR pairs (op_sel / op_sel_hi) that the mov-free neighbour scheme uses.
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_exp/max3_nan scripts/ubench/max3_nan.hip && gpurun_exp/max3_nan
// Questions: does maximum3(v, 0, 0) return the NaN it was given (sign, payload)?  What does it make of -0?  Of -inf / +inf?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, int n, float hi, unsigned* out) {
    int o = 0;
    for (int i = 0; i < n; ++i) {
        const float v = in[i];
        float r, q, ref;
        asm volatile("v_maximum3_f32 %0, %1, 0, 0" : "=v"(r) : "v"(v));
        asm volatile("v_minimum3_f32 %0, %1, %2, %2" : "=v"(q) : "v"(r), "v"(hi));
        ref = (v < 0.0f) ? 0.0f : v;
        ref = (ref > hi) ? hi : ref;
        out[o++] = __float_as_uint(v);
        out[o++] = __float_as_uint(r);
        out[o++] = __float_as_uint(q);
        out[o++] = __float_as_uint(ref);
    }
    // packed swaps: C = (1, 2), D = (10, 20)
    const f2 C = {in[n], in[n + 1]}, D = {in[n + 2], in[n + 3]};
    f2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(D), "v"(C));   // (D.hi + C.hi, D.lo + C.lo)
    out[o++] = __float_as_uint(r.x); out[o++] = __float_as_uint(r.y);
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(D), "v"(C));   // (D.lo + C.hi, D.hi + C.lo)
    out[o++] = __float_as_uint(r.x); out[o++] = __float_as_uint(r.y);
    const f2 acc = {100.0f, 200.0f};
    unsigned long long w;   // SGPR pair (3, 5)
    {
        const unsigned lo = __builtin_amdgcn_readfirstlane(__float_as_uint(in[n + 4])), hi2 = __builtin_amdgcn_readfirstlane(__float_as_uint(in[n + 5]));
        w = ((unsigned long long)hi2 << 32) | lo;
    }
    // src0 swapped (lo half takes D.hi, hi half takes D.lo), weights straight (lo half w.lo, hi half w.hi)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(D), "s"(w), "v"(acc));   // (20*3+100, 10*5+200)
    out[o++] = __float_as_uint(r.x); out[o++] = __float_as_uint(r.y);
    // src0 swapped, weights swapped (lo half w.hi, hi half w.lo)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(C), "s"(w), "v"(acc));   // (2*5+100, 1*3+200)
    out[o++] = __float_as_uint(r.x); out[o++] = __float_as_uint(r.y);
}
int main() {
    unsigned bits[] = {0x7fc00000u, 0xffc00000u, 0xffc12345u, 0x7f800001u /* sNaN */, 0x80000000u /* -0 */, 0x00000000u, 0xbf800000u /* -1 */,
                       0x3f800000u, 0x43960000u /* 300 */, 0x7f800000u /* +inf */, 0xff800000u /* -inf */, 0x80000001u /* -denormal */, 0x00000001u};
    const int n = sizeof(bits) / 4;
    float h[n + 6];
    memcpy(h, bits, sizeof(bits));
    h[n] = 1.0f; h[n + 1] = 2.0f; h[n + 2] = 10.0f; h[n + 3] = 20.0f; h[n + 4] = 3.0f; h[n + 5] = 5.0f;
    float* d; unsigned* o; unsigned r[4 * 16 + 8] = {0};
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, 255.0f, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("%-12s %-12s %-12s %-12s\n", "v", "maximum3(v,0,0)", "minimum3(.,255)", "(v<0?0:v)>255?255:.");
    for (int i = 0; i < n; ++i) printf("%08x     %08x     %08x     %08x%s\n", r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3], r[4 * i + 2] == r[4 * i + 3] ? "" : "   <- differs");
    float f[8];
    memcpy(f, r + 4 * n, sizeof(f));
    printf("pk_add D.hi+C.hi, D.lo+C.lo = (%g, %g)   want (22, 11)\n", f[0], f[1]);
    printf("pk_add D.lo+C.hi, D.hi+C.lo = (%g, %g)   want (12, 21)\n", f[2], f[3]);
    printf("pk_fma Dswapped * w          = (%g, %g)   want (160, 250)\n", f[4], f[5]);
    printf("pk_fma Cswapped * wswapped   = (%g, %g)   want (110, 203)\n", f[6], f[7]);
    return 0;
}
