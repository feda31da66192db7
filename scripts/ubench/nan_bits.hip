// What NaN does gfx950 produce / propagate?  (decides whether relu can be an integer max: a NaN with the sign bit set would
// be flushed to 0 by v_max_i32(bits, 0), which (x < 0) ? 0 : x does not do.)
//   hipcc --offload-arch=gfx950 -O2 -o gpurun_exp/nan_bits scripts/ubench/nan_bits.hip && gpurun_exp/nan_bits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out) {
    const float inf = in[0], zero = in[1], pnan = in[2], nnan = in[3], neg = in[4], one = in[5];
    int i = 0;
    auto put = [&](float v) { out[i++] = __float_as_uint(v); };
    put(zero * inf);                       // 0: invalid -> default NaN
    put(inf - inf);                        // 1
    put(__builtin_fmaf(zero, inf, one));   // 2
    put(pnan * neg);                       // 3: +NaN times a negative weight
    put(nnan * one);                       // 4: -NaN propagated
    put(__builtin_fmaf(nnan, neg, one));   // 5
    put(__builtin_fmaf(pnan, neg, one));   // 6
    put(nnan + one);                       // 7
    f2 a = {pnan, nnan}, b = {neg, neg}, c = {one, one}, r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    put(r.x); put(r.y);                    // 8, 9
    f2 z = {zero, zero}, q = {inf, -inf};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(z), "v"(q), "v"(c));
    put(r.x); put(r.y);                    // 10, 11: 0 * inf, 0 * -inf
    put(__builtin_amdgcn_exp2f(pnan));     // 12
    put(__builtin_amdgcn_logf(-one));      // 13: log of a negative -> NaN
    put(one * __builtin_amdgcn_exp2f(-0.1f * __builtin_amdgcn_logf(zero)));   // 14: regulator factor at m = 0 (inf)
    put(zero * (one * __builtin_amdgcn_exp2f(-0.1f * __builtin_amdgcn_logf(zero))));   // 15: 0 * inf
    put(sqrtf(-one));                      // 16
    put(zero / zero);                      // 17
}
int main() {
    float h[6] = {INFINITY, 0.0f, 0.0f, 0.0f, -0.25f, 1.0f};
    unsigned p = 0x7fc00000u, n = 0xffc12345u;
    memcpy(&h[2], &p, 4); memcpy(&h[3], &n, 4);
    float* d; unsigned* o; unsigned r[32] = {0};
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d, o);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    const char* names[] = {"0*inf", "inf-inf", "fma(0,inf,1)", "+nan*neg", "-nan*1", "fma(-nan,neg,1)", "fma(+nan,neg,1)", "-nan+1",
                           "pk_fma(+nan,neg,1)", "pk_fma(-nan,neg,1)", "pk_fma(0,inf,1)", "pk_fma(0,-inf,1)", "exp2(nan)", "log2(-1)",
                           "exp2(-.1*log2(0))", "0*exp2(-.1*log2(0))", "sqrt(-1)", "0/0"};
    for (int i = 0; i < 18; ++i) printf("%-22s %08x\n", names[i], r[i]);
    return 0;
}
