// Where does global_load_lds_dwordx3 put a lane's 12 bytes?  (hipcc --offload-arch=gfx950 -O2 lds_dma_b96.hip -o lds_dma_b96)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* glb_ptr;
__global__ void k(const float* src, float* out, int nlanes) {
    __shared__ __attribute__((aligned(16))) float s[512];
    for (int i = threadIdx.x; i < 512; i += 64) s[i] = -1.0f;
    __syncthreads();
    if ((int)threadIdx.x < nlanes) __builtin_amdgcn_global_load_lds((glb_ptr)(src + threadIdx.x * 3), (lds_ptr)s, 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = s[i];
}
int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    float *d, *o;
    hipMalloc(&d, 1024); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    for (int nl : {64, 8}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, nl);
        std::vector<float> r(512);
        hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
        printf("lanes %d:", nl);
        for (int i = 0; i < 272; ++i) printf(" %g", r[i]);
        printf("\n");
    }
    return 0;
}
