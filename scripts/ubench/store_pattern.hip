// Store-pattern microbenchmark for gfx950: how fast does HBM take the write stream of gray_stream_kernel
// (per wave and row: 224 B + 224 B + 896 B into three arrays, rows 7680 B apart) compared with a contiguous fill of the
// same bytes?   build: hipcc --offload-arch=gfx950 -O3 -o store_pattern store_pattern.hip ; run: ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int W = 1920, H = 1080, TH = 16, COLS = 56;

// MODE 0: tile pattern (block = 4 waves side by side, 16 rows); MODE 1: same bytes, every wave writes contiguous memory;
// MODE 2: tile pattern with ROWS_PER_STEP rows' worth written per "row step" as one contiguous run (upper bound of a wider tile)
__device__ __forceinline__ unsigned xcd_swizzle(unsigned bid, unsigned n) {
    const unsigned per = n / 8, rem = n % 8, x = bid % 8, k = bid / 8;
    return x * per + (x < rem ? x : rem) + k;
}

template <int MODE, bool NT, int WAVES = 4, bool XCD = false, bool BAR = false>
__global__ __launch_bounds__(WAVES * 64) void pattern(float* __restrict__ a, float* __restrict__ b, float4* __restrict__ e,
                                                      int tiles_x, int tiles_per_frame) {
    constexpr int TW = WAVES * COLS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned bid = XCD ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
    const int frame = bid / tiles_per_frame, rem = bid - frame * tiles_per_frame;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const long long base = (long long)frame * W * H;
    const float v = (float)bid;
    if (MODE == 0) {
        const int x = tx * TW + wave * COLS + lane;
        if (lane >= COLS || x >= W) return;
        for (int r = 0; r < TH; ++r) {
            const int y = ty * TH + r;
            if (y >= H) break;
            if (BAR) __builtin_amdgcn_s_barrier();      // the block's waves store a row at the same time
            const long long p = base + (long long)y * W + x;
            if (NT) {
                __builtin_nontemporal_store(v, a + p);
                __builtin_nontemporal_store(v, b + p);
            } else {
                a[p] = v;
                b[p] = v;
            }
            e[p] = make_float4(v, v, v, v);
        }
    } else {
        // the same number of bytes per block (16 rows x 224 px x 24 B), laid out contiguously per block
        const long long blk = (long long)bid * TH * TW;
        for (int r = 0; r < TH; ++r) {
            const long long p = blk + r * TW + wave * COLS + lane;
            if (lane >= COLS) continue;
            if (NT) {
                __builtin_nontemporal_store(v, a + p);
                __builtin_nontemporal_store(v, b + p);
            } else {
                a[p] = v;
                b[p] = v;
            }
            e[p] = make_float4(v, v, v, v);
        }
    }
}

template <int MODE, bool NT, int WAVES = 4, bool XCD = false, bool BAR = false>
void run(const char* name, float* a, float* b, float4* e, int frames) {
    constexpr int TW = WAVES * COLS;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int blocks = tiles_x * tiles_y * frames;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int it = 0; it < 12; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((pattern<MODE, NT, WAVES, XCD, BAR>), dim3(blocks), dim3(WAVES * 64), 0, 0, a, b, e, tiles_x, tiles_x * tiles_y);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = MODE == 0 ? (double)frames * W * H * 24.0 : (double)blocks * TH * TW * 24.0;
    printf("%-44s %.3f ms  %.0f GB/s\n", name, ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6);
}

// PX pixels per lane: a wave covers 56 * PX columns (a lane's pixels are adjacent: float2 / float4 stores for the 1-channel maps)
template <int PX, bool NT, bool XCD>
__global__ __launch_bounds__(256) void pattern_wide(float* __restrict__ a, float* __restrict__ b, float4* __restrict__ e,
                                                    int tiles_x, int tiles_per_frame) {
    constexpr int TW = 4 * COLS * PX;
    typedef float vec __attribute__((ext_vector_type(PX)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned bid = XCD ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
    const int frame = bid / tiles_per_frame, rem = bid - frame * tiles_per_frame;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const long long base = (long long)frame * W * H;
    const float v = (float)bid;
    const int x = tx * TW + (wave * COLS + lane) * PX;
    if (lane >= COLS || x + PX > W) return;
    vec vv;
    for (int i = 0; i < PX; ++i) vv[i] = v;
    for (int r = 0; r < TH; ++r) {
        const int y = ty * TH + r;
        if (y >= H) break;
        const long long p = base + (long long)y * W + x;
        if (NT) {
            __builtin_nontemporal_store(vv, (vec*)(a + p));
            __builtin_nontemporal_store(vv, (vec*)(b + p));
        } else {
            *(vec*)(a + p) = vv;
            *(vec*)(b + p) = vv;
        }
        for (int i = 0; i < PX; ++i) e[p + i] = make_float4(v, v, v, v);
    }
}

template <int PX, bool NT, bool XCD>
void run_wide(const char* name, float* a, float* b, float4* e, int frames) {
    constexpr int TW = 4 * COLS * PX;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int blocks = tiles_x * tiles_y * frames;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int it = 0; it < 12; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((pattern_wide<PX, NT, XCD>), dim3(blocks), dim3(256), 0, 0, a, b, e, tiles_x, tiles_x * tiles_y);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    // columns beyond the last whole group of PX are not written: count what is
    const double cols = (double)(W / PX) * PX;
    const double bytes = (double)frames * cols * H * 24.0;
    printf("%-44s %.3f ms  %.0f GB/s\n", name, ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6);
}

// NS adjacent 56-column strips per wave, written row by row with separate 4-byte / 16-byte stores issued back to back
template <int NS, bool NT>
__global__ __launch_bounds__(256) void pattern_strips(float* __restrict__ a, float* __restrict__ b, float4* __restrict__ e,
                                                      int tiles_x, int tiles_per_frame) {
    constexpr int TW = 4 * COLS * NS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned bid = blockIdx.x;
    const int frame = bid / tiles_per_frame, rem = bid - frame * tiles_per_frame;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const long long base = (long long)frame * W * H;
    const float v = (float)bid;
    const int x0 = tx * TW + wave * COLS * NS + lane;
    if (lane >= COLS) return;
    for (int r = 0; r < TH; ++r) {
        const int y = ty * TH + r;
        if (y >= H) break;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int x = x0 + s_ * COLS;
            if (x >= W) continue;
            const long long p = base + (long long)y * W + x;
            if (NT) __builtin_nontemporal_store(v, a + p);
            else a[p] = v;
        }
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int x = x0 + s_ * COLS;
            if (x >= W) continue;
            const long long p = base + (long long)y * W + x;
            if (NT) __builtin_nontemporal_store(v, b + p);
            else b[p] = v;
        }
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int x = x0 + s_ * COLS;
            if (x >= W) continue;
            e[base + (long long)y * W + x] = make_float4(v, v, v, v);
        }
    }
}

template <int NS, bool NT>
void run_strips(const char* name, float* a, float* b, float4* e, int frames) {
    constexpr int TW = 4 * COLS * NS;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int blocks = tiles_x * tiles_y * frames;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> ts;
    for (int it = 0; it < 12; ++it) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((pattern_strips<NS, NT>), dim3(blocks), dim3(256), 0, 0, a, b, e, tiles_x, tiles_x * tiles_y);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)frames * W * H * 24.0;
    printf("%-44s %.3f ms  %.0f GB/s\n", name, ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6);
}

int main() {
    const int frames = 64;
    // the contiguous variant writes whole 224 x 16 tiles for every block, edge tiles included: size for that
    // the contiguous variant writes whole tiles for every block, edge tiles included: size for the widest tile (16 waves)
    const size_t tiles = (size_t)((W + 224 - 1) / 224) * ((H + TH - 1) / TH) * frames;
    const size_t tiles16 = (size_t)((W + 896 - 1) / 896) * ((H + TH - 1) / TH) * frames;
    const size_t px = std::max({(size_t)frames * W * H, tiles * TH * 224, tiles16 * TH * 896}) + (1 << 22);
    float *a, *b;
    float4* e;
    if (hipMalloc(&a, px * 4) != hipSuccess || hipMalloc(&b, px * 4) != hipSuccess || hipMalloc(&e, px * 16) != hipSuccess) return 1;
    run<0, false>("tile pattern (3 arrays, rows 7680 B apart)", a, b, e, frames);
    run<0, true>("tile pattern, nt on the 4-byte stores", a, b, e, frames);
    run<1, false>("same bytes, contiguous per block", a, b, e, frames);
    run<1, true>("contiguous per block, nt on the 4-byte stores", a, b, e, frames);
    run<0, false, 4, true>("tile pattern, XCD-contiguous block order", a, b, e, frames);
    run<0, true, 4, true>("tile pattern, XCD order, nt", a, b, e, frames);
    run<0, false, 16>("tile pattern, 16 waves side by side (896 px)", a, b, e, frames);
    run<0, false, 16, true>("16 waves side by side, XCD order", a, b, e, frames);
    run<0, false, 8>("tile pattern, 8 waves side by side (448 px)", a, b, e, frames);
    run_wide<2, false, false>("2 px per lane (wave = 112 px, block 448)", a, b, e, frames);
    run_wide<2, true, false>("2 px per lane, nt", a, b, e, frames);
    run_wide<4, false, false>("4 px per lane (wave = 224 px, block 896)", a, b, e, frames);
    run_wide<4, true, false>("4 px per lane, nt", a, b, e, frames);
    run_wide<4, false, true>("4 px per lane, XCD order", a, b, e, frames);
    run<0, false, 4, false, true>("tile pattern, barrier before every row", a, b, e, frames);
    run<0, true, 4, false, true>("tile pattern, barrier per row, nt", a, b, e, frames);
    run<0, true, 16, false, true>("16 waves side by side, barrier per row, nt", a, b, e, frames);
    run_strips<2, false>("2 adjacent strips per wave, scalar stores", a, b, e, frames);
    run_strips<2, true>("2 adjacent strips per wave, nt", a, b, e, frames);
    run_strips<3, true>("3 adjacent strips per wave, nt", a, b, e, frames);
    run_strips<4, true>("4 adjacent strips per wave, nt", a, b, e, frames);
    return 0;
}
