// Experiment helpers for profiles/r06/placement.md (not part of the product library):
//  * vmm_alloc: a device buffer assembled from `chunk`-byte physical allocations (hipMemCreate) mapped into ONE virtual range in a
//    chosen order -- sequential, shuffled, reversed -- so that the physical contiguity of a map can be varied on purpose;
//  * wp_run: a pure write kernel with the tile / run geometry of gray_stream_kernel's K-orientation map (a block of 4 waves writes
//    tile_rows runs of 4 * run_bytes, one image row after the other), to see which write patterns feel the placement.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

struct VmmBuf {
    void* ptr;
    size_t bytes, chunk;
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
static std::vector<VmmBuf*> g_bufs;

#define TRY(x)                                                                                 \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "vmm_probe: %s -> %s\n", #x, hipGetErrorString(e_));               \
            return (int)e_;                                                                    \
        }                                                                                      \
    } while (0)

extern "C" int vmm_granularity(int device, size_t* min_g, size_t* rec_g) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    TRY(hipMemGetAllocationGranularity(min_g, &prop, hipMemAllocationGranularityMinimum));
    TRY(hipMemGetAllocationGranularity(rec_g, &prop, hipMemAllocationGranularityRecommended));
    return 0;
}

// order: 0 chunks mapped in creation order, 1 shuffled (seed), 2 reversed, 3 even chunks first then odd ones
extern "C" int vmm_alloc(int device, size_t bytes, size_t chunk, int order, unsigned seed, void** out) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    const size_t n = (bytes + chunk - 1) / chunk;
    VmmBuf* b = new VmmBuf{nullptr, n * chunk, chunk, {}};
    TRY(hipMemAddressReserve(&b->ptr, b->bytes, chunk < (size_t)(2 << 20) ? (2 << 20) : chunk > ((size_t)1 << 30) ? ((size_t)1 << 30) : chunk, nullptr, 0));
    b->handles.resize(n);
    for (size_t i = 0; i < n; ++i) TRY(hipMemCreate(&b->handles[i], chunk, &prop, 0));
    std::vector<size_t> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = i;
    if (order == 1) {
        std::mt19937 rng(seed);
        std::shuffle(perm.begin(), perm.end(), rng);
    } else if (order == 2) {
        std::reverse(perm.begin(), perm.end());
    } else if (order == 3) {
        size_t k = 0;
        for (size_t i = 0; i < n; i += 2) perm[k++] = i;
        for (size_t i = 1; i < n; i += 2) perm[k++] = i;
    }
    for (size_t i = 0; i < n; ++i) TRY(hipMemMap((char*)b->ptr + i * chunk, chunk, 0, b->handles[perm[i]], 0));
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    TRY(hipMemSetAccess(b->ptr, b->bytes, &acc, 1));
    g_bufs.push_back(b);
    *out = b->ptr;
    return 0;
}

// n maps, each contiguous in its own virtual range, their PHYSICAL chunks created in proportional interleaved order: the chunk that
// backs fraction t of map a is created right next to (in time, hence -- a sequential allocator -- in memory) the chunks that back
// fraction t of the other maps.  So what a wave writes at the same pixel index of several maps lies in one physical neighbourhood.
extern "C" int vmm_striped(int device, int n, const size_t* bytes, size_t chunk, void** out) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<VmmBuf*> bufs(n);
    std::vector<size_t> chunks(n), done(n, 0);
    for (int a = 0; a < n; ++a) {
        chunks[a] = (bytes[a] + chunk - 1) / chunk;
        bufs[a] = new VmmBuf{nullptr, chunks[a] * chunk, chunk, {}};
        TRY(hipMemAddressReserve(&bufs[a]->ptr, bufs[a]->bytes, (size_t)2 << 20, nullptr, 0));
    }
    for (;;) {
        // the map that is furthest behind (smallest done / chunks) gets the next physical chunk
        int pick = -1;
        for (int a = 0; a < n; ++a)
            if (done[a] < chunks[a] && (pick < 0 || (double)done[a] / chunks[a] < (double)done[pick] / chunks[pick])) pick = a;
        if (pick < 0) break;
        hipMemGenericAllocationHandle_t h;
        TRY(hipMemCreate(&h, chunk, &prop, 0));
        bufs[pick]->handles.push_back(h);
        TRY(hipMemMap((char*)bufs[pick]->ptr + done[pick] * chunk, chunk, 0, h, 0));
        ++done[pick];
    }
    for (int a = 0; a < n; ++a) {
        TRY(hipMemSetAccess(bufs[a]->ptr, bufs[a]->bytes, &acc, 1));
        g_bufs.push_back(bufs[a]);
        out[a] = bufs[a]->ptr;
    }
    return 0;
}

// Physical memory only (no mapping): what a spacer needs.  Returns the handle through *out; vmm_phys_release frees it.
extern "C" int vmm_phys_create(int device, size_t bytes, unsigned long long* out) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemGenericAllocationHandle_t h;
    TRY(hipMemCreate(&h, bytes, &prop, 0));
    *out = (unsigned long long)(uintptr_t)h;
    return 0;
}
extern "C" int vmm_phys_release(unsigned long long handle) {
    TRY(hipMemRelease((hipMemGenericAllocationHandle_t)(uintptr_t)handle));
    return 0;
}

extern "C" int vmm_free(void* ptr) {
    for (size_t k = 0; k < g_bufs.size(); ++k)
        if (g_bufs[k]->ptr == ptr) {
            VmmBuf* b = g_bufs[k];
            TRY(hipDeviceSynchronize());
            TRY(hipMemUnmap(b->ptr, b->bytes));
            for (auto h : b->handles) TRY(hipMemRelease(h));
            TRY(hipMemAddressFree(b->ptr, b->bytes));
            g_bufs.erase(g_bufs.begin() + k);
            delete b;
            return 0;
        }
    return -1;
}

// One block = 4 waves side by side; wave w writes `run_bytes` (a multiple of 16) of each of the tile's rows, row after row.
typedef float nf4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void wp_kernel(char* __restrict__ dst, long long row_bytes, int rows, int run_bytes, int tile_rows,
                                                 int tiles_x, long long image_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned bid = blockIdx.x;
    const int tiles_y = (rows + tile_rows - 1) / tile_rows;
    const int img = bid / (tiles_x * tiles_y);
    const int rem = bid - img * tiles_x * tiles_y;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const long long x0 = ((long long)tx * 4 + wave) * run_bytes;
    if (x0 >= row_bytes) return;
    const int n = (int)min((long long)run_bytes, row_bytes - x0);
    char* base = dst + img * image_bytes + x0;
    const nf4 v = {1.0f * lane, 2.0f, 3.0f * wave, 4.0f};
    for (int r = 0; r < tile_rows; ++r) {
        const int y = ty * tile_rows + r;
        if (y >= rows) break;
        char* p = base + (long long)y * row_bytes;
        for (int o = lane * 16; o < n; o += 1024) {
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<nf4*>(p + o));
            else *reinterpret_cast<nf4*>(p + o) = v;
        }
    }
}

// Returns the mean milliseconds of `reps` launches (after 2 warm-up launches), or a negative HIP error.
extern "C" float wp_run(void* dst, long long row_bytes, int rows, int images, int run_bytes, int tile_rows, int nt, int reps) {
    const int tiles_x = (int)((row_bytes + 4LL * run_bytes - 1) / (4LL * run_bytes));
    const int tiles_y = (rows + tile_rows - 1) / tile_rows;
    const unsigned grid = (unsigned)(tiles_x * tiles_y * images);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < reps + 2; ++i) {
        if (i == 2) hipEventRecord(a, nullptr);
        if (nt) wp_kernel<true><<<grid, 256>>>((char*)dst, row_bytes, rows, run_bytes, tile_rows, tiles_x, row_bytes * rows);
        else wp_kernel<false><<<grid, 256>>>((char*)dst, row_bytes, rows, run_bytes, tile_rows, tiles_x, row_bytes * rows);
    }
    hipEventRecord(b, nullptr);
    hipError_t e = hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a);
    hipEventDestroy(b);
    if (e != hipSuccess) return -(float)e;
    return ms / reps;
}


// wp2: the write pattern of gray_stream_kernel's unit level with its other streams, switchable: per tile row a wave writes a run of
// 56 pixels x 4K bytes of the K-orientation map (always), and -- flags -- 1: first reads 24 rows of the frame (64 lanes x 4 B each,
// all requested up front), 2: writes the 56-float run of the CS map (224 B = 1.75 lines), 4: the same for the pyramid's level 0,
// 8: those 1-channel stores temporal instead of non-temporal, 16: the K-map stores temporal.
template <int K, int COLS = 56>
__global__ __launch_bounds__(256) void wp2_kernel(float* __restrict__ end, float* __restrict__ cs, float* __restrict__ pyr,
                                                  const float* __restrict__ frames, int W, int H, long long map_px_per_frame, int flags) {
    constexpr int R = 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_x = (W + 4 * COLS - 1) / (4 * COLS), tiles_y = (H + R - 1) / R;
    const unsigned bid = blockIdx.x;
    const int img = bid / (tiles_x * tiles_y);
    const int rem = bid - img * tiles_x * tiles_y;
    const int ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int xw0 = tx * 4 * COLS + wave * COLS;
    if (xw0 >= W && !(flags & (32 | 64))) return;
    const int y0 = ty * R;
    float acc = 0.0f;
    if (flags & 1) {
        const float* src = frames + (long long)img * W * H;
        const int sx = min(max(xw0 + lane - 4, 0), W - 1);
        float in[R + 8];
#pragma unroll
        for (int i = 0; i < R + 8; ++i) in[i] = src[(long long)min(max(y0 - 4 + i, 0), H - 1) * W + sx];
#pragma unroll
        for (int i = 0; i < R + 8; ++i) acc += in[i];
    }
    const bool out_lane = lane < COLS && xw0 + lane < W;
    const long long base = (long long)img * map_px_per_frame + xw0;
    for (int r = 0; r < R; ++r) {
        const int y = y0 + r;
        if (y >= H) break;
        const long long px = base + (long long)y * W;
        if (flags & 32) {
            // block-wide runs: wave w writes the WHOLE tile row (224 px = 896 B = 7 aligned lines) of rows r = 4 j + w, one float4 per lane
            if ((r & 3) == wave) {
                const long long bpx = (long long)img * map_px_per_frame + (long long)tx * 4 * COLS + (long long)y * W;
                const int n4 = (min(4 * COLS, W - tx * 4 * COLS) + 3) / 4;
                const nf4 q = {acc, acc, acc, acc};
                // (all four waves of the block run this loop; the rows of the three other waves are written by them)
                if (lane < n4) {
                    if (flags & 4) {
                        if (flags & 8) reinterpret_cast<nf4*>(pyr + bpx)[lane] = q;
                        else __builtin_nontemporal_store(q, reinterpret_cast<nf4*>(pyr + bpx) + lane);
                    }
                    if (flags & 2) {
                        if (flags & 8) reinterpret_cast<nf4*>(cs + bpx)[lane] = q;
                        else __builtin_nontemporal_store(q, reinterpret_cast<nf4*>(cs + bpx) + lane);
                    }
                }
            }
        } else {
        if (flags & 4) {
            if (out_lane) {
                if (flags & 8) pyr[px + lane] = acc;
                else __builtin_nontemporal_store(acc, pyr + px + lane);
            }
        }
        if (flags & 2) {
            if (out_lane) {
                if (flags & 8) cs[px + lane] = acc + 1.0f;
                else __builtin_nontemporal_store(acc + 1.0f, cs + px + lane);
            }
        }
        }
        if (flags & 64) {
            // the K map in block-wide runs too: wave w writes the whole tile row (224 px x 4K bytes) of rows r = 4 j + w
            if ((r & 3) == wave) {
                const long long bpx = (long long)img * map_px_per_frame + (long long)tx * 4 * COLS + (long long)y * W;
                nf4* o4 = reinterpret_cast<nf4*>(end + bpx * K);
                const int pieces = min(4 * COLS, W - tx * 4 * COLS) * (K / 4);
                const nf4 v = {acc, 2.0f, 3.0f, (float)r};
                for (int o = lane; o < pieces; o += 64) {
                    if (flags & 16) o4[o] = v;
                    else __builtin_nontemporal_store(v, o4 + o);
                }
            }
            continue;
        }
        // the K floats of a pixel: K / 4 float4 pieces; lane l writes pieces l, l + 64, ... of the wave's run (contiguous 1 KiB per instruction)
        if (xw0 >= W) continue;
        nf4* out4 = reinterpret_cast<nf4*>(end + px * K);
        const int pieces = min(COLS, W - xw0) * (K / 4);
        const nf4 v = {acc, 2.0f, 3.0f, (float)r};
        for (int o = lane; o < pieces; o += 64) {
            if (flags & 16) out4[o] = v;
            else __builtin_nontemporal_store(v, out4 + o);
        }
    }
}

extern "C" float wp2_run(void* end, void* cs, void* pyr, const void* frames, int K, int W, int H, int images, long long map_px_per_frame,
                         int flags, int reps) {
    // flags & 128: 64 output columns per wave (every run starts on a 64-byte boundary and is whole 64-byte pieces) instead of 56
    const int cols = (flags & 128) ? 64 : (flags & 256) ? 48 : 56;   // flags & 256: 48 columns per wave (192-byte runs, 64-byte aligned)
    const int tiles_x = (W + 4 * cols - 1) / (4 * cols), tiles_y = (H + 15) / 16;
    const unsigned grid = (unsigned)(tiles_x * tiles_y * images);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < reps + 2; ++i) {
        if (i == 2) (void)hipEventRecord(a, nullptr);
        if (flags & 256) {
            if (K == 8) wp2_kernel<8, 48><<<grid, 256>>>((float*)end, (float*)cs, (float*)pyr, (const float*)frames, W, H, map_px_per_frame, flags);
            else wp2_kernel<4, 48><<<grid, 256>>>((float*)end, (float*)cs, (float*)pyr, (const float*)frames, W, H, map_px_per_frame, flags);
        } else if (flags & 128) {
            if (K == 8) wp2_kernel<8, 64><<<grid, 256>>>((float*)end, (float*)cs, (float*)pyr, (const float*)frames, W, H, map_px_per_frame, flags);
            else wp2_kernel<4, 64><<<grid, 256>>>((float*)end, (float*)cs, (float*)pyr, (const float*)frames, W, H, map_px_per_frame, flags);
        } else {
            if (K == 8) wp2_kernel<8><<<grid, 256>>>((float*)end, (float*)cs, (float*)pyr, (const float*)frames, W, H, map_px_per_frame, flags);
            else wp2_kernel<4><<<grid, 256>>>((float*)end, (float*)cs, (float*)pyr, (const float*)frames, W, H, map_px_per_frame, flags);
        }
    }
    (void)hipEventRecord(b, nullptr);
    hipError_t e = hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (e != hipSuccess) return -(float)e;
    return ms / reps;
}
