// Zoom pyramid: un-prefiltered quintic B-spline resampling of a frame crop (scipy.ndimage.zoom
// order=5, prefilter=False semantics; reference call site util/zoom/from_image.py:55-59),
// separable, with the per-axis tap tables built on the host in float64.
#pragma once

#include "silent_common.h"

namespace silent {

constexpr int kPyrTW = 64;    // output columns per tile (one wave per row segment)
constexpr int kPyrMaxTH = 16; // output rows per tile (shrinks with the vertical step)
constexpr int kPyrMaxRows = 48;  // LDS rows of horizontally filtered source lines per tile

struct PyrLevelDev {
    int src_y0, src_x0, src_h, src_w;
    int zoom_h, zoom_w, out_h, out_w;
    int tile_h;    // output rows per tile of this level
    int xtab_off;  // entry offset (in output columns) of this level in the x tables
    int ytab_off;  // entry offset (in output rows) in the y tables
};

struct PyrTab {
    int n_levels;
    int tiles_per_frame;
    int H, W, C;
    long long frame_px_out;  // pixels of one output pyramid
    PyrLevelDev lv[kMaxLevels];
    int tiles_x[kMaxLevels];
    int tile_start[kMaxLevels + 1];
    long long px_off[kMaxLevels];
    // device tables (one allocation owned by the plan)
    const int* xidx;   // [cols][6] source column, mirrored, relative to the crop
    const float* xw;   // [cols][6]
    const int* ybase;  // [rows] floor(coordinate), unmirrored, relative to the crop
    const float* yw;   // [rows][6]
};

__device__ __forceinline__ int mirror_index(int i, int n) {
    // scipy 'mirror' extension (d c b | a b c d | c b a)
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    if (i < 0) i = -i;
    i %= period;
    return i >= n ? period - i : i;
}

// One tile = 64 output columns x tile_h output rows of one level of one frame.
// Phase 1: every source line the tile's rows touch is filtered horizontally into LDS
//          (slot s <-> unmirrored source row base(oy0) - 2 + s; lane = output column).
// Phase 2: 6-tap vertical combination from LDS; coalesced NHWC store.
template <int C>
__global__ __launch_bounds__(256) void pyramid_kernel(const float* __restrict__ frames,
                                                      float* __restrict__ pyr, const PyrTab tab) {
    __shared__ float s_h[kPyrMaxRows * kPyrTW * C];

    const unsigned bid = blockIdx.x;
    const int frame = (int)(bid / (unsigned)tab.tiles_per_frame);
    int rem = (int)(bid - (unsigned)frame * (unsigned)tab.tiles_per_frame);
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < tab.n_levels && rem >= tab.tile_start[i]) l = i;
    rem -= tab.tile_start[l];
    const PyrLevelDev lv = tab.lv[l];
    const int ty = rem / tab.tiles_x[l];
    const int tx = rem - ty * tab.tiles_x[l];

    const int tid = threadIdx.x;
    const int col = tid & 63, wave = tid >> 6;
    const int ox = tx * kPyrTW + col;
    const int oy0 = ty * lv.tile_h;
    // rows of this tile that the resampler actually produces
    const int zy1 = min(oy0 + lv.tile_h, min(lv.zoom_h, lv.out_h));
    const float* __restrict__ src = frames + (long long)frame * tab.H * tab.W * C;
    float* __restrict__ dst = pyr + ((long long)frame * tab.frame_px_out + tab.px_off[l]) * C;

    int nrows = 0, rstart = 0;
    if (oy0 < zy1) {
        rstart = tab.ybase[lv.ytab_off + oy0] - 2;
        nrows = tab.ybase[lv.ytab_off + zy1 - 1] + 3 - rstart + 1;
        nrows = min(nrows, kPyrMaxRows);  // host guarantees <=; never index past LDS
    }
    const bool col_live = ox < lv.zoom_w && ox < lv.out_w;

    if (nrows > 0) {
        int xi[6];
        float xw[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            xi[i] = col_live ? (tab.xidx[(long long)(lv.xtab_off + ox) * 6 + i] + lv.src_x0) * C : 0;
            xw[i] = col_live ? tab.xw[(long long)(lv.xtab_off + ox) * 6 + i] : 0.0f;
        }
        for (int s = wave; s < nrows; s += 4) {
            const int r = mirror_index(rstart + s, lv.src_h) + lv.src_y0;
            const float* __restrict__ row = src + (long long)r * tab.W * C;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float acc = 0.0f;
#pragma unroll
                for (int i = 0; i < 6; ++i) acc = __builtin_fmaf(xw[i], row[xi[i] + c], acc);
                s_h[(s * kPyrTW + col) * C + c] = acc;
            }
        }
    }
    __syncthreads();

    if (ox >= lv.out_w) return;
    for (int oy = oy0 + wave; oy < oy0 + lv.tile_h && oy < lv.out_h; oy += 4) {
        float v[C];
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = 0.0f;
        if (oy < zy1 && col_live) {
            const int s0 = tab.ybase[lv.ytab_off + oy] - 2 - rstart;
            const float* __restrict__ wy = tab.yw + (long long)(lv.ytab_off + oy) * 6;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float w = wy[j];
                const int s = min(s0 + j, kPyrMaxRows - 1);
#pragma unroll
                for (int c = 0; c < C; ++c) v[c] = __builtin_fmaf(w, s_h[(s * kPyrTW + col) * C + c], v[c]);
            }
        }
        float* __restrict__ po = dst + ((long long)oy * lv.out_w + ox) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) po[c] = v[c];
    }
}

}  // namespace silent
