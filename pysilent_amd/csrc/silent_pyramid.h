// Zoom pyramid: un-prefiltered quintic B-spline resampling of a frame crop (scipy.ndimage.zoom
// order=5, prefilter=False semantics; reference call site util/zoom/from_image.py:55-59),
// separable, with the per-axis tap tables built on the host in float64.
//
// One launch covers every level of every frame; a tile takes one of three block-uniform paths:
//   UNIT    zoom factor exactly 1 (level 0, 75 % of all pixels): the resampler degenerates to the fixed
//           5-tap smoother [1,26,66,26,1]/120 per axis -> LDS-tiled separable stencil, float4 staging.
//   DENSE   step <= ~4 source pixels per output pixel: the contiguous source footprint of the tile is
//           staged into LDS once (coalesced), filtered vertically, then horizontally.
//   SPARSE  larger steps (tiny levels): 36 gathered taps per output straight from L2.
#pragma once

#include "silent_common.h"

namespace silent {

enum { kPyrUnit = 0, kPyrDense = 1, kPyrSparse = 2 };

constexpr int kUnitTW = 64, kUnitTH = 32;         // UNIT output tile
constexpr int kUnitSW = kUnitTW + 8;              // staged columns x0-4 .. x0+67 (16-byte aligned start)
constexpr int kUnitSH = kUnitTH + 4;              // staged rows    y0-2 .. y0+33
constexpr int kDenseMaxTW = 64, kDenseMaxTH = 16; // DENSE output tile (shrinks with the step)
constexpr int kDenseMaxSH = 40;                   // staged source rows per tile
__host__ __device__ constexpr int dense_max_sw(int C) { return C == 1 ? 136 : 72; }  // staged source columns
constexpr int kSparseTW = 64, kSparseTH = 4;

template <int C>
__host__ __device__ constexpr int pyr_lds_floats() {
    constexpr int unit = (kUnitSH * kUnitSW + kUnitSH * kUnitTW) * C;
    constexpr int dense = (kDenseMaxSH + kDenseMaxTH) * dense_max_sw(C) * C;
    return unit > dense ? unit : dense;
}

struct PyrLevelDev {
    int src_y0, src_x0, src_h, src_w;
    int zoom_h, zoom_w, out_h, out_w;
    int kind, tile_w, tile_h;
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
    // device tables (one allocation owned by the plan); "base" = floor(coordinate), unmirrored,
    // relative to the crop; idx = the 6 mirrored tap positions relative to the crop
    const int* xbase;
    const int* xidx;   // [cols][6]
    const float* xw;   // [cols][6]
    const int* ybase;
    const int* yidx;   // [rows][6]
    const float* yw;   // [rows][6]
};

__device__ __forceinline__ int mirror_index(int i, int n) {
    // scipy 'mirror' extension (d c b | a b c d | c b a)
    if ((unsigned)i < (unsigned)n) return i;
    if (n == 1) return 0;
    const int period = 2 * (n - 1);
    if (i < 0) i = -i;
    i %= period;
    return i >= n ? period - i : i;
}

// ------------------------------------------------------------------------------------------ UNIT
template <int C>
__device__ __forceinline__ void pyr_unit_tile(const float* __restrict__ src, float* __restrict__ dst,
                                              const PyrTab& tab, const PyrLevelDev& lv, int ty, int tx,
                                              float* smem) {
    constexpr int TW = kUnitTW, TH = kUnitTH, SW = kUnitSW, SH = kUnitSH;
    float* s_src = smem;                 // [SH][SW][C]
    float* s_h = smem + SH * SW * C;     // [SH][TW][C]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = tx * TW, y0 = ty * TH;
    const int W = tab.W;
    float wx[5], wy[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        wx[i] = tab.xw[(long long)lv.xtab_off * 6 + i];
        wy[i] = tab.yw[(long long)lv.ytab_off * 6 + i];
    }

    // stage rows y0-2 .. y0+33, columns x0-4 .. x0+67 (mirrored at the crop border)
    const bool fast = (x0 - 4 >= 0) && (x0 + TW + 4 <= lv.src_w) && (((lv.src_x0 * C) & 3) == 0) &&
                      (((W * C) & 3) == 0);
    // Loads are issued in predicated batches (clamped index, no branch around a load) so that every
    // thread has several requests in flight before the first wait.
    if (fast) {
        constexpr int V4 = SW * C / 4;  // float4 per staged row
        constexpr int NB = (SH * V4 + 255) / 256;
        float4 v[NB];
        const float* __restrict__ base = src + ((long long)lv.src_y0 * W + lv.src_x0 + x0 - 4) * C;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = min(tid + 256 * k, SH * V4 - 1);
            const int r = p / V4, q = p - r * V4;
            const int sy = mirror_index(y0 - 2 + r, lv.src_h);
            v[k] = *reinterpret_cast<const float4*>(base + (long long)sy * W * C + q * 4);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = tid + 256 * k;
            const int r = p / V4, q = p - r * V4;
            if (p < SH * V4) *reinterpret_cast<float4*>(s_src + (r * SW) * C + q * 4) = v[k];
        }
    } else {
        constexpr int NB = (SH * SW + 255) / 256;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            float v[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int p = min(tid + 256 * k, SH * SW - 1);
                const int r = p / SW, c = p - r * SW;
                const int sy = mirror_index(y0 - 2 + r, lv.src_h) + lv.src_y0;
                const int sx = mirror_index(x0 - 4 + c, lv.src_w) + lv.src_x0;
                v[k] = src[((long long)sy * W + sx) * C + ch];
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int p = tid + 256 * k;
                if (p < SH * SW) s_src[p * C + ch] = v[k];
            }
        }
    }
    __syncthreads();

    // horizontal 5 taps: output column j reads staged columns j+2 .. j+6
    for (int r = wave; r < SH; r += 4) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 5; ++i) acc = __builtin_fmaf(wx[i], s_src[(r * SW + lane + 2 + i) * C + ch], acc);
            s_h[(r * TW + lane) * C + ch] = acc;
        }
    }
    __syncthreads();

    // vertical 5 taps, a 5-row register window sliding down the wave's 8 rows
    const int ox = x0 + lane;
    if (ox >= lv.out_w) return;
    constexpr int R = TH / 4;
    const int r0 = wave * R;
    float win[5][C];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) win[j + 1][ch] = s_h[((r0 + j) * TW + lane) * C + ch];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int oy = y0 + r0 + rr;
        if (oy >= lv.out_h) break;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
#pragma unroll
            for (int j = 0; j < 4; ++j) win[j][ch] = win[j + 1][ch];
            win[4][ch] = s_h[((r0 + rr + 4) * TW + lane) * C + ch];
        }
        const bool live = oy < lv.zoom_h && ox < lv.zoom_w;
        float* __restrict__ po = dst + ((long long)oy * lv.out_w + ox) * C;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 5; ++j) acc = __builtin_fmaf(wy[j], win[j][ch], acc);
            po[ch] = live ? acc : 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------ DENSE
template <int C>
__device__ __forceinline__ void pyr_dense_tile(const float* __restrict__ src, float* __restrict__ dst,
                                               const PyrTab& tab, const PyrLevelDev& lv, int ty, int tx,
                                               float* smem) {
    constexpr int SWM = dense_max_sw(C);
    float* s_src = smem;                          // [kDenseMaxSH][SWM][C]
    float* s_v = smem + kDenseMaxSH * SWM * C;    // [kDenseMaxTH][SWM][C]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ox0 = tx * lv.tile_w, oy0 = ty * lv.tile_h;
    const int W = tab.W;
    // output rows / columns of this tile that the resampler produces (the rest of the canvas is 0)
    const int ncols = min(lv.tile_w, min(lv.zoom_w, lv.out_w) - ox0);
    const int nrows = min(lv.tile_h, min(lv.zoom_h, lv.out_h) - oy0);
    int cx0 = 0, SW = 0, ry0 = 0, SH = 0;
    if (ncols > 0 && nrows > 0) {
        cx0 = tab.xbase[lv.xtab_off + ox0] - 2;
        SW = min(tab.xbase[lv.xtab_off + ox0 + ncols - 1] + 3 - cx0 + 1, SWM);  // host guarantees <=
        ry0 = tab.ybase[lv.ytab_off + oy0] - 2;
        SH = min(tab.ybase[lv.ytab_off + oy0 + nrows - 1] + 3 - ry0 + 1, kDenseMaxSH);
    }
    // per-thread horizontal taps (column = lane; tile_w <= 64); loads are unconditional on a clamped index
    const int oxc = lv.xtab_off + min(ox0 + lane, max(lv.zoom_w - 1, 0));
    const int xo = tab.xbase[oxc] - 2 - cx0;
    float wx[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) wx[i] = tab.xw[(long long)oxc * 6 + i];

    // A: stage the contiguous source footprint (mirrored at the crop border).  Lanes = consecutive staged
    // floats, the wave's rows are wave-uniform; loads go out in predicated batches of KR per thread.
    if (SH > 0 && SW > 0) {
        constexpr int KC = (SWM * C + 63) / 64;
        constexpr int KR = kDenseMaxSH / 4;
        long long rowoff[KR];
#pragma unroll
        for (int j = 0; j < KR; ++j)
            rowoff[j] = (long long)(mirror_index(ry0 + min(wave + 4 * j, SH - 1), lv.src_h) + lv.src_y0) * W;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int cf = lane + 64 * k;
            const int c = cf / C, ch = cf - c * C;
            const int sx = mirror_index(cx0 + min(c, SW - 1), lv.src_w) + lv.src_x0;
            float v[KR];
#pragma unroll
            for (int j = 0; j < KR; ++j) v[j] = src[(rowoff[j] + sx) * C + ch];
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const int r = wave + 4 * j;
                if (c < SW && r < SH) s_src[(r * SWM) * C + cf] = v[j];
            }
        }
    }
    __syncthreads();
    // B: vertical 6 taps for every staged column (conflict-free: lanes = consecutive columns)
    for (int orow = wave; orow < nrows; orow += 4) {
        const int oy = lv.ytab_off + oy0 + orow;
        const int s0 = min(max(tab.ybase[oy] - 2 - ry0, 0), kDenseMaxSH - 6);
        float wy[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) wy[j] = tab.yw[(long long)oy * 6 + j];
        for (int c = lane; c < SW; c += 64) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 6; ++j) acc = __builtin_fmaf(wy[j], s_src[((s0 + j) * SWM + c) * C + ch], acc);
                s_v[(orow * SWM + c) * C + ch] = acc;
            }
        }
    }
    __syncthreads();
    // C: horizontal 6 taps and store
    const int ox = ox0 + lane;
    if (lane >= lv.tile_w || ox >= lv.out_w) return;
    const int xs = min(max(xo, 0), SWM - 6);
    for (int orow = wave; orow < lv.tile_h; orow += 4) {
        const int oy = oy0 + orow;
        if (oy >= lv.out_h) break;
        const bool live = orow < nrows && lane < ncols;
        float* __restrict__ po = dst + ((long long)oy * lv.out_w + ox) * C;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; ++i) acc = __builtin_fmaf(wx[i], s_v[(orow * SWM + xs + i) * C + ch], acc);
            po[ch] = live ? acc : 0.0f;
        }
    }
}

// ------------------------------------------------------------------------------------------ SPARSE
template <int C>
__device__ __forceinline__ void pyr_sparse_tile(const float* __restrict__ src, float* __restrict__ dst,
                                                const PyrTab& tab, const PyrLevelDev& lv, int ty, int tx) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ox = tx * kSparseTW + lane, oy = ty * kSparseTH + wave;
    if (ox >= lv.out_w || oy >= lv.out_h) return;
    const bool live = ox < lv.zoom_w && oy < lv.zoom_h;
    const long long xe = lv.xtab_off + min(ox, lv.zoom_w - 1), ye = lv.ytab_off + min(oy, lv.zoom_h - 1);
    int xi[6], yi[6];
    float wx[6], wy[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        xi[i] = (tab.xidx[xe * 6 + i] + lv.src_x0) * C;
        wx[i] = tab.xw[xe * 6 + i];
        yi[i] = tab.yidx[ye * 6 + i] + lv.src_y0;
        wy[i] = tab.yw[ye * 6 + i];
    }
    float acc[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) acc[ch] = 0.0f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const float* __restrict__ row = src + (long long)yi[j] * tab.W * C;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            float h = 0.0f;
#pragma unroll
            for (int i = 0; i < 6; ++i) h = __builtin_fmaf(wx[i], row[xi[i] + ch], h);
            acc[ch] = __builtin_fmaf(wy[j], h, acc[ch]);
        }
    }
    float* __restrict__ po = dst + ((long long)oy * lv.out_w + ox) * C;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) po[ch] = live ? acc[ch] : 0.0f;
}

template <int C>
__global__ __launch_bounds__(256) void pyramid_kernel(const float* __restrict__ frames,
                                                      float* __restrict__ pyr, const PyrTab tab) {
    __shared__ __attribute__((aligned(16))) float smem[pyr_lds_floats<C>()];

    const unsigned bid = blockIdx.x;
    const int frame = (int)(bid / (unsigned)tab.tiles_per_frame);
    int rem = (int)(bid - (unsigned)frame * (unsigned)tab.tiles_per_frame);
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < tab.n_levels && rem >= tab.tile_start[i]) l = i;
    rem -= tab.tile_start[l];
    const PyrLevelDev& lv = tab.lv[l];
    const int ty = rem / tab.tiles_x[l];
    const int tx = rem - ty * tab.tiles_x[l];
    const float* __restrict__ src = frames + (long long)frame * tab.H * tab.W * C;
    float* __restrict__ dst = pyr + ((long long)frame * tab.frame_px_out + tab.px_off[l]) * C;
    if (lv.kind == kPyrUnit)
        pyr_unit_tile<C>(src, dst, tab, lv, ty, tx, smem);
    else if (lv.kind == kPyrDense)
        pyr_dense_tile<C>(src, dst, tab, lv, ty, tx, smem);
    else
        pyr_sparse_tile<C>(src, dst, tab, lv, ty, tx);
}

}  // namespace silent
