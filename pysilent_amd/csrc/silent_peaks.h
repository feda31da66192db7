// Pointwise ops, per-level reductions, 3x3 NMS and ordered keypoint compaction.
#pragma once

#include "silent_common.h"

namespace silent {

// ---- a-7 pad_inwards: out = mask * in (multiplication, like the reference: 0 * NaN stays NaN)
__global__ __launch_bounds__(256) void pad_inwards_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          const LevelTab tab, int C, int pt, int pb, int pl,
                                                          int pr) {
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const int npx = H * W;
    for (int k = 0; k < 4; ++k) {
        const int p = tc.tx * kChunk + k * 256 + threadIdx.x;
        if (p >= npx) break;
        const int y = p / W, x = p - y * W;
        const float m = (y >= pt && y < H - pb && x >= pl && x < W - pr) ? 1.0f : 0.0f;
        for (int c = 0; c < C; ++c) out[(base_px + p) * C + c] = m * in[(base_px + p) * C + c];
    }
}

// ---- a-8 get_value_from_color: ((x0 + x1) + x2 ...) * float32(1/C)
__global__ __launch_bounds__(256) void value_from_color_kernel(const float* __restrict__ in,
                                                               float* __restrict__ out, long long npx, int C) {
    const float inv = 1.0f / (float)C;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npx; p += (long long)gridDim.x * 256) {
        float s = in[p * C];
        for (int c = 1; c < C; ++c) s = __fadd_rn(s, in[p * C + c]);
        out[p] = __fmul_rn(s, inv);
    }
}

// ---- get_bw_from_color: (x0 + x1 + ... != 0) ? 1 : 0 (NaN != 0 is true)
__global__ __launch_bounds__(256) void bw_from_color_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            long long npx, int C) {
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npx; p += (long long)gridDim.x * 256) {
        float s = in[p * C];
        for (int c = 1; c < C; ++c) s = __fadd_rn(s, in[p * C + c]);
        out[p] = (s != 0.0f) ? 1.0f : 0.0f;
    }
}

// ---- a-9 3x3 non-max suppression (max-pool SAME ignores out-of-level taps)
__global__ __launch_bounds__(256) void nms3x3_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                     const LevelTab tab, int C, int mode) {
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const float* __restrict__ src = in + base_px * C;
    float* __restrict__ dst = out + base_px * C;
    const int npx = H * W;
    for (int k = 0; k < 4; ++k) {
        const int p = tc.tx * kChunk + k * 256 + threadIdx.x;
        if (p >= npx) break;
        const int y = p / W, x = p - y * W;
        for (int c = 0; c < C; ++c) {
            float m = kPoolLowest;
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    m = pool_max(m, src[((long long)yy * W + xx) * C + c]);
                }
            }
            const float v = src[(long long)p * C + c];
            const bool is_max = v == m;
            dst[(long long)p * C + c] = mode == SILENT_NMS_FIRED ? (is_max ? 1.0f : 0.0f) : v * (is_max ? v : 0.0f);
        }
    }
}

// ---- a-10 per-level max / min: wave shuffle reduction -> LDS -> one atomic pair per block.
// mm[(frame * n_levels + level) * 2 + {0,1}] = ordered-uint max_pool(v) / max_pool(-v); pre-set by init_maxmin_kernel.
// NaN values are ignored by both (pool_max); read back with level_max / level_min.
__global__ void init_maxmin_kernel(unsigned* mm, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        mm[2 * i] = pool_lowest_ord();      // max_pool(v)
        mm[2 * i + 1] = pool_lowest_ord();  // max_pool(-v): the minimum is -1.0 * this (top_value_points.py:19-21)
    }
}

// one launch for the tables of the selection tail: extrema slots and cell maxima start from lowest(); the sparse tail's counters,
// hit masks and flags (one 16-byte aligned region of the workspace, n_zero16 pieces) from 0
__global__ void init_select_kernel(unsigned* mm, int n_mm2, unsigned* cells, long long n_cells, uint4* zero, long long n_zero16) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n_mm2) mm[i] = pool_lowest_ord();
    if (i < n_cells) cells[i] = pool_lowest_ord();
    for (long long k = i; k < n_zero16; k += (long long)gridDim.x * 256) zero[k] = uint4{0u, 0u, 0u, 0u};
}

// If `color` is non-null the value is computed on the fly as get_value_from_color does.
// kRedChunk pixels per block (64 per thread, 8 requested at a time): with 1024-pixel blocks this pass ran at
// 0.65 TB/s (86 k blocks of a few microseconds each, two global atomics per block).
constexpr int kRedChunk = 16384;
__global__ __launch_bounds__(256) void level_maxmin_kernel(const float* __restrict__ value,
                                                           const float* __restrict__ color, int C,
                                                           const LevelTab tab, unsigned* __restrict__ mm) {
    __shared__ float s_mx[4], s_nmn[4];
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int npx = tab.h[tc.level] * tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const float inv = 1.0f / (float)C;
    float mx = kPoolLowest, nmn = kPoolLowest;   // running max_pool(v), max_pool(-v)
    const int p0 = tc.tx * kRedChunk + threadIdx.x;
    for (int k0 = 0; k0 < kRedChunk / 256; k0 += 8) {
        if (p0 + k0 * 256 >= npx) break;  // (the first lane of the block decides: block-uniform enough, see clamp)
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            // clamped index: re-reading the last pixel of the level changes neither the max nor the min
            const long long p = base_px + min(p0 + (k0 + k) * 256, npx - 1);
            if (value) {
                v[k] = value[p];
            } else {
                float t = color[p * C];
                for (int c = 1; c < C; ++c) t = __fadd_rn(t, color[p * C + c]);
                v[k] = __fmul_rn(t, inv);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            mx = pool_max(mx, v[k]);
            nmn = pool_max(nmn, -v[k]);
        }
    }
    mx = wave_max(mx);
    nmn = wave_max(nmn);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_mx[wave] = mx;
        s_nmn[wave] = nmn;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) {
            mx = pool_max(mx, s_mx[i]);
            nmn = pool_max(nmn, s_nmn[i]);
        }
        unsigned* slot = mm + ((long long)tc.frame * tab.n_levels + tc.level) * 2;
        atomicMax(slot, f2ord(mx));
        atomicMax(slot + 1, f2ord(nmn));
    }
}

__device__ __forceinline__ float level_max(const unsigned* slot) { return ord2f(slot[0]); }
__device__ __forceinline__ float level_min(const unsigned* slot) { return __fmul_rn(-1.0f, ord2f(slot[1])); }

// thr = (1-p)*max + p*min with every op rounded to float32 and NOT fused (TF runs mul, mul, add)
__global__ __launch_bounds__(256) void top_value_points_kernel(const float* __restrict__ color,
                                                               const float* __restrict__ value,
                                                               float* __restrict__ out, const LevelTab tab, int C,
                                                               float one_minus_p, float p_f,
                                                               const unsigned* __restrict__ mm) {
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int npx = tab.h[tc.level] * tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const unsigned* slot = mm + ((long long)tc.frame * tab.n_levels + tc.level) * 2;
    const float thr = __fadd_rn(__fmul_rn(one_minus_p, level_max(slot)), __fmul_rn(p_f, level_min(slot)));
    const float inv = 1.0f / (float)C;
    for (int k = 0; k < 4; ++k) {
        const int p = tc.tx * kChunk + k * 256 + threadIdx.x;
        if (p >= npx) break;
        float v;
        if (value) {
            v = value[base_px + p];
        } else {
            v = color[(base_px + p) * C];
            for (int c = 1; c < C; ++c) v = __fadd_rn(v, color[(base_px + p) * C + c]);
            v = __fmul_rn(v, inv);
        }
        const float m = v >= thr ? 1.0f : 0.0f;
        for (int c = 0; c < C; ++c) out[(base_px + p) * C + c] = color[(base_px + p) * C + c] * m;
    }
}

// ---- a-11 max_value_indices_region
// The TF1 op is max_pool(k = full extent, stride = region, SAME): every window is the level clipped to
// a shifted copy of itself, i.e. a PREFIX or a SUFFIX of rows (and of columns).  The distinct window
// edges cut each axis into <= kMaxSeg segments; one pass computes the max of every (row segment x
// column segment) cell, and a window maximum is the max over the cells it covers.
constexpr int kMaxWin = 4;  // windows per axis (the reference uses 2); keeps RegionTab inside the kernarg budget
constexpr int kMaxSeg = 8;  // segments per axis (<= 2 * windows)

struct RegionLevel {
    // general path (any number of windows): TF1 geometry as arithmetic, tables in the workspace
    int ry, rx;          // strides = region extents
    int pad_y, pad_x;    // pad_before of max_pool SAME: window j starts at j * stride - pad_before
    long long m1_off;    // offset of this level's [h][ow] row maxima inside one frame's M1 table
    long long pooled_off;  // offset of this level's [oh][ow] window maxima inside one frame's pooled table
    int oh, ow;          // windows per axis
    int nrs, ncs;        // segments per axis
    int rcut[kMaxSeg + 1];  // row segment s = [rcut[s], rcut[s+1])
    int ccut[kMaxSeg + 1];
    int wy_lo[kMaxWin], wy_hi[kMaxWin];  // window j covers row SEGMENTS [lo, hi)
    int wx_lo[kMaxWin], wx_hi[kMaxWin];
    float yscale, xscale;  // float32 (windows / extent): TF1 CalculateResizeScale
};

struct RegionTab {
    RegionLevel lv[kMaxLevels];
    long long m1_per_frame, pooled_per_frame;   // general path: floats per frame of the two tables
};

constexpr int kCells = kMaxSeg * kMaxSeg;

// ---- a-9 as a streaming stencil (C = 1 or 3; other channel counts use nms3x3_kernel above): lane = column (halo 2
// like the other streaming kernels), rows walk down, 3x3 maximum = row maximum by DPP then a 3-row window; taps
// outside the image (and NaN taps) are ignored, see pool_max.  The per-pixel kernel with its 9 strided loads per channel ran at 1.0 TB/s on 3-channel
// 1080p maps.  Same comparisons in the same order: identical results, NaNs included.
constexpr int kNmsCols = 60, kNmsTW = 4 * kNmsCols, kNmsTH = 32;

template <int C>
__global__ __launch_bounds__(256) void nms3x3_stream_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                            const LevelTab tab, int mode) {
    constexpr int R = kNmsTH, CH = 6;
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = tc.tx * kNmsTW + wave * kNmsCols;
    if (xw0 >= W) return;  // wave-uniform
    const int y0 = tc.ty * R;
    const int x = xw0 + lane - 2;
    const bool col_ok = x >= 0 && x < W;
    const int xc = min(max(x, 0), W - 1);
    const bool out_lane = lane >= 2 && lane < 2 + kNmsCols && x < W;
    float hm[C][2], ctr[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        hm[c][0] = hm[c][1] = kPoolLowest;
        ctr[c] = 0.0f;
    }
#pragma unroll 1
    for (int i0 = 0; i0 < R + 2; i0 += CH) {
        float v[CH][C];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int y = y0 - 1 + i0 + j;
            const long long px = base_px + (long long)min(max(y, 0), H - 1) * W + xc;
#pragma unroll
            for (int c = 0; c < C; ++c) v[j][c] = in[px * C + c];
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int i = i0 + j;
            if (i >= R + 2) break;  // wave-uniform
            const int y = y0 - 1 + i;  // arriving row; row y - 1 completes
            const bool in_img = y >= 0 && y < H && col_ok;
            float o[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float tp = in_img ? v[j][c] : kPoolLowest;
                const float l = from_lane_below(tp), r = from_lane_above(tp);
                // (the DPP shift feeds 0 into the wave's edge lanes: those are halo lanes whose result is never stored)
                const float h = pool_max(pool_max(pool_max(kPoolLowest, l), tp), r);
                const float mx = pool_max(pool_max(hm[c][0], hm[c][1]), h);
                const float xv = ctr[c];
                const bool is_max = xv == mx;
                o[c] = mode == SILENT_NMS_FIRED ? (is_max ? 1.0f : 0.0f) : xv * (is_max ? xv : 0.0f);
                hm[c][0] = hm[c][1];
                hm[c][1] = h;
                ctr[c] = v[j][c];
            }
            const int yo = y - 1;
            if (i >= 2 && yo < H && out_lane) {
                float* __restrict__ po = out + (base_px + (long long)yo * W + x) * C;
#pragma unroll
                for (int c = 0; c < C; ++c) po[c] = o[c];
            }
        }
    }
}

// Sparse tail: a kernel that usually finds nothing to do runs with a BOUNDED grid.  Block b owns the tiles b, b + G, b + 2 G ...
// (at most 64 of them: host-checked, sparse_grid); lane i of every wave looks at tile b + i G -- one round trip -- and the block
// then visits the live ones only.  select_peaks_kernel on config 3: 11 600 blocks that all return at once took 9.4 us, 2048 blocks
// take 4.7.  (Measured and not kept: the same for region_count_kernel and region_write_kernel -- their time is the work of the
// "zero map" levels, every pixel of a window without a peak is a keypoint -- and more summary entries per thread in
// sparse_select_kernel: a wave handles the groups that reach the threshold one after the other, 16x the entries per wave made
// the kernel 1.6 - 2.7x longer.  profiles/r05_experiments.txt)
// `live` must give every wave of the block the same answer.
template <class Live, class Body>
__device__ __forceinline__ void for_live_tiles(unsigned n_tiles, Live live, Body body) {
    if (gridDim.x >= n_tiles) {
        body(blockIdx.x);
        return;
    }
    const unsigned t = blockIdx.x + (threadIdx.x & 63u) * gridDim.x;
    unsigned long long todo = __ballot(t < n_tiles && live(t));
    while (todo) {   // block-uniform
        const int i = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        body(blockIdx.x + (unsigned)i * gridDim.x);
    }
}

// ---- a-10 -> a-9 -> a-8 in one streaming pass ("top 10 %, NMS" between the chain and the keypoints)
//   top   = color * (value >= thr ? 1 : 0)                          top_value_points_kernel
//   peaks = top * (top == maxpool3x3 SAME(top) ? top : 0)           nms3x3_kernel, SILENT_NMS_PRODUCT
//   pv    = (sum_c peaks) * float32(1 / C)                          value_from_color_kernel
// Same operations in the same order as the three separate kernels (bit-identical, tested), but the two
// intermediate colour maps stay in registers unless the caller asks for them: 16 + 4 bytes per pixel instead of 72.
// Wave-autonomous streaming like gray_line_end_kernel: lane = column (halo 2), rows walk down, the 3x3 maximum is a
// row maximum by DPP followed by a 3-row window; out-of-image taps are -inf (max_pool ignores them).
constexpr int kSelCols = 60, kSelTW = 4 * kSelCols, kSelTH = 32;

// CELLS: also fold the cell maxima of max_value_indices_region (region_cell_max_kernel) over the peak value into this
// pass -- the lane's column segment is fixed, the row segment is wave-uniform, so it is one running maximum per lane
// and a segmented wave reduction + one atomic per (row segment, column segment) and tile.
template <int C, bool CELLS>
__device__ __forceinline__ void select_peaks_tile(const float* __restrict__ color, const float* __restrict__ value,
                                                  float* __restrict__ top_out, float* __restrict__ peaks_out,
                                                  float* __restrict__ pv_out, const LevelTab& tab, float one_minus_p, float p_f,
                                                  const unsigned* __restrict__ mm, const RegionTab& rt, unsigned* __restrict__ cells,
                                                  const int* __restrict__ dense_flags, unsigned tile) {
    constexpr int R = kSelTH;
    const TileCoord tc = locate_tile(tab, tile);
    // sparse tail (sparse_modes_kernel): only the (frame, level)s in dense mode run this pass
    if (dense_flags && dense_flags[tc.frame * tab.n_levels + tc.level] != kTailDense) return;
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = tc.tx * kSelTW + wave * kSelCols;
    if (xw0 >= W) return;  // wave-uniform
    const int y0 = tc.ty * R;
    const int x = xw0 + lane - 2;
    const bool col_ok = x >= 0 && x < W;
    const int xc = min(max(x, 0), W - 1);
    const bool out_lane = lane >= 2 && lane < 2 + kSelCols && x < W;
    const unsigned* slot = mm + ((long long)tc.frame * tab.n_levels + tc.level) * 2;
    const float thr = __fadd_rn(__fmul_rn(one_minus_p, level_max(slot)), __fmul_rn(p_f, level_min(slot)));
    const float inv = 1.0f / (float)C;

    // CELLS state: column segment of this lane, row segment of the rows seen so far, running maximum
    const RegionLevel& rl = rt.lv[tc.level];
    int cseg = 0, cur_rs = -1;
    float cmax = kPoolLowest;
    if constexpr (CELLS) {
        for (int s_ = 1; s_ < rl.ncs; ++s_) cseg = x >= rl.ccut[s_] ? s_ : cseg;
    }
    auto flush_cells = [&]() {
        if (cur_rs < 0) return;  // wave-uniform
        unsigned* dst = cells + ((long long)tc.frame * tab.n_levels + tc.level) * kCells + cur_rs * kMaxSeg;
        int key = out_lane ? cseg : -1;
        unsigned long long left = __ballot(key >= 0);
        while (left) {
            const int first = __ffsll((long long)left) - 1;
            const int k = __builtin_amdgcn_readlane(key, first);
            const bool mine = key == k;
            const float m = wave_max(mine ? cmax : kPoolLowest);
            if (lane == first) atomicMax(dst + k, f2ord(m));
            key = mine ? -1 : key;
            left = __ballot(key >= 0);
        }
        cmax = kPoolLowest;
    };
    float hm[C][2], ctr[C];  // row maxima of rows y-2, y-1; centre values of row y-1
#pragma unroll
    for (int c = 0; c < C; ++c) {
        hm[c][0] = hm[c][1] = kPoolLowest;
        ctr[c] = 0.0f;
    }
    constexpr int CH = 8;  // rows requested per batch
    static_assert((R + 2) % 2 == 0, "");
#pragma unroll 1
    for (int i0 = 0; i0 < R + 2; i0 += CH) {
        float col[CH][C], val[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int y = y0 - 1 + i0 + j;
            const long long px = base_px + (long long)min(max(y, 0), H - 1) * W + xc;
#pragma unroll
            for (int c = 0; c < C; ++c) col[j][c] = color[px * C + c];
            val[j] = value ? value[px] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int i = i0 + j;
            if (i >= R + 2) break;  // wave-uniform
            const int y = y0 - 1 + i;  // arriving row; row y - 1 completes
            const bool in_img = y >= 0 && y < H && col_ok;
            float v = val[j];
            if (!value) {
                v = col[j][0];
#pragma unroll
                for (int c = 1; c < C; ++c) v = __fadd_rn(v, col[j][c]);
                v = __fmul_rn(v, inv);
            }
            const float m = v >= thr ? 1.0f : 0.0f;
            float t[C], o[C];
#pragma unroll
            for (int c = 0; c < C; ++c) {
                t[c] = __fmul_rn(col[j][c], m);
                const float tp = in_img ? t[c] : kPoolLowest;  // max_pool SAME ignores taps outside the image
                const float l = from_lane_below(tp), r = from_lane_above(tp);
                const float h = pool_max(pool_max(pool_max(kPoolLowest, l), tp), r);   // NaN taps are ignored too
                const float mx = pool_max(pool_max(hm[c][0], hm[c][1]), h);
                const float xv = ctr[c];
                o[c] = __fmul_rn(xv, xv == mx ? xv : 0.0f);
                hm[c][0] = hm[c][1];
                hm[c][1] = h;
            }
            const int yo = y - 1;
            if (i >= 2 && yo < H && out_lane) {  // rows y0 .. y0 + R - 1
                const long long px = base_px + (long long)yo * W + x;
                if (top_out) {
#pragma unroll
                    for (int c = 0; c < C; ++c) top_out[px * C + c] = ctr[c];
                }
                if (peaks_out) {
#pragma unroll
                    for (int c = 0; c < C; ++c) peaks_out[px * C + c] = o[c];
                }
                if (pv_out || CELLS) {
                    float pv = o[0];
#pragma unroll
                    for (int c = 1; c < C; ++c) pv = __fadd_rn(pv, o[c]);
                    pv = __fmul_rn(pv, inv);
                    if (pv_out) pv_out[px] = pv;
                    if constexpr (CELLS) cmax = pool_max(cmax, pv);
                }
            }
            if constexpr (CELLS) {
                // the NEXT output row may start a new row segment: hand the finished one over first (wave-uniform)
                const int yn = yo + 1;
                if (i >= 1 && yn < H && yn < y0 + R) {
                    int rs = 0;
                    for (int s_ = 1; s_ < rl.nrs; ++s_) rs = yn >= rl.rcut[s_] ? s_ : rs;
                    if (rs != cur_rs) {
                        flush_cells();
                        cur_rs = rs;
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < C; ++c) ctr[c] = t[c];
        }
    }
    if constexpr (CELLS) flush_cells();
}

// n_tiles: tiles of `tab` (the sparse tail launches fewer blocks than that: for_live_tiles)
template <int C, bool CELLS>
__global__ __launch_bounds__(256) void select_peaks_kernel(const float* __restrict__ color,
                                                           const float* __restrict__ value,
                                                           float* __restrict__ top_out, float* __restrict__ peaks_out,
                                                           float* __restrict__ pv_out, const LevelTab tab,
                                                           float one_minus_p, float p_f,
                                                           const unsigned* __restrict__ mm, const RegionTab rt,
                                                           unsigned* __restrict__ cells,
                                                           const int* __restrict__ dense_flags, unsigned n_tiles) {
    for_live_tiles(
        n_tiles,
        [&](unsigned t) {
            const TileCoord tc = locate_tile(tab, t);
            return dense_flags[tc.frame * tab.n_levels + tc.level] == kTailDense;
        },
        [&](unsigned t) { select_peaks_tile<C, CELLS>(color, value, top_out, peaks_out, pv_out, tab, one_minus_p, p_f, mm, rt, cells, dense_flags, t); });
}



// (y, x) of pixel p are carried by the caller: the kernels walk their pixels with a fixed stride, so one integer
// division per thread replaces one per pixel
__device__ __forceinline__ void advance_yx(int& y, int& x, int step, int W) {
    x += step;
    while (x >= W) {
        x -= W;
        ++y;
    }
}

__global__ void init_cells_kernel(unsigned* cells, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) cells[i] = pool_lowest_ord();
}

// cells[(frame * n_levels + level) * kCells + rs * kMaxSeg + cs] = ordered-uint max of the cell
__global__ __launch_bounds__(256) void region_cell_max_kernel(const float* __restrict__ value, const LevelTab tab,
                                                              const RegionTab rt, unsigned* __restrict__ cells) {
    __shared__ unsigned s_cell[kCells];
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const RegionLevel& rl = rt.lv[tc.level];
    const int W = tab.w[tc.level];
    const int npx = tab.h[tc.level] * W;
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    for (int i = threadIdx.x; i < kCells; i += 256) s_cell[i] = pool_lowest_ord();
    __syncthreads();
    int y = (tc.tx * kRedChunk + (int)threadIdx.x) / W, x = tc.tx * kRedChunk + (int)threadIdx.x - y * W;
    for (int k = 0; k < kRedChunk / 256; ++k, advance_yx(y, x, 256, W)) {
        const int p = tc.tx * kRedChunk + k * 256 + threadIdx.x;
        if (tc.tx * kRedChunk + k * 256 >= npx) break;  // block-uniform
        const bool live = p < npx;
        int cell = -1;
        float v = kPoolLowest;
        if (live) {
            int rs = 0, cs = 0;
            for (int s = 1; s < rl.nrs; ++s) rs = y >= rl.rcut[s] ? s : rs;
            for (int s = 1; s < rl.ncs; ++s) cs = x >= rl.ccut[s] ? s : cs;
            cell = rs * kMaxSeg + cs;
            v = pool_max(kPoolLowest, value[base_px + p]);   // a NaN never reaches the reductions below
        }
        // common case: the whole wave sits in one cell -> one LDS atomic per wave
        const int first = __builtin_amdgcn_readfirstlane(cell);
        if (__all(cell == first)) {
            const float m = wave_max(v);
            if ((threadIdx.x & 63) == 0 && first >= 0) atomicMax(&s_cell[first], f2ord(m));
        } else if (live) {
            atomicMax(&s_cell[cell], f2ord(v));
        }
    }
    __syncthreads();
    unsigned* dst = cells + ((long long)tc.frame * tab.n_levels + tc.level) * kCells;
    for (int i = threadIdx.x; i < kCells; i += 256)
        if (s_cell[i] != pool_lowest_ord()) atomicMax(dst + i, s_cell[i]);
}

// window maxima of one (frame, level) into LDS: pooled[j * kMaxWin + i]
// `cells`: the level's kCells cell maxima -- in LDS (pooled_from_cells), or in memory: staged with ONE round trip first (a thread
// that walks its window's cells in memory pays a dependent load per cell: 9 in a row for a window of 3 x 3 segments)
__device__ __forceinline__ void pooled_from_cells(const unsigned* cells, const RegionLevel& rl, float* s_pooled) {
    for (int wi = threadIdx.x; wi < rl.oh * rl.ow; wi += 256) {
        const int j = wi / rl.ow, i = wi - j * rl.ow;
        unsigned m = pool_lowest_ord();
        for (int rs = rl.wy_lo[j]; rs < rl.wy_hi[j]; ++rs)
            for (int cs = rl.wx_lo[i]; cs < rl.wx_hi[i]; ++cs) {
                const unsigned c = cells[rs * kMaxSeg + cs];
                m = c > m ? c : m;
            }
        s_pooled[j * kMaxWin + i] = ord2f(m);
    }
}
__device__ __forceinline__ void load_pooled(const unsigned* __restrict__ cells, const RegionLevel& rl, float* s_pooled, unsigned* s_cells) {
    if (threadIdx.x < kCells) s_cells[threadIdx.x] = cells[threadIdx.x];
    __syncthreads();
    pooled_from_cells(s_cells, rl, s_pooled);
}


// threshold of pixel (y, x): the window maximum its cell maps back to.  `pooled` is the LDS copy of the (<= 4 x 4)
// window maxima with row stride kMaxWin, or (GEN) the level's [oh][ow] table in the workspace.
template <bool GEN>
__device__ __forceinline__ float region_thr(int y, int x, const RegionLevel& rl, const float* pooled) {
    // TF1 ResizeNearestNeighbor: min(floorf(dst * scale), in - 1), float32
    int sy = (int)floorf(__fmul_rn((float)y, rl.yscale));
    int sx = (int)floorf(__fmul_rn((float)x, rl.xscale));
    sy = sy > rl.oh - 1 ? rl.oh - 1 : sy;
    sx = sx > rl.ow - 1 ? rl.ow - 1 : sx;
    return pooled[sy * (GEN ? rl.ow : kMaxWin) + sx];
}

// ---- a-11, any region_shape (more than kMaxWin windows per axis): the windows of max_pool(k = full extent, stride = region,
// SAME) are PREFIXES (start j * stride - pad <= 0) or SUFFIXES of the rows and of the columns, so the window maxima are
// separable prefix / suffix maxima: pass 1 gives every row its maxima over the column windows (M1[y][i]), pass 2 walks
// each column of M1 down (prefix windows) and up (suffix windows).  max is exact and order-free (pool_max ignores NaN),
// so the result is bit-identical to the cell path and to the oracle.
// One block per (frame, level, row).  The row is walked in CHUNKS of kRowChunk pixels staged in LDS -- left to right for the
// prefix windows (a running prefix maximum is carried from chunk to chunk), right to left for the suffix windows -- so any
// level width works (round 3 staged the whole row twice and refused levels wider than 16 384 pixels).  Inside a chunk: every
// thread scans its slice, the slice totals are combined, and the windows whose last (prefix) / first (suffix) pixel lies in
// the chunk are written.
constexpr int kRowChunk = 8192;
__global__ __launch_bounds__(256) void region_rowmax_kernel(const float* __restrict__ value, const LevelTab tab,
                                                            const RegionTab rt, float* __restrict__ m1) {
    __shared__ float s_v[kRowChunk];
    __shared__ float s_tot[256];
    const TileCoord tc = locate_tile(tab, blockIdx.x);   // tile = one row: ty = y
    const RegionLevel& rl = rt.lv[tc.level];
    const int W = tab.w[tc.level], y = tc.ty;
    const float* __restrict__ row = value + (long long)tc.frame * tab.frame_px + tab.px_off[tc.level] + (long long)y * W;
    float* __restrict__ out = m1 + (long long)tc.frame * rt.m1_per_frame + rl.m1_off + (long long)y * rl.ow;
    const int tid = threadIdx.x;
    const int n_chunks = (W + kRowChunk - 1) / kRowChunk;
    for (int dir = 0; dir < 2; ++dir) {                 // 0: prefix maxima, chunks ascending; 1: suffix maxima, chunks descending
        float carry = kPoolLowest;                      // maximum of everything before (after) the current chunk
        for (int cc = 0; cc < n_chunks; ++cc) {
            const int c = dir == 0 ? cc : n_chunks - 1 - cc;
            const int x0 = c * kRowChunk, n = min(kRowChunk, W - x0);
            for (int x = tid; x < n; x += 256) s_v[x] = pool_max(kPoolLowest, row[x0 + x]);   // a NaN never wins
            __syncthreads();
            const int per = (n + 255) / 256, lo = min(tid * per, n), hi = min(lo + per, n);
            float m = kPoolLowest;
            if (dir == 0) {
                for (int x = lo; x < hi; ++x) {
                    m = pool_max(m, s_v[x]);
                    s_v[x] = m;
                }
            } else {
                for (int x = hi - 1; x >= lo; --x) {
                    m = pool_max(m, s_v[x]);
                    s_v[x] = m;
                }
            }
            s_tot[tid] = m;
            __syncthreads();
            float before = carry;                       // carry + the slices before (after) this thread's slice
            if (dir == 0) {
                for (int t = 0; t < tid; ++t) before = pool_max(before, s_tot[t]);
            } else {
                for (int t = tid + 1; t < 256; ++t) before = pool_max(before, s_tot[t]);
            }
            for (int x = lo; x < hi; ++x) s_v[x] = pool_max(before, s_v[x]);
            float all = carry;
            for (int t = 0; t < 256; ++t) all = pool_max(all, s_tot[t]);
            __syncthreads();
            // windows of max_pool(k = full extent, stride = region, SAME): start a = i * rx - pad_x; a <= 0: the prefix that ends
            // at min(a + W, W) - 1, else the suffix that starts at a
            for (int i = tid; i < rl.ow; i += 256) {
                const int a = i * rl.rx - rl.pad_x;
                const int key = a <= 0 ? min(a + W, W) - 1 : a;
                if ((a <= 0) == (dir == 0) && key >= x0 && key < x0 + n) out[i] = s_v[key - x0];
            }
            carry = all;
            __syncthreads();
        }
    }
}

// pass 2: one thread per (frame, level, column window i)
__global__ __launch_bounds__(256) void region_colmax_kernel(const float* __restrict__ m1, const LevelTab tab,
                                                            const RegionTab rt, int n_frames, long long cols_per_frame,
                                                            float* __restrict__ pooled) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int frame = (int)(gid / cols_per_frame);
    if (frame >= n_frames) return;
    long long rem = gid - (long long)frame * cols_per_frame;
    int l = 0;
    while (l + 1 < tab.n_levels && rem >= rt.lv[l].ow) {
        rem -= rt.lv[l].ow;
        ++l;
    }
    const RegionLevel& rl = rt.lv[l];
    const int i = (int)rem, H = tab.h[l];
    const float* __restrict__ col = m1 + (long long)frame * rt.m1_per_frame + rl.m1_off + i;
    float* __restrict__ out = pooled + (long long)frame * rt.pooled_per_frame + rl.pooled_off + i;
    // prefix windows (start <= 0), ends ascending with j
    float m = kPoolLowest;
    int j = 0;
    for (int y = 0; y < H; ++y) {
        m = pool_max(m, col[(long long)y * rl.ow]);
        while (j < rl.oh && j * rl.ry - rl.pad_y <= 0 && min(j * rl.ry - rl.pad_y + H, H) - 1 == y) {
            out[(long long)j * rl.ow] = m;
            ++j;
        }
    }
    // suffix windows (start > 0), starts descending with j from the last window
    m = kPoolLowest;
    int k = rl.oh - 1;
    for (int y = H - 1; y >= 0 && k >= j; --y) {
        m = pool_max(m, col[(long long)y * rl.ow]);
        while (k >= j && k * rl.ry - rl.pad_y == y) {
            out[(long long)k * rl.ow] = m;
            --k;
        }
    }
}

// pass 1: matches per chunk.  A chunk is kKpChunk pixels: wave w owns the w-th quarter, lane l the pixels
// quarter + k * 64 + l (coalesced 256-byte loads); hits are wave ballots, so ranks stay row-major without any
// per-thread bookkeeping.  (1024-pixel chunks with 4 consecutive pixels per thread ran at 1.7 TB/s.)
constexpr int kKpChunk = 4096, kKpPer = kKpChunk / 256;
template <bool GEN>
__global__ __launch_bounds__(256) void region_count_kernel(const float* __restrict__ value, const LevelTab tab,
                                                           const RegionTab rt, const unsigned* __restrict__ cells,
                                                           const float* __restrict__ pooled_g,
                                                           int* __restrict__ chunk_counts,
                                                           unsigned long long* __restrict__ hit_masks,
                                                           const int* __restrict__ dense_flags) {
    __shared__ float s_pooled[kMaxWin * kMaxWin];
    __shared__ unsigned s_cells[kCells];
    __shared__ int s_cnt[4];
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    // sparse tail: kTailSparse levels are settled by the candidates alone; kTailZero levels hold no NaN and every pixel that is
    // not a candidate has peak value 0, so the map is synthesised instead of read
    const int mode = dense_flags ? dense_flags[tc.frame * tab.n_levels + tc.level] : kTailDense;
    if (mode == kTailSparse) return;
    const RegionLevel& rl = rt.lv[tc.level];
    const int W = tab.w[tc.level];
    const int npx = tab.h[tc.level] * W;
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = tc.tx * kKpChunk + wave * (kKpChunk / 4);
    // the chunk's values are requested before the window maxima are staged: the two latencies overlap
    float v[kKpPer];
    if (mode == kTailDense) {
#pragma unroll
        for (int k = 0; k < kKpPer; ++k) v[k] = value[base_px + min(seg + k * 64 + lane, npx - 1)];
    } else {
#pragma unroll
        for (int k = 0; k < kKpPer; ++k) v[k] = 0.0f;
    }
    const float* pooled = s_pooled;
    if constexpr (GEN) {
        pooled = pooled_g + (long long)tc.frame * rt.pooled_per_frame + rl.pooled_off;
    } else {
        load_pooled(cells + ((long long)tc.frame * tab.n_levels + tc.level) * kCells, rl, s_pooled, s_cells);
        __syncthreads();
    }
    int n = 0;
    int y = (seg + lane) / W, x = seg + lane - y * W;
#pragma unroll
    for (int k = 0; k < kKpPer; ++k, advance_yx(y, x, 64, W)) {
        const int p = seg + k * 64 + lane;
        const bool hit = p < npx && v[k] >= region_thr<GEN>(y, x, rl, pooled);
        const unsigned long long m = __ballot(hit);
        n += __popcll(m);
        // the 64 hit bits of this group, kept for the write pass (1 bit per pixel instead of a second read of the value map
        // and a second evaluation of the thresholds)
        if (lane == 0) hit_masks[((long long)blockIdx.x * 4 + wave) * kKpPer + k] = m;
    }
    if (lane == 0) s_cnt[wave] = n;
    __syncthreads();
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
}

// ---- sparse selection tail (config 3 without a peak-value map): a-10 -> a-9 -> a-8 -> a-11 looking only where something can be.
// top_value_points keeps the pixels whose value reaches thr = (1 - p) * max + p * min of their level -- on natural and on
// noise frames a handful per level.  Everything after it is local to those "passers":
//   * a non-passer's peak value is 0 (or NaN where line_end holds a NaN): top = color * 0, and 0 * where(..) = 0 whatever
//     the 3x3 maximum is;
//   * a passer's peak value depends on the passers among its 8 neighbours only (non-passers and NaNs contribute 0 / nothing
//     to max_pool, and a passer's own t_c >= 0 is in its window);
//   * a window maximum of max_value_indices_region that is > 0 is the maximum over the passers in the window, and then the
//     keypoints of the pixels mapped to that window are passers too (0 >= positive is false, NaN >= x is false).
// So: the chain kernel leaves max_pool(value) per (pixel pair x kSumRows rows); sparse_select_kernel visits the entries that
// reach thr, evaluates the passers with the SAME operations as select_peaks_kernel (bit-identical peak values), folds them
// into the cell maxima and appends them to a per-frame candidate list; sparse_finish_kernel (one block per frame) checks per
// level that every window maximum is > 0 -- if not (an empty window: every non-NaN pixel mapped to it is a keypoint), or if
// the candidate list overflowed, the (frame, level) is flagged and the dense kernels run for it, every other block of them
// exits at once -- and turns the candidates that reach their window maximum into hit bits + chunk counts, the form the
// count pass leaves them in, so that scan and ordered write run unchanged (row-major order like tf.where by construction).
constexpr int kCandCap = 16384;   // candidates per frame; more -> the frame runs the dense kernels
struct Candidate {
    int level, y, x;
    float pv;
};

// t_c = color_c * (value >= thr ? 1 : 0) of pixel (y, x) as select_peaks_kernel computes it; lowest() outside the level
// (max_pool SAME ignores such taps).  The address is clamped and the load unconditional: the 9 taps of a passer are requested
// together, not one dependent round trip after the other.
__device__ __forceinline__ void sparse_top(const float* __restrict__ lev, int H, int W, int y, int x, float thr, float (&t)[3],
                                           bool* passer = nullptr) {
    const bool inside = y >= 0 && y < H && x >= 0 && x < W;
    const float* __restrict__ px = lev + ((long long)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) * 3;
    const float c0 = px[0], c1 = px[1], c2 = px[2];
    const float v = __fmul_rn(__fadd_rn(__fadd_rn(c0, c1), c2), 1.0f / 3.0f);
    const float m = v >= thr ? 1.0f : 0.0f;
    if (passer) *passer = inside && v >= thr;
    t[0] = inside ? __fmul_rn(c0, m) : kPoolLowest;
    t[1] = inside ? __fmul_rn(c1, m) : kPoolLowest;
    t[2] = inside ? __fmul_rn(c2, m) : kPoolLowest;
}

// grid: x = 256-entry chunks of one frame's summary, y = frame.  A lane looks at one entry; the entries that reach their level's
// threshold are then handled one after the other by the WHOLE wave (lane i takes pixel i of the entry's 2 x kSumRows pixels),
// so that a group costs two memory round trips instead of 32 dependent ones.
__global__ __launch_bounds__(256) void sparse_select_kernel(const float* __restrict__ color, const LevelTab tab, const SumTab st,
                                                            const float* __restrict__ sum, int n_frames, float one_minus_p,
                                                            float p_f, const unsigned* __restrict__ mm, const RegionTab rt,
                                                            unsigned* __restrict__ cells, Candidate* __restrict__ cand,
                                                            int* __restrict__ cand_n) {
    static_assert(2 * kSumRows <= 64, "one lane per pixel of a group");
    const int frame = blockIdx.y;
    const int e = (int)(blockIdx.x * 256 + threadIdx.x);   // (entries per frame < 2^31: host-checked)
    const int lane = threadIdx.x & 63;
    const bool live = e < (int)st.frame_entries;
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < tab.n_levels && e >= (int)st.off[i]) l = i;
    const unsigned* slot = mm + ((long long)frame * tab.n_levels + l) * 2;
    const float thr = __fadd_rn(__fmul_rn(one_minus_p, level_max(slot)), __fmul_rn(p_f, level_min(slot)));
    const float s = live ? sum[(long long)frame * st.frame_entries + e] : kPoolLowest;
    // (groups below the last image row hold no data: whatever is there, their row range is empty)
    unsigned long long todo = __ballot(live && s >= thr);
    while (todo) {   // wave-uniform
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int gl = __builtin_amdgcn_readlane(l, src);
        const float gthr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(thr), src));
        const int rem = __builtin_amdgcn_readlane(e, src) - (int)st.off[gl];
        const int H = tab.h[gl], W = tab.w[gl], nxp = (W + 1) >> 1;
        const int gy = rem / nxp, xp = rem - gy * nxp;
        const int ty = gy / st.gpt, k = gy - ty * st.gpt;
        const int r0 = ty * st.th + k * kSumRows;
        const int r1 = min(min(r0 + kSumRows, (ty + 1) * st.th), H);
        const float* __restrict__ lev = color + ((long long)frame * tab.frame_px + tab.px_off[gl]) * 3;
        const int y = r0 + (lane >> 1), x = 2 * xp + (lane & 1);
        float tc_[3];
        bool passer = false;
        sparse_top(lev, H, W, min(y, H - 1), x, gthr, tc_, &passer);
        passer = passer && lane < 2 * kSumRows && y < r1;
        if (!passer) continue;   // (no wave-level operation below)
        float mx[3] = {kPoolLowest, kPoolLowest, kPoolLowest};
        float t[9][3];
#pragma unroll
        for (int j = 0; j < 9; ++j) sparse_top(lev, H, W, y + j / 3 - 1, x + j % 3 - 1, gthr, t[j]);
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int c = 0; c < 3; ++c) mx[c] = pool_max(mx[c], t[j][c]);   // NaN taps are ignored like out-of-level ones
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = __fmul_rn(tc_[c], tc_[c] == mx[c] ? tc_[c] : 0.0f);
        const float pv = __fmul_rn(__fadd_rn(__fadd_rn(o[0], o[1]), o[2]), 1.0f / 3.0f);
        const RegionLevel& rl = rt.lv[gl];
        int rs = 0, cs = 0;
        for (int s_ = 1; s_ < rl.nrs; ++s_) rs = y >= rl.rcut[s_] ? s_ : rs;
        for (int s_ = 1; s_ < rl.ncs; ++s_) cs = x >= rl.ccut[s_] ? s_ : cs;
        atomicMax(cells + ((long long)frame * tab.n_levels + gl) * kCells + rs * kMaxSeg + cs, f2ord(pool_max(kPoolLowest, pv)));
        const int n = atomicAdd(cand_n + frame, 1);
        if (n < kCandCap) cand[(long long)frame * kCandCap + n] = Candidate{gl, y, x, pv};
    }
}

// one block per frame: the mode of every level (dense_flags[frame][level]), before the dense kernels.  The frame's cell maxima come
// in with one round trip (a thread that walks its windows' cells in memory pays a dependent load per cell)
__global__ __launch_bounds__(256) void sparse_modes_kernel(const LevelTab tab, const RegionTab rt, const unsigned* __restrict__ cells,
                                                           const int* __restrict__ cand_n, const int* __restrict__ nan_flags,
                                                           int* __restrict__ dense_flags, int map_wanted) {
    __shared__ unsigned s_cells[kMaxLevels * kCells];
    const int f = blockIdx.x, l = threadIdx.x;
    const unsigned* fc = cells + (long long)f * tab.n_levels * kCells;
    for (int i = threadIdx.x; i < tab.n_levels * kCells; i += 256) s_cells[i] = fc[i];
    const bool overflow = cand_n[f] > kCandCap;
    const bool has_nan = l < tab.n_levels && nan_flags[f * tab.n_levels + l] != 0;
    __syncthreads();
    if (l >= tab.n_levels) return;
    const RegionLevel& rl = rt.lv[l];
    const unsigned* c = s_cells + l * kCells;
    bool all_pos = true;
    for (int j = 0; j < rl.oh; ++j)
        for (int i = 0; i < rl.ow; ++i) {
            unsigned m = pool_lowest_ord();
            for (int rs = rl.wy_lo[j]; rs < rl.wy_hi[j]; ++rs)
                for (int cs = rl.wx_lo[i]; cs < rl.wx_hi[i]; ++cs) m = max(m, c[rs * kMaxSeg + cs]);
            all_pos = all_pos && ord2f(m) > 0.0f;
        }
    int mode = kTailDense;
    // map_wanted: the caller takes the peak-value MAP too.  Without NaNs in the level it is zeros + the candidates' values
    // (sparse_fill_map_kernel, sparse_finish_kernel); a NaN pixel's peak value is a NaN, which only the dense pass can place
    if (!overflow) mode = (map_wanted && has_nan) ? kTailDense : (all_pos ? kTailSparse : (has_nan ? kTailDense : kTailZero));
    dense_flags[f * tab.n_levels + l] = mode;
}

// zero-fill of the caller's peak-value map on the (frame, level)s the dense pass does not write (`tab`: kKpChunk pixels per block)
__global__ __launch_bounds__(256) void sparse_fill_map_kernel(const LevelTab tab, const int* __restrict__ dense_flags,
                                                              float* __restrict__ pv_map) {
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    if (dense_flags[tc.frame * tab.n_levels + tc.level] == kTailDense) return;
    const int npx = tab.h[tc.level] * tab.w[tc.level];
    float* __restrict__ dst = pv_map + (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
#pragma unroll
    for (int k = 0; k < kKpPer; ++k) {
        const int p = tc.tx * kKpChunk + k * 256 + threadIdx.x;
        if (p < npx) dst[p] = 0.0f;
    }
}

// a load that sees what other workgroups' atomics have left (it bypasses this CU's L1)
__device__ __forceinline__ int coherent_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// exclusive scan of the chunk counts of frame f by one block.  COHERENT: the counts hold this block's own atomics (and a cache line
// may be shared with the neighbouring frame's counts, which another block on this CU may have read before): read around the L1
template <bool COHERENT>
__device__ __forceinline__ void scan_frame_chunks(const int* __restrict__ chunk_counts, long long* __restrict__ chunk_offsets,
                                                  int chunks_per_frame, int64_t* __restrict__ counts, int f) {
    __shared__ long long s_part[256];
    const int* cc = chunk_counts + (long long)f * chunks_per_frame;
    long long* off = chunk_offsets + (long long)f * chunks_per_frame;
    const int per = (chunks_per_frame + 255) / 256;
    const int lo = min((int)threadIdx.x * per, chunks_per_frame), hi = min(lo + per, chunks_per_frame);
    auto at = [&](int i) { return COHERENT ? coherent_load(cc + i) : cc[i]; };
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += at(i);
    s_part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long run = 0;
        for (int i = 0; i < 256; ++i) {
            const long long t = s_part[i];
            s_part[i] = run;
            run += t;
        }
        counts[f] = run;
    }
    __syncthreads();
    long long run = s_part[threadIdx.x];
    for (int i = lo; i < hi; ++i) {
        off[i] = run;
        run += at(i);
    }
}

// one block per frame, AFTER the count pass; `tab` is the count / write pass's chunk table (kKpChunk pixels per block): the
// candidates that reach a POSITIVE window maximum become hit bits + chunk counts (in a window without a positive peak the
// count pass has already set every pixel of a kTailZero level); then the scan of the frame's chunk counts (pass 2: the same
// grid, round 4 spent a launch on it)
__global__ __launch_bounds__(256) void sparse_finish_kernel(const LevelTab tab, const RegionTab rt, const unsigned* __restrict__ cells,
                                                            const Candidate* __restrict__ cand, const int* __restrict__ cand_n,
                                                            const int* __restrict__ dense_flags, unsigned long long* __restrict__ hit_masks,
                                                            int* __restrict__ chunk_counts, float* __restrict__ pv_map,
                                                            long long* __restrict__ chunk_offsets, int64_t* __restrict__ counts) {
    __shared__ float s_pooled[kMaxLevels][kMaxWin * kMaxWin];
    __shared__ unsigned s_cells[kMaxLevels * kCells];
    const int f = blockIdx.x;
    const int n = cand_n[f];
    if (n <= kCandCap) {     // block-uniform (more: every level of this frame ran the dense kernels)
        const unsigned* fc = cells + (long long)f * tab.n_levels * kCells;
        for (int i = threadIdx.x; i < tab.n_levels * kCells; i += 256) s_cells[i] = fc[i];   // one round trip for every level
        __syncthreads();
        for (int l = 0; l < tab.n_levels; ++l) pooled_from_cells(s_cells + l * kCells, rt.lv[l], s_pooled[l]);
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            const Candidate c = cand[(long long)f * kCandCap + i];
            if (dense_flags[f * tab.n_levels + c.level] == kTailDense) continue;
            if (pv_map)      // the caller's peak-value map: zero-filled for this level (sparse_fill_map_kernel), the passers' values go in
                pv_map[(long long)f * tab.frame_px + tab.px_off[c.level] + (long long)c.y * tab.w[c.level] + c.x] = c.pv;
            const float thr = region_thr<false>(c.y, c.x, rt.lv[c.level], s_pooled[c.level]);
            if (!(thr > 0.0f && c.pv >= thr)) continue;
            const int p = c.y * tab.w[c.level] + c.x;
            const long long blk = (long long)f * tab.tiles_per_frame + tab.tile_start[c.level] + (p / kKpChunk);
            atomicOr(hit_masks + blk * (kKpChunk / 64) + ((p % kKpChunk) >> 6), 1ull << (p & 63));
            atomicAdd(chunk_counts + blk, 1);
        }
    }
    // This block's atomics have been performed (device-scope atomics act at the memory side and are acknowledged from there) once the
    // counter is at zero; the scan reads the counts with loads that go to the same place.  (No __threadfence(): its release half
    // writes the L2 back, per wave -- in a kernel that every block ends with one it cost 10x the kernel's time.)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    scan_frame_chunks<true>(chunk_counts, chunk_offsets, tab.tiles_per_frame, counts, f);
}

// pass 2: exclusive scan of the chunk counts of one frame (one block per frame)
__global__ __launch_bounds__(256) void region_scan_kernel(const int* __restrict__ chunk_counts,
                                                          long long* __restrict__ chunk_offsets,
                                                          int chunks_per_frame, int64_t* __restrict__ counts) {
    scan_frame_chunks<false>(chunk_counts, chunk_offsets, chunks_per_frame, counts, blockIdx.x);
}

// pass 3: ordered write of (level, y, x, 0) rows: rank = chunk offset + hits of the lower waves + hits of this wave's
// earlier 64-pixel groups + hits of the lower lanes of this group
__global__ __launch_bounds__(256) void region_write_kernel(const LevelTab tab, const unsigned long long* __restrict__ hit_masks,
                                                           const long long* __restrict__ chunk_offsets,
                                                           int64_t* __restrict__ idx, long long cap_per_frame,
                                                           const int* __restrict__ chunk_counts) {
    __shared__ int s_wave[4];
    if (chunk_counts && chunk_counts[blockIdx.x] == 0) return;   // block-uniform: nothing to write in this chunk
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int W = tab.w[tc.level];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seg = tc.tx * kKpChunk + wave * (kKpChunk / 4);
    // the hit bits the count pass left: kKpPer 64-bit masks per wave (wave-uniform loads)
    const unsigned long long* __restrict__ mk = hit_masks + ((long long)blockIdx.x * 4 + wave) * kKpPer;
    unsigned long long hits[kKpPer];
    int n = 0;
#pragma unroll
    for (int k = 0; k < kKpPer; ++k) {
        hits[k] = mk[k];
        n += __popcll(hits[k]);
    }
    if (lane == 0) s_wave[wave] = n;
    __syncthreads();
    if (s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3] == 0) return;   // block-uniform: nothing to write in this chunk
    long long pos = chunk_offsets[blockIdx.x];
    for (int i = 0; i < wave; ++i) pos += s_wave[i];
    int64_t* dst = idx + (long long)tc.frame * cap_per_frame * 4;
    const unsigned long long below = (1ull << lane) - 1ull;
    int y = (seg + lane) / W, x = seg + lane - y * W;
#pragma unroll
    for (int k = 0; k < kKpPer; ++k, advance_yx(y, x, 64, W)) {
        const unsigned long long m = hits[k];
        if ((m >> lane) & 1ull) {
            const long long r = pos + __popcll(m & below);
            if (r < cap_per_frame) {
                dst[r * 4 + 0] = tc.level;
                dst[r * 4 + 1] = y;
                dst[r * 4 + 2] = x;
                dst[r * 4 + 3] = 0;
            }
        }
        pos += __popcll(m);
    }
}

// ---- centroids (reference slam_recognition/util/centroids.py:21-46)
// cells: box sums (window = stride = region, TF SAME geometry) of v, x*v, y*v per cell; the corrected centroid
// (sum / total, 0/0 = NaN like the reference) goes to a workspace, total_pool to the caller.
struct CellTab {
    int rh, rw;
    int oh[kMaxLevels], ow[kMaxLevels];          // cells per level
    int y_first[kMaxLevels], x_first[kMaxLevels];  // first input index of cell 0 (<= 0: SAME padding)
    long long cell_off[kMaxLevels];              // cell offset of level l inside one frame
    long long frame_cells;
    float yscale[kMaxLevels], xscale[kMaxLevels];  // float32 cells / extent: nearest-neighbour resize back
};

// cell (j, i) of one H x W map v: total of the box, and its centroid (0 / 0 = NaN in an empty cell, like the reference)
// (load(y, x): the map's value at a pixel -- a plain array for the per-op kernel, the displayer's tail computes value / 255 and the
// nearest-neighbour resize on the fly there)
template <class Load>
__device__ __forceinline__ void centroid_cell_of(Load load, int H, int W, int y_first, int x_first, int rh, int rw, int j, int i,
                                                 float* tot_out, float* cx, float* cy) {
    const int y0 = max(y_first + j * rh, 0), y1 = min(y_first + j * rh + rh, H);
    const int x0 = max(x_first + i * rw, 0), x1 = min(x_first + i * rw + rw, W);
    float tot = 0.0f, sx = 0.0f, sy = 0.0f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            const float val = load(y, x);
            sx = __fadd_rn(sx, __fmul_rn((float)x, val));   // ind_tens * value, then the box sum (float32)
            sy = __fadd_rn(sy, __fmul_rn((float)y, val));
            const float half = __fmul_rn(val, 0.5f);        // value tiled to 2 channels times the 1/2 tap
            tot = __fadd_rn(__fadd_rn(tot, half), half);
        }
    *tot_out = tot;
    *cx = sx / tot;
    *cy = sy / tot;
}
__device__ __forceinline__ void centroid_cell(const float* __restrict__ v, int H, int W, int y_first, int x_first, int rh, int rw, int j, int i,
                                              float* tot_out, float* cx, float* cy) {
    centroid_cell_of([&](int y, int x) { return v[(long long)y * W + x]; }, H, W, y_first, x_first, rh, rw, j, i, tot_out, cx, cy);
}

__global__ __launch_bounds__(256) void centroid_cells_kernel(const float* __restrict__ value, const LevelTab tab,
                                                             const CellTab ct, int n_frames,
                                                             float* __restrict__ total_out, float* __restrict__ cxy) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int frame = (int)(gid / ct.frame_cells);
    if (frame >= n_frames) return;
    long long rem = gid - (long long)frame * ct.frame_cells;
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < tab.n_levels && rem >= ct.cell_off[i]) l = i;
    rem -= ct.cell_off[l];
    const int j = (int)(rem / ct.ow[l]), i = (int)(rem - (long long)j * ct.ow[l]);
    const float* __restrict__ v = value + (long long)frame * tab.frame_px + tab.px_off[l];
    const long long cell = (long long)frame * ct.frame_cells + ct.cell_off[l] + rem;
    centroid_cell(v, tab.h[l], tab.w[l], ct.y_first[l], ct.x_first[l], ct.rh, ct.rw, j, i, &total_out[cell], &cxy[cell * 2], &cxy[cell * 2 + 1]);
}

// per pixel: |cx(cell) - x| + |cy(cell) - y| with the TF1 nearest-neighbour cell lookup (c: the map's [oh][ow][2] centroids)
__device__ __forceinline__ float centroid_dist_px(const float* __restrict__ c, int y, int x, int oh, int ow, float yscale, float xscale) {
    int sy = (int)floorf(__fmul_rn((float)y, yscale));
    int sx = (int)floorf(__fmul_rn((float)x, xscale));
    sy = min(sy, oh - 1);
    sx = min(sx, ow - 1);
    const float cx = c[((long long)sy * ow + sx) * 2], cy = c[((long long)sy * ow + sx) * 2 + 1];
    return __fadd_rn(fabsf(cx - (float)x), fabsf(cy - (float)y));
}
__global__ __launch_bounds__(256) void centroid_dist_kernel(const LevelTab tab, const CellTab ct,
                                                            const float* __restrict__ cxy, float* __restrict__ out) {
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int l = tc.level, W = tab.w[l];
    const int npx = tab.h[l] * W;
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[l];
    const float* __restrict__ c = cxy + ((long long)tc.frame * ct.frame_cells + ct.cell_off[l]) * 2;
    for (int k = 0; k < 4; ++k) {
        const int p = tc.tx * kChunk + k * 256 + threadIdx.x;
        if (p >= npx) break;
        const int y = p / W, x = p - y * W;
        out[base_px + p] = centroid_dist_px(c, y, x, ct.oh[l], ct.ow[l], ct.yscale[l], ct.xscale[l]);
    }
}

// ---- boosting state (reference slam_recognition/util/energy/boosting.py:10-42)
// step 1: memory_biased_values = input ** energy.  pow in float64 rounded once to float32 (the reference's float32
// pow is libm dependent; this is the correctly rounded value, see oracle/silent_oracle.py boosting_power).
__global__ __launch_bounds__(256) void boost_power_kernel(const float* __restrict__ x, const float* __restrict__ energy,
                                                          float* __restrict__ m, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) m[i] = (float)pow((double)x[i], (double)energy[i]);
}

struct BoostP {
    float lo, hi;             // clip range of the state: [-exhaustion_max, +excitation_max]
    int recovery_mode;        // bit 0 constant, bit 1 input based
    float recovery_amount, recovery_percentage;
    int visualize;            // outputs carry 3 identical channels: fired * x, energy * normer + centerer
    float normer, centerer;
};

// step 2: has_fired = (m == maxpool3x3 SAME(m)); energy <- clip((energy*255 - fired*255 + recovery) / 255, lo, hi)
// in place (a pixel's update reads only its own energy; the neighbours enter through m, which is a separate buffer).
// One pixel p of an H x W map m: returns the new energy; *f_out / *e_out = what the fired / energy maps show.
__device__ __forceinline__ float boost_update_px(const float* __restrict__ src, int H, int W, int p, float xin, float e, const BoostP& bp,
                                                 float* f_out, float* e_out) {
    const int y = p / W, xx0 = p - y * W;
    float mx = kPoolLowest;
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = xx0 + dx;
            if (xx < 0 || xx >= W) continue;
            mx = pool_max(mx, src[(long long)yy * W + xx]);
        }
    }
    const float fired = src[p] == mx ? 1.0f : 0.0f;
    const float strength = __fmul_rn(fired, xin);
    const float r_in = __fmul_rn(strength, bp.recovery_percentage);
    float rec = bp.recovery_amount;
    if (bp.recovery_mode == 2) rec = r_in;
    else if (bp.recovery_mode == 3) rec = r_in < bp.recovery_amount ? bp.recovery_amount : r_in;
    float u = __fadd_rn(__fsub_rn(__fmul_rn(e, 255.0f), __fmul_rn(fired, 255.0f)), rec) / 255.0f;
    u = u < bp.lo ? bp.lo : u;
    u = u > bp.hi ? bp.hi : u;
    *f_out = bp.visualize ? strength : fired;
    *e_out = bp.visualize ? __fadd_rn(__fmul_rn(u, bp.normer), bp.centerer) : u;
    return u;
}
__global__ __launch_bounds__(256) void boost_update_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                           float* __restrict__ energy, float* __restrict__ fired_out,
                                                           float* __restrict__ energy_out, const LevelTab tab,
                                                           const BoostP bp) {
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const float* __restrict__ src = m + base_px;
    const int npx = H * W;
    const int C = bp.visualize ? 3 : 1;
    for (int k = 0; k < 4; ++k) {
        const int p = tc.tx * kChunk + k * 256 + threadIdx.x;
        if (p >= npx) break;
        float f_out, e_out;
        energy[base_px + p] = boost_update_px(src, H, W, p, x[base_px + p], energy[base_px + p], bp, &f_out, &e_out);
        for (int c = 0; c < C; ++c) {
            fired_out[(base_px + p) * C + c] = f_out;
            if (energy_out) energy_out[(base_px + p) * C + c] = e_out;
        }
    }
}

// ---- display-graph glue (reference recognition_testing.py:79-83, :99)
// out = clip(in * mul / div + add, lo, hi) + post_add, each operation rounded to float32 like the TF scalar ops.
struct AffineP {
    float mul, div, add, lo, hi, post_add;
};

__device__ __forceinline__ float affine_px(float v, const AffineP& ap) {
    float y = __fadd_rn(__fdiv_rn(__fmul_rn(v, ap.mul), ap.div), ap.add);
    y = y < ap.lo ? ap.lo : y;
    y = y > ap.hi ? ap.hi : y;
    return __fadd_rn(y, ap.post_add);
}

__global__ __launch_bounds__(256) void affine_clip_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          long long n, const AffineP ap) {
    // 8 elements per thread, 256 apart (coalesced), all requested before the first use
    const long long i0 = (long long)blockIdx.x * 2048 + threadIdx.x;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = in[min(i0 + k * 256, n - 1)];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long i = i0 + k * 256;
        if (i < n) out[i] = affine_px(v[k], ap);
    }
}

// np.asarray(frame, dtype=float32) (recognition_testing.py:141) and the per-colour-plane copies of image_to_zoom_tensor
// (from_image.py:54-64) as one strided cast: out[p * out_stride + out_off + k] = float32(in[p * in_stride + in_off + k]),
// k < count.  Widening uint8 / int / float64 camera frames, cutting a colour plane out of an interleaved image and
// interleaving planes again are all instances.  The conversion is the C cast (exact up to 2^24 in magnitude, round-to-nearest
// beyond for int32 / int64 / float64, like NumPy's astype).
template <class T>
__global__ __launch_bounds__(256) void cast_interleave_kernel(const T* __restrict__ in, float* __restrict__ out, long long n_px,
                                                              int in_stride, int in_off, int count, int out_stride, int out_off) {
    const long long total = n_px * count;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long p = i / count;
        const int k = (int)(i - p * count);
        out[p * out_stride + out_off + k] = (float)in[p * in_stride + in_off + k];
    }
}

// The same cast for a RECTANGLE of an interleaved [H][W][C] frame (rows y0 .. y0 + h, columns x0 .. x0 + w), written to the same
// positions of the float32 frame: the displayer converts only the part of a camera frame its pyramid reads (the union of the
// reference layout's centre crops is about half of the frame -- half of the bytes that cross the link when the frame is read straight
// from pinned host memory).
template <class T>
__global__ __launch_bounds__(256) void cast_rect_kernel(const T* __restrict__ in, float* __restrict__ out, int W, int C, int y0, int x0, int h, int w) {
    const long long row_n = (long long)w * C, total = row_n * h;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / row_n;
        const long long at = ((long long)(y0 + r) * W + x0) * C + (i - r * row_n);
        out[at] = (float)in[at];
    }
}

// tf.image.resize_nearest_neighbor (TF1, align_corners = False): src = min(floor(dst * float32(in / out)), in - 1).
// ``tab`` describes the OUTPUT maps; the input level l has extents (ih[l], iw[l]) at in_off[l] of a frame of in_px.
struct ResizeTab {
    int ih[kMaxLevels], iw[kMaxLevels];
    long long in_off[kMaxLevels];
    long long in_px;
    float yscale[kMaxLevels], xscale[kMaxLevels];
};

__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             const LevelTab tab, const ResizeTab rt, int C) {
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int l = tc.level, W = tab.w[l];
    const int npx = tab.h[l] * W;
    const float* __restrict__ src = in + ((long long)tc.frame * rt.in_px + rt.in_off[l]) * C;
    float* __restrict__ dst = out + ((long long)tc.frame * tab.frame_px + tab.px_off[l]) * C;
    for (int k = 0; k < 4; ++k) {
        const int p = tc.tx * kChunk + k * 256 + threadIdx.x;
        if (p >= npx) break;
        const int y = p / W, x = p - y * W;
        const int sy = min((int)floorf(__fmul_rn((float)y, rt.yscale[l])), rt.ih[l] - 1);
        const int sx = min((int)floorf(__fmul_rn((float)x, rt.xscale[l])), rt.iw[l] - 1);
        for (int c = 0; c < C; ++c) dst[(long long)p * C + c] = src[((long long)sy * rt.iw[l] + sx) * C + c];
    }
}

// ---- the tail of the reference's application graph for ONE frame (silent_displayer_step; recognition_testing.py:77-100) in five
// launches instead of fifteen: every launch of the per-op sequence is at its 4.7 us floor on a 640 x 480 frame, so the graph's
// device time was a third small-kernel floors.  Each kernel below is several of the per-op kernels side by side over ONE index
// space -- the per-element arithmetic is the per-op kernels' own (affine_px, centroid_cell, centroid_dist_px, boost_update_px), so
// the results are bit-identical (test_line_end_displayer_three_frames: native against per-op).
// Geometry: L maps of h x w (the pyramid's levels are the batch), cells of rh x rw -> ch x cw; the nearest-neighbour resized maps
// h2 x w2 -> ch2 x cw2 cells.
struct DispTail {
    int L, h, w, ch, cw, h2, w2, ch2, cw2, rh, rw;
    int y_first, x_first, y_first2, x_first2;      // first input index of cell 0 (TF SAME padding), both maps
    float yscale_c, xscale_c, yscale_c2, xscale_c2; // float32 cells / extent: nearest-neighbour cell lookup, both maps
    float yscale_r, xscale_r;                      // float32 h / h2, w / w2: nearest-neighbour resize
    AffineP by255, imp, inv, x255;
    BoostP bp;
    const float* value;                            // [L, h, w]
    float *g, *im2n;                               // (value / 255; resized value / 255: intermediates of the per-op path, not written here since round 6)
    float *cxy, *cxy2, *tot1;                      // centroids of both maps [cells][2]; cell totals of the first
    float *imp_out, *m;                            // importances; importances ** energy
    float *energy;                                 // [L, ch, cw] state, advanced in place
    float *out1, *out2, *out3, *update;            // 255 - dist * 255 (both maps), fired * 255, update_importances
    // completion signal (NULL: none): disp_signal_kernel, one thread behind the last kernel, bumps *seq and stores it to *flag -- a
    // word in pinned host memory the caller polls instead of paying a stream synchronisation (silent_displayer_step)
    unsigned long long* seq;                       // frames completed (device)
    unsigned long long* flag;                      // = *seq, in pinned host memory
};

// Round 6: TWO launches instead of four (each costs ~5 us on the critical path of a camera frame).
// (1) the centroid cells of both maps with value / 255 and the nearest-neighbour resize computed on the fly (the per-op path's
//     affine_clip + resize_nearest + affine_clip kernels wrote them to memory first: same operations on the same floats), and for
//     the cells of the first map the importances (total_pool * 255 / 4 clipped to [1, 256], - 1) and importances ** energy.
__global__ __launch_bounds__(256) void disp_cells_kernel(const DispTail t) {
    const long long c1 = (long long)t.L * t.ch * t.cw, c2 = (long long)t.L * t.ch2 * t.cw2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid < c1) {
        const int f = (int)(gid / ((long long)t.ch * t.cw)), r = (int)(gid - (long long)f * t.ch * t.cw);
        const float* __restrict__ v = t.value + (long long)f * t.h * t.w;
        float tot;
        centroid_cell_of([&](int y, int x) { return affine_px(v[(long long)y * t.w + x], t.by255); }, t.h, t.w, t.y_first, t.x_first, t.rh, t.rw,
                         r / t.cw, r % t.cw, &tot, &t.cxy[gid * 2], &t.cxy[gid * 2 + 1]);
        t.tot1[gid] = tot;
        const float imp = affine_px(tot, t.imp);
        t.imp_out[gid] = imp;
        t.m[gid] = (float)pow((double)imp, (double)t.energy[gid]);          // boost_power_kernel
    } else if (gid < c1 + c2) {
        const long long q = gid - c1;
        const int f = (int)(q / ((long long)t.ch2 * t.cw2)), r = (int)(q - (long long)f * t.ch2 * t.cw2);
        const float* __restrict__ v = t.value + (long long)f * t.h * t.w;
        float tot;
        centroid_cell_of(
            [&](int y, int x) {
                const int sy = min((int)floorf(__fmul_rn((float)y, t.yscale_r)), t.h - 1);
                const int sx = min((int)floorf(__fmul_rn((float)x, t.xscale_r)), t.w - 1);
                return affine_px(v[(long long)sy * t.w + sx], t.by255);
            },
            t.h2, t.w2, t.y_first2, t.x_first2, t.rh, t.rw, r / t.cw2, r % t.cw2, &tot, &t.cxy2[q * 2], &t.cxy2[q * 2 + 1]);
    }
}

// (2) 255 - dist * 255 of both maps, and -- beside them, it needs nothing of theirs -- boost_update_kernel + fired * 255
//     (C channels: 3 with bp.visualize) on the cells
__global__ __launch_bounds__(256) void disp_dist_boost_kernel(const DispTail t) {
    const long long px = (long long)t.L * t.h * t.w, px2 = (long long)t.L * t.h2 * t.w2, c1 = (long long)t.L * t.ch * t.cw;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid < px) {
        const int f = (int)(gid / ((long long)t.h * t.w)), p = (int)(gid - (long long)f * t.h * t.w);
        const float d = centroid_dist_px(t.cxy + (long long)f * t.ch * t.cw * 2, p / t.w, p % t.w, t.ch, t.cw, t.yscale_c, t.xscale_c);
        t.out1[gid] = affine_px(d, t.inv);
    } else if (gid < px + px2) {
        const long long q = gid - px;
        const int f = (int)(q / ((long long)t.h2 * t.w2)), p = (int)(q - (long long)f * t.h2 * t.w2);
        const float d = centroid_dist_px(t.cxy2 + (long long)f * t.ch2 * t.cw2 * 2, p / t.w2, p % t.w2, t.ch2, t.cw2, t.yscale_c2, t.xscale_c2);
        t.out2[q] = affine_px(d, t.inv);
    } else if (gid < px + px2 + c1) {
        const long long q = gid - px - px2;
        const int f = (int)(q / ((long long)t.ch * t.cw)), p = (int)(q - (long long)f * t.ch * t.cw);
        float f_out, e_out;
        t.energy[q] = boost_update_px(t.m + (long long)f * t.ch * t.cw, t.ch, t.cw, p, t.imp_out[q], t.energy[q], t.bp, &f_out, &e_out);
        const int C = t.bp.visualize ? 3 : 1;
        const float shown = affine_px(f_out, t.x255);
        for (int c = 0; c < C; ++c) {
            t.out3[q * C + c] = shown;
            t.update[q * C + c] = e_out;
        }
    }
}

// The frame's completion word.  A kernel of its own BEHIND the last one: the kernels in front have ended, their stores (some of them
// into pinned host memory) are visible system-wide.  (Counting finished blocks inside the last kernel, with a system-scope fence per
// block, made that kernel wait for the link 700 times: + 60 us per frame, measured.)
__global__ __launch_bounds__(64) void disp_signal_kernel(unsigned long long* seq, unsigned long long* flag) {
    if (threadIdx.x == 0) {
        const unsigned long long n = *seq + 1;
        *seq = n;
        __hip_atomic_store(flag, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace silent
