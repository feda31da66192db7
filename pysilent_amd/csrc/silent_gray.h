// Grayscale kernels (BASELINE configs 1 / 2 / 5): the streaming CS -> line-end stencil, the unit level fused with its pyramid
// smoother, and the single-read stream kernels that produce the whole pyramid from one read of the frame.
#pragma once

#include "silent_common.h"

namespace silent {

// ---------------------------------------------------------------------------------------------
// Fused grayscale pass (BASELINE configs 1/2/5):  cs = relu(conv3x3(x, cs_k));
//                                                 end = clip(relu(conv3x3(cs, end_bank)), 0, hi)
//
// Wave-autonomous streaming stencil: no LDS, no barrier.  A wave owns a strip of 64 columns (60 of
// them produce outputs) and walks down R rows.  Each lane loads its own column (one coalesced dword
// per row, all R+4 rows requested up front), receives the left / right neighbours by a DPP wave shift,
// keeps a 3x3 input window and a 3x3 CS window in registers, and emits one CS value and one K-vector
// per row.  Lanes 0/63 only feed their neighbours' CS, lanes 1/62 only feed their neighbours' outputs,
// so lanes 2..61 own the 60 output columns.  CS values outside the level are forced to 0 because the
// second convolution's SAME padding pads the CS MAP, not the input.  The dominant traffic (K floats
// per pixel, NHWC) leaves as one contiguous 60 x 4K-byte run per wave instruction.
struct GrayW {
    float cs[9];
    float end[9 * 8];  // [dy][dx][k], k < K
};

constexpr int kGrayCols = 60;            // output columns per wave
constexpr int kGrayTW = 4 * kGrayCols;   // 4 waves side by side
constexpr int kGrayTH = 16;              // rows per tile (R)

// K = 8: a pixel's 8 floats are 32 bytes, so "one float4 pair per lane" makes every store instruction write
// 16-byte pieces at a 32-byte stride (measured 3.1 TB/s vs 5.2 for K = 4).  Instead the wave transposes the
// row through a wave-private LDS slab: lane l then stores the l-th 16-byte piece of the row, so that each of
// the two store instructions covers 1 KiB of contiguous memory.  first / count: the lanes (= pixels of the
// wave's 64 columns) that may be written.  Every lane of the wave must call this (LDS exchange).
// NT: non-temporal stores (gray_stream_kernel: the map is a result nobody on the GPU reads back, and keeping it out of
// the caches leaves the Infinity Cache to the pyramid levels that gray_line_end_kernel reads next).
template <bool NT = false>
__device__ __forceinline__ void store_row_k8(float* __restrict__ row_base /* address of pixel of lane 0 */,
                                             const float (&acc)[8], float* s_slab /* 512 floats, wave private */,
                                             int lane, int first, int count) {
    typedef float nf4 __attribute__((ext_vector_type(4)));
    nf4* slab4 = reinterpret_cast<nf4*>(s_slab);
    slab4[lane * 2 + 0] = nf4{acc[0], acc[1], acc[2], acc[3]};
    slab4[lane * 2 + 1] = nf4{acc[4], acc[5], acc[6], acc[7]};
    __builtin_amdgcn_wave_barrier();
    const nf4 a = slab4[lane], b = slab4[64 + lane];
    __builtin_amdgcn_wave_barrier();
    // piece q (16 bytes) belongs to pixel q / 2
    const int pa = lane >> 1, pb = 32 + (lane >> 1);
    nf4* out4 = reinterpret_cast<nf4*>(row_base);
    if constexpr (NT) {
        if (pa >= first && pa < first + count) __builtin_nontemporal_store(a, out4 + lane);
        if (pb >= first && pb < first + count) __builtin_nontemporal_store(b, out4 + 64 + lane);
    } else {
        if (pa >= first && pa < first + count) out4[lane] = a;
        if (pb >= first && pb < first + count) out4[64 + lane] = b;
    }
}

template <int K, int R>
__global__ __launch_bounds__(256) void gray_line_end_kernel(const float* __restrict__ pyr,
                                                            float* __restrict__ cs_out,
                                                            float* __restrict__ end_out, const LevelTab tab,
                                                            const GrayW wts, float clip_hi, unsigned opts) {
    __shared__ __attribute__((aligned(16))) float s_slab[K == 8 ? 4 * 512 : 4];  // K = 8 store transpose, per wave
    const TileCoord tc = locate_tile(tab, (opts & 1u) ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x);
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const float* __restrict__ src = pyr + base_px;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = tc.tx * kGrayTW + wave * kGrayCols;  // first output column of this wave
    if (xw0 >= W) return;                                // wave-uniform
    const int y0 = tc.ty * R;
    const int x = xw0 + lane - 2;
    const bool col_ok = x >= 0 && x < W;
    const int xc = min(max(x, 0), W - 1);

    // all R+4 input rows of this lane's column, requested before the first use (clamped address +
    // select: no branch around a load)
    float in[R + 4];
#pragma unroll
    for (int i = 0; i < R + 4; ++i) {
        const int y = y0 - 2 + i;
        const bool row_ok = y >= 0 && y < H;
        const float v = src[(long long)min(max(y, 0), H - 1) * W + xc];
        in[i] = (row_ok && col_ok) ? v : 0.0f;
    }
    // Retire every requested row here.  Loads and stores share vmcnt in issue order, so a counted wait
    // for a late row inside the loop below would also wait for the previous rows' STORES to be
    // acknowledged; with the loads retired up front the row loop carries no vector-memory wait at all.
#pragma unroll
    for (int i = 0; i < R + 4; ++i) asm volatile("" ::"v"(in[i]));

    float iw[3][3], cw[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) iw[a][b] = cw[a][b] = 0.0f;
    const bool out_lane = lane >= 2 && lane < 2 + kGrayCols && x < W;

#pragma unroll
    for (int i = 0; i < R + 4; ++i) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            iw[0][b] = iw[1][b];
            iw[1][b] = iw[2][b];
        }
        iw[2][1] = in[i];
        iw[2][0] = from_lane_below(in[i]);
        iw[2][2] = from_lane_above(in[i]);
        if (i >= 2) {
            const int cy = y0 + i - 3;  // CS row produced by this step
            float acc = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc = __builtin_fmaf(iw[dy][dx], wts.cs[dy * 3 + dx], acc);
            float cs = relu_tf(acc);
            cs = (cy >= 0 && cy < H && col_ok) ? cs : 0.0f;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                cw[0][b] = cw[1][b];
                cw[1][b] = cw[2][b];
            }
            cw[2][1] = cs;
            cw[2][0] = from_lane_below(cs);
            cw[2][2] = from_lane_above(cs);
        }
        if (i >= 4) {
            const int y = y0 + i - 4;  // output row
            if (y < H) {               // wave-uniform
                const long long px = base_px + (long long)y * W + x;
                if (cs_out && out_lane) {
                    cs_out[px] = cw[1][1];  // (non-temporal here: within noise, unlike in gray_stream_kernel)
                }
                if (end_out) {
                    float acc[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) acc[k] = 0.0f;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                            for (int k = 0; k < K; ++k)
                                acc[k] = __builtin_fmaf(cw[dy][dx], wts.end[(dy * 3 + dx) * K + k], acc[k]);
                    relu_clip_tf(acc, clip_hi);
                    if constexpr (K == 8) {
                        // lane 0's pixel is column xw0 - 2; valid pixels are lanes 2 .. 2 + ncols
                        const int ncols = min(kGrayCols, W - xw0);
                        store_row_k8(end_out + (base_px + (long long)y * W + (xw0 - 2)) * 8, acc, s_slab + wave * 512, lane,
                                     2, ncols);
                    } else if (out_lane) {
                        float* __restrict__ po = end_out + px * K;
                        if constexpr (K == 4) {
                            typedef float nf4 __attribute__((ext_vector_type(4)));
                            const nf4 v4 = {acc[0], acc[1], acc[2], acc[3]};
                            *reinterpret_cast<nf4*>(po) = v4;
                        } else {
#pragma unroll
                            for (int k = 0; k < K; ++k) po[k] = acc[k];
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Unit-zoom level, pyramid AND filters in one pass (the whole hot path for level 0, 75 % of all pixels):
//     p   = smooth5x5(frame)               -> pyramid level   (scipy zoom factor 1 = [1,26,66,26,1]/120 per axis)
//     cs  = relu(conv3x3(p, cs_k))         -> CS map
//     end = clip(relu(conv3x3(cs, bank)))  -> K-orientation line-end maps
// Same wave-autonomous streaming structure as gray_line_end_kernel, with the 5-tap smoother in front: the
// level is never re-read from HBM by the filter pass.  Each stage costs halo lanes (2 + 1 + 1 per side):
// 56 of the 64 lanes produce outputs; a tile is 224 columns x 16 rows and streams 24 source rows.
#ifndef SILENT_FUSED_COLS
#define SILENT_FUSED_COLS 56
#endif
constexpr int kFusedCols = SILENT_FUSED_COLS;
static_assert(kFusedCols >= 16 && kFusedCols <= 56 && kFusedCols % 4 == 0, "64 lanes = output columns + 4 halo lanes per side");
#ifndef SILENT_FUSED_WAVES
#define SILENT_FUSED_WAVES 4
#endif
// waves side by side in a block of the fused / stream kernels (autonomous: no barrier).  4 x 56 columns = 896 bytes = 7 whole
// 128-byte lines per tile row; measured on config 2 (gray_stream_kernel alone): 1 wave 1.06 ms, 2 waves 0.93, 4 waves 0.85,
// 5 waves (1120 bytes: tile edges off the line grid) 1.03, 8 waves 0.89
constexpr int kFusedWaves = SILENT_FUSED_WAVES;
constexpr int kFusedTW = kFusedWaves * kFusedCols;
#ifndef SILENT_FUSED_TH
#define SILENT_FUSED_TH 16
#endif
// output rows per tile of the fused / stream kernels; a tile streams TH + 8 source rows.  Round 4 tried 24 / 32 / 40 / 48 rows
// (32: the stream kernel fetches 40 rows per 32 produced instead of 24 per 16 -- read amplification 1.43x instead of 1.79x with
// the column halo -- at 40 KB instead of 24 KB of LDS per block, 4 instead of 6 waves / SIMD): in bursts of 5 steps on a fast box
// 32 rows won 1.9 % of the config-2 step (profiles/r04/evidence/ab_gray_tile_height.txt), through bench.py's settled 30-step
// windows on a slow box they LOST 8 % (step 1.154 against 1.066 ms, kernel 0.944 against 0.852; 40 rows the same, 48 rows 16 %;
// config 5: +1 % / +9 % / +10 %; profiles/r04/evidence/ab_gray_tile_height_bench.txt).  16 stays: the contract number is the
// settled one, and most boxes of the pool are of the slow kind.
constexpr int kFusedTH = SILENT_FUSED_TH;

struct FusedLevel {  // the unit levels of a pyramid plan, as the kernel needs them
    int src_y0, src_x0, src_h, src_w;  // crop of the frame (mirror extension happens inside the crop)
    int zoom_h, zoom_w, out_h, out_w;  // resampler extents (== crop) and canvas extents
    int tiles_x, tile_start;
    long long px_off;                  // pixel offset of the level inside one pyramid
};

struct FusedTab {
    int n, tiles_per_frame, H, W;
    long long frame_px;                // pixels of one whole pyramid (all levels)
    float wx[6], wy[6];                // scipy's SIX taps of a zoom-1 level: [1, 26, 66, 26, 1] / 120 and 2^-53 (unit_taps6)
    FusedLevel lv[kMaxLevels];
};

template <int K, int R>
__global__ __launch_bounds__(64 * kFusedWaves) void gray_unit_fused_kernel(const float* __restrict__ frames,
                                                              float* __restrict__ pyr, float* __restrict__ cs_out,
                                                              float* __restrict__ end_out, const FusedTab tab,
                                                              const GrayW wts, float clip_hi) {
    __shared__ __attribute__((aligned(16))) float s_slab[K == 8 ? kFusedWaves * 512 : 4];  // K = 8 store transpose, per wave
    const unsigned bid = blockIdx.x;
    const int frame = (int)(bid / (unsigned)tab.tiles_per_frame);
    int rem = (int)(bid - (unsigned)frame * (unsigned)tab.tiles_per_frame);
    int li = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < tab.n && rem >= tab.lv[i].tile_start) li = i;
    const FusedLevel& lv = tab.lv[li];
    rem -= lv.tile_start;
    const int ty = rem / lv.tiles_x, tx = rem - ty * lv.tiles_x;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = tx * kFusedTW + wave * kFusedCols;
    if (xw0 >= lv.out_w) return;  // wave-uniform
    const int y0 = ty * R;
    const int ox = xw0 + lane - 4;
    const int W = tab.W;
    const float* __restrict__ src = frames + (long long)frame * tab.H * W;
    const long long base_px = (long long)frame * tab.frame_px + lv.px_off;

    // scipy 'mirror' inside the crop, then the crop's offset in the frame
    const long long sx = mirror_near(ox, lv.src_w) + lv.src_x0;
    float in[R + 9];                                                  // stream rows y0 - 4 .. y0 + R + 4 (the sixth tap: + 1 row)
#pragma unroll
    for (int i = 0; i < R + 9; ++i)
        in[i] = src[(long long)(mirror_near(y0 - 4 + i, lv.src_h) + lv.src_y0) * W + sx];
    float xcol = unit_edge_column(src, W, xw0 + 60, lv.src_w, lv.src_x0, y0 - 4, R + 9, lv.src_h, lv.src_y0, lane);
#pragma unroll
    for (int i = 0; i < R + 9; ++i) asm volatile("" ::"v"(in[i]));  // retire loads before the first store
    asm volatile("" : "+v"(xcol));

    const bool col_in = ox >= 0 && ox < lv.out_w;                  // inside the level (SAME padding is 0 outside)
    const bool out_lane = lane >= 4 && lane < 4 + kFusedCols && ox < lv.out_w;
    float hw[6] = {0, 0, 0, 0, 0, 0};
    float iw[3][3], cw[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) iw[a][b] = cw[a][b] = 0.0f;

#pragma unroll
    for (int i = 0; i < R + 9; ++i) {
        // ---- horizontal taps of source row y0 - 4 + i
        {
            const float h = unit_taps6(in[i], unit_edge(xcol, i), tab.wx);
#pragma unroll
            for (int j = 0; j < 5; ++j) hw[j] = hw[j + 1];
            hw[5] = h;
        }
        if (i >= 5) {
            // ---- pyramid row p = y0 + i - 7 (vertical taps p - 2 .. p + 3)
            const int p = y0 + i - 7;
            float v = tab.wy[0] * hw[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) v = __builtin_fmaf(tab.wy[j], hw[j], v);
            v = (p < lv.zoom_h && ox < lv.zoom_w) ? v : 0.0f;                 // canvas beyond the zoomed crop
            if (p >= y0 && p < y0 + R && p < lv.out_h && out_lane) pyr[base_px + (long long)p * lv.out_w + ox] = v;
            v = (p >= 0 && p < lv.out_h && col_in) ? v : 0.0f;                // SAME zero padding of the first conv
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                iw[0][b] = iw[1][b];
                iw[1][b] = iw[2][b];
            }
            iw[2][1] = v;
            iw[2][0] = from_lane_below(v);
            iw[2][2] = from_lane_above(v);
        }
        if (i >= 7) {
            // ---- CS row c = y0 + i - 8
            const int c = y0 + i - 8;
            float acc = 0.0f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) acc = __builtin_fmaf(iw[dy][dx], wts.cs[dy * 3 + dx], acc);
            float cs = relu_tf(acc);
            cs = (c >= 0 && c < lv.out_h && col_in) ? cs : 0.0f;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                cw[0][b] = cw[1][b];
                cw[1][b] = cw[2][b];
            }
            cw[2][1] = cs;
            cw[2][0] = from_lane_below(cs);
            cw[2][2] = from_lane_above(cs);
        }
        if (i >= 9) {
            // ---- output row y = y0 + i - 9
            const int y = y0 + i - 9;
            if (y < lv.out_h) {  // wave-uniform
                const long long px = base_px + (long long)y * lv.out_w + ox;
                if (cs_out && out_lane) cs_out[px] = cw[1][1];  // (non-temporal here: 6 % SLOWER, unlike in gray_stream_kernel)
                if (end_out) {
                    float acc[K];
#pragma unroll
                    for (int k = 0; k < K; ++k) acc[k] = 0.0f;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                            for (int k = 0; k < K; ++k)
                                acc[k] = __builtin_fmaf(cw[dy][dx], wts.end[(dy * 3 + dx) * K + k], acc[k]);
                    relu_clip_tf(acc, clip_hi);
                    if constexpr (K == 8) {
                        const int ncols = min(kFusedCols, lv.out_w - xw0);
                        store_row_k8(end_out + (base_px + (long long)y * lv.out_w + (xw0 - 4)) * 8, acc,
                                     s_slab + wave * 512, lane, 4, ncols);
                    } else if (out_lane) {
                        float* __restrict__ po = end_out + px * K;
                        if constexpr (K == 4) {
                            *reinterpret_cast<float4*>(po) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                        } else {
#pragma unroll
                            for (int k = 0; k < K; ++k) po[k] = acc[k];
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gray_stream_kernel: the fused unit-level kernel above, PLUS the pyramid of every other level, from the
// same single stream of frame rows (the frame is read once for the whole pyramid).
//
// Eligible plans (host-checked): one unit level, and every other level resamples the SAME crop with a
// step of at least 1.4 source pixels per output pixel (classic whole-frame pyramids; the reference layout, whose
// levels are different crops, keeps the region kernel).
//
// Pass 1 (registers, fully unrolled) is exactly gray_unit_fused_kernel; on the way the wave also drops its
// 24 streamed rows into a wave-private LDS slab.  Pass 2 is a compact ROLLED loop over those rows for the
// other levels.  Vertical pass of level g: the lane holds its own column, so it keeps up to 4 output rows of
// level g in flight (slot = output row mod 4) and adds w * value per row.  Which slot gets which weight at
// which stream row, which slot restarts and which one completes is a tiny per-tile ROW PROGRAM built by the
// host from the same float64 tap tables (8 dwords per level and stream row), staged into LDS once per block.
// Horizontal pass of a completed row: the outputs anchored in the wave's 56 columns gather their 6 taps from
// the other lanes with ds_bpermute (lane = tap position - first column of the wave) and store one coalesced
// run.  Arithmetic order = region kernel's (vertical then horizontal, taps ascending): bit-identical output.
constexpr int kStreamSlots = 5;   // (the most of any layout)
// Output rows of general level number g that can be in flight at once (their vertical taps overlap), host-checked per plan.
// Layout 0: 4, 3, 2, 2, then 1 -- enough for zoom ladders of ratio >= e^0.5 (the reference's default) and 2.  Layout 1 (round 5,
// "dense" ladders down to a ratio of 1.4, e.g. sqrt 2): 5, 4, 3, 2, 2, 1, 1, always instantiated for 7 levels.  The host tries 0 first.
__host__ __device__ constexpr int stream_slots(int layout, int g) {
    return layout == 0 ? (g == 0 ? 4 : (g == 1 ? 3 : (g <= 3 ? 2 : 1))) : (g == 0 ? 5 : (g == 1 ? 4 : (g == 2 ? 3 : (g <= 4 ? 2 : 1))));
}
constexpr int kStreamRows = kFusedTH + 8;
// Row program, one record per stream row of a tile row, padded to the kernel's template G (4 or 7 levels):
//   [meta(0) .. meta(Gp-1)] [weights of level 0 (4)] [level 1 (3)] [level 2 (2)] ... ; kStreamProgRow(layout, Gp) dwords.
// It is wave-uniform data: the kernel reads it with scalar loads, one record ahead of the row it is working on.
__host__ __device__ constexpr int stream_pad_levels(int g) { return g <= 4 ? 4 : 7; }
__host__ __device__ constexpr int stream_w_off(int layout, int gp, int g) {
    int o = gp;
    for (int h = 0; h < g; ++h) o += stream_slots(layout, h);
    return o;
}
__host__ __device__ constexpr int kStreamProgRow(int layout, int gp) { return (stream_w_off(layout, gp, gp) + 3) / 4 * 4; }   // layout 0: 16 / 24
// meta: bits 0.. "slot restarts", 3 bits completing slot (7 = none), bit 7 "row feeds this level", then the output row
__host__ __device__ constexpr int stream_done_shift(int layout) { return layout == 0 ? 4 : 8; }
__host__ __device__ constexpr int stream_row_shift(int layout) { return layout == 0 ? 8 : 12; }

struct StreamTab {
    int G;                        // general levels handled here (<= template G; extra ones are inert)
    int tiles_y, waves_x;         // tile rows of the unit level, 56-column wave tiles per row
    const int* row_prog;          // [tiles_y][kStreamRows][kStreamProgRow(Gp)], Gp = stream_pad_levels(G)
    const int* col_hdr;           // [G][waves_x][2]: first output column, number of outputs
    const int* col_rec;           // [G][waves_x][64][8]: lane of tap 0, 6 weight bits, pad
    long long px_off[8];          // pixel offset of level g inside one pyramid
    int out_w[8];
};

// A lane's column record of general level g: where its run of outputs starts, how many outputs the wave has, and for output
// `lane` of the run the lane that holds tap 0 (as a ds_bpermute byte index) + the 6 horizontal weights.
struct StreamCol {
    int x0, n, lane4;
    float w[6];
};
__device__ __forceinline__ StreamCol stream_col(const StreamTab& st, int g, int wx_tile, int lane) {
    const int gg = min(g, st.G - 1);
    const int* __restrict__ h = st.col_hdr + ((long long)gg * st.waves_x + wx_tile) * 2;
    const int4* __restrict__ rec = reinterpret_cast<const int4*>(st.col_rec + (((long long)gg * st.waves_x + wx_tile) * 64 + lane) * 8);
    const int4 a = rec[0], b = rec[1];
    StreamCol c;
    c.x0 = h[0];
    c.n = g < st.G ? h[1] : 0;
    c.lane4 = a.x * 4;
    c.w[0] = __int_as_float(a.y);
    c.w[1] = __int_as_float(a.z);
    c.w[2] = __int_as_float(a.w);
    c.w[3] = __int_as_float(b.x);
    c.w[4] = __int_as_float(b.y);
    c.w[5] = __int_as_float(b.z);
    return c;
}
// The first kStreamRegLevels general levels keep their column records in registers for the whole tile (a row of level 0 completes
// every other stream row); the smaller levels -- one completed row per 2, 4, 8 tiles -- fetch theirs when a row completes.  With all
// seven resident the <K, 7> instantiations sat at 125 VGPRs = 4 waves / SIMD (63 registers of records); now they run at the <K, 4>
// instantiations' 5 - 6 waves (profiles/r05_config5/README.md).
#ifndef SILENT_STREAM_REG_LEVELS
#define SILENT_STREAM_REG_LEVELS 3
#endif
constexpr int kStreamRegLevels = SILENT_STREAM_REG_LEVELS;

template <int K, int G, int L = 0>
__global__ __launch_bounds__(64 * kFusedWaves) void gray_stream_kernel(const float* __restrict__ frames, float* __restrict__ pyr,
                                                          float* __restrict__ cs_out, float* __restrict__ end_out,
                                                          const FusedTab tab, const StreamTab st, const GrayW wts,
                                                          float clip_hi, unsigned opts) {
    constexpr int R = kFusedTH, NR = kStreamRows;
    __shared__ __attribute__((aligned(16))) float s_slab[K == 8 ? kFusedWaves * 512 : 4];
    __shared__ __attribute__((aligned(16))) float s_rows[kFusedWaves][NR][64];  // the streamed rows of each wave (wave private)
    const unsigned bid = (opts & 1u) ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
    const int frame = (int)(bid / (unsigned)tab.tiles_per_frame);
    const int rem = (int)(bid - (unsigned)frame * (unsigned)tab.tiles_per_frame);
    const FusedLevel& lv = tab.lv[0];  // the one unit level
    const int ty = rem / lv.tiles_x, tx = rem - ty * lv.tiles_x;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = tx * kFusedTW + wave * kFusedCols;
    const bool live = xw0 < lv.out_w;  // wave-uniform: this wave has columns of the level
    const int wx_tile = tx * kFusedWaves + wave;
    const int y0 = ty * R;
    const int ox = xw0 + lane - 4;
    const int W = tab.W;
    const float* __restrict__ src = frames + (long long)frame * tab.H * W;
    const long long frame_px0 = (long long)frame * tab.frame_px;
    const long long base_px = frame_px0 + lv.px_off;

    // request the 24 frame rows and the per-level column records of this lane (output j of the wave's run = lane j)
    const long long sx = mirror_near(ox, lv.src_w) + lv.src_x0;
    float in[R + 8];
    float in_last = 0.0f, xcol = 0.0f;   // pass 1's sixth taps: stream row R + 8 and the column right of the wave's 64 (unit_taps6)
    constexpr int GR = G < kStreamRegLevels ? G : kStreamRegLevels;
    StreamCol col[GR > 0 ? GR : 1];
#pragma unroll
    for (int i = 0; i < R + 8; ++i) in[i] = 0.0f;
#pragma unroll
    for (int g = 0; g < GR; ++g) {
        col[g].x0 = col[g].n = col[g].lane4 = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) col[g].w[j] = 0.0f;
    }
    if (live) {
#pragma unroll
        for (int i = 0; i < R + 8; ++i)
            in[i] = src[(long long)(mirror_near(y0 - 4 + i, lv.src_h) + lv.src_y0) * W + sx];
        in_last = src[(long long)(mirror_near(y0 + R + 4, lv.src_h) + lv.src_y0) * W + sx];
        xcol = unit_edge_column(src, W, xw0 + 60, lv.src_w, lv.src_x0, y0 - 4, R + 9, lv.src_h, lv.src_y0, lane);
#pragma unroll
        for (int g = 0; g < GR; ++g) col[g] = stream_col(st, g, wx_tile, lane);
    }
    if (!live) return;
#pragma unroll
    for (int i = 0; i < R + 8; ++i) asm volatile("" ::"v"(in[i]));  // retire loads before the first store
    asm volatile("" : "+v"(in_last), "+v"(xcol));
#pragma unroll
    for (int g = 0; g < GR; ++g) {
        asm volatile("" ::"v"(col[g].lane4));
#pragma unroll
        for (int j = 0; j < 6; ++j) asm volatile("" ::"v"(col[g].w[j]));
    }
#pragma unroll
    for (int i = 0; i < R + 8; ++i) s_rows[wave][i][lane] = in[i];

    // ================= pass 2 (runs first: its registers die before pass 1): every other level of the pyramid,
    // rolled loop over the rows in LDS =================
    float vacc[G][kStreamSlots];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k = 0; k < kStreamSlots; ++k) vacc[g][k] = 0.0f;
    static_assert(G == stream_pad_levels(G), "row programs are padded to 4 or 7 levels");
    constexpr int PR = kStreamProgRow(L, G);
    // constant address space: with a wave-uniform address these are s_load_dwordx16 (no VGPR, no readfirstlane)
    typedef const __attribute__((address_space(4))) int* const_int_ptr;
    const_int_ptr prog = (const_int_ptr)(st.row_prog + (long long)ty * (NR * PR));
    int cur[PR], nxt[PR];
#pragma unroll
    for (int e = 0; e < PR; ++e) cur[e] = prog[e];
    float c0 = s_rows[wave][0][lane];
    const int nr_run = NR;
#pragma unroll 1
    for (int i = 0; i < nr_run; ++i) {
        // record and row of the NEXT step are requested before this step's work
        const int in = min(i + 1, NR - 1);
#pragma unroll
        for (int e = 0; e < PR; ++e) nxt[e] = prog[in * PR + e];
        const float c_next = s_rows[wave][in][lane];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int meta = cur[g];
            if (!(meta & 128)) continue;  // wave-uniform: this stream row carries no tap of level g
#pragma unroll
            for (int k = 0; k < stream_slots(L, g); ++k) {
                const float w = __int_as_float(cur[stream_w_off(L, G, g) + k]);
                const float prev = (meta >> k) & 1 ? 0.0f : vacc[g][k];
                vacc[g][k] = __builtin_fmaf(w, c0, prev);
            }
            const int done = (meta >> stream_done_shift(L)) & 7;
            if (done != 7) {  // wave-uniform: slot `done` holds a finished output row of level g
                const int oy = meta >> stream_row_shift(L);
                float v = vacc[g][0];
#pragma unroll
                for (int k = 1; k < stream_slots(L, g); ++k) v = done == k ? vacc[g][k] : v;
                const int vbits = __float_as_int(v);
                const StreamCol cr = g < GR ? col[g < GR ? g : 0] : stream_col(st, g, wx_tile, lane);   // (g is a constant here)
                float acc = cr.w[0] * __int_as_float(__builtin_amdgcn_ds_bpermute(cr.lane4, vbits));
#pragma unroll
                for (int t = 1; t < 6; ++t)
                    acc = __builtin_fmaf(cr.w[t], __int_as_float(__builtin_amdgcn_ds_bpermute(cr.lane4 + 4 * t, vbits)), acc);
                if (lane < cr.n) pyr[frame_px0 + st.px_off[g] + (long long)oy * st.out_w[g] + cr.x0 + lane] = acc;
            }
        }
#pragma unroll
        for (int e = 0; e < PR; ++e) cur[e] = nxt[e];
        c0 = c_next;
    }

    // ================= pass 1: unit level (pyramid + CS + end), rows back from LDS =================
    // The conv weights live in VGPRs here (K <= 4): a VALU fma whose sources are all VGPRs issues at about twice
    // the rate of one that reads an SGPR once two waves per SIMD are ready (2.7 vs 4.2 cycles per wave
    // instruction, scripts/ubench/valu_rate.hip), and the ~50 SGPRs they would take no longer force reloads of
    // the weights from the kernarg segment in every row.
    constexpr bool VW = K <= 4;
    float wv[6], csw[9], endw[VW ? 9 * K : 1];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        wv[j] = tab.wx[j];  // the unit level's taps are the same on both axes ([1,26,66,26,1]/120 and scipy's sixth, 2^-53)
        if constexpr (VW) asm volatile("" : "+v"(wv[j]));
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        csw[j] = wts.cs[j];
        if constexpr (VW) asm volatile("" : "+v"(csw[j]));
    }
    if constexpr (VW) {
#pragma unroll
        for (int j = 0; j < 9 * K; ++j) {
            endw[j] = wts.end[j];
            asm volatile("" : "+v"(endw[j]));
        }
    }
    const int eff_h = min(lv.zoom_h, lv.out_h), eff_w = min(lv.zoom_w, lv.out_w);
    const bool col_eff = ox >= 0 && ox < eff_w;       // inside the zoomed crop (zero outside it)
    const bool col_in = ox >= 0 && ox < lv.out_w;     // inside the level (zero padding of the convolutions)
    const bool out_lane = lane >= 4 && lane < 4 + kFusedCols && ox < lv.out_w;
    const long long wave_px = base_px + (xw0 - 4);    // + row * out_w + lane: wave-uniform part of every address
    // The two 1-channel maps of the unit level are written with NON-TEMPORAL stores (nt: streamed through L2 without
    // displacing the frame rows that neighbouring tiles re-read): -6...-9 % in alternating A/B.  Measured with it:
    // nt on the 16-byte end-map stores as well +3 %, on those alone 0; nt on the small line-sharing stores of pass 2
    // +18 % (they then reach memory as partial lines); the same nt stores in gray_unit_fused_kernel +6 % (!), in
    // pyramid_unit_kernel and gray_line_end_kernel within noise; a run-time switch between the two kinds of store
    // costs 2.5 % by itself, hence no knob.
    // (Tried and dropped: a 3-instruction relu+clip (v_med3 + NaN select) instead of 4: no measurable change, the kernel
    // is not VALU-bound any more.  Parking 4 finished rows of the 1-channel maps in consumed LDS rows and writing them with one
    // global_store_dwordx4 per 4 rows -- 24 instead of 48 stores per tile -- was 3.5 % slower in an alternating A/B.
    // Round 6, block-wide rows: every wave parks its 56-float pieces of four rows of both maps in consumed LDS rows and, after one
    // barrier per four rows, wave w writes the whole 224-pixel tile row 4 g + w (896 B, 7 whole lines, one 16-byte store per lane).
    // Bit-identical; on the SAME buffers (profiles/r06/evidence/ab_block_rows_same_buffers.txt): config 5 - 8 ... - 13 % on slow
    // placements and + 0.5 ... 1.3 % on fast ones, config 2 + 10 ... 14 % on every placement (the barrier ties four waves that
    // otherwise drift freely).  With the placement tuner choosing the fast relation it only costs: dropped (commit 02244a6 has it).)
    {
        float hw[6] = {0, 0, 0, 0, 0, 0};
        float iw[3][3], cw[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) iw[a][b] = cw[a][b] = 0.0f;
#pragma unroll
        for (int i = 0; i < R + 9; ++i) {
            {
                const float c0 = i < R + 8 ? s_rows[wave][i < R + 8 ? i : 0][lane] : in_last;   // (the last row stayed in a register)
                const float h = unit_taps6(c0, unit_edge(xcol, i), wv);
#pragma unroll
                for (int j = 0; j < 5; ++j) hw[j] = hw[j + 1];
                hw[5] = h;
            }
            if (i >= 5) {
                const int p = y0 + i - 7;
                float v = wv[0] * hw[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) v = __builtin_fmaf(wv[j], hw[j], v);
                v = (p >= 0 && p < eff_h && col_eff) ? v : 0.0f;
                if (p >= y0 && p < y0 + R && p < lv.out_h) {  // wave-uniform
                    float* __restrict__ prow = pyr + (wave_px + (long long)p * lv.out_w);
                    if (out_lane) __builtin_nontemporal_store(v, prow + lane);
                }
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    iw[0][b] = iw[1][b];
                    iw[1][b] = iw[2][b];
                }
                iw[2][1] = v;
                iw[2][0] = from_lane_below(v);
                iw[2][2] = from_lane_above(v);
            }
            if (i >= 7) {
                const int c = y0 + i - 8;
                float acc = 0.0f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) acc = __builtin_fmaf(iw[dy][dx], csw[dy * 3 + dx], acc);
                // relu (a NaN stays a NaN) and the zero padding of the end convolution in one select
                const float cs = (c >= 0 && c < lv.out_h && col_in && !(acc < 0.0f)) ? acc : 0.0f;
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    cw[0][b] = cw[1][b];
                    cw[1][b] = cw[2][b];
                }
                cw[2][1] = cs;
                cw[2][0] = from_lane_below(cs);
                cw[2][2] = from_lane_above(cs);
            }
            if (i >= 9) {
                const int y = y0 + i - 9;
                if (y < lv.out_h) {  // wave-uniform
                    const long long row_px = wave_px + (long long)y * lv.out_w;
                    if (cs_out) {
                        float* __restrict__ crow = cs_out + row_px;
                        if (out_lane) __builtin_nontemporal_store(cw[1][1], crow + lane);
                    }
                    if (end_out) {
                        float acc[K];
#pragma unroll
                        for (int k = 0; k < K; ++k) acc[k] = 0.0f;
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                                for (int k = 0; k < K; ++k) {
                                    const int wi = (dy * 3 + dx) * K + k;
                                    if constexpr (VW) acc[k] = __builtin_fmaf(cw[dy][dx], endw[wi], acc[k]);
                                    else acc[k] = __builtin_fmaf(cw[dy][dx], wts.end[wi], acc[k]);
                                }
                        relu_clip_tf(acc, clip_hi);
                        if constexpr (K == 8) {
                            const int ncols = min(kFusedCols, lv.out_w - xw0);
                            store_row_k8<true>(end_out + row_px * 8, acc, s_slab + wave * 512, lane, 4, ncols);
                        } else if constexpr (K == 4) {
                            // non-temporal like the two 4-byte maps: whole pass -1.6 % (the kernel alone -0.7 %: the rest is
                            // gray_line_end_kernel finding more of the pyramid's levels >= 1 in the Infinity Cache)
                            typedef float nf4 __attribute__((ext_vector_type(4)));
                            nf4* __restrict__ erow = reinterpret_cast<nf4*>(end_out + row_px * 4);
                            if (out_lane) __builtin_nontemporal_store(nf4{acc[0], acc[1], acc[2], acc[3]}, erow + lane);
                        } else {
                            float* __restrict__ po = end_out + row_px * K;
                            if (out_lane) {
#pragma unroll
                                for (int k = 0; k < K; ++k) po[lane * K + k] = acc[k];
                            }
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pyramid_stream_kernel<C, G>: the single-read PYRAMID (no filters) for silent_pyramid: gray_stream_kernel without
// the CS / end stages, looped over the C interleaved channels of the frame.  Lane = pixel column exactly as in the
// gray kernel (same plan tables, same row programs, same column records for every channel); channel ch of a row is
// read and written at a stride of C floats.  Used for C = 1 only: two-step gray pyramids 0.58 -> 0.51 ms per 64 1080p
// frames; with C = 3 the stride-3 loads and partial-line stores made it 1.5 ms against 1.0 ms for unit + region
// kernels on 32 RGB frames (measured, bit-identical either way), so RGB plans are not marked streamable.
template <int C, int G, int L = 0>
__global__ __launch_bounds__(64 * kFusedWaves) void pyramid_stream_kernel(const float* __restrict__ frames, float* __restrict__ pyr,
                                                             const FusedTab tab, const StreamTab st) {
    constexpr int R = kFusedTH, NR = kStreamRows;
    __shared__ float s_rows[kFusedWaves][NR][64];
    const unsigned bid = blockIdx.x;
    const int frame = (int)(bid / (unsigned)tab.tiles_per_frame);
    const int rem = (int)(bid - (unsigned)frame * (unsigned)tab.tiles_per_frame);
    const FusedLevel& lv = tab.lv[0];
    const int ty = rem / lv.tiles_x, tx = rem - ty * lv.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = tx * kFusedTW + wave * kFusedCols;
    if (xw0 >= lv.out_w) return;  // wave-uniform (no barrier in this kernel)
    const int wx_tile = tx * kFusedWaves + wave;
    const int y0 = ty * R;
    const int ox = xw0 + lane - 4;
    const long long WC = (long long)tab.W * C;
    const float* __restrict__ src = frames + (long long)frame * tab.H * WC;
    const long long frame_px0 = (long long)frame * tab.frame_px;
    const long long base_px = frame_px0 + lv.px_off;
    const long long sx = (long long)(mirror_near(ox, lv.src_w) + lv.src_x0) * C;

    constexpr int GR = G < kStreamRegLevels ? G : kStreamRegLevels;
    StreamCol col[GR > 0 ? GR : 1];
#pragma unroll
    for (int g = 0; g < GR; ++g) col[g] = stream_col(st, g, wx_tile, lane);
    const int eff_h = min(lv.zoom_h, lv.out_h), eff_w = min(lv.zoom_w, lv.out_w);
    const bool col_eff = ox >= 0 && ox < eff_w;
    const bool out_lane = lane >= 4 && lane < 4 + kFusedCols && ox < lv.out_w;
    static_assert(G == stream_pad_levels(G), "row programs are padded to 4 or 7 levels");
    constexpr int PR = kStreamProgRow(L, G);
    typedef const __attribute__((address_space(4))) int* const_int_ptr;
    const_int_ptr prog = (const_int_ptr)(st.row_prog + (long long)ty * (NR * PR));

#pragma unroll 1
    for (int ch = 0; ch < C; ++ch) {
        float in[R + 8];
#pragma unroll
        for (int i = 0; i < R + 8; ++i)
            in[i] = src[(long long)(mirror_near(y0 - 4 + i, lv.src_h) + lv.src_y0) * WC + sx + ch];
#pragma unroll
        for (int i = 0; i < R + 8; ++i) asm volatile("" ::"v"(in[i]));  // retire loads before the first store
#pragma unroll
        for (int i = 0; i < R + 8; ++i) s_rows[wave][i][lane] = in[i];

        // ---- pass 2: the other levels
        {
            float vacc[G][kStreamSlots];
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int k = 0; k < kStreamSlots; ++k) vacc[g][k] = 0.0f;
            int cur[PR], nxt[PR];
#pragma unroll
            for (int e = 0; e < PR; ++e) cur[e] = prog[e];
            float c0 = s_rows[wave][0][lane];
#pragma unroll 1
            for (int i = 0; i < NR; ++i) {
                const int inx = min(i + 1, NR - 1);
#pragma unroll
                for (int e = 0; e < PR; ++e) nxt[e] = prog[inx * PR + e];
                const float c_next = s_rows[wave][inx][lane];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int meta = cur[g];
                    if (!(meta & 128)) continue;
#pragma unroll
                    for (int k = 0; k < stream_slots(L, g); ++k) {
                        const float w = __int_as_float(cur[stream_w_off(L, G, g) + k]);
                        const float prev = (meta >> k) & 1 ? 0.0f : vacc[g][k];
                        vacc[g][k] = __builtin_fmaf(w, c0, prev);
                    }
                    const int done = (meta >> stream_done_shift(L)) & 7;
                    if (done != 7) {
                        const int oy = meta >> stream_row_shift(L);
                        float v = vacc[g][0];
#pragma unroll
                        for (int k = 1; k < stream_slots(L, g); ++k) v = done == k ? vacc[g][k] : v;
                        const int vbits = __float_as_int(v);
                        const StreamCol cr = g < GR ? col[g < GR ? g : 0] : stream_col(st, g, wx_tile, lane);   // (g is a constant here)
                        float acc = cr.w[0] * __int_as_float(__builtin_amdgcn_ds_bpermute(cr.lane4, vbits));
#pragma unroll
                        for (int q = 1; q < 6; ++q)
                            acc = __builtin_fmaf(cr.w[q], __int_as_float(__builtin_amdgcn_ds_bpermute(cr.lane4 + 4 * q, vbits)), acc);
                        if (lane < cr.n)
                            pyr[(frame_px0 + st.px_off[g] + (long long)oy * st.out_w[g] + cr.x0 + lane) * C + ch] = acc;
                    }
                }
#pragma unroll
                for (int e = 0; e < PR; ++e) cur[e] = nxt[e];
                c0 = c_next;
            }
        }

        // ---- pass 1: the unit level (same fma order as pyramid_unit_kernel).  The tile's 24 rows and the wave's 4 halo lanes per
        // side hold scipy's sixth taps (row p + 3, column x + 3) of every output without extra loads: no edge column here
        float hw[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < R + 8; ++i) {
            const float h = unit_taps6<false>(s_rows[wave][i][lane], 0.0f, tab.wx);
#pragma unroll
            for (int j = 0; j < 5; ++j) hw[j] = hw[j + 1];
            hw[5] = h;
            if (i >= 7 && i < R + 7) {
                const int p = y0 + i - 7;
                float v = tab.wy[0] * hw[0];
#pragma unroll
                for (int j = 1; j < 6; ++j) v = __builtin_fmaf(tab.wy[j], hw[j], v);
                v = (p < eff_h && col_eff) ? v : 0.0f;
                if (p < lv.out_h && out_lane) pyr[(base_px + (long long)p * lv.out_w + ox) * C + ch] = v;
            }
        }
    }
}

}  // namespace silent
