// Zoom pyramid: un-prefiltered quintic B-spline resampling of a frame crop (scipy.ndimage.zoom
// order=5, prefilter=False semantics; reference call site util/zoom/from_image.py:55-59),
// separable, with the per-axis tap tables built on the host in float64.
//
// The frame is read TWICE whatever the number of levels:
//   pyramid_unit_kernel    levels whose zoom factor is exactly 1 (level 0, 75 % of all pixels).  The
//                          resampler degenerates to the fixed 5-tap smoother [1,26,66,26,1]/120 per axis;
//                          wave-autonomous streaming stencil (DPP neighbour exchange, no LDS, no barrier).
//   pyramid_region_kernel  every other level.  One block per source region (128 x 32 px for 1 channel, 64 x 16 for 3): the region (+ halo)
//                          is staged into LDS once with float4 loads, and each level's outputs ANCHORED in
//                          the region (floor(source coordinate) inside it) are produced from that copy:
//                          6-tap vertical pass into LDS, 6-tap horizontal pass, coalesced store.
//                          (Variants tried on the GPU and rejected, DESIGN.md section 6: region-major tap
//                          records staged in LDS, persistent blocks with register prefetch.)
//   pyramid_zero_kernel    canvas pixels the zoomed crop does not cover (reference: uninitialised; here 0),
//                          launched only when a non-unit level has such pixels.
#pragma once

#include "silent_common.h"

namespace silent {

enum { kPyrUnit = 0, kPyrGeneral = 1 };

// ---- unit kernel geometry
constexpr int kUnitCols = 60;             // output columns per wave (64 lanes - 2 halo lanes each side)
constexpr int kUnitTW = 4 * kUnitCols;    // 4 waves side by side
constexpr int kUnitTH = 16;               // rows per tile

// ---- region kernel geometry
__host__ __device__ constexpr int region_w(int C) { return C == 1 ? 128 : 64; }
// rows per region: the LDS copy of a region bounds the blocks per CU, and the kernel is latency-bound (staging
// loads, LDS round trips, barriers), so 3-channel frames use half-height regions: 22 KB -> 7 blocks per CU
// instead of 3 (measured -19 % on 32 x 1080p RGB; for 1 channel 32 rows stay better, 25 KB -> 6 blocks)
__host__ __device__ constexpr int region_h(int C) { return C == 1 ? 32 : 16; }
constexpr int kRegionHaloL = 4, kRegionHaloR = 4;   // staged columns [X0-4, X0+RW+4): float4 aligned, covers taps -3..+3
constexpr int kRegionHaloT = 3, kRegionHaloB = 3;   // staged rows    [Y0-3, Y0+RH+3)
__host__ __device__ constexpr int region_vr(int C) { return C == 1 ? 8 : 4; }   // output rows per vertical-pass chunk
__host__ __device__ constexpr int region_sw(int C) { return region_w(C) + kRegionHaloL + kRegionHaloR; }
__host__ __device__ constexpr int region_sh(int C) { return region_h(C) + kRegionHaloT + kRegionHaloB; }

struct PyrLevelDev {
    int src_y0, src_x0, src_h, src_w;
    int zoom_h, zoom_w, out_h, out_w;
    int kind;
    int xtab_off, ytab_off;  // entry offsets (output columns / rows) of this level in the tap tables
    int xreg_off, yreg_off;  // entry offsets in the region-start tables (regions_x+1 / regions_y+1 entries)
};

struct PyrTab {
    int n_levels;
    int H, W, C;
    long long frame_px_out;  // pixels of one output pyramid
    PyrLevelDev lv[kMaxLevels];
    long long px_off[kMaxLevels];
    // unit launch: tiles of kUnitTW x kUnitTH canvas pixels over the unit levels only
    int unit_tiles_per_frame;
    int unit_tiles_x[kMaxLevels];
    int unit_tile_start[kMaxLevels + 1];
    // region launch
    int regions_x, regions_y, n_general;
    // zero-fill launch (chunks of 1024 canvas pixels over the general levels that need it)
    int zero_chunks_per_frame;
    int zero_chunk_start[kMaxLevels + 1];
    // device tables (one allocation owned by the plan); idx = the 6 mirrored tap positions relative to the crop
    const int* xidx;   // [cols][6]
    const float* xw;   // [cols][6]
    const int* yidx;   // [rows][6]
    const float* yw;   // [rows][6]
    const int* xreg;   // per general level: first output column anchored at or right of region column rx
    const int* yreg;
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
// A wave owns 64 source columns (60 outputs) and walks down kUnitTH rows: per row one coalesced load of
// its own column (all rows requested up front), DPP shifts for the neighbours x - 2 .. x + 3, 6 horizontal
// FMAs, a 6-row register window, 6 vertical FMAs, one store (scipy's six taps at zoom 1: unit_taps6, silent_gray.h).
template <int C>
__global__ __launch_bounds__(256) void pyramid_unit_kernel(const float* __restrict__ frames,
                                                           float* __restrict__ pyr, const PyrTab tab) {
    constexpr int R = kUnitTH;
    const unsigned bid = blockIdx.x;
    const int frame = (int)(bid / (unsigned)tab.unit_tiles_per_frame);
    int rem = (int)(bid - (unsigned)frame * (unsigned)tab.unit_tiles_per_frame);
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < tab.n_levels && rem >= tab.unit_tile_start[i]) l = i;
    rem -= tab.unit_tile_start[l];
    const PyrLevelDev& lv = tab.lv[l];
    const int ty = rem / tab.unit_tiles_x[l];
    const int tx = rem - ty * tab.unit_tiles_x[l];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = tx * kUnitTW + wave * kUnitCols;
    if (xw0 >= lv.out_w) return;  // wave-uniform
    const int y0 = ty * R;
    const int ox = xw0 + lane - 2;
    const int W = tab.W;
    const float* __restrict__ src = frames + (long long)frame * tab.H * W * C;
    float* __restrict__ dst = pyr + ((long long)frame * tab.frame_px_out + tab.px_off[l]) * C;

    float wx[6], wy[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        wx[i] = tab.xw[(long long)lv.xtab_off * 6 + i];
        wy[i] = tab.yw[(long long)lv.ytab_off * 6 + i];
    }
    const long long sx = (long long)(mirror_near(ox, lv.src_w) + lv.src_x0) * C;
    float in[R + 5][C];                                   // stream rows y0 - 2 .. y0 + R + 2
    float xcol[C];                                        // the column right of the wave's 64: lane 61's sixth tap, row i in lane i
#pragma unroll
    for (int i = 0; i < R + 5; ++i) {
        const long long sy = mirror_near(y0 - 2 + i, lv.src_h) + lv.src_y0;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) in[i][ch] = src[sy * W * C + sx + ch];
    }
#pragma unroll
    for (int ch = 0; ch < C; ++ch)
        xcol[ch] = unit_edge_column(src + ch, (long long)W * C, xw0 + 62, lv.src_w, lv.src_x0, y0 - 2, R + 5, lv.src_h, lv.src_y0, lane, C);
    // retire the loads before the first store (see gray_line_end_kernel)
#pragma unroll
    for (int i = 0; i < R + 5; ++i)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) asm volatile("" ::"v"(in[i][ch]));
#pragma unroll
    for (int ch = 0; ch < C; ++ch) asm volatile("" : "+v"(xcol[ch]));

    const bool out_lane = lane >= 2 && lane < 2 + kUnitCols && ox < lv.out_w;
    float hw[6][C];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) hw[j][ch] = 0.0f;
#pragma unroll
    for (int i = 0; i < R + 5; ++i) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) {
            const float h = unit_taps6(in[i][ch], unit_edge(xcol[ch], i), wx);
#pragma unroll
            for (int j = 0; j < 5; ++j) hw[j][ch] = hw[j + 1][ch];
            hw[5][ch] = h;
        }
        if (i >= 5) {
            const int oy = y0 + i - 5;
            if (oy < lv.out_h && out_lane) {
                const bool live = oy < lv.zoom_h && ox < lv.zoom_w;
                float* __restrict__ po = dst + ((long long)oy * lv.out_w + ox) * C;
#pragma unroll
                for (int ch = 0; ch < C; ++ch) {
                    float v = wy[0] * hw[0][ch];
#pragma unroll
                    for (int j = 1; j < 6; ++j) v = __builtin_fmaf(wy[j], hw[j][ch], v);
                    po[ch] = live ? v : 0.0f;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ REGION
// One block per source region: the region (+ halo) is staged into LDS once (batched float4 loads), then each
// general level's outputs anchored in the region are produced from that copy: 6-tap vertical pass into LDS
// for every staged column, barrier, 6-tap horizontal pass, coalesced store.  The kernel is latency-bound
// (staging loads, LDS round trips, barriers), so the LDS footprint is kept small: 25 KB (1 channel) / 22 KB
// (3 channels) -> 6 / 7 blocks per CU that cover each other's waits.
template <int C>
__global__ __launch_bounds__(256) void pyramid_region_kernel(const float* __restrict__ frames,
                                                             float* __restrict__ pyr, const PyrTab tab) {
    constexpr int RW = region_w(C), SW = region_sw(C), SH = region_sh(C), ROWF = SW * C, VR = region_vr(C);
    __shared__ __attribute__((aligned(16))) float s_src[SH * ROWF];
    __shared__ __attribute__((aligned(16))) float s_v[VR * ROWF];

    const int per_frame = tab.regions_x * tab.regions_y;
    // XCD-contiguous order: regions that share halo rows / columns meet in one L2 (-1.5 % on RGB; the same
    // remap made the streaming filter kernels slower)
    const unsigned bid = xcd_swizzle(blockIdx.x, gridDim.x);
    const int frame = (int)(bid / (unsigned)per_frame);
    const int rem = (int)(bid - (unsigned)frame * (unsigned)per_frame);
    const int ry = rem / tab.regions_x, rx = rem - ry * tab.regions_x;
    const int X0 = rx * RW, Y0 = ry * region_h(C);
    const int W = tab.W, H = tab.H;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ src = frames + (long long)frame * H * W * C;

    // A region no level has an output in is not staged at all (crop layouts: the reference's nested centre crops leave about half
    // of a 1080p frame's regions outside the outermost crop -- round 5: those blocks used to load their region and find nothing
    // to do with it; reference_layout_gray pyramid 0.187 -> see profiles/r05_experiments.txt 9).  Block-uniform.
    {
        typedef const __attribute__((address_space(4))) int* const_int_ptr;
        const_int_ptr xreg = (const_int_ptr)tab.xreg;
        const_int_ptr yreg = (const_int_ptr)tab.yreg;
        bool any = false;
        for (int l = 0; l < tab.n_levels; ++l) {
            const PyrLevelDev& lv = tab.lv[l];
            if (lv.kind != kPyrGeneral) continue;
            any = any || (xreg[lv.xreg_off + rx] < xreg[lv.xreg_off + rx + 1] && yreg[lv.yreg_off + ry] < yreg[lv.yreg_off + ry + 1]);
        }
        if (!any) return;
    }

    // stage rows [Y0-3, Y0+RH+3) x columns [X0-4, X0+RW+4).  Taps are mirrored INTO the crop, so positions
    // outside the frame are never referenced: rows are clamped, out-of-frame column groups are skipped.
    if (((W * C) & 3) == 0) {
        constexpr int V4 = ROWF / 4, NB = (SH * V4 + 255) / 256;
        float4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = min(tid + 256 * k, SH * V4 - 1);
            const int r = p / V4, q = p - r * V4;
            const int sy = min(max(Y0 - kRegionHaloT + r, 0), H - 1);
            long long f = (long long)(X0 - kRegionHaloL) * C + q * 4;  // first float of this group in the row
            f = min(max(f, 0ll), (long long)W * C - 4);
            v[k] = *reinterpret_cast<const float4*>(src + (long long)sy * W * C + f);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = tid + 256 * k;
            const int r = p / V4, q = p - r * V4;
            const long long f = (long long)(X0 - kRegionHaloL) * C + q * 4;
            if (p < SH * V4 && f >= 0 && f + 4 <= (long long)W * C)
                *reinterpret_cast<float4*>(s_src + r * ROWF + q * 4) = v[k];
        }
    } else {
        constexpr int NB = (SH * ROWF + 255) / 256;
        float v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = min(tid + 256 * k, SH * ROWF - 1);
            const int r = p / ROWF, q = p - r * ROWF;
            const int sy = min(max(Y0 - kRegionHaloT + r, 0), H - 1);
            const long long f = min(max((long long)(X0 - kRegionHaloL) * C + q, 0ll), (long long)W * C - 1);
            v[k] = src[(long long)sy * W * C + f];
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = tid + 256 * k;
            if (p < SH * ROWF) s_src[p] = v[k];
        }
    }
    __syncthreads();

    for (int l = 0; l < tab.n_levels; ++l) {
        const PyrLevelDev& lv = tab.lv[l];
        if (lv.kind != kPyrGeneral) continue;
        const int xs = tab.xreg[lv.xreg_off + rx], xe = tab.xreg[lv.xreg_off + rx + 1];
        const int ys = tab.yreg[lv.yreg_off + ry], ye = tab.yreg[lv.yreg_off + ry + 1];
        if (xs >= xe || ys >= ye) continue;  // block-uniform
        float* __restrict__ dst = pyr + ((long long)frame * tab.frame_px_out + tab.px_off[l]) * C;
        const int xshift = lv.src_x0 - (X0 - kRegionHaloL), yshift = lv.src_y0 - (Y0 - kRegionHaloT);
        // this lane's horizontal taps (first 64 output columns of the region) are requested before the vertical
        // pass, so their latency is covered by it; the vertical taps are wave-uniform -> scalar loads
        int co0[6];
        float wx0[6];
        {
            const long long xe6 = (long long)(lv.xtab_off + min(xs + lane, xe - 1)) * 6;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                co0[i] = min(max(tab.xidx[xe6 + i] + xshift, 0), SW - 1) * C;
                wx0[i] = tab.xw[xe6 + i];
            }
        }
        typedef const __attribute__((address_space(4))) int* const_int_ptr;
        typedef const __attribute__((address_space(4))) float* const_float_ptr;
        for (int yc = ys; yc < ye; yc += VR) {
            const int nr = min(VR, ye - yc);
            // vertical 6 taps for every staged float of the rows this chunk needs (lanes = consecutive floats)
            for (int orow = wave; orow < nr; orow += 4) {
                const long long ye6 = (long long)(lv.ytab_off + yc + orow) * 6;
                const_int_ptr yi = (const_int_ptr)(tab.yidx + ye6);
                const_float_ptr yw = (const_float_ptr)(tab.yw + ye6);
                int ro[6];
                float wy[6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    ro[j] = min(max(yi[j] + yshift, 0), SH - 1) * ROWF;
                    wy[j] = yw[j];
                }
                for (int c = lane; c < ROWF; c += 64) {
                    float acc = wy[0] * s_src[ro[0] + c];
#pragma unroll
                    for (int j = 1; j < 6; ++j) acc = __builtin_fmaf(wy[j], s_src[ro[j] + c], acc);
                    s_v[orow * ROWF + c] = acc;
                }
            }
            __syncthreads();
            // horizontal 6 taps and store (lanes = consecutive output columns)
            for (int oc = xs + lane; oc < xe; oc += 64) {
                int co[6];
                float wx[6];
                if (oc < xs + 64) {  // wave-uniform
#pragma unroll
                    for (int i = 0; i < 6; ++i) co[i] = co0[i], wx[i] = wx0[i];
                } else {
                    const long long xe6 = (long long)(lv.xtab_off + oc) * 6;
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        co[i] = min(max(tab.xidx[xe6 + i] + xshift, 0), SW - 1) * C;
                        wx[i] = tab.xw[xe6 + i];
                    }
                }
                for (int orow = wave; orow < nr; orow += 4) {
                    float* __restrict__ po = dst + ((long long)(yc + orow) * lv.out_w + oc) * C;
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) {
                        float acc = wx[0] * s_v[orow * ROWF + co[0] + ch];
#pragma unroll
                        for (int i = 1; i < 6; ++i) acc = __builtin_fmaf(wx[i], s_v[orow * ROWF + co[i] + ch], acc);
                        po[ch] = acc;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------ BORDER
// The reference's layout (image_to_zoom_tensor, from_image.py:45-64) resamples NESTED centre crops.  pyramid_walk3_kernel reads the
// outermost crop once and serves the inner levels from the same rows wherever an output's 6 x 6 taps lie inside ITS level's crop
// (silent_pyramid_api.hip, "union" plans).  What is left is each inner level's frame of outputs whose taps scipy mirrors at that
// level's own crop edge -- the first / last output row and column, a few thousand pixels per frame: one thread per (pixel, channel),
// taps straight from the frame through the plan's mirrored tap tables, in the walk's order of operations (vertical 6 fmas from
// fmaf(w0, x0, 0), then w0 * v0 and 5 fmas): bit-identical to what the level's own plan produced.
struct BorderLevel {
    int level;                         // index into PyrTab::lv
    int oy_lo, oy_hi, ox_lo, ox_hi;    // interior outputs [oy_lo, oy_hi) x [ox_lo, ox_hi): served by the union walk
    int zr, zc;                        // outputs the resampler produces (min(zoom, canvas))
    int start;                         // first border pixel of this level among one frame's border pixels
};
struct BorderTab {
    int n, per_frame;
    BorderLevel lv[kMaxLevels];
};

// border pixel x channel number gid of the launch (frames x bt.per_frame x C of them).  LEAN: one tap column at a time (6 loads in
// flight instead of 36) -- inside the walk kernel, whose 7 waves per SIMD must not pay for this path's registers
template <int C, bool LEAN = false>
__device__ __forceinline__ void pyramid_border_px(const float* __restrict__ frames, float* __restrict__ pyr, const PyrTab& tab,
                                                  const BorderTab& bt, long long gid, int n_frames) {
    const long long per_frame = (long long)bt.per_frame * C;
    if (gid >= per_frame * n_frames) return;
    const int frame = (int)(gid / per_frame);
    int k = (int)(gid - (long long)frame * per_frame);
    const int ch = k % C;
    k /= C;
    int bi = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < bt.n && k >= bt.lv[i].start) bi = i;
    const BorderLevel& b = bt.lv[bi];
    k -= b.start;
    // border pixels in this order: the rows above the interior, the rows below it, then per interior row the columns left / right of it
    const int top = b.oy_lo * b.zc, bottom = (b.zr - b.oy_hi) * b.zc, ih = b.oy_hi - b.oy_lo, side = b.ox_lo + (b.zc - b.ox_hi);
    int oy, ox;
    if (k < top) {
        oy = k / b.zc;
        ox = k - oy * b.zc;
    } else if (k < top + bottom) {
        const int q = k - top;
        oy = b.oy_hi + q / b.zc;
        ox = q - (q / b.zc) * b.zc;
    } else {
        const int q = k - top - bottom;
        oy = b.oy_lo + q / side;
        const int r = q - (q / side) * side;
        ox = r < b.ox_lo ? r : b.ox_hi + (r - b.ox_lo);
    }
    (void)ih;
    const PyrLevelDev& lv = tab.lv[b.level];
    const float* __restrict__ src = frames + (long long)frame * tab.H * tab.W * C;
    const int* __restrict__ yi = tab.yidx + (long long)(lv.ytab_off + oy) * 6;
    const int* __restrict__ xi = tab.xidx + (long long)(lv.xtab_off + ox) * 6;
    const float* __restrict__ wy = tab.yw + (long long)(lv.ytab_off + oy) * 6;
    const float* __restrict__ wx = tab.xw + (long long)(lv.xtab_off + ox) * 6;
    auto column = [&](int j) {
        const long long col = (long long)(xi[j] + lv.src_x0) * C + ch;
        float a = __builtin_fmaf(wy[0], src[(long long)(yi[0] + lv.src_y0) * tab.W * C + col], 0.0f);
#pragma unroll
        for (int i = 1; i < 6; ++i) a = __builtin_fmaf(wy[i], src[(long long)(yi[i] + lv.src_y0) * tab.W * C + col], a);
        return a;
    };
    float acc;
    if constexpr (LEAN) {
        acc = wx[0] * column(0);
#pragma unroll 1
        for (int j = 1; j < 6; ++j) acc = __builtin_fmaf(wx[j], column(j), acc);
    } else {
        float v[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) v[j] = column(j);
        acc = wx[0] * v[0];
#pragma unroll
        for (int j = 1; j < 6; ++j) acc = __builtin_fmaf(wx[j], v[j], acc);
    }
    pyr[((long long)frame * tab.frame_px_out + tab.px_off[b.level] + (long long)oy * lv.out_w + ox) * C + ch] = acc;
}

// (on its own: PYRAMID knob 8; by default these pixels are the first blocks of the walk's launch, silent_walk_rgb.h)
template <int C>
__global__ __launch_bounds__(256) void pyramid_border_kernel(const float* __restrict__ frames, float* __restrict__ pyr, const PyrTab tab,
                                                             const BorderTab bt, int n_frames) {
    pyramid_border_px<C>(frames, pyr, tab, bt, (long long)blockIdx.x * 256 + threadIdx.x, n_frames);
}

// ------------------------------------------------------------------------------------------ ZERO FILL
// Canvas pixels the resampler does not produce -- rows zoom_h .. out_h - 1 over the full width, columns zoom_w .. out_w - 1 of the
// rows above them (a canvas larger than its zoomed crop; scipy's mode-'constant' artefact row / column) -- and only those: the
// chunks of 1024 enumerate exactly that L-shaped set (host: the same count in silent_pyramid_plan_create).
__host__ __device__ inline long long pyramid_zero_count(int zoom_h, int zoom_w, int out_h, int out_w) {
    const int zh = zoom_h < out_h ? zoom_h : out_h, zw = zoom_w < out_w ? zoom_w : out_w;
    return (long long)(out_h - zh) * out_w + (long long)zh * (out_w - zw);
}
template <int C>
__global__ __launch_bounds__(256) void pyramid_zero_kernel(float* __restrict__ pyr, const PyrTab tab) {
    const unsigned bid = blockIdx.x;
    const int frame = (int)(bid / (unsigned)tab.zero_chunks_per_frame);
    int rem = (int)(bid - (unsigned)frame * (unsigned)tab.zero_chunks_per_frame);
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < tab.n_levels && rem >= tab.zero_chunk_start[i]) l = i;
    rem -= tab.zero_chunk_start[l];
    const PyrLevelDev& lv = tab.lv[l];
    float* __restrict__ dst = pyr + ((long long)frame * tab.frame_px_out + tab.px_off[l]) * C;
    const int zh = min(lv.zoom_h, lv.out_h), zw = min(lv.zoom_w, lv.out_w);
    const int n_below = (lv.out_h - zh) * lv.out_w, wide = lv.out_w - zw, n_right = zh * wide;
    for (int k = 0; k < 4; ++k) {
        const int p = rem * 1024 + k * 256 + threadIdx.x;
        if (p >= n_below + n_right) break;
        int y, x;
        if (p < n_below) {
            y = zh + p / lv.out_w;
            x = p - (p / lv.out_w) * lv.out_w;
        } else {
            const int q = p - n_below;
            y = q / wide;
            x = zw + (q - y * wide);
        }
        const long long px = (long long)y * lv.out_w + x;
        for (int ch = 0; ch < C; ++ch) dst[px * C + ch] = 0.0f;
    }
}

}  // namespace silent
