// pyramid_walk3_kernel: the whole zoom pyramid of a 3-channel (interleaved RGB) frame from ONE read of the frame, as a
// strip walk with a dedicated loader wave -- the structure of gray_walk_kernel (silent_walk.h) applied to the pyramid of
// BASELINE config 3 (the gray kernel is kept as a record under scripts/ubench/walk_kernels/), where round 1 ran two kernels (pyramid_unit_kernel<3> + pyramid_region_kernel<3>: the frame is fetched
// 2.95x, the region kernel moves 1.64x its algorithmic bytes and is latency-bound at 1.9 TB/s).
//
// Everything is indexed in FLOATS of the interleaved row (pixel p, channel c <-> float 3p + c):
//   * a block owns a strip of 144 pixels of one frame segment; wave 4 (the loader) LDS-DMAs the strip's rows (+ 4 / 3 halo
//     pixels) into a ring of 3 chunks x 8 rows, two chunks ahead, together with the chunk's row records;
//   * waves 0-3 (consumers) own 36 pixels each = 108 floats, two floats per lane, plus a halo of 2 / 3 pixels that only
//     the gather of the other levels needs: the unit level's 5 horizontal taps sit 3 floats apart and are read straight
//     from the ring (no DPP, hence no halo lanes for the smoother), the 5 vertical taps are a register window;
//   * the other levels: every output row of every level is a 6-tap combination of 6 CONSECUTIVE source rows (the spline's
//     support; the zoom step only says how often a row completes), so the lane keeps a window of its last 6 source rows and
//     evaluates a row when the row program (one record per source row, DMA'd into the ring with the rows) says it completes:
//     6 weights from the record, 6 fmas per float, taps ascending.  No rolling accumulators, no slots, no restart logic --
//     the first version carried gray_walk_kernel's accumulate-every-row scheme and spent 0.32 of its 0.87 ms there.  The
//     completed row goes through a wave-private LDS line, and lane 3j + c gathers the 6 taps of channel c of output pixel j
//     (3 floats apart) with column records staged in LDS -- the arithmetic order of the region kernel, bit-identical.
// Same protocol as the gray walk: one raw s_barrier per chunk, uniform trip counts, the consumers' vmcnt queue holds
// stores only.
// Round 3: a launch runs up to kW3MaxPlans WALK PLANS side by side (block -> frame, plan, segment, strip).  A plan is one crop
// of the frame with an optional unit level and the general levels that resample THAT crop.  A classic pyramid is one plan
// (unit level + every other level on the whole frame).  The reference's layout (image_to_zoom_tensor, from_image.py:45-64:
// nested centre crops, each resampled to one fixed size) is one plan per level: the unit level on the innermost crop, every
// other level alone on its own crop -- each mirrors at ITS crop's edge exactly like scipy does on the cropped array, and the
// nested re-reads of one frame sit next to each other in the launch (Infinity Cache).  The ring row is addressed in floats
// of the FRAME row from a 16-byte aligned origin (any crop offset), PX = 36 or 32 pixels per consumer wave (32: zoom steps
// down to 1.6, the reference's e ** .5).  Eligibility (host): W a multiple of 4, <= 21 outputs per wave tile and level.
#pragma once

#include <type_traits>

#include "silent_common.h"
#include "silent_gray.h"
#include "silent_pyramid.h"

namespace silent {

// ---- strip-walk declarations
constexpr int kWalkCH = 8;                          // rows per record chunk the host pads the row program to
#ifndef SILENT_W3_SLOTS
#define SILENT_W3_SLOTS 3
#endif
constexpr int kWalkSlots = SILENT_W3_SLOTS;         // chunks in the ring (the loader runs kWalkSlots - 1 chunks ahead of the consumers)

// tables of the in-walk pyramid (device memory owned by the plan)
struct WalkPyr {
    int G;                        // general levels (<= template G; the rest are inert)
    const int* row_prog;          // [out_h + 8 (+ padding)][w3_prog_row(Gp)]: record of stream row y at index y + 4
    const int* col_hdr;           // [G][waves_x][2]: first output column, number of outputs of the wave tile
    const int* col_rec;           // [waves_x][w3_rec_total(Gp)][8]: float index of tap 0 in the wave's line, 6 weights, pad
    long long px_off[8];          // pixel offset of level g inside one pyramid
    int out_w[8];
};

// compile-time loop: the body gets its index as an integral constant
template <int I, int N, class F>
__device__ __forceinline__ void walk_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        walk_static_for<I + 1, N>(f);
    }
}

typedef __attribute__((address_space(3))) void* walk_lds_ptr;
typedef const __attribute__((address_space(1))) void* walk_glb_ptr;

constexpr int kW3NC = 4;                        // consumer waves per block
constexpr int kW3HaloL = 4, kW3HaloR = 3;       // ring halo in pixels
constexpr int kW3MaxPlans = 8;                  // walk plans per launch
constexpr int kW3TileL = 2;                     // a wave's line starts 2 pixels left of its first pixel ...
constexpr int kW3TileF = 124;                   // ... and holds 124 floats (41 pixels + 1 float): taps -2 .. +3 of its anchors
// floats per ring row: the strip's 4 PX pixels + 7 of halo, + up to 3 floats of alignment shift, in whole 16-byte pieces (PX = 36: 456)
__host__ __device__ constexpr int w3_row_f(int px) { return ((kW3NC * px + kW3HaloL + kW3HaloR) * 3 + 3 + 3) / 4 * 4; }
constexpr int kW3RingTail = 16;                 // floats behind the ring's last row: a lane's six taps are tap 0 + 0, 3 .. 15 floats, and the
                                                // idle tail lanes of a wave's line read up to 12 floats past their row (into the next one)
constexpr int kW3Threads = (kW3NC + 1) * 64;
#ifndef SILENT_W3_CH
#define SILENT_W3_CH 4
#endif
constexpr int kW3CH = SILENT_W3_CH;             // rows per chunk: a 12-row ring (22 KB) instead of 24 rows -- the loader is far from
                                                // being the limit (all loads alone: 0.14 ms), the consumers are latency-bound and
                                                // want waves: 31 KB of LDS per block = 4 blocks (16 consumer waves) per CU instead of 3
// column records per wave tile and level: output PIXELS anchored in a wave's PX pixels (36: zoom steps >= 1.875 ^ (g + 1);
// 32: >= 1.6 ^ (g + 1); 28: >= 1.4 ^ (g + 1), e.g. sqrt 2; 24: >= 1.2 ^ (g + 1), e.g. 2 ^ (1/3)); never more than 21 (the gather
// works on lane 3 j + c).  The host tries 36, 32, 28, 24 in this order and takes the first whose capacities hold the plan.
__host__ __device__ constexpr int w3_rec_cap(int px, int g) {
    return px == 36   ? (g == 0 ? 21 : (g == 1 ? 11 : (g == 2 ? 6 : (g == 3 ? 4 : (g <= 5 ? 2 : 1)))))
           : px == 32 ? (g == 0 ? 21 : (g == 1 ? 14 : (g == 2 ? 9 : (g == 3 ? 6 : (g == 4 ? 5 : 3)))))
           : px == 28 ? (g == 0 ? 21 : (g == 1 ? 15 : (g == 2 ? 11 : (g == 3 ? 8 : (g == 4 ? 6 : (g == 5 ? 4 : 3))))))
                      : (g == 0 ? 21 : (g == 1 ? 17 : (g == 2 ? 14 : (g == 3 ? 12 : (g == 4 ? 10 : (g == 5 ? 9 : 7))))));
}
__host__ __device__ constexpr int w3_rec_base(int px, int g) {
    int n = 0;
    for (int i = 0; i < g; ++i) n += w3_rec_cap(px, i);
    return n;
}
__host__ __device__ constexpr int w3_rec_total(int px, int g) { return w3_rec_base(px, g); }

// one walk plan (see the header)
struct Walk3Plan {
    int src_y0, src_x0, src_h, src_w;    // the crop this plan walks (frame coordinates)
    int shift;                           // (src_x0 * 3) mod 4: floats between the aligned ring origin and the crop's pixel - 4
    int has_unit;                        // the plan stores a unit (zoom 1) level
    int out_h, out_w, eff_h, eff_w;      // walk extents: the unit level's canvas and the part the crop covers (no unit level: the crop)
    int strips_x, segs_y, seg_rows;      // decomposition of this plan
    int block0;                          // first block of this plan among one frame's blocks
    long long px_off;                    // offset of the unit level in one pyramid
    WalkPyr pyr;                         // general levels of this plan (G may be 0)
};
struct Walk3Args {
    int H, W;                            // frame extents
    int n_plans, blocks_per_frame;
    long long frame_px;                  // pixels of one pyramid
    float wx[6];                         // [1, 26, 66, 26, 1] / 120 and scipy's sixth tap, 2^-53, as float32 (both axes)
    Walk3Plan plan[kW3MaxPlans];
};
// Union plans: the inner levels' first / last output rows and columns (pyramid_border_px, silent_pyramid.h: one thread per pixel and
// channel, taps straight from the frame) are the LAST blocks of the walk's launch, from block `first` on -- they fill the slots the
// walk's last round of blocks leaves idle instead of running behind it in a launch of their own.
struct WalkBorderArgs {
    int first, n_frames;                 // first = INT_MAX: none
    PyrTab tab;
    BorderTab bt;
};
// row record of the walk: [meta(0) .. meta(Gp-1)] [6 weights of level 0] ... [6 weights of level Gp-1], padded to a multiple
// of 4 dwords.  meta: bit 0 "an output row of this level completes with this source row", bits 8.. that output row.
__host__ __device__ constexpr int w3_prog_row(int gp) { return (gp * 7 + 3) / 4 * 4; }

template <int G, int PX>
__global__ __launch_bounds__(kW3Threads) void pyramid_walk3_kernel(const float* __restrict__ frames, float* __restrict__ pyr,
                                                                    const Walk3Args args, const WalkBorderArgs border) {
    static_assert(G == stream_pad_levels(G), "row programs are padded to 4 or 7 levels");
    static_assert(PX == 36 || PX == 32 || PX == 28 || PX == 24, "");
    constexpr int kW3Px = PX, kW3StripPx = kW3NC * PX, kW3RowF = w3_row_f(PX);
    static_assert(kW3RowF > 256 && kW3RowF <= 512, "two DMA loads of 64 x 16 bytes per ring row");
    constexpr int PR = w3_prog_row(G);
    constexpr int kRecTotal = w3_rec_total(PX, G);
    __shared__ __attribute__((aligned(16))) float s_ring_f[kWalkSlots * kW3CH * kW3RowF + kW3RingTail];
    float (*s_ring)[kW3RowF] = reinterpret_cast<float (*)[kW3RowF]>(s_ring_f);
    __shared__ __attribute__((aligned(16))) int s_prog[kWalkSlots * kW3CH * PR];                // row records of the ring's chunks
    __shared__ __attribute__((aligned(16))) float s_line[kW3NC * 128];                            // a completed row, per wave
    __shared__ __attribute__((aligned(16))) int s_rec[kW3NC * kRecTotal * 8];                     // column records, per wave

    if (blockIdx.x >= (unsigned)border.first) {   // block-uniform
        if (threadIdx.x < 256)
            pyramid_border_px<3, true>(frames, pyr, border.tab, border.bt, (long long)(blockIdx.x - (unsigned)border.first) * 256 + threadIdx.x, border.n_frames);
        return;
    }
    const unsigned bid = blockIdx.x;
    const int frame = (int)(bid / (unsigned)args.blocks_per_frame);
    int rest = (int)(bid - (unsigned)frame * (unsigned)args.blocks_per_frame);
    int pi = 0;
#pragma unroll
    for (int i = 1; i < kW3MaxPlans; ++i)
        if (i < args.n_plans && rest >= args.plan[i].block0) pi = i;
    const Walk3Plan& tab = args.plan[pi];          // (scalar loads: the plan index is block-uniform)
    const WalkPyr& wp = tab.pyr;
    rest -= tab.block0;
    const int strip = rest % tab.strips_x;
    const int seg = rest / tab.strips_x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg_y0 = seg * tab.seg_rows;
    const int seg_h = min(tab.seg_rows, tab.out_h - seg_y0);
    const int n_rows = seg_h + 8;                              // stream rows seg_y0 - 4 .. seg_y0 + seg_h + 3
    const int n_chunks = (n_rows + kW3CH - 1) / kW3CH;
    const int X0 = strip * kW3StripPx;                         // first pixel of the strip; the ring row starts at pixel X0 - 4
    const int R0 = X0 - kW3HaloL;

    if (wave == kW3NC) {
        // ------------------------------------------------------------------ loader: LDS-DMA only
        const float* __restrict__ src = frames + (long long)frame * args.H * args.W * 3;
        const int rowf = args.W * 3;                           // floats of a FRAME row
        // ring float r <-> frame-row float A + r, A = the crop's pixel R0 minus the alignment shift.  Frame widths that are a
        // multiple of 4 (every row starts on 16 bytes): A is a multiple of 4 and a lane fetches a 16-byte aligned group of 4 floats,
        // two DMA loads per ring row.  Other widths (round 5): no shift, a lane fetches ONE FLOAT, 5 - 8 loads per ring row -- the
        // loader has the time (one row per ~2000 cycles of its consumers).  (The 12-byte DMA, one pixel per lane, is no way out: it
        // writes lane l's 12 bytes at 16 l, scripts/ubench/lds_dma_b96.hip.)  Pieces that stick out of the row are clamped to a valid
        // address; whatever lies outside the CROP is never read (the consumers read mirrored pixels instead)
        const int A = (tab.src_x0 + R0) * 3 - tab.shift;
        auto loader = [&](auto dword_dma) {
            constexpr bool DWORD_DMA = decltype(dword_dma)::value;
            constexpr int kRowLoads = DWORD_DMA ? (kW3RowF + 63) / 64 : 2;   // DMA instructions per ring row
            int f[kRowLoads];
#pragma unroll
            for (int k = 0; k < kRowLoads; ++k)
                f[k] = DWORD_DMA ? min(max(A + lane + 64 * k, 0), rowf - 1) : min(max(A + 256 * k + lane * 4, 0), rowf - 4);
            auto issue = [&](int c, int slot) {
#pragma unroll
                for (int r = 0; r < kW3CH; ++r) {
                    const int y = seg_y0 - 4 + c * kW3CH + r;
                    const float* rp = src + (long long)(mirror_near(y, tab.src_h) + tab.src_y0) * args.W * 3;
                    float* dst = &s_ring[slot * kW3CH + r][0];
                    if constexpr (DWORD_DMA) {
#pragma unroll
                        for (int k = 0; k < kRowLoads; ++k)
                            if (k + 1 < kRowLoads || lane < kW3RowF - 64 * k)
                                __builtin_amdgcn_global_load_lds((walk_glb_ptr)(rp + f[k]), (walk_lds_ptr)(dst + 64 * k), 4, 0, 0);
                    } else {
                        __builtin_amdgcn_global_load_lds((walk_glb_ptr)(rp + f[0]), (walk_lds_ptr)dst, 16, 0, 0);
                        if (lane < (kW3RowF - 256) / 4)
                            __builtin_amdgcn_global_load_lds((walk_glb_ptr)(rp + f[1]), (walk_lds_ptr)(dst + 256), 16, 0, 0);
                    }
                }
                const int* rp = wp.row_prog + ((long long)seg_y0 + (long long)c * kW3CH) * PR + lane * 4;
                int* dst = s_prog + slot * (kW3CH * PR);
                constexpr int NV = kW3CH * PR / 4;               // 16-byte pieces of a chunk's records
                if (lane < NV) __builtin_amdgcn_global_load_lds((walk_glb_ptr)rp, (walk_lds_ptr)dst, 16, 0, 0);
                if constexpr (NV > 64) {
                    if (lane < NV - 64) __builtin_amdgcn_global_load_lds((walk_glb_ptr)(rp + 256), (walk_lds_ptr)(dst + 256), 16, 0, 0);
                }
            };
            // scipy mirrors every tap at the CROP's edge (d c b | a b c d | c b a).  The ring row of a strip at the crop's left / right
            // edge holds whatever lies beside the crop in the frame (or a clamped address): the loader writes the mirrored pixels over
            // those halo positions -- up to 4 pixels left (q = -4 .. -1 <- -q) and 3 right (q = n .. n + 2 <- 2 (n - 1) - q) -- once per
            // chunk, so that the consumers read every tap at a fixed offset from tap 0 (no per-tap offsets, no mirror arithmetic)
            int fix_dst = 0, fix_src = 0;
            bool fix = false;
            {
                const int n = tab.src_w;
                const int q = lane < 12 ? -4 + lane / 3 : n + (lane - 12) / 3;
                const int ch = lane < 12 ? lane % 3 : (lane - 12) % 3;
                const int qs = lane < 12 ? -q : 2 * (n - 1) - q;
                fix = lane < 21 && q - R0 >= 0 && q - R0 < kW3StripPx + kW3HaloL + kW3HaloR && qs - R0 >= 0 && qs - R0 < kW3StripPx + kW3HaloL + kW3HaloR;
                fix_dst = (q - R0) * 3 + ch + tab.shift;
                fix_src = (qs - R0) * 3 + ch + tab.shift;
            }
            const bool any_fix = __any(fix);                        // wave-uniform: an edge strip
            static_assert(kWalkSlots == 2 || kWalkSlots == 3, "");
            issue(0, 0);
            if (kWalkSlots == 3 && n_chunks > 1) issue(1, 1);
            int slot2 = kWalkSlots - 1, slot0 = 0;
            for (int c = 0; c < n_chunks; ++c) {
                // chunk c has landed when at most the next chunk's loads (three slots: one chunk in flight behind it) are outstanding
                if (kWalkSlots == 3 && c + 1 < n_chunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kRowLoads * kW3CH + (kW3CH * PR / 4 > 64 ? 2 : 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (any_fix) {
#pragma unroll
                    for (int r = 0; r < kW3CH; ++r) {
                        float* row = &s_ring[slot0 * kW3CH + r][0];
                        if (fix) row[fix_dst] = row[fix_src];
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                slot0 = slot0 == kWalkSlots - 1 ? 0 : slot0 + 1;
                __builtin_amdgcn_s_barrier();
                if (c + kWalkSlots - 1 < n_chunks) issue(c + kWalkSlots - 1, slot2);
                slot2 = slot2 == kWalkSlots - 1 ? 0 : slot2 + 1;
            }
        };
        if (rowf & 3) loader(std::true_type{});                 // block-uniform
        else loader(std::false_type{});
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wx0 = X0 + wave * kW3Px;                          // first pixel of this wave
    const bool live = wx0 < tab.out_w;                          // wave-uniform; a dead wave still meets every barrier
    // the lane's two floats: line floats lane and lane + 64 <-> pixel wx0 - 2 + i / 3, channel i % 3 (i = lane + 64 k).  Stride 1
    // across the lanes: the ring reads and the line writes are free of bank conflicts (floats 2 lane, 2 lane + 1 gave 2-way
    // conflicts on every ds_read_b32: SQ_LDS_BANK_CONFLICT was 1.6x the LDS instruction cycles), and a store instruction writes
    // 256 contiguous bytes without any alignment condition on the level
    // ring offset (in floats) of tap 0 (pixel - 2) of each float; tap d is 3 d floats further (the loader has mirrored the crop's
    // edges into the ring): one address per float, the six taps as immediate offsets (pairs of them share a ds_read2_b32)
    int tap0[2];
    int px[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = lane + 64 * k;
        const int p = wx0 - kW3TileL + i / 3, c = i % 3;
        px[k] = p;
        tap0[k] = min(max((p - 2 - R0) * 3 + c + tab.shift, 0), kW3RowF - 1);
    }
    const bool out0 = px[0] >= wx0 && px[0] < wx0 + kW3Px && px[0] < tab.out_w;
    const bool out1 = px[1] >= wx0 && px[1] < wx0 + kW3Px && px[1] < tab.out_w && lane + 64 < kW3TileF;
    const bool eff0 = px[0] < tab.eff_w, eff1 = px[1] < tab.eff_w;
    const long long base_f = ((long long)frame * args.frame_px + tab.px_off) * 3 + (long long)(wx0 - kW3TileL) * 3 + lane;
    const bool has_unit = tab.has_unit != 0;                    // block-uniform

    float hist[6][2];                                           // the lane's two floats on the last 6 source rows
#pragma unroll
    for (int j = 0; j < 6; ++j) hist[j][0] = hist[j][1] = 0.0f;
    int gx0[G], gn[G];
    const long long frame_px0 = (long long)frame * args.frame_px;
    int* const my_rec = s_rec + wave * (kRecTotal * 8);
    float* const my_line = s_line + wave * 128;
    {
        const int wx_tile = strip * kW3NC + wave;
        const int4* __restrict__ src4 = reinterpret_cast<const int4*>(wp.col_rec) + (long long)wx_tile * (kRecTotal * 2);
        int4* dst4 = reinterpret_cast<int4*>(my_rec);
        if (wp.G > 0)
            for (int i = lane; i < kRecTotal * 2; i += 64) dst4[i] = src4[i];
        typedef const __attribute__((address_space(4))) int* const_int_ptr;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int gg = max(min(g, wp.G - 1), 0);
            const_int_ptr h = (const_int_ptr)(wp.col_hdr + ((long long)gg * (tab.strips_x * kW3NC) + wx_tile) * 2);
            gx0[g] = wp.G > 0 ? h[0] : 0;
            gn[g] = g < wp.G ? h[1] : 0;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // records staged (the only vector loads of a consumer)
        __builtin_amdgcn_wave_barrier();
    }
    const int gj = lane / 3, gc = lane - 3 * gj;               // gather role: output pixel gj, channel gc

    float hw[6][2];                                             // horizontally smoothed rows y-5 .. y of the two floats
#pragma unroll
    for (int j = 0; j < 6; ++j) hw[j][0] = hw[j][1] = 0.0f;

    int slot = 0;
    for (int c = 0; c < n_chunks; ++c) {
        __builtin_amdgcn_s_barrier();                           // barrier c: chunk c (rows + records) is in the ring
        asm volatile("" ::: "memory");
        if (live) {
#pragma unroll
            for (int r = 0; r < kW3CH; ++r) {
                const int s = c * kW3CH + r;                  // stream row; source row y = seg_y0 - 4 + s
                if (s >= n_rows) break;                         // wave-uniform
                const float* __restrict__ row = &s_ring[slot * kW3CH + r][0];
                float t[2][6];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int d = 0; d < 6; ++d) t[k][d] = row[tap0[k] + 3 * d];
                const int* __restrict__ prow = s_prog + (slot * kW3CH + r) * PR;
                int meta_v[(G + 3) / 4 * 4];
#pragma unroll
                for (int e = 0; e < (G + 3) / 4; ++e) {
                    const int4 q = reinterpret_cast<const int4*>(prow)[e];   // every lane reads the same record (LDS broadcast)
                    meta_v[4 * e] = q.x; meta_v[4 * e + 1] = q.y; meta_v[4 * e + 2] = q.z; meta_v[4 * e + 3] = q.w;
                }
                // ---- unit level: scipy's six horizontal taps x - 2 .. x + 3 (same fma order as pyramid_unit_kernel / unit_taps6:
                // the sixth, 2^-53, carries a NaN / inf pixel to the outputs three to its left and above), then the vertical window
                if (has_unit) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    float h = args.wx[0] * t[k][0];
                    h = __builtin_fmaf(args.wx[1], t[k][1], h);
                    h = __builtin_fmaf(args.wx[2], t[k][2], h);
                    h = __builtin_fmaf(args.wx[3], t[k][3], h);
                    h = __builtin_fmaf(args.wx[4], t[k][4], h);
                    h = __builtin_fmaf(args.wx[5], t[k][5], h);
#pragma unroll
                    for (int j = 0; j < 5; ++j) hw[j][k] = hw[j + 1][k];
                    hw[5][k] = h;
                }
                }
                const int p = seg_y0 + s - 7;                   // level-0 row that completes with source row y = p + 3
                if (has_unit && p >= seg_y0 && p < seg_y0 + seg_h) {        // wave-uniform (rows above are warm-up)
                    float v0 = args.wx[0] * hw[0][0], v1 = args.wx[0] * hw[0][1];
#pragma unroll
                    for (int j = 1; j < 6; ++j) {
                        v0 = __builtin_fmaf(args.wx[j], hw[j][0], v0);
                        v1 = __builtin_fmaf(args.wx[j], hw[j][1], v1);
                    }
                    const bool prow = p < tab.eff_h;
                    v0 = (prow && eff0) ? v0 : 0.0f;            // canvas beyond the zoomed crop
                    v1 = (prow && eff1) ? v1 : 0.0f;
                    float* __restrict__ po = pyr + base_f + (long long)p * tab.out_w * 3;
                    if (out0) po[0] = v0;
                    if (out1) po[64] = v1;
                }
                // ---- the other levels: the window of the last 6 source rows, and a row of level g when the record says so
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    hist[j][0] = hist[j + 1][0];
                    hist[j][1] = hist[j + 1][1];
                }
                hist[5][0] = t[0][2];                           // the lane's own floats (horizontal tap d = 0)
                hist[5][1] = t[1][2];
                const int anchor = seg_y0 + s - 7;              // anchor row of a row completing now: stored by the segment that owns it
                if (anchor >= seg_y0 && anchor < seg_y0 + seg_h) {   // wave-uniform
                    walk_static_for<0, G>([&](auto gcst) {
                        constexpr int g = decltype(gcst)::value;
                        const int meta = __builtin_amdgcn_readfirstlane(meta_v[g]);
                        if (!(meta & 1)) return;                // wave-uniform: no row of level g completes here
                        const int oy = meta >> 8;
                        const int jj = min(gj, w3_rec_cap(PX, g) - 1);
                        const int4* __restrict__ rc = reinterpret_cast<const int4*>(my_rec + (w3_rec_base(PX, g) + jj) * 8);
                        const int4 ra = rc[0], rb = rc[1];
                        const float* __restrict__ wv = reinterpret_cast<const float*>(prow + G + 6 * g);   // 6 vertical weights (broadcast)
                        float v0 = __builtin_fmaf(wv[0], hist[0][0], 0.0f), v1 = __builtin_fmaf(wv[0], hist[0][1], 0.0f);
#pragma unroll
                        for (int j = 1; j < 6; ++j) {
                            v0 = __builtin_fmaf(wv[j], hist[j][0], v0);
                            v1 = __builtin_fmaf(wv[j], hist[j][1], v1);
                        }
                        my_line[lane] = v0;
                        my_line[lane + 64] = v1;
                        __builtin_amdgcn_wave_barrier();
                        const float* tp = my_line + min(ra.x + gc, kW3TileF - 16);   // taps 3 floats apart (clamped: idle lanes)
                        float acc = __int_as_float(ra.y) * tp[0];
                        acc = __builtin_fmaf(__int_as_float(ra.z), tp[3], acc);
                        acc = __builtin_fmaf(__int_as_float(ra.w), tp[6], acc);
                        acc = __builtin_fmaf(__int_as_float(rb.x), tp[9], acc);
                        acc = __builtin_fmaf(__int_as_float(rb.y), tp[12], acc);
                        acc = __builtin_fmaf(__int_as_float(rb.z), tp[15], acc);
                        __builtin_amdgcn_wave_barrier();
                        if (gj < gn[g])
                            pyr[(frame_px0 + wp.px_off[g] + (long long)oy * wp.out_w[g] + gx0[g]) * 3 + lane] = acc;
                    });
                }
            }
        }
        slot = slot == kWalkSlots - 1 ? 0 : slot + 1;
    }
}

}  // namespace silent
