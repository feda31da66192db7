// gray_walk_kernel: the unit level of the gray pass (pyramid + CS + K-orientation line-end) as a STRIP WALK with a
// dedicated loader wave -- the round-2 answer to what round 1 measured on gray_stream_kernel:
//   * that kernel's time is the time of its write stream, and the write stream is slow because it leaves each wave as
//     224-byte (1-channel maps) / 896-byte (end maps) row fragments; a store-only kernel with 2 pixels per lane
//     (448 / 1792-byte runs) reaches the contiguous-fill rate (profiles/r01d/store_pattern.txt);
//   * every attempt to widen the runs inside the tile kernel lost the occupancy it needs to hide the frame loads:
//     loads and stores share one in-order vmcnt queue per wave, so a wave cannot prefetch behind its own stores.
// Here the two are separated.  A block owns a 480-column strip of one frame segment and walks DOWN it:
//   * wave 4 (the loader) does nothing but LDS-DMA (global_load_lds_dwordx4) of frame rows into a ring of 3 chunks x 8
//     rows, two chunks ahead; its vmcnt queue holds loads only, so `s_waitcnt vmcnt(16)` means "chunk c has landed";
//   * waves 0-3 (the consumers) own 120 output columns each, TWO ADJACENT PIXELS PER LANE (lane l = columns 2l, 2l+1 of
//     the wave's 128, halo 4 columns = 2 lanes per side), read their two values of a row from the ring and run the same
//     arithmetic, in the same order, as gray_unit_fused_kernel (bit-identical, tested): 5x5 smoother -> level 0 -> CS ->
//     ReLU -> end bank -> ReLU -> clip.  Their vmcnt queue holds stores only and is never waited on.
//   * one raw s_barrier per chunk hands chunk c to the consumers and the slot of chunk c-1 back to the loader.  Every
//     wave of the block executes exactly n_chunks barriers (uniform trip counts, no spinning: nothing can hang).
// Walking down removes the row halo of the tiles (24 streamed rows per 16 outputs -> seg_rows + 8 per seg_rows), two
// pixels per lane halve the DPP neighbour traffic and take the column halo from 64/56 to 128/120.
// Eligibility is host-checked (silent_api.hip, walk_plan): single-channel unit level, 16-byte aligned rows and crop
// (W, src_x0, out_w multiples of 4; even pyramid offsets), K = 4 or 8.  Everything else keeps gray_stream_kernel.
//
// The OTHER levels of the pyramid (template G > 0) come out of the same walk, in the consumers, with the arithmetic
// order of gray_stream_kernel's pass 2 and of the region kernel (vertical 6 taps in the lane, then horizontal 6 taps by
// gather: bit-identical, tested).  A consumer keeps, per level, up to walk_slots(g) output rows in flight for its two
// columns; a wave-uniform ROW PROGRAM (one record per source row of the level-0 crop, built by the host from the float64
// tap tables) says which slot takes which weight, which restarts, which completes.  The loader DMAs the records of a
// chunk into the ring beside its rows and the consumer reads them from LDS with the row's own values.  (Measured
// alternatives: records by scalar loads inside the consumers -- s_load shares lgkmcnt with the ring reads and the records
// overflowed the SGPR file, 2x slower; a sixth "pyramid wave" owning all 488 columns, 8 per lane -- one wave cannot keep
// up: ~350 instructions and ~5 LDS round trips per row against a budget of ~3000 cycles, kernel 0.73 -> 1.05 ms.)
// A completed row goes through a wave-private 128-float LDS line; the outputs ANCHORED in the wave's 120 columns (one per
// lane; host-checked <= 64 >> g, i.e. zoom steps >= 1.875 per level) gather their 6 taps from it, with column records (tap
// index + 6 weights) that the wave staged into LDS at its start.  A segment stores the rows whose anchor row it owns;
// their taps lie inside its streamed rows (4 halo rows per side).
#pragma once

#include <type_traits>

#include "silent_common.h"
#include "silent_conv.h"

namespace silent {

constexpr int kWalkNC = 4;                          // consumer waves per block
constexpr int kWalkCols = 120;                      // output columns per consumer wave
constexpr int kWalkStripW = kWalkNC * kWalkCols;    // 480 output columns per block
constexpr int kWalkRowF = 512;                      // floats per ring row (488 used: strip + 4 halo columns per side)
constexpr int kWalkCH = 8;                          // rows per chunk
constexpr int kWalkSlots = 3;                       // chunks in the ring
constexpr int kWalkLoadsPerChunk = 2 * kWalkCH;     // LDS-DMA instructions the loader issues per chunk (+1 with a pyramid wave)
constexpr int kWalkMaxOut = 64;                     // outputs of general level 0 anchored in a wave's 120 columns (one per lane)
__host__ __device__ constexpr int walk_threads(int) { return (kWalkNC + 1) * 64; }
// output rows of general level g in flight at once in the pyramid wave.  The walk only takes zoom steps >= 1.875 per
// level (host-checked through the slot-conflict test), so 3 / 2 / 1 rows suffice where gray_stream_kernel keeps 4 / 3 / 2.
__host__ __device__ constexpr int walk_slots(int g) { return g == 0 ? 3 : (g == 1 ? 2 : 1); }
constexpr int kWalkMaxSlots = 3;
// row record: [meta(0) .. meta(Gp-1)] [weights of level 0 (3)] [level 1 (2)] [level 2 (1)] ...; padded to 12 / 20 dwords
__host__ __device__ constexpr int walk_w_off(int gp, int g) {
    int o = gp;
    for (int h = 0; h < g; ++h) o += (h == 0 ? 3 : (h == 1 ? 2 : 1));
    return o;
}
__host__ __device__ constexpr int walk_prog_row(int gp) { return gp <= 4 ? 12 : 20; }   // >= walk_w_off(gp, gp), multiple of 4
// LDS column-record table of a consumer wave: level g holds max(kWalkMaxOut >> g, 1) slots of 8 dwords (level g has at
// most half the outputs of level g - 1 per wave tile; host-checked)
__host__ __device__ constexpr int walk_rec_cap(int g) { return (kWalkMaxOut >> g) > 1 ? (kWalkMaxOut >> g) : 1; }

// tables of the in-walk pyramid (device memory owned by the plan)
struct WalkPyr {
    int G;                        // general levels (<= template G; the rest are inert)
    const int* row_prog;          // [out_h + 8 (+ padding)][walk_prog_row(Gp)]: record of stream row y at index y + 4
    const int* col_hdr;           // [G][waves_x][2]: first output column, number of outputs of the 120-column wave tile
    const int* col_rec;           // [waves_x][walk_rec_total(Gp)][8]: index of tap 0 in the wave's 128 columns, 6 weights, pad
    long long px_off[8];          // pixel offset of level g inside one pyramid
    int out_w[8];
};

struct WalkTab {
    int H, W;                            // frame extents
    int src_y0, src_x0, src_h, src_w;    // crop the unit level resamples (zoom 1)
    int out_h, out_w, eff_h, eff_w;      // canvas, and the part of it the zoomed crop covers
    int strips_x, segs_y, seg_rows;      // decomposition: block = (frame, segment of seg_rows output rows, strip)
    long long frame_px, px_off;          // pixels of one pyramid, offset of the unit level in it
    float wx[5];                         // [1, 26, 66, 26, 1] / 120 as float32 (both axes)
};

// compile-time loop: the body gets its index as an integral constant (per-level array sizes in the pyramid wave)
template <int I, int N, class F>
__device__ __forceinline__ void walk_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        walk_static_for<I + 1, N>(f);
    }
}

typedef __attribute__((address_space(3))) void* walk_lds_ptr;
typedef const __attribute__((address_space(1))) void* walk_glb_ptr;

// total record slots of the LDS column table for a kernel compiled for G levels
__host__ __device__ constexpr int walk_rec_total(int g) {
    int n = 0;
    for (int i = 0; i < g; ++i) n += (kWalkMaxOut >> i) > 1 ? (kWalkMaxOut >> i) : 1;
    return n;
}
__host__ __device__ constexpr int walk_rec_base(int g) {
    int n = 0;
    for (int i = 0; i < g; ++i) n += (kWalkMaxOut >> i) > 1 ? (kWalkMaxOut >> i) : 1;
    return n;
}

// (amdgpu_waves_per_eu(3, ..): two 6-wave blocks per CU = 3 waves per SIMD = at most 168 VGPRs)
template <int K, bool NT, int G>
__global__ __launch_bounds__(walk_threads(G)) __attribute__((amdgpu_waves_per_eu(3, 8))) void gray_walk_kernel(const float* __restrict__ frames,
                                                                     float* __restrict__ pyr, float* __restrict__ cs_out,
                                                                     float* __restrict__ end_out, const WalkTab tab,
                                                                     const WalkPyr wp, const GrayW wts, float clip_hi) {
    static_assert(K == 4 || K == 8, "two-pixel store layouts exist for K = 4 and K = 8");
    static_assert(G == 0 || G == stream_pad_levels(G), "row programs are padded to 4 or 7 levels");
    constexpr int GG = G > 0 ? G : 1;
    constexpr int PR = G > 0 ? walk_prog_row(GG) : 4;           // dwords per row record
    __shared__ __attribute__((aligned(16))) float s_ring[kWalkSlots * kWalkCH][kWalkRowF];   // 48 KB
    __shared__ __attribute__((aligned(16))) float s_slab[K == 8 ? kWalkNC * 512 : 4];        // K = 8 store transpose
    __shared__ __attribute__((aligned(16))) int s_prog[G > 0 ? kWalkSlots * kWalkCH * PR : 4];   // row records of the ring's chunks
    __shared__ __attribute__((aligned(16))) float s_line[G > 0 ? kWalkNC * 128 : 4];              // a completed row of another level, per wave
    __shared__ __attribute__((aligned(16))) int s_rec[G > 0 ? kWalkNC * walk_rec_total(GG) * 8 : 4];   // column records, per wave

    const unsigned bid = blockIdx.x;
    const int strip = (int)(bid % (unsigned)tab.strips_x);
    const unsigned rest = bid / (unsigned)tab.strips_x;
    const int seg = (int)(rest % (unsigned)tab.segs_y);
    const int frame = (int)(rest / (unsigned)tab.segs_y);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg_y0 = seg * tab.seg_rows;
    const int seg_h = min(tab.seg_rows, tab.out_h - seg_y0);
    const int n_rows = seg_h + 8;                              // stream rows seg_y0 - 4 .. seg_y0 + seg_h + 3
    const int n_chunks = (n_rows + kWalkCH - 1) / kWalkCH;
    const int X0 = strip * kWalkStripW;

    if (wave == kWalkNC) {
        // ------------------------------------------------------------------ loader: LDS-DMA only
        const float* __restrict__ src = frames + (long long)frame * tab.H * tab.W;
        // ring row = columns X0 - 4 .. X0 + 507 of the crop, 16 bytes per lane and instruction; 4-column groups are
        // aligned (host-checked), so a group lies wholly inside the crop or wholly outside; outside groups are
        // clamped to a valid address and never read (the consumers read mirrored columns instead)
        const int c0 = min(max(X0 - 4 + lane * 4, 0), tab.src_w - 4) + tab.src_x0;
        const int c1 = min(max(X0 + 252 + lane * 4, 0), tab.src_w - 4) + tab.src_x0;
        auto issue = [&](int c, int slot) {
#pragma unroll
            for (int r = 0; r < kWalkCH; ++r) {
                const int y = seg_y0 - 4 + c * kWalkCH + r;     // rows past the segment (padding of the last chunk) clamp
                const float* rp = src + (long long)(mirror_near(y, tab.src_h) + tab.src_y0) * tab.W;
                float* dst = &s_ring[slot * kWalkCH + r][0];
                __builtin_amdgcn_global_load_lds((walk_glb_ptr)(rp + c0), (walk_lds_ptr)dst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((walk_glb_ptr)(rp + c1), (walk_lds_ptr)(dst + 256), 16, 0, 0);
            }
            if constexpr (G > 0) {
                // the chunk's kWalkCH row records (contiguous in the table: record of stream row s at (seg_y0 + s) * PR)
                const int* rp = wp.row_prog + ((long long)seg_y0 + (long long)c * kWalkCH) * PR + lane * 4;
                int* dst = s_prog + slot * (kWalkCH * PR);
                if (lane < kWalkCH * PR / 4)
                    __builtin_amdgcn_global_load_lds((walk_glb_ptr)rp, (walk_lds_ptr)dst, 16, 0, 0);
            }
        };
        issue(0, 0);
        if (n_chunks > 1) issue(1, 1);
        int slot2 = 2;                                          // slot of chunk c + 2
        for (int c = 0; c < n_chunks; ++c) {
            // chunks 0 .. c + 1 have been issued: leave only the newest one in flight -> chunk c has landed
            if (c + 1 < n_chunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kWalkLoadsPerChunk + (G > 0 ? 1 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // barrier c: consumers are done with chunk c - 1
            if (c + 2 < n_chunks) issue(c + 2, slot2);          // ... whose slot is the one chunk c + 2 goes to
            slot2 = slot2 == kWalkSlots - 1 ? 0 : slot2 + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int xw0 = X0 + wave * kWalkCols;                      // first output column of this wave
    const bool live = xw0 < tab.out_w;                          // wave-uniform; a dead wave still meets every barrier
    const int colA = xw0 - 4 + 2 * lane, colB = colA + 1;       // the lane's two columns (level = crop coordinates)
    // ring offsets of the (mirrored) columns: the ring row starts at column X0 - 4
    const int offA = min(max(mirror_near(colA, tab.src_w) - (X0 - 4), 0), kWalkRowF - 1);
    const int offB = min(max(mirror_near(colB, tab.src_w) - (X0 - 4), 0), kWalkRowF - 1);
    const bool effA = colA >= 0 && colA < tab.eff_w, effB = colB >= 0 && colB < tab.eff_w;
    const bool inA = colA >= 0 && colA < tab.out_w, inB = colB >= 0 && colB < tab.out_w;
    const bool out_lane = lane >= 2 && lane < 62 && colA < tab.out_w;   // out_w is even: both columns or neither
    const long long base_px = (long long)frame * tab.frame_px + tab.px_off;
    const long long lane_px = base_px + colA;                   // + row * out_w

    // conv weights in VGPRs (all-VGPR fmas issue at ~2.7 cycles, SGPR-operand ones at ~4.2: profiles/r01b/valu_rate.txt)
    constexpr bool VW = K <= 4 && G == 0;   // (with the in-walk pyramid state the 36 end weights would push the wave past 168 VGPRs)
    float wv[5], csw[9], endw[VW ? 9 * K : 1];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        wv[j] = tab.wx[j];
        asm volatile("" : "+v"(wv[j]));
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        csw[j] = wts.cs[j];
        asm volatile("" : "+v"(csw[j]));
    }
    if constexpr (VW) {
#pragma unroll
        for (int j = 0; j < 9 * K; ++j) {
            endw[j] = wts.end[j];
            asm volatile("" : "+v"(endw[j]));
        }
    }

    // ---- in-walk pyramid state (G > 0)
    float vacc[GG][kWalkMaxSlots][2];
    int gx0[GG], gn[GG];
    const long long frame_px0 = (long long)frame * tab.frame_px;
    int* const my_rec = s_rec + wave * (walk_rec_total(GG) * 8);
    float* const my_line = s_line + wave * 128;
    if constexpr (G > 0) {
        const int wx_tile = strip * kWalkNC + wave;
        // stage this wave tile's column records into LDS (wave-private: no block-level sync needed); these are the only
        // vector loads of a consumer and they precede all of its stores
        {
            const int4* __restrict__ src4 = reinterpret_cast<const int4*>(wp.col_rec) + (long long)wx_tile * (walk_rec_total(G) * 2);
            int4* dst4 = reinterpret_cast<int4*>(my_rec);
            for (int i = lane; i < walk_rec_total(G) * 2; i += 64) dst4[i] = src4[i];
        }
        typedef const __attribute__((address_space(4))) int* const_int_ptr;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int gg = min(g, wp.G - 1);
            const_int_ptr h = (const_int_ptr)(wp.col_hdr + ((long long)gg * (tab.strips_x * kWalkNC) + wx_tile) * 2);
            gx0[g] = h[0];
            gn[g] = g < wp.G ? h[1] : 0;
#pragma unroll
            for (int k = 0; k < kWalkMaxSlots; ++k) vacc[g][k][0] = vacc[g][k][1] = 0.0f;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // records staged
        __builtin_amdgcn_wave_barrier();
    }

    float hA[5] = {0, 0, 0, 0, 0}, hB[5] = {0, 0, 0, 0, 0};   // horizontally smoothed rows y-4 .. y of the two columns
    float iw[3][4], cw[3][4];                                   // level-0 rows / CS rows x columns (A-1, A, B, B+1)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) iw[a][b] = cw[a][b] = 0.0f;

    int slot = 0;
    for (int c = 0; c < n_chunks; ++c) {
        __builtin_amdgcn_s_barrier();                           // barrier c: chunk c is in the ring
        asm volatile("" ::: "memory");
        // One source row.  CHECK = false is the steady state (an INTERIOR chunk: every row of it is a stream row of the
        // segment whose level-0 / CS / output rows are all stored and inside the image): no row conditions at all, so the
        // 8 unrolled rows are one straight-line block and the register windows rotate by renaming instead of moves.
        auto do_row = [&](auto check_c, int r) {
            constexpr bool CHECK = decltype(check_c)::value;
            {
                const int s = c * kWalkCH + r;                  // stream row index; source row y = seg_y0 - 4 + s
                if (CHECK && s >= n_rows) return;               // wave-uniform (padding of the last chunk)
                const float* __restrict__ row = &s_ring[slot * kWalkCH + r][0];
                const float a = row[offA], b = row[offB];
                // ---- the other levels: vertical taps of this source row, gather + store of a row that completes
                if constexpr (G > 0) {
                    const int4* __restrict__ rec4 = reinterpret_cast<const int4*>(s_prog + (slot * kWalkCH + r) * PR);
                    int cur[PR];
#pragma unroll
                    for (int e = 0; e < PR / 4; ++e) {
                        const int4 t = rec4[e];                 // every lane reads the same record (LDS broadcast)
                        cur[4 * e] = t.x; cur[4 * e + 1] = t.y; cur[4 * e + 2] = t.z; cur[4 * e + 3] = t.w;
                    }
                    walk_static_for<0, G>([&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        const int meta = __builtin_amdgcn_readfirstlane(cur[g]);
                        if (!(meta & 128)) return;              // wave-uniform: this source row carries no tap of level g
#pragma unroll
                        for (int k = 0; k < walk_slots(g); ++k) {
                            const float w = __int_as_float(cur[walk_w_off(G, g) + k]);
                            const bool restart = (meta >> k) & 1;   // a slot that restarts accumulates onto +0
                            vacc[g][k][0] = __builtin_fmaf(w, a, restart ? 0.0f : vacc[g][k][0]);
                            vacc[g][k][1] = __builtin_fmaf(w, b, restart ? 0.0f : vacc[g][k][1]);
                        }
                        const int done = (meta >> 4) & 7;
                        // the row's anchor is source row y - 3 = seg_y0 + s - 7: stored by the segment that owns it
                        const int anchor = seg_y0 + s - 7;
                        if (done < kWalkMaxSlots && (!CHECK || (anchor >= seg_y0 && anchor < seg_y0 + seg_h))) {   // wave-uniform
                            const int oy = meta >> 8;
                            // the lane's column record is requested before the row is written: one LDS round trip less
                            const int jj = min(lane, walk_rec_cap(g) - 1);
                            const int4* __restrict__ rc = reinterpret_cast<const int4*>(my_rec + (walk_rec_base(g) + jj) * 8);
                            const int4 ra = rc[0], rb = rc[1];
                            float v0 = vacc[g][0][0], v1 = vacc[g][0][1];
#pragma unroll
                            for (int k = 1; k < walk_slots(g); ++k) {
                                v0 = done == k ? vacc[g][k][0] : v0;
                                v1 = done == k ? vacc[g][k][1] : v1;
                            }
                            typedef float nf2 __attribute__((ext_vector_type(2)));
                            *reinterpret_cast<nf2*>(my_line + 2 * lane) = nf2{v0, v1};
                            __builtin_amdgcn_wave_barrier();
                            const float* tp = my_line + ra.x;
                            float acc = __int_as_float(ra.y) * tp[0];
                            acc = __builtin_fmaf(__int_as_float(ra.z), tp[1], acc);
                            acc = __builtin_fmaf(__int_as_float(ra.w), tp[2], acc);
                            acc = __builtin_fmaf(__int_as_float(rb.x), tp[3], acc);
                            acc = __builtin_fmaf(__int_as_float(rb.y), tp[4], acc);
                            acc = __builtin_fmaf(__int_as_float(rb.z), tp[5], acc);
                            __builtin_amdgcn_wave_barrier();
                            if (lane < gn[g])
                                pyr[frame_px0 + wp.px_off[g] + (long long)oy * wp.out_w[g] + gx0[g] + lane] = acc;
                        }
                    });
                }
                // ---- horizontal 5 taps (same fma order as gray_unit_fused_kernel)
                {
                    const float La = from_lane_below(a), Lb = from_lane_below(b);
                    const float Ra = from_lane_above(a), Rb = from_lane_above(b);
                    float h0 = wv[0] * La;
                    h0 = __builtin_fmaf(wv[1], Lb, h0);
                    h0 = __builtin_fmaf(wv[2], a, h0);
                    h0 = __builtin_fmaf(wv[3], b, h0);
                    h0 = __builtin_fmaf(wv[4], Ra, h0);
                    float h1 = wv[0] * Lb;
                    h1 = __builtin_fmaf(wv[1], a, h1);
                    h1 = __builtin_fmaf(wv[2], b, h1);
                    h1 = __builtin_fmaf(wv[3], Ra, h1);
                    h1 = __builtin_fmaf(wv[4], Rb, h1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        hA[j] = hA[j + 1];
                        hB[j] = hB[j + 1];
                    }
                    hA[4] = h0;
                    hB[4] = h1;
                }
                // ---- level-0 row p = y - 2 (vertical 5 taps); rows above the segment's first are warm-up garbage that
                // is never stored and has left every window before the first stored row needs it
                const int p = seg_y0 + s - 6;
                {
                    float v0 = wv[0] * hA[0], v1 = wv[0] * hB[0];
#pragma unroll
                    for (int j = 1; j < 5; ++j) {
                        v0 = __builtin_fmaf(wv[j], hA[j], v0);
                        v1 = __builtin_fmaf(wv[j], hB[j], v1);
                    }
                    const bool prow = !CHECK || (p >= 0 && p < tab.eff_h);
                    v0 = (prow && effA) ? v0 : 0.0f;
                    v1 = (prow && effB) ? v1 : 0.0f;
                    if ((!CHECK || (p >= seg_y0 && p < seg_y0 + seg_h)) && out_lane) {
                        typedef float nf2 __attribute__((ext_vector_type(2)));
                        nf2* dst = reinterpret_cast<nf2*>(pyr + (lane_px + (long long)p * tab.out_w));
                        const nf2 v = {v0, v1};
                        if constexpr (NT) __builtin_nontemporal_store(v, dst);
                        else *dst = v;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        iw[0][q] = iw[1][q];
                        iw[1][q] = iw[2][q];
                    }
                    iw[2][1] = v0;
                    iw[2][2] = v1;
                    iw[2][0] = from_lane_below(v1);
                    iw[2][3] = from_lane_above(v0);
                }
                // ---- CS row cr = p - 1
                const int cr = p - 1;
                {
                    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            a0 = __builtin_fmaf(iw[dy][dx], csw[dy * 3 + dx], a0);
                            a1 = __builtin_fmaf(iw[dy][dx + 1], csw[dy * 3 + dx], a1);
                        }
                    const bool crow = !CHECK || (cr >= 0 && cr < tab.out_h);
                    // relu (a NaN stays a NaN) and the zero padding of the end convolution in one select
                    const float cs0 = (crow && inA && !(a0 < 0.0f)) ? a0 : 0.0f;
                    const float cs1 = (crow && inB && !(a1 < 0.0f)) ? a1 : 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        cw[0][q] = cw[1][q];
                        cw[1][q] = cw[2][q];
                    }
                    cw[2][1] = cs0;
                    cw[2][2] = cs1;
                    cw[2][0] = from_lane_below(cs1);
                    cw[2][3] = from_lane_above(cs0);
                }
                // ---- output row yo = p - 2
                const int yo = p - 2;
                if (!CHECK || (yo >= seg_y0 && yo < seg_y0 + seg_h)) {      // wave-uniform
                    const long long row_px = lane_px + (long long)yo * tab.out_w;
                    if (cs_out && out_lane) {
                        typedef float nf2 __attribute__((ext_vector_type(2)));
                        nf2* dst = reinterpret_cast<nf2*>(cs_out + row_px);
                        const nf2 v = {cw[1][1], cw[1][2]};
                        if constexpr (NT) __builtin_nontemporal_store(v, dst);
                        else *dst = v;
                    }
                    if (end_out) {
                        float e0[K], e1[K];
#pragma unroll
                        for (int k = 0; k < K; ++k) e0[k] = e1[k] = 0.0f;
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                                for (int k = 0; k < K; ++k) {
                                    const int wi = (dy * 3 + dx) * K + k;
                                    const float w = VW ? endw[VW ? wi : 0] : wts.end[wi];
                                    e0[k] = __builtin_fmaf(cw[dy][dx], w, e0[k]);
                                    e1[k] = __builtin_fmaf(cw[dy][dx + 1], w, e1[k]);
                                }
                        // relu + clip.  The reference's forms ((x < 0) ? 0 : x, (x > hi) ? hi : x) keep a NaN a NaN; as compare +
                        // select pairs they are 4 instructions per value, serialised through VCC with hazard nops (a quarter of
                        // the row's issue slots).  For every non-NaN x they equal the median of (x, 0, hi) -- one v_med3_f32 --
                        // when hi >= 0 (the chains start from +0, so x is never -0).  The 2K raw values are summed first: the sum is
                        // a NaN iff one of them is (or +inf meets -inf), and only then the wave takes the select form.
                        {
                            float chk = e0[0] + e1[0];
#pragma unroll
                            for (int k = 1; k < K; ++k) chk += e0[k] + e1[k];
                            if (clip_hi >= 0.0f && !__any(chk != chk)) {   // wave-uniform
#pragma unroll
                                for (int k = 0; k < K; ++k) {
                                    e0[k] = __builtin_amdgcn_fmed3f(e0[k], 0.0f, clip_hi);
                                    e1[k] = __builtin_amdgcn_fmed3f(e1[k], 0.0f, clip_hi);
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < K; ++k) {
                                    e0[k] = clip_hi_tf(relu_tf(e0[k]), clip_hi);
                                    e1[k] = clip_hi_tf(relu_tf(e1[k]), clip_hi);
                                }
                            }
                        }
                        typedef float nf4 __attribute__((ext_vector_type(4)));
                        if constexpr (K == 4) {
                            if (out_lane) {
                                nf4* dst = reinterpret_cast<nf4*>(end_out + row_px * 4);
                                dst[0] = nf4{e0[0], e0[1], e0[2], e0[3]};
                                dst[1] = nf4{e1[0], e1[1], e1[2], e1[3]};
                            }
                        } else {
                            // K = 8: the lane's two pixels are 64 contiguous bytes; four 16-byte stores per lane would each
                            // write 16-byte pieces at a 64-byte stride.  Transpose through a wave-private LDS slab instead:
                            // every store instruction writes 64 consecutive pieces of the row, 1 KiB contiguous,
                            // in two halves (lanes 0-31, then 32-63), so that the slab is 2 KB per wave.
                            nf4* slab = reinterpret_cast<nf4*>(s_slab + wave * 512);
                            const int ncols = min(kWalkCols, tab.out_w - xw0);
                            nf4* dst = reinterpret_cast<nf4*>(end_out + (base_px + (long long)yo * tab.out_w + (xw0 - 4)) * 8);
#pragma unroll
                            for (int hlf = 0; hlf < 2; ++hlf) {
                                if ((lane >> 5) == hlf) {
                                    const int l5 = lane & 31;
                                    slab[l5 * 4 + 0] = nf4{e0[0], e0[1], e0[2], e0[3]};
                                    slab[l5 * 4 + 1] = nf4{e0[4], e0[5], e0[6], e0[7]};
                                    slab[l5 * 4 + 2] = nf4{e1[0], e1[1], e1[2], e1[3]};
                                    slab[l5 * 4 + 3] = nf4{e1[4], e1[5], e1[6], e1[7]};
                                }
                                __builtin_amdgcn_wave_barrier();
                                const nf4 p0 = slab[lane], p1 = slab[64 + lane];
                                __builtin_amdgcn_wave_barrier();
                                // piece i of the row (16 bytes) belongs to column (xw0 - 4) + i / 2; this half holds pieces
                                // 128 hlf .. 128 hlf + 127; stored columns: xw0 .. xw0 + ncols - 1
                                const int i0 = hlf * 128 + lane, i1 = i0 + 64;
                                if ((i0 >> 1) >= 4 && (i0 >> 1) < 4 + ncols) dst[i0] = p0;
                                if ((i1 >> 1) >= 4 && (i1 >> 1) < 4 + ncols) dst[i1] = p1;
                            }
                        }
                    }
                }
            }
        };
        if (live) {
            // interior: rows s = 8c .. 8c + 7 all have s >= 8 (first stored output row reached), level-0 row p = seg_y0 + s - 6
            // below the segment's end and inside the zoomed crop
            const bool interior = c >= 1 && c * kWalkCH + 1 < seg_h && seg_y0 + c * kWalkCH + 1 < tab.eff_h;
            if (interior) {
#pragma unroll
                for (int r = 0; r < kWalkCH; ++r) do_row(std::false_type{}, r);
            } else {
#pragma unroll
                for (int r = 0; r < kWalkCH; ++r) do_row(std::true_type{}, r);
            }
        }
        slot = slot == kWalkSlots - 1 ? 0 : slot + 1;
    }
}

}  // namespace silent
