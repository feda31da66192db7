// gray_walk1_kernel: the whole gray pass of the unit level plus the pyramid of every other level, as a strip walk with a
// dedicated loader wave and ONE pixel per lane -- the occupancy-first sibling of gray_walk_kernel (silent_walk.h).
//
// What round 2 learned on the way here (profiles/r02/walk_kernel.txt, rgb_pyramid_walk.txt):
//   * a walk's consumers are latency-bound (LDS round trips, store issue): they need WAVES.  gray_walk_kernel (two pixels
//     per lane, 102-168 VGPRs, 48 KB ring) gets 8-12 consumer waves per CU and loses to the tile kernel's 20; the RGB pyramid
//     walk went from 0.87 to 0.56 ms by halving its ring (3 -> 4 blocks per CU);
//   * accumulating the other levels on every source row costs more than evaluating a level's row from a 6-row window when
//     it completes (every output row of every level is a 6-tap combination of 6 consecutive source rows).
// So: 56 output columns per consumer wave (64 lanes - 4 halo lanes per side, exactly gray_stream_kernel's pass 1, same
// arithmetic in the same order), a 12-row ring of 1 KB rows (ONE LDS-DMA instruction per row), column records in LDS instead
// of registers, ~95 VGPRs and ~22 KB of LDS per block: the 20 waves per CU of the tile kernel, without its row halo
// (24 streamed rows per 16 outputs -> seg_rows + 8 per seg_rows) and with the frame loads off the consumers' vmcnt queue.
// Horizontal gather of a completed row by ds_bpermute, like gray_stream_kernel.  Bit-identical to the tile path (tested).
#pragma once

#include <type_traits>

#include "silent_common.h"
#include "silent_conv.h"
#include "silent_walk.h"
#include "silent_walk_rgb.h"

namespace silent {

constexpr int kW1NC = 4;                         // consumer waves per block
constexpr int kW1Cols = 56;                      // output columns per consumer wave
constexpr int kW1StripW = kW1NC * kW1Cols;       // 224 output columns per block
constexpr int kW1RowF = 256;                     // floats per ring row (232 used): one global_load_lds_dwordx4 per row
constexpr int kW1CH = 4;                         // rows per chunk, 3 chunks in the ring
constexpr int kW1Threads = (kW1NC + 1) * 64;
// column records per wave tile and level (outputs anchored in a wave's 56 columns at zoom step >= 1.875 ^ (g + 1))
__host__ __device__ constexpr int w1_rec_cap(int g) { return (32 >> g) > 1 ? (32 >> g) : 1; }
__host__ __device__ constexpr int w1_rec_base(int g) {
    int n = 0;
    for (int i = 0; i < g; ++i) n += w1_rec_cap(i);
    return n;
}
__host__ __device__ constexpr int w1_rec_total(int g) { return w1_rec_base(g) > 0 ? w1_rec_base(g) : 1; }

template <int K, int G>
__global__ __launch_bounds__(kW1Threads) void gray_walk1_kernel(const float* __restrict__ frames, float* __restrict__ pyr,
                                                                float* __restrict__ cs_out, float* __restrict__ end_out,
                                                                const WalkTab tab, const WalkPyr wp, const GrayW wts,
                                                                float clip_hi) {
    static_assert(G == 0 || G == stream_pad_levels(G), "row programs are padded to 4 or 7 levels");
    constexpr int GG = G > 0 ? G : 1;
    constexpr int PR = G > 0 ? w3_prog_row(GG) : 4;              // the record format of silent_walk_rgb.h: flags + 6 weights per level
    __shared__ __attribute__((aligned(16))) float s_ring[kWalkSlots * kW1CH][kW1RowF];          // 12 KB
    __shared__ __attribute__((aligned(16))) int s_prog[G > 0 ? kWalkSlots * kW1CH * PR : 4];
    __shared__ __attribute__((aligned(16))) int s_rec[G > 0 ? kW1NC * w1_rec_total(GG) * 8 : 4];
    __shared__ __attribute__((aligned(16))) float s_slab[K == 8 ? kW1NC * 512 : 4];              // K = 8 store transpose, per wave

    const unsigned bid = blockIdx.x;
    const int strip = (int)(bid % (unsigned)tab.strips_x);
    const unsigned rest = bid / (unsigned)tab.strips_x;
    const int seg = (int)(rest % (unsigned)tab.segs_y);
    const int frame = (int)(rest / (unsigned)tab.segs_y);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg_y0 = seg * tab.seg_rows;
    const int seg_h = min(tab.seg_rows, tab.out_h - seg_y0);
    const int n_rows = seg_h + 8;                               // stream rows seg_y0 - 4 .. seg_y0 + seg_h + 3
    const int n_chunks = (n_rows + kW1CH - 1) / kW1CH;
    const int X0 = strip * kW1StripW;

    if (wave == kW1NC) {
        // ------------------------------------------------------------------ loader: LDS-DMA only
        const float* __restrict__ src = frames + (long long)frame * tab.H * tab.W;
        // ring row = columns X0 - 4 .. X0 + 251 of the crop; aligned 4-column groups lie wholly inside or outside the crop;
        // outside groups are clamped to a valid address and never read (the consumers read mirrored columns instead)
        const int c0 = min(max(X0 - 4 + lane * 4, 0), tab.src_w - 4) + tab.src_x0;
        auto issue = [&](int c, int slot) {
#pragma unroll
            for (int r = 0; r < kW1CH; ++r) {
                const int y = seg_y0 - 4 + c * kW1CH + r;
                const float* rp = src + (long long)(mirror_near(y, tab.src_h) + tab.src_y0) * tab.W;
                __builtin_amdgcn_global_load_lds((walk_glb_ptr)(rp + c0), (walk_lds_ptr)&s_ring[slot * kW1CH + r][0], 16, 0, 0);
            }
            if constexpr (G > 0) {
                const int* rp = wp.row_prog + ((long long)seg_y0 + (long long)c * kW1CH) * PR + lane * 4;
                if (lane < kW1CH * PR / 4)
                    __builtin_amdgcn_global_load_lds((walk_glb_ptr)rp, (walk_lds_ptr)(s_prog + slot * (kW1CH * PR)), 16, 0, 0);
            }
        };
        issue(0, 0);
        if (n_chunks > 1) issue(1, 1);
        int slot2 = 2;
        for (int c = 0; c < n_chunks; ++c) {
            if (c + 1 < n_chunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kW1CH + (G > 0 ? 1 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // barrier c: consumers are done with chunk c - 1
            if (c + 2 < n_chunks) issue(c + 2, slot2);
            slot2 = slot2 == kWalkSlots - 1 ? 0 : slot2 + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------- consumers (lane = column, halo 4 lanes per side)
    const int xw0 = X0 + wave * kW1Cols;
    const bool live = xw0 < tab.out_w;                          // wave-uniform; a dead wave still meets every barrier
    const int ox = xw0 + lane - 4;
    const int off = min(max(mirror_near(ox, tab.src_w) - (X0 - 4), 0), kW1RowF - 1);
    const bool col_eff = ox >= 0 && ox < tab.eff_w;             // inside the zoomed crop (zero outside it)
    const bool col_in = ox >= 0 && ox < tab.out_w;              // inside the level (zero padding of the convolutions)
    const bool out_lane = lane >= 4 && lane < 4 + kW1Cols && ox < tab.out_w;
    const long long frame_px0 = (long long)frame * tab.frame_px;
    const long long wave_px = frame_px0 + tab.px_off + (xw0 - 4);   // + row * out_w + lane

    // conv weights in VGPRs for K <= 4 (all-VGPR fmas issue faster than SGPR-operand ones, profiles/r01b/valu_rate.txt)
    constexpr bool VW = K <= 4;
    float wv[5], csw[9], endw[VW ? 9 * K : 1];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        wv[j] = tab.wx[j];
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

    // in-walk pyramid: this wave tile's column records staged in LDS, the 6-row window of the lane's column
    int gx0[GG], gn[GG];
    int* const my_rec = s_rec + wave * (w1_rec_total(GG) * 8);
    float hist[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (G > 0) {
        const int wx_tile = strip * kW1NC + wave;
        const int4* __restrict__ src4 = reinterpret_cast<const int4*>(wp.col_rec) + (long long)wx_tile * (w1_rec_total(G) * 2);
        int4* dst4 = reinterpret_cast<int4*>(my_rec);
        for (int i = lane; i < w1_rec_total(G) * 2; i += 64) dst4[i] = src4[i];
        typedef const __attribute__((address_space(4))) int* const_int_ptr;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int gg = min(g, wp.G - 1);
            const_int_ptr h = (const_int_ptr)(wp.col_hdr + ((long long)gg * (tab.strips_x * kW1NC) + wx_tile) * 2);
            gx0[g] = h[0];
            gn[g] = g < wp.G ? h[1] : 0;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // records staged (the only vector loads of a consumer)
        __builtin_amdgcn_wave_barrier();
    }

    float hw[5] = {0, 0, 0, 0, 0};
    float iw[3][3], cw[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) iw[a][b] = cw[a][b] = 0.0f;

    int slot = 0;
    for (int c = 0; c < n_chunks; ++c) {
        __builtin_amdgcn_s_barrier();                           // barrier c: chunk c (rows + records) is in the ring
        asm volatile("" ::: "memory");
        if (live) {
#pragma unroll
            for (int r = 0; r < kW1CH; ++r) {
                const int s = c * kW1CH + r;                    // stream row; source row y = seg_y0 - 4 + s
                if (s >= n_rows) break;                         // wave-uniform (padding of the last chunk)
                const float c0 = s_ring[slot * kW1CH + r][off];
                // ---- the other levels: 6-row window; a row of level g is evaluated when the record says it completes
                if constexpr (G > 0) {
                    const int* __restrict__ prow = s_prog + (slot * kW1CH + r) * PR;
                    int meta_v[(GG + 3) / 4 * 4];
#pragma unroll
                    for (int e = 0; e < (GG + 3) / 4; ++e) {
                        const int4 q = reinterpret_cast<const int4*>(prow)[e];   // every lane reads the same record (LDS broadcast)
                        meta_v[4 * e] = q.x; meta_v[4 * e + 1] = q.y; meta_v[4 * e + 2] = q.z; meta_v[4 * e + 3] = q.w;
                    }
#pragma unroll
                    for (int j = 0; j < 5; ++j) hist[j] = hist[j + 1];
                    hist[5] = c0;
                    const int anchor = seg_y0 + s - 7;          // anchor row of a row completing now: stored by the segment that owns it
                    if (anchor >= seg_y0 && anchor < seg_y0 + seg_h) {   // wave-uniform
                        walk_static_for<0, G>([&](auto gcst) {
                            constexpr int g = decltype(gcst)::value;
                            const int meta = __builtin_amdgcn_readfirstlane(meta_v[g]);
                            if (!(meta & 1)) return;            // wave-uniform: no row of level g completes here
                            const int oy = meta >> 8;
                            const int jj = min(lane, w1_rec_cap(g) - 1);
                            const int4* __restrict__ rc = reinterpret_cast<const int4*>(my_rec + (w1_rec_base(g) + jj) * 8);
                            const int4 ra = rc[0], rb = rc[1];
                            const float* __restrict__ wy = reinterpret_cast<const float*>(prow + G + 6 * g);   // 6 vertical weights
                            float v = __builtin_fmaf(wy[0], hist[0], 0.0f);
#pragma unroll
                            for (int j = 1; j < 6; ++j) v = __builtin_fmaf(wy[j], hist[j], v);
                            const int vbits = __float_as_int(v);
                            const int l4 = min(ra.x, 58) * 4;   // lane of tap 0 (idle lanes clamped), byte index for ds_bpermute
                            float acc = __int_as_float(ra.y) * __int_as_float(__builtin_amdgcn_ds_bpermute(l4, vbits));
                            acc = __builtin_fmaf(__int_as_float(ra.z), __int_as_float(__builtin_amdgcn_ds_bpermute(l4 + 4, vbits)), acc);
                            acc = __builtin_fmaf(__int_as_float(ra.w), __int_as_float(__builtin_amdgcn_ds_bpermute(l4 + 8, vbits)), acc);
                            acc = __builtin_fmaf(__int_as_float(rb.x), __int_as_float(__builtin_amdgcn_ds_bpermute(l4 + 12, vbits)), acc);
                            acc = __builtin_fmaf(__int_as_float(rb.y), __int_as_float(__builtin_amdgcn_ds_bpermute(l4 + 16, vbits)), acc);
                            acc = __builtin_fmaf(__int_as_float(rb.z), __int_as_float(__builtin_amdgcn_ds_bpermute(l4 + 20, vbits)), acc);
                            if (lane < gn[g]) pyr[frame_px0 + wp.px_off[g] + (long long)oy * wp.out_w[g] + gx0[g] + lane] = acc;
                        });
                    }
                }
                // ---- unit level: exactly pass 1 of gray_stream_kernel
                {
                    const float l1 = from_lane_below(c0), l2 = from_lane_below(l1);
                    const float r1 = from_lane_above(c0), r2 = from_lane_above(r1);
                    float h = wv[0] * l2;
                    h = __builtin_fmaf(wv[1], l1, h);
                    h = __builtin_fmaf(wv[2], c0, h);
                    h = __builtin_fmaf(wv[3], r1, h);
                    h = __builtin_fmaf(wv[4], r2, h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) hw[j] = hw[j + 1];
                    hw[4] = h;
                }
                const int p = seg_y0 + s - 6;                   // level-0 row completing with source row y = p + 2
                {
                    float v = wv[0] * hw[0];
#pragma unroll
                    for (int j = 1; j < 5; ++j) v = __builtin_fmaf(wv[j], hw[j], v);
                    v = (p >= 0 && p < tab.eff_h && col_eff) ? v : 0.0f;
                    if (p >= seg_y0 && p < seg_y0 + seg_h && out_lane)   // rows above the segment are warm-up
                        __builtin_nontemporal_store(v, pyr + (wave_px + (long long)p * tab.out_w) + lane);
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        iw[0][b] = iw[1][b];
                        iw[1][b] = iw[2][b];
                    }
                    iw[2][1] = v;
                    iw[2][0] = from_lane_below(v);
                    iw[2][2] = from_lane_above(v);
                }
                const int cr = p - 1;
                {
                    float acc = 0.0f;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) acc = __builtin_fmaf(iw[dy][dx], csw[dy * 3 + dx], acc);
                    // relu (a NaN stays a NaN) and the zero padding of the end convolution in one select
                    const float cs = (cr >= 0 && cr < tab.out_h && col_in && !(acc < 0.0f)) ? acc : 0.0f;
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        cw[0][b] = cw[1][b];
                        cw[1][b] = cw[2][b];
                    }
                    cw[2][1] = cs;
                    cw[2][0] = from_lane_below(cs);
                    cw[2][2] = from_lane_above(cs);
                }
                const int yo = p - 2;
                if (yo >= seg_y0 && yo < seg_y0 + seg_h) {      // wave-uniform
                    const long long row_px = wave_px + (long long)yo * tab.out_w;
                    if (cs_out && out_lane) __builtin_nontemporal_store(cw[1][1], cs_out + row_px + lane);
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
                            const int ncols = min(kW1Cols, tab.out_w - xw0);
                            store_row_k8(end_out + row_px * 8, acc, s_slab + wave * 512, lane, 4, ncols);
                        } else if constexpr (K == 4) {
                            if (out_lane) reinterpret_cast<float4*>(end_out + row_px * 4)[lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                        } else {
                            if (out_lane) {
#pragma unroll
                                for (int k = 0; k < K; ++k) end_out[(row_px + lane) * K + k] = acc[k];
                            }
                        }
                    }
                }
            }
        }
        slot = slot == kWalkSlots - 1 ? 0 : slot + 1;
    }
}

}  // namespace silent
