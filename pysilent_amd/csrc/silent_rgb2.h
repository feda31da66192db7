// rgb_line_end2_kernel: the fused RGB chain of silent_rgb.h with TWO ADJACENT PIXELS PER LANE on packed f32 instructions.
//
// Why (profiles/r02/pk_rate.txt): rgb_line_end_kernel is VALU-issue bound on fmas whose weight is an SGPR operand,
// 4.2 cycles per wave instruction.  v_pk_fma_f32 takes the same SGPR weight for BOTH halves (op_sel / op_sel_hi pick the
// low or the high dword of an aligned SGPR pair) at 4.4-4.7 cycles per instruction = 2.2-2.35 cycles per fma.  So a lane owns
// the pixel pair (x, x + 1) and every quantity of the chain is a register pair (lo = pixel x, hi = pixel x + 1):
//   * every fma chain of silent_rgb.h becomes the same chain of v_pk_fma_f32, each accumulator seeing its terms in the same
//     (dx, i) order: each half is the fmaf chain of its pixel, so the result is BIT-IDENTICAL to rgb_line_end_kernel
//     (tested on NaN / inf inputs, odd widths, both flat policies); pending rows of one output advance side by side
//     (independent accumulators back to back: a dependent packed fma costs a wait state);
//   * left / right neighbours: (below(hi), lo) and (hi, above(lo)) -- one DPP shift + one move per pair instead of two
//     DPP shifts per pixel; the wave covers 128 columns of which 112 produce outputs (halo 8 px = 4 lanes per side;
//     7 are needed): column redundancy 1.14x instead of 1.28x;
//   * weights: a STREAM in consumption order (below), read in 16-float blocks with one block of prefetch;
//   * memory: raw buffer loads / stores whose range check does every bounds test (per-level resource for the loads, per-tile
//     resources for the maps: zero padding / dropped store, no branch around a store so that the in-order vmcnt distance is
//     static), input rows fetched two steps ahead, output rows transposed through LDS so that every store instruction writes a
//     run of whole pixels, nt on the maps nobody on the GPU reads back;
//   * the three stripe rows that wait for their blur row live in LDS: 128 VGPRs, 4 waves / SIMD.
// All packed fmas are `asm volatile`: they stay in program order between the scalar loads around them, which is what the
// result-laundering did in silent_rgb.h.
// Round 3: the SYM instantiation (the symmetric forms of the reference's kernels: rgc folded over both mirror axes, rgby as
// channel mix -> one profile -> centre mix, left / right taps as swapped-half packed fmas on SGPR pairs, NaN-propagating
// v_maximum3 / v_minimum3 for relu and clip) -- the one tier that is NOT bit-identical to the one-pixel kernel.
// History and leave-one-out measurements: profiles/r02/rgb_pair_kernel.txt, profiles/r03_experiments.txt, DESIGN.md 4.5.
#pragma once

#include "silent_rgb.h"

namespace silent {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
typedef const __attribute__((address_space(4))) u64* ku64_p;

constexpr int kRgb2Halo = 8;                     // pixels per side (4 lanes); 7 are consumed by the five stages
constexpr int kRgb2Cols = 128 - 2 * kRgb2Halo;   // 112 output columns per wave
#ifndef RGB2_WAVES
#define RGB2_WAVES 2
#endif
constexpr int kRgb2Waves = RGB2_WAVES;          // waves side by side in a block
constexpr int kRgb2TW = kRgb2Waves * kRgb2Cols;
constexpr int kRgb2RowHalo = 7;                  // rows above / below a tile (same as silent_rgb.h)
#ifndef RGB2_STORE_AUX
#define RGB2_STORE_AUX 2
#endif
constexpr int kRgb2StoreAux = RGB2_STORE_AUX;      // 2 = nt: orient / line_end are results nobody on the GPU reads back
constexpr int kRgb2Out = 0x40000000;             // byte offset of "outside": lane + row parts add up to >= this; levels are smaller

// ---- the weight stream -------------------------------------------------------------------------------------------
// The kernel consumes its weights in ONE fixed order per row step, so the host lays them out in that order
// (rgb2_fill_stream below) and the kernel reads the stream in 16-float blocks
// (s_load_dwordx16 = 8 aligned SGPR pairs) into two alternating buffers: at the first use of block b the wave waits for it
// (lgkmcnt(0)) and only then requests block b + 1 into the other buffer, which is dead by then -- the request is in flight
// during the 16 packed fmas of block b (scalar loads return out of order: any later wait is lgkmcnt(0) and would otherwise
// wait for the prefetch too).  The stream is cyclic (the last block's first use requests block 0 of the next row step);
// the block count is even so that the buffer parity survives the wrap.
constexpr int kRgb2Blk = 16;
constexpr int kRgb2StreamMax = SILENT_RGB_STREAM_MAX;  // 4 x 81 + 49 = 373 floats for the dense instantiation, padded to whole block pairs

constexpr int rgb2_popc(unsigned v) { return v ? (int)(v & 1u) + rgb2_popc(v >> 1) : 0; }

// The "symmetric" forms (SYM instantiation; host-checked structure of the weights, silent_rgb_api.hip analyze_rgb_chain):
//   rgc   channel-diagonal, every channel's 3x3 kernel mirror-symmetric in both axes (midget_rgc: a centre-surround profile):
//         per channel [corner, edge_h (above / below the centre), edge_v (left / right), centre];
//   rgby  K[t][i][o] = S[t] * A[i][o] for the 8 taps around the centre, S mirror-symmetric in both axes, + B[i][o] at the centre
//         (rgby_3: one surround profile times one channel-mix matrix, another matrix at the centre): the channels are mixed FIRST
//         (Z_o = sum_i A[i][o] x_i, all nine terms: a zero entry still carries a NaN / inf on like the reference's 0 * x does),
//         then ONE symmetric 3x3 filter per output;
//   stripe (channel sum) in the paired form below.
struct RgbSym {
    float rgc[12];    // [channel][corner, edge_h, edge_v, centre]
    float rgby[27];   // A[i][o] at 0..8, S[dy][dx] at 9..17 (centre 0), B[i][o] at 18..26
};
constexpr int kSymRgc = 12, kSymRgby = 27, kSymStripe = 48;

// Stage sizes / bases in the stream for an instantiation
template <unsigned RGC_PAIRS, bool STRIPE_SUM, bool RGBY_TWO, bool END_TWO, bool SYM = false>
struct Rgb2Layout {
    static constexpr int n_rgc = SYM ? kSymRgc : 9 * rgb2_popc(RGC_PAIRS & 0x1ffu);
    static constexpr int n_rgby = SYM ? kSymRgby : RGBY_TWO ? 45 : 81;
    static constexpr int n_stripe = SYM ? kSymStripe : STRIPE_SUM ? 27 : 81;
    // the two-group instantiation also takes the blur in its MIRROR-SYMMETRIC form (host-checked, blur_tensor's profile is a
    // function of the distance): 16 weights w[|dy|][min(dx, 6 - dx)] instead of 49, see the blur stage of the kernel
    static constexpr bool blur_sym = RGBY_TWO && END_TWO;
    static constexpr int n_blur = blur_sym ? 16 : 49;
    static constexpr int n_end = END_TWO ? 45 : 81;
    // (SYM: the stripe block holds SGPR PAIRS (left weight, right weight) and starts on an even position)
    static constexpr int b_rgc = 0, b_rgby = b_rgc + n_rgc, b_stripe = (b_rgby + n_rgby + (SYM ? 1 : 0)) & ~(SYM ? 1 : 0),
                         b_blur = b_stripe + n_stripe, b_end = b_blur + n_blur, total = b_end + n_end;
    static constexpr int blocks = ((total + 2 * kRgb2Blk - 1) / (2 * kRgb2Blk)) * 2;
    static_assert(!SYM || (STRIPE_SUM && RGBY_TWO && END_TWO), "the symmetric forms extend the two-group instantiation");
};

// stream position of term (o, dy, dx, i) of a pair-masked 3x3x3->3 stage: for o: for active (dx, i): for dy = 2, 1, 0
constexpr int rgb2_conv_pos(unsigned pairs, int o, int dy, int dx, int i) {
    const unsigned row = (pairs >> (o * 3)) & 7u;
    const int before = 9 * rgb2_popc(pairs & ((1u << (o * 3)) - 1u));
    const int rank = dx * rgb2_popc(row) + rgb2_popc(row & ((1u << i) - 1u));
    return before + rank * 3 + (2 - dy);
}
// host side: fill the stream from the RgbW block (same enumeration as the kernel's); sym: the symmetric forms (two-group only)
inline int rgb2_fill_stream(const RgbW& w, unsigned rgc_pairs, bool stripe_sum, bool rgby_two, bool end_two, float* out,
                            const RgbSym* sym = nullptr) {
    const bool blur_sym = rgby_two && end_two;   // (Rgb2Layout::blur_sym; the host only picks the two-group form for a symmetric blur)
    int n = 0;
    auto conv = [&](const float* k, unsigned pairs) {  // k: [o][dy][dx][i]
        const int base = n;
        for (int o = 0; o < 3; ++o)
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx)
                    for (int i = 0; i < 3; ++i)
                        if ((pairs >> (o * 3 + i)) & 1u) {
                            out[base + rgb2_conv_pos(pairs, o, dy, dx, i)] = k[((o * 3 + dy) * 3 + dx) * 3 + i];
                            ++n;
                        }
    };
    auto two = [&](const float* k) {  // scale[dy][dx][i] at 0..26, mixA[i][o] 27.., mixB[i][o] 36..
        for (int j = 0; j < 9; ++j)
            for (int dy = 0; dy < 3; ++dy) out[n + j * 3 + (2 - dy)] = k[dy * 9 + j];
        for (int t = 0; t < 18; ++t) out[n + 27 + t] = k[27 + t];  // term t = group * 3 + i, then o: consumed (t, o) in this order
        n += 45;
    };
    if (sym) {
        // rgc: the three channels side by side: corner, edge_v (with the side sum), edge_h, centre (with the pixel itself)
        for (int c = 0; c < 3; ++c) {
            out[n + 0 + c] = sym->rgc[c * 4 + 0];
            out[n + 3 + c] = sym->rgc[c * 4 + 2];
            out[n + 6 + c] = sym->rgc[c * 4 + 1];
            out[n + 9 + c] = sym->rgc[c * 4 + 3];
        }
        n += kSymRgc;
        // rgby: A[i][o] (i outer), the profile's corner / edge_v / edge_h once per output, B[i][o] (i outer)
        for (int t = 0; t < 9; ++t) out[n + t] = sym->rgby[t];
        for (int o = 0; o < 3; ++o) {
            out[n + 9 + o] = sym->rgby[9 + 0];     // corner
            out[n + 12 + o] = sym->rgby[9 + 3];    // edge_v: left / right of the centre
            out[n + 15 + o] = sym->rgby[9 + 1];    // edge_h: above / below
        }
        for (int t = 0; t < 9; ++t) out[n + 18 + t] = sym->rgby[18 + t];
        n += kSymRgby;
        if (n & 1) out[n++] = 0.0f;
        // stripe of the channel sum, per output: (left, right) pairs of rows dy = 2, 1, 0; the three centres + a pad; (right, left) pairs
        for (int o = 0; o < 3; ++o) {
            for (int dy = 0; dy < 3; ++dy) {
                const float l = w.stripe[((o * 3 + dy) * 3 + 0) * 3], c = w.stripe[((o * 3 + dy) * 3 + 1) * 3], r = w.stripe[((o * 3 + dy) * 3 + 2) * 3];
                out[n + (2 - dy) * 2] = l;
                out[n + (2 - dy) * 2 + 1] = r;
                out[n + 6 + (2 - dy)] = c;
                out[n + 10 + (2 - dy) * 2] = r;
                out[n + 10 + (2 - dy) * 2 + 1] = l;
            }
            out[n + 9] = 0.0f;
            n += 16;
        }
    } else {
    conv(w.rgc, rgc_pairs & 0x1ffu);
    if (rgby_two) two(w.rgby);
    else conv(w.rgby, 0x1ffu);
    if (stripe_sum) {
        for (int o = 0; o < 3; ++o)
            for (int dx = 0; dx < 3; ++dx)
                for (int dy = 0; dy < 3; ++dy) out[n + o * 9 + dx * 3 + (2 - dy)] = w.stripe[((o * 3 + dy) * 3 + dx) * 3];
        n += 27;
    } else {
        conv(w.stripe, 0x1ffu);
    }
    }
    if (blur_sym) {   // for folded column j = 0..3 (dx = j and 6 - j): for d = |dy| = 0..3
        for (int j = 0; j < 4; ++j)
            for (int d = 0; d < 4; ++d) out[n + j * 4 + d] = w.blur[(3 + d) * 7 + j];
        n += 16;
    } else {
        for (int dx = 0; dx < 7; ++dx)
            for (int k = 0; k < 7; ++k) out[n + dx * 7 + k] = w.blur[(6 - k) * 7 + dx];
        n += 49;
    }
    if (end_two) two(w.end);
    else conv(w.end, 0x1ffu);
    const int total = n;
    while (n % (2 * kRgb2Blk)) out[n++] = 0.0f;
    return total;
}

struct Rgb2Args {
    const float* pyr;
    float* orient_out;
    float* line_out;
    float* value_out;
    LevelTab tab;
    RgbP prm;
    int th;   // output rows per tile (even)
    unsigned* mm;   // optional [n_frames][n_levels][2] ordered-uint slots (silent_peaks.h): max_pool(value), max_pool(-value) per level
    // optional value summary for the sparse selection tail (silent_peaks.h, SumTab): per frame and level a [tiles_y * gpt][ceil(W / 2)]
    // array, entry = max_pool(value) over one lane's pixel pair x kSumRows rows (MM instantiation only)
    float* sum;
    int* nan_flags;                    // with sum: [n_frames][n_levels], set to 1 when a value of that level is a NaN (zero-initialised)
    long long sum_frame;               // entries per frame
    long long sum_off[kMaxLevels];     // entry offset of level l inside a frame
    alignas(64) float ws[kRgb2StreamMax];
};
static_assert(offsetof(Rgb2Args, ws) % 64 == 0, "weight blocks are whole 64-byte lines of the kernarg segment");

template <int NP>
struct WPairs {
    u64 p[NP];
};
__device__ __forceinline__ void wait_weights() { __builtin_amdgcn_s_waitcnt(0xC07F); }  // lgkmcnt(0)

template <int NB>
struct WStream {
    WPairs<kRgb2Blk / 2> A, B;
    ku64_p wp;
    template <int BLK>
    __device__ __forceinline__ void request() {
        ku64_p q = wp;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" : "+s"(q));
        if constexpr (BLK & 1) {
#pragma unroll
            for (int k = 0; k < kRgb2Blk / 2; ++k) B.p[k] = q[BLK * (kRgb2Blk / 2) + k];
        } else {
#pragma unroll
            for (int k = 0; k < kRgb2Blk / 2; ++k) A.p[k] = q[BLK * (kRgb2Blk / 2) + k];
        }
    }
    // the SGPR pair that holds stream float IDX; positions must be visited in increasing order
    template <int IDX>
    __device__ __forceinline__ u64 pair() {
        constexpr int blk = IDX / kRgb2Blk;
        if constexpr (IDX % kRgb2Blk == 0) {
            wait_weights();
            request<(blk + 1) % NB>();
        }
        if constexpr (blk & 1) return B.p[(IDX % kRgb2Blk) / 2];
        else return A.p[(IDX % kRgb2Blk) / 2];
    }
    // x * w[IDX] + c  (each half is fmaf(x, w, c) of its pixel)
    template <int IDX>
    __device__ __forceinline__ f2 fma(const f2& x, const f2& c) {
        const u64 w = pair<IDX>();
        f2 r;
        if constexpr ((IDX & 1) == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "s"(w), "v"(c));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "s"(w), "v"(c));
        return r;
    }
    // x * w[IDX] + 0: the first term of a chain that starts from +0
    template <int IDX>
    __device__ __forceinline__ f2 fma0(const f2& x) {
        const u64 w = pair<IDX>();
        f2 r;
        if constexpr ((IDX & 1) == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(x), "s"(w));
        else asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(x), "s"(w));
        return r;
    }
    // x * w[IDX]
    template <int IDX>
    __device__ __forceinline__ f2 mul(const f2& x) {
        const u64 w = pair<IDX>();
        f2 r;
        if constexpr ((IDX & 1) == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(x), "s"(w));
        else asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(x), "s"(w));
        return r;
    }
    // The mov-free neighbour forms (SYM instantiation).  IDX even: the SGPR pair (w[IDX], w[IDX + 1]) as it lies, the halves of x
    // SWAPPED: lo = x.hi * w[IDX] + c.lo, hi = x.lo * w[IDX + 1] + c.hi.  With x = (above(c.x), below(c.y)) and the pair
    // (left weight, right weight) that is the left tap of pixel lo and the right tap of pixel hi in ONE instruction; with x = the
    // pixel pair itself and the pair (right weight, left weight) the right tap of lo (its neighbour hi) and the left tap of hi.
    template <int IDX>
    __device__ __forceinline__ f2 fma_sw(const f2& x, const f2& c) {
        static_assert((IDX & 1) == 0, "an aligned SGPR pair");
        const u64 w = pair<IDX>();
        f2 r;
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "s"(w), "v"(c));
        return r;
    }
    template <int IDX>
    __device__ __forceinline__ f2 fma0_sw(const f2& x) {
        static_assert((IDX & 1) == 0, "an aligned SGPR pair");
        const u64 w = pair<IDX>();
        f2 r;
        asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(x), "s"(w));
        return r;
    }
    // skip stream positions [FROM, TO) that carry no fma (padding): still crosses block boundaries in order
    template <int FROM, int TO>
    __device__ __forceinline__ void skip() {
        if constexpr (FROM < TO) {
            constexpr int nb = (FROM + kRgb2Blk - 1) / kRgb2Blk * kRgb2Blk;  // next block start at or after FROM
            if constexpr (nb < TO) {
                (void)pair<nb>();
                skip<nb + 1, TO>();
            }
        }
    }
};

template <int I>
using ic = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void rgb2_for(F&& f) {
    if constexpr (B < E) {
        f(ic<B>{});
        rgb2_for<B + 1, E>(f);
    }
}

// l = left neighbours, r = right neighbours of the pixel pair c
__device__ __forceinline__ void neighbours2(const f2& c, f2& l, f2& r) {
    l.x = from_lane_below(c.y);
    l.y = c.x;
    r.x = c.y;
    r.y = from_lane_above(c.x);
}

// The same neighbours WITHOUT the two moves: packed f32 instructions pick either dword of a VGPR pair for either half (op_sel /
// op_sel_hi), so with side = (above(c.x), below(c.y)) -- two DPP writes into one pair --
//   l + r       = (side.hi + c.hi, side.lo + c.lo)                      one v_pk_add_f32 with both operands' halves swapped
//   wl l + wr r = fma_sw(side, (wl, wr)) then fma_sw(c, (wr, wl))       (WStream::fma_sw)
__device__ __forceinline__ f2 side_pair(const f2& c) { return f2{from_lane_above(c.x), from_lane_below(c.y)}; }
__device__ __forceinline__ f2 pk_add_swap_both(const f2& a, const f2& b) {   // (a.hi + b.hi, a.lo + b.lo)
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 pk_add_swap_second(const f2& a, const f2& b) {   // (a.lo + b.hi, a.hi + b.lo)
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// clip(relu(v), hi) with the reference's NaN behaviour in two instructions: gfx950's v_maximum3_f32 / v_minimum3_f32 are the
// IEEE-754-2019 maximum / minimum -- a NaN operand comes back as it is, sign and payload (scripts/ubench/max3_nan.hip).  Against
// (v < 0 ? 0 : v) the only difference is -0 -> +0.
__device__ __forceinline__ float relu_clip_max3(float v, float hi) {
    float r;
    asm("v_maximum3_f32 %0, %1, 0, 0" : "=v"(r) : "v"(v));
    asm("v_minimum3_f32 %0, %1, %2, %2" : "=v"(r) : "v"(r), "s"(hi));
    return r;
}

// One arriving row of a dense / pair-masked 3x3 x 3->3 convolution (conv3_roll of silent_rgb.h on pixel pairs).  The three
// pending rows of an output advance side by side (three independent accumulators: no back-to-back dependent packed fmas);
// each accumulator still sees its terms in (dx, i) order.
template <unsigned PAIRS, int BASE, class WS>
__device__ __forceinline__ void conv3_roll2(const f2 (&v)[3][3], WS& ws, f2 (&pa)[3], f2 (&pb)[3], f2 (&done)[3]) {
    f2 npa[3], npb[3];
    rgb2_for<0, 3>([&](auto oo) {
        constexpr int o = decltype(oo)::value;
        constexpr unsigned row = (PAIRS >> (o * 3)) & 7u;
        f2 t2 = pa[o], t1 = pb[o], t0 = f2{0.0f, 0.0f};
        if constexpr (row != 0) {
            constexpr int first_i = (row & 1u) ? 0 : ((row & 2u) ? 1 : 2);
            rgb2_for<0, 9>([&](auto jj) {
                constexpr int j = decltype(jj)::value, dx = j / 3, i = j % 3;
                if constexpr ((row >> i) & 1u) {
                    constexpr int p2 = BASE + rgb2_conv_pos(PAIRS, o, 2, dx, i);
                    t2 = ws.template fma<p2>(v[dx][i], t2);
                    t1 = ws.template fma<p2 + 1>(v[dx][i], t1);
                    if constexpr (dx == 0 && i == first_i) t0 = ws.template fma0<p2 + 2>(v[dx][i]);
                    else t0 = ws.template fma<p2 + 2>(v[dx][i], t0);
                }
            });
        }
        done[o] = t2;
        npa[o] = t1;
        npb[o] = t0;
    });
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        pa[o] = npa[o];
        pb[o] = npb[o];
    }
}

// Two-group kernel (conv3_roll_struct of silent_rgb.h on pixel pairs): pa / pb = [0..2] group A per input, [3..5] group B.
template <unsigned M0, unsigned M1, unsigned M2, int BASE, class WS>
__device__ __forceinline__ void conv3_roll2_struct(const f2 (&v)[3][3], WS& ws, f2 (&pa)[6], f2 (&pb)[6], f2 (&done)[3]) {
    constexpr unsigned M[3] = {M0, M1, M2};
    f2 acc2[6], acc1[6], acc0[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        acc2[q] = pa[q];
        acc1[q] = pb[q];
        acc0[q] = f2{0.0f, 0.0f};
    }
    rgb2_for<0, 9>([&](auto jj) {
        constexpr int j = decltype(jj)::value, dx = j / 3, i = j % 3;
        constexpr int p = BASE + j * 3;
        constexpr int q2 = ((M[i] >> (2 * 3 + dx)) & 1u) ? i : 3 + i;
        constexpr int q1 = ((M[i] >> (1 * 3 + dx)) & 1u) ? i : 3 + i;
        constexpr bool a0 = (M[i] >> dx) & 1u;
        constexpr int q0 = a0 ? i : 3 + i;
        acc2[q2] = ws.template fma<p>(v[dx][i], acc2[q2]);
        acc1[q1] = ws.template fma<p + 1>(v[dx][i], acc1[q1]);
        // first tap of (q0, dy = 0) in (dx, i) order starts from +0
        constexpr unsigned taps = M[i] & 7u;
        constexpr bool seen = (dx >= 1 && (((taps >> 0) & 1u) != 0) == a0) || (dx >= 2 && (((taps >> 1) & 1u) != 0) == a0);
        if constexpr (!seen) acc0[q0] = ws.template fma0<p + 2>(v[dx][i]);
        else acc0[q0] = ws.template fma<p + 2>(v[dx][i], acc0[q0]);
    });
    // the completed row: out[o] = sum_i mixA[i][o] * A_i + mixB[i][o] * B_i, the three outputs side by side
    {
        constexpr int PM = BASE + 27;
        f2 t[3];
        rgb2_for<0, 3>([&](auto oo) {
            constexpr int o = decltype(oo)::value;
            t[o] = ws.template mul<PM + o>(acc2[0]);
        });
        rgb2_for<1, 6>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            rgb2_for<0, 3>([&](auto oo) {
                constexpr int o = decltype(oo)::value;
                t[o] = ws.template fma<PM + k * 3 + o>(acc2[k], t[o]);
            });
        });
#pragma unroll
        for (int o = 0; o < 3; ++o) done[o] = t[o];
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        pa[q] = acc1[q];
        pb[q] = acc0[q];
    }
}

// Channel-sum kernel (conv3_roll_sum of silent_rgb.h on pixel pairs): s[dx] = channel sum at x-1, x, x+1.
template <int BASE, class WS>
__device__ __forceinline__ void conv3_roll2_sum(const f2 (&s)[3], WS& ws, f2 (&pa)[3], f2 (&pb)[3], f2 (&done)[3]) {
    f2 npa[3], npb[3];
    rgb2_for<0, 3>([&](auto oo) {
        constexpr int o = decltype(oo)::value;
        f2 t2 = pa[o], t1 = pb[o], t0;
        rgb2_for<0, 3>([&](auto xx) {
            constexpr int dx = decltype(xx)::value, p = BASE + o * 9 + dx * 3;
            t2 = ws.template fma<p>(s[dx], t2);
            t1 = ws.template fma<p + 1>(s[dx], t1);
            if constexpr (dx == 0) t0 = ws.template fma0<p + 2>(s[dx]);
            else t0 = ws.template fma<p + 2>(s[dx], t0);
        });
        done[o] = t2;
        npa[o] = t1;
        npb[o] = t0;
    });
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        pa[o] = npa[o];
        pb[o] = npb[o];
    }
}

__device__ __forceinline__ float relu_ok(float v, bool ok) { return ok ? relu_tf(v) : 0.0f; }
__device__ __forceinline__ float relu_max3(float v) {   // relu_tf in one instruction (see relu_clip_max3)
    float r;
    asm("v_maximum3_f32 %0, %1, 0, 0" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ float relu_max3_poisoned(float v, float u) {   // u: +0, or a NaN that the result must become
    float r;
    asm("v_maximum3_f32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(u));
    return r;
}

// MM: also accumulate the per-level extrema of the value map (args.mm) -- a separate instantiation with a 3-waves/SIMD register
// budget: the plain kernel sits at the 128-VGPR edge, and forced into that budget the extra pointer, masks and accumulators
// spill to scratch (+16 %; with 3 waves +4 %)
// ST4 (round 4, opt-in by RGB knob bit 7): orient / line_end leave as 16-byte-per-lane stores (buffer_store_dwordx4), three per
// PAIR of rows instead of four 12-byte ones -- see the store section below; host-checked: every level's width, pixel offset and
// the map pointers are multiples of 4 pixels / 16 bytes (all BASELINE extents are).  Bit-identical and, measured, exactly as
// fast as the 12-byte form (0.806 vs 0.804 ms per config-3 launch): not the default.
template <unsigned RGC_PAIRS, bool STRIPE_SUM, unsigned RGBY_A, unsigned END_A0, unsigned END_A1, unsigned END_A2, bool MM = false, bool SYM = false,
          bool ST4 = false>
__global__ __launch_bounds__(64 * kRgb2Waves) __attribute__((amdgpu_waves_per_eu((MM || ST4) ? 3 : 4, 8))) void rgb_line_end2_kernel(const Rgb2Args args) {
    typedef Rgb2Layout<RGC_PAIRS, STRIPE_SUM, RGBY_A != kDense, END_A0 != kDense, SYM> L;
    static_assert(L::blocks * kRgb2Blk <= kRgb2StreamMax, "stream fits its kernarg array");
    static_assert((RGC_PAIRS & 0x1ffu) == 0x1ffu || (RGC_PAIRS & 0x1ffu) == 0x111u, "rgc: dense or channel-diagonal");
    constexpr bool kRgcDiag = (RGC_PAIRS & 0x1ffu) == 0x111u;   // zero weights skipped: the outputs are poisoned (rgc stage)
    const int R = args.th, NROWS = R + 2 * kRgb2RowHalo;
    const float* __restrict__ pyr = args.pyr;
    float* __restrict__ orient_out = args.orient_out;
    float* __restrict__ line_out = args.line_out;
    float* __restrict__ value_out = args.value_out;
    const LevelTab& tab = args.tab;
    const RgbP& prm = args.prm;

    typedef const __attribute__((address_space(4))) char* kchar_p;
    WStream<L::blocks> ws;
    ws.wp = (ku64_p)((kchar_p)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(Rgb2Args, ws));
    ws.template request<0>();

    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const float* __restrict__ src = pyr + base_px * 3;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = tc.tx * kRgb2TW + wave * kRgb2Cols;
    if (xw0 >= W) return;  // wave-uniform
    const int y0 = tc.ty * R;
    const int x0 = xw0 + 2 * lane - kRgb2Halo, x1 = x0 + 1;  // the lane's pixel pair
    const bool col0 = x0 >= 0 && x0 < W, col1 = x1 >= 0 && x1 < W;
    const bool out_lane = lane >= kRgb2Halo / 2 && lane < 64 - kRgb2Halo / 2;
    const bool out0 = out_lane && x0 < W, out1 = out_lane && x1 < W;
    // All pixel traffic goes through RAW BUFFER instructions whose range check does the bounds work: the pyramid is read through a
    // per-LEVEL resource (base = this frame's level, num_records = its bytes) with the byte offset row * row bytes + lane part --
    // a row above the level makes it negative (= huge as the unsigned number the hardware compares), a row below it or a lane
    // outside (lane part kRgb2Out) makes it >= num_records: a load of 0, the zero padding of the SAME convolutions.  The maps are
    // written through per-TILE resources (base = the tile's first row, num_records = its rows), offsets relative to that row: the
    // rows of the pipeline's fill and drain (above / below the tile) and the lanes outside are dropped stores.  So there is no
    // clamping, no select, no scalar row test and, above all, NO BRANCH around a store: every step issues the same 2 loads + 6
    // stores, the compiler knows the in-order vmcnt distance exactly, and the wait for a row fetched two steps ago does not wait
    // for the stores issued since (with exec-masked store blocks it cannot know how many were issued and waits for all of them:
    // 1.06 ms, 0.83 without stores).  Offsets are computed in unsigned arithmetic; the host keeps (H + 16) rows below kRgb2Out
    // and (H + tile height + 16) rows below 2^32 - kRgb2Out, so no sum wraps back into range.
    const int rb = W * 12;                                   // bytes per row of a 3-channel map
    const int rows_t = R < H - y0 ? R : H - y0;             // output rows of this tile (>= 1)
    const unsigned lvl_bytes = (unsigned)H * (unsigned)rb;
    const unsigned tile_bytes = (unsigned)rows_t * (unsigned)rb;
    const long long tile_px = base_px + (long long)y0 * W;
    const __amdgpu_buffer_rsrc_t r_src = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, lvl_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_orient = __builtin_amdgcn_make_buffer_rsrc((void*)(orient_out + tile_px * 3), 0, orient_out ? tile_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_line = __builtin_amdgcn_make_buffer_rsrc((void*)(line_out + tile_px * 3), 0, line_out ? tile_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_value = __builtin_amdgcn_make_buffer_rsrc((void*)(value_out + tile_px), 0, value_out ? tile_bytes / 3u : 0u, 0x00020000);
    const int in0 = col0 ? x0 * 12 : kRgb2Out, in1 = col1 ? x1 * 12 : kRgb2Out;     // byte offsets inside a row, or out of range
    // Stores: a lane's own pixel pair would make every x3 store instruction write 12 of every 24 bytes (its partner instruction
    // the other 12) -- measured 15 % slower than runs of whole pixels per instruction (profiles/r02/rgb_pair_kernel.txt).  So an
    // output row goes through a per-wave LDS row (s_tr: written as the lanes hold it, read back pixel-major): store A writes
    // pixels 0..63 of the wave's 128 columns (768 contiguous bytes), store B pixels 64..127.
    const int xa = xw0 - kRgb2Halo + lane, xb = xa + 64;
    const int sta = (lane >= kRgb2Halo && xa < W) ? xa * 12 : kRgb2Out, stb = (lane < 64 - kRgb2Halo && xb < W) ? xb * 12 : kRgb2Out;
    // value: the lane's two floats are adjacent in memory: one 8-byte store where both pixels exist, a 4-byte one for a last odd column
    const int sv2 = (out0 && out1) ? x0 * 4 : kRgb2Out, sv1 = (out0 && !out1) ? x0 * 4 : kRgb2Out;
    const bool padc0 = x0 >= prm.pad && x0 < W - prm.pad, padc1 = x1 >= prm.pad && x1 < W - prm.pad;
    const f2 inv3p = {1.0f / 3.0f, 1.0f / 3.0f};
    const f2 zero2 = {0.0f, 0.0f};

    // rolling state (pairs)
    f2 a1[3] = {zero2, zero2, zero2}, b1[3] = {zero2, zero2, zero2};
    f2 a2[6] = {zero2, zero2, zero2, zero2, zero2, zero2}, b2[6] = {zero2, zero2, zero2, zero2, zero2, zero2};
    f2 a3[3] = {zero2, zero2, zero2}, b3[3] = {zero2, zero2, zero2};
    f2 a5[6] = {zero2, zero2, zero2, zero2, zero2, zero2}, b5[6] = {zero2, zero2, zero2, zero2, zero2, zero2};
    f2 pb[7] = {zero2, zero2, zero2, zero2, zero2, zero2, zero2};
    // The stripe rows wait three steps for their blur row: a 4-slot delay line per wave in LDS (each lane reads back what it
    // wrote itself: no barrier) instead of 24 VGPRs -- with them the kernel is over 128 registers, i.e. 3 instead of 4 waves/SIMD.
    __shared__ f2 s_hist[kRgb2Waves][4][3][64];   // [wave][slot = row & 3][channel][lane]
    // MM: per-lane running max_pool(value) / max_pool(-value) of this tile, for the per-level extrema of a-10 (args.mm)
    float mm_mx = kPoolLowest, mm_nmn = kPoolLowest;
    // MM: the same maximum per group of kSumRows output rows, stored per lane (= per pixel pair) when the group ends: the sparse
    // selection tail looks only at the groups whose maximum reaches the level's threshold (silent_peaks.h, sparse_select_kernel).
    // A raw buffer store like all the others: out of range (dropped) on every step but a group's last.
    float mm_grp = kPoolLowest;
    unsigned long long mm_nan = 0;   // lanes that have seen a NaN value (scalar registers)
    const int gpt = (R + kSumRows - 1) / kSumRows, nxp = (W + 1) >> 1;
    const unsigned sum_bytes = (MM && args.sum) ? (unsigned)((rows_t + kSumRows - 1) / kSumRows) * (unsigned)nxp * 4u : 0u;   // this tile's groups
    const __amdgpu_buffer_rsrc_t r_sum = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(args.sum + (long long)tc.frame * args.sum_frame + args.sum_off[tc.level] + (long long)tc.ty * gpt * nxp), 0, sum_bytes, 0x00020000);
    const int ssum = out0 ? (x0 >> 1) * 4 : kRgb2Out;
    // [wave][one output row of the wave's 128 columns x 3 channels]; ST4: [wave][map: orient, line_end][row A, row B][384]
    __shared__ __attribute__((aligned(16))) float s_tr[kRgb2Waves][ST4 ? 4 * 384 : 384];
    float* const tr = s_tr[wave];
    // lane-major in AS HELD -- [lane][channel][half]: three 8-byte writes of the register pairs, no shuffling moves --, pixel-major
    // out: pixel p of the wave's 128 columns is half p & 1 of lane p >> 1, its channels 2 floats apart.  The reading lane j takes
    // pixel j (store A) and pixel 64 + j (store B): float addresses 6 (j >> 1) + (j & 1) + {0, 2, 4} hit 64 different banks.  LDS
    // operations of one wave execute in order, so no wait / barrier in between.
    typedef int i3 __attribute__((ext_vector_type(3)));
    f2* const tr_in = reinterpret_cast<f2*>(tr) + lane * 3;
    const float* const tr_out = tr + (lane >> 1) * 6 + (lane & 1);
    auto store_row3 = [&](const f2 (&val)[3], __amdgpu_buffer_rsrc_t rsrc, int ro) {
#pragma unroll
        for (int c = 0; c < 3; ++c) tr_in[c] = val[c];
        const i3 da = {__float_as_int(tr_out[0]), __float_as_int(tr_out[2]), __float_as_int(tr_out[4])};
        const i3 db = {__float_as_int(tr_out[192]), __float_as_int(tr_out[194]), __float_as_int(tr_out[196])};
        __builtin_amdgcn_raw_buffer_store_b96(da, rsrc, (int)((unsigned)sta + (unsigned)ro), 0, kRgb2StoreAux);
        __builtin_amdgcn_raw_buffer_store_b96(db, rsrc, (int)((unsigned)stb + (unsigned)ro), 0, kRgb2StoreAux);
    };
    // ST4: a row of the wave's 128 columns is 1536 bytes = 96 sixteen-byte chunks, of which chunks 6 .. 89 (the 112 output columns;
    // 4 pixels = 3 chunks and everything is a multiple of 4 pixels, so no chunk straddles a column bound) are stored.  84 chunks
    // are 1.3 wave stores, so rows go out in PAIRS (A = the even row, B = A + 1): 168 chunks = three stores instead of four x3 ones
    //   S1: lane j        -> row A chunk 6 + j
    //   S2: lanes 0 .. 19 -> row A chunk 70 + j,   lanes 20 .. 63 -> row B chunk j - 14
    //   S3: lanes 0 .. 39 -> row B chunk 50 + j    (lanes 40 .. 63 out of range)
    // Row A waits one step in LDS: written PIXEL-MAJOR (float 3 P + c: three ds_write2_b32 of the register pairs as they lie),
    // read back as one ds_read_b128 per lane and store.  Offsets are lane constant + row A's offset (row B = + one row, folded
    // into the constant); rows outside the tile and chunks outside the columns are dropped by the range check as before.
    typedef int i4v __attribute__((ext_vector_type(4)));
    int st4_off[3] = {0, 0, 0}, st4_lds[3] = {0, 0, 0};
    if constexpr (ST4) {
        const int colb = (xw0 - kRgb2Halo) * 12;                       // byte offset of chunk 0 inside a row (a multiple of 16)
        const int wend = (xw0 + kRgb2Cols < W ? xw0 + kRgb2Cols : W);  // one past the wave's last output column
        auto chunk = [&](int q, int rowb, int ldsb, int& off, int& lds) {
            const int x_last = xw0 - kRgb2Halo + (4 * q + 3) / 3;      // last pixel the chunk touches
            const bool ok = q >= 6 && q < 90 && x_last < wend;
            off = ok ? rowb + colb + 16 * q : kRgb2Out;
            lds = ldsb + 16 * (q < 96 ? q : 95);
        };
        chunk(6 + lane, 0, 0, st4_off[0], st4_lds[0]);
        if (lane < 20) chunk(70 + lane, 0, 0, st4_off[1], st4_lds[1]);
        else chunk(lane - 14, rb, 1536, st4_off[1], st4_lds[1]);
        chunk(lane < 40 ? 50 + lane : 95, rb, 1536, st4_off[2], st4_lds[2]);
        if (lane >= 40) st4_off[2] = kRgb2Out;
    }
    // write this step's row of map `m` (0 orient, 1 line_end) into its pair slot; slot B completes the pair: three stores
    auto st4_put = [&](const f2 (&val)[3], int m, int slot) {
        float* const dst = tr + (m * 2 + slot) * 384 + lane * 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dst[c] = val[c].x;
            dst[3 + c] = val[c].y;
        }
    };
    auto st4_flush = [&](int m, __amdgpu_buffer_rsrc_t rsrc, int roA) {
        const char* const base = reinterpret_cast<const char*>(tr + m * 2 * 384);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const i4v d = *reinterpret_cast<const i4v*>(base + st4_lds[k]);
            __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, (int)((unsigned)st4_off[k] + (unsigned)roA), 0, kRgb2StoreAux);
        }
    };
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) s_hist[wave][k][c][lane] = zero2;

    // nb[r & 1]: input row r, fetched TWO steps ahead; the row loop is unrolled by two so that the two buffers alternate without
    // a move (a move would consume the load early)
    i3 nb0[2], nb1[2];
    auto fetch = [&](i3 (&buf)[2], int row) {
        const unsigned ro = (unsigned)((y0 - kRgb2RowHalo + row) * rb);   // (a row outside the level: out of range by itself)
        buf[0] = __builtin_amdgcn_raw_buffer_load_b96(r_src, (int)((unsigned)in0 + ro), 0, 0);
        buf[1] = __builtin_amdgcn_raw_buffer_load_b96(r_src, (int)((unsigned)in1 + ro), 0, 0);
    };
    fetch(nb0, 0);
    fetch(nb1, 1);
    // The wait at the head of the row loop merges two paths: the back edge (14 younger operations behind the loads it waits for:
    // 6 + 6 stores and 2 loads) and this prologue (2: the second fetch) -- the compiler takes the minimum, vmcnt(2), which on
    // the back edge waits for the previous iteration's 12 stores.  Twelve out-of-range (dropped) stores make both paths 14
    // (MM: one summary store more per step, 14 make both 16).
#pragma unroll
    for (int k = 0; k < (MM ? 14 : 12); ++k) __builtin_amdgcn_raw_buffer_store_b32(k, r_value, kRgb2Out + 64 * k, 0, 0);   // (distinct: identical ones are merged)

    // Masks.  Every stage's row is forced to 0 outside the level (the next stage's SAME padding) and the line-end map inside the
    // pad border.  For most steps of most waves nothing of that can bite: `inside` (one unsigned compare per step) says that the
    // seven rows this step touches lie in the level, the output row outside the pad border, and -- cols_ok, per wave -- all 128
    // columns inside both; then the selects and the mask product are skipped (wave-uniform branches).  Otherwise the row tests
    // are (unsigned)y < H.
    const bool cols_ok = __builtin_amdgcn_readfirstlane(__all(col0 && col1 && padc0 && padc1) ? 1 : 0) != 0;
    const int in_hi = (H - prm.pad < H - 6 ? H - prm.pad : H - 6) - prm.pad;   // yout in [pad, pad + in_hi): rows yout .. yout + 6 in the level
    const unsigned in_len = (cols_ok && in_hi > 0) ? (unsigned)in_hi : 0u;
    auto mask_stage = [&](f2 (&g)[3], int y, int outside) {
        if (outside != 0) {
            const bool rok = (unsigned)y < (unsigned)H;
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = f2{(rok && col0) ? g[c].x : 0.0f, (rok && col1) ? g[c].y : 0.0f};
        }
    };
    auto relu_stage = [&](f2 (&g)[3], int y, int outside) {
        if constexpr (SYM) {
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = f2{relu_max3(g[c].x), relu_max3(g[c].y)};
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = f2{relu_tf(g[c].x), relu_tf(g[c].y)};
        }
        mask_stage(g, y, outside);
    };
    // one row step: `mine` holds this row; row + 2 is fetched into it as soon as the row has been taken
    auto step = [&](int row, i3 (&mine)[2], auto parity) {
        constexpr int PAR = decltype(parity)::value;   // row & 1 (the loop is unrolled by two)
        const int yin = y0 - kRgb2RowHalo + row;  // input row of this step
        // 0 inside, -1 otherwise: kept as an INTEGER (both differences non-negative <=> inside) and compared where it is used -- a
        // bool that lives across blocks travels as a lane mask through a VGPR (v_cndmask + v_cmp per use)
        const int outside = ((yin - 7 - prm.pad) | ((int)in_len - 1 - (yin - 7 - prm.pad))) >> 31;
        f2 v[3][3], g[3];
        // ---- rgc: completes row yin - 1
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[1][c] = f2{__int_as_float(mine[0][c]), __int_as_float(mine[1][c])};
        }
        asm volatile("" : "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]));   // taken: the buffer is free
        fetch(mine, row + 2);   // (rows past the tile's last input row are fetched and not used)
        if constexpr (SYM) {
            // per channel a 3x3 kernel that is mirror-symmetric in both axes: F = l + r; the arriving row gives the rows above and
            // below it corner * F + edge_h * c and its own row edge_v * F + centre * c: 4 fmas + 2 adds instead of 9 fmas
            f2 F[3], E[3], M[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) F[c] = pk_add_swap_both(side_pair(v[1][c]), v[1][c]);
            rgb2_for<0, 3>([&](auto cc) { constexpr int c = decltype(cc)::value; E[c] = ws.template mul<L::b_rgc + c>(F[c]); });
            rgb2_for<0, 3>([&](auto cc) { constexpr int c = decltype(cc)::value; M[c] = ws.template fma<L::b_rgc + 3 + c>(F[c], b1[c]); });
            rgb2_for<0, 3>([&](auto cc) { constexpr int c = decltype(cc)::value; E[c] = ws.template fma<L::b_rgc + 6 + c>(v[1][c], E[c]); });
            rgb2_for<0, 3>([&](auto cc) { constexpr int c = decltype(cc)::value; M[c] = ws.template fma<L::b_rgc + 9 + c>(v[1][c], M[c]); });
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                g[c] = a1[c] + E[c];
                a1[c] = M[c];
                b1[c] = E[c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) neighbours2(v[1][c], v[0][c], v[2][c]);
            conv3_roll2<RGC_PAIRS & 0x1ffu, L::b_rgc>(v, ws, a1, b1, g);
        }
        {
            // The reference's rgc convolution is DENSE: its 54 zero weights (midget_rgc is channel-diagonal) still multiply, and
            // 0 * x is NaN for a NaN / inf pixel -- a non-finite value in ONE channel of the 3 x 3 window makes ALL THREE outputs
            // non-finite there.  The diagonal forms skip those products, so the outputs are "poisoned" instead: s = the sum of the
            // three outputs is non-finite exactly where some channel's window holds a non-finite value (every tap of a channel's
            // own 3 x 3 profile is non-zero), u = s - s is +0 there ... or NaN, and relu takes u along -- v_maximum3_f32(g, 0, u)
            // = max(g, 0) for u = +0 and NaN for u = NaN (no instruction more than the plain relu; the non-SYM tiers add u first).
            // The rgc map itself is not an output of this kernel; after the next stage (rgby: every product kept) the NaN / inf
            // footprint is the reference's (tests/test_gpu_parity.py::test_rgb_chain_nonfinite_pixels_against_the_oracle).
            if constexpr (kRgcDiag) {
                const f2 sg = (g[0] + g[1]) + g[2];
                const f2 u = sg - sg;
                if constexpr (SYM) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) g[c] = f2{relu_max3_poisoned(g[c].x, u.x), relu_max3_poisoned(g[c].y, u.y)};
                } else {
#pragma unroll
                    for (int c = 0; c < 3; ++c) g[c] = g[c] + u;
#pragma unroll
                    for (int c = 0; c < 3; ++c) g[c] = f2{relu_tf(g[c].x), relu_tf(g[c].y)};
                }
                mask_stage(g, yin - 1, outside);
            } else {
                relu_stage(g, yin - 1, outside);
            }
        }
        // ---- rgby: completes row yin - 2
        if constexpr (SYM) {
            // K = S (x) A around the centre + B at the centre: mix the channels first (Z_o = sum_i A[i][o] x_i), then one
            // mirror-symmetric profile per output (S's centre is 0) and the centre's own mix: 9 + 3 + 6 + 3 + 3 + 9 = 33 instead of 45
            constexpr int B = L::b_rgby;
            f2 Z[3], F[3], E[3], M[3];
            rgb2_for<0, 9>([&](auto tt) {
                constexpr int t = decltype(tt)::value, i = t / 3, o = t % 3;
                if constexpr (i == 0) Z[o] = ws.template mul<B + t>(g[0]);
                else Z[o] = ws.template fma<B + t>(g[i], Z[o]);
            });
#pragma unroll
            for (int o = 0; o < 3; ++o) F[o] = pk_add_swap_both(side_pair(Z[o]), Z[o]);
            rgb2_for<0, 3>([&](auto oo) { constexpr int o = decltype(oo)::value; E[o] = ws.template mul<B + 9 + o>(F[o]); });
            rgb2_for<0, 3>([&](auto oo) { constexpr int o = decltype(oo)::value; M[o] = ws.template fma<B + 12 + o>(F[o], b2[o]); });
            rgb2_for<0, 3>([&](auto oo) { constexpr int o = decltype(oo)::value; E[o] = ws.template fma<B + 15 + o>(Z[o], E[o]); });
            rgb2_for<0, 9>([&](auto tt) {
                constexpr int t = decltype(tt)::value, i = t / 3, o = t % 3;
                M[o] = ws.template fma<B + 18 + t>(g[i], M[o]);
            });
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                g[o] = a2[o] + E[o];
                a2[o] = M[o];
                b2[o] = E[o];
            }
        } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[1][c] = g[c];
            neighbours2(v[1][c], v[0][c], v[2][c]);
        }
        if constexpr (RGBY_A != kDense) conv3_roll2_struct<RGBY_A, RGBY_A, RGBY_A, L::b_rgby>(v, ws, a2, b2, g);
        else conv3_roll2<0x1ffu, L::b_rgby>(v, ws, reinterpret_cast<f2 (&)[3]>(a2), reinterpret_cast<f2 (&)[3]>(b2), g);
        }
        {
            relu_stage(g, yin - 2, outside);
        }
        // ---- stripe: completes row q = yin - 3
        if constexpr (SYM) {
            // stripe of the channel sum, left / right taps in the paired forms (no moves): per output 3 x (side pair, centre, self pair)
            const f2 S = (g[0] + g[1]) + g[2];
            const f2 D = side_pair(S);
            f2 npa[3], npb[3];
            rgb2_for<0, 3>([&](auto oo) {
                constexpr int o = decltype(oo)::value, B = L::b_stripe + o * 16;
                f2 t2 = a3[o], t1 = b3[o], t0;
                t2 = ws.template fma_sw<B + 0>(D, t2);
                t1 = ws.template fma_sw<B + 2>(D, t1);
                t0 = ws.template fma0_sw<B + 4>(D);
                t2 = ws.template fma<B + 6>(S, t2);
                t1 = ws.template fma<B + 7>(S, t1);
                t0 = ws.template fma<B + 8>(S, t0);
                t2 = ws.template fma_sw<B + 10>(S, t2);
                t1 = ws.template fma_sw<B + 12>(S, t1);
                t0 = ws.template fma_sw<B + 14>(S, t0);
                g[o] = t2;
                npa[o] = t1;
                npb[o] = t0;
            });
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                a3[o] = npa[o];
                b3[o] = npb[o];
            }
        } else if constexpr (STRIPE_SUM) {
            f2 s3[3];
            s3[1] = (g[0] + g[1]) + g[2];
            neighbours2(s3[1], s3[0], s3[2]);
            conv3_roll2_sum<L::b_stripe>(s3, ws, a3, b3, g);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                v[1][c] = g[c];
                neighbours2(v[1][c], v[0][c], v[2][c]);
            }
            conv3_roll2<0x1ffu, L::b_stripe>(v, ws, a3, b3, g);
        }
        {
            relu_stage(g, yin - 3, outside);
        }
        f2 xs3[3];   // the stripe row of three steps ago
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s_hist[wave][row & 3][c][lane] = g[c];
            xs3[c] = s_hist[wave][(row + 1) & 3][c][lane];
        }
        // ---- blur of the channel sum: stripe row q feeds blur rows q-3 .. q+3; row t = q - 3 completes
        f2 bdone;
        {
            const f2 s = (g[0] + g[1]) + g[2];
            // T[k] = (sum at x0 + k - 3, sum at x1 + k - 3)
            f2 T[7];
            f2 nb[7];
            if constexpr (SYM) {
                // the folded columns straight from three DPP pairs (a = below(s.x), b = below(s.y), c = above(s.x), d = above(s.y),
                // e = below(b), f = above(c)): F0 = T0 + T6 = (e + d, a + f), F1 = T1 + T5 = (a + c, b + d), F2 = T2 + T4 =
                // (b + s.y, s.x + c) -- the same sums as below without building the seven shifted pairs (8 moves)
                const f2 X = side_pair(s);                                          // (c, b)
                const f2 Y = f2{from_lane_below(s.x), from_lane_above(s.y)};        // (a, d)
                const f2 Q = f2{from_lane_below(X.y), from_lane_above(X.x)};        // (e, f)
                T[0] = pk_add_swap_second(Q, Y);
                T[1] = Y + X;
                T[2] = pk_add_swap_both(X, s);
                T[3] = s;
            } else {
            const float a = from_lane_below(s.x), b = from_lane_below(s.y), c = from_lane_above(s.x), d = from_lane_above(s.y);
            const float e = from_lane_below(b), f = from_lane_above(c);
            T[0] = f2{e, a};
            T[1] = f2{a, b};
            T[2] = f2{b, s.x};
            T[3] = s;
            T[4] = f2{s.y, c};
            T[5] = f2{c, d};
            T[6] = f2{d, f};
            }
            if constexpr (L::blur_sym) {
                // Mirror-symmetric blur (w[dy][dx] = w[6 - dy][dx] = w[dy][6 - dx]): fold the columns (3 adds), one partial sum
                // per |dy| (P[d] = sum_j w[d][j] F[j], 16 packed fmas), and every pending row takes the partial sum of ITS |dy|
                // (7 adds) -- 26 instead of 49 instructions.  All terms are >= 0 (relu outputs, positive weights), so "the blur
                // is exactly 0" still means "every value of the window is 0" in this order too.
                f2 F[4];
                if constexpr (SYM) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) F[j] = T[j];
                } else {
                    F[0] = T[0] + T[6];
                    F[1] = T[1] + T[5];
                    F[2] = T[2] + T[4];
                    F[3] = T[3];
                }
                f2 P[4];
                rgb2_for<0, 4>([&](auto jj) {
                    constexpr int j = decltype(jj)::value;
                    rgb2_for<0, 4>([&](auto dd) {      // the four chains side by side (stream order j * 4 + d)
                        constexpr int d = decltype(dd)::value, p = L::b_blur + j * 4 + d;
                        if constexpr (j == 0) P[d] = ws.template mul<p>(F[0]);
                        else P[d] = ws.template fma<p>(F[j], P[d]);
                    });
                });
                nb[0] = pb[0] + P[3];
                nb[1] = pb[1] + P[2];
                nb[2] = pb[2] + P[1];
                nb[3] = pb[3] + P[0];
                nb[4] = pb[4] + P[1];
                nb[5] = pb[5] + P[2];
                nb[6] = P[3];
            } else {
            // the seven pending blur rows advance side by side: for dx: for k (stream order dx * 7 + k)
            rgb2_for<0, 7>([&](auto xx) {
                constexpr int dx = decltype(xx)::value;
                rgb2_for<0, 7>([&](auto kk) {
                    constexpr int k = decltype(kk)::value, p = L::b_blur + dx * 7 + k;
                    if constexpr (dx == 0) {
                        if constexpr (k == 6) nb[k] = ws.template fma0<p>(T[0]);
                        else nb[k] = ws.template fma<p>(T[0], pb[k]);
                    } else {
                        nb[k] = ws.template fma<p>(T[dx], nb[k]);
                    }
                });
            });
            }
            bdone = nb[0];
#pragma unroll
            for (int k = 0; k < 6; ++k) pb[k] = nb[k + 1];
        }
        // ---- regulate row t = yin - 6 with the stripe row kept 3 steps back
        const int t = yin - 6;
        f2 o3[3];
        {
            f2 rr;
            if constexpr (SYM) {
                // regulator_ratio (silent_rgb.h) for both halves side by side: min(b, 1) as v_minimum3_f32 (a NaN stays), the two
                // log2 / exp2 chains interleaved, ONE rarely taken branch for the denormal / root = 0 cases of either half
                float m0, m1;
                asm("v_minimum3_f32 %0, %1, 1.0, 1.0" : "=v"(m0) : "v"(bdone.x));
                asm("v_minimum3_f32 %0, %1, 1.0, 1.0" : "=v"(m1) : "v"(bdone.y));
                rr = f2{prm.rv * __builtin_amdgcn_exp2f(-prm.root * __builtin_amdgcn_logf(m0)),
                        prm.rv * __builtin_amdgcn_exp2f(-prm.root * __builtin_amdgcn_logf(m1))};
                const bool s0 = (m0 > 0.0f && m0 < 7.8886e-31f) || prm.root == 0.0f, s1 = (m1 > 0.0f && m1 < 7.8886e-31f) || prm.root == 0.0f;
                if (s0 || s1) {
                    if (s0) rr.x = prm.rv / powf(m0, prm.root);
                    if (s1) rr.y = prm.rv / powf(m1, prm.root);
                }
            } else {
                rr = f2{regulator_ratio(bdone.x, prm.rv, prm.root), regulator_ratio(bdone.y, prm.rv, prm.root)};
            }
            int flat_policy = prm.flat_policy;
            asm volatile("" : "+s"(flat_policy));   // compared here, every step: hoisted out of the loop the bool travels through a VGPR
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f2 xs = xs3[c];
                f2 y = xs * rr;
                if (flat_policy == SILENT_FLAT_ZERO) {
                    if (xs.x == 0.0f) y.x = 0.0f;
                    if (xs.y == 0.0f) y.y = 0.0f;
                }
                o3[c] = y;
            }
            if (outside != 0) {
                const bool rok = (unsigned)t < (unsigned)H;
#pragma unroll
                for (int c = 0; c < 3; ++c) o3[c] = f2{(rok && col0) ? o3[c].x : 0.0f, (rok && col1) ? o3[c].y : 0.0f};
            }
        }
        if constexpr (ST4) {
            // orient row t - y0 = row - 13: even on the odd steps (slot A), the pair completes on the even ones
            st4_put(o3, 0, PAR == 1 ? 0 : 1);
            if constexpr (PAR == 0) st4_flush(0, r_orient, (t - y0 - 1) * rb);
        } else {
            store_row3(o3, r_orient, (t - y0) * rb);
        }
        // ---- end bank: completes row yout = yin - 7
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v[1][c] = o3[c];
            neighbours2(v[1][c], v[0][c], v[2][c]);
        }
        if constexpr (END_A0 != kDense) conv3_roll2_struct<END_A0, END_A1, END_A2, L::b_end>(v, ws, a5, b5, g);
        else conv3_roll2<0x1ffu, L::b_end>(v, ws, reinterpret_cast<f2 (&)[3]>(a5), reinterpret_cast<f2 (&)[3]>(b5), g);
        const int yout = yin - 7, k = yout - y0;   // k: output row inside the tile
        {
            f2 le[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                le[c] = SYM ? f2{relu_clip_max3(g[c].x, prm.clip_hi), relu_clip_max3(g[c].y, prm.clip_hi)}
                            : f2{clip_hi_tf(relu_tf(g[c].x), prm.clip_hi), clip_hi_tf(relu_tf(g[c].y), prm.clip_hi)};
            if (outside != 0) {   // pad_inwards: the product with a 0 / 1 mask (a NaN in the border stays a NaN, like the reference's)
                const bool padr = yout >= prm.pad && yout < H - prm.pad;
                const f2 mk = {(padc0 && padr) ? 1.0f : 0.0f, (padc1 && padr) ? 1.0f : 0.0f};
#pragma unroll
                for (int c = 0; c < 3; ++c) le[c] = mk * le[c];
            }
            if constexpr (ST4) {
                // line_end row k = row - 14: even on the even steps (slot A), the pair completes on the odd ones
                st4_put(le, 1, PAR == 0 ? 0 : 1);
                if constexpr (PAR == 1) st4_flush(1, r_line, (k - 1) * rb);
            } else {
                store_row3(le, r_line, k * rb);
            }
            const f2 val = ((le[0] + le[1]) + le[2]) * inv3p;
            typedef int i2 __attribute__((ext_vector_type(2)));
            const int norow = (k | (rows_t - 1 - k)) >> 31;   // 0: an output row of this tile (wave-uniform, an integer like `outside`)
            if (MM && norow == 0) {   // v_max3_f32 drops NaN operands like pool_max does (max_pool semantics, silent_peaks.h)
                const float a = out0 ? val.x : kPoolLowest, b = out1 ? val.y : kPoolLowest;
                const float na = out0 ? -val.x : kPoolLowest, nb = out1 ? -val.y : kPoolLowest;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mm_mx) : "v"(mm_mx), "v"(a), "v"(b));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mm_nmn) : "v"(mm_nmn), "v"(na), "v"(nb));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mm_grp) : "v"(mm_grp), "v"(a), "v"(b));
                mm_nan |= __ballot((out0 && val.x != val.x) || (out1 && val.y != val.y));
            }
            if constexpr (MM) {
                const bool gend = norow == 0 && (((k + 1) & (kSumRows - 1)) == 0 || k == rows_t - 1);   // wave-uniform
                const unsigned so = gend ? (unsigned)((k >> kSumRowsLog2) * nxp * 4) : (unsigned)kRgb2Out;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(mm_grp), r_sum, (int)((unsigned)ssum + so), 0, 0);
                mm_grp = gend ? kPoolLowest : mm_grp;
            }
            const unsigned rv = (unsigned)(k * W * 4);
            __builtin_amdgcn_raw_buffer_store_b64(i2{__float_as_int(val.x), __float_as_int(val.y)}, r_value, (int)((unsigned)sv2 + rv), 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(val.x), r_value, (int)((unsigned)sv1 + rv), 0, 0);
        }
        ws.template skip<L::total, L::blocks * kRgb2Blk>();
    };
#pragma unroll 1
    for (int row = 0; row < NROWS; row += 2) {   // th is even: whole pairs of rows
        step(row, nb0, ic<0>{});
        step(row + 1, nb1, ic<1>{});
    }
    if constexpr (MM) {
        const float mx = wave_max(mm_mx), nmn = wave_max(mm_nmn);
        if (lane == 0) {
            unsigned* slot = args.mm + ((long long)tc.frame * tab.n_levels + tc.level) * 2;
            atomicMax(slot, f2ord(mx));
            atomicMax(slot + 1, f2ord(nmn));
            if (mm_nan && args.nan_flags) args.nan_flags[tc.frame * tab.n_levels + tc.level] = 1;
        }
    }
}

}  // namespace silent
