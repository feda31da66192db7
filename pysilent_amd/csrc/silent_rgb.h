// Fused RGB chain (BASELINE config 3; reference graph recognition_testing.py:69-77):
//   rgc -> ReLU -> rgby -> ReLU -> stripe -> ReLU -> regulate(blur 7x7) -> end -> ReLU -> clip -> pad_inwards -> value
// in ONE launch: the pyramid is read once and only the returned maps (orient, line_end, value) are written.
//
// Wave-autonomous streaming, no barrier in the row loop (same idea as gray_line_end_kernel).  A wave owns 64
// columns and walks down R + 14 input rows; lane = column.  Every convolution keeps ROLLING ROW ACCUMULATORS
// instead of a window: an arriving row adds its dy = 2 terms to the oldest pending output row (which
// completes), its dy = 1 terms to the next, and opens a new one with its dy = 0 terms -- the fma chain of an
// output is still (dy, dx, i) ordered, bit-identical to conv2d_same_kernel.  Left / right neighbours come from
// DPP wave shifts.  Each 3x3 stage costs one lane per side, the 7x7 blur three: lanes 7..56 (50 columns)
// produce outputs.  The blur must be channel-uniform (blur_tensor writes one profile into every (in, out)
// pair, gaussian_blur.py:51-52), so it acts on the channel SUM: 49 taps instead of 441.  The host checks
// this and falls back to the stage-per-launch path otherwise.
//
// Weights: 373 floats do not fit the ~100 SGPRs of a wave.  Left to itself hipcc hoists every (loop
// invariant) kernarg load out of the row loop and then spills SGPRs into VGPR lanes (v_writelane /
// v_readlane + hazard nops), 9x below the VALU estimate.  What works: every fma chain re-reads its 7..9
// weights from the kernarg segment (scalar cache) through a pointer laundered by an empty asm -- not
// hoistable, not mergeable -- AND the chain's result is laundered too: SelectionDAG emits side-effecting
// nodes in program order but lets pure fmas float, so without the second fence all weight loads of a row
// step are emitted first and spilled.  ~20 weight SGPRs live, 90-140 VGPRs, no LDS.
// (An LDS-resident weight table read by broadcast ds_read_b128 measured the same time: LDS issue and VALU
// were co-limiting there.)
#pragma once

#include <cstddef>

#include "silent_common.h"

namespace silent {

struct RgbW {
    float rgc[81], rgby[81], stripe[81], end[81];  // [o][dy][dx][i] (repacked from HWIO by the host)
    float blur[49];                                // profile [dy][dx]
};
// Two-group ("centre / surround") form of a 3x3x3->3 kernel, stored in the first 45 floats of its dense slot:
//   K[t][i][o] = scale[t][i] * (tap t of input i is in group A ? mixA[i][o] : mixB[i][o]),   scale >= 0
// scale [dy][dx][i] at 0..26, mixA [i][o] at 27..35, mixB [i][o] at 36..44.  Which taps are in group A is a
// compile-time mask per input channel (bit dy * 3 + dx).  This is how the reference's generators build
// rgby_3 (one 3x3 profile, centre tap -> one channel-mix matrix, the 8 others -> another) and rgb_2d_end_tensors
// (one-hot input per orientation, taps with z >= 0 -> center_out, z < 0 -> surround_out): 27 + 18 instead of 81 fmas.
constexpr int kStructMixA = 27, kStructMixB = 36;

struct RgbP {
    float rv, root;
    int flat_policy;
    float clip_hi;
    int pad;
};

constexpr int kRgbHalo = 7;
constexpr int kRgbCols = 64 - 2 * kRgbHalo;  // 50 output columns per wave
constexpr int kRgbTW = 4 * kRgbCols;         // 4 waves side by side
constexpr int kRgbTH = 90;                   // round 1's fixed tile height (50 / 72 / 90 / 108 / 120 measured: 1.44 / 1.36 / 1.33 / 1.34 / 1.40 ms)
// Round 2: the tile height is a launch parameter (RgbArgs::th, even, kRgbTHMin .. kRgbTHMax).  Launches of about one round of
// resident tiles or less get the height that minimises ceil(tiles / resident tiles) x (th + 14) row steps -- short tiles, the
// generalisation of round 1's 18-row instantiation; launches that fill the chip several times keep 90 rows (the sweep in
// silent_rgb_api.hip: +-4 % without a trend).
constexpr int kRgbTHMin = 18, kRgbTHMax = 160;
constexpr int kRgbChunk = 2;                 // input rows per prefetch chunk; (TH + 14) % chunk == 0
// A wave walks its TH + 14 rows one after the other (~1.8 us per row when it has a SIMD to itself), so a launch with
// few tiles -- one 480p frame of the reference application has 36 waves' worth -- takes 104 row steps = 190 us whatever
// its size.  Such launches use 18-row tiles: 5x as many waves, 32 row steps each (1.78x the rows of work instead of 1.16x,
// which only matters once the chip is full).
static_assert((kRgbTH + 2 * kRgbHalo) % kRgbChunk == 0 && (kRgbTHMin + 2 * kRgbHalo) % kRgbChunk == 0,
              "row pipeline works in whole chunks");


// Weights stay in the kernarg segment and are re-read per fma chain with scalar loads (scalar cache) through
// a laundered pointer; the next chain is requested before the current chain's fmas.
typedef const __attribute__((address_space(4))) float* kfloat_p;
struct Chain3 {
    float k[9];
};
__device__ __forceinline__ Chain3 load_chain3(kfloat_p wp, int off) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" : "+s"(wp));
    Chain3 c;
#pragma unroll
    for (int j = 0; j < 9; ++j) c.k[j] = wp[off + j];
    return c;
}

// One arriving row of a 3x3 x 3->3 convolution.  v[dx][i]: the row's values at x-1, x, x+1.
// pa: output row with dy = 0,1 already in; pb: output row with dy = 0 in.  Returns the completed row in done.
// PAIRS: bit (o * 3 + i) set = output o reads input i.  The host clears a bit only when all 9 taps of that pair
// are exactly 0 (midget_rgc is channel-diagonal: 27 of its 81 weights are non-zero), so for finite inputs the
// skipped fmas would have added exactly 0.
template <unsigned PAIRS>
__device__ __forceinline__ void conv3_roll(const float (&v)[3][3], kfloat_p wstage, float (&pa)[3], float (&pb)[3],
                                           float (&done)[3]) {
    float acc[3][3];  // [o][dy]
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        acc[o][0] = 0.0f;   // chain dy continues the row that already holds dy' < dy
        acc[o][1] = pb[o];
        acc[o][2] = pa[o];
    }
    // chain order: c = o * 3 + (2 - dy); host layout [o][dy][dx][i] -> chain (o, dy) at (o * 3 + dy) * 9
    Chain3 w0 = load_chain3(wstage, (0 * 3 + 2) * 9);
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        const int o = c / 3, dy = 2 - c % 3;
        Chain3 w1 = w0;
        if (c + 1 < 9) {
            const int o1 = (c + 1) / 3, dy1 = 2 - (c + 1) % 3;
            w1 = load_chain3(wstage, (o1 * 3 + dy1) * 9);
        }
        float t = acc[o][dy];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if ((PAIRS >> (o * 3 + i)) & 1u) t = __builtin_fmaf(v[dx][i], w0.k[dx * 3 + i], t);
        asm volatile("" : "+v"(t));  // pins the chain's fmas between the loads around it
        acc[o][dy] = t;
        w0 = w1;
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        done[o] = acc[o][2];
        pa[o] = acc[o][1];
        pb[o] = acc[o][0];
    }
}

// One arriving row of a two-group kernel.  pa / pb hold, per pending output row, the partial sums of group A and of
// group B for every input channel: [0..2] = A_i, [3..5] = B_i.  Rounding differs from the 81-term chain (re-association).
template <unsigned M0, unsigned M1, unsigned M2>
__device__ __forceinline__ void conv3_roll_struct(const float (&v)[3][3], kfloat_p wstage, float (&pa)[6], float (&pb)[6],
                                                  float (&done)[3]) {
    float acc[6][3];  // [group * 3 + i][dy]
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        acc[q][0] = 0.0f;
        acc[q][1] = pb[q];
        acc[q][2] = pa[q];
    }
    constexpr unsigned M[3] = {M0, M1, M2};
#pragma unroll
    for (int dy = 2; dy >= 0; --dy) {
        const Chain3 w = load_chain3(wstage, dy * 9);  // scale[dy][dx][i]
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int q = ((M[i] >> (dy * 3 + dx)) & 1u) ? i : 3 + i;
                acc[q][dy] = __builtin_fmaf(v[dx][i], w.k[dx * 3 + i], acc[q][dy]);
            }
#pragma unroll
        for (int q = 0; q < 6; ++q) asm volatile("" : "+v"(acc[q][dy]));
    }
    // the completed row: out[o] = sum_i mixA[i][o] * A_i + mixB[i][o] * B_i
    const Chain3 ma = load_chain3(wstage, kStructMixA), mb = load_chain3(wstage, kStructMixB);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        float t = ma.k[0 * 3 + o] * acc[0][2];
        t = __builtin_fmaf(ma.k[1 * 3 + o], acc[1][2], t);
        t = __builtin_fmaf(ma.k[2 * 3 + o], acc[2][2], t);
        t = __builtin_fmaf(mb.k[0 * 3 + o], acc[3][2], t);
        t = __builtin_fmaf(mb.k[1 * 3 + o], acc[4][2], t);
        t = __builtin_fmaf(mb.k[2 * 3 + o], acc[5][2], t);
        done[o] = t;
    }
    asm volatile("" : "+v"(done[0]), "+v"(done[1]), "+v"(done[2]));
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        pa[q] = acc[q][1];
        pb[q] = acc[q][0];
    }
}

// The same for a kernel that does not depend on the input channel (rgb_2d_stripe_tensors with its default
// in_channel = (1, 1, 1): every orientation reads the channel SUM): 27 fmas on the sum instead of 81.
// s[dx]: channel sum at x-1, x, x+1.  Rounding differs from the 81-term chain like the blur's does.
__device__ __forceinline__ void conv3_roll_sum(const float (&s)[3], kfloat_p wstage, float (&pa)[3], float (&pb)[3],
                                               float (&done)[3]) {
    float acc[3][3];  // [o][dy]
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        acc[o][0] = 0.0f;
        acc[o][1] = pb[o];
        acc[o][2] = pa[o];
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        Chain3 w[3];  // the three rows of output o: 3 x 9 weights of which every third (i = 0) is used
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) w[dy] = load_chain3(wstage, (o * 3 + dy) * 9);
#pragma unroll
        for (int dy = 2; dy >= 0; --dy) {
            float t = acc[o][dy];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) t = __builtin_fmaf(s[dx], w[dy].k[dx * 3], t);
            acc[o][dy] = t;
        }
        asm volatile("" : "+v"(acc[o][0]), "+v"(acc[o][1]), "+v"(acc[o][2]));
    }
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        done[o] = acc[o][2];
        pa[o] = acc[o][1];
        pb[o] = acc[o][0];
    }
}

// The regulator's factor rv / min(b, 1) ** root, evaluated as rv * exp2(-root * log2(m)) on the transcendental unit
// (v_log_f32, v_exp_f32: 1 ulp each; <= 3 ulp in all since |root * log2 m| stays small) -- no powf (~100 instructions) and no
// IEEE division (~10).  m = 0 gives exp2(+inf) = inf like rv / 0.  v_log_f32 treats denormal inputs as 0: below 2^-100 the
// accurate powf and a true division run (a lane-divergent branch that no lane takes in practice), and for root = 0
// (0 * -inf; pow(0, 0) = 1).  silent_regulate / conv2d_same_kernel keep powf.
__device__ __forceinline__ float regulator_ratio(float bd, float rv, float root) {
    const float m = bd > 1.0f ? 1.0f : bd;
    if ((m > 0.0f && m < 7.8886e-31f) || root == 0.0f) return rv / powf(m, root);
    return rv * __builtin_amdgcn_exp2f(-root * __builtin_amdgcn_logf(m));
}

__device__ __forceinline__ void with_neighbours(const float (&c)[3], float (&v)[3][3]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[0][i] = from_lane_below(c[i]);
        v[1][i] = c[i];
        v[2][i] = from_lane_above(c[i]);
    }
}

// The kernel takes ONE by-value struct so that the offset of the weights inside the kernarg segment is
// offsetof(RgbArgs, w) by construction.
struct RgbArgs {
    const float* pyr;
    float* orient_out;
    float* line_out;
    float* value_out;
    LevelTab tab;
    RgbW w;
    RgbP prm;
    int th;   // output rows per tile (even)
};

// RGC_PAIRS: (o, i) pairs of the rgc kernel that are not identically zero; STRIPE_SUM: the stripe kernel does not
// depend on the input channel.  The host checks both on the actual weights and launches <0x1ff, false> otherwise.
// RGBY_A: group-A tap mask of the two-group form of rgby (same for the three inputs), END_A0..2: of the end bank per
// input channel; kDense = the dense 81-fma form.
constexpr unsigned kDense = 0xffffffffu;
template <unsigned RGC_PAIRS, bool STRIPE_SUM, unsigned RGBY_A, unsigned END_A0, unsigned END_A1, unsigned END_A2>
__global__ __launch_bounds__(256) void rgb_line_end_kernel(const RgbArgs args) {
    constexpr int D = kRgbChunk;
    const int R = args.th, NCH = (R + 2 * kRgbHalo) / D;       // th is even: whole chunks
    const float* __restrict__ pyr = args.pyr;
    float* __restrict__ orient_out = args.orient_out;
    float* __restrict__ line_out = args.line_out;
    float* __restrict__ value_out = args.value_out;
    const LevelTab& tab = args.tab;
    const RgbP& prm = args.prm;

    typedef const __attribute__((address_space(4))) char* kchar_p;
    const kfloat_p wp = (kfloat_p)((kchar_p)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(RgbArgs, w));

    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const float* __restrict__ src = pyr + base_px * 3;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = tc.tx * kRgbTW + wave * kRgbCols;
    if (xw0 >= W) return;  // wave-uniform
    const int y0 = tc.ty * R;
    const int x = xw0 + lane - kRgbHalo;
    const bool col_ok = x >= 0 && x < W;
    const long long xoff = (long long)min(max(x, 0), W - 1) * 3;
    const bool out_lane = lane >= kRgbHalo && lane < kRgbHalo + kRgbCols && x < W;
    const bool pad_col = x >= prm.pad && x < W - prm.pad;
    const float inv3 = 1.0f / 3.0f;

    // rolling state
    float a1[3] = {0, 0, 0}, b1[3] = {0, 0, 0};  // rgc
    float a2[6] = {0, 0, 0, 0, 0, 0}, b2[6] = {0, 0, 0, 0, 0, 0};  // rgby (dense form: [0..2] only)
    float a3[3] = {0, 0, 0}, b3[3] = {0, 0, 0};  // stripe
    float a5[6] = {0, 0, 0, 0, 0, 0}, b5[6] = {0, 0, 0, 0, 0, 0};  // end
    float pb[7] = {0, 0, 0, 0, 0, 0, 0};         // blur: pb[k] = pending output row (newest stripe row) - 3 + k
    float hist[4][3];                            // stripe rows q, q-1, q-2, q-3 (own column)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) hist[k][c] = 0.0f;

    float cur[D][3], nxt[D][3];
    auto fetch = [&](float (&buf)[D][3], int chunk) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int y = y0 - kRgbHalo + chunk * D + d;
            const bool ok = y >= 0 && y < H && col_ok;
            const float* __restrict__ p = src + (long long)min(max(y, 0), H - 1) * W * 3 + xoff;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t = p[c];
                buf[d][c] = ok ? t : 0.0f;
            }
        }
    };
    fetch(cur, 0);

#pragma unroll 1
    for (int chunk = 0; chunk < NCH; ++chunk) {
        if (chunk + 1 < NCH) fetch(nxt, chunk + 1);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int yin = y0 - kRgbHalo + chunk * D + d;  // input row of this step
            float v[3][3], g[3];
            // ---- rgc: completes row yin - 1
            with_neighbours(cur[d], v);
            conv3_roll<RGC_PAIRS>(v, wp + 0 * 81, a1, b1, g);
            if constexpr ((RGC_PAIRS & 0x1ffu) != 0x1ffu) {
                // channel-diagonal rgc: the skipped 0 * x products of the reference's dense convolution are NaN for a NaN / inf
                // pixel -- poison all three outputs where any channel's window holds one (silent_rgb2.h, rgc stage)
                const float sg = (g[0] + g[1]) + g[2];
                const float u = sg - sg;
#pragma unroll
                for (int c = 0; c < 3; ++c) g[c] = g[c] + u;
            }
            {
                const bool ok = yin - 1 >= 0 && yin - 1 < H && col_ok;
#pragma unroll
                for (int c = 0; c < 3; ++c) g[c] = ok ? relu_tf(g[c]) : 0.0f;
            }
            // ---- rgby: completes row yin - 2
            with_neighbours(g, v);
            if constexpr (RGBY_A != kDense) conv3_roll_struct<RGBY_A, RGBY_A, RGBY_A>(v, wp + 1 * 81, a2, b2, g);
            else conv3_roll<0x1ffu>(v, wp + 1 * 81, reinterpret_cast<float (&)[3]>(a2), reinterpret_cast<float (&)[3]>(b2), g);
            {
                const bool ok = yin - 2 >= 0 && yin - 2 < H && col_ok;
#pragma unroll
                for (int c = 0; c < 3; ++c) g[c] = ok ? relu_tf(g[c]) : 0.0f;
            }
            // ---- stripe: completes row q = yin - 3
            if constexpr (STRIPE_SUM) {
                float s3[3];
                s3[1] = (g[0] + g[1]) + g[2];
                s3[0] = from_lane_below(s3[1]);
                s3[2] = from_lane_above(s3[1]);
                conv3_roll_sum(s3, wp + 2 * 81, a3, b3, g);
            } else {
                with_neighbours(g, v);
                conv3_roll<0x1ffu>(v, wp + 2 * 81, a3, b3, g);
            }
            {
                const bool ok = yin - 3 >= 0 && yin - 3 < H && col_ok;
#pragma unroll
                for (int c = 0; c < 3; ++c) g[c] = ok ? relu_tf(g[c]) : 0.0f;
            }
#pragma unroll
            for (int k = 3; k > 0; --k)
#pragma unroll
                for (int c = 0; c < 3; ++c) hist[k][c] = hist[k - 1][c];
#pragma unroll
            for (int c = 0; c < 3; ++c) hist[0][c] = g[c];
            // ---- blur of the channel sum: stripe row q feeds blur rows q-3 .. q+3; row t = q - 3 completes
            float s7[7];
            s7[3] = (g[0] + g[1]) + g[2];
            s7[2] = from_lane_below(s7[3]);
            s7[1] = from_lane_below(s7[2]);
            s7[0] = from_lane_below(s7[1]);
            s7[4] = from_lane_above(s7[3]);
            s7[5] = from_lane_above(s7[4]);
            s7[6] = from_lane_above(s7[5]);
            float bdone;
            if constexpr (RGBY_A != kDense && END_A0 != kDense) {
                // mirror-symmetric blur, folded exactly like rgb_line_end2_kernel's (silent_rgb2.h): same operations, same order
                kfloat_p kb = wp + 4 * 81;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" : "+s"(kb));
                const float F[4] = {s7[0] + s7[6], s7[1] + s7[5], s7[2] + s7[4], s7[3]};
                float P[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) P[d] = F[0] * kb[(3 + d) * 7 + 0];
#pragma unroll
                for (int j = 1; j < 4; ++j)
#pragma unroll
                    for (int d = 0; d < 4; ++d) P[d] = __builtin_fmaf(F[j], kb[(3 + d) * 7 + j], P[d]);
                asm volatile("" : "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]));
                bdone = pb[0] + P[3];
                pb[0] = pb[1] + P[2];
                pb[1] = pb[2] + P[1];
                pb[2] = pb[3] + P[0];
                pb[3] = pb[4] + P[1];
                pb[4] = pb[5] + P[2];
                pb[5] = P[3];
                pb[6] = 0.0f;
            } else {
                // output row q-3+k takes kernel row dy = 6 - k from this stripe row
                float nb[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    kfloat_p kb = wp + 4 * 81 + (6 - k) * 7;
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("" : "+s"(kb));
                    float kw[7];
#pragma unroll
                    for (int dx = 0; dx < 7; ++dx) kw[dx] = kb[dx];
                    float acc = pb[k];
#pragma unroll
                    for (int dx = 0; dx < 7; ++dx) acc = __builtin_fmaf(s7[dx], kw[dx], acc);
                    asm volatile("" : "+v"(acc));
                    nb[k] = acc;
                }
                bdone = nb[0];
#pragma unroll
                for (int k = 0; k < 6; ++k) pb[k] = nb[k + 1];
                pb[6] = 0.0f;
            }
            // ---- regulate row t = yin - 6 with the stripe row kept 3 steps back
            const int t = yin - 6;
            float o3[3];
            {
                const float r = regulator_ratio(bdone, prm.rv, prm.root);
                const bool ok = t >= 0 && t < H && col_ok;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float xs = hist[3][c];
                    float y = xs * r;
                    if (prm.flat_policy == SILENT_FLAT_ZERO && xs == 0.0f) y = 0.0f;
                    o3[c] = ok ? y : 0.0f;
                }
            }
            if (orient_out && t >= y0 && t < y0 + R && t < H && out_lane) {
                float* __restrict__ po = orient_out + (base_px + (long long)t * W + x) * 3;
                po[0] = o3[0];
                po[1] = o3[1];
                po[2] = o3[2];
            }
            // ---- end bank: completes row yout = yin - 7
            with_neighbours(o3, v);
            if constexpr (END_A0 != kDense) conv3_roll_struct<END_A0, END_A1, END_A2>(v, wp + 3 * 81, a5, b5, g);
            else conv3_roll<0x1ffu>(v, wp + 3 * 81, reinterpret_cast<float (&)[3]>(a5), reinterpret_cast<float (&)[3]>(b5), g);
            const int yout = yin - 7;
            if (yout >= y0 && yout < H && out_lane) {
                const float mk = (pad_col && yout >= prm.pad && yout < H - prm.pad) ? 1.0f : 0.0f;
                float le[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) le[c] = mk * clip_hi_tf(relu_tf(g[c]), prm.clip_hi);
                const long long px = base_px + (long long)yout * W + x;
                if (line_out) {
                    line_out[px * 3 + 0] = le[0];
                    line_out[px * 3 + 1] = le[1];
                    line_out[px * 3 + 2] = le[2];
                }
                if (value_out) value_out[px] = __fmul_rn(__fadd_rn(__fadd_rn(le[0], le[1]), le[2]), inv3);
            }
        }
        if (chunk + 1 < NCH) {
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int c = 0; c < 3; ++c) cur[d][c] = nxt[d][c];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// regulate_sum_kernel: silent_regulate for a 7x7 channel-uniform blur on 3-channel maps (what blur_tensor generates and
// orientation_filter uses, filters/orientation.py:20,31-33) -- the regulator stage of rgb_line_end_kernel on its own:
// the blur is a 49-tap filter of the channel SUM (rolling row accumulators, DPP neighbours, halo 3 lanes per side),
// then y = x * (rv / pow(min(b, 1), root)) with powf.  The dense 441-fma form (conv2d_same_kernel<7,7,3,3,true>) ran at
// 0.6 TB/s on 1080p maps.
constexpr int kRegCols = 58, kRegTW = 4 * kRegCols, kRegTH = 58;  // 58 + 6 = 64 streamed rows per tile

struct RegArgs {
    const float* in;
    float* out;
    LevelTab tab;
    float blur[49];  // profile [dy][dx]
    float rv, root;
    int flat_policy;
};

__global__ __launch_bounds__(256) void regulate_sum_kernel(const RegArgs args) {
    constexpr int R = kRegTH;
    const LevelTab& tab = args.tab;
    typedef const __attribute__((address_space(4))) char* kchar_p;
    const kfloat_p kb0 = (kfloat_p)((kchar_p)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(RegArgs, blur));
    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const float* __restrict__ src = args.in + base_px * 3;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = tc.tx * kRegTW + wave * kRegCols;
    if (xw0 >= W) return;  // wave-uniform
    const int y0 = tc.ty * R;
    const int x = xw0 + lane - 3;
    const bool col_ok = x >= 0 && x < W;
    const long long xoff = (long long)min(max(x, 0), W - 1) * 3;
    const bool out_lane = lane >= 3 && lane < 3 + kRegCols && x < W;
    float pb[7] = {0, 0, 0, 0, 0, 0, 0};  // pending blur rows: pb[k] = output row (newest input row) - 3 + k
    float hist[4][3];                     // input rows q, q-1, q-2, q-3 (own column)
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) hist[k][c] = 0.0f;
#pragma unroll 1
    for (int i0 = 0; i0 < R + 6; i0 += 4) {
        float g4[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 - 3 + i0 + j;
            const bool ok = y >= 0 && y < H && col_ok;
            const float* __restrict__ p = src + (long long)min(max(y, 0), H - 1) * W * 3 + xoff;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t = p[c];
                g4[j][c] = ok ? t : 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = y0 - 3 + i0 + j;  // arriving input row; blur row t = q - 3 completes
#pragma unroll
            for (int k = 3; k > 0; --k)
#pragma unroll
                for (int c = 0; c < 3; ++c) hist[k][c] = hist[k - 1][c];
#pragma unroll
            for (int c = 0; c < 3; ++c) hist[0][c] = g4[j][c];
            float s7[7];
            s7[3] = (g4[j][0] + g4[j][1]) + g4[j][2];
            s7[2] = from_lane_below(s7[3]);
            s7[1] = from_lane_below(s7[2]);
            s7[0] = from_lane_below(s7[1]);
            s7[4] = from_lane_above(s7[3]);
            s7[5] = from_lane_above(s7[4]);
            s7[6] = from_lane_above(s7[5]);
            float nb[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                kfloat_p kb = kb0 + (6 - k) * 7;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" : "+s"(kb));
                float kw[7];
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) kw[dx] = kb[dx];
                float acc = pb[k];
#pragma unroll
                for (int dx = 0; dx < 7; ++dx) acc = __builtin_fmaf(s7[dx], kw[dx], acc);
                asm volatile("" : "+v"(acc));
                nb[k] = acc;
            }
            const float bdone = nb[0];
#pragma unroll
            for (int k = 0; k < 6; ++k) pb[k] = nb[k + 1];
            pb[6] = 0.0f;
            const int t = q - 3;
            if (t >= y0 && t < y0 + R && t < H && out_lane) {
                const float m = bdone > 1.0f ? 1.0f : bdone;
                const float r = args.rv / powf(m, args.root);
                float* __restrict__ po = args.out + (base_px + (long long)t * W + x) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float xs = hist[3][c];
                    float yv = xs * r;
                    if (args.flat_policy == SILENT_FLAT_ZERO && xs == 0.0f) yv = 0.0f;
                    po[c] = yv;
                }
            }
        }
    }
}

}  // namespace silent
