// Shared host/device declarations for libsilent_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/silent_hip.h"

namespace silent {

constexpr int kMaxLevels = SILENT_MAX_LEVELS;
constexpr int kWave = 64;  // CDNA wavefront

// Tile -> (frame, level, tile_y, tile_x) lookup for one launch over a packed pyramid batch.
// Travels as a kernel argument (SGPR-resident after s_load), so the lookup is scalar work.
struct LevelTab {
    int n_levels;
    int tiles_per_frame;
    long long frame_px;  // pixels of one whole pyramid (all levels)
    int h[kMaxLevels];
    int w[kMaxLevels];
    int tiles_x[kMaxLevels];
    int tile_start[kMaxLevels + 1];  // prefix sum of tiles per level
    long long px_off[kMaxLevels];    // pixel offset of level l inside one pyramid
};

constexpr int kChunk = 1024;  // pixels per block for the 1-D (flattened level) kernels: 256 threads x 4

struct TileCoord {
    int frame, level, ty, tx;
};

// XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own 4 MiB L2), so
// with the identity order two tiles that share halo rows land on different L2s.  Remap so that every XCD walks
// a contiguous run of tiles: vertical neighbours are then a few blocks apart on the SAME L2.  Bijective for any
// grid size; placement is a speed hint only, never a correctness assumption.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned bid, unsigned n) {
    constexpr unsigned kXcd = 8;
    const unsigned per = n / kXcd, rem = n % kXcd;   // XCD x owns per (+1 if x < rem) consecutive tiles
    const unsigned x = bid % kXcd, k = bid / kXcd;
    return x * per + (x < rem ? x : rem) + k;
}

__device__ __forceinline__ TileCoord locate_tile(const LevelTab& tab, unsigned bid) {
    TileCoord t;
    t.frame = (int)(bid / (unsigned)tab.tiles_per_frame);
    int rem = (int)(bid - (unsigned)t.frame * (unsigned)tab.tiles_per_frame);
    int l = 0;
#pragma unroll
    for (int i = 1; i < kMaxLevels; ++i)
        if (i < tab.n_levels && rem >= tab.tile_start[i]) l = i;
    t.level = l;
    rem -= tab.tile_start[l];
    t.ty = rem / tab.tiles_x[l];
    t.tx = rem - t.ty * tab.tiles_x[l];
    return t;
}

// Neighbour exchange inside a wave by DPP wavefront shifts (VALU, no LDS round trip).
// lane i receives lane i-1 (lane 0 gets 0) / lane i+1 (lane 63 gets 0): bound_ctrl with a zero "old" value is a
// single v_mov_b32_dpp; keeping the lane's own value instead costs an extra v_mov per shift.  The edge lanes of a
// wave are halo lanes in every kernel that uses these.
__device__ __forceinline__ float from_lane_below(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_lane_above(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

// scipy 'mirror' extension (d c b | a b c d | c b a) for an index within ONE reflection of [0, n): exact for
// -(n-1) <= i <= 2(n-1); anything further out is clamped into range (such taps only feed outputs that are
// never stored).  The streaming kernels reach at most 4 pixels outside, hence kMirrorNearMin.
constexpr int kMirrorNearMin = 5;
__device__ __forceinline__ int mirror_near(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    return min(max(i, 0), n - 1);
}

// The horizontal taps of a unit (zoom 1) level for one streamed row of a lane-per-column wave.  scipy.ndimage.zoom(order=5)
// (from_image.py:55-59) evaluates SIX taps, x - 2 .. x + 3; at zoom 1 the sixth weight is what double arithmetic leaves of
// 1 - (1 + 26 + 66 + 26 + 1) / 120 = 2^-53: nothing next to finite neighbours, but a NaN / inf pixel reaches the outputs three
// to its left (and three above it) as well, and a lone bright pixel leaves 2^-53 of itself there.  The sixth tap of lane 61 is
// the column right of the wave's 64: `edge` carries it (wave-uniform: one lane of a per-tile load, unit_edge_column) and enters
// as what lane 63 "receives from lane 64" in the first shift (a DPP shift without bound_ctrl keeps the destination's old value
// in the lane that has no source), so r2 of lane 62 and r3 of lane 61 see it too.
// EDGE = false: lane 61 is no output's tap (kernels whose halo lanes already cover x + 3).
template <bool EDGE = true>
__device__ __forceinline__ float unit_taps6(float c0, float edge, const float (&w)[6]) {
    const float l1 = from_lane_below(c0), l2 = from_lane_below(l1);
    float r1;
    if constexpr (EDGE)
        r1 = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(c0), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
    else
        r1 = from_lane_above(c0);
    const float r2 = from_lane_above(r1), r3 = from_lane_above(r2);
    float h = w[0] * l2;
    h = __builtin_fmaf(w[1], l1, h);
    h = __builtin_fmaf(w[2], c0, h);
    h = __builtin_fmaf(w[3], r1, h);
    h = __builtin_fmaf(w[4], r2, h);
    return __builtin_fmaf(w[5], r3, h);
}
// Stream row i of the column `col` (level coordinates, mirrored inside the crop like every tap; px_stride floats per pixel) in lane i: ONE load per tile
// for the sixth taps of a wave's last smoothing lane; row i is read back with unit_edge(xcol, i).
__device__ __forceinline__ float unit_edge_column(const float* __restrict__ src, long long row_stride, int col, int src_w, int src_x0,
                                                  int y_first, int n_rows, int src_h, int src_y0, int lane, int px_stride = 1) {
    const long long sx = (long long)(mirror_near(col, src_w) + src_x0) * px_stride;
    return src[(long long)(mirror_near(y_first + min(lane, n_rows - 1), src_h) + src_y0) * row_stride + sx];
}
__device__ __forceinline__ float unit_edge(float xcol, int i) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xcol), i));
}

// tf.maximum(x, [0]) with Eigen's CPU functor: a NaN stays a NaN (oracle: relu_tf).
__device__ __forceinline__ float relu_tf(float v) { return v < 0.0f ? 0.0f : v; }
__device__ __forceinline__ float clip_hi_tf(float v, float hi) { return v > hi ? hi : v; }

// relu + clip of N accumulators of one pixel: clip(relu(x), hi) with the reference's NaN behaviour.  The select forms
// ((x < 0) ? 0 : x, (x > hi) ? hi : x) keep a NaN a NaN but are 4 instructions per value, serialised through VCC with hazard
// nops.  For every non-NaN x they equal the median of (x, 0, hi) -- one v_med3_f32 -- when hi >= 0 (the fma chains start
// from +0, so x is never -0).  The N raw values are summed first: the sum is a NaN iff one of them is (or +inf meets -inf),
// and only then the wave takes the select form.  Wave-uniform branch; every lane of the wave must call this.
template <int N>
__device__ __forceinline__ void relu_clip_tf(float (&v)[N], float hi) {
    float chk = v[0];
#pragma unroll
    for (int k = 1; k < N; ++k) chk += v[k];
    if (hi >= 0.0f && !__any(chk != chk)) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = __builtin_amdgcn_fmed3f(v[k], 0.0f, hi);
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = clip_hi_tf(relu_tf(v[k]), hi);
    }
}

// Value summary written by the fused RGB chain (rgb_line_end2_kernel, MM instantiation) for the sparse selection tail:
// entry = max_pool(value) over one pixel PAIR x kSumRows rows of a chain tile.
constexpr int kSumRowsLog2 = 4, kSumRows = 1 << kSumRowsLog2;
// How the sparse tail (sparse_select_kernel, silent_peaks.h) settles a (frame, level): by its candidates alone (every window maximum > 0); candidates + a synthesised all-zero
// map in the count pass (some window without a positive peak, but the level holds no NaN: every pixel mapped to such a window
// is a keypoint); or the dense kernels (such a window AND NaNs in the level, or too many candidates)
constexpr int kTailSparse = 0, kTailDense = 1, kTailZero = 2;

// order-preserving float <-> uint map so that integer atomics give float max / min
__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// tf.nn.max_pool as the reference's device path evaluates it.  The reference pins its graph to '/device:GPU:0'
// (slam_recognition/recognition_testing.py:64); there TF 1.x runs MaxPoolForwardNHWC (`maxval = lowest(); if (x > maxval)
// maxval = x`) or cuDNN with CUDNN_NOT_PROPAGATE_NAN (TF_ENABLE_MAXPOOL_NANPROP defaults to false): a NaN never wins, the
// result is independent of the tap order, a window with nothing above lowest() yields lowest() = -FLT_MAX.
// Every maximum in silent_peaks.h (3x3 NMS, per-level max / min, window and cell maxima) goes through pool_max with the
// RUNNING maximum as first argument, so an accumulator is never a NaN and atomics only ever see ordered floats.
// Oracle: pool_max in oracle/silent_oracle.py.
constexpr float kPoolLowest = -3.402823466e+38f;
__device__ __forceinline__ float pool_max(float m, float v) { return v > m ? v : m; }
__device__ __forceinline__ unsigned pool_lowest_ord() { return f2ord(kPoolLowest); }

// v must already be NaN-free (a pool_max accumulator)
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = pool_max(v, __shfl_xor(v, o));
    return v;
}

// Geometry of that value summary (host: build_sum_tab in silent_peaks_api.hip)
struct SumTab {
    int th, gpt;                       // tile height of the chain launch, groups per tile = ceil(th / kSumRows)
    long long frame_entries;
    long long off[kMaxLevels + 1];     // entry offset of level l inside a frame (off[n_levels] = frame_entries)
};

}  // namespace silent
