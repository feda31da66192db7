// Stencil kernels: generic SAME conv2d (NHWC x HWIO) and the fused grayscale CS -> end-bank pass.
// HBM-bound stencil/pointwise work: coalesced tile loads into LDS, filters unrolled in registers,
// weights held in SGPRs (they arrive as kernel arguments), no MFMA.
#pragma once

#include "silent_common.h"

namespace silent {

struct ConvW {
    float w[SILENT_MAX_KERNEL_FLOATS];
};

constexpr int kConvTW = 64;  // one wave spans a tile row: 64 consecutive pixels
constexpr int kConvTH = 16;

// out[y,x,o] = sum_{dy,dx,i} in[y+dy-PH, x+dx-PW, i] * K[dy,dx,i,o]   (zero outside the level)
// Block = 256 threads = 4 waves; lane = column, each wave owns 4 rows of the 64 x 16 tile.
//
// REG = true turns the kernel into regulate_tensor (gaussian_regulator_tensor.py:34-36): the
// convolution is the blur (C_out == C_in) and the epilogue is y = x * (rv / pow(min(b, 1), root)),
// x taken from the already staged tile.
struct Epilogue {
    unsigned flags;   // SILENT_RELU | SILENT_CLIP           (REG = false)
    float clip_hi;
    float rv, root;   // regulation value / root             (REG = true)
    int flat_policy;  // SILENT_FLAT_*
};

__device__ __forceinline__ float regulate_px(float x, float b, const Epilogue& ep) {
    const float m = b > 1.0f ? 1.0f : b;  // tf.minimum(b, [1])
    const float p = powf(m, ep.root);
    const float r = ep.rv / p;
    float y = x * r;
    if (ep.flat_policy == SILENT_FLAT_ZERO && x == 0.0f) y = 0.0f;
    return y;
}

template <int KH, int KW, int CIN, int COUT, bool REG>
__global__ __launch_bounds__(256) void conv2d_same_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          const LevelTab tab, const ConvW wts, const Epilogue ep) {
    static_assert(!REG || CIN == COUT, "regulate needs a square blur");
    constexpr int TW = kConvTW, TH = kConvTH;
    constexpr int PH = (KH - 1) / 2, PW = (KW - 1) / 2;
    constexpr int IW = TW + KW - 1, IH = TH + KH - 1;
    constexpr int ROWF = IW * CIN;  // floats per LDS row
    __shared__ float s_in[IH * ROWF];

    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const float* __restrict__ src = in + base_px * CIN;
    float* __restrict__ dst = out + base_px * COUT;
    const int x0 = tc.tx * TW, y0 = tc.ty * TH;
    const int tid = threadIdx.x;

    // stage the (TH+KH-1) x (TW+KW-1) x CIN halo tile; consecutive threads read consecutive floats.
    // Predicated batch (clamped address + select, no branch around a load): every thread has all its
    // requests in flight before the first wait.
    {
        constexpr int NB = (IH * ROWF + 255) / 256;
        float v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = min(tid + 256 * k, IH * ROWF - 1);
            const int r = p / ROWF;
            const int rem = p - r * ROWF;
            const int c = rem / CIN;
            const int ch = rem - c * CIN;
            const int y = y0 + r - PH, x = x0 + c - PW;
            const bool ok = y >= 0 && y < H && x >= 0 && x < W;
            const float t = src[((long long)min(max(y, 0), H - 1) * W + min(max(x, 0), W - 1)) * CIN + ch];
            v[k] = ok ? t : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int p = tid + 256 * k;
            if (p < IH * ROWF) s_in[p] = v[k];
        }
    }
    __syncthreads();

    const int col = tid & 63, wave = tid >> 6;
    const int x = x0 + col;
    if (x >= W) return;
#pragma unroll 1
    for (int rr = 0; rr < TH / 4; ++rr) {
        const int r = wave * (TH / 4) + rr;
        const int y = y0 + r;
        if (y >= H) break;
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = 0.0f;
#pragma unroll
        for (int dy = 0; dy < KH; ++dy)
#pragma unroll
            for (int dx = 0; dx < KW; ++dx)
#pragma unroll
                for (int i = 0; i < CIN; ++i) {
                    const float v = s_in[(r + dy) * ROWF + (col + dx) * CIN + i];
#pragma unroll
                    for (int o = 0; o < COUT; ++o)
                        acc[o] = __builtin_fmaf(v, wts.w[((dy * KW + dx) * CIN + i) * COUT + o], acc[o]);
                }
        float* __restrict__ po = dst + ((long long)y * W + x) * COUT;
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            float v = acc[o];
            if constexpr (REG) {
                v = regulate_px(s_in[(r + PH) * ROWF + (col + PW) * CIN + o], v, ep);
            } else {
                if (ep.flags & SILENT_RELU) v = relu_tf(v);
                if (ep.flags & SILENT_CLIP) v = clip_hi_tf(relu_tf(v), ep.clip_hi);
            }
            acc[o] = v;
        }
        if constexpr (COUT == 4) {
            *reinterpret_cast<float4*>(po) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        } else if constexpr (COUT == 8) {
            reinterpret_cast<float4*>(po)[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            reinterpret_cast<float4*>(po)[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        } else {
#pragma unroll
            for (int o = 0; o < COUT; ++o) po[o] = acc[o];
        }
    }
}

// Any other (kh, kw, C_in, C_out) within the weight budget: same tiling, runtime loops,
// dynamic LDS.  Correct, not tuned -- the compiled set above covers every reference call site.
__global__ __launch_bounds__(256) void conv2d_same_generic_kernel(const float* __restrict__ in,
                                                                  float* __restrict__ out, const LevelTab tab,
                                                                  const ConvW wts, int KH, int KW, int CIN,
                                                                  int COUT, int reg, const Epilogue ep) {
    constexpr int TW = kConvTW, TH = kConvTH;
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    const int PH = (KH - 1) / 2, PW = (KW - 1) / 2;
    const int IW = TW + KW - 1, IH = TH + KH - 1;
    const int ROWF = IW * CIN;

    const TileCoord tc = locate_tile(tab, blockIdx.x);
    const int H = tab.h[tc.level], W = tab.w[tc.level];
    const long long base_px = (long long)tc.frame * tab.frame_px + tab.px_off[tc.level];
    const float* __restrict__ src = in + base_px * CIN;
    float* __restrict__ dst = out + base_px * COUT;
    const int x0 = tc.tx * TW, y0 = tc.ty * TH;
    const int tid = threadIdx.x;

    for (int p = tid; p < IH * ROWF; p += 256) {
        const int r = p / ROWF;
        const int rem = p - r * ROWF;
        const int c = rem / CIN;
        const int ch = rem - c * CIN;
        const int y = y0 + r - PH, x = x0 + c - PW;
        float v = 0.0f;
        if (y >= 0 && y < H && x >= 0 && x < W) v = src[((long long)y * W + x) * CIN + ch];
        s_dyn[p] = v;
    }
    __syncthreads();

    const int col = tid & 63, wave = tid >> 6;
    const int x = x0 + col;
    if (x >= W) return;
    for (int rr = 0; rr < TH / 4; ++rr) {
        const int r = wave * (TH / 4) + rr;
        const int y = y0 + r;
        if (y >= H) break;
        float* __restrict__ po = dst + ((long long)y * W + x) * COUT;
        for (int o = 0; o < COUT; ++o) {
            float acc = 0.0f;
            for (int dy = 0; dy < KH; ++dy)
                for (int dx = 0; dx < KW; ++dx)
                    for (int i = 0; i < CIN; ++i)
                        acc = __builtin_fmaf(s_dyn[(r + dy) * ROWF + (col + dx) * CIN + i],
                                             wts.w[((dy * KW + dx) * CIN + i) * COUT + o], acc);
            if (reg) {
                acc = regulate_px(s_dyn[(r + PH) * ROWF + (col + PW) * CIN + o], acc, ep);
            } else {
                if (ep.flags & SILENT_RELU) acc = relu_tf(acc);
                if (ep.flags & SILENT_CLIP) acc = clip_hi_tf(relu_tf(acc), ep.clip_hi);
            }
            po[o] = acc;
        }
    }
}

}  // namespace silent
