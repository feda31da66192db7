/* CPU ORACLE (C, OpenMP) for the pySILEnT line-end hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Same algorithms as oracle/silent_oracle.py (see its header for the reference file:line of
 * every op and for the pinning status: constant kernels pinned by the reference's generators,
 * per-frame ops PARITY UNPINNED against the reference because TensorFlow cannot run here;
 * the resampler is pinned by scipy.ndimage.zoom itself).  Exists so that (a) parity tests can
 * run at full 1080p/4K sizes in seconds and (b) bench.py can time a multi-core CPU baseline
 * ("cpu_baseline.kind" = "port") on the GPU box's host cores.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared).   No reference source is compiled or copied.
 *
 * Conventions: activations NHWC float32, kernels HWIO float32, stride 1, SAME zero padding,
 * cross-correlation.  Inside one op: float64 accumulation, one rounding to float32.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SO_API __attribute__((visibility("default")))

SO_API int so_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

SO_API void so_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* tf.maximum(x, 0) with Eigen's functor: a NaN stays a NaN */
static inline float relu_tf(float v) { return v < 0.0f ? 0.0f : v; }
static inline float clip_hi_tf(float v, float hi) { return v > hi ? hi : v; }

/* flags: bit0 relu, bit1 clip to [0, clip_hi] */
SO_API void so_conv2d_same(const float* in, int n, int h, int w, int ci, const float* k, int kh, int kw,
                           int co, int flags, float clip_hi, float* out) {
    const int ph = (kh - 1) / 2, pw = (kw - 1) / 2;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int y = 0; y < h; ++y) {
            double acc[16];
            for (int x = 0; x < w; ++x) {
                for (int o = 0; o < co; ++o) acc[o] = 0.0;
                for (int dy = 0; dy < kh; ++dy) {
                    const int yy = y + dy - ph;
                    if (yy < 0 || yy >= h) continue;
                    for (int dx = 0; dx < kw; ++dx) {
                        const int xx = x + dx - pw;
                        if (xx < 0 || xx >= w) continue;
                        const float* px = in + (((size_t)b * h + yy) * w + xx) * ci;
                        const float* kk = k + ((size_t)(dy * kw + dx) * ci) * co;
                        for (int i = 0; i < ci; ++i) {
                            const double v = px[i];
                            for (int o = 0; o < co; ++o) acc[o] += v * (double)kk[i * co + o];
                        }
                    }
                }
                float* po = out + (((size_t)b * h + y) * w + x) * co;
                for (int o = 0; o < co; ++o) {
                    float v = (float)acc[o];
                    if (flags & 1) v = relu_tf(v);
                    if (flags & 2) v = clip_hi_tf(relu_tf(v), clip_hi);
                    po[o] = v;
                }
            }
        }
}

/* y = x * (rv / pow(min(blur(x), 1), root)); flat_policy 0 = ieee, 1 = zero */
SO_API void so_regulate(const float* in, int n, int h, int w, int c, const float* blur, int kh, int kw,
                        float rv, float root, int flat_policy, float* out) {
    float* b = (float*)malloc((size_t)n * h * w * c * sizeof(float));
    so_conv2d_same(in, n, h, w, c, blur, kh, kw, c, 0, 0.0f, b);
    const size_t tot = (size_t)n * h * w * c;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < tot; ++i) {
        const float m = b[i] > 1.0f ? 1.0f : b[i];
        const float p = (float)pow((double)m, (double)root);
        const float r = rv / p;
        float y = in[i] * r;
        if (flat_policy == 1 && in[i] == 0.0f) y = 0.0f;
        out[i] = y;
    }
    free(b);
}

SO_API void so_pad_inwards(const float* in, int n, int h, int w, int c, int pt, int pb, int pl, int pr,
                           float* out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const float m = (y >= pt && y < h - pb && x >= pl && x < w - pr) ? 1.0f : 0.0f;
                const size_t o = (((size_t)b * h + y) * w + x) * c;
                for (int i = 0; i < c; ++i) out[o + i] = m * in[o + i];
            }
}

SO_API void so_value_from_color(const float* in, size_t npx, int c, float* out) {
    const float inv = 1.0f / (float)c;
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < npx; ++p) {
        float s = in[p * c];
        for (int i = 1; i < c; ++i) s = s + in[p * c + i];
        out[p] = s * inv;
    }
}

/* tf.nn.max_pool on the reference's device path ('/device:GPU:0'): TF 1.x MaxPoolForwardNHWC starts from
 * lowest() and takes `x > maxval`, cuDNN runs with CUDNN_NOT_PROPAGATE_NAN (TF_ENABLE_MAXPOOL_NANPROP = false):
 * a NaN never wins, an all-NaN window gives lowest().  See pool_max in silent_oracle.py. */
#define POOL_LOWEST (-3.402823466e+38f)
static inline float pool_max(float m, float v) { return v > m ? v : m; }

/* mode 0: x * where(x == maxpool3x3(x), x, 0); mode 1: fired mask */
SO_API void so_nms3x3(const float* in, int n, int h, int w, int c, int mode, float* out) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; ++b)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                for (int i = 0; i < c; ++i) {
                    float m = POOL_LOWEST;
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int yy = y + dy;
                        if (yy < 0 || yy >= h) continue;
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int xx = x + dx;
                            if (xx < 0 || xx >= w) continue;
                            m = pool_max(m, in[(((size_t)b * h + yy) * w + xx) * c + i]);
                        }
                    }
                    const size_t o = (((size_t)b * h + y) * w + x) * c + i;
                    const float v = in[o];
                    const int is_max = (v == m);
                    out[o] = mode == 1 ? (is_max ? 1.0f : 0.0f) : v * (is_max ? v : 0.0f);
                }
}

SO_API void so_level_max_min(const float* v, int n, size_t npx, float* mx, float* mn) {
    for (int b = 0; b < n; ++b) {
        /* max_pool(v) and -1.0 * max_pool(-v), NaNs ignored (top_value_points.py:16-21) */
        float hi = POOL_LOWEST, nlo = POOL_LOWEST;
        const float* p = v + (size_t)b * npx;
#pragma omp parallel for reduction(max : hi) reduction(max : nlo) schedule(static)
        for (size_t i = 0; i < npx; ++i) {
            if (p[i] > hi) hi = p[i];
            if (-p[i] > nlo) nlo = -p[i];
        }
        mx[b] = hi;
        mn[b] = -1.0f * nlo;
    }
}

SO_API void so_top_value_points(const float* color, const float* value, int n, int h, int w, int c,
                                double top_percent, float* out) {
    const size_t npx = (size_t)h * w;
    float* mx = (float*)malloc(sizeof(float) * n * 2);
    float* mn = mx + n;
    so_level_max_min(value, n, npx, mx, mn);
    const float a = (float)(1.0 - top_percent); /* python: float32(1.0 - p) */
    const float pf = (float)top_percent;
    for (int b = 0; b < n; ++b) {
        volatile float t0 = a * mx[b];
        volatile float t1 = pf * mn[b];
        const float thr = t0 + t1;
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < npx; ++p) {
            const float m = value[(size_t)b * npx + p] >= thr ? 1.0f : 0.0f;
            for (int i = 0; i < c; ++i) out[((size_t)b * npx + p) * c + i] = color[((size_t)b * npx + p) * c + i] * m;
        }
    }
    free(mx);
}

static void region_geometry(int size, int stride, int* n_out, int* lo, int* hi, int* src) {
    const int out = (size + stride - 1) / stride;
    const int pad_before = ((out - 1) * stride) / 2;
    for (int j = 0; j < out; ++j) {
        int a = j * stride - pad_before, b = a + size;
        lo[j] = a < 0 ? 0 : a;
        hi[j] = b > size ? size : b;
    }
    const float scale = (float)out / (float)size;
    for (int y = 0; y < size; ++y) {
        int s = (int)floorf((float)y * scale);
        src[y] = s > out - 1 ? out - 1 : s;
    }
    *n_out = out;
}

/* writes rows (n, y, x, 0) in row-major order; returns the total count (may exceed cap: then
 * only the first cap rows were written) */
SO_API int64_t so_max_value_indices_region(const float* value, int n, int h, int w, int rh, int rw,
                                           int64_t* idx, int64_t cap) {
    int oh, ow;
    int* ylo = (int*)malloc(sizeof(int) * (size_t)(3 * h + 3 * w));
    int *yhi = ylo + h, *ysrc = yhi + h, *xlo = ysrc + h, *xhi = xlo + w, *xsrc = xhi + w;
    region_geometry(h, rh, &oh, ylo, yhi, ysrc);
    region_geometry(w, rw, &ow, xlo, xhi, xsrc);
    float* pooled = (float*)malloc(sizeof(float) * (size_t)oh * ow);
    int64_t count = 0;
    for (int b = 0; b < n; ++b) {
        const float* v = value + (size_t)b * h * w;
        for (int j = 0; j < oh; ++j)
            for (int i = 0; i < ow; ++i) {
                float m = POOL_LOWEST;
                for (int y = ylo[j]; y < yhi[j]; ++y)
                    for (int x = xlo[i]; x < xhi[i]; ++x) m = pool_max(m, v[(size_t)y * w + x]);
                pooled[j * ow + i] = m;
            }
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x)
                if (v[(size_t)y * w + x] >= pooled[ysrc[y] * ow + xsrc[x]]) {
                    if (count < cap) {
                        idx[count * 4 + 0] = b;
                        idx[count * 4 + 1] = y;
                        idx[count * 4 + 2] = x;
                        idx[count * 4 + 3] = 0;
                    }
                    ++count;
                }
    }
    free(pooled);
    free(ylo);
    return count;
}

/* ------------------------------------------------------------------------------------------ pyramid */

static void spline5_weights(double t, double* w) {
    const double y = t, z = 1.0 - t;
    double t2 = y * y;
    w[2] = t2 * (t2 * (0.25 - y / 12.0) - 0.5) + 0.55;
    t2 = z * z;
    w[3] = t2 * (t2 * (0.25 - z / 12.0) - 0.5) + 0.55;
    const double y1 = y + 1.0;
    w[1] = y1 * (y1 * (y1 * (y1 * (y1 / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
    const double z1 = z + 1.0;
    w[4] = z1 * (z1 * (z1 * (z1 * (z1 / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
    const double y2 = 1.0 - y;
    w[0] = y2 * y2 * y2 * y2 * y2 / 120.0;
    w[5] = 1.0 - w[0] - w[1] - w[2] - w[3] - w[4];
}

static int mirror_index(long i, int n) {
    if (n == 1) return 0;
    const long period = 2L * (n - 1);
    if (i < 0) i = -i;
    i %= period;
    return (int)(i >= n ? period - i : i);
}

static void axis_table(int n_in, int n_out, int* idx, double* wts) {
    const double step = n_out > 1 ? (double)(n_in - 1) / (double)(n_out - 1) : 1.0;
    for (int o = 0; o < n_out; ++o) {
        const double c = (double)o * step;
        const long b = (long)floor(c);
        if (c >= 0.0 && c <= (double)(n_in - 1))
            spline5_weights(c - (double)b, wts + 6 * o);
        else
            for (int j = 0; j < 6; ++j) wts[6 * o + j] = 0.0; /* scipy mode='constant': out of bounds -> cval 0 */
        for (int j = 0; j < 6; ++j) idx[6 * o + j] = mirror_index(b - 2 + j, n_in);
    }
}

/* One level of the zoom pyramid: crop [y0,y0+ch) x [x0,x0+cw) of an H x W x C frame, resampled by the
 * un-prefiltered quintic B-spline to zh x zw, copied into the top-left of an oh x ow canvas (rest 0). */
SO_API void so_zoom_level(const float* frame, int H, int W, int C, int y0, int x0, int ch, int cw, int zh,
                          int zw, int oh, int ow, float* out) {
    (void)H;
    int* iy = (int*)malloc(sizeof(int) * 6 * (size_t)(zh + zw));
    int* ix = iy + 6 * (size_t)zh;
    double* wy = (double*)malloc(sizeof(double) * 6 * (size_t)(zh + zw));
    double* wx = wy + 6 * (size_t)zh;
    axis_table(ch, zh, iy, wy);
    axis_table(cw, zw, ix, wx);
    const int ym = zh < oh ? zh : oh, xm = zw < ow ? zw : ow;
    memset(out, 0, sizeof(float) * (size_t)oh * ow * C);
#pragma omp parallel for schedule(static)
    for (int oy = 0; oy < ym; ++oy)
        for (int ox = 0; ox < xm; ++ox)
            for (int c = 0; c < C; ++c) {
                double acc = 0.0;
                for (int a = 0; a < 6; ++a) {
                    const float* row = frame + ((size_t)(y0 + iy[6 * oy + a]) * W + x0) * C + c;
                    for (int b = 0; b < 6; ++b)
                        acc += (wy[6 * oy + a] * wx[6 * ox + b]) * (double)row[(size_t)ix[6 * ox + b] * C];
                }
                out[((size_t)oy * ow + ox) * C + c] = (float)acc;
            }
    free(iy);
    free(wy);
}

/* BASELINE config 1/2/5 chain on one level (n = 1): CS -> ReLU -> K end bank -> ReLU -> clip */
SO_API void so_gray_line_end_level(const float* lev, int h, int w, const float* cs_k, const float* end_k,
                                   int K, float clip_hi, float* cs_out, float* end_out) {
    so_conv2d_same(lev, 1, h, w, 1, cs_k, 3, 3, 1, 1, 0.0f, cs_out);
    so_conv2d_same(cs_out, 1, h, w, 1, end_k, 3, 3, K, 3, clip_hi, end_out);
}

/* Whole gray pass (BASELINE configs 2/5) on a BATCH of frames, one frame per OpenMP thread: classic pyramid
 * (every level = the whole frame resampled to extents[l]) -> CS -> ReLU -> K end bank -> ReLU -> clip.
 * The per-op `omp parallel for`s above become serial inside this region (nested parallelism is off by
 * default), so the host cores are used across frames -- bench.py's cpu_baseline leg; outputs are dropped
 * except for a checksum that keeps the work alive.  Returns the sum of all end-map values. */
SO_API double so_gray_pass_frames(const float* frames, int n_frames, int H, int W, const int* extents /* [L][2] */,
                                  int L, const float* cs_k, const float* end_k, int K, float clip_hi) {
    double total = 0.0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int f = 0; f < n_frames; ++f) {
        const float* frame = frames + (size_t)f * H * W;
        double acc = 0.0;
        for (int l = 0; l < L; ++l) {
            const int h = extents[2 * l], w = extents[2 * l + 1];
            float* lev = (float*)malloc(sizeof(float) * (size_t)h * w * (2 + K));
            float* cs = lev + (size_t)h * w;
            float* end = cs + (size_t)h * w;
            so_zoom_level(frame, H, W, 1, 0, 0, H, W, h, w, h, w, lev);
            so_gray_line_end_level(lev, h, w, cs_k, end_k, K, clip_hi, cs, end);
            for (size_t i = 0; i < (size_t)h * w * K; i += 97) acc += end[i];
            free(lev);
        }
        total += acc;
    }
    return total;
}

/* BASELINE config 3 on a BATCH of frames, one frame per OpenMP thread (bench.py's cpu_baseline leg for the RGB workload; also
 * checked against the per-op composition in tests/test_oracle.py): classic pyramid -> per level the reference graph
 * recognition_testing.py:69-77 (rgc > rgby > stripe > regulate(blur 7x7, rv, root) > end > relu > clip > pad_inwards > value)
 * -> top_value_points(p) (a-10) -> 3x3 NMS, product form (a-9) -> value (a-8) -> max_value_indices_region with
 * region = (h / 2, w / 2) (a-11).  Rows (level, y, x, 0), levels in order, row-major inside a level, go to
 * idx + f * cap * 4 when idx is not NULL (only the first cap of a frame); counts[f] = rows of frame f.  Returns the total. */
SO_API int64_t so_rgb_pass_frames(const float* frames, int n_frames, int H, int W, const int* extents /* [L][2] */, int L,
                                  const float* rgc, const float* rgby, const float* stripe, const float* blur,
                                  const float* end, float rv, float root, int flat_policy, float clip_hi, int pad,
                                  double top_percent, int64_t* idx, int64_t cap, int64_t* counts) {
    int64_t total = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
    for (int f = 0; f < n_frames; ++f) {
        const float* frame = frames + (size_t)f * H * W * 3;
        int64_t n_rows = 0;
        for (int l = 0; l < L; ++l) {
            const int h = extents[2 * l], w = extents[2 * l + 1];
            const size_t n3 = (size_t)h * w * 3;
            float* a = (float*)malloc(sizeof(float) * (3 * n3 + (size_t)h * w));
            float *b = a + n3, *c = b + n3, *v = c + n3;
            so_zoom_level(frame, H, W, 3, 0, 0, H, W, h, w, h, w, a);
            so_conv2d_same(a, 1, h, w, 3, rgc, 3, 3, 3, 1, 0.0f, b);
            so_conv2d_same(b, 1, h, w, 3, rgby, 3, 3, 3, 1, 0.0f, a);
            so_conv2d_same(a, 1, h, w, 3, stripe, 3, 3, 3, 1, 0.0f, b);
            so_regulate(b, 1, h, w, 3, blur, 7, 7, rv, root, flat_policy, a);          /* orient */
            so_conv2d_same(a, 1, h, w, 3, end, 3, 3, 3, 3, clip_hi, b);                /* line_end */
            so_pad_inwards(b, 1, h, w, 3, pad, pad, pad, pad, a);                      /* padded */
            so_value_from_color(a, (size_t)h * w, 3, v);
            so_top_value_points(a, v, 1, h, w, 3, top_percent, b);
            so_nms3x3(b, 1, h, w, 3, 0, c);
            so_value_from_color(c, (size_t)h * w, 3, v);                               /* peak value */
            int64_t* dst = idx ? idx + ((size_t)f * cap + (size_t)(n_rows < cap ? n_rows : cap)) * 4 : NULL;
            const int64_t room = idx ? (n_rows < cap ? cap - n_rows : 0) : 0;
            const int64_t got = so_max_value_indices_region(v, 1, h, w, h / 2 > 1 ? h / 2 : 1, w / 2 > 1 ? w / 2 : 1, dst, room);
            if (dst)
                for (int64_t r = 0; r < (got < room ? got : room); ++r) dst[r * 4] = l;
            n_rows += got;
            free(a);
        }
        if (counts) counts[f] = n_rows;
        total += n_rows;
    }
    return total;
}
