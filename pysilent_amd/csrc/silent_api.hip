// libsilent_hip.so -- C ABI (include/silent_hip.h) over the gfx950 kernels.
// Host side: argument validation, tile tables, tap tables (float64, scipy-identical), staging arena.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "silent_host_shim.h"   // (inert unless SILENT_HOST_ONLY: the CPU container's sanitizer build of the host side)
#include "silent_common.h"
#include "silent_conv.h"
#include "silent_peaks.h"
#include "silent_pyramid.h"
#include "silent_rgb.h"
#include "silent_rgb2.h"
#include "silent_walk_rgb.h"

using namespace silent;

#define SILENT_EXPORT extern "C" __attribute__((visibility("default")))

// ------------------------------------------------------------------------------------------ context

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct silent_ctx {
    int device = 0;
    std::string err;
    std::string name;
    int n_cus = 256;
    DevBuf arena;  // staging for the host-pointer entry points
    DevBuf ws;     // scratch for reductions / compaction / the RGB chain temporaries
    hipStream_t ws_stream = nullptr;   // the stream whose work last used ws (see workspace())
    bool ws_used = false;
    bool profiling = false;
    // HIP-event sampling of the dominant kernel of silent_gray_pass_dev: every prof_period-th call records a pair
    // into a ring of kProfPairs, silent_profile_elapsed_ms averages the recorded ones
    static constexpr int kProfPairs = 8;
    hipEvent_t prof_ev[kProfPairs][2] = {};
    int prof_period = 1, prof_calls = 0, prof_recorded = 0;
    bool prof_sample = false;
    long long prof_pixels = 0;
    // kernel-selection knobs (silent_set_tuning; initial values from SILENT_GRAY_OPTS / SILENT_RGB_OPTS /
    // SILENT_PYRAMID_OPTS read ONCE, in silent_create): tests and A/B scripts pick alternative kernels with them
    unsigned tune[SILENT_TUNE_COUNT] = {0, 0, 0};
    // the last silent_rgb_keypoints_dev call's sparse tail (silent_sparse_tail_stats): where its flags / counters live in ws
    bool sparse_ran = false;
    hipStream_t sparse_stream = nullptr;
    size_t sparse_flags_off = 0, sparse_candn_off = 0;
    int sparse_pairs = 0, sparse_frames = 0;
};

// Entry points run on the context's device and put the caller's device back before they return: torch tracks its
// current device through hipGetDevice, so a context on another GPU must not move it.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) {
            (void)hipGetLastError();
            cur = -1;
        }
        if (cur != dev) {
            ok = hipSetDevice(dev) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            prev = cur;
        }
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

struct silent_pyramid_plan {
    silent_ctx* ctx = nullptr;
    PyrTab tab{};
    std::vector<silent_extent> extents;
    void* tables = nullptr;
    float unit_w[5] = {0, 0, 0, 0, 0};  // taps of a unit-zoom level ([1,26,66,26,1]/120 as float32)
    // single-read "stream" path (gray_stream_kernel): row programs + column records, when the plan is eligible
    bool stream_ok = false;
    void* stream_tables = nullptr;
    StreamTab stream{};
    int stream_unit_level = -1;
    // walk plans of pyramid_walk3_kernel (silent_walk_rgb.h; 3 channels): a classic pyramid is ONE plan (unit level + every
    // other level on the same crop), a crop layout like the reference's one plan per level; row programs (completion records)
    // + column records per wave tile live in walk_tables
    bool walk_pyr_ok = false;
    int walk_px = 36;                    // pixels per consumer wave: 36, or 32 for zoom steps below 1.875
    int walk_G = 4;                      // general levels the kernel is instantiated for (4 or 7)
    void* walk_tables = nullptr;
    Walk3Args walk{};                    // everything but the per-launch decomposition (strips / segments / block0)
};

static thread_local std::string g_create_err;

static int fail(silent_ctx* ctx, int code, const std::string& msg) {
    if (ctx)
        ctx->err = msg;
    else
        g_create_err = msg;
    return code;
}

// ------------------------------------------------------------------------------------------ exception barrier
// include/silent_hip.h promises that nothing throws or aborts across the ABI.  The host side allocates (std::vector, std::string):
// every extern "C" entry point is a function-try-block whose handler turns std::bad_alloc into SILENT_E_NOMEM and anything else
// into SILENT_E_INVALID (the message says what was thrown).  The handler itself must not throw: setting the message allocates.
static int on_exception(silent_ctx* ctx, const char* who) noexcept {
    int code = SILENT_E_INVALID;
    try {
        throw;
    } catch (const std::bad_alloc&) {
        code = SILENT_E_NOMEM;
        try {
            fail(ctx, code, std::string(who) + ": out of host memory");
        } catch (...) {
        }
    } catch (const std::exception& e) {
        try {
            fail(ctx, code, std::string(who) + ": unexpected exception: " + e.what());
        } catch (...) {
        }
    } catch (...) {
        try {
            fail(ctx, code, std::string(who) + ": unexpected exception");
        } catch (...) {
        }
    }
    return code;
}

#define HIP_TRY(ctx, call)                                                                            \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            (void)hipGetLastError();                                                                  \
            return fail(ctx, e_ == hipErrorOutOfMemory ? SILENT_E_NOMEM : SILENT_E_HIP,               \
                        std::string(#call) + ": " + hipGetErrorString(e_));                           \
        }                                                                                             \
    } while (0)

#define TRY(expr)                   \
    do {                            \
        int rc_ = (expr);           \
        if (rc_ != SILENT_OK) return rc_; \
    } while (0)

static int grow(silent_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return SILENT_OK;
    if (b.p) {
        HIP_TRY(ctx, hipFree(b.p));  // synchronises with work still using the old block
        b.p = nullptr;
        b.cap = 0;
    }
    const size_t want = bytes + bytes / 4 + (1u << 20);
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    return SILENT_OK;
}

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// The context has ONE workspace, so its users are ordered by the stream they run on.  A caller that moves to another
// stream is not an error: the previous stream is drained first (rare path), then the workspace belongs to the new one.
static int workspace(silent_ctx* ctx, hipStream_t s, size_t bytes) {
    if (ctx->ws_used && ctx->ws_stream != s) {
        // (not while `s` is capturing a HIP graph: a host synchronisation is illegal there, and a caller that captures has
        // ordered its warm-up stream against the capture stream itself -- pysilent_amd.recognition_testing does)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) (void)hipGetLastError();
        if (cap != hipStreamCaptureStatusActive) HIP_TRY(ctx, hipStreamSynchronize(ctx->ws_stream));
    }
    ctx->ws_stream = s;
    ctx->ws_used = true;
    // whoever lays the workspace out anew invalidates what silent_sparse_tail_stats would read back (silent_rgb_keypoints_dev
    // sets the flag again AFTER its own layout)
    ctx->sparse_ran = false;
    return grow(ctx, ctx->ws, bytes);
}

SILENT_EXPORT int silent_abi_version(void) { return SILENT_ABI_VERSION; }

SILENT_EXPORT int silent_device_count(int* count) try {
    if (!count) return fail(nullptr, SILENT_E_INVALID, "silent_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
        return fail(nullptr, SILENT_E_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return SILENT_OK;
} catch (...) {
    return on_exception(nullptr, "silent_device_count");
}

SILENT_EXPORT int silent_create(int device, silent_ctx** out) try {
    if (!out) return fail(nullptr, SILENT_E_INVALID, "silent_create: out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&n));
    if (device < 0 || device >= n)
        return fail(nullptr, SILENT_E_INVALID,
                    "silent_create: device " + std::to_string(device) + " out of range (" + std::to_string(n) +
                        " visible)");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(nullptr, SILENT_E_HIP, "silent_create: hipSetDevice failed");
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    silent_ctx* ctx = new (std::nothrow) silent_ctx();
    if (!ctx) return fail(nullptr, SILENT_E_NOMEM, "silent_create: out of host memory");
    ctx->device = device;
    ctx->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        std::string m = "silent_create: device is " + ctx->name + "; this library holds gfx950 code objects only";
        delete ctx;
        return fail(nullptr, SILENT_E_UNSUPPORTED, m);
    }
    const char* names[SILENT_TUNE_COUNT] = {"SILENT_GRAY_OPTS", "SILENT_RGB_OPTS", "SILENT_PYRAMID_OPTS"};
    for (int i = 0; i < SILENT_TUNE_COUNT; ++i)
        if (const char* e = std::getenv(names[i])) ctx->tune[i] = (unsigned)std::strtoul(e, nullptr, 0);
    *out = ctx;
    return SILENT_OK;
} catch (...) {
    return on_exception(nullptr, "silent_create");
}

SILENT_EXPORT int silent_set_tuning(silent_ctx* ctx, int which, unsigned value) try {
    if (!ctx) return fail(nullptr, SILENT_E_INVALID, "silent_set_tuning: ctx is NULL");
    if (which < 0 || which >= SILENT_TUNE_COUNT) return fail(ctx, SILENT_E_INVALID, "silent_set_tuning: unknown knob");
    ctx->tune[which] = value;
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_set_tuning");
}

SILENT_EXPORT int silent_get_tuning(const silent_ctx* ctx, int which, unsigned* value) try {
    if (!ctx || !value || which < 0 || which >= SILENT_TUNE_COUNT) return SILENT_E_INVALID;
    *value = ctx->tune[which];
    return SILENT_OK;
} catch (...) {
    return on_exception(nullptr, "silent_get_tuning");
}

SILENT_EXPORT void silent_destroy(silent_ctx* ctx) try {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    if (ctx->arena.p) (void)hipFree(ctx->arena.p);
    if (ctx->ws.p) (void)hipFree(ctx->ws.p);
    for (auto& pr : ctx->prof_ev)
        for (hipEvent_t e : pr)
            if (e) (void)hipEventDestroy(e);
    delete ctx;
} catch (...) {
}

SILENT_EXPORT const char* silent_last_error(const silent_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_create_err.c_str();
}

SILENT_EXPORT int silent_device_name(const silent_ctx* ctx, char* buf, size_t len) try {
    if (!ctx || !buf || len == 0) return SILENT_E_INVALID;
    std::snprintf(buf, len, "%s", ctx->name.c_str());
    return SILENT_OK;
} catch (...) {
    return on_exception(nullptr, "silent_device_name");
}

#define NEED_CTX(ctx)                  \
    SILENT_FAULT_POINT();              \
    if (!(ctx)) return fail(nullptr, SILENT_E_INVALID, std::string(__func__) + ": ctx is NULL"); \
    DeviceGuard device_guard_((ctx)->device);                                                    \
    if (!device_guard_.ok) return fail(ctx, SILENT_E_HIP, std::string(__func__) + ": hipSetDevice failed")

SILENT_EXPORT int silent_malloc(silent_ctx* ctx, size_t bytes, void** dptr) try {
    NEED_CTX(ctx);
    if (!dptr) return fail(ctx, SILENT_E_INVALID, "silent_malloc: dptr is NULL");
    *dptr = nullptr;
    HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_malloc");
}

SILENT_EXPORT int silent_free(silent_ctx* ctx, void* dptr) try {
    NEED_CTX(ctx);
    if (dptr) HIP_TRY(ctx, hipFree(dptr));
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_free");
}

SILENT_EXPORT int silent_memcpy_h2d(silent_ctx* ctx, void* dst, const void* src, size_t bytes, silent_stream stream) try {
    NEED_CTX(ctx);
    if (bytes && (!dst || !src)) return fail(ctx, SILENT_E_INVALID, "silent_memcpy_h2d: NULL pointer");
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_memcpy_h2d");
}

SILENT_EXPORT int silent_memcpy_d2h(silent_ctx* ctx, void* dst, const void* src, size_t bytes, silent_stream stream) try {
    NEED_CTX(ctx);
    if (bytes && (!dst || !src)) return fail(ctx, SILENT_E_INVALID, "silent_memcpy_d2h: NULL pointer");
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_memcpy_d2h");
}

SILENT_EXPORT int silent_gather_d2h(silent_ctx* ctx, void* dst_host, const void* const* src_dev, const size_t* bytes,
                                    int n, silent_stream stream) try {
    NEED_CTX(ctx);
    if (n < 0 || (n && (!dst_host || !src_dev || !bytes))) return fail(ctx, SILENT_E_INVALID, "silent_gather_d2h: NULL pointer");
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] && !src_dev[i]) return fail(ctx, SILENT_E_INVALID, "silent_gather_d2h: NULL source");
        total += bytes[i];
    }
    hipStream_t s = (hipStream_t)stream;
    if (n == 1) {
        HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_dev[0], bytes[0], hipMemcpyDeviceToHost, s));
    } else if (n > 1 && total) {
        // device-to-device into the context's staging arena (asynchronous, no host round trip each), then ONE copy to the
        // host: n separate copies into pageable memory cost a staging synchronisation each
        TRY(grow(ctx, ctx->arena, total));
        char* d = (char*)ctx->arena.p;
        for (int i = 0; i < n; ++i) {
            if (bytes[i]) HIP_TRY(ctx, hipMemcpyAsync(d, src_dev[i], bytes[i], hipMemcpyDeviceToDevice, s));
            d += bytes[i];
        }
        HIP_TRY(ctx, hipMemcpyAsync(dst_host, ctx->arena.p, total, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_gather_d2h");
}

SILENT_EXPORT int silent_synchronize(silent_ctx* ctx, silent_stream stream) try {
    NEED_CTX(ctx);
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_synchronize");
}

// ------------------------------------------------------------------------------------------ tile tables

// tile_h == 0 selects the 1-D "chunk" decomposition (kChunk flattened pixels per block; tile_w > 0: that many).
static int build_level_tab(silent_ctx* ctx, const char* who, const silent_extent* levels, int n_levels, int n_frames,
                           int tile_w, int tile_h, LevelTab* tab, long long* n_blocks,
                           const bool* skip = nullptr) {
    if (!levels) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": levels is NULL");
    if (n_levels < 1 || n_levels > kMaxLevels)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": n_levels must be in [1, " + std::to_string(kMaxLevels) + "]");
    if (n_frames < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": n_frames must be >= 1");
    std::memset(tab, 0, sizeof(*tab));
    tab->n_levels = n_levels;
    long long px = 0, tiles = 0;
    for (int l = 0; l < n_levels; ++l) {
        const long long h = levels[l].h, w = levels[l].w;
        if (h < 1 || w < 1 || h * w > (1ll << 30))
            return fail(ctx, SILENT_E_INVALID, std::string(who) + ": level " + std::to_string(l) + " extent " +
                                                   std::to_string(h) + "x" + std::to_string(w) + " is invalid");
        tab->h[l] = (int)h;
        tab->w[l] = (int)w;
        tab->px_off[l] = px;
        tab->tile_start[l] = (int)tiles;
        long long tx, ty;
        if (tile_h == 0) {
            const long long chunk = tile_w > 0 ? tile_w : kChunk;
            tx = (h * w + chunk - 1) / chunk;
            ty = 1;
        } else {
            tx = (w + tile_w - 1) / tile_w;
            ty = (h + tile_h - 1) / tile_h;
        }
        tab->tiles_x[l] = (int)tx;
        if (!(skip && skip[l])) tiles += tx * ty;  // a skipped level keeps its place in the layout, gets no tiles
        px += h * w;
    }
    tab->tile_start[n_levels] = (int)tiles;
    tab->tiles_per_frame = (int)tiles;
    tab->frame_px = px;
    const long long total = tiles * (long long)n_frames;
    if (total > 0x7fffffffll) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": too many tiles for one launch");
    *n_blocks = total;
    return SILENT_OK;
}

static int check_launch(silent_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SILENT_E_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
    return SILENT_OK;
}

static long long pyramid_px(const silent_extent* levels, int n_levels) {
    long long px = 0;
    for (int l = 0; l < n_levels; ++l) px += (long long)levels[l].h * levels[l].w;
    return px;
}

// ------------------------------------------------------------------------------------------ convolution

template <int KH, int KW, int CIN, int COUT, bool REG>
static void launch_conv(const float* in, float* out, const LevelTab& tab, const ConvW& w, const Epilogue& ep,
                        long long blocks, hipStream_t s) {
    hipLaunchKernelGGL((conv2d_same_kernel<KH, KW, CIN, COUT, REG>), dim3((unsigned)blocks), dim3(256), 0, s, in, out,
                       tab, w, ep);
}

template <int KH, int KW, int CI, int CO>
static void conv_case(bool reg, const float* in, float* out, const LevelTab& tab, const ConvW& w, const Epilogue& ep,
                      long long blocks, hipStream_t s) {
    if constexpr (CI == CO) {
        if (reg) {
            launch_conv<KH, KW, CI, CO, true>(in, out, tab, w, ep, blocks, s);
            return;
        }
    }
    launch_conv<KH, KW, CI, CO, false>(in, out, tab, w, ep, blocks, s);
}

static int conv_dispatch(silent_ctx* ctx, const char* who, const float* in, const silent_extent* levels, int n_levels,
                         int n_frames, int cin, const float* k, int kh, int kw, int cout, bool reg,
                         const Epilogue& ep, float* out, hipStream_t s) {
    if (!in || !out || !k) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (kh < 1 || kw < 1 || cin < 1 || cout < 1 || kh > 15 || kw > 15 || cin > 16 || cout > 16)
        return fail(ctx, SILENT_E_UNSUPPORTED, std::string(who) + ": kernel shape out of range (k <= 15, channels <= 16)");
    if ((long long)kh * kw * cin * cout > SILENT_MAX_KERNEL_FLOATS)
        return fail(ctx, SILENT_E_UNSUPPORTED,
                    std::string(who) + ": kh*kw*C_in*C_out exceeds " + std::to_string(SILENT_MAX_KERNEL_FLOATS));
    if (reg && cin != cout) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": blur must be [kh,kw,C,C]");
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kConvTW, kConvTH, &tab, &blocks));
    ConvW w;
    std::memset(&w, 0, sizeof(w));
    std::memcpy(w.w, k, sizeof(float) * kh * kw * cin * cout);

#define CONV_CASE(KH_, KW_, CI_, CO_)                                      \
    if (kh == KH_ && kw == KW_ && cin == CI_ && cout == CO_) {             \
        conv_case<KH_, KW_, CI_, CO_>(reg, in, out, tab, w, ep, blocks, s); \
        return check_launch(ctx, who);                                     \
    }
    CONV_CASE(3, 3, 1, 1)
    CONV_CASE(3, 3, 1, 3)
    CONV_CASE(3, 3, 1, 4)
    CONV_CASE(3, 3, 1, 8)
    CONV_CASE(3, 3, 3, 1)
    CONV_CASE(3, 3, 3, 3)
    CONV_CASE(3, 3, 3, 4)
    CONV_CASE(7, 7, 1, 1)
    CONV_CASE(7, 7, 3, 3)
#undef CONV_CASE
    const int IW = kConvTW + kw - 1, IH = kConvTH + kh - 1;
    const size_t lds = sizeof(float) * (size_t)IW * IH * cin;
    if (lds > 64 * 1024) return fail(ctx, SILENT_E_UNSUPPORTED, std::string(who) + ": halo tile exceeds 64 KiB of LDS");
    hipLaunchKernelGGL(conv2d_same_generic_kernel, dim3((unsigned)blocks), dim3(256), lds, s, in, out, tab, w, kh, kw,
                       cin, cout, reg ? 1 : 0, ep);
    return check_launch(ctx, who);
}

SILENT_EXPORT int silent_conv2d_same_dev(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                         int n_frames, int c_in, const float* kernel_hwio, int kh, int kw, int c_out,
                                         unsigned flags, float clip_hi, float* out, silent_stream stream) try {
    NEED_CTX(ctx);
    Epilogue ep{flags, clip_hi, 0.f, 0.f, 0};
    return conv_dispatch(ctx, "silent_conv2d_same", in, levels, n_levels, n_frames, c_in, kernel_hwio, kh, kw, c_out,
                         false, ep, out, (hipStream_t)stream);
} catch (...) {
    return on_exception(ctx, "silent_conv2d_same_dev");
}

SILENT_EXPORT int silent_regulate_dev(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                      int n_frames, int channels, const float* blur_hwio, int kh, int kw,
                                      float regulation_value, float regulation_root, int flat_policy, float* out,
                                      silent_stream stream) try {
    NEED_CTX(ctx);
    if (flat_policy != SILENT_FLAT_IEEE && flat_policy != SILENT_FLAT_ZERO)
        return fail(ctx, SILENT_E_INVALID, "silent_regulate: flat_policy must be SILENT_FLAT_IEEE or SILENT_FLAT_ZERO");
    // 7x7 channel-uniform blur on 3-channel maps (the reference's orientation_filter): 49-tap filter of the channel sum
    if (channels == 3 && kh == 7 && kw == 7 && in && out && blur_hwio && levels) {
        bool uniform = true;
        for (int t = 0; t < 49 && uniform; ++t)
            for (int io = 1; io < 9; ++io)
                if (blur_hwio[t * 9 + io] != blur_hwio[t * 9]) uniform = false;
        const unsigned kopts = ctx->tune[SILENT_TUNE_RGB];
        if (uniform && !(kopts & 1u)) {
            RegArgs a;
            long long blocks;
            TRY(build_level_tab(ctx, "silent_regulate", levels, n_levels, n_frames, kRegTW, kRegTH, &a.tab, &blocks));
            a.in = in;
            a.out = out;
            for (int t = 0; t < 49; ++t) a.blur[t] = blur_hwio[t * 9];
            a.rv = regulation_value;
            a.root = regulation_root;
            a.flat_policy = flat_policy;
            hipLaunchKernelGGL(regulate_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
            return check_launch(ctx, "silent_regulate");
        }
    }
    Epilogue ep{0u, 0.f, regulation_value, regulation_root, flat_policy};
    return conv_dispatch(ctx, "silent_regulate", in, levels, n_levels, n_frames, channels, blur_hwio, kh, kw, channels,
                         true, ep, out, (hipStream_t)stream);
} catch (...) {
    return on_exception(ctx, "silent_regulate_dev");
}

// ------------------------------------------------------------------------------------------ fused gray pass

static int launch_gray(silent_ctx* ctx, const char* who, const float* pyr, const silent_extent* levels, int n_levels,
                       int n_frames, const float* cs_kernel, const float* end_bank, int n_orient, float clip_hi,
                       float* cs_out, float* end_out, hipStream_t s, const bool* skip) {
    if (!pyr || !cs_kernel) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (!cs_out && !end_out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": both outputs are NULL");
    if (end_out && !end_bank) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": end_bank is NULL");
    if (n_orient != 3 && n_orient != 4 && n_orient != 8)
        return fail(ctx, SILENT_E_UNSUPPORTED, std::string(who) + ": n_orient must be 3, 4 or 8");
    // development knob for interleaved A/B timing (scripts/ab_gray.py): bit0 XCD-aware tile order (measured
    // 7 % slower, off), bit1 32-row tiles (3 % slower, off), bit2 non-temporal stores (no effect, off)
    const unsigned opts = ctx->tune[SILENT_TUNE_GRAY];
    const int th = (opts & 2u) ? 32 : kGrayTH;
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kGrayTW, th, &tab, &blocks, skip));
    if (blocks == 0) return SILENT_OK;
    GrayW w;
    std::memset(&w, 0, sizeof(w));
    std::memcpy(w.cs, cs_kernel, sizeof(float) * 9);
    if (end_bank) std::memcpy(w.end, end_bank, sizeof(float) * 9 * n_orient);
#define GRAY_LAUNCH(K_, R_) \
    hipLaunchKernelGGL((gray_line_end_kernel<K_, R_>), dim3((unsigned)blocks), dim3(256), 0, s, pyr, cs_out, end_out, tab, w, clip_hi, opts)
    if (th == 32) {
        if (n_orient == 3) GRAY_LAUNCH(3, 32);
        else if (n_orient == 4) GRAY_LAUNCH(4, 32);
        else GRAY_LAUNCH(8, 32);
    } else {
        if (n_orient == 3) GRAY_LAUNCH(3, kGrayTH);
        else if (n_orient == 4) GRAY_LAUNCH(4, kGrayTH);
        else GRAY_LAUNCH(8, kGrayTH);
    }
#undef GRAY_LAUNCH
    return check_launch(ctx, who);
}

SILENT_EXPORT int silent_gray_line_end_dev(silent_ctx* ctx, const float* pyr, const silent_extent* levels,
                                           int n_levels, int n_frames, const float* cs_kernel, const float* end_bank,
                                           int n_orient, float clip_hi, float* cs_out, float* end_out,
                                           silent_stream stream) try {
    NEED_CTX(ctx);
    return launch_gray(ctx, "silent_gray_line_end", pyr, levels, n_levels, n_frames, cs_kernel, end_bank, n_orient,
                       clip_hi, cs_out, end_out, (hipStream_t)stream, nullptr);
} catch (...) {
    return on_exception(ctx, "silent_gray_line_end_dev");
}

// ------------------------------------------------------------------------------------------ pointwise / nms

SILENT_EXPORT int silent_pad_inwards_dev(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                         int n_frames, int channels, int pt, int pb, int pl, int pr, float* out,
                                         silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_pad_inwards";
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (channels < 1 || pt < 0 || pb < 0 || pl < 0 || pr < 0)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": channels must be >= 1 and paddings >= 0");
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    hipLaunchKernelGGL(pad_inwards_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, tab,
                       channels, pt, pb, pl, pr);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_pad_inwards_dev");
}

SILENT_EXPORT int silent_value_from_color_dev(silent_ctx* ctx, const float* in, const silent_extent* levels,
                                              int n_levels, int n_frames, int channels, float* out,
                                              silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_value_from_color";
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": channels must be >= 1");
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    const long long npx = tab.frame_px * n_frames;
    const long long grid = std::min<long long>((npx + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(value_from_color_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, in, out, npx,
                       channels);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_value_from_color_dev");
}

SILENT_EXPORT int silent_bw_from_color_dev(silent_ctx* ctx, const float* in, const silent_extent* levels,
                                              int n_levels, int n_frames, int channels, float* out,
                                              silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_bw_from_color";
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": channels must be >= 1");
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    const long long npx = tab.frame_px * n_frames;
    const long long grid = std::min<long long>((npx + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(bw_from_color_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, in, out, npx,
                       channels);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_bw_from_color_dev");
}

SILENT_EXPORT int silent_nms3x3_dev(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                    int n_frames, int channels, int mode, float* out, silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_nms3x3";
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": channels must be >= 1");
    if (mode != SILENT_NMS_PRODUCT && mode != SILENT_NMS_FIRED)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": mode must be SILENT_NMS_PRODUCT or SILENT_NMS_FIRED");
    LevelTab tab;
    long long blocks;
    if (channels == 1 || channels == 3) {  // streaming stencil
        TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kNmsTW, kNmsTH, &tab, &blocks));
        if (channels == 3)
            hipLaunchKernelGGL(nms3x3_stream_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, tab, mode);
        else
            hipLaunchKernelGGL(nms3x3_stream_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, tab, mode);
        return check_launch(ctx, who);
    }
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    hipLaunchKernelGGL(nms3x3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, tab, channels,
                       mode);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_nms3x3_dev");
}

// ------------------------------------------------------------------------------------------ top-percent threshold

SILENT_EXPORT int silent_top_value_points_dev(silent_ctx* ctx, const float* color, const float* value,
                                              const silent_extent* levels, int n_levels, int n_frames, int channels,
                                              double top_percent, float* out, silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_top_value_points";
    if (!color || !out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": channels must be >= 1");
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    LevelTab rtab;
    long long rblocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kRedChunk, 0, &rtab, &rblocks));
    const int nmm = n_frames * n_levels;
    TRY(workspace(ctx, (hipStream_t)stream, sizeof(unsigned) * 2 * (size_t)nmm));
    unsigned* mm = (unsigned*)ctx->ws.p;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(init_maxmin_kernel, dim3((nmm + 255) / 256), dim3(256), 0, s, mm, nmm);
    hipLaunchKernelGGL(level_maxmin_kernel, dim3((unsigned)rblocks), dim3(256), 0, s, value, value ? nullptr : color,
                       channels, rtab, mm);
    // python: (1.0 - top_percent) and top_percent are doubles that TF casts to float32 constants
    const float a = (float)(1.0 - top_percent), b = (float)top_percent;
    hipLaunchKernelGGL(top_value_points_kernel, dim3((unsigned)blocks), dim3(256), 0, s, color, value, out, tab,
                       channels, a, b, mm);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_top_value_points_dev");
}

SILENT_EXPORT int silent_select_peaks_dev(silent_ctx* ctx, const float* color, const float* value,
                                          const silent_extent* levels, int n_levels, int n_frames, int channels,
                                          double top_percent, float* top_out, float* peaks_out, float* peak_value_out,
                                          silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_select_peaks";
    if (!color) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (!top_out && !peaks_out && !peak_value_out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": all outputs are NULL");
    if (channels != 1 && channels != 3) return fail(ctx, SILENT_E_UNSUPPORTED, std::string(who) + ": channels must be 1 or 3");
    LevelTab rtab, tab;
    long long rblocks, blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kRedChunk, 0, &rtab, &rblocks));
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kSelTW, kSelTH, &tab, &blocks));
    const int nmm = n_frames * n_levels;
    TRY(workspace(ctx, (hipStream_t)stream, sizeof(unsigned) * 2 * (size_t)nmm));
    unsigned* mm = (unsigned*)ctx->ws.p;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(init_maxmin_kernel, dim3((nmm + 255) / 256), dim3(256), 0, s, mm, nmm);
    hipLaunchKernelGGL(level_maxmin_kernel, dim3((unsigned)rblocks), dim3(256), 0, s, value, value ? nullptr : color,
                       channels, rtab, mm);
    const float a = (float)(1.0 - top_percent), b = (float)top_percent;
    RegionTab no_regions;
    std::memset(&no_regions, 0, sizeof(no_regions));
    if (channels == 3)
        hipLaunchKernelGGL((select_peaks_kernel<3, false>), dim3((unsigned)blocks), dim3(256), 0, s, color, value, top_out,
                           peaks_out, peak_value_out, tab, a, b, mm, no_regions, nullptr, nullptr);
    else
        hipLaunchKernelGGL((select_peaks_kernel<1, false>), dim3((unsigned)blocks), dim3(256), 0, s, color, value, top_out,
                           peaks_out, peak_value_out, tab, a, b, mm, no_regions, nullptr, nullptr);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_select_peaks_dev");
}

// ------------------------------------------------------------------------------------------ keypoint indices

// TF1 max_pool SAME geometry with window == full extent
static int region_axis(int size, int stride, int* n_win, int* nseg, int* cut, int* w_lo, int* w_hi, float* scale) {
    if (stride < 1) return -1;
    const int out = (size + stride - 1) / stride;
    *n_win = out;
    *scale = (float)out / (float)size;
    if (out > kMaxWin) return -2;
    const int pad_before = ((out - 1) * stride) / 2;
    int lo[kMaxWin], hi[kMaxWin];
    std::vector<int> cuts = {0, size};
    for (int j = 0; j < out; ++j) {
        const int a = j * stride - pad_before, b = a + size;
        lo[j] = a < 0 ? 0 : a;
        hi[j] = b > size ? size : b;
        cuts.push_back(lo[j]);
        cuts.push_back(hi[j]);
    }
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    const int ns = (int)cuts.size() - 1;
    if (ns > kMaxSeg) return -2;
    for (int i = 0; i <= ns; ++i) cut[i] = cuts[i];
    for (int j = 0; j < out; ++j) {
        int a = 0, b = 0;
        for (int i = 0; i <= ns; ++i) {
            if (cuts[i] == lo[j]) a = i;
            if (cuts[i] == hi[j]) b = i;
        }
        w_lo[j] = a;
        w_hi[j] = b;
    }
    *n_win = out;
    *nseg = ns;
    *scale = (float)out / (float)size;
    return 0;
}

// Region tables of max_value_indices_region (TF1 max_pool geometry per level).  *general = true when some level has more
// than kMaxWin windows per axis: the cell tables (kernarg-resident) do not apply then and the separable prefix / suffix
// path runs (region_rowmax_kernel / region_colmax_kernel) -- any region_shape the reference accepts
// (slam_recognition/util/selection/top_value_points.py:32-45).
static int build_region_tab(silent_ctx* ctx, const char* who, const silent_extent* levels, int n_levels,
                            const silent_extent* regions, RegionTab* rt, bool* general) {
    std::memset(rt, 0, sizeof(*rt));
    *general = false;
    long long m1 = 0, pooled = 0;
    for (int l = 0; l < n_levels; ++l) {
        RegionLevel& r = rt->lv[l];
        if (regions[l].h < 1 || regions[l].w < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": region extents must be >= 1");
        const int rc0 = region_axis(levels[l].h, regions[l].h, &r.oh, &r.nrs, r.rcut, r.wy_lo, r.wy_hi, &r.yscale);
        const int rc1 = region_axis(levels[l].w, regions[l].w, &r.ow, &r.ncs, r.ccut, r.wx_lo, r.wx_hi, &r.xscale);
        if (rc0 == -2 || rc1 == -2) *general = true;
        r.ry = regions[l].h;
        r.rx = regions[l].w;
        r.pad_y = ((r.oh - 1) * r.ry) / 2;
        r.pad_x = ((r.ow - 1) * r.rx) / 2;
        r.m1_off = m1;
        r.pooled_off = pooled;
        m1 += (long long)levels[l].h * r.ow;
        pooled += (long long)r.oh * r.ow;
    }
    rt->m1_per_frame = m1;
    rt->pooled_per_frame = pooled;
    return SILENT_OK;
}

// workspace of the keypoint passes, after `reserve` bytes the caller keeps for itself:
// cells | chunk_counts | hit_masks | cand_n | nan flags | offsets | m1 | pooled | summary | candidates | modes | peak-value map
struct KeypointWs {
    unsigned* cells;
    int* chunk_counts;
    long long* chunk_offsets;
    size_t n_cells;
    float* m1;       // general path: row maxima over the column windows
    float* pooled;   // general path: window maxima
    unsigned long long* hit_masks;   // 1 bit per pixel: the count pass's ballots, read by the write pass
    // sparse tail (silent_rgb_keypoints without a peak-value map; silent_peaks.h, sparse_select_kernel)
    float* sum = nullptr;            // the chain kernel's value summary (SumTab geometry)
    Candidate* cand = nullptr;       // [n_frames][kCandCap]
    int* cand_n = nullptr;           // [n_frames]
    int* dense_flags = nullptr;      // [n_frames][n_levels]: kTailSparse / kTailDense / kTailZero
    int* nan_flags = nullptr;        // [n_frames][n_levels]: the chain kernel saw a NaN value in that level
    float* pv = nullptr;             // peak-value map of the (frame, level)s that run the dense kernels
    void* zero_from = nullptr;       // chunk_counts | hit_masks | cand_n | nan flags: one memset before a sparse tail's chain launch
    size_t zero_bytes = 0;
};

static int keypoint_workspace(silent_ctx* ctx, hipStream_t stream, int n_levels, int n_frames, long long blocks, size_t reserve,
                              const RegionTab& rt, bool general, KeypointWs* w, long long sum_entries = 0, long long pv_px = 0) {
    w->n_cells = (size_t)n_frames * n_levels * kCells;
    const size_t off_cells = align_up(reserve);
    const size_t off_counts = off_cells + align_up(w->n_cells * sizeof(unsigned));
    const size_t off_masks = off_counts + align_up((size_t)blocks * sizeof(int));
    const size_t off_candn = off_masks + align_up((size_t)blocks * 4 * kKpPer * sizeof(unsigned long long));
    const size_t off_nanf = off_candn + align_up((size_t)n_frames * sizeof(int));
    const size_t off_offsets = off_nanf + align_up((size_t)n_frames * n_levels * sizeof(int));
    const size_t off_m1 = off_offsets + align_up((size_t)blocks * sizeof(long long));
    const size_t off_pooled = off_m1 + (general ? align_up((size_t)n_frames * rt.m1_per_frame * sizeof(float)) : 0);
    const size_t off_sum = off_pooled + (general ? align_up((size_t)n_frames * rt.pooled_per_frame * sizeof(float)) : 0);
    const size_t off_cand = off_sum + align_up((size_t)sum_entries * n_frames * sizeof(float));
    const size_t off_flags = off_cand + (sum_entries ? align_up((size_t)n_frames * kCandCap * sizeof(Candidate)) : 0);
    const size_t off_pv = off_flags + align_up((size_t)n_frames * n_levels * sizeof(int));
    const size_t total = off_pv + align_up((size_t)pv_px * n_frames * sizeof(float));
    TRY(workspace(ctx, (hipStream_t)stream, total));
    char* base = (char*)ctx->ws.p;
    w->cells = (unsigned*)(base + off_cells);
    w->chunk_counts = (int*)(base + off_counts);
    w->chunk_offsets = (long long*)(base + off_offsets);
    w->m1 = (float*)(base + off_m1);
    w->pooled = (float*)(base + off_pooled);
    w->hit_masks = (unsigned long long*)(base + off_masks);
    w->cand_n = (int*)(base + off_candn);
    w->nan_flags = (int*)(base + off_nanf);
    w->sum = sum_entries ? (float*)(base + off_sum) : nullptr;
    w->cand = sum_entries ? (Candidate*)(base + off_cand) : nullptr;
    w->dense_flags = (int*)(base + off_flags);
    w->pv = pv_px ? (float*)(base + off_pv) : nullptr;
    w->zero_from = base + off_counts;
    w->zero_bytes = off_offsets - off_counts;
    return SILENT_OK;
}

// general path: window maxima of every level into w.pooled (two launches)
static int region_window_maxima(silent_ctx* ctx, const char* who, const float* value, const silent_extent* levels, int n_levels,
                                int n_frames, const RegionTab& rt, const KeypointWs& w, hipStream_t s) {
    LevelTab rowtab;
    long long rows;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 1 << 30, 1, &rowtab, &rows));   // one tile per row
    long long cols = 0;
    for (int l = 0; l < n_levels; ++l) cols += rt.lv[l].ow;
    hipLaunchKernelGGL(region_rowmax_kernel, dim3((unsigned)rows), dim3(256), 0, s, value, rowtab, rt, w.m1);
    const long long threads = cols * n_frames;
    hipLaunchKernelGGL(region_colmax_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, w.m1, rowtab, rt, n_frames,
                       cols, w.pooled);
    return check_launch(ctx, who);
}

// count -> scan -> ordered write, given the cell maxima
static void keypoint_passes(const float* value, const LevelTab& tab, long long blocks, const RegionTab& rt, const KeypointWs& w,
                            bool general, int n_frames, int64_t* idx, size_t cap_per_frame, int64_t* counts, hipStream_t s,
                            const int* dense_flags = nullptr, float* caller_map = nullptr) {
    if (general)
        hipLaunchKernelGGL(region_count_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, value, tab, rt, w.cells, w.pooled, w.chunk_counts, w.hit_masks, dense_flags);
    else
        hipLaunchKernelGGL(region_count_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, value, tab, rt, w.cells, w.pooled, w.chunk_counts, w.hit_masks, dense_flags);
    if (dense_flags)   // sparse tail: the candidates' hits join what the count pass left
        hipLaunchKernelGGL(sparse_finish_kernel, dim3((unsigned)n_frames), dim3(256), 0, s, tab, rt, w.cells, w.cand, w.cand_n, dense_flags,
                           w.hit_masks, w.chunk_counts, caller_map);
    hipLaunchKernelGGL(region_scan_kernel, dim3((unsigned)n_frames), dim3(256), 0, s, w.chunk_counts, w.chunk_offsets,
                       tab.tiles_per_frame, counts);
    if (!cap_per_frame) return;
    hipLaunchKernelGGL(region_write_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tab, w.hit_masks, w.chunk_offsets, idx,
                       (long long)cap_per_frame, w.chunk_counts);
}

SILENT_EXPORT int silent_max_value_indices_region_dev(silent_ctx* ctx, const float* value, const silent_extent* levels,
                                                      int n_levels, int n_frames, const silent_extent* regions,
                                                      int64_t* idx, size_t cap_per_frame, int64_t* counts,
                                                      silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_max_value_indices_region";
    if (!value || !regions || !counts || (!idx && cap_per_frame))
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    LevelTab tab, ctab;
    long long blocks, cblocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kKpChunk, 0, &tab, &blocks));   // count / write chunks
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kRedChunk, 0, &ctab, &cblocks));  // cell maxima
    RegionTab rt;
    bool general;
    TRY(build_region_tab(ctx, who, levels, n_levels, regions, &rt, &general));
    KeypointWs w;
    TRY(keypoint_workspace(ctx, (hipStream_t)stream, n_levels, n_frames, blocks, 0, rt, general, &w));
    hipStream_t s = (hipStream_t)stream;
    if (general) {
        TRY(region_window_maxima(ctx, who, value, levels, n_levels, n_frames, rt, w, s));
    } else {
        hipLaunchKernelGGL(init_cells_kernel, dim3((unsigned)((w.n_cells + 255) / 256)), dim3(256), 0, s, w.cells, (long long)w.n_cells);
        hipLaunchKernelGGL(region_cell_max_kernel, dim3((unsigned)cblocks), dim3(256), 0, s, value, ctab, rt, w.cells);
    }
    keypoint_passes(value, tab, blocks, rt, w, general, n_frames, idx, cap_per_frame, counts, s);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_max_value_indices_region_dev");
}

// SURVEY 8d config 3 as one call: top-percent -> NMS -> value -> per-region keypoint indices of the peak value.
// = silent_select_peaks + silent_max_value_indices_region, with the cell maxima folded into the selection pass.
// Two halves so that silent_rgb_keypoints can run the chain kernel in between (it fills the per-level extrema itself).
struct SelectPlan {
    LevelTab rtab, stab, tab;
    long long rblocks, sblocks, blocks;
    RegionTab rt;
    bool general;
    KeypointWs w;
    unsigned* mm;   // [n_frames][n_levels][2] ordered-uint extrema, at the head of the context workspace
    int nmm;
    SumTab st;      // sparse tail: geometry of the chain kernel's value summary (frame_entries = 0: none)
};

// Geometry of the value summary a chain launch with tile height th leaves (silent_rgb2.h): per level
// [tiles_y * gpt][ceil(w / 2)] entries, gpt = ceil(th / kSumRows).
static void build_sum_tab(const silent_extent* levels, int n_levels, int th, SumTab* st) {
    std::memset(st, 0, sizeof(*st));
    st->th = th;
    st->gpt = (th + kSumRows - 1) / kSumRows;
    long long e = 0;
    for (int l = 0; l < n_levels; ++l) {
        st->off[l] = e;
        e += (long long)((levels[l].h + th - 1) / th) * st->gpt * ((levels[l].w + 1) / 2);
    }
    for (int l = n_levels; l <= kMaxLevels; ++l) st->off[l] = e;
    st->frame_entries = e;
}

// tables, workspace, and the init kernels.  sparse_th > 0: also lay out the sparse tail for a chain launch with that tile
// height; pv_ws: keep room for a peak-value map in the workspace (the caller has none)
static int select_prepare(silent_ctx* ctx, const char* who, const silent_extent* levels, int n_levels, int n_frames,
                          const silent_extent* regions, hipStream_t s, SelectPlan* sp, int sparse_th = 0, bool pv_ws = false) {
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kRedChunk, 0, &sp->rtab, &sp->rblocks));
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kSelTW, kSelTH, &sp->stab, &sp->sblocks));
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, kKpChunk, 0, &sp->tab, &sp->blocks));
    TRY(build_region_tab(ctx, who, levels, n_levels, regions, &sp->rt, &sp->general));
    sp->nmm = n_frames * n_levels;
    std::memset(&sp->st, 0, sizeof(sp->st));
    if (sparse_th > 0 && !sp->general && n_frames <= 65535) {
        build_sum_tab(levels, n_levels, sparse_th, &sp->st);
        if (sp->st.frame_entries >= (1ll << 31)) std::memset(&sp->st, 0, sizeof(sp->st));   // (sparse_select_kernel indexes a frame with int)
    }
    TRY(keypoint_workspace(ctx, s, n_levels, n_frames, sp->blocks, sizeof(unsigned) * 2 * (size_t)sp->nmm, sp->rt, sp->general, &sp->w,
                           sp->st.frame_entries, pv_ws ? sp->tab.frame_px : 0));
    sp->mm = (unsigned*)ctx->ws.p;
    const long long n_init = std::max<long long>(2ll * sp->nmm, (long long)sp->w.n_cells);
    const long long n_zero16 = sp->st.frame_entries > 0 ? (long long)(sp->w.zero_bytes / 16) : 0;   // (every piece of the workspace is align_up'ed)
    const long long init_blocks = std::max((n_init + 255) / 256, std::min<long long>((n_zero16 + 255) / 256, 8ll * ctx->n_cus));
    hipLaunchKernelGGL(init_select_kernel, dim3((unsigned)init_blocks), dim3(256), 0, s, sp->mm, 2 * sp->nmm, sp->w.cells,
                       (long long)sp->w.n_cells, (uint4*)sp->w.zero_from, n_zero16);
    return SILENT_OK;
}

// have_mm: the extrema are already in sp.mm (no reduction pass).  sparse: the chain kernel left its value summary in sp.w.sum
// (geometry sp.st) -- the sparse tail runs and the dense kernels only where it could not settle a (frame, level).
static int select_run(silent_ctx* ctx, const char* who, const float* color, const float* value, const silent_extent* levels,
                      int n_levels, int n_frames, int channels, double top_percent, const SelectPlan& sp, bool have_mm,
                      float* peak_value_out, int64_t* idx, size_t cap_per_frame, int64_t* counts, hipStream_t s,
                      bool sparse = false) {
    unsigned* mm = sp.mm;
    const RegionTab& rt = sp.rt;
    const KeypointWs& w = sp.w;
    if (!have_mm)
        hipLaunchKernelGGL(level_maxmin_kernel, dim3((unsigned)sp.rblocks), dim3(256), 0, s, value, value ? nullptr : color,
                           channels, sp.rtab, mm);
    const float a = (float)(1.0 - top_percent), b = (float)top_percent;
    const int* dense_flags = nullptr;
    float* const caller_map = peak_value_out;
    if (sparse) {
        // (sparse implies: 3 channels, cell tables, extrema present, no caller-side peak-value map; select_prepare zeroed the counters)
        hipLaunchKernelGGL(sparse_select_kernel, dim3((unsigned)((sp.st.frame_entries + 255) / 256), (unsigned)n_frames), dim3(256), 0, s, color, sp.tab, sp.st, w.sum,
                           n_frames, a, b, mm, rt, w.cells, w.cand, w.cand_n);
        hipLaunchKernelGGL(sparse_modes_kernel, dim3((unsigned)n_frames), dim3(64), 0, s, sp.tab, rt, w.cells, w.cand_n, w.nan_flags,
                           w.dense_flags, peak_value_out ? 1 : 0);
        if (peak_value_out)   // the map the caller takes: zeros wherever the dense pass will not write
            hipLaunchKernelGGL(sparse_fill_map_kernel, dim3((unsigned)sp.blocks), dim3(256), 0, s, sp.tab, w.dense_flags, peak_value_out);
        dense_flags = w.dense_flags;
    }
    if (!peak_value_out) peak_value_out = w.pv;
    if (sp.general) {
        // many windows: the selection pass without the folded cell maxima, then the separable window maxima
        if (channels == 3)
            hipLaunchKernelGGL((select_peaks_kernel<3, false>), dim3((unsigned)sp.sblocks), dim3(256), 0, s, color, value, nullptr, nullptr,
                               peak_value_out, sp.stab, a, b, mm, rt, nullptr, nullptr);
        else
            hipLaunchKernelGGL((select_peaks_kernel<1, false>), dim3((unsigned)sp.sblocks), dim3(256), 0, s, color, value, nullptr, nullptr,
                               peak_value_out, sp.stab, a, b, mm, rt, nullptr, nullptr);
        TRY(region_window_maxima(ctx, who, peak_value_out, levels, n_levels, n_frames, rt, w, s));
    } else if (channels == 3) {
        hipLaunchKernelGGL((select_peaks_kernel<3, true>), dim3((unsigned)sp.sblocks), dim3(256), 0, s, color, value, nullptr, nullptr,
                           peak_value_out, sp.stab, a, b, mm, rt, w.cells, dense_flags);
    } else {
        hipLaunchKernelGGL((select_peaks_kernel<1, true>), dim3((unsigned)sp.sblocks), dim3(256), 0, s, color, value, nullptr, nullptr,
                           peak_value_out, sp.stab, a, b, mm, rt, w.cells, dense_flags);
    }
    keypoint_passes(peak_value_out, sp.tab, sp.blocks, rt, w, sp.general, n_frames, idx, cap_per_frame, counts, s, dense_flags,
                    dense_flags ? caller_map : nullptr);
    return check_launch(ctx, who);
}

SILENT_EXPORT int silent_select_keypoints_dev(silent_ctx* ctx, const float* color, const float* value,
                                              const silent_extent* levels, int n_levels, int n_frames, int channels,
                                              double top_percent, const silent_extent* regions, float* peak_value_out,
                                              int64_t* idx, size_t cap_per_frame, int64_t* counts, silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_select_keypoints";
    if (!color || !regions || !counts || (!idx && cap_per_frame))
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (channels != 1 && channels != 3) return fail(ctx, SILENT_E_UNSUPPORTED, std::string(who) + ": channels must be 1 or 3");
    SelectPlan sp;
    TRY(select_prepare(ctx, who, levels, n_levels, n_frames, regions, (hipStream_t)stream, &sp, 0, !peak_value_out));
    return select_run(ctx, who, color, value, levels, n_levels, n_frames, channels, top_percent, sp, false, peak_value_out, idx,
                      cap_per_frame, counts, (hipStream_t)stream);
} catch (...) {
    return on_exception(ctx, "silent_select_keypoints_dev");
}

// ------------------------------------------------------------------------------------------ centroids

static int build_cell_tab(silent_ctx* ctx, const char* who, const silent_extent* levels, int n_levels, int rh, int rw,
                          CellTab* ct) {
    if (rh < 1 || rw < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": region extents must be >= 1");
    std::memset(ct, 0, sizeof(*ct));
    ct->rh = rh;
    ct->rw = rw;
    long long cells = 0;
    for (int l = 0; l < n_levels; ++l) {
        const int h = levels[l].h, w = levels[l].w;
        ct->oh[l] = (h + rh - 1) / rh;
        ct->ow[l] = (w + rw - 1) / rw;
        ct->y_first[l] = -(std::max((ct->oh[l] - 1) * rh + rh - h, 0) / 2);
        ct->x_first[l] = -(std::max((ct->ow[l] - 1) * rw + rw - w, 0) / 2);
        ct->yscale[l] = (float)ct->oh[l] / (float)h;
        ct->xscale[l] = (float)ct->ow[l] / (float)w;
        ct->cell_off[l] = cells;
        cells += (long long)ct->oh[l] * ct->ow[l];
    }
    ct->frame_cells = cells;
    return SILENT_OK;
}

SILENT_EXPORT int silent_centroids_dev(silent_ctx* ctx, const float* value, const silent_extent* levels, int n_levels,
                                       int n_frames, int region_h, int region_w, float* dist_out, float* total_out,
                                       silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_centroids";
    if (!value || !dist_out || !total_out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    CellTab ct;
    TRY(build_cell_tab(ctx, who, levels, n_levels, region_h, region_w, &ct));
    const long long cells = ct.frame_cells * n_frames;
    if ((cells + 255) / 256 > 0x7fffffffll) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": too many cells");
    TRY(workspace(ctx, (hipStream_t)stream, (size_t)cells * 2 * sizeof(float)));
    float* cxy = (float*)ctx->ws.p;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(centroid_cells_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, s, value, tab, ct, n_frames,
                       total_out, cxy);
    hipLaunchKernelGGL(centroid_dist_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tab, ct, cxy, dist_out);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_centroids_dev");
}

// ------------------------------------------------------------------------------------------ boosting state

static int check_boost_params(silent_ctx* ctx, const silent_boosting_params* p, BoostP* bp) {
    if (!p) return fail(ctx, SILENT_E_INVALID, "silent_boosting_step: NULL params");
    if (p->recovery_mode < 1 || p->recovery_mode > 3)
        return fail(ctx, SILENT_E_INVALID, "silent_boosting_step: You must choose a type of recovery");
    bp->lo = -p->exhaustion_max;
    bp->hi = p->excitation_max;
    bp->recovery_mode = (int)p->recovery_mode;
    bp->recovery_amount = p->recovery_amount;
    bp->recovery_percentage = p->recovery_percentage;
    bp->visualize = p->visualize ? 1 : 0;
    const double span = (double)p->exhaustion_max + (double)p->excitation_max;
    bp->normer = (float)(255.0 / span);
    bp->centerer = (float)(((double)p->excitation_max / span) * 255.0);
    return SILENT_OK;
}

SILENT_EXPORT int silent_boosting_step_dev(silent_ctx* ctx, const float* input, const silent_extent* levels,
                                           int n_levels, int n_frames, const silent_boosting_params* params,
                                           float* energy, float* fired_out, float* energy_out, silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_boosting_step";
    if (!input || !energy || !fired_out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    BoostP bp;
    TRY(check_boost_params(ctx, params, &bp));
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    const long long n = tab.frame_px * n_frames;
    if ((n + 255) / 256 > 0x7fffffffll) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": too many pixels");
    TRY(workspace(ctx, (hipStream_t)stream, (size_t)n * sizeof(float)));
    float* m = (float*)ctx->ws.p;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(boost_power_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, input, energy, m, n);
    hipLaunchKernelGGL(boost_update_kernel, dim3((unsigned)blocks), dim3(256), 0, s, input, m, energy, fired_out,
                       energy_out, tab, bp);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_boosting_step_dev");
}

// ------------------------------------------------------------------------------------------ display-graph glue

SILENT_EXPORT int silent_affine_clip_dev(silent_ctx* ctx, const float* in, size_t n_values,
                                         const silent_affine_params* params, float* out, silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_affine_clip";
    if (!in || !out || !params) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (n_values == 0) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": empty tensor");
    if ((n_values + 2047) / 2048 > 0x7fffffffull) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": too many values");
    const AffineP ap = {params->mul, params->div, params->add, params->lo, params->hi, params->post_add};
    hipLaunchKernelGGL(affine_clip_kernel, dim3((unsigned)((n_values + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream,
                       in, out, (long long)n_values, ap);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_affine_clip_dev");
}

static size_t dtype_size(int dt) {
    switch (dt) {
        case SILENT_DT_U8: return 1;
        case SILENT_DT_U16: case SILENT_DT_I16: return 2;
        case SILENT_DT_F32: case SILENT_DT_I32: return 4;
        case SILENT_DT_F64: case SILENT_DT_I64: return 8;
        default: return 0;
    }
}

SILENT_EXPORT int silent_cast_interleave_dev(silent_ctx* ctx, const void* in, int in_dtype, size_t n_pixels, int in_stride,
                                             int in_offset, int count, float* out, int out_stride, int out_offset,
                                             silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_cast_interleave";
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (!dtype_size(in_dtype)) return fail(ctx, SILENT_E_UNSUPPORTED, std::string(who) + ": unknown in_dtype");
    if (n_pixels == 0) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": empty tensor");
    if (count < 1 || in_offset < 0 || out_offset < 0 || in_stride < in_offset + count || out_stride < out_offset + count)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": need 0 <= offset and offset + count <= stride on both sides");
    const long long total = (long long)n_pixels * count;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256ll * 64);
    hipStream_t s = (hipStream_t)stream;
    const long long n = (long long)n_pixels;
#define CAST_CASE(DT, T) \
    case DT: hipLaunchKernelGGL(cast_interleave_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)in, out, n, in_stride, in_offset, count, out_stride, out_offset); break
    switch (in_dtype) {
        CAST_CASE(SILENT_DT_U8, unsigned char);
        CAST_CASE(SILENT_DT_F32, float);
        CAST_CASE(SILENT_DT_F64, double);
        CAST_CASE(SILENT_DT_I32, int);
        CAST_CASE(SILENT_DT_U16, unsigned short);
        CAST_CASE(SILENT_DT_I16, short);
        CAST_CASE(SILENT_DT_I64, long long);
    }
#undef CAST_CASE
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_cast_interleave_dev");
}

SILENT_EXPORT int silent_resize_nearest_dev(silent_ctx* ctx, const float* in, const silent_extent* in_levels,
                                            int n_levels, int n_frames, int channels, const silent_extent* out_levels,
                                            float* out, silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_resize_nearest";
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": channels must be >= 1");
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, in_levels, n_levels, n_frames, 0, 0, &tab, &blocks));   // validates the input side
    TRY(build_level_tab(ctx, who, out_levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    ResizeTab rt;
    std::memset(&rt, 0, sizeof(rt));
    long long off = 0;
    for (int l = 0; l < n_levels; ++l) {
        rt.ih[l] = in_levels[l].h;
        rt.iw[l] = in_levels[l].w;
        rt.in_off[l] = off;
        off += (long long)in_levels[l].h * in_levels[l].w;
        rt.yscale[l] = (float)in_levels[l].h / (float)out_levels[l].h;
        rt.xscale[l] = (float)in_levels[l].w / (float)out_levels[l].w;
    }
    rt.in_px = off;
    hipLaunchKernelGGL(resize_nearest_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, tab, rt,
                       channels);
    return check_launch(ctx, who);
} catch (...) {
    return on_exception(ctx, "silent_resize_nearest_dev");
}

// ------------------------------------------------------------------------------------------ RGB chain

// Two-group structure of input channel i of a HWIO [3][3][3][3] kernel: every tap vector K[t][i][:] is a scalar
// multiple of one of two vectors.  Group A is the one that holds the centre tap.  Returns false when the 9 tap
// vectors need more than two directions (tolerance: 2e-7 of the largest weight, i.e. float32 rounding of the
// generators' float64 products).
static bool two_group_channel(const float* k, int i, unsigned* mask_a, float scale[9], float mix_a[3], float mix_b[3]) {
    double kmax = 0.0;
    for (int j = 0; j < 81; ++j) kmax = std::max(kmax, (double)std::fabs(k[j]));
    const double tol = 2e-7 * kmax;
    auto vec = [&](int t, int o) { return (double)k[(t * 3 + i) * 3 + o]; };
    auto fit = [&](int t, int ref, double* c) {  // is tap t a multiple of tap ref?
        double num = 0.0, den = 0.0;
        for (int o = 0; o < 3; ++o) {
            num += vec(t, o) * vec(ref, o);
            den += vec(ref, o) * vec(ref, o);
        }
        if (den == 0.0) return false;
        *c = num / den;
        for (int o = 0; o < 3; ++o)
            if (std::fabs(vec(t, o) - *c * vec(ref, o)) > tol) return false;
        return true;
    };
    auto norm = [&](int t) { return std::max(std::fabs(vec(t, 0)), std::max(std::fabs(vec(t, 1)), std::fabs(vec(t, 2)))); };
    int ref_a = 4;
    if (norm(4) <= tol) {  // centre tap is zero: take the largest tap instead
        for (int t = 0; t < 9; ++t)
            if (norm(t) > norm(ref_a)) ref_a = t;
    }
    int group[9];
    int ref_b = -1;
    for (int t = 0; t < 9; ++t) {
        double c = 0.0;
        if (norm(t) <= tol) {
            group[t] = 0;
            scale[t] = 0.0f;
        } else if (fit(t, ref_a, &c)) {
            group[t] = 0;
            scale[t] = (float)c;
        } else {
            group[t] = 1;
            if (ref_b < 0 || norm(t) > norm(ref_b)) ref_b = t;
        }
    }
    for (int t = 0; t < 9; ++t) {
        if (group[t] != 1) continue;
        double c = 0.0;
        if (!fit(t, ref_b, &c)) return false;
        scale[t] = (float)c;
    }
    *mask_a = 0;
    for (int t = 0; t < 9; ++t)
        if (group[t] == 0) *mask_a |= 1u << t;
    for (int o = 0; o < 3; ++o) {
        mix_a[o] = (float)vec(ref_a, o);
        mix_b[o] = ref_b >= 0 ? (float)vec(ref_b, o) : 0.0f;
    }
    return true;
}

// group-A masks of the kernels the reference's generators produce (rgby_3(2): centre tap; rgb_2d_end_tensors():
// per orientation the taps on the centre's side of the facet); pinned by tests/test_host_logic.py through
// silent_rgb_chain_structure
constexpr unsigned kRgbyA = 0x010u, kEndA0 = 0x1f9u, kEndA1 = 0x119u, kEndA2 = 0x11fu;

struct RgbStructure {
    bool rgc_diag, stripe_sum, rgby_two, end_two;
    unsigned rgby_mask[3], end_mask[3];
    float rgby_w[45], end_w[45];  // structured weight blocks: scale[dy][dx][i], mixA[i][o], mixB[i][o]
    // the symmetric forms of silent_rgb2.h (RgbSym): rgc per channel mirror-symmetric in both axes; rgby = S (x) A around the
    // centre + B at the centre with S mirror-symmetric in both axes
    bool rgc_sym, rgby_mix;
    RgbSym sym;
};

// K[t][i][o] (HWIO, t = dy * 3 + dx) = S[t] * A[i][o] for t != centre, with S[t] = S[mirror(t)]?  A is the tap vector of the
// largest off-centre tap (S = 1 there), S the least-squares factor of every other tap; accepted when the float32 factors
// reproduce every weight within 2e-7 of the largest one (the tolerance of two_group_channel).  B = the centre tap as it is.
static bool rgby_mix_form(const float* k, RgbSym* sym) {
    double kmax = 0.0;
    for (int j = 0; j < 81; ++j) kmax = std::max(kmax, (double)std::fabs(k[j]));
    const double tol = 2e-7 * kmax;
    auto at = [&](int t, int io) { return (double)k[t * 9 + io]; };
    int ref = -1;
    double best = 0.0;
    for (int t = 0; t < 9; ++t) {
        if (t == 4) continue;
        double n = 0.0;
        for (int io = 0; io < 9; ++io) n = std::max(n, std::fabs(at(t, io)));
        if (n > best) {
            best = n;
            ref = t;
        }
    }
    if (ref < 0 || best <= tol) return false;
    float S[9];
    double den = 0.0;
    for (int io = 0; io < 9; ++io) den += at(ref, io) * at(ref, io);
    for (int t = 0; t < 9; ++t) {
        double num = 0.0;
        for (int io = 0; io < 9; ++io) num += at(t, io) * at(ref, io);
        S[t] = t == 4 ? 0.0f : (float)(num / den);
    }
    for (int t = 0; t < 9; ++t)
        for (int io = 0; io < 9 && t != 4; ++io)
            if (std::fabs((double)S[t] * (double)k[ref * 9 + io] - at(t, io)) > tol) return false;
    if (S[0] != S[2] || S[0] != S[6] || S[0] != S[8] || S[1] != S[7] || S[3] != S[5]) return false;
    for (int io = 0; io < 9; ++io) {
        sym->rgby[io] = k[ref * 9 + io];        // A[i][o]
        sym->rgby[9 + io] = S[io];              // S[dy][dx] (io used as t)
        sym->rgby[18 + io] = k[4 * 9 + io];     // B[i][o]
    }
    return true;
}

static void analyze_rgb_chain(const silent_rgb_chain_params* p, RgbStructure* r) {
    r->rgc_diag = r->stripe_sum = true;
    for (int t = 0; t < 9; ++t)
        for (int i = 0; i < 3; ++i)
            for (int o = 0; o < 3; ++o) {
                if (i != o && p->rgc[(t * 3 + i) * 3 + o] != 0.0f) r->rgc_diag = false;
                if (p->stripe[(t * 3 + i) * 3 + o] != p->stripe[(t * 3 + 0) * 3 + o]) r->stripe_sum = false;
            }
    auto two = [](const float* k, unsigned mask[3], float w[45]) {
        std::memset(w, 0, sizeof(float) * 45);
        for (int i = 0; i < 3; ++i) {
            float sc[9], ma[3], mb[3];
            if (!two_group_channel(k, i, &mask[i], sc, ma, mb)) return false;
            for (int t = 0; t < 9; ++t) w[t * 3 + i] = sc[t];
            for (int o = 0; o < 3; ++o) {
                w[kStructMixA + i * 3 + o] = ma[o];
                w[kStructMixB + i * 3 + o] = mb[o];
            }
        }
        return true;
    };
    r->rgby_two = two(p->rgby, r->rgby_mask, r->rgby_w);
    r->end_two = two(p->end, r->end_mask, r->end_w);
    std::memset(&r->sym, 0, sizeof(r->sym));
    r->rgc_sym = r->rgc_diag;
    for (int c = 0; c < 3 && r->rgc_sym; ++c) {
        auto w = [&](int dy, int dx) { return p->rgc[((dy * 3 + dx) * 3 + c) * 3 + c]; };
        if (w(0, 0) != w(0, 2) || w(0, 0) != w(2, 0) || w(0, 0) != w(2, 2) || w(0, 1) != w(2, 1) || w(1, 0) != w(1, 2)) r->rgc_sym = false;
        r->sym.rgc[c * 4 + 0] = w(0, 0);
        r->sym.rgc[c * 4 + 1] = w(0, 1);
        r->sym.rgc[c * 4 + 2] = w(1, 0);
        r->sym.rgc[c * 4 + 3] = w(1, 1);
    }
    r->rgby_mix = rgby_mix_form(p->rgby, &r->sym);
}

// The RGB chain's weights as the fused kernels take them: HWIO -> [o][dy][dx][i], the blur's profile, and -- where the
// host FINDS the structure in the actual weights (what the reference's generators produce, but checked, not assumed) --
// the two-group blocks: rgc channel-diagonal (27 fmas), stripe a filter of the channel sum (27), rgby and the end bank
// two-group (27 + 18 each), and the blur mirror-symmetric (16 fmas + 10 adds instead of 49 fmas): 160 weights instead of 373
// per pixel.  Anything else runs the basic (diagonal rgc + channel-sum stripe) or the dense instantiation.
// kopts: SILENT_TUNE_RGB bits 0 (dense) and 1 (no two-group form).
static void pack_rgb_weights(const silent_rgb_chain_params* p, unsigned kopts, RgbW* w, bool* basic, bool* two, RgbSym* sym = nullptr,
                             bool* use_sym = nullptr) {
    auto repack = [](const float* hwio, float* dst) {  // HWIO [dy][dx][i][o] -> [o][dy][dx][i]
        for (int o = 0; o < 3; ++o)
            for (int dy = 0; dy < 3; ++dy)
                for (int dx = 0; dx < 3; ++dx)
                    for (int i = 0; i < 3; ++i) dst[((o * 3 + dy) * 3 + dx) * 3 + i] = hwio[((dy * 3 + dx) * 3 + i) * 3 + o];
    };
    repack(p->rgc, w->rgc);
    repack(p->rgby, w->rgby);
    repack(p->stripe, w->stripe);
    repack(p->end, w->end);
    for (int t = 0; t < 49; ++t) w->blur[t] = p->blur[t * 9];
    RgbStructure rs;
    analyze_rgb_chain(p, &rs);
    *basic = rs.rgc_diag && rs.stripe_sum && !(kopts & 1u);
    // the blur's mirror symmetry (what blur_tensor generates: a function of the distance), folded by the two-group kernels
    bool blur_sym = true;
    for (int dy = 0; dy < 7; ++dy)
        for (int dx = 0; dx < 7; ++dx)
            if (w->blur[dy * 7 + dx] != w->blur[(6 - dy) * 7 + dx] || w->blur[dy * 7 + dx] != w->blur[dy * 7 + (6 - dx)]) blur_sym = false;
    *two = *basic && !(kopts & 2u) && blur_sym && rs.rgby_two && rs.end_two && rs.rgby_mask[0] == kRgbyA && rs.rgby_mask[1] == kRgbyA &&
           rs.rgby_mask[2] == kRgbyA && rs.end_mask[0] == kEndA0 && rs.end_mask[1] == kEndA1 && rs.end_mask[2] == kEndA2;
    if (*two) {
        std::memcpy(w->rgby, rs.rgby_w, sizeof(rs.rgby_w));
        std::memcpy(w->end, rs.end_w, sizeof(rs.end_w));
    }
    // the symmetric forms on top of the two-group ones (pair kernel only; kopts bit 6 keeps the two-group instantiation)
    if (use_sym) *use_sym = *two && !(kopts & 64u) && rs.rgc_sym && rs.rgby_mix;
    if (sym) *sym = rs.sym;
}

SILENT_EXPORT int silent_rgb_chain_stream(const silent_rgb_chain_params* params, unsigned knobs, float* stream, int* n_used,
                                          int* variant) try {
    if (!params || !stream || !n_used || !variant || !params->rgc || !params->rgby || !params->stripe || !params->blur || !params->end)
        return SILENT_E_INVALID;
    RgbW w;
    RgbSym sym;
    bool basic, two, use_sym;
    pack_rgb_weights(params, knobs, &w, &basic, &two, &sym, &use_sym);
    std::memset(stream, 0, sizeof(float) * SILENT_RGB_STREAM_MAX);
    *n_used = rgb2_fill_stream(w, basic ? 0x111u : 0x1ffu, basic, two, two, stream, use_sym ? &sym : nullptr);
    *variant = use_sym ? 3 : two ? 2 : basic ? 1 : 0;
    return SILENT_OK;
} catch (...) {
    return on_exception(nullptr, "silent_rgb_chain_stream");
}

SILENT_EXPORT int silent_rgb_chain_structure(const silent_rgb_chain_params* params, unsigned* flags, unsigned* masks) try {
    if (!params || !flags || !params->rgc || !params->rgby || !params->stripe || !params->end) return SILENT_E_INVALID;
    RgbStructure r;
    analyze_rgb_chain(params, &r);
    *flags = (r.rgc_diag ? 1u : 0u) | (r.stripe_sum ? 2u : 0u) | (r.rgby_two ? 4u : 0u) | (r.end_two ? 8u : 0u) | (r.rgc_sym ? 16u : 0u) |
             (r.rgby_mix ? 32u : 0u);
    if (masks)
        for (int i = 0; i < 3; ++i) {
            masks[i] = r.rgby_two ? r.rgby_mask[i] : 0u;
            masks[3 + i] = r.end_two ? r.end_mask[i] : 0u;
        }
    return SILENT_OK;
} catch (...) {
    return on_exception(nullptr, "silent_rgb_chain_structure");
}


// Which fused kernel a chain launch over these levels uses and its tile height (output rows per tile).
static int rgb_chain_tile_height(const silent_ctx* ctx, const silent_extent* levels, int n_levels, int n_frames, bool* pair) {
    const unsigned kopts = ctx->tune[SILENT_TUNE_RGB];  // 1: dense, 2: no two-group, 8: 90-row tiles, bits 8-15: tile height / 2
    // two pixels per lane on packed f32 (silent_rgb2.h; its buffer addressing wants levels below 2^30 bytes per map); 16: one pixel per lane
    bool pair_kernel = !(kopts & 16u);
    // (its range-check addressing: (H + 16) rows of a map below kRgb2Out, (H + tile height + 16) rows below 2^32 - kRgb2Out)
    for (int l = 0; l < n_levels; ++l)
        if ((long long)(levels[l].h + 16) * levels[l].w * 12 >= (long long)kRgb2Out ||
            (long long)(levels[l].h + 512 + 16) * levels[l].w * 12 >= (1ll << 32) - (long long)kRgb2Out)
            pair_kernel = false;
    // tile height: the one that minimises ceil(tiles / resident tiles) x (th + 14) row steps (silent_rgb.h)
    int th = kRgbTH;
    if ((kopts >> 8) & 0xffu) {
        th = std::min(std::max((int)((kopts >> 8) & 0xffu) * 2, 2), 400);
    } else if (!(kopts & 8u)) {
        const long long resident = (pair_kernel ? 16ll / kRgb2Waves : 5ll) * ctx->n_cus;   // 128 / 94 VGPRs, 256 threads: 4 / 5 tiles per CU
        const int tw = pair_kernel ? kRgb2TW : kRgbTW;
        long long best = -1;
        // Launches that fill the chip several times over keep round 1's 90 rows: a sweep on config 3 (scripts/sweep_rgb_th.py:
        // 50 ... 156 rows = 1.40 1.32 1.30 1.37 1.35 1.40 1.32 1.36 1.30 1.37 ms) shows +-4 % with no trend the rounds
        // model predicts (tiles are not equal: ragged edges, small levels).  The model decides where it is sharp: launches
        // of about one round or less, where it picks short tiles (the latency of one wave's row walk sets the time).
        long long tiles90 = 0;
        for (int l = 0; l < n_levels; ++l)
            tiles90 += (long long)((levels[l].w + tw - 1) / tw) * ((levels[l].h + kRgbTH - 1) / kRgbTH);
        const bool model = tiles90 * n_frames < 2 * resident;
        for (int cand = kRgbTHMin; model && cand <= kRgbTHMax; cand += 2) {
            long long tiles = 0;
            for (int l = 0; l < n_levels; ++l)
                tiles += (long long)((levels[l].w + tw - 1) / tw) * ((levels[l].h + cand - 1) / cand);
            tiles *= n_frames;
            const long long cost = ((tiles + resident - 1) / resident) * (cand + 2 * kRgbHalo);
            if (best < 0 || cost < best) {
                best = cost;
                th = cand;
            }
        }
    }
    *pair = pair_kernel;
    return th;
}

// mm: optional per-level extrema slots (already initialised); *mm_done tells whether the launch filled them (only the pair
// kernel's two-group instantiation does -- everything else leaves them to level_maxmin_kernel)
static int rgb_chain_launch(silent_ctx* ctx, const char* who, const float* pyr, const silent_extent* levels, int n_levels,
                            int n_frames, const silent_rgb_chain_params* p, float* orient_out, float* line_end_out,
                            float* value_out, unsigned* mm, bool* mm_done, silent_stream stream, const SumTab* st = nullptr,
                            float* sum = nullptr, int* nan_flags = nullptr) {
    if (mm_done) *mm_done = false;
    if (!pyr || !p) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (!p->rgc || !p->rgby || !p->stripe || !p->blur || !p->end)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": a kernel pointer in params is NULL");
    if (!orient_out && !line_end_out && !value_out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": all outputs are NULL");
    if (p->pad < 0) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": pad must be >= 0");
    if (!levels || n_levels < 1 || n_levels > kMaxLevels || n_frames < 1)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": bad levels / n_frames");
    for (int l = 0; l < n_levels; ++l)
        if (levels[l].h < 1 || levels[l].w < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": bad level extent");
    hipStream_t s = (hipStream_t)stream;
    // Fused single-launch path: needs a channel-uniform blur (what blur_tensor generates), so that the 7x7x3x3
    // blur is a 49-tap filter of the channel sum.
    bool uniform_blur = true;
    for (int t = 0; t < 49 && uniform_blur; ++t)
        for (int io = 1; io < 9; ++io)
            if (p->blur[t * 9 + io] != p->blur[t * 9]) uniform_blur = false;
    if (uniform_blur) {
        LevelTab tab;
        long long blocks;
        const unsigned kopts = ctx->tune[SILENT_TUNE_RGB];  // 1: dense, 2: no two-group, 8: 90-row tiles, bits 8-15: tile height / 2
        bool pair_kernel;
        const int th = rgb_chain_tile_height(ctx, levels, n_levels, n_frames, &pair_kernel);
        TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, pair_kernel ? kRgb2TW : kRgbTW, th, &tab, &blocks));
        if (p->flat_policy != SILENT_FLAT_IEEE && p->flat_policy != SILENT_FLAT_ZERO)
            return fail(ctx, SILENT_E_INVALID, std::string(who) + ": flat_policy must be SILENT_FLAT_IEEE or SILENT_FLAT_ZERO");
        RgbArgs a;
        a.pyr = pyr;
        a.orient_out = orient_out;
        a.line_out = line_end_out;
        a.value_out = value_out;
        a.tab = tab;
        a.th = th;
        a.prm = RgbP{p->regulation_value, p->regulation_root, p->flat_policy, p->clip_hi, p->pad};
        bool basic, two, use_sym;
        RgbSym sym;
        pack_rgb_weights(p, kopts, &a.w, &basic, &two, &sym, &use_sym);
        if (pair_kernel) {
            Rgb2Args a2;
            a2.pyr = a.pyr;
            a2.orient_out = a.orient_out;
            a2.line_out = a.line_out;
            a2.value_out = a.value_out;
            a2.tab = a.tab;
            a2.prm = a.prm;
            a2.th = a.th;
            a2.mm = nullptr;
            a2.sum = nullptr;
            a2.nan_flags = nullptr;
            a2.sum_frame = 0;
            std::memset(a2.sum_off, 0, sizeof(a2.sum_off));
            std::memset(a2.ws, 0, sizeof(a2.ws));
            rgb2_fill_stream(a.w, basic ? 0x111u : 0x1ffu, basic, two, two, a2.ws, use_sym ? &sym : nullptr);
            // 16-byte-per-lane stores of orient / line_end (ST4, silent_rgb2.h; RGB knob bit 7 -- an alternating A/B on config 3 put
            // it at 0.8063 against 0.8043 ms per launch for the 12-byte form, profiles/r04/evidence/ab_st4.txt: the store
            // instructions are not what the kernel waits for, so the simpler form stays the default): every row of every map must
            // start on a 16-byte boundary -- widths, level offsets and the frame stride multiples of 4 pixels, the map pointers
            // 16-byte aligned (all BASELINE extents; anything else keeps the 12-byte form).
            bool st4 = (kopts & 128u) && tab.frame_px % 4 == 0 && (uintptr_t)orient_out % 16 == 0 && (uintptr_t)line_end_out % 16 == 0;
            for (int l = 0; l < n_levels && st4; ++l) st4 = levels[l].w % 4 == 0 && tab.px_off[l] % 4 == 0;
            // silent_set_profiling: HIP events around THIS launch, on the stream it runs on (the fused RGB chain is the dominant
            // kernel of silent_rgb_line_end / silent_rgb_keypoints, like gray_stream_kernel is of silent_gray_pass)
            const bool prof = ctx->profiling && (ctx->prof_calls++ % ctx->prof_period) == 0;
            const int prof_slot = ctx->prof_recorded % silent_ctx::kProfPairs;
            if (prof) HIP_TRY(ctx, hipEventRecord(ctx->prof_ev[prof_slot][0], s));
            struct ProfEnd {
                silent_ctx* c; bool on; int slot; hipStream_t st; long long px;
                ~ProfEnd() {
                    if (!on) return;
                    if (hipEventRecord(c->prof_ev[slot][1], st) == hipSuccess) {
                        ++c->prof_recorded;
                        c->prof_pixels = px;
                    } else (void)hipGetLastError();
                }
            } prof_end{ctx, prof, prof_slot, s, tab.frame_px * n_frames};
            if (two && mm) {
                a2.mm = mm;
                if (st && sum && st->frame_entries > 0 && st->th == th) {   // value summary for the sparse selection tail
                    a2.sum = sum;
                    a2.nan_flags = nan_flags;
                    a2.sum_frame = st->frame_entries;
                    for (int l = 0; l < kMaxLevels; ++l) a2.sum_off[l] = st->off[l];
                }
                if (use_sym && st4) hipLaunchKernelGGL((rgb_line_end2_kernel<0x111u, true, kRgbyA, kEndA0, kEndA1, kEndA2, true, true, true>), dim3((unsigned)blocks), dim3(64 * kRgb2Waves), 0, s, a2);
                else if (use_sym) hipLaunchKernelGGL((rgb_line_end2_kernel<0x111u, true, kRgbyA, kEndA0, kEndA1, kEndA2, true, true>), dim3((unsigned)blocks), dim3(64 * kRgb2Waves), 0, s, a2);
                else hipLaunchKernelGGL((rgb_line_end2_kernel<0x111u, true, kRgbyA, kEndA0, kEndA1, kEndA2, true>), dim3((unsigned)blocks), dim3(64 * kRgb2Waves), 0, s, a2);
                if (mm_done) *mm_done = true;
            } else
            if (use_sym && st4) hipLaunchKernelGGL((rgb_line_end2_kernel<0x111u, true, kRgbyA, kEndA0, kEndA1, kEndA2, false, true, true>), dim3((unsigned)blocks), dim3(64 * kRgb2Waves), 0, s, a2);
            else if (use_sym) hipLaunchKernelGGL((rgb_line_end2_kernel<0x111u, true, kRgbyA, kEndA0, kEndA1, kEndA2, false, true>), dim3((unsigned)blocks), dim3(64 * kRgb2Waves), 0, s, a2);
            else if (two) hipLaunchKernelGGL((rgb_line_end2_kernel<0x111u, true, kRgbyA, kEndA0, kEndA1, kEndA2>), dim3((unsigned)blocks), dim3(64 * kRgb2Waves), 0, s, a2);
            else if (basic) hipLaunchKernelGGL((rgb_line_end2_kernel<0x111u, true, kDense, kDense, kDense, kDense>), dim3((unsigned)blocks), dim3(64 * kRgb2Waves), 0, s, a2);
            else hipLaunchKernelGGL((rgb_line_end2_kernel<0x1ffu, false, kDense, kDense, kDense, kDense>), dim3((unsigned)blocks), dim3(64 * kRgb2Waves), 0, s, a2);
        } else if (two) {
            hipLaunchKernelGGL((rgb_line_end_kernel<0x111u, true, kRgbyA, kEndA0, kEndA1, kEndA2>), dim3((unsigned)blocks), dim3(256), 0, s, a);
        } else if (basic) {
            hipLaunchKernelGGL((rgb_line_end_kernel<0x111u, true, kDense, kDense, kDense, kDense>), dim3((unsigned)blocks), dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL((rgb_line_end_kernel<0x1ffu, false, kDense, kDense, kDense, kDense>), dim3((unsigned)blocks), dim3(256), 0, s, a);
        }
        return check_launch(ctx, who);
    }
    // General blur: stage-per-launch composition through ping-pong temporaries in the context workspace.
    const size_t n = (size_t)pyramid_px(levels, n_levels) * n_frames * 3;
    const size_t bytes = align_up(n * sizeof(float));
    TRY(workspace(ctx, (hipStream_t)stream, 3 * bytes));
    float* t0 = (float*)ctx->ws.p;
    float* t1 = (float*)((char*)ctx->ws.p + bytes);
    float* t2 = (float*)((char*)ctx->ws.p + 2 * bytes);
    const Epilogue relu{SILENT_RELU, 0.f, 0.f, 0.f, 0};
    TRY(conv_dispatch(ctx, who, pyr, levels, n_levels, n_frames, 3, p->rgc, 3, 3, 3, false, relu, t0, s));
    TRY(conv_dispatch(ctx, who, t0, levels, n_levels, n_frames, 3, p->rgby, 3, 3, 3, false, relu, t1, s));
    TRY(conv_dispatch(ctx, who, t1, levels, n_levels, n_frames, 3, p->stripe, 3, 3, 3, false, relu, t0, s));
    float* orient = orient_out ? orient_out : t1;
    const Epilogue reg{0u, 0.f, p->regulation_value, p->regulation_root, p->flat_policy};
    TRY(conv_dispatch(ctx, who, t0, levels, n_levels, n_frames, 3, p->blur, 7, 7, 3, true, reg, orient, s));
    if (!line_end_out && !value_out) return SILENT_OK;
    const Epilogue rc{SILENT_RELU | SILENT_CLIP, p->clip_hi, 0.f, 0.f, 0};
    TRY(conv_dispatch(ctx, who, orient, levels, n_levels, n_frames, 3, p->end, 3, 3, 3, false, rc, t0, s));
    float* padded = line_end_out ? line_end_out : t2;
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    hipLaunchKernelGGL(pad_inwards_kernel, dim3((unsigned)blocks), dim3(256), 0, s, t0, padded, tab, 3, p->pad, p->pad,
                       p->pad, p->pad);
    if (value_out) {
        const long long npx = tab.frame_px * n_frames;
        const long long grid = std::min<long long>((npx + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(value_from_color_kernel, dim3((unsigned)grid), dim3(256), 0, s, padded, value_out, npx, 3);
    }
    return check_launch(ctx, who);
}

SILENT_EXPORT int silent_rgb_line_end_dev(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels,
                                          int n_frames, const silent_rgb_chain_params* p, float* orient_out,
                                          float* line_end_out, float* value_out, silent_stream stream) try {
    NEED_CTX(ctx);
    return rgb_chain_launch(ctx, "silent_rgb_line_end", pyr, levels, n_levels, n_frames, p, orient_out, line_end_out, value_out,
                            nullptr, nullptr, stream);
} catch (...) {
    return on_exception(ctx, "silent_rgb_line_end_dev");
}

// Config 3 from the pyramid on in one call: silent_rgb_line_end + silent_select_keypoints on its line_end / value maps.
// When the chain runs as the pair kernel's two-group instantiation, that kernel also accumulates the per-level extrema of the
// value map (a-10's max / min), so the reduction pass is skipped and nobody needs the value map in memory: the selection pass
// takes the value from line_end (same three operations, same bits) and the map is written only if the caller asks for it.
SILENT_EXPORT int silent_rgb_keypoints_dev(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels,
                                           int n_frames, const silent_rgb_chain_params* p, double top_percent,
                                           const silent_extent* regions, float* orient_out, float* line_end_out,
                                           float* value_out, float* peak_value_out, int64_t* idx, size_t cap_per_frame,
                                           int64_t* counts, silent_stream stream) try {
    NEED_CTX(ctx);
    const char* who = "silent_rgb_keypoints";
    if (!pyr || !p || !regions || !line_end_out || !counts || (!idx && cap_per_frame))
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (!p->rgc || !p->rgby || !p->stripe || !p->blur || !p->end)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": a kernel pointer in params is NULL");
    bool uniform_blur = true;   // anything else makes the chain use the context workspace itself: plain sequence then
    for (int t = 0; t < 49 && uniform_blur; ++t)
        for (int io = 1; io < 9; ++io)
            if (p->blur[t * 9 + io] != p->blur[t * 9]) uniform_blur = false;
    hipStream_t s = (hipStream_t)stream;
    if (!levels || n_levels < 1 || n_levels > kMaxLevels || n_frames < 1)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": bad levels / n_frames");
    for (int l = 0; l < n_levels; ++l)
        if (levels[l].h < 1 || levels[l].w < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": bad level extent");
    if (!uniform_blur) {
        TRY(rgb_chain_launch(ctx, who, pyr, levels, n_levels, n_frames, p, orient_out, line_end_out, value_out, nullptr, nullptr, stream));
        return silent_select_keypoints_dev(ctx, line_end_out, value_out, levels, n_levels, n_frames, 3, top_percent, regions,
                                           peak_value_out, idx, cap_per_frame, counts, stream);
    }
    // Without a caller-side peak-value map the tail runs sparse (silent_peaks.h, sparse_select_kernel): the chain kernel leaves
    // a per-group maximum of the value map, and selection / NMS / keypoint search look only at the groups that reach their
    // level's threshold; whatever that cannot settle exactly runs the dense kernels on a map in the workspace.
    bool pair_kernel = false;
    const int th = rgb_chain_tile_height(ctx, levels, n_levels, n_frames, &pair_kernel);
    const bool want_sparse = pair_kernel && !(ctx->tune[SILENT_TUNE_RGB] & 32u);
    SelectPlan sp;
    TRY(select_prepare(ctx, who, levels, n_levels, n_frames, regions, s, &sp, want_sparse ? th : 0, !peak_value_out));
    bool mm_done = false;
    TRY(rgb_chain_launch(ctx, who, pyr, levels, n_levels, n_frames, p, orient_out, line_end_out, value_out, sp.mm, &mm_done, stream,
                         &sp.st, sp.w.sum, sp.w.nan_flags));
    const bool sparse = want_sparse && mm_done && sp.st.frame_entries > 0;
    ctx->sparse_ran = sparse;
    ctx->sparse_stream = s;
    ctx->sparse_flags_off = (size_t)((char*)sp.w.dense_flags - (char*)ctx->ws.p);
    ctx->sparse_candn_off = (size_t)((char*)sp.w.cand_n - (char*)ctx->ws.p);
    ctx->sparse_pairs = n_frames * n_levels;
    ctx->sparse_frames = n_frames;
    return select_run(ctx, who, line_end_out, mm_done ? nullptr : value_out, levels, n_levels, n_frames, 3, top_percent, sp, mm_done,
                      peak_value_out, idx, cap_per_frame, counts, s, sparse);
} catch (...) {
    return on_exception(ctx, "silent_rgb_keypoints_dev");
}


SILENT_EXPORT int silent_sparse_tail_stats(silent_ctx* ctx, int64_t* stats) try {
    NEED_CTX(ctx);
    if (!stats) return fail(ctx, SILENT_E_INVALID, "silent_sparse_tail_stats: stats is NULL");
    stats[0] = ctx->sparse_ran ? 1 : 0;
    stats[1] = stats[2] = stats[3] = stats[4] = 0;
    if (!ctx->sparse_ran || !ctx->ws.p) return SILENT_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->sparse_stream));
    std::vector<int> flags((size_t)ctx->sparse_pairs), cn((size_t)ctx->sparse_frames);
    HIP_TRY(ctx, hipMemcpy(flags.data(), (char*)ctx->ws.p + ctx->sparse_flags_off, flags.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(cn.data(), (char*)ctx->ws.p + ctx->sparse_candn_off, cn.size() * sizeof(int), hipMemcpyDeviceToHost));
    stats[1] = ctx->sparse_pairs;
    for (int f : flags) stats[2] += f == kTailDense ? 1 : 0;
    for (int f : flags) stats[4] += f == kTailZero ? 1 : 0;
    for (int c : cn) stats[3] += c;
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_sparse_tail_stats");
}

// ------------------------------------------------------------------------------------------ pyramid plan

static void spline5_weights(double t, double* w) {
    // quintic cardinal B-spline at taps floor(c)-2 .. floor(c)+3; last tap by partition of unity
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

static int host_mirror(long i, int n) {
    if (n == 1) return 0;
    const long period = 2L * (n - 1);
    if (i < 0) i = -i;
    i %= period;
    return (int)(i >= n ? period - i : i);
}

// scipy.ndimage.zoom, grid_mode=False: output o samples o * (n_in-1)/(n_out-1); mode 'constant'
// declares a coordinate outside [0, n_in-1] out of bounds (-> cval 0 for the whole row/column).
static void axis_table(int n_in, int n_out, int* base, int* idx, float* wts) {
    const double step = n_out > 1 ? (double)(n_in - 1) / (double)(n_out - 1) : 1.0;
    for (int o = 0; o < n_out; ++o) {
        const double c = (double)o * step;
        const long b = (long)std::floor(c);
        double w[6] = {0, 0, 0, 0, 0, 0};
        if (c >= 0.0 && c <= (double)(n_in - 1)) spline5_weights(c - (double)b, w);
        base[o] = (int)b;
        for (int j = 0; j < 6; ++j) {
            wts[6 * o + j] = (float)w[j];
            idx[6 * o + j] = host_mirror(b - 2 + j, n_in);
        }
    }
}

SILENT_EXPORT int silent_pyramid_plan_create(silent_ctx* ctx, int frame_h, int frame_w, int channels,
                                             const silent_pyr_level* levels, int n_levels,
                                             silent_pyramid_plan** out) try {
    NEED_CTX(ctx);
    const char* who = "silent_pyramid_plan_create";
    if (!out || !levels) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    *out = nullptr;
    if (frame_h < 1 || frame_w < 1 || (long long)frame_h * frame_w > (1ll << 30))
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": bad frame extent");
    if (channels != 1 && channels != 3)
        return fail(ctx, SILENT_E_UNSUPPORTED, std::string(who) + ": channels must be 1 or 3");
    if (n_levels < 1 || n_levels > kMaxLevels)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": n_levels must be in [1, " + std::to_string(kMaxLevels) + "]");
    silent_pyramid_plan* plan = new (std::nothrow) silent_pyramid_plan();
    if (!plan) return fail(ctx, SILENT_E_NOMEM, std::string(who) + ": out of host memory");
    plan->ctx = ctx;
    PyrTab& tab = plan->tab;
    tab.n_levels = n_levels;
    tab.H = frame_h;
    tab.W = frame_w;
    tab.C = channels;
    const int RW = channels == 1 ? region_w(1) : region_w(3);
    tab.regions_x = (frame_w + RW - 1) / RW;
    const int RH = channels == 1 ? region_h(1) : region_h(3);
    tab.regions_y = (frame_h + RH - 1) / RH;
    long long cols = 0, rows = 0, px = 0;
    for (int l = 0; l < n_levels; ++l) {
        const silent_pyr_level& L = levels[l];
        const bool ok = L.src_h >= 1 && L.src_w >= 1 && L.src_y0 >= 0 && L.src_x0 >= 0 &&
                        (long long)L.src_y0 + L.src_h <= frame_h && (long long)L.src_x0 + L.src_w <= frame_w &&
                        L.zoom_h >= 1 && L.zoom_w >= 1 && L.out_h >= 1 && L.out_w >= 1 &&
                        (long long)L.out_h * L.out_w <= (1ll << 30) && (long long)L.zoom_h * L.zoom_w <= (1ll << 30);
        if (!ok) {
            delete plan;
            return fail(ctx, SILENT_E_INVALID, std::string(who) + ": level " + std::to_string(l) + " geometry is invalid");
        }
        PyrLevelDev& d = tab.lv[l];
        d.src_y0 = L.src_y0; d.src_x0 = L.src_x0; d.src_h = L.src_h; d.src_w = L.src_w;
        d.zoom_h = L.zoom_h; d.zoom_w = L.zoom_w; d.out_h = L.out_h; d.out_w = L.out_w;
        d.xtab_off = (int)cols;
        d.ytab_off = (int)rows;
        cols += L.zoom_w;
        rows += L.zoom_h;
        tab.px_off[l] = px;
        px += (long long)L.out_h * L.out_w;
        plan->extents.push_back(silent_extent{L.out_h, L.out_w});
    }
    tab.frame_px_out = px;
    std::vector<int> xbase(cols), xidx(cols * 6), ybase(rows), yidx(rows * 6), xreg, yreg;
    std::vector<float> xw(cols * 6), yw(rows * 6);
    long long unit_tiles = 0, zero_chunks = 0;
    tab.n_general = 0;
    bool tap_range_ok = true;
    for (int l = 0; l < n_levels; ++l) {
        PyrLevelDev& d = tab.lv[l];
        int* xb = xbase.data() + d.xtab_off;
        int* yb = ybase.data() + d.ytab_off;
        float* xwl = xw.data() + (size_t)d.xtab_off * 6;
        float* ywl = yw.data() + (size_t)d.ytab_off * 6;
        axis_table(d.src_w, d.zoom_w, xb, xidx.data() + (size_t)d.xtab_off * 6, xwl);
        axis_table(d.src_h, d.zoom_h, yb, yidx.data() + (size_t)d.ytab_off * 6, ywl);
        // zoom factor exactly 1 <=> every output samples an integer coordinate: weights [1,26,66,26,1,~0]/120
        // (the streaming unit kernels mirror with one reflection: needs at least kMirrorNearMin source pixels per axis)
        const bool unit = d.zoom_h == d.src_h && d.zoom_w == d.src_w && std::fabs(xwl[5]) < 1e-12f &&
                          std::fabs(ywl[5]) < 1e-12f && d.src_h >= kMirrorNearMin && d.src_w >= kMirrorNearMin;
        d.kind = unit ? kPyrUnit : kPyrGeneral;
        tab.unit_tile_start[l] = (int)unit_tiles;
        tab.zero_chunk_start[l] = (int)zero_chunks;
        tab.unit_tiles_x[l] = (d.out_w + kUnitTW - 1) / kUnitTW;
        d.xreg_off = (int)xreg.size();
        d.yreg_off = (int)yreg.size();
        if (unit) {
            unit_tiles += (long long)tab.unit_tiles_x[l] * ((d.out_h + kUnitTH - 1) / kUnitTH);
            for (int j = 0; j < 5; ++j) plan->unit_w[j] = xwl[j];
            continue;
        }
        ++tab.n_general;
        // outputs are owned by the region that holds their ANCHOR = floor(source coordinate), frame coordinates
        const int zc = std::min(d.zoom_w, d.out_w), zr = std::min(d.zoom_h, d.out_h);
        int o = 0;
        for (int r = 0; r <= tab.regions_x; ++r) {
            while (o < zc && xb[o] + d.src_x0 < r * RW) ++o;
            xreg.push_back(r == tab.regions_x ? zc : o);
        }
        o = 0;
        for (int r = 0; r <= tab.regions_y; ++r) {
            while (o < zr && yb[o] + d.src_y0 < r * RH) ++o;
            yreg.push_back(r == tab.regions_y ? zr : o);
        }
        // every mirrored tap of an anchored output must lie inside its region's staged tile (see the kernel)
        const int* xi = xidx.data() + (size_t)d.xtab_off * 6;
        const int* yi = yidx.data() + (size_t)d.ytab_off * 6;
        for (int ox = 0; ox < zc; ++ox) {
            const int X0 = ((xb[ox] + d.src_x0) / RW) * RW;
            for (int j = 0; j < 6; ++j) {
                const int p = xi[(size_t)ox * 6 + j] + d.src_x0 - (X0 - kRegionHaloL);
                if (p < 0 || p >= RW + kRegionHaloL + kRegionHaloR) tap_range_ok = false;
            }
        }
        for (int oy = 0; oy < zr; ++oy) {
            const int Y0 = ((yb[oy] + d.src_y0) / RH) * RH;
            for (int j = 0; j < 6; ++j) {
                const int p = yi[(size_t)oy * 6 + j] + d.src_y0 - (Y0 - kRegionHaloT);
                if (p < 0 || p >= RH + kRegionHaloT + kRegionHaloB) tap_range_ok = false;
            }
        }
        if (d.out_h > d.zoom_h || d.out_w > d.zoom_w) zero_chunks += ((long long)d.out_h * d.out_w + 1023) / 1024;
    }
    tab.unit_tile_start[n_levels] = (int)unit_tiles;
    tab.unit_tiles_per_frame = (int)unit_tiles;
    tab.zero_chunk_start[n_levels] = (int)zero_chunks;
    tab.zero_chunks_per_frame = (int)zero_chunks;
    if (!tap_range_ok) {
        delete plan;
        return fail(ctx, SILENT_E_HIP, std::string(who) + ": internal error: a tap fell outside its staged region");
    }
    if (xreg.empty()) xreg.push_back(0);
    if (yreg.empty()) yreg.push_back(0);
    const std::vector<std::pair<const void*, size_t>> blobs = {
        {xidx.data(), xidx.size() * 4}, {xw.data(), xw.size() * 4},     {yidx.data(), yidx.size() * 4},
        {yw.data(), yw.size() * 4},     {xreg.data(), xreg.size() * 4}, {yreg.data(), yreg.size() * 4}};
    size_t total = 0;
    for (const auto& bl : blobs) total += align_up(bl.second);
    hipError_t e = hipMalloc(&plan->tables, total);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        delete plan;
        return fail(ctx, SILENT_E_NOMEM, std::string(who) + ": hipMalloc: " + hipGetErrorString(e));
    }
    const void* dptr[6];
    size_t off = 0;
    for (size_t i = 0; i < blobs.size(); ++i) {
        dptr[i] = (char*)plan->tables + off;
        e = hipMemcpy((void*)dptr[i], blobs[i].first, blobs[i].second, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(plan->tables);
            delete plan;
            return fail(ctx, SILENT_E_HIP, std::string(who) + ": hipMemcpy: " + hipGetErrorString(e));
        }
        off += align_up(blobs[i].second);
    }
    tab.xidx = (const int*)dptr[0];
    tab.xw = (const float*)dptr[1];
    tab.yidx = (const int*)dptr[2];
    tab.yw = (const float*)dptr[3];
    tab.xreg = (const int*)dptr[4];
    tab.yreg = (const int*)dptr[5];
    // ---- single-read stream path (see gray_stream_kernel): eligible when there is exactly one unit level and
    // every other level resamples the same crop with a step large enough for 4 vertical slots
    {
        int unit = -1, n_unit = 0;
        for (int l = 0; l < n_levels; ++l)
            if (tab.lv[l].kind == kPyrUnit) { unit = l; ++n_unit; }
        bool ok = channels == 1 && n_unit == 1 && tab.n_general >= 1 && tab.n_general <= 7;
        if (ok) {
            const PyrLevelDev& u = tab.lv[unit];
            ok = u.out_h >= u.src_h && u.out_w >= u.src_w;
            for (int l = 0; l < n_levels && ok; ++l) {
                const PyrLevelDev& d = tab.lv[l];
                if (d.kind != kPyrGeneral) continue;
                ok = d.src_y0 == u.src_y0 && d.src_x0 == u.src_x0 && d.src_h == u.src_h && d.src_w == u.src_w;
            }
        }
        if (ok) {
            const PyrLevelDev& u = tab.lv[unit];
            const int G = tab.n_general;
            const int tiles_y = (u.out_h + kFusedTH - 1) / kFusedTH;
            const int waves_x = ((u.out_w + kFusedTW - 1) / kFusedTW) * kFusedWaves;
            const int Gp = stream_pad_levels(G), PR = kStreamProgRow(Gp);
            const size_t n_rec = (size_t)tiles_y * kStreamRows;
            std::vector<int> prog(n_rec * PR, 0), hdr((size_t)G * waves_x * 2, 0), rec((size_t)G * waves_x * 64 * 8, 0);
            for (size_t r = 0; r < n_rec; ++r)
                for (int gg = 0; gg < Gp; ++gg) prog[r * PR + gg] = 7 << 4;  // inert: feeds nothing, no slot completes
            std::vector<char> used(n_rec * G * kStreamSlots, 0);
            int g = 0;
            for (int l = 0; l < n_levels && ok; ++l) {
                const PyrLevelDev& d = tab.lv[l];
                if (d.kind != kPyrGeneral) continue;
                const int zc = std::min(d.zoom_w, d.out_w), zr = std::min(d.zoom_h, d.out_h);
                const int* yb = ybase.data() + d.ytab_off;
                const int* xb = xbase.data() + d.xtab_off;
                for (int oy = 0; oy < zr && ok; ++oy) {
                    const int t = yb[oy] / kFusedTH;
                    if (yb[oy] < 0 || t >= tiles_y) { ok = false; break; }
                    const int slot = oy % stream_slots(g);
                    for (int j = 0; j < 6; ++j) {
                        const int i = yb[oy] - t * kFusedTH + 2 + j;  // stream row of tap j (a tile streams rows y0-4 ..)
                        if (i < 0 || i >= kStreamRows) { ok = false; break; }
                        const size_t r = (size_t)t * kStreamRows + i;
                        const size_t e = r * G + g;
                        if (used[e * kStreamSlots + slot]) { ok = false; break; }  // two live rows in one slot: step too small
                        used[e * kStreamSlots + slot] = 1;
                        int* pr = prog.data() + r * PR;
                        int& meta = pr[g];
                        std::memcpy(pr + stream_w_off(Gp, g) + slot, &yw[(size_t)(d.ytab_off + oy) * 6 + j], 4);
                        meta |= 128;  // this stream row feeds level g
                        if (j == 0) meta |= 1 << slot;
                        if (j == 5) {
                            if (((meta >> 4) & 7) != 7) { ok = false; break; }  // two rows completing together
                            meta = (meta & 0x8f) | (slot << 4) | (oy << 8);
                        }
                    }
                }
                int ox = 0;
                for (int wx = 0; wx < waves_x && ok; ++wx) {
                    const int xw0 = wx * kFusedCols;
                    while (ox < zc && xb[ox] < xw0) ++ox;
                    int n = 0;
                    while (ox + n < zc && xb[ox + n] < xw0 + kFusedCols) ++n;
                    if (n > 64) { ok = false; break; }
                    hdr[((size_t)g * waves_x + wx) * 2] = ox;
                    hdr[((size_t)g * waves_x + wx) * 2 + 1] = n;
                    for (int j = 0; j < n; ++j) {
                        int* r = rec.data() + (((size_t)g * waves_x + wx) * 64 + j) * 8;
                        r[0] = xb[ox + j] - xw0 + 2;  // lane holding tap 0 (lane 0 <-> column xw0 - 4)
                        if (r[0] < 0 || r[0] + 5 > 63) { ok = false; break; }
                        std::memcpy(r + 1, &xw[(size_t)(d.xtab_off + ox + j) * 6], 24);
                    }
                    ox += n;
                }
                plan->stream.px_off[g] = tab.px_off[l];
                plan->stream.out_w[g] = d.out_w;
                ++g;
            }
            if (ok) {
                const size_t b0 = align_up(prog.size() * 4), b1 = align_up(hdr.size() * 4), b2 = align_up(rec.size() * 4);
                hipError_t se = hipMalloc(&plan->stream_tables, b0 + b1 + b2);
                if (se == hipSuccess) se = hipMemcpy(plan->stream_tables, prog.data(), prog.size() * 4, hipMemcpyHostToDevice);
                if (se == hipSuccess) se = hipMemcpy((char*)plan->stream_tables + b0, hdr.data(), hdr.size() * 4, hipMemcpyHostToDevice);
                if (se == hipSuccess) se = hipMemcpy((char*)plan->stream_tables + b0 + b1, rec.data(), rec.size() * 4, hipMemcpyHostToDevice);
                if (se != hipSuccess) {
                    (void)hipGetLastError();
                    if (plan->stream_tables) (void)hipFree(plan->stream_tables);
                    plan->stream_tables = nullptr;
                } else {
                    plan->stream.G = G;
                    plan->stream.tiles_y = tiles_y;
                    plan->stream.waves_x = waves_x;
                    plan->stream.row_prog = (const int*)plan->stream_tables;
                    plan->stream.col_hdr = (const int*)((char*)plan->stream_tables + b0);
                    plan->stream.col_rec = (const int*)((char*)plan->stream_tables + b0 + b1);
                    plan->stream_unit_level = unit;
                    plan->stream_ok = true;
                }
            }
        }
    }
    // ---- walk plans of pyramid_walk3_kernel (3 channels).  Classic pyramid (one unit level whose crop every other level
    // resamples): one plan.  Anything else (the reference's nested centre crops): one plan per level -- a unit level alone, or a
    // general level alone on its own crop.  Per plan: a row program (one record per source row of the crop: "an output row of
    // level g completes here" + its 6 vertical weights) and column records per PX-pixel wave tile.
    if (channels == 3 && tab.W % 4 == 0 && n_levels <= 7 + kW3MaxPlans) {
        struct HostPlan {
            int unit;                 // level index of the plan's unit level, or -1
            std::vector<int> gen;     // its general levels
        };
        std::vector<HostPlan> hp;
        {
            int unit = -1, n_unit = 0;
            for (int l = 0; l < n_levels; ++l)
                if (tab.lv[l].kind == kPyrUnit) { unit = l; ++n_unit; }
            bool same_crop = n_unit == 1 && tab.n_general >= 1 && tab.n_general <= 7;
            if (same_crop) {
                const PyrLevelDev& u = tab.lv[unit];
                same_crop = u.out_h >= u.src_h && u.out_w >= u.src_w;
                for (int l = 0; l < n_levels && same_crop; ++l) {
                    const PyrLevelDev& d = tab.lv[l];
                    if (d.kind != kPyrGeneral) continue;
                    same_crop = d.src_y0 == u.src_y0 && d.src_x0 == u.src_x0 && d.src_h == u.src_h && d.src_w == u.src_w;
                }
            }
            if (same_crop) {
                HostPlan h{unit, {}};
                for (int l = 0; l < n_levels; ++l)
                    if (tab.lv[l].kind == kPyrGeneral) h.gen.push_back(l);
                hp.push_back(h);
            } else {
                for (int l = 0; l < n_levels; ++l) {
                    if (tab.lv[l].kind == kPyrUnit) hp.push_back(HostPlan{l, {}});
                    else if (tab.lv[l].kind == kPyrGeneral) hp.push_back(HostPlan{-1, {l}});
                }
            }
        }
        bool usable = !hp.empty() && (int)hp.size() <= kW3MaxPlans;
        for (const HostPlan& h : hp) {
            if (h.unit >= 0) {
                const PyrLevelDev& u = tab.lv[h.unit];   // canvas at least as large as the crop
                if (u.out_h < u.src_h || u.out_w < u.src_w) usable = false;
            }
            const PyrLevelDev& c = tab.lv[h.unit >= 0 ? h.unit : h.gen[0]];
            if (c.src_w < 8) usable = false;
        }
        int maxg = 0;
        for (const HostPlan& h : hp) maxg = std::max(maxg, (int)h.gen.size());
        const int Gp = stream_pad_levels(std::max(maxg, 1)), PR = w3_prog_row(Gp);
        for (int px : {36, 32}) {
            if (!usable || plan->walk_pyr_ok) break;
            const int rec_total = w3_rec_total(px, Gp);
            std::vector<int> blob;                        // all tables of all plans, offsets in ints
            struct Off { size_t prog, hdr, rec; };
            std::vector<Off> offs;
            bool ok = true;
            Walk3Args wa;
            std::memset(&wa, 0, sizeof(wa));
            for (size_t pi = 0; pi < hp.size() && ok; ++pi) {
                const HostPlan& h = hp[pi];
                const PyrLevelDev& c = tab.lv[h.unit >= 0 ? h.unit : h.gen[0]];
                const int walk_h = h.unit >= 0 ? c.out_h : c.src_h, walk_w = h.unit >= 0 ? c.out_w : c.src_w;
                const int G = (int)h.gen.size();
                const int waves_x = ((walk_w + kW3NC * px - 1) / (kW3NC * px)) * kW3NC;
                const size_t n_rec = (size_t)walk_h + 8;                 // stream rows y = -4 .. walk_h + 3 at index y + 4
                const size_t n_rec_pad = n_rec + 2 * kWalkCH;            // the loader fetches whole chunks of records
                std::vector<int> prog(n_rec_pad * PR, 0), hdr((size_t)std::max(G, 1) * waves_x * 2, 0), rec((size_t)waves_x * rec_total * 8, 0);
                Walk3Plan& wp3 = wa.plan[pi];
                for (int g = 0; g < G && ok; ++g) {
                    const PyrLevelDev& d = tab.lv[h.gen[g]];
                    const int zc = std::min(d.zoom_w, d.out_w), zr = std::min(d.zoom_h, d.out_h);
                    const int* yb = ybase.data() + d.ytab_off;
                    const int* xb = xbase.data() + d.xtab_off;
                    for (int oy = 0; oy < zr && ok; ++oy) {
                        // one record entry per COMPLETING row: flag + output row, 6 weights
                        if (yb[oy] < 0 || yb[oy] >= walk_h) { ok = false; break; }
                        const size_t r = (size_t)(yb[oy] + 7);           // the last tap sits on stream row y = yb + 3, index y + 4
                        if (r >= n_rec) { ok = false; break; }
                        int* pr = prog.data() + r * PR;
                        if (pr[g] & 1) { ok = false; break; }             // two rows of one level completing together: step < 1
                        pr[g] = 1 | (oy << 8);
                        std::memcpy(pr + Gp + 6 * g, &yw[(size_t)(d.ytab_off + oy) * 6], 24);
                    }
                    int ox = 0;
                    for (int wx = 0; wx < waves_x && ok; ++wx) {
                        const int xw0 = wx * px;
                        while (ox < zc && xb[ox] < xw0) ++ox;
                        int n = 0;
                        while (ox + n < zc && xb[ox + n] < xw0 + px) ++n;
                        if (n > w3_rec_cap(px, g)) { ok = false; break; }  // outputs per wave tile (the gather takes <= 21)
                        hdr[((size_t)g * waves_x + wx) * 2] = ox;
                        hdr[((size_t)g * waves_x + wx) * 2 + 1] = n;
                        for (int j = 0; j < n; ++j) {
                            int* r = rec.data() + ((size_t)wx * rec_total + w3_rec_base(px, g) + j) * 8;
                            r[0] = (xb[ox + j] - xw0) * 3;               // FLOAT index of tap 0, channel 0 (line starts at pixel xw0 - 2)
                            if (r[0] < 0 || r[0] + 2 + 15 > kW3TileF - 1) { ok = false; break; }
                            std::memcpy(r + 1, &xw[(size_t)(d.xtab_off + ox + j) * 6], 24);
                        }
                        ox += n;
                    }
                    wp3.pyr.px_off[g] = tab.px_off[h.gen[g]];
                    wp3.pyr.out_w[g] = d.out_w;
                }
                if (!ok) break;
                wp3.src_y0 = c.src_y0; wp3.src_x0 = c.src_x0; wp3.src_h = c.src_h; wp3.src_w = c.src_w;
                wp3.shift = (c.src_x0 * 3) % 4;
                wp3.has_unit = h.unit >= 0 ? 1 : 0;
                wp3.out_h = walk_h; wp3.out_w = walk_w;
                wp3.eff_h = h.unit >= 0 ? std::min(c.zoom_h, c.out_h) : walk_h;
                wp3.eff_w = h.unit >= 0 ? std::min(c.zoom_w, c.out_w) : walk_w;
                wp3.px_off = h.unit >= 0 ? tab.px_off[h.unit] : 0;
                wp3.pyr.G = G;
                auto put = [&](const std::vector<int>& v) {
                    while (blob.size() % 64) blob.push_back(0);          // 256-byte aligned tables
                    const size_t at = blob.size();
                    blob.insert(blob.end(), v.begin(), v.end());
                    return at;
                };
                Off o;
                o.prog = put(prog);
                o.hdr = put(hdr);
                o.rec = put(rec);
                offs.push_back(o);
            }
            if (!ok) continue;
            hipError_t se = hipMalloc(&plan->walk_tables, blob.size() * 4);
            if (se == hipSuccess) se = hipMemcpy(plan->walk_tables, blob.data(), blob.size() * 4, hipMemcpyHostToDevice);
            if (se != hipSuccess) {
                (void)hipGetLastError();
                if (plan->walk_tables) (void)hipFree(plan->walk_tables);
                plan->walk_tables = nullptr;
                break;
            }
            const int* base = (const int*)plan->walk_tables;
            for (size_t pi = 0; pi < hp.size(); ++pi) {
                wa.plan[pi].pyr.row_prog = base + offs[pi].prog;
                wa.plan[pi].pyr.col_hdr = base + offs[pi].hdr;
                wa.plan[pi].pyr.col_rec = base + offs[pi].rec;
            }
            wa.H = tab.H;
            wa.W = tab.W;
            wa.n_plans = (int)hp.size();
            wa.frame_px = tab.frame_px_out;
            for (int j = 0; j < 5; ++j) wa.wx[j] = plan->unit_w[j];
            plan->walk = wa;
            plan->walk_px = px;
            plan->walk_G = Gp;
            plan->walk_pyr_ok = true;
        }
    }
    *out = plan;
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_pyramid_plan_create");
}

SILENT_EXPORT void silent_pyramid_plan_destroy(silent_pyramid_plan* plan) try {
    if (!plan) return;
    DeviceGuard guard(plan->ctx ? plan->ctx->device : 0);
    if (plan->tables) (void)hipFree(plan->tables);
    if (plan->stream_tables) (void)hipFree(plan->stream_tables);
    if (plan->walk_tables) (void)hipFree(plan->walk_tables);
    delete plan;
} catch (...) {
}

template <int G, int PX>
static int walk3_blocks_per_cu() {
    int per_cu = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pyramid_walk3_kernel<G, PX>, kW3Threads, 0);
    (void)hipGetLastError();
    return std::max(per_cu, 1);
}

// Per-launch decomposition of the plan's walks (any batch size): strips of 4 x PX pixels, and per walk the segment height
// that minimises ceil(blocks / resident blocks) x (segment rows + 8 halo rows) row steps (896 blocks on a chip that holds 768
// run TWO rounds: measured 1.07 ms against 0.66 ms for 5 segments per frame).
static bool walk3_plan(const silent_ctx* ctx, const silent_pyramid_plan* plan, int n_frames, Walk3Args* wa) {
    if (plan->tab.C != 3 || !plan->walk_pyr_ok) return false;
    *wa = plan->walk;
    const int px = plan->walk_px;
    int per_cu;
    if (px == 36) per_cu = plan->walk_G <= 4 ? walk3_blocks_per_cu<4, 36>() : walk3_blocks_per_cu<7, 36>();
    else per_cu = plan->walk_G <= 4 ? walk3_blocks_per_cu<4, 32>() : walk3_blocks_per_cu<7, 32>();
    const long long resident = (long long)per_cu * ctx->n_cus;
    long long block0 = 0;
    for (int pi = 0; pi < wa->n_plans; ++pi) {
        Walk3Plan& w = wa->plan[pi];
        w.strips_x = (w.out_w + kW3NC * px - 1) / (kW3NC * px);
        const long long per_seg = (long long)n_frames * w.strips_x;
        const int max_segs = std::max(1, w.out_h / 32);
        long long best_cost = -1;
        int seg_rows = w.out_h;
        for (int segs = 1; segs <= max_segs; ++segs) {
            int rows = (w.out_h + segs - 1) / segs;
            rows = (rows + kWalkCH - 1) / kWalkCH * kWalkCH;
            const long long n_seg = (w.out_h + rows - 1) / rows;
            const long long cost = ((per_seg * n_seg + resident - 1) / resident) * (rows + 8);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                seg_rows = rows;
            }
        }
        w.seg_rows = seg_rows;
        w.segs_y = (w.out_h + seg_rows - 1) / seg_rows;
        w.block0 = (int)block0;
        block0 += (long long)w.strips_x * w.segs_y;
    }
    if (block0 * n_frames > 0x7fffffffll) return false;
    wa->blocks_per_frame = (int)block0;
    return true;
}

static int launch_pyramid(silent_ctx* ctx, const char* who, const silent_pyramid_plan* plan, const float* frames,
                          int n_frames, float* pyr, hipStream_t s, bool with_unit, bool with_region = true) {
    if (!plan || !frames || !pyr) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (plan->ctx != ctx) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": plan belongs to another context");
    if (n_frames < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": n_frames must be >= 1");
    const PyrTab& tab = plan->tab;
    const long long b_unit = with_unit ? (long long)tab.unit_tiles_per_frame * n_frames : 0;
    const long long b_region = (with_region && tab.n_general) ? (long long)tab.regions_x * tab.regions_y * n_frames : 0;
    const long long b_zero = (long long)tab.zero_chunks_per_frame * n_frames;
    if (b_unit > 0x7fffffffll || b_region > 0x7fffffffll || b_zero > 0x7fffffffll)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": too many tiles for one launch");
    const unsigned kopts = ctx->tune[SILENT_TUNE_PYRAMID];  // 1: no stream kernel
    Walk3Args w3t;
    if (plan->stream_ok && with_unit && with_region && !(kopts & 1u)) {
        // single-read pyramid: frame -> every level in one kernel (pyramid_stream_kernel; single-channel plans only:
        // on interleaved RGB the stride-3 accesses of the same kernel made it 1.5x SLOWER than unit + region kernels)
        const PyrLevelDev& d = tab.lv[plan->stream_unit_level];
        FusedTab ft;
        std::memset(&ft, 0, sizeof(ft));
        ft.n = 1;
        for (int j = 0; j < 5; ++j) ft.wx[j] = ft.wy[j] = plan->unit_w[j];
        FusedLevel& f = ft.lv[0];
        f.src_y0 = d.src_y0; f.src_x0 = d.src_x0; f.src_h = d.src_h; f.src_w = d.src_w;
        f.zoom_h = d.zoom_h; f.zoom_w = d.zoom_w; f.out_h = d.out_h; f.out_w = d.out_w;
        f.tiles_x = (d.out_w + kFusedTW - 1) / kFusedTW;
        f.px_off = tab.px_off[plan->stream_unit_level];
        ft.tiles_per_frame = f.tiles_x * ((d.out_h + kFusedTH - 1) / kFusedTH);
        ft.H = tab.H;
        ft.W = tab.W;
        ft.frame_px = tab.frame_px_out;
        const long long blocks = (long long)ft.tiles_per_frame * n_frames;
        if (blocks > 0x7fffffffll) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": too many tiles for one launch");
#define PYR_STREAM(G_) \
    hipLaunchKernelGGL((pyramid_stream_kernel<1, G_>), dim3((unsigned)blocks), dim3(64 * kFusedWaves), 0, s, frames, pyr, ft, plan->stream)
        if (plan->stream.G <= 4) PYR_STREAM(4);
        else PYR_STREAM(7);
#undef PYR_STREAM
    } else if (tab.C == 3 && plan->walk_pyr_ok && with_unit && with_region && !(kopts & 3u) && walk3_plan(ctx, plan, n_frames, &w3t)) {
        // single-read RGB pyramid (pyramid_walk3_kernel, silent_walk_rgb.h); PYRAMID knob bits 1 / 2: unit + region kernels
        const long long wblocks = (long long)n_frames * w3t.blocks_per_frame;
#define WALK3(G_, PX_) hipLaunchKernelGGL((pyramid_walk3_kernel<G_, PX_>), dim3((unsigned)wblocks), dim3(kW3Threads), 0, s, frames, pyr, w3t)
        if (plan->walk_px == 36) {
            if (plan->walk_G <= 4) WALK3(4, 36);
            else WALK3(7, 36);
        } else {
            if (plan->walk_G <= 4) WALK3(4, 32);
            else WALK3(7, 32);
        }
#undef WALK3
    } else if (tab.C == 1) {
        if (b_unit) hipLaunchKernelGGL(pyramid_unit_kernel<1>, dim3((unsigned)b_unit), dim3(256), 0, s, frames, pyr, tab);
        if (b_region) hipLaunchKernelGGL(pyramid_region_kernel<1>, dim3((unsigned)b_region), dim3(256), 0, s, frames, pyr, tab);
    } else {
        if (b_unit) hipLaunchKernelGGL(pyramid_unit_kernel<3>, dim3((unsigned)b_unit), dim3(256), 0, s, frames, pyr, tab);
        if (b_region) hipLaunchKernelGGL(pyramid_region_kernel<3>, dim3((unsigned)b_region), dim3(256), 0, s, frames, pyr, tab);
    }
    if (b_zero) hipLaunchKernelGGL(pyramid_zero_kernel, dim3((unsigned)b_zero), dim3(256), 0, s, pyr, tab);
    return check_launch(ctx, who);
}

SILENT_EXPORT int silent_pyramid_dev(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames,
                                     int n_frames, float* pyr, silent_stream stream) try {
    NEED_CTX(ctx);
    return launch_pyramid(ctx, "silent_pyramid", plan, frames, n_frames, pyr, (hipStream_t)stream, true);
} catch (...) {
    return on_exception(ctx, "silent_pyramid_dev");
}

// ------------------------------------------------------------------------------------------ whole gray pass

SILENT_EXPORT int silent_pyramid_plan_is_streamable(const silent_pyramid_plan* plan) try {
    return plan && plan->stream_ok ? 1 : 0;
} catch (...) {
    return on_exception(nullptr, "silent_pyramid_plan_is_streamable");
}

SILENT_EXPORT int silent_pyramid_plan_walk_plans(const silent_pyramid_plan* plan, int* pixels_per_wave) try {
    if (pixels_per_wave) *pixels_per_wave = plan && plan->walk_pyr_ok ? plan->walk_px : 0;
    return plan && plan->walk_pyr_ok ? plan->walk.n_plans : 0;
} catch (...) {
    return on_exception(nullptr, "silent_pyramid_plan_walk_plans");
}

SILENT_EXPORT int silent_set_profiling(silent_ctx* ctx, int enable) try {
    NEED_CTX(ctx);
    if (enable < 0) return fail(ctx, SILENT_E_INVALID, "silent_set_profiling: enable must be >= 0");
    if (enable && !ctx->prof_ev[0][0])
        for (auto& pr : ctx->prof_ev)
            for (hipEvent_t& e : pr) HIP_TRY(ctx, hipEventCreate(&e));
    ctx->profiling = enable != 0;
    ctx->prof_period = enable > 0 ? enable : 1;
    ctx->prof_calls = ctx->prof_recorded = 0;
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_set_profiling");
}

SILENT_EXPORT int silent_profile_elapsed_ms(silent_ctx* ctx, float* ms, int64_t* pixels) try {
    NEED_CTX(ctx);
    if (!ms) return fail(ctx, SILENT_E_INVALID, "silent_profile_elapsed_ms: ms is NULL");
    if (!ctx->prof_recorded) return fail(ctx, SILENT_E_INVALID, "silent_profile_elapsed_ms: no profiled launch recorded");
    const int n = std::min(ctx->prof_recorded, silent_ctx::kProfPairs);
    double sum = 0.0;
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        HIP_TRY(ctx, hipEventSynchronize(ctx->prof_ev[i][1]));
        HIP_TRY(ctx, hipEventElapsedTime(&t, ctx->prof_ev[i][0], ctx->prof_ev[i][1]));
        sum += t;
    }
    *ms = (float)(sum / n);
    if (pixels) *pixels = ctx->prof_pixels;
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_profile_elapsed_ms");
}

// parts: bit 0 = the pyramid of every level + CS / end of the unit levels (steps 1 and 2), bit 1 = CS + end of the remaining levels
// (step 3, which reads the pyramid steps 1 and 2 wrote)
static int gray_pass_parts(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames, int n_frames,
                           const float* cs_kernel, const float* end_bank, int n_orient, float clip_hi, float* pyr,
                           float* cs_out, float* end_out, unsigned parts, silent_stream stream) {
    const char* who = "silent_gray_pass";
    if (!plan || !frames || !pyr || !cs_kernel) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    if (plan->ctx != ctx) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": plan belongs to another context");
    if (plan->tab.C != 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": the plan must be single-channel");
    if (!cs_out && !end_out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": both outputs are NULL");
    if (end_out && !end_bank) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": end_bank is NULL");
    if (n_orient != 3 && n_orient != 4 && n_orient != 8)
        return fail(ctx, SILENT_E_UNSUPPORTED, std::string(who) + ": n_orient must be 3, 4 or 8");
    if (n_frames < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": n_frames must be >= 1");
    hipStream_t s = (hipStream_t)stream;
    const PyrTab& pt = plan->tab;
    const int kopts = (int)ctx->tune[SILENT_TUNE_GRAY];  // A/B knobs: bit3 32-row fused tiles, bit4 disable the stream path
    const bool stream_path = plan->stream_ok && !(kopts & 16);
    // 1. non-unit levels of the pyramid: by the region kernel, unless the stream kernel of step 2 produces them
    //    from the same single read of the frame; plus the zero fill of canvases larger than their zoomed crop
    if (!(parts & 3u)) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": parts must name step 1 + 2 (bit 0) and / or step 3 (bit 1)");
    if (parts & 1u) TRY(launch_pyramid(ctx, who, plan, frames, n_frames, pyr, s, false, !stream_path));
    // 2. unit levels: pyramid + CS + end in one kernel
    const int fth = (!stream_path && (kopts & 8)) ? 32 : kFusedTH;
    FusedTab ft;
    std::memset(&ft, 0, sizeof(ft));
    bool is_unit[kMaxLevels] = {false};
    long long tiles = 0, unit_px = 0;
    for (int l = 0; l < pt.n_levels; ++l) {
        const PyrLevelDev& d = pt.lv[l];
        if (d.kind != kPyrUnit) continue;
        is_unit[l] = true;
        if (ft.n == 0)
            for (int j = 0; j < 5; ++j) {  // every unit level has the same [1,26,66,26,1]/120 taps
                ft.wx[j] = plan->unit_w[j];
                ft.wy[j] = plan->unit_w[j];
            }
        FusedLevel& f = ft.lv[ft.n++];
        f.src_y0 = d.src_y0; f.src_x0 = d.src_x0; f.src_h = d.src_h; f.src_w = d.src_w;
        f.zoom_h = d.zoom_h; f.zoom_w = d.zoom_w; f.out_h = d.out_h; f.out_w = d.out_w;
        f.tiles_x = (d.out_w + kFusedTW - 1) / kFusedTW;
        f.tile_start = (int)tiles;
        f.px_off = pt.px_off[l];
        tiles += (long long)f.tiles_x * ((d.out_h + fth - 1) / fth);
        unit_px += (long long)d.out_h * d.out_w;
    }
    ft.tiles_per_frame = (int)tiles;
    ft.H = pt.H;
    ft.W = pt.W;
    ft.frame_px = pt.frame_px_out;
    const long long blocks = tiles * n_frames;
    if (blocks > 0x7fffffffll) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": too many tiles for one launch");
    if (blocks && (parts & 1u)) {
        GrayW w;
        std::memset(&w, 0, sizeof(w));
        std::memcpy(w.cs, cs_kernel, sizeof(float) * 9);
        if (end_bank) std::memcpy(w.end, end_bank, sizeof(float) * 9 * n_orient);
        ctx->prof_sample = ctx->profiling && (ctx->prof_calls++ % ctx->prof_period) == 0;
        const int prof_slot = ctx->prof_recorded % silent_ctx::kProfPairs;
        if (ctx->prof_sample) HIP_TRY(ctx, hipEventRecord(ctx->prof_ev[prof_slot][0], s));
        if (stream_path) {
            const StreamTab& st = plan->stream;
#define STREAM_LAUNCH(K_, G_) \
    hipLaunchKernelGGL((gray_stream_kernel<K_, G_>), dim3((unsigned)blocks), dim3(64 * kFusedWaves), 0, s, frames, pyr, cs_out, end_out, ft, st, w, clip_hi, (unsigned)((kopts >> 5) & 1))
            if (st.G <= 4) {
                if (n_orient == 3) STREAM_LAUNCH(3, 4);
                else if (n_orient == 4) STREAM_LAUNCH(4, 4);
                else STREAM_LAUNCH(8, 4);
            } else {
                if (n_orient == 3) STREAM_LAUNCH(3, 7);
                else if (n_orient == 4) STREAM_LAUNCH(4, 7);
                else STREAM_LAUNCH(8, 7);
            }
#undef STREAM_LAUNCH
        } else {
#define FUSED_LAUNCH(K_, R_) \
    hipLaunchKernelGGL((gray_unit_fused_kernel<K_, R_>), dim3((unsigned)blocks), dim3(64 * kFusedWaves), 0, s, frames, pyr, cs_out, end_out, ft, w, clip_hi)
            if (fth == 32) {
                if (n_orient == 3) FUSED_LAUNCH(3, 32);
                else if (n_orient == 4) FUSED_LAUNCH(4, 32);
                else FUSED_LAUNCH(8, 32);
            } else {
                if (n_orient == 3) FUSED_LAUNCH(3, kFusedTH);
                else if (n_orient == 4) FUSED_LAUNCH(4, kFusedTH);
                else FUSED_LAUNCH(8, kFusedTH);
            }
#undef FUSED_LAUNCH
        }
        if (ctx->prof_sample) {
            HIP_TRY(ctx, hipEventRecord(ctx->prof_ev[prof_slot][1], s));
            ++ctx->prof_recorded;
            ctx->prof_pixels = unit_px * n_frames;
        }
        TRY(check_launch(ctx, who));
    }
    // 3. CS + end on the remaining levels (they read the pyramid written in step 1)
    if (pt.n_general && (parts & 2u))
        TRY(launch_gray(ctx, who, pyr, plan->extents.data(), pt.n_levels, n_frames, cs_kernel, end_bank, n_orient,
                        clip_hi, cs_out, end_out, s, is_unit));
    return SILENT_OK;
}

SILENT_EXPORT int silent_gray_pass_dev(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames,
                                       int n_frames, const float* cs_kernel, const float* end_bank, int n_orient,
                                       float clip_hi, float* pyr, float* cs_out, float* end_out,
                                       silent_stream stream) try {
    NEED_CTX(ctx);
    return gray_pass_parts(ctx, plan, frames, n_frames, cs_kernel, end_bank, n_orient, clip_hi, pyr, cs_out, end_out, 3u, stream);
} catch (...) {
    return on_exception(ctx, "silent_gray_pass_dev");
}

SILENT_EXPORT int silent_gray_pass_parts_dev(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames,
                                             int n_frames, const float* cs_kernel, const float* end_bank, int n_orient,
                                             float clip_hi, float* pyr, float* cs_out, float* end_out, unsigned parts,
                                             silent_stream stream) try {
    NEED_CTX(ctx);
    return gray_pass_parts(ctx, plan, frames, n_frames, cs_kernel, end_bank, n_orient, clip_hi, pyr, cs_out, end_out, parts, stream);
} catch (...) {
    return on_exception(ctx, "silent_gray_pass_parts_dev");
}

// ------------------------------------------------------------------------------------------ host-pointer twins
// Synchronous: stage inputs into the context arena, run the *_dev twin on the default stream, copy back.

struct Stage {
    silent_ctx* ctx;
    size_t used = 0;
    std::vector<size_t> offs;
    explicit Stage(silent_ctx* c) : ctx(c) {}
    size_t add(size_t bytes) {
        offs.push_back(used);
        used += align_up(bytes ? bytes : 1);
        return offs.size() - 1;
    }
    int commit() { return grow(ctx, ctx->arena, used); }
    template <class T>
    T* ptr(size_t i) const { return (T*)((char*)ctx->arena.p + offs[i]); }
};

static int h2d(silent_ctx* ctx, void* d, const void* h, size_t bytes) {
    HIP_TRY(ctx, hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return SILENT_OK;
}
static int d2h(silent_ctx* ctx, void* h, const void* d, size_t bytes) {
    HIP_TRY(ctx, hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    return SILENT_OK;
}
static int sync0(silent_ctx* ctx) {
    HIP_TRY(ctx, hipStreamSynchronize(nullptr));
    return SILENT_OK;
}

static int check_levels(silent_ctx* ctx, const char* who, const silent_extent* levels, int n_levels, int n_frames,
                        long long* px) {
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    *px = tab.frame_px * n_frames;
    return SILENT_OK;
}

SILENT_EXPORT int silent_conv2d_same(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                     int n_frames, int c_in, const float* k, int kh, int kw, int c_out, unsigned flags,
                                     float clip_hi, float* out) try {
    NEED_CTX(ctx);
    if (!in || !out || !k) return fail(ctx, SILENT_E_INVALID, "silent_conv2d_same: NULL pointer");
    if (c_in < 1 || c_out < 1) return fail(ctx, SILENT_E_INVALID, "silent_conv2d_same: channels must be >= 1");
    long long px;
    TRY(check_levels(ctx, "silent_conv2d_same", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t bi = (size_t)px * c_in * 4, bo = (size_t)px * c_out * 4;
    const size_t i_in = st.add(bi), i_out = st.add(bo);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), in, bi));
    TRY(silent_conv2d_same_dev(ctx, st.ptr<float>(i_in), levels, n_levels, n_frames, c_in, k, kh, kw, c_out, flags,
                               clip_hi, st.ptr<float>(i_out), nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, out, st.ptr<float>(i_out), bo);
} catch (...) {
    return on_exception(ctx, "silent_conv2d_same");
}

SILENT_EXPORT int silent_regulate(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                  int n_frames, int channels, const float* blur, int kh, int kw, float rv, float root,
                                  int flat_policy, float* out) try {
    NEED_CTX(ctx);
    if (!in || !out || !blur) return fail(ctx, SILENT_E_INVALID, "silent_regulate: NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, "silent_regulate: channels must be >= 1");
    long long px;
    TRY(check_levels(ctx, "silent_regulate", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t b = (size_t)px * channels * 4;
    const size_t i_in = st.add(b), i_out = st.add(b);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), in, b));
    TRY(silent_regulate_dev(ctx, st.ptr<float>(i_in), levels, n_levels, n_frames, channels, blur, kh, kw, rv, root,
                            flat_policy, st.ptr<float>(i_out), nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, out, st.ptr<float>(i_out), b);
} catch (...) {
    return on_exception(ctx, "silent_regulate");
}

SILENT_EXPORT int silent_gray_line_end(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels,
                                       int n_frames, const float* cs_kernel, const float* end_bank, int n_orient,
                                       float clip_hi, float* cs_out, float* end_out) try {
    NEED_CTX(ctx);
    if (!pyr) return fail(ctx, SILENT_E_INVALID, "silent_gray_line_end: NULL pointer");
    if (n_orient < 1 || n_orient > 8) return fail(ctx, SILENT_E_UNSUPPORTED, "silent_gray_line_end: n_orient must be 3, 4 or 8");
    long long px;
    TRY(check_levels(ctx, "silent_gray_line_end", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t b1 = (size_t)px * 4, bk = (size_t)px * n_orient * 4;
    const size_t i_in = st.add(b1), i_cs = st.add(b1), i_end = st.add(bk);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), pyr, b1));
    TRY(silent_gray_line_end_dev(ctx, st.ptr<float>(i_in), levels, n_levels, n_frames, cs_kernel, end_bank, n_orient,
                                 clip_hi, cs_out ? st.ptr<float>(i_cs) : nullptr,
                                 end_out ? st.ptr<float>(i_end) : nullptr, nullptr));
    TRY(sync0(ctx));
    if (cs_out) TRY(d2h(ctx, cs_out, st.ptr<float>(i_cs), b1));
    if (end_out) TRY(d2h(ctx, end_out, st.ptr<float>(i_end), bk));
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_gray_line_end");
}

SILENT_EXPORT int silent_pad_inwards(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                     int n_frames, int channels, int pt, int pb, int pl, int pr, float* out) try {
    NEED_CTX(ctx);
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, "silent_pad_inwards: NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, "silent_pad_inwards: channels must be >= 1");
    long long px;
    TRY(check_levels(ctx, "silent_pad_inwards", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t b = (size_t)px * channels * 4;
    const size_t i_in = st.add(b), i_out = st.add(b);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), in, b));
    TRY(silent_pad_inwards_dev(ctx, st.ptr<float>(i_in), levels, n_levels, n_frames, channels, pt, pb, pl, pr,
                               st.ptr<float>(i_out), nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, out, st.ptr<float>(i_out), b);
} catch (...) {
    return on_exception(ctx, "silent_pad_inwards");
}

SILENT_EXPORT int silent_value_from_color(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                          int n_frames, int channels, float* out) try {
    NEED_CTX(ctx);
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, "silent_value_from_color: NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, "silent_value_from_color: channels must be >= 1");
    long long px;
    TRY(check_levels(ctx, "silent_value_from_color", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t bi = (size_t)px * channels * 4, bo = (size_t)px * 4;
    const size_t i_in = st.add(bi), i_out = st.add(bo);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), in, bi));
    TRY(silent_value_from_color_dev(ctx, st.ptr<float>(i_in), levels, n_levels, n_frames, channels,
                                    st.ptr<float>(i_out), nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, out, st.ptr<float>(i_out), bo);
} catch (...) {
    return on_exception(ctx, "silent_value_from_color");
}

SILENT_EXPORT int silent_bw_from_color(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                          int n_frames, int channels, float* out) try {
    NEED_CTX(ctx);
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, "silent_bw_from_color: NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, "silent_bw_from_color: channels must be >= 1");
    long long px;
    TRY(check_levels(ctx, "silent_bw_from_color", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t bi = (size_t)px * channels * 4, bo = (size_t)px * 4;
    const size_t i_in = st.add(bi), i_out = st.add(bo);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), in, bi));
    TRY(silent_bw_from_color_dev(ctx, st.ptr<float>(i_in), levels, n_levels, n_frames, channels,
                                    st.ptr<float>(i_out), nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, out, st.ptr<float>(i_out), bo);
} catch (...) {
    return on_exception(ctx, "silent_bw_from_color");
}

SILENT_EXPORT int silent_nms3x3(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                int n_frames, int channels, int mode, float* out) try {
    NEED_CTX(ctx);
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, "silent_nms3x3: NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, "silent_nms3x3: channels must be >= 1");
    long long px;
    TRY(check_levels(ctx, "silent_nms3x3", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t b = (size_t)px * channels * 4;
    const size_t i_in = st.add(b), i_out = st.add(b);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), in, b));
    TRY(silent_nms3x3_dev(ctx, st.ptr<float>(i_in), levels, n_levels, n_frames, channels, mode, st.ptr<float>(i_out),
                          nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, out, st.ptr<float>(i_out), b);
} catch (...) {
    return on_exception(ctx, "silent_nms3x3");
}

SILENT_EXPORT int silent_top_value_points(silent_ctx* ctx, const float* color, const float* value,
                                          const silent_extent* levels, int n_levels, int n_frames, int channels,
                                          double top_percent, float* out) try {
    NEED_CTX(ctx);
    if (!color || !out) return fail(ctx, SILENT_E_INVALID, "silent_top_value_points: NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, "silent_top_value_points: channels must be >= 1");
    long long px;
    TRY(check_levels(ctx, "silent_top_value_points", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t bc = (size_t)px * channels * 4, bv = (size_t)px * 4;
    const size_t i_c = st.add(bc), i_v = st.add(bv), i_o = st.add(bc);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_c), color, bc));
    if (value) TRY(h2d(ctx, st.ptr<float>(i_v), value, bv));
    TRY(silent_top_value_points_dev(ctx, st.ptr<float>(i_c), value ? st.ptr<float>(i_v) : nullptr, levels, n_levels,
                                    n_frames, channels, top_percent, st.ptr<float>(i_o), nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, out, st.ptr<float>(i_o), bc);
} catch (...) {
    return on_exception(ctx, "silent_top_value_points");
}

SILENT_EXPORT int silent_max_value_indices_region(silent_ctx* ctx, const float* value, const silent_extent* levels,
                                                  int n_levels, int n_frames, const silent_extent* regions,
                                                  int64_t* idx, size_t cap_per_frame, int64_t* counts) try {
    NEED_CTX(ctx);
    if (!value || !counts) return fail(ctx, SILENT_E_INVALID, "silent_max_value_indices_region: NULL pointer");
    long long px;
    TRY(check_levels(ctx, "silent_max_value_indices_region", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t bv = (size_t)px * 4, bi = (size_t)n_frames * cap_per_frame * 4 * sizeof(int64_t);
    const size_t bc = (size_t)n_frames * sizeof(int64_t);
    const size_t i_v = st.add(bv), i_i = st.add(bi), i_c = st.add(bc);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_v), value, bv));
    TRY(silent_max_value_indices_region_dev(ctx, st.ptr<float>(i_v), levels, n_levels, n_frames, regions,
                                            st.ptr<int64_t>(i_i), cap_per_frame, st.ptr<int64_t>(i_c), nullptr));
    TRY(sync0(ctx));
    TRY(d2h(ctx, counts, st.ptr<int64_t>(i_c), bc));
    bool over = false;
    for (int f = 0; f < n_frames; ++f) {
        const size_t n = (size_t)std::min<int64_t>(counts[f], (int64_t)cap_per_frame);
        if (counts[f] > (int64_t)cap_per_frame) over = true;
        if (n) TRY(d2h(ctx, idx + (size_t)f * cap_per_frame * 4, st.ptr<int64_t>(i_i) + (size_t)f * cap_per_frame * 4, n * 4 * sizeof(int64_t)));
    }
    if (over) return fail(ctx, SILENT_E_CAPACITY, "silent_max_value_indices_region: cap_per_frame too small; counts hold the need");
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_max_value_indices_region");
}

SILENT_EXPORT int silent_rgb_line_end(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels,
                                      int n_frames, const silent_rgb_chain_params* p, float* orient_out,
                                      float* line_end_out, float* value_out) try {
    NEED_CTX(ctx);
    if (!pyr || !p) return fail(ctx, SILENT_E_INVALID, "silent_rgb_line_end: NULL pointer");
    long long px;
    TRY(check_levels(ctx, "silent_rgb_line_end", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t b3 = (size_t)px * 3 * 4, b1 = (size_t)px * 4;
    const size_t i_in = st.add(b3), i_o = st.add(b3), i_l = st.add(b3), i_v = st.add(b1);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), pyr, b3));
    TRY(silent_rgb_line_end_dev(ctx, st.ptr<float>(i_in), levels, n_levels, n_frames, p,
                                orient_out ? st.ptr<float>(i_o) : nullptr, line_end_out ? st.ptr<float>(i_l) : nullptr,
                                value_out ? st.ptr<float>(i_v) : nullptr, nullptr));
    TRY(sync0(ctx));
    if (orient_out) TRY(d2h(ctx, orient_out, st.ptr<float>(i_o), b3));
    if (line_end_out) TRY(d2h(ctx, line_end_out, st.ptr<float>(i_l), b3));
    if (value_out) TRY(d2h(ctx, value_out, st.ptr<float>(i_v), b1));
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_rgb_line_end");
}

SILENT_EXPORT int silent_rgb_keypoints(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels,
                                       int n_frames, const silent_rgb_chain_params* p, double top_percent,
                                       const silent_extent* regions, float* orient_out, float* line_end_out, float* value_out,
                                       float* peak_value_out, int64_t* idx, size_t cap_per_frame, int64_t* counts) try {
    NEED_CTX(ctx);
    if (!pyr || !p || !regions || !counts || (!idx && cap_per_frame))
        return fail(ctx, SILENT_E_INVALID, "silent_rgb_keypoints: NULL pointer");
    long long px;
    TRY(check_levels(ctx, "silent_rgb_keypoints", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t b3 = (size_t)px * 3 * 4, b1 = (size_t)px * 4;
    const size_t bi = (size_t)n_frames * cap_per_frame * 4 * sizeof(int64_t), bn = (size_t)n_frames * sizeof(int64_t);
    const size_t i_in = st.add(b3), i_o = st.add(b3), i_l = st.add(b3), i_v = st.add(b1), i_p = st.add(b1), i_i = st.add(bi),
                 i_n = st.add(bn);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), pyr, b3));
    TRY(silent_rgb_keypoints_dev(ctx, st.ptr<float>(i_in), levels, n_levels, n_frames, p, top_percent, regions,
                                 orient_out ? st.ptr<float>(i_o) : nullptr, st.ptr<float>(i_l),
                                 value_out ? st.ptr<float>(i_v) : nullptr, peak_value_out ? st.ptr<float>(i_p) : nullptr,
                                 st.ptr<int64_t>(i_i), cap_per_frame, st.ptr<int64_t>(i_n), nullptr));
    TRY(sync0(ctx));
    if (orient_out) TRY(d2h(ctx, orient_out, st.ptr<float>(i_o), b3));
    if (line_end_out) TRY(d2h(ctx, line_end_out, st.ptr<float>(i_l), b3));
    if (value_out) TRY(d2h(ctx, value_out, st.ptr<float>(i_v), b1));
    if (peak_value_out) TRY(d2h(ctx, peak_value_out, st.ptr<float>(i_p), b1));
    if (cap_per_frame) TRY(d2h(ctx, idx, st.ptr<int64_t>(i_i), bi));
    return d2h(ctx, counts, st.ptr<int64_t>(i_n), bn);
} catch (...) {
    return on_exception(ctx, "silent_rgb_keypoints");
}

SILENT_EXPORT int silent_pyramid(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames, int n_frames,
                                 float* pyr) try {
    NEED_CTX(ctx);
    if (!plan || !frames || !pyr) return fail(ctx, SILENT_E_INVALID, "silent_pyramid: NULL pointer");
    if (n_frames < 1) return fail(ctx, SILENT_E_INVALID, "silent_pyramid: n_frames must be >= 1");
    Stage st(ctx);
    const size_t bi = (size_t)plan->tab.H * plan->tab.W * plan->tab.C * 4 * n_frames;
    const size_t bo = (size_t)plan->tab.frame_px_out * plan->tab.C * 4 * n_frames;
    const size_t i_in = st.add(bi), i_out = st.add(bo);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), frames, bi));
    TRY(silent_pyramid_dev(ctx, plan, st.ptr<float>(i_in), n_frames, st.ptr<float>(i_out), nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, pyr, st.ptr<float>(i_out), bo);
} catch (...) {
    return on_exception(ctx, "silent_pyramid");
}

SILENT_EXPORT int silent_gray_pass(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames, int n_frames,
                                   const float* cs_kernel, const float* end_bank, int n_orient, float clip_hi,
                                   float* pyr, float* cs_out, float* end_out) try {
    NEED_CTX(ctx);
    if (!plan || !frames || !pyr) return fail(ctx, SILENT_E_INVALID, "silent_gray_pass: NULL pointer");
    if (n_frames < 1) return fail(ctx, SILENT_E_INVALID, "silent_gray_pass: n_frames must be >= 1");
    if (n_orient < 1 || n_orient > 8) return fail(ctx, SILENT_E_UNSUPPORTED, "silent_gray_pass: n_orient must be 3, 4 or 8");
    Stage st(ctx);
    const size_t px = (size_t)plan->tab.frame_px_out * n_frames;
    const size_t bi = (size_t)plan->tab.H * plan->tab.W * plan->tab.C * 4 * n_frames;
    const size_t b1 = px * 4, bk = px * n_orient * 4;
    const size_t i_in = st.add(bi), i_p = st.add(b1), i_cs = st.add(b1), i_end = st.add(bk);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_in), frames, bi));
    TRY(silent_gray_pass_dev(ctx, plan, st.ptr<float>(i_in), n_frames, cs_kernel, end_bank, n_orient, clip_hi,
                             st.ptr<float>(i_p), cs_out ? st.ptr<float>(i_cs) : nullptr,
                             end_out ? st.ptr<float>(i_end) : nullptr, nullptr));
    TRY(sync0(ctx));
    TRY(d2h(ctx, pyr, st.ptr<float>(i_p), b1));
    if (cs_out) TRY(d2h(ctx, cs_out, st.ptr<float>(i_cs), b1));
    if (end_out) TRY(d2h(ctx, end_out, st.ptr<float>(i_end), bk));
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_gray_pass");
}

SILENT_EXPORT int silent_centroids(silent_ctx* ctx, const float* value, const silent_extent* levels, int n_levels,
                                   int n_frames, int region_h, int region_w, float* dist_out, float* total_out) try {
    NEED_CTX(ctx);
    if (!value || !dist_out || !total_out) return fail(ctx, SILENT_E_INVALID, "silent_centroids: NULL pointer");
    long long px;
    TRY(check_levels(ctx, "silent_centroids", levels, n_levels, n_frames, &px));
    CellTab ct;
    TRY(build_cell_tab(ctx, "silent_centroids", levels, n_levels, region_h, region_w, &ct));
    Stage st(ctx);
    const size_t bv = (size_t)px * 4, bt = (size_t)ct.frame_cells * n_frames * 4;
    const size_t i_v = st.add(bv), i_d = st.add(bv), i_t = st.add(bt);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_v), value, bv));
    TRY(silent_centroids_dev(ctx, st.ptr<float>(i_v), levels, n_levels, n_frames, region_h, region_w,
                             st.ptr<float>(i_d), st.ptr<float>(i_t), nullptr));
    TRY(sync0(ctx));
    TRY(d2h(ctx, dist_out, st.ptr<float>(i_d), bv));
    return d2h(ctx, total_out, st.ptr<float>(i_t), bt);
} catch (...) {
    return on_exception(ctx, "silent_centroids");
}

SILENT_EXPORT int silent_boosting_step(silent_ctx* ctx, const float* input, const silent_extent* levels, int n_levels,
                                       int n_frames, const silent_boosting_params* params, float* energy,
                                       float* fired_out, float* energy_out) try {
    NEED_CTX(ctx);
    if (!input || !energy || !fired_out) return fail(ctx, SILENT_E_INVALID, "silent_boosting_step: NULL pointer");
    BoostP bp;
    TRY(check_boost_params(ctx, params, &bp));
    long long px;
    TRY(check_levels(ctx, "silent_boosting_step", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t b1 = (size_t)px * 4, bc = b1 * (bp.visualize ? 3 : 1);
    const size_t i_x = st.add(b1), i_e = st.add(b1), i_f = st.add(bc), i_o = st.add(bc);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_x), input, b1));
    TRY(h2d(ctx, st.ptr<float>(i_e), energy, b1));
    TRY(silent_boosting_step_dev(ctx, st.ptr<float>(i_x), levels, n_levels, n_frames, params, st.ptr<float>(i_e),
                                 st.ptr<float>(i_f), energy_out ? st.ptr<float>(i_o) : nullptr, nullptr));
    TRY(sync0(ctx));
    TRY(d2h(ctx, energy, st.ptr<float>(i_e), b1));
    TRY(d2h(ctx, fired_out, st.ptr<float>(i_f), bc));
    return energy_out ? d2h(ctx, energy_out, st.ptr<float>(i_o), bc) : SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_boosting_step");
}

SILENT_EXPORT int silent_affine_clip(silent_ctx* ctx, const float* in, size_t n_values,
                                     const silent_affine_params* params, float* out) try {
    NEED_CTX(ctx);
    if (!in || !out || !params) return fail(ctx, SILENT_E_INVALID, "silent_affine_clip: NULL pointer");
    if (n_values == 0) return fail(ctx, SILENT_E_INVALID, "silent_affine_clip: empty tensor");
    Stage st(ctx);
    const size_t b = n_values * 4;
    const size_t i_x = st.add(b);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_x), in, b));
    TRY(silent_affine_clip_dev(ctx, st.ptr<float>(i_x), n_values, params, st.ptr<float>(i_x), nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, out, st.ptr<float>(i_x), b);
} catch (...) {
    return on_exception(ctx, "silent_affine_clip");
}

SILENT_EXPORT int silent_cast_interleave(silent_ctx* ctx, const void* in, int in_dtype, size_t n_pixels, int in_stride,
                                         int in_offset, int count, float* out, int out_stride, int out_offset) try {
    NEED_CTX(ctx);
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, "silent_cast_interleave: NULL pointer");
    const size_t es = dtype_size(in_dtype);
    if (!es) return fail(ctx, SILENT_E_UNSUPPORTED, "silent_cast_interleave: unknown in_dtype");
    if (n_pixels == 0 || in_stride < 1 || out_stride < 1) return fail(ctx, SILENT_E_INVALID, "silent_cast_interleave: empty tensor");
    if (count < 1 || in_offset < 0 || out_offset < 0 || in_stride < in_offset + count || out_stride < out_offset + count)
        return fail(ctx, SILENT_E_INVALID, "silent_cast_interleave: need 0 <= offset and offset + count <= stride on both sides");
    Stage st(ctx);
    // the last pixel needs offset + count elements, not a whole stride: a caller's buffer may end with its last used element
    const size_t bi = ((n_pixels - 1) * (size_t)in_stride + (size_t)(in_offset + count)) * es;
    const size_t bo = ((n_pixels - 1) * (size_t)out_stride + (size_t)(out_offset + count)) * 4;
    const size_t i_x = st.add(bi), i_o = st.add(bo);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<char>(i_x), in, bi));
    TRY(h2d(ctx, st.ptr<float>(i_o), out, bo));      // the elements of out that this call does not write keep their values
    TRY(silent_cast_interleave_dev(ctx, st.ptr<char>(i_x), in_dtype, n_pixels, in_stride, in_offset, count, st.ptr<float>(i_o),
                                   out_stride, out_offset, nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, out, st.ptr<float>(i_o), bo);
} catch (...) {
    return on_exception(ctx, "silent_cast_interleave");
}

SILENT_EXPORT int silent_resize_nearest(silent_ctx* ctx, const float* in, const silent_extent* in_levels, int n_levels,
                                        int n_frames, int channels, const silent_extent* out_levels, float* out) try {
    NEED_CTX(ctx);
    if (!in || !out) return fail(ctx, SILENT_E_INVALID, "silent_resize_nearest: NULL pointer");
    if (channels < 1) return fail(ctx, SILENT_E_INVALID, "silent_resize_nearest: channels must be >= 1");
    long long ipx, opx;
    TRY(check_levels(ctx, "silent_resize_nearest", in_levels, n_levels, n_frames, &ipx));
    TRY(check_levels(ctx, "silent_resize_nearest", out_levels, n_levels, n_frames, &opx));
    Stage st(ctx);
    const size_t bi = (size_t)ipx * channels * 4, bo = (size_t)opx * channels * 4;
    const size_t i_x = st.add(bi), i_o = st.add(bo);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_x), in, bi));
    TRY(silent_resize_nearest_dev(ctx, st.ptr<float>(i_x), in_levels, n_levels, n_frames, channels, out_levels,
                                  st.ptr<float>(i_o), nullptr));
    TRY(sync0(ctx));
    return d2h(ctx, out, st.ptr<float>(i_o), bo);
} catch (...) {
    return on_exception(ctx, "silent_resize_nearest");
}

SILENT_EXPORT int silent_select_peaks(silent_ctx* ctx, const float* color, const float* value, const silent_extent* levels,
                                      int n_levels, int n_frames, int channels, double top_percent, float* top_out,
                                      float* peaks_out, float* peak_value_out) try {
    NEED_CTX(ctx);
    if (!color) return fail(ctx, SILENT_E_INVALID, "silent_select_peaks: NULL pointer");
    if (!top_out && !peaks_out && !peak_value_out) return fail(ctx, SILENT_E_INVALID, "silent_select_peaks: all outputs are NULL");
    if (channels != 1 && channels != 3) return fail(ctx, SILENT_E_UNSUPPORTED, "silent_select_peaks: channels must be 1 or 3");
    long long px;
    TRY(check_levels(ctx, "silent_select_peaks", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t bc = (size_t)px * channels * 4, bv = (size_t)px * 4;
    const size_t i_c = st.add(bc), i_v = st.add(bv), i_t = st.add(bc), i_p = st.add(bc), i_o = st.add(bv);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_c), color, bc));
    if (value) TRY(h2d(ctx, st.ptr<float>(i_v), value, bv));
    TRY(silent_select_peaks_dev(ctx, st.ptr<float>(i_c), value ? st.ptr<float>(i_v) : nullptr, levels, n_levels, n_frames,
                                channels, top_percent, top_out ? st.ptr<float>(i_t) : nullptr,
                                peaks_out ? st.ptr<float>(i_p) : nullptr, peak_value_out ? st.ptr<float>(i_o) : nullptr,
                                nullptr));
    TRY(sync0(ctx));
    if (top_out) TRY(d2h(ctx, top_out, st.ptr<float>(i_t), bc));
    if (peaks_out) TRY(d2h(ctx, peaks_out, st.ptr<float>(i_p), bc));
    if (peak_value_out) TRY(d2h(ctx, peak_value_out, st.ptr<float>(i_o), bv));
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_select_peaks");
}

SILENT_EXPORT int silent_select_keypoints(silent_ctx* ctx, const float* color, const float* value, const silent_extent* levels,
                                          int n_levels, int n_frames, int channels, double top_percent,
                                          const silent_extent* regions, float* peak_value_out, int64_t* idx,
                                          size_t cap_per_frame, int64_t* counts) try {
    NEED_CTX(ctx);
    if (!color || !regions || !counts || (!idx && cap_per_frame))
        return fail(ctx, SILENT_E_INVALID, "silent_select_keypoints: NULL pointer");
    if (channels != 1 && channels != 3) return fail(ctx, SILENT_E_UNSUPPORTED, "silent_select_keypoints: channels must be 1 or 3");
    long long px;
    TRY(check_levels(ctx, "silent_select_keypoints", levels, n_levels, n_frames, &px));
    Stage st(ctx);
    const size_t bc = (size_t)px * channels * 4, bv = (size_t)px * 4;
    const size_t bi = (size_t)n_frames * cap_per_frame * 4 * sizeof(int64_t), bn = (size_t)n_frames * sizeof(int64_t);
    const size_t i_c = st.add(bc), i_v = st.add(bv), i_o = st.add(bv), i_i = st.add(bi), i_n = st.add(bn);
    TRY(st.commit());
    TRY(h2d(ctx, st.ptr<float>(i_c), color, bc));
    if (value) TRY(h2d(ctx, st.ptr<float>(i_v), value, bv));
    TRY(silent_select_keypoints_dev(ctx, st.ptr<float>(i_c), value ? st.ptr<float>(i_v) : nullptr, levels, n_levels, n_frames,
                                    channels, top_percent, regions, peak_value_out ? st.ptr<float>(i_o) : nullptr,
                                    st.ptr<int64_t>(i_i), cap_per_frame, st.ptr<int64_t>(i_n), nullptr));
    TRY(sync0(ctx));
    if (peak_value_out) TRY(d2h(ctx, peak_value_out, st.ptr<float>(i_o), bv));
    if (cap_per_frame) TRY(d2h(ctx, idx, st.ptr<int64_t>(i_i), bi));
    return d2h(ctx, counts, st.ptr<int64_t>(i_n), bn);
} catch (...) {
    return on_exception(ctx, "silent_select_keypoints");
}

#ifdef SILENT_HOST_ONLY
// fault injectors of the host-only sanitizer build (silent_host_shim.h); not part of include/silent_hip.h
SILENT_EXPORT void silent_host_arm_fault(long countdown) { silent_host::fault_countdown() = countdown; }
SILENT_EXPORT void silent_host_fail_new_after(long countdown) { silent_host::new_countdown() = countdown; }
#endif
