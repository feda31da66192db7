// libsilent_hip.so -- C ABI (include/silent_hip.h) over the gfx950 kernels.  This translation unit: the context, memory and
// profiling entry points, the exception barrier, workspace and tile tables.  The kernel families have one translation unit
// each (silent_*_api.hip); silent_internal.h is what they share.
#include "silent_internal.h"

using namespace silent;

static thread_local std::string g_create_err;

int fail(silent_ctx* ctx, int code, const std::string& msg) {
    if (ctx)
        ctx->err = msg;
    else
        g_create_err = msg;
    return code;
}

int on_exception(silent_ctx* ctx, const char* who) noexcept {
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

int grow(silent_ctx* ctx, DevBuf& b, size_t bytes) {
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

int workspace(silent_ctx* ctx, hipStream_t s, size_t bytes) {
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
    // (owned until the last line: a throw while the name / knobs are filled in -- std::string allocates -- must not leak it)
    std::unique_ptr<silent_ctx> ctx(new (std::nothrow) silent_ctx());
    if (!ctx) return fail(nullptr, SILENT_E_NOMEM, "silent_create: out of host memory");
    ctx->device = device;
    ctx->name = std::string(prop.name) + " (" + prop.gcnArchName + ")";
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(nullptr, SILENT_E_UNSUPPORTED, "silent_create: device is " + ctx->name + "; this library holds gfx950 code objects only");
    const char* names[SILENT_TUNE_COUNT] = {"SILENT_GRAY_OPTS", "SILENT_RGB_OPTS", "SILENT_PYRAMID_OPTS"};
    for (int i = 0; i < SILENT_TUNE_COUNT; ++i)
        if (const char* e = std::getenv(names[i])) ctx->tune[i] = (unsigned)std::strtoul(e, nullptr, 0);
    *out = ctx.release();
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

// One wavefront polling the constant-rate clock (100 MHz on gfx950: s_memrealtime) until `ticks` have passed.  Bounded by the
// host (<= 1 s), so every wave reaches its exit.
__global__ __launch_bounds__(64) void busy_wait_kernel(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

SILENT_EXPORT int silent_busy_wait_dev(silent_ctx* ctx, unsigned microseconds, silent_stream stream) try {
    NEED_CTX(ctx);
    if (microseconds > 1000000u) return fail(ctx, SILENT_E_INVALID, "silent_busy_wait: at most 1 000 000 microseconds");
    int khz = 100000;   // wall_clock64 ticks per millisecond
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device) != hipSuccess || khz <= 0) {
        (void)hipGetLastError();
        khz = 100000;
    }
    hipLaunchKernelGGL(busy_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)microseconds * (unsigned long long)khz / 1000ull);
    return check_launch(ctx, "silent_busy_wait");
} catch (...) {
    return on_exception(ctx, "silent_busy_wait_dev");
}

// A named no-op in the kernel trace: bench.py launches it once between its tuners and the settle / timed steps, and
// scripts/summarize_profile.py computes the per-kernel statistics over what FOLLOWS it (the busy-wait kernel cannot serve as the
// marker: the stream-concurrency probe launches it too, before and after the timed steps).
__global__ __launch_bounds__(64) void trace_marker_kernel() {}

SILENT_EXPORT int silent_trace_marker_dev(silent_ctx* ctx, silent_stream stream) try {
    NEED_CTX(ctx);
    hipLaunchKernelGGL(trace_marker_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream);
    return check_launch(ctx, "silent_trace_marker");
} catch (...) {
    return on_exception(ctx, "silent_trace_marker_dev");
}

// ------------------------------------------------------------------------------------------ tile tables

// tile_h == 0 selects the 1-D "chunk" decomposition (kChunk flattened pixels per block; tile_w > 0: that many).
int build_level_tab(silent_ctx* ctx, const char* who, const silent_extent* levels, int n_levels, int n_frames,
                           int tile_w, int tile_h, LevelTab* tab, long long* n_blocks,
                           const bool* skip) {
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

int check_launch(silent_ctx* ctx, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SILENT_E_HIP, std::string(what) + " launch: " + hipGetErrorString(e));
    return SILENT_OK;
}

long long pyramid_px(const silent_extent* levels, int n_levels) {
    long long px = 0;
    for (int l = 0; l < n_levels; ++l) px += (long long)levels[l].h * levels[l].w;
    return px;
}

// ------------------------------------------------------------------------------------------ host-pointer twins: staging

int h2d(silent_ctx* ctx, void* d, const void* h, size_t bytes) {
    HIP_TRY(ctx, hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return SILENT_OK;
}
int d2h(silent_ctx* ctx, void* h, const void* d, size_t bytes) {
    HIP_TRY(ctx, hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    return SILENT_OK;
}
int sync0(silent_ctx* ctx) {
    HIP_TRY(ctx, hipStreamSynchronize(nullptr));
    return SILENT_OK;
}

int check_levels(silent_ctx* ctx, const char* who, const silent_extent* levels, int n_levels, int n_frames,
                        long long* px) {
    LevelTab tab;
    long long blocks;
    TRY(build_level_tab(ctx, who, levels, n_levels, n_frames, 0, 0, &tab, &blocks));
    *px = tab.frame_px * n_frames;
    return SILENT_OK;
}

#ifdef SILENT_HOST_ONLY
// fault injectors of the host-only sanitizer build (silent_host_shim.h); not part of include/silent_hip.h
SILENT_EXPORT void silent_host_arm_fault(long countdown) { silent_host::fault_countdown() = countdown; }
SILENT_EXPORT void silent_host_fail_new_after(long countdown) { silent_host::new_countdown() = countdown; }
#endif
