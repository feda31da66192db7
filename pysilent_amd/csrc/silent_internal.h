// Host side shared by the translation units of libsilent_hip.so (one per kernel family: silent_core.hip, silent_conv_api.hip,
// silent_gray_api.hip, silent_peaks_api.hip, silent_rgb_api.hip, silent_pyramid_api.hip): the context, the exception barrier
// of the C ABI, workspace / staging helpers and the tile tables.  Nothing here is exported (-fvisibility=hidden).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "silent_host_shim.h"   // (inert unless SILENT_HOST_ONLY: the CPU container's sanitizer build of the host side)
#include "silent_common.h"

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

// (silent_core.hip) ctx == nullptr: the message goes to the thread's create-error slot
int fail(silent_ctx* ctx, int code, const std::string& msg);

// ------------------------------------------------------------------------------------------ exception barrier
// include/silent_hip.h promises that nothing throws or aborts across the ABI.  The host side allocates (std::vector, std::string):
// every extern "C" entry point is a function-try-block whose handler turns std::bad_alloc into SILENT_E_NOMEM and anything else
// into SILENT_E_INVALID (the message says what was thrown).  The handler itself must not throw: setting the message allocates.
int on_exception(silent_ctx* ctx, const char* who) noexcept;

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

#define NEED_CTX(ctx)                  \
    SILENT_FAULT_POINT();              \
    if (!(ctx)) return fail(nullptr, SILENT_E_INVALID, std::string(__func__) + ": ctx is NULL"); \
    DeviceGuard device_guard_((ctx)->device);                                                    \
    if (!device_guard_.ok) return fail(ctx, SILENT_E_HIP, std::string(__func__) + ": hipSetDevice failed")

int grow(silent_ctx* ctx, DevBuf& b, size_t bytes);

static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// The context has ONE workspace, so its users are ordered by the stream they run on.  A caller that moves to another
// stream is not an error: the previous stream is drained first (rare path), then the workspace belongs to the new one.
int workspace(silent_ctx* ctx, hipStream_t s, size_t bytes);

// ------------------------------------------------------------------------------------------ tile tables

// tile_h == 0 selects the 1-D "chunk" decomposition (kChunk flattened pixels per block; tile_w > 0: that many).
int build_level_tab(silent_ctx* ctx, const char* who, const silent_extent* levels, int n_levels, int n_frames, int tile_w, int tile_h,
                    silent::LevelTab* tab, long long* n_blocks, const bool* skip = nullptr);
int check_launch(silent_ctx* ctx, const char* what);
long long pyramid_px(const silent_extent* levels, int n_levels);

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

int h2d(silent_ctx* ctx, void* d, const void* h, size_t bytes);
int d2h(silent_ctx* ctx, void* h, const void* d, size_t bytes);
int sync0(silent_ctx* ctx);
int check_levels(silent_ctx* ctx, const char* who, const silent_extent* levels, int n_levels, int n_frames, long long* px);

// ------------------------------------------------------------------------------------------ across the families
// (silent_rgb_api.hip) the channel-uniform 7x7 regulator of silent_regulate_dev: regulate_sum_kernel lives with the RGB chain
int launch_regulate_sum(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels, int n_frames, const float* blur_hwio,
                        float regulation_value, float regulation_root, int flat_policy, float* out, hipStream_t s);
// (silent_rgb_api.hip) which fused kernel a chain launch over these levels uses and its tile height (output rows per tile)
int rgb_chain_tile_height(const silent_ctx* ctx, const silent_extent* levels, int n_levels, int n_frames, bool* pair, bool with_extrema);
// (silent_rgb_api.hip) mm: optional per-level extrema slots (already initialised); *mm_done tells whether the launch filled them
// (only the pair kernel's two-group instantiation does -- everything else leaves them to level_maxmin_kernel); st / sum /
// nan_flags: the value summary of the sparse keypoint tail (silent_peaks_api.hip)
int rgb_chain_launch(silent_ctx* ctx, const char* who, const float* pyr, const silent_extent* levels, int n_levels, int n_frames,
                     const silent_rgb_chain_params* p, float* orient_out, float* line_end_out, float* value_out, unsigned* mm,
                     bool* mm_done, silent_stream stream, const silent::SumTab* st = nullptr, float* sum = nullptr,
                     int* nan_flags = nullptr);
// (silent_peaks_api.hip) the widening cast of a rectangle of an interleaved frame (the displayer: the part of the frame its pyramid reads)
int cast_rect_launch(silent_ctx* ctx, const void* in, int in_dtype, int W, int C, int y0, int x0, int h, int w, float* out, hipStream_t s);
// (silent_peaks_api.hip) the tail of the displayer's graph after the chain, fused into two launches (silent_peaks.h, DispTail); seq / flag: the completion signal
int displayer_tail(silent_ctx* ctx, int L, int h, int w, int rh, int rw, int h2, int w2, const silent_boosting_params* boost, const float* value,
                   float* g, float* im2n, float* tot1, float* imp, float* energy, float* out1, float* out2, float* out3, float* update,
                   hipStream_t s, unsigned long long* seq = nullptr, unsigned long long* flag = nullptr);
