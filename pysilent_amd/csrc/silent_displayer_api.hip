// libsilent_hip.so -- the reference's application graph for ONE camera frame as one call (include/silent_hip.h,
// silent_displayer_*): LineEndDisplayer.callback + run, slam_recognition/recognition_testing.py:106-144 -- frame in host memory ->
// np.asarray(float32) -> zoom.from_image -> the graph of compile() (:60-100) -> the six fetched tensors in host memory.  The
// reference pays a feed, a session.run and six fetches per frame; here the whole frame is one HIP graph: an upload node, ~20
// kernel nodes (every one a kernel of the other translation units, enqueued through their *_dev entry points while the stream
// captures), a download node.  The object owns its buffers, its stream, a context of its own (nobody else regrows the workspace
// the graph's nodes point into) and the boosting state (energy_values, recognition_testing.py:56).
#include <chrono>
#include <vector>

#include "silent_internal.h"

using namespace silent;

struct silent_displayer {
    silent_ctx* owner = nullptr;          // the caller's context (errors are reported there; it must outlive the displayer's CALLS,
                                          // not its destruction: silent_displayer_destroy only uses `device`)
    int device = 0;                       // the owner's device, kept here so that destroy never reads the owner
    silent_ctx* ctx = nullptr;            // private context: workspace of the graph's nodes
    silent_pyramid_plan* plan = nullptr;
    hipStream_t stream = nullptr;
    std::vector<hipGraph_t> graph;        // one per result slot (the kernels' output pointers point into the slot)
    std::vector<hipGraphExec_t> exec;
    hipEvent_t ev[2] = {nullptr, nullptr};
    silent_displayer_params prm{};
    float kernels[4 * 81 + 441];          // private copy of the chain's constant kernels (the caller's arrays may go away)
    int L = 0, h = 0, w = 0, ch = 0, cw = 0, hh = 0, hw = 0;
    int cast_y0 = 0, cast_x0 = 0, cast_h = 0, cast_w = 0;   // the rectangle of the frame the pyramid reads (union of the levels' crops + margin)
    size_t in_bytes = 0;
    // device
    void* slab = nullptr;
    void* d_raw = nullptr;
    float *d_frame = nullptr, *d_pyr = nullptr, *d_value = nullptr, *d_tot1 = nullptr, *d_imp = nullptr, *d_energy = nullptr;
    // pinned host: the frame, and the result slots -- two at creation (silent_displayer_step alternates between them: the results of
    // step n stay valid until step n + 2), more on demand (silent_displayer_add_slot / silent_displayer_step_slot: a caller that hands
    // the results out zero-copy steps into a slot nobody holds any more)
    void* h_in = nullptr;
    std::vector<float*> h_out;
    size_t out_floats[6] = {0, 0, 0, 0, 0, 0}, out_off[6] = {0, 0, 0, 0, 0, 0}, out_total = 0;   // (offsets / total in floats, 64-byte steps)
    // (round 6: the six results have no device copy any more -- the kernels write them into the pinned slot, displayer_enqueue)
    int slot = 0;
    long long steps = 0;
    // completion signal (round 6): a one-thread kernel behind the last one stores the number of completed frames into h_flag (pinned host
    // memory); a step that does not ask for the device time polls it instead of synchronising the stream
    unsigned long long* d_seq = nullptr;
    unsigned long long* h_flag = nullptr;
    unsigned long long frames_signalled = 0;   // what h_flag will read once every launched frame has completed
};

static size_t dt_size(int dt) {
    switch (dt) {
        case SILENT_DT_U8: return 1;
        case SILENT_DT_U16: case SILENT_DT_I16: return 2;
        case SILENT_DT_F32: case SILENT_DT_I32: return 4;
        case SILENT_DT_F64: case SILENT_DT_I64: return 8;
        default: return 0;
    }
}

static void displayer_free(silent_displayer* d) {
    if (!d) return;
    for (hipGraphExec_t e : d->exec)
        if (e) (void)hipGraphExecDestroy(e);
    for (hipGraph_t g : d->graph)
        if (g) (void)hipGraphDestroy(g);
    for (hipEvent_t e : d->ev)
        if (e) (void)hipEventDestroy(e);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    if (d->slab) (void)hipFree(d->slab);
    if (d->h_flag) (void)hipHostFree((void*)d->h_flag);
    if (d->h_in) (void)hipHostFree(d->h_in);
    for (float* p : d->h_out)
        if (p) (void)hipHostFree(p);
    if (d->plan) silent_pyramid_plan_destroy(d->plan);
    if (d->ctx) silent_destroy(d->ctx);
    delete d;
}

// the kernels of one frame on d->stream, between the upload and the download (recognition_testing.py:141-142, :69-100)
static int displayer_enqueue(silent_displayer* d, int slot) {
    silent_ctx* c = d->ctx;
    silent_stream s = (silent_stream)d->stream;
    const int L = d->L;
    const silent_extent lev = {d->h, d->w};
    // Round 6, zero copy at both ends.  The pinned buffers are device-accessible (hipHostMalloc: mapped, coherent): a uint8 camera
    // frame is read by the cast kernel straight from the pinned input buffer (no upload node), and the six results are WRITTEN by
    // the kernels that produce them straight into the pinned result slot -- posted writes over the link, beside the arithmetic --
    // instead of one 3.55 MB download behind the last kernel (64 of the frame's 151 us at 640 x 480).
    float* const out = d->h_out[slot];
    // np.asarray(frame, dtype=float32) (:141)
    if (d->prm.frame_dtype != SILENT_DT_F32) {
        // (only the rectangle the pyramid reads: the union of the levels' crops, about half of the frame in the reference's layout;
        // the rest of d_frame stays 0 from creation)
        TRY(cast_rect_launch(c, d->h_in, d->prm.frame_dtype, d->prm.frame_w, 3, d->cast_y0, d->cast_x0, d->cast_h, d->cast_w, d->d_frame, d->stream));
    } else {
        // (float32 frames: the walk re-reads halo columns and rows -- from device memory, behind one upload)
        HIP_TRY(d->owner, hipMemcpyAsync(d->d_raw, d->h_in, d->in_bytes, hipMemcpyHostToDevice, d->stream));
    }
    // zoom.from_image (:142)
    TRY(silent_pyramid_dev(c, d->plan, d->prm.frame_dtype == SILENT_DT_F32 ? (const float*)d->d_raw : d->d_frame, 1, d->d_pyr, s));
    // rgc -> rgby -> orientation -> line-end -> clip -> pad_inwards; get_value_from_color (:69-77): the pyramid's levels are the batch
    silent_rgb_chain_params cp = d->prm.chain;
    cp.rgc = d->kernels; cp.rgby = d->kernels + 81; cp.stripe = d->kernels + 162; cp.end = d->kernels + 243; cp.blur = d->kernels + 324;
    TRY(silent_rgb_line_end_dev(c, d->d_pyr, &lev, 1, L, &cp, out + d->out_off[0], out + d->out_off[5], d->d_value, s));
    // centroids of gray / 255 (:79-80), importances (:81); the same on the nearest-neighbour half-size map (:82-84); get_boosting
    // (:86, advances energy_values); what the reference fetches (:99-100): 255 - centroids * 255 (both), fired * 255, update --
    // fifteen launches of the per-op path (affine, centroid cells / dist, resize, boost power / update) as two (silent_peaks.h, DispTail)
    TRY(displayer_tail(c, L, d->h, d->w, d->prm.centroid_region_h, d->prm.centroid_region_w, d->hh, d->hw, &d->prm.boosting, d->d_value,
                       nullptr, nullptr, d->d_tot1, d->d_imp, d->d_energy, out + d->out_off[1], out + d->out_off[2], out + d->out_off[3],
                       out + d->out_off[4], d->stream, d->d_seq, d->h_flag));
    return SILENT_OK;
}

SILENT_EXPORT int silent_displayer_create(silent_ctx* ctx, const silent_displayer_params* p, const silent_pyr_level* levels, int n_levels,
                                          silent_displayer** out) try {
    NEED_CTX(ctx);
    const char* who = "silent_displayer_create";
    if (!p || !levels || !out) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    *out = nullptr;
    if (!p->chain.rgc || !p->chain.rgby || !p->chain.stripe || !p->chain.blur || !p->chain.end)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": a kernel pointer in params is NULL");
    if (!dt_size(p->frame_dtype)) return fail(ctx, SILENT_E_UNSUPPORTED, std::string(who) + ": unknown frame_dtype");
    if (p->frame_h < 1 || p->frame_w < 1 || n_levels < 1 || n_levels > kMaxLevels || p->centroid_region_h < 1 || p->centroid_region_w < 1)
        return fail(ctx, SILENT_E_INVALID, std::string(who) + ": bad extents");
    for (int l = 0; l < n_levels; ++l)
        if (levels[l].out_h != levels[0].out_h || levels[l].out_w != levels[0].out_w)
            return fail(ctx, SILENT_E_INVALID, std::string(who) + ": the reference's pyramid has ONE canvas extent for every level");
    struct Guard {
        silent_displayer* d;
        ~Guard() { displayer_free(d); }
    } g{new silent_displayer()};
    silent_displayer* d = g.d;
    d->owner = ctx;
    d->device = ctx->device;
    d->prm = *p;
    std::memcpy(d->kernels, p->chain.rgc, 81 * 4);
    std::memcpy(d->kernels + 81, p->chain.rgby, 81 * 4);
    std::memcpy(d->kernels + 162, p->chain.stripe, 81 * 4);
    std::memcpy(d->kernels + 243, p->chain.end, 81 * 4);
    std::memcpy(d->kernels + 324, p->chain.blur, 441 * 4);
    int rc = silent_create(ctx->device, &d->ctx);
    if (rc != SILENT_OK) return fail(ctx, rc, std::string(who) + ": " + silent_last_error(nullptr));
    for (int i = 0; i < SILENT_TUNE_COUNT; ++i) d->ctx->tune[i] = ctx->tune[i];
    // the chain kernel writes orient / line_end straight into pinned host memory here: 16-byte stores (knob bit 7, same bits as
    // the 12-byte form, same speed into device memory) make better use of the link (in-place p50 0.174 -> 0.166 ms at 640 x 480)
    d->ctx->tune[SILENT_TUNE_RGB] |= 128u;
    rc = silent_pyramid_plan_create(d->ctx, p->frame_h, p->frame_w, 3, levels, n_levels, &d->plan);
    if (rc != SILENT_OK) return fail(ctx, rc, std::string(who) + ": " + silent_last_error(d->ctx));
    {
        // union of the crops the levels resample, + a margin for the walk's ring halo / alignment slack (read, never used), clipped
        int y0 = p->frame_h, x0 = p->frame_w, y1 = 0, x1 = 0;
        for (int l = 0; l < n_levels; ++l) {
            y0 = std::min(y0, levels[l].src_y0);
            x0 = std::min(x0, levels[l].src_x0);
            y1 = std::max(y1, levels[l].src_y0 + levels[l].src_h);
            x1 = std::max(x1, levels[l].src_x0 + levels[l].src_w);
        }
        y0 = std::max(0, y0 - 8); x0 = std::max(0, x0 - 16);
        y1 = std::min(p->frame_h, y1 + 8); x1 = std::min(p->frame_w, x1 + 16);
        d->cast_y0 = y0; d->cast_x0 = x0; d->cast_h = std::max(1, y1 - y0); d->cast_w = std::max(1, x1 - x0);
    }
    d->L = n_levels;
    d->h = levels[0].out_h;
    d->w = levels[0].out_w;
    d->ch = (d->h + p->centroid_region_h - 1) / p->centroid_region_h;
    d->cw = (d->w + p->centroid_region_w - 1) / p->centroid_region_w;
    // tf.image.resize_images(gray, [int(h / e ** .5), int(w / e ** .5)]) with the float32 arithmetic of recognition_testing.py:82
    const float root_e = (float)std::exp(0.5);
    d->hh = std::max(1, (int)((float)d->h / root_e));
    d->hw = std::max(1, (int)((float)d->w / root_e));
    const size_t px = (size_t)d->L * d->h * d->w, cells = (size_t)d->L * d->ch * d->cw, px2 = (size_t)d->L * d->hh * d->hw;
    const int vis = p->boosting.visualize ? 3 : 1;
    d->in_bytes = (size_t)p->frame_h * p->frame_w * 3 * dt_size(p->frame_dtype);
    // the six results back to back in 64-byte steps: the layout of a pinned host slot
    const size_t outs[6] = {px * 3, px, px2, cells * vis, cells * vis, px * 3};
    for (int i = 0; i < 6; ++i) {
        d->out_floats[i] = outs[i];
        d->out_off[i] = d->out_total;
        d->out_total += (outs[i] + 15) / 16 * 16;
    }
    (void)px2;
    const size_t want[] = {d->in_bytes, (size_t)p->frame_h * p->frame_w * 12, px * 12, px * 4, cells * 4, cells * 4, cells * 4, 64};
    size_t total = 0;
    for (size_t b : want) total += align_up(b);
    HIP_TRY(ctx, hipMalloc(&d->slab, total));
    char* at = (char*)d->slab;
    auto take = [&](size_t b) { char* r = at; at += align_up(b); return r; };
    d->d_raw = take(want[0]);
    float** f[] = {&d->d_frame, &d->d_pyr, &d->d_value, &d->d_tot1, &d->d_imp, &d->d_energy};
    for (size_t i = 0; i < sizeof(f) / sizeof(f[0]); ++i) *f[i] = (float*)take(want[i + 1]);
    {
        char* sig = take(64);
        HIP_TRY(ctx, hipMemset(sig, 0, 64));
        d->d_seq = (unsigned long long*)sig;
    }
    HIP_TRY(ctx, hipMemset(d->d_frame, 0, (size_t)p->frame_h * p->frame_w * 12));   // what the cast never writes reads as 0
    HIP_TRY(ctx, hipHostMalloc((void**)&d->h_flag, 64, hipHostMallocDefault));
    *d->h_flag = 0;
    HIP_TRY(ctx, hipHostMalloc(&d->h_in, d->in_bytes, hipHostMallocDefault));
    for (int k = 0; k < 2; ++k) {
        float* hp = nullptr;
        HIP_TRY(ctx, hipHostMalloc((void**)&hp, d->out_total * 4, hipHostMallocDefault));
        d->h_out.push_back(hp);
        d->graph.push_back(nullptr);
        d->exec.push_back(nullptr);
    }
    HIP_TRY(ctx, hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    for (hipEvent_t& e : d->ev) HIP_TRY(ctx, hipEventCreate(&e));
    // initialize_boosting: 8 everywhere (boosting.py:6-7, recognition_testing.py:56)
    std::vector<float> eight(cells, 8.0f);
    HIP_TRY(ctx, hipMemcpy(d->d_energy, eight.data(), cells * 4, hipMemcpyHostToDevice));
    *out = d;
    g.d = nullptr;
    return SILENT_OK;
} catch (...) {
    return on_exception(ctx, "silent_displayer_create");
}

SILENT_EXPORT void silent_displayer_destroy(silent_displayer* d) try {
    if (!d) return;
    DeviceGuard guard(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    displayer_free(d);
} catch (...) {
}

SILENT_EXPORT int silent_displayer_shape(const silent_displayer* d, int32_t* shape7, size_t* out_floats6) try {
    if (!d || !shape7) return SILENT_E_INVALID;
    const int32_t s[7] = {d->L, d->h, d->w, d->ch, d->cw, d->hh, d->hw};
    std::memcpy(shape7, s, sizeof(s));
    if (out_floats6)
        for (int i = 0; i < 6; ++i) out_floats6[i] = d->out_floats[i];
    return SILENT_OK;
} catch (...) {
    return on_exception(nullptr, "silent_displayer_shape");
}

static int displayer_fail(silent_displayer* d, int rc, const char* who) {
    // (a failing *_dev call left its message in the private context)
    if (!d->ctx->err.empty()) return fail(d->owner, rc, std::string(who) + ": " + d->ctx->err);
    return rc;
}

// One more pinned result slot (its graph is captured the first time a frame is stepped into it); *slot_index: its number.
SILENT_EXPORT int silent_displayer_add_slot(silent_displayer* d, int* slot_index) try {
    if (!d || !slot_index) return fail(d ? d->owner : nullptr, SILENT_E_INVALID, "silent_displayer_add_slot: NULL pointer");
    silent_ctx* ctx = d->owner;
    NEED_CTX(ctx);
    if (d->h_out.size() >= 64) return fail(ctx, SILENT_E_INVALID, "silent_displayer_add_slot: at most 64 result slots");
    float* hp = nullptr;
    HIP_TRY(ctx, hipHostMalloc((void**)&hp, d->out_total * 4, hipHostMallocDefault));
    d->h_out.push_back(hp);
    d->graph.push_back(nullptr);
    d->exec.push_back(nullptr);
    *slot_index = (int)d->h_out.size() - 1;
    return SILENT_OK;
} catch (...) {
    return on_exception(d ? d->owner : nullptr, "silent_displayer_add_slot");
}

static int displayer_step_slot(silent_displayer* d, const void* frame_host, int slot, const float** results, float* gpu_ms, const char* who);

// frame_host: the camera frame [frame_h, frame_w, 3] of the dtype the displayer was created for.  results[0 .. 5]: pointers INTO
// the displayer's pinned result slot (layouts: silent_displayer_shape), valid until the second next step (slots 0 and 1 alternate).
// gpu_ms (may be NULL): device time of the frame from events around the graph.  Synchronous.
SILENT_EXPORT int silent_displayer_step(silent_displayer* d, const void* frame_host, const float** results, float* gpu_ms) try {
    if (!d) return fail(nullptr, SILENT_E_INVALID, "silent_displayer_step: displayer is NULL");
    const int rc = displayer_step_slot(d, frame_host, d->slot, results, gpu_ms, "silent_displayer_step");
    if (rc == SILENT_OK) d->slot ^= 1;
    return rc;
} catch (...) {
    return on_exception(d ? d->owner : nullptr, "silent_displayer_step");
}

SILENT_EXPORT int silent_displayer_step_slot(silent_displayer* d, const void* frame_host, int slot, const float** results, float* gpu_ms) try {
    if (!d) return fail(nullptr, SILENT_E_INVALID, "silent_displayer_step_slot: displayer is NULL");
    if (slot < 0 || slot >= (int)d->h_out.size())
        return fail(d->owner, SILENT_E_INVALID, "silent_displayer_step_slot: no such result slot (silent_displayer_add_slot)");
    return displayer_step_slot(d, frame_host, slot, results, gpu_ms, "silent_displayer_step_slot");
} catch (...) {
    return on_exception(d ? d->owner : nullptr, "silent_displayer_step_slot");
}

static int displayer_step_slot(silent_displayer* d, const void* frame_host, int slot, const float** results, float* gpu_ms, const char* who) {
    silent_ctx* ctx = d->owner;
    NEED_CTX(ctx);
    if (!frame_host || !results) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": NULL pointer");
    // (a capture loop may grab INTO silent_displayer_input; a frame that is a view at an offset into that buffer overlaps it: memmove)
    if (frame_host != d->h_in) std::memmove(d->h_in, frame_host, d->in_bytes);
    d->ctx->err.clear();
    if (gpu_ms) HIP_TRY(ctx, hipEventRecord(d->ev[0], d->stream));
    if (d->steps == 0) {
        // the first frame runs eagerly: the private context's workspace grows to its final size outside any capture
        const int rc = displayer_enqueue(d, slot);
        if (rc != SILENT_OK) return displayer_fail(d, rc, who);
    } else {
        if (!d->exec[slot]) {
            // the same sequence under stream capture, once per result slot (the kernels' output pointers point into the slot)
            HIP_TRY(ctx, hipStreamBeginCapture(d->stream, hipStreamCaptureModeRelaxed));
            const int rc = displayer_enqueue(d, slot);
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(d->stream, &g);
            if (rc != SILENT_OK) {
                if (g) (void)hipGraphDestroy(g);
                (void)hipGetLastError();
                return displayer_fail(d, rc, who);
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                return fail(ctx, SILENT_E_HIP, std::string(who) + ": hipStreamEndCapture: " + hipGetErrorString(e));
            }
            d->graph[slot] = g;
            HIP_TRY(ctx, hipGraphInstantiate(&d->exec[slot], g, nullptr, nullptr, 0));
        }
        HIP_TRY(ctx, hipGraphLaunch(d->exec[slot], d->stream));
    }
    ++d->frames_signalled;                       // one more frame in the queue: its signal kernel will store this number into h_flag
    if (gpu_ms) {
        HIP_TRY(ctx, hipEventRecord(d->ev[1], d->stream));
        HIP_TRY(ctx, hipStreamSynchronize(d->stream));
        HIP_TRY(ctx, hipEventElapsedTime(gpu_ms, d->ev[0], d->ev[1]));
    } else {
        // Nobody asked for the device time: wait for the frame's own completion word in pinned memory (a one-thread kernel behind
        // the last one writes it; the results are in the same kind of memory, stored by kernels that have ended) instead of a stream synchronisation --
        // the driver's wake-up costs more than the last two kernels of a 640 x 480 frame.  Bounded: after 20 ms the stream is
        // synchronised the ordinary way (a faulting kernel never signals; the error then surfaces there).
        const unsigned long long want = d->frames_signalled;
        const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
        bool seen = false;
#ifdef SILENT_HOST_ONLY
        for (unsigned spin = 0; spin < 1; ++spin) {        // (no device behind the sanitizer build: nothing will ever signal)
#else
        for (unsigned spin = 0;; ++spin) {
#endif
            if (__atomic_load_n(d->h_flag, __ATOMIC_ACQUIRE) >= want) {
                seen = true;
                break;
            }
            __builtin_ia32_pause();
            if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() > t_end) break;
        }
        if (!seen) HIP_TRY(ctx, hipStreamSynchronize(d->stream));
    }
    for (int i = 0; i < 6; ++i) results[i] = d->h_out[slot] + d->out_off[i];
    ++d->steps;
    return SILENT_OK;
}

// The displayer's pinned input buffer (frame_h x frame_w x 3 of the frame dtype): a frame written there and passed to
// silent_displayer_step as this very pointer is uploaded without the staging copy.
SILENT_EXPORT int silent_displayer_input(silent_displayer* d, void** frame_buffer, size_t* bytes) try {
    if (!d || !frame_buffer) return fail(d ? d->owner : nullptr, SILENT_E_INVALID, "silent_displayer_input: NULL pointer");
    *frame_buffer = d->h_in;
    if (bytes) *bytes = d->in_bytes;
    return SILENT_OK;
} catch (...) {
    return on_exception(d ? d->owner : nullptr, "silent_displayer_input");
}

// The boosting state (energy_values, recognition_testing.py:56): [levels, ceil(h / region_h), ceil(w / region_w)] float32.
SILENT_EXPORT int silent_displayer_get_state(silent_displayer* d, float* energy_host) try {
    if (!d || !energy_host) return fail(d ? d->owner : nullptr, SILENT_E_INVALID, "silent_displayer_get_state: NULL pointer");
    silent_ctx* ctx = d->owner;
    NEED_CTX(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(d->stream));
    HIP_TRY(ctx, hipMemcpy(energy_host, d->d_energy, (size_t)d->L * d->ch * d->cw * 4, hipMemcpyDeviceToHost));
    return SILENT_OK;
} catch (...) {
    return on_exception(d ? d->owner : nullptr, "silent_displayer_get_state");
}

SILENT_EXPORT int silent_displayer_set_state(silent_displayer* d, const float* energy_host) try {
    if (!d || !energy_host) return fail(d ? d->owner : nullptr, SILENT_E_INVALID, "silent_displayer_set_state: NULL pointer");
    silent_ctx* ctx = d->owner;
    NEED_CTX(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(d->stream));
    HIP_TRY(ctx, hipMemcpy(d->d_energy, energy_host, (size_t)d->L * d->ch * d->cw * 4, hipMemcpyHostToDevice));
    return SILENT_OK;
} catch (...) {
    return on_exception(d ? d->owner : nullptr, "silent_displayer_set_state");
}
