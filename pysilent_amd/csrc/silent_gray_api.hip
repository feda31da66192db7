// libsilent_hip.so -- the grayscale pass (silent_gray.h): CS -> line-end bank on a pyramid, and the whole pass from the frame
// (pyramid + unit levels in one read of the frame, then the remaining levels).
#include "silent_plan.h"

using namespace silent;

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
    // 7 % slower, off), bit2 non-temporal stores (no effect, off).  (Bit1 selected 32-row tiles until round 5: 3 % slower in every
    // A/B and 145 - 156 SGPR spills; the instantiations are gone since round 6, the bit is ignored.)
    const unsigned opts = ctx->tune[SILENT_TUNE_GRAY];
    const int th = kGrayTH;
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
    if (n_orient == 3) GRAY_LAUNCH(3, kGrayTH);
    else if (n_orient == 4) GRAY_LAUNCH(4, kGrayTH);
    else GRAY_LAUNCH(8, kGrayTH);
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
    const int kopts = (int)ctx->tune[SILENT_TUNE_GRAY];  // A/B knob: bit4 disables the stream path (bit3, 32-row fused tiles, is ignored since round 6)
    const bool stream_path = plan->stream_ok && !(kopts & 16);
    // 1. non-unit levels of the pyramid: by the region kernel, unless the stream kernel of step 2 produces them
    //    from the same single read of the frame; plus the zero fill of canvases larger than their zoomed crop
    if (!(parts & 3u)) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": parts must name step 1 + 2 (bit 0) and / or step 3 (bit 1)");
    if (parts & 1u) TRY(launch_pyramid(ctx, who, plan, frames, n_frames, pyr, s, false, !stream_path));
    // 2. unit levels: pyramid + CS + end in one kernel
    const int fth = kFusedTH;
    FusedTab ft;
    std::memset(&ft, 0, sizeof(ft));
    bool is_unit[kMaxLevels] = {false};
    long long tiles = 0, unit_px = 0;
    for (int l = 0; l < pt.n_levels; ++l) {
        const PyrLevelDev& d = pt.lv[l];
        if (d.kind != kPyrUnit) continue;
        is_unit[l] = true;
        if (ft.n == 0)
            for (int j = 0; j < 6; ++j) {  // every unit level has the same taps ([1,26,66,26,1]/120 and the sixth, 2^-53)
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
#define STREAM_LAUNCH(K_, G_, L_) \
    hipLaunchKernelGGL((gray_stream_kernel<K_, G_, L_>), dim3((unsigned)blocks), dim3(64 * kFusedWaves), 0, s, frames, pyr, cs_out, end_out, ft, st, w, clip_hi, (unsigned)((kopts >> 5) & 1))
            if (plan->stream_layout == 1) {          // zoom ladders of ratio 1.4 .. e^.5: five rows of the first level in flight
                if (n_orient == 3) STREAM_LAUNCH(3, 7, 1);
                else if (n_orient == 4) STREAM_LAUNCH(4, 7, 1);
                else STREAM_LAUNCH(8, 7, 1);
            } else if (st.G <= 4) {
                if (n_orient == 3) STREAM_LAUNCH(3, 4, 0);
                else if (n_orient == 4) STREAM_LAUNCH(4, 4, 0);
                else STREAM_LAUNCH(8, 4, 0);
            } else {
                if (n_orient == 3) STREAM_LAUNCH(3, 7, 0);
                else if (n_orient == 4) STREAM_LAUNCH(4, 7, 0);
                else STREAM_LAUNCH(8, 7, 0);
            }
#undef STREAM_LAUNCH
        } else {
#define FUSED_LAUNCH(K_, R_) \
    hipLaunchKernelGGL((gray_unit_fused_kernel<K_, R_>), dim3((unsigned)blocks), dim3(64 * kFusedWaves), 0, s, frames, pyr, cs_out, end_out, ft, w, clip_hi)
            if (n_orient == 3) FUSED_LAUNCH(3, kFusedTH);
            else if (n_orient == 4) FUSED_LAUNCH(4, kFusedTH);
            else FUSED_LAUNCH(8, kFusedTH);
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
