// libsilent_hip.so -- pointwise ops, per-level thresholds, 3x3 NMS, keypoint indices, centroids, boosting state and the
// display-graph glue (silent_peaks.h); silent_rgb_keypoints = the fused RGB chain (silent_rgb_api.hip) + the keypoint tail.
#include "silent_internal.h"
#include "silent_peaks.h"

using namespace silent;

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
                           peaks_out, peak_value_out, tab, a, b, mm, no_regions, nullptr, nullptr, (unsigned)blocks);
    else
        hipLaunchKernelGGL((select_peaks_kernel<1, false>), dim3((unsigned)blocks), dim3(256), 0, s, color, value, top_out,
                           peaks_out, peak_value_out, tab, a, b, mm, no_regions, nullptr, nullptr, (unsigned)blocks);
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

// Blocks of a sparse-tail launch over n tiles whose blocks mostly find nothing to do (for_live_tiles, silent_peaks.h): a block
// looks at up to 64 tiles; 2048 blocks keep the chip busy where a frame does run the dense kernels.
static unsigned sparse_grid(long long n) { return (unsigned)std::min<long long>(n, std::max<long long>((n + 63) / 64, 2048)); }

// count -> scan -> ordered write, given the cell maxima
static void keypoint_passes(const float* value, const LevelTab& tab, long long blocks, const RegionTab& rt, const KeypointWs& w,
                            bool general, int n_frames, int64_t* idx, size_t cap_per_frame, int64_t* counts, hipStream_t s,
                            const int* dense_flags = nullptr, float* caller_map = nullptr) {
    if (general)
        hipLaunchKernelGGL(region_count_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, value, tab, rt, w.cells, w.pooled, w.chunk_counts, w.hit_masks, dense_flags);
    else
        hipLaunchKernelGGL(region_count_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, value, tab, rt, w.cells, w.pooled, w.chunk_counts, w.hit_masks, dense_flags);
    if (dense_flags)   // sparse tail: the candidates' hits join what the count pass left, then the scan (one block per frame both)
        hipLaunchKernelGGL(sparse_finish_kernel, dim3((unsigned)n_frames), dim3(256), 0, s, tab, rt, w.cells, w.cand, w.cand_n, dense_flags,
                           w.hit_masks, w.chunk_counts, caller_map, w.chunk_offsets, counts);
    else
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
        hipLaunchKernelGGL(sparse_modes_kernel, dim3((unsigned)n_frames), dim3(256), 0, s, sp.tab, rt, w.cells, w.cand_n, w.nan_flags,
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
                               peak_value_out, sp.stab, a, b, mm, rt, nullptr, nullptr, (unsigned)sp.sblocks);
        else
            hipLaunchKernelGGL((select_peaks_kernel<1, false>), dim3((unsigned)sp.sblocks), dim3(256), 0, s, color, value, nullptr, nullptr,
                               peak_value_out, sp.stab, a, b, mm, rt, nullptr, nullptr, (unsigned)sp.sblocks);
        TRY(region_window_maxima(ctx, who, peak_value_out, levels, n_levels, n_frames, rt, w, s));
    } else if (channels == 3) {
        // (sparse tail: a bounded grid -- usually no (frame, level) runs this pass)
        hipLaunchKernelGGL((select_peaks_kernel<3, true>), dim3(dense_flags ? sparse_grid(sp.sblocks) : (unsigned)sp.sblocks), dim3(256), 0, s, color, value, nullptr, nullptr,
                           peak_value_out, sp.stab, a, b, mm, rt, w.cells, dense_flags, (unsigned)sp.sblocks);
    } else {
        hipLaunchKernelGGL((select_peaks_kernel<1, true>), dim3((unsigned)sp.sblocks), dim3(256), 0, s, color, value, nullptr, nullptr,
                           peak_value_out, sp.stab, a, b, mm, rt, w.cells, dense_flags, (unsigned)sp.sblocks);
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

// (silent_displayer_api.hip) np.asarray(frame, float32) for the rectangle of the frame the displayer's pyramid reads
int cast_rect_launch(silent_ctx* ctx, const void* in, int in_dtype, int W, int C, int y0, int x0, int h, int w, float* out, hipStream_t s) {
    const char* who = "silent_displayer_step";
    if (!in || !out || h < 1 || w < 1) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": bad cast rectangle");
    const long long total = (long long)h * w * C;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256ll * 64);
#define CAST_CASE(DT, T) \
    case DT: hipLaunchKernelGGL(cast_rect_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)in, out, W, C, y0, x0, h, w); break
    switch (in_dtype) {
        CAST_CASE(SILENT_DT_U8, unsigned char);
        CAST_CASE(SILENT_DT_F32, float);
        CAST_CASE(SILENT_DT_F64, double);
        CAST_CASE(SILENT_DT_I32, int);
        CAST_CASE(SILENT_DT_U16, unsigned short);
        CAST_CASE(SILENT_DT_I16, short);
        CAST_CASE(SILENT_DT_I64, long long);
        default: return fail(ctx, SILENT_E_UNSUPPORTED, std::string(who) + ": frame dtype");
    }
#undef CAST_CASE
    return check_launch(ctx, who);
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

// ------------------------------------------------------------------------------------------ the displayer's fused tail
// (silent_displayer_api.hip) value [L, h, w] -> the four small results + the advanced state, in two launches (silent_peaks.h, DispTail);
// g / im2n: unused since round 6 (value / 255 and its resize are computed on the fly), may be NULL
int displayer_tail(silent_ctx* ctx, int L, int h, int w, int rh, int rw, int h2, int w2, const silent_boosting_params* boost, const float* value,
                   float* g, float* im2n, float* tot1, float* imp, float* energy, float* out1, float* out2, float* out3, float* update,
                   hipStream_t s, unsigned long long* seq, unsigned long long* flag) {
    const char* who = "silent_displayer_step";
    DispTail t;
    std::memset(&t, 0, sizeof(t));
    TRY(check_boost_params(ctx, boost, &t.bp));
    const silent_extent full = {h, w}, half = {h2, w2};
    CellTab c1, c2;
    TRY(build_cell_tab(ctx, who, &full, 1, rh, rw, &c1));
    TRY(build_cell_tab(ctx, who, &half, 1, rh, rw, &c2));
    t.L = L; t.h = h; t.w = w; t.h2 = h2; t.w2 = w2; t.rh = rh; t.rw = rw;
    t.ch = c1.oh[0]; t.cw = c1.ow[0]; t.ch2 = c2.oh[0]; t.cw2 = c2.ow[0];
    t.y_first = c1.y_first[0]; t.x_first = c1.x_first[0]; t.y_first2 = c2.y_first[0]; t.x_first2 = c2.x_first[0];
    t.yscale_c = c1.yscale[0]; t.xscale_c = c1.xscale[0]; t.yscale_c2 = c2.yscale[0]; t.xscale_c2 = c2.xscale[0];
    t.yscale_r = (float)h / (float)h2;      // (silent_resize_nearest_dev: float32 in / out)
    t.xscale_r = (float)w / (float)w2;
    t.by255 = AffineP{1.f, 255.f, 0.f, -INFINITY, INFINITY, 0.f};
    t.imp = AffineP{255.f / 4.0f, 1.f, 0.f, 1.f, 256.f, -1.f};
    t.inv = AffineP{-255.f, 1.f, 255.f, -INFINITY, INFINITY, 0.f};
    t.x255 = AffineP{255.f, 1.f, 0.f, -INFINITY, INFINITY, 0.f};
    const long long px = (long long)L * h * w, px2 = (long long)L * h2 * w2, cells = (long long)L * t.ch * t.cw, cells2 = (long long)L * t.ch2 * t.cw2;
    TRY(workspace(ctx, s, (size_t)(cells * 2 + cells2 * 2 + cells) * sizeof(float)));
    t.cxy = (float*)ctx->ws.p;
    t.cxy2 = t.cxy + cells * 2;
    t.m = t.cxy2 + cells2 * 2;
    t.value = value; t.g = g; t.im2n = im2n; t.tot1 = tot1; t.imp_out = imp; t.energy = energy;
    t.out1 = out1; t.out2 = out2; t.out3 = out3; t.update = update;
    t.seq = seq; t.flag = flag;
    auto grid = [](long long n) { return dim3((unsigned)((n + 255) / 256)); };
    hipLaunchKernelGGL(disp_cells_kernel, grid(cells + cells2), dim3(256), 0, s, t);
    hipLaunchKernelGGL(disp_dist_boost_kernel, grid(px + px2 + cells), dim3(256), 0, s, t);
    if (flag) hipLaunchKernelGGL(disp_signal_kernel, dim3(1), dim3(64), 0, s, seq, flag);
    return check_launch(ctx, who);
}

// ------------------------------------------------------------------------------------------ RGB chain + keypoints

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
    const int th = rgb_chain_tile_height(ctx, levels, n_levels, n_frames, &pair_kernel, true);
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

// ------------------------------------------------------------------------------------------ host-pointer twins

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
