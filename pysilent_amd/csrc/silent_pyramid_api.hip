// libsilent_hip.so -- the zoom pyramid (silent_pyramid.h, silent_walk_rgb.h, pyramid_stream_kernel of silent_gray.h): plans
// (float64 tap tables, scipy-identical; row programs and column records of the single-read kernels) and their launches.
#include "silent_plan.h"

#include <limits>

using namespace silent;

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
            // a weight that is not 0 in scipy's double arithmetic must not become 0 in float32: an outer tap's (1 - t)^5 / 120 falls
            // below 2^-149 when t rounds to just under 1, and 0 * inf = NaN where scipy has tiny * inf = inf.  The smallest float32
            // magnitude keeps the sign and the non-finite arithmetic; next to finite pixels it is as invisible as the true weight
            float f = (float)w[j];
            if (w[j] != 0.0 && f == 0.0f) f = std::copysign(std::numeric_limits<float>::denorm_min(), (float)(w[j] < 0.0 ? -1.0 : 1.0));
            wts[6 * o + j] = f;
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
            for (int j = 0; j < 6; ++j) plan->unit_w[j] = xwl[j];
            continue;
        }
        ++tab.n_general;
        // scipy's mode-'constant' artefact: the last output coordinate (n_out - 1) * step can round to just above n_in - 1, and the
        // whole row / column is then cval = 0 WITHOUT a pixel being read (axis_table leaves its six weights 0) -- a NaN / inf pixel
        // under it stays out of the level.  0 * NaN would not: such a row / column leaves the resampler and joins the zero fill of
        // "canvas beyond the zoomed crop" (pyramid_zero_kernel); d.zoom_* from here on = the outputs the resampler produces
        auto dead = [](const float* w6) { return w6[0] == 0.0f && w6[1] == 0.0f && w6[2] == 0.0f && w6[3] == 0.0f && w6[4] == 0.0f && w6[5] == 0.0f; };
        if (d.zoom_w > 1 && dead(xwl + (size_t)(d.zoom_w - 1) * 6)) --d.zoom_w;
        if (d.zoom_h > 1 && dead(ywl + (size_t)(d.zoom_h - 1) * 6)) --d.zoom_h;
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
        zero_chunks += (pyramid_zero_count(d.zoom_h, d.zoom_w, d.out_h, d.out_w) + 1023) / 1024;   // (exactly the pixels the resampler leaves)
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
        const bool eligible = ok;
        // slot layouts (silent_gray.h, stream_slots): 0 for ladders of ratio >= e^.5, 1 ("dense", kernels for 7 levels only) down to 1.4
        for (int layout = 0; layout < 2 && eligible && !plan->stream_ok; ++layout) {
            ok = true;
            const PyrLevelDev& u = tab.lv[unit];
            const int G = tab.n_general;
            const int tiles_y = (u.out_h + kFusedTH - 1) / kFusedTH;
            const int waves_x = ((u.out_w + kFusedTW - 1) / kFusedTW) * kFusedWaves;
            const int Gp = layout ? 7 : stream_pad_levels(G), PR = kStreamProgRow(layout, Gp);
            const size_t n_rec = (size_t)tiles_y * kStreamRows;
            std::vector<int> prog(n_rec * PR, 0), hdr((size_t)G * waves_x * 2, 0), rec((size_t)G * waves_x * 64 * 8, 0);
            for (size_t r = 0; r < n_rec; ++r)
                for (int gg = 0; gg < Gp; ++gg) prog[r * PR + gg] = 7 << stream_done_shift(layout);  // inert: feeds nothing, no slot completes
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
                    const int slot = oy % stream_slots(layout, g);
                    for (int j = 0; j < 6; ++j) {
                        const int i = yb[oy] - t * kFusedTH + 2 + j;  // stream row of tap j (a tile streams rows y0-4 ..)
                        if (i < 0 || i >= kStreamRows) { ok = false; break; }
                        const size_t r = (size_t)t * kStreamRows + i;
                        const size_t e = r * G + g;
                        if (used[e * kStreamSlots + slot]) { ok = false; break; }  // two live rows in one slot: step too small
                        used[e * kStreamSlots + slot] = 1;
                        int* pr = prog.data() + r * PR;
                        int& meta = pr[g];
                        std::memcpy(pr + stream_w_off(layout, Gp, g) + slot, &yw[(size_t)(d.ytab_off + oy) * 6 + j], 4);
                        meta |= 128;  // this stream row feeds level g
                        if (j == 0) meta |= 1 << slot;
                        if (j == 5) {
                            const int ds = stream_done_shift(layout);
                            if (((meta >> ds) & 7) != 7) { ok = false; break; }  // two rows completing together
                            meta = (meta & ~(7 << ds)) | (slot << ds) | (oy << stream_row_shift(layout));
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
                    plan->stream_layout = layout;
                    plan->stream_ok = true;
                }
            }
        }
    }
    // ---- walk plans of pyramid_walk3_kernel (3 channels).  Classic pyramid (one unit level whose crop every other level
    // resamples): one plan.  Anything else (the reference's nested centre crops): a unit level alone; and the general levels
    // either as ONE "union" plan -- the walk over the OUTERMOST crop serves every level whose crop lies inside it, wherever an
    // output's 6 x 6 taps stay inside that level's own crop (no mirroring: the frame pixels ARE the taps); the inner levels' first /
    // last output rows and columns go to pyramid_border_kernel -- or, where that is not eligible, one plan per level on its own crop.
    // Per plan: a row program (one record per source row of the crop: "an output row of level g completes here" + its 6 vertical
    // weights) and column records per PX-pixel wave tile.
    // A classic pyramid of more than 7 general levels stays on the unit + region kernels: the levels beyond the seventh as walk plans
    // of their own, or left to pyramid_region_kernel behind the walk, cost a second pass over the frame for a few small levels
    // (32 x 1080p, sqrt 2, 12 levels: 1.38 / 1.35 ms against 1.27 ms; profiles/r05_experiments.txt).
    if (channels == 3) {
        struct HostPlan {
            int unit;                 // level index of the plan's unit level, or -1
            int crop;                 // the level whose crop the plan walks
            std::vector<int> gen;     // its general levels (union plans: finest first; those other than `crop` interior-only)
        };
        std::vector<std::vector<HostPlan>> candidates;
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
                HostPlan h{unit, unit, {}};
                for (int l = 0; l < n_levels; ++l)
                    if (tab.lv[l].kind == kPyrGeneral) h.gen.push_back(l);
                // finest level first: the column records' capacity per wave tile shrinks with the position (w3_rec_cap)
                std::stable_sort(h.gen.begin(), h.gen.end(), [&](int x, int y) {
                    return (double)tab.lv[x].zoom_w / tab.lv[x].src_w > (double)tab.lv[y].zoom_w / tab.lv[y].src_w;
                });
                candidates.push_back({h});
            } else {
                std::vector<HostPlan> units, singles;
                std::vector<int> gens;
                for (int l = 0; l < n_levels; ++l) {
                    if (tab.lv[l].kind == kPyrUnit) units.push_back(HostPlan{l, l, {}});
                    else if (tab.lv[l].kind == kPyrGeneral) { singles.push_back(HostPlan{-1, l, {l}}); gens.push_back(l); }
                }
                // the union: an outermost general level whose crop holds every other general level's crop
                int outer = -1;
                for (int o : gens) {
                    bool holds = true;
                    for (int l : gens) {
                        const PyrLevelDev &a = tab.lv[o], &d = tab.lv[l];
                        holds = holds && d.src_y0 >= a.src_y0 && d.src_x0 >= a.src_x0 && d.src_y0 + d.src_h <= a.src_y0 + a.src_h &&
                                d.src_x0 + d.src_w <= a.src_x0 + a.src_w;
                    }
                    if (holds) { outer = o; break; }
                }
                if (outer >= 0 && gens.size() >= 2 && gens.size() <= 7 && !(ctx->tune[SILENT_TUNE_PYRAMID] & 4u)) {
                    // finest level first: the column records' capacity per wave tile shrinks with the position (w3_rec_cap)
                    std::vector<int> order = gens;
                    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
                        return (double)tab.lv[x].zoom_w / tab.lv[x].src_w > (double)tab.lv[y].zoom_w / tab.lv[y].src_w;
                    });
                    std::vector<HostPlan> u2 = units;
                    u2.push_back(HostPlan{-1, outer, order});
                    candidates.push_back(u2);
                }
                std::vector<HostPlan> per_level = units;
                per_level.insert(per_level.end(), singles.begin(), singles.end());
                candidates.push_back(per_level);
            }
        }
        for (const std::vector<HostPlan>& hp : candidates) {
            if (plan->walk_pyr_ok) break;
            bool usable = !hp.empty() && (int)hp.size() <= kW3MaxPlans;
            for (const HostPlan& h : hp) {
                if (h.unit >= 0) {
                    const PyrLevelDev& u = tab.lv[h.unit];   // canvas at least as large as the crop
                    if (u.out_h < u.src_h || u.out_w < u.src_w) usable = false;
                }
                if (tab.lv[h.crop].src_w < 8) usable = false;
            }
            int maxg = 0;
            for (const HostPlan& h : hp) maxg = std::max(maxg, (int)h.gen.size());
            const int Gp = stream_pad_levels(std::max(maxg, 1)), PR = w3_prog_row(Gp);
            for (int px : {36, 32, 28, 24}) {
                if (!usable || plan->walk_pyr_ok) break;
                const int rec_total = w3_rec_total(px, Gp);
                std::vector<int> blob;                        // all tables of all plans, offsets in ints
                struct Off { size_t prog, hdr, rec; };
                std::vector<Off> offs;
                bool ok = true;
                Walk3Args wa;
                std::memset(&wa, 0, sizeof(wa));
                BorderTab bt;
                std::memset(&bt, 0, sizeof(bt));
                for (size_t pi = 0; pi < hp.size() && ok; ++pi) {
                    const HostPlan& h = hp[pi];
                    const PyrLevelDev& c = tab.lv[h.crop];
                    const int walk_h = h.unit >= 0 ? c.out_h : c.src_h;
                    int walk_w = h.unit >= 0 ? c.out_w : c.src_w;
                    const int G = (int)h.gen.size();
                    // A last strip that holds a sliver of the crop costs a whole strip of row steps (the reference layout on 1080p: 10
                    // of 128 pixels, one block in eleven).  Plans of general levels only give it up: the few output columns anchored in
                    // it join the border pixels (pyramid_border_px: taps straight from the frame), the walk ends at the strip boundary --
                    // the pixels behind it are still in the frame for the last strip's halo.
                    const int strip_px = kW3NC * px;
                    if (h.unit < 0 && G >= 2 && walk_w > strip_px && walk_w % strip_px != 0 && walk_w % strip_px <= px / 2)
                        walk_w -= walk_w % strip_px;
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
                        // a level on ANOTHER crop than the walk's: offset of its crop inside the walk, and the outputs whose taps
                        // (rows / columns base - 2 .. base + 3) stay inside its own crop
                        const int dy = d.src_y0 - c.src_y0, dx = d.src_x0 - c.src_x0;
                        int oy_lo = 0, oy_hi = zr, ox_lo = 0, ox_hi = zc;
                        if (d.src_y0 != c.src_y0 || d.src_x0 != c.src_x0 || d.src_h != c.src_h || d.src_w != c.src_w) {
                            while (oy_lo < zr && yb[oy_lo] - 2 < 0) ++oy_lo;
                            while (oy_hi > oy_lo && yb[oy_hi - 1] + 3 > d.src_h - 1) --oy_hi;
                            while (ox_lo < zc && xb[ox_lo] - 2 < 0) ++ox_lo;
                            while (ox_hi > ox_lo && xb[ox_hi - 1] + 3 > d.src_w - 1) --ox_hi;
                        }
                        while (ox_hi > ox_lo && xb[ox_hi - 1] + dx >= walk_w) --ox_hi;   // anchored behind a trimmed walk
                        if (oy_lo > 0 || oy_hi < zr || ox_lo > 0 || ox_hi < zc) {
                            if (oy_hi <= oy_lo || ox_hi <= ox_lo || bt.n >= kMaxLevels) { ok = false; break; }
                            BorderLevel& bl = bt.lv[bt.n++];
                            bl.level = h.gen[g];
                            bl.oy_lo = oy_lo; bl.oy_hi = oy_hi; bl.ox_lo = ox_lo; bl.ox_hi = ox_hi;
                            bl.zr = zr; bl.zc = zc;
                            bl.start = bt.per_frame;
                            bt.per_frame += zr * zc - (oy_hi - oy_lo) * (ox_hi - ox_lo);
                        }
                        for (int oy = oy_lo; oy < oy_hi && ok; ++oy) {
                            // one record entry per COMPLETING row: flag + output row, 6 weights
                            const int y = yb[oy] + dy;
                            if (y < 0 || y >= walk_h) { ok = false; break; }
                            const size_t r = (size_t)(y + 7);                // the last tap sits on stream row y + 3, index y + 4
                            if (r >= n_rec) { ok = false; break; }
                            int* pr = prog.data() + r * PR;
                            if (pr[g] & 1) { ok = false; break; }             // two rows of one level completing together: step < 1
                            pr[g] = 1 | (oy << 8);
                            std::memcpy(pr + Gp + 6 * g, &yw[(size_t)(d.ytab_off + oy) * 6], 24);
                        }
                        int ox = ox_lo;
                        for (int wx = 0; wx < waves_x && ok; ++wx) {
                            const int xw0 = wx * px;
                            while (ox < ox_hi && xb[ox] + dx < xw0) ++ox;
                            int n = 0;
                            while (ox + n < ox_hi && xb[ox + n] + dx < xw0 + px) ++n;
                            if (n > w3_rec_cap(px, g)) { ok = false; break; }  // outputs per wave tile (the gather takes <= 21)
                            hdr[((size_t)g * waves_x + wx) * 2] = ox;
                            hdr[((size_t)g * waves_x + wx) * 2 + 1] = n;
                            for (int j = 0; j < n; ++j) {
                                int* r = rec.data() + ((size_t)wx * rec_total + w3_rec_base(px, g) + j) * 8;
                                r[0] = (xb[ox + j] + dx - xw0) * 3;          // FLOAT index of tap 0, channel 0 (line starts at pixel xw0 - 2)
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
                    wp3.shift = tab.W % 4 ? 0 : (c.src_x0 * 3) % 4;   // (other widths: the loader fetches single floats, not 16-byte groups)
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
                for (int j = 0; j < 6; ++j) wa.wx[j] = plan->unit_w[j];
                plan->walk = wa;
                plan->walk_border = bt;
                plan->walk_px = px;
                plan->walk_G = Gp;
                plan->walk_pyr_ok = true;
            }
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

// Per-launch decomposition of the plan's walks (any batch size): strips of 4 x PX pixels, cut into segments of ONE height for
// every plan of the launch -- the height that makes about 3.5 blocks per block the chip holds (but not below 32 rows: every
// segment walks 8 rows of halo).  Rounds 2 - 4 minimised ceil(blocks / resident blocks) x (segment rows + 8) per plan; that
// model is right for one plan of equal blocks that fills the chip a few times (config 3: 4 ... 20 segments per frame measured
// flat within 3 %, both rules land there) and wrong for the reference layout, where it gave the union plan exactly one round of
// 224-row blocks beside the unit plan's short ones: 0.199 ms, against 0.165 ms with 12 - 20 segments (profiles/r05_experiments.txt 7).
static bool walk3_plan(const silent_ctx* ctx, const silent_pyramid_plan* plan, int n_frames, Walk3Args* wa) {
    if (plan->tab.C != 3 || !plan->walk_pyr_ok) return false;
    *wa = plan->walk;
    const int px = plan->walk_px;
    int per_cu = 1;
#define PER_CU(PX_) case PX_: per_cu = plan->walk_G <= 4 ? walk3_blocks_per_cu<4, PX_>() : walk3_blocks_per_cu<7, PX_>(); break
    switch (px) { PER_CU(36); PER_CU(32); PER_CU(28); PER_CU(24); default: return false; }
#undef PER_CU
    const long long resident = (long long)per_cu * ctx->n_cus;
    long long strip_rows = 0;                                   // rows x strips of every plan, one frame
    for (int pi = 0; pi < wa->n_plans; ++pi) {
        Walk3Plan& w = wa->plan[pi];
        w.strips_x = (w.out_w + kW3NC * px - 1) / (kW3NC * px);
        strip_rows += (long long)w.strips_x * w.out_h;
    }
    // (round 6: a launch that stays within a quarter of the chip even in 8-row segments -- a single camera frame -- is pure latency,
    // one block's walk of segment rows + 8: 8-row segments there, profiles/r06_experiments.txt 8)
    const long long min_rows = (strip_rows * n_frames + 7) / 8 * 4 <= resident ? 8 : 32;
    const long long target = std::max<long long>(min_rows, (2 * strip_rows * n_frames + 7 * resident - 1) / (7 * resident));   // rows per segment
    long long block0 = 0;
    for (int pi = 0; pi < wa->n_plans; ++pi) {
        Walk3Plan& w = wa->plan[pi];
        const int segs = (int)std::max<long long>(1, (w.out_h + target / 2) / target);
        int seg_rows = (w.out_h + segs - 1) / segs;
        seg_rows = (seg_rows + kWalkCH - 1) / kWalkCH * kWalkCH;
        w.seg_rows = seg_rows;
        w.segs_y = (w.out_h + seg_rows - 1) / seg_rows;
        w.block0 = (int)block0;
        block0 += (long long)w.strips_x * w.segs_y;
    }
    if (block0 * n_frames > 0x7fffffffll) return false;
    wa->blocks_per_frame = (int)block0;
    return true;
}

int launch_pyramid(silent_ctx* ctx, const char* who, const silent_pyramid_plan* plan, const float* frames,
                          int n_frames, float* pyr, hipStream_t s, bool with_unit, bool with_region) {
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
        for (int j = 0; j < 6; ++j) ft.wx[j] = ft.wy[j] = plan->unit_w[j];
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
#define PYR_STREAM(G_, L_) \
    hipLaunchKernelGGL((pyramid_stream_kernel<1, G_, L_>), dim3((unsigned)blocks), dim3(64 * kFusedWaves), 0, s, frames, pyr, ft, plan->stream)
        if (plan->stream_layout == 1) PYR_STREAM(7, 1);
        else if (plan->stream.G <= 4) PYR_STREAM(4, 0);
        else PYR_STREAM(7, 0);
#undef PYR_STREAM
    } else if (tab.C == 3 && plan->walk_pyr_ok && with_unit && with_region && !(kopts & 3u) && walk3_plan(ctx, plan, n_frames, &w3t)) {
        // single-read RGB pyramid (pyramid_walk3_kernel, silent_walk_rgb.h); PYRAMID knob bits 1 / 2: unit + region kernels
        // union plans: the inner levels' first / last output rows and columns -- the last blocks of the same launch (PYRAMID knob 8:
        // a launch of their own behind the walk)
        const long long bthreads = (long long)plan->walk_border.per_frame * 3 * n_frames;
        const long long bblocks = (bthreads + 255) / 256;
        const bool border_inside = bblocks > 0 && !(kopts & 8u);
        const long long wblocks = (long long)n_frames * w3t.blocks_per_frame + (border_inside ? bblocks : 0);
        if (wblocks > 0x7fffffffll || bblocks > 0x7fffffffll) return fail(ctx, SILENT_E_INVALID, std::string(who) + ": too many blocks for one launch");
        WalkBorderArgs wb;
        std::memset(&wb, 0, sizeof(wb));
        wb.first = 0x7fffffff;
        if (border_inside) {
            wb.first = (int)((long long)n_frames * w3t.blocks_per_frame);
            wb.n_frames = n_frames;
            wb.tab = tab;
            wb.bt = plan->walk_border;
        }
#define WALK3(G_, PX_) hipLaunchKernelGGL((pyramid_walk3_kernel<G_, PX_>), dim3((unsigned)wblocks), dim3(kW3Threads), 0, s, frames, pyr, w3t, wb)
#define WALK3_PX(PX_) case PX_: if (plan->walk_G <= 4) WALK3(4, PX_); else WALK3(7, PX_); break
        switch (plan->walk_px) { WALK3_PX(36); WALK3_PX(32); WALK3_PX(28); WALK3_PX(24); default: break; }   // (walk3_plan refuses any other)
#undef WALK3_PX
#undef WALK3
        if (bblocks > 0 && !border_inside)
            hipLaunchKernelGGL(pyramid_border_kernel<3>, dim3((unsigned)bblocks), dim3(256), 0, s, frames, pyr, tab, plan->walk_border, n_frames);
    } else if (tab.C == 1) {
        if (b_unit) hipLaunchKernelGGL(pyramid_unit_kernel<1>, dim3((unsigned)b_unit), dim3(256), 0, s, frames, pyr, tab);
        if (b_region) hipLaunchKernelGGL(pyramid_region_kernel<1>, dim3((unsigned)b_region), dim3(256), 0, s, frames, pyr, tab);
    } else {
        if (b_unit) hipLaunchKernelGGL(pyramid_unit_kernel<3>, dim3((unsigned)b_unit), dim3(256), 0, s, frames, pyr, tab);
        if (b_region) hipLaunchKernelGGL(pyramid_region_kernel<3>, dim3((unsigned)b_region), dim3(256), 0, s, frames, pyr, tab);
    }
    if (b_zero && tab.C == 1) hipLaunchKernelGGL(pyramid_zero_kernel<1>, dim3((unsigned)b_zero), dim3(256), 0, s, pyr, tab);
    else if (b_zero) hipLaunchKernelGGL(pyramid_zero_kernel<3>, dim3((unsigned)b_zero), dim3(256), 0, s, pyr, tab);
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
