// libsilent_hip.so -- the fused RGB chain (silent_rgb.h, silent_rgb2.h): rgc -> rgby -> stripe -> regulate -> line-end -> clip ->
// pad -> value in one launch, the host analysis that finds the kernels' structure in the actual weights, and the
// channel-uniform regulator.
#include "silent_internal.h"
#include "silent_rgb.h"
#include "silent_rgb2.h"

using namespace silent;

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
int rgb_chain_tile_height(const silent_ctx* ctx, const silent_extent* levels, int n_levels, int n_frames, bool* pair, bool with_extrema) {
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
        // resident tiles: the pair kernel runs 4 waves per SIMD, 3 where it also leaves the extrema and the value summary of the
        // keypoint tail (149 VGPRs; round 5: the model counted 4 there as well and picked 24-row tiles for the reference layout --
        // 2048 tiles on a chip that holds 1536; with 32 rows it is one round: step 0.357 -> 0.349 ms); the one-pixel kernel 5 tiles
        const long long resident = (pair_kernel ? (with_extrema ? 12ll : 16ll) / kRgb2Waves : 5ll) * ctx->n_cus;
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
        // Round 6: launches far below one round (a camera frame: 2 - 4 levels of 192 x 288 = a few hundred tiles on a chip that
        // holds 1536) are pure latency -- one wave's row walk, th + 14 steps of about a microsecond -- and the pair kernel has no
        // chunking that ties the tile height to kRgbTHMin: down to 4 rows while the launch stays within a QUARTER of a round
        // (measured on the application graph, 640 x 480: chain 44 -> 24 us, frame 0.2085 -> 0.1873 ms at 4 rows; 2 rows 0.1904,
        // 6: 0.1898, 8: 0.1918, 12: 0.2013; profiles/r06_experiments.txt 8).  Fuller launches keep kRgbTHMin: there the rows a
        // tile re-reads (14 per tile) cost bandwidth and issue slots, not just latency.
        for (int cand = pair_kernel ? 4 : kRgbTHMin; model && cand <= kRgbTHMax; cand += 2) {
            long long tiles = 0;
            for (int l = 0; l < n_levels; ++l)
                tiles += (long long)((levels[l].w + tw - 1) / tw) * ((levels[l].h + cand - 1) / cand);
            tiles *= n_frames;
            if (cand < kRgbTHMin && tiles * 4 > resident) continue;
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
int rgb_chain_launch(silent_ctx* ctx, const char* who, const float* pyr, const silent_extent* levels, int n_levels,
                            int n_frames, const silent_rgb_chain_params* p, float* orient_out, float* line_end_out,
                            float* value_out, unsigned* mm, bool* mm_done, silent_stream stream, const SumTab* st,
                            float* sum, int* nan_flags) {
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
        const int th = rgb_chain_tile_height(ctx, levels, n_levels, n_frames, &pair_kernel, mm != nullptr);
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
    // General blur: stage-per-launch composition through ping-pong temporaries in the context workspace, by the single-op
    // entry points of the other families (silent_conv_api.hip, silent_peaks_api.hip).
    const size_t n = (size_t)pyramid_px(levels, n_levels) * n_frames * 3;
    const size_t bytes = align_up(n * sizeof(float));
    TRY(workspace(ctx, (hipStream_t)stream, 3 * bytes));
    float* t0 = (float*)ctx->ws.p;
    float* t1 = (float*)((char*)ctx->ws.p + bytes);
    float* t2 = (float*)((char*)ctx->ws.p + 2 * bytes);
    TRY(silent_conv2d_same_dev(ctx, pyr, levels, n_levels, n_frames, 3, p->rgc, 3, 3, 3, SILENT_RELU, 0.f, t0, stream));
    TRY(silent_conv2d_same_dev(ctx, t0, levels, n_levels, n_frames, 3, p->rgby, 3, 3, 3, SILENT_RELU, 0.f, t1, stream));
    TRY(silent_conv2d_same_dev(ctx, t1, levels, n_levels, n_frames, 3, p->stripe, 3, 3, 3, SILENT_RELU, 0.f, t0, stream));
    float* orient = orient_out ? orient_out : t1;
    TRY(silent_regulate_dev(ctx, t0, levels, n_levels, n_frames, 3, p->blur, 7, 7, p->regulation_value, p->regulation_root, p->flat_policy,
                            orient, stream));
    if (!line_end_out && !value_out) return SILENT_OK;
    TRY(silent_conv2d_same_dev(ctx, orient, levels, n_levels, n_frames, 3, p->end, 3, 3, 3, SILENT_RELU | SILENT_CLIP, p->clip_hi, t0, stream));
    float* padded = line_end_out ? line_end_out : t2;
    TRY(silent_pad_inwards_dev(ctx, t0, levels, n_levels, n_frames, 3, p->pad, p->pad, p->pad, p->pad, padded, stream));
    if (value_out) TRY(silent_value_from_color_dev(ctx, padded, levels, n_levels, n_frames, 3, value_out, stream));
    return SILENT_OK;
}

int launch_regulate_sum(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels, int n_frames, const float* blur_hwio,
                        float regulation_value, float regulation_root, int flat_policy, float* out, hipStream_t s) {
    RegArgs a;
    long long blocks;
    TRY(build_level_tab(ctx, "silent_regulate", levels, n_levels, n_frames, kRegTW, kRegTH, &a.tab, &blocks));
    a.in = in;
    a.out = out;
    for (int t = 0; t < 49; ++t) a.blur[t] = blur_hwio[t * 9];
    a.rv = regulation_value;
    a.root = regulation_root;
    a.flat_policy = flat_policy;
    hipLaunchKernelGGL(regulate_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    return check_launch(ctx, "silent_regulate");
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
