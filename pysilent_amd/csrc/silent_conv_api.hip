// libsilent_hip.so -- generic SAME convolution and the regulator (silent_conv.h): apply_filter / rgc_filter / rgby_filter /
// regulate_tensor of the reference as single ops.
#include "silent_internal.h"
#include "silent_conv.h"

using namespace silent;

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
        if (uniform && !(ctx->tune[SILENT_TUNE_RGB] & 1u))
            return launch_regulate_sum(ctx, in, levels, n_levels, n_frames, blur_hwio, regulation_value, regulation_root, flat_policy, out,
                                       (hipStream_t)stream);
    }
    Epilogue ep{0u, 0.f, regulation_value, regulation_root, flat_policy};
    return conv_dispatch(ctx, "silent_regulate", in, levels, n_levels, n_frames, channels, blur_hwio, kh, kw, channels,
                         true, ep, out, (hipStream_t)stream);
} catch (...) {
    return on_exception(ctx, "silent_regulate_dev");
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
