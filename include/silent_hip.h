/* silent_hip.h -- C ABI of libsilent_hip.so: the MI355X (gfx950) implementation of pySILEnT's
 * scale-space center-surround / oriented line-end detector hot path.
 *
 * The reference (SimLeek/pySILEnT, /root/reference) has NO plugin / operator / FFI layer: its
 * boundary is a set of plain Python functions that build TensorFlow-1.x graph nodes
 * (SURVEY.md section 8b).  Each entry point below therefore cites the reference *function* it
 * replaces (path:line relative to /root/reference); the Python binding a maintainer would add
 * (ctypes) is shown in INTEGRATION.md and implemented in pysilent_amd/_lib.py.
 *
 * Conventions
 *   - Activations: float32, NHWC, "packed pyramid batch": n_frames consecutive pyramids, each
 *     pyramid = level 0 .. level n_levels-1 back to back, level l = h_l x w_l x C row-major.
 *     The reference's fixed-size layout [L, h, w, C] (util/zoom/from_image.py:47) is the special
 *     case n_frames = 1, all extents equal -- byte-identical to the NumPy array it feeds TF.
 *   - Kernels: float32, HWIO [kh, kw, C_in, C_out] (what tf.constant(k, dtype=tf.float32) holds).
 *   - Convolutions: stride 1, SAME zero padding (pad_before = (k-1)/2), cross-correlation.
 *   - Every function returns SILENT_OK (0) or a negative silent_status; the message is available
 *     from silent_last_error().  Nothing throws or aborts across the ABI.
 *   - Ownership: the caller owns every buffer; the library never keeps a caller pointer past
 *     return.  silent_ctx / silent_pyramid_plan are library-owned handles.
 *   - Threading: a silent_ctx is not thread-safe; use one per (host thread, GPU).  Several entry points keep
 *     temporaries (reduction slots, chunk counts, staged intermediates) in ONE workspace owned by the context.  The
 *     library enforces the ordering this needs: when a workspace-using *_dev call arrives on another stream than the
 *     previous one, that previous stream is drained first (correct, but serialising: streams that should overlap need
 *     a context each).  The entry points run on the context's device and restore the caller's current HIP device.
 *   - Entry points without suffix take HOST pointers and are synchronous (they stage through the
 *     context's device arena).  The *_dev twins take DEVICE pointers plus a hipStream_t (passed as
 *     void*; NULL = the legacy default stream), are stream-ordered and do not synchronise.
 */
#ifndef SILENT_HIP_H
#define SILENT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SILENT_ABI_VERSION 4
#define SILENT_MAX_LEVELS 16
#define SILENT_MAX_KERNEL_FLOATS 784 /* kh*kw*C_in*C_out limit (weights travel as kernel arguments) */

typedef enum silent_status {
    SILENT_OK = 0,
    SILENT_E_INVALID = -1,     /* bad argument (NULL pointer, extent <= 0, channel mismatch ...)  -> ValueError */
    SILENT_E_HIP = -2,         /* a HIP runtime call failed                                       -> RuntimeError */
    SILENT_E_CAPACITY = -3,    /* caller-provided index buffer too small; counts hold the need    -> ValueError */
    SILENT_E_UNSUPPORTED = -4, /* shape outside the compiled set                                  -> ValueError */
    SILENT_E_NOMEM = -5        /* device allocation failed                                        -> MemoryError */
} silent_status;

typedef struct silent_ctx silent_ctx;
typedef struct silent_pyramid_plan silent_pyramid_plan;
typedef void* silent_stream; /* hipStream_t */

typedef struct silent_extent {
    int32_t h, w;
} silent_extent;

/* conv flags */
#define SILENT_RELU 1u /* tf.maximum(x, [0])            filters/rgc.py:13-16 */
#define SILENT_CLIP 2u /* tf.clip_by_value(x, 0, hi)    recognition_testing.py:74 */

/* regulate_tensor flat-region policy (hard part 3) */
#define SILENT_FLAT_IEEE 0 /* literal: 0 * (rv / pow(0, root)) = NaN */
#define SILENT_FLAT_ZERO 1 /* output 0 where the input is exactly 0 */

/* nms modes */
#define SILENT_NMS_PRODUCT 0 /* x * where(x == maxpool3x3(x), x, 0)   _experimental/vision_filter.py:88-89 */
#define SILENT_NMS_FIRED 1   /* where(x == maxpool3x3(x), 1, 0)       util/energy/boosting.py:18-22 */

/* ---------------------------------------------------------------------------- context */

int silent_abi_version(void);
/* Number of HIP devices visible (0 and SILENT_E_HIP when the runtime cannot initialise). */
int silent_device_count(int* count);
int silent_create(int device, silent_ctx** out);
void silent_destroy(silent_ctx* ctx);
/* Last error text of this context; ctx may be NULL (then: the last error of a failed silent_create). */
const char* silent_last_error(const silent_ctx* ctx);
int silent_device_name(const silent_ctx* ctx, char* buf, size_t len);

/* Device-memory helpers so that a host without PyTorch can stay device-resident. */
int silent_malloc(silent_ctx* ctx, size_t bytes, void** dptr);
int silent_free(silent_ctx* ctx, void* dptr);
int silent_memcpy_h2d(silent_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes, silent_stream stream);
int silent_memcpy_d2h(silent_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes, silent_stream stream);
/* n device buffers -> ONE host buffer, back to back, with a single synchronisation at the end: the fetch of
 * session.run([t0 .. t5]) (slam_recognition/recognition_testing.py:132) for results that live on the device. */
int silent_gather_d2h(silent_ctx* ctx, void* dst_host, const void* const* src_dev, const size_t* bytes, int n,
                      silent_stream stream);
int silent_synchronize(silent_ctx* ctx, silent_stream stream);
/* Occupies `stream` for about `microseconds` (one wavefront polling the constant-rate clock; <= 1 s): the probe a host
 * uses to find out whether two streams really run side by side (pysilent_amd.pipeline.pick_concurrent_stream).  No
 * counterpart in the reference (it has one stream: tf.device('/device:GPU:0'), recognition_testing.py:64). */
int silent_busy_wait_dev(silent_ctx* ctx, unsigned microseconds, silent_stream stream);
/* One empty kernel named trace_marker_kernel on `stream`: a mark in a rocprofv3 kernel trace (bench.py puts it between its tuners
 * and the measured steps; scripts/summarize_profile.py reads the statistics behind it).  No counterpart in the reference. */
int silent_trace_marker_dev(silent_ctx* ctx, silent_stream stream);

/* ---------------------------------------------------------------------------- a-1 pyramid
 * Replaces image_to_zoom_tensor, slam_recognition/util/zoom/from_image.py:10-69: per level a crop of
 * the frame resampled by scipy.ndimage.zoom(plane, z, prefilter=False, order=5) (from_image.py:55-59:
 * un-prefiltered quintic B-spline, output o -> o*(in-1)/(out-1), mirror tap extension, mode
 * 'constant' out-of-range rule) and copied to the top-left of a canvas (from_image.py:61-64).
 * Canvas pixels the zoomed crop does not cover are uninitialised in the reference; here they are 0.
 * The HOST decides crop and zoom extents (Python's round() is part of scipy's contract) and passes
 * them explicitly; "classic" whole-frame levels are the case crop = frame, out = zoom extents. */
typedef struct silent_pyr_level {
    int32_t src_y0, src_x0, src_h, src_w; /* crop of the frame that the resampler sees        */
    int32_t zoom_h, zoom_w;               /* resampler output extents = round(src * zoom)     */
    int32_t out_h, out_w;                 /* canvas (level) extents; copied region = min(zoom, out) */
} silent_pyr_level;

int silent_pyramid_plan_create(silent_ctx* ctx, int frame_h, int frame_w, int channels,
                               const silent_pyr_level* levels, int n_levels, silent_pyramid_plan** out);
void silent_pyramid_plan_destroy(silent_pyramid_plan* plan);
/* frames: n_frames x [H, W, C] float32 (values as the reference feeds them: uint8 range cast to f32,
 * recognition_testing.py:141).  pyr: packed pyramid batch with the plan's canvas extents. */
int silent_pyramid(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames, int n_frames, float* pyr);
int silent_pyramid_dev(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames, int n_frames,
                       float* pyr, silent_stream stream);

/* ---------------------------------------------------------------------------- a-2..a-5 convolution
 * Replaces the tf.nn.conv2d(+tf.maximum(+tf.clip_by_value)) call sites:
 *   rgc_filter          slam_recognition/filters/rgc.py:6-18
 *   rgby_filter         slam_recognition/filters/rgby.py:6-14
 *   orientation_filter  slam_recognition/filters/orientation.py:17-29 (the stripe conv)
 *   apply_filter        slam_recognition/util/apply_filter.py:4-7 (+ relu/clip recognition_testing.py:73-74)
 * flags: SILENT_RELU | SILENT_CLIP (clip to [0, clip_hi]); NaN handling is Eigen's (x < 0 ? 0 : x). */
int silent_conv2d_same(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels, int n_frames,
                       int c_in, const float* kernel_hwio, int kh, int kw, int c_out, unsigned flags,
                       float clip_hi, float* out);
int silent_conv2d_same_dev(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                           int n_frames, int c_in, const float* kernel_hwio, int kh, int kw, int c_out,
                           unsigned flags, float clip_hi, float* out, silent_stream stream);

/* ---------------------------------------------------------------------------- fused grayscale pass
 * BASELINE configs 1/2/5: per level
 *     cs  = relu(conv3x3(x,  cs_kernel[3,3,1,1]))
 *     end = clip(relu(conv3x3(cs, end_bank[3,3,1,K])), 0, clip_hi)
 * i.e. the reference chain order recognition_testing.py:69-74 restricted to one input channel, in one
 * kernel launch (the CS map never leaves LDS between the two convolutions).  K in {3, 4, 8}.
 * cs_out: packed [.., h_l, w_l, 1]; end_out: packed [.., h_l, w_l, K]; either may be NULL to skip it. */
int silent_gray_line_end(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels, int n_frames,
                         const float* cs_kernel, const float* end_bank, int n_orient, float clip_hi,
                         float* cs_out, float* end_out);
int silent_gray_line_end_dev(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels,
                             int n_frames, const float* cs_kernel, const float* end_bank, int n_orient,
                             float clip_hi, float* cs_out, float* end_out, silent_stream stream);

/* ---------------------------------------------------------------------------- whole grayscale hot path
 * frames -> zoom pyramid -> CS -> ReLU -> K-orientation end bank -> ReLU -> clip in ONE call: the op order of
 * LineEndDisplayer.callback + compile (recognition_testing.py:136-144, :69-74) restricted to one channel.
 * Results are identical to silent_pyramid_dev followed by silent_gray_line_end_dev; the difference is traffic:
 * levels whose zoom factor is exactly 1 (level 0, 75 % of the pixels) are smoothed, center-surround filtered
 * and line-end filtered in one kernel, so that level is written once and never re-read.
 * pyr / cs_out / end_out: packed pyramid batches with the plan's extents (1, 1 and K channels); pyr is
 * always written (it is an output of the reference's from_image); cs_out or end_out may be NULL. */
int silent_gray_pass(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames, int n_frames,
                     const float* cs_kernel, const float* end_bank, int n_orient, float clip_hi, float* pyr,
                     float* cs_out, float* end_out);
int silent_gray_pass_dev(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames, int n_frames,
                         const float* cs_kernel, const float* end_bank, int n_orient, float clip_hi, float* pyr,
                         float* cs_out, float* end_out, silent_stream stream);
/* The same pass in two halves, for callers that overlap consecutive batches on two streams (the reference runs one frame per
 * session.run, recognition_testing.py:132; a batched caller can run the second half of batch n beside the first half of batch
 * n + 1).  parts bit 0 (SILENT_GRAY_PART_PYRAMID): the pyramid of every level + CS / end of the unit-zoom levels -- reads
 * frames, writes pyr and the unit levels of cs_out / end_out; bit 1 (SILENT_GRAY_PART_FILTER): CS + end of the remaining
 * levels -- reads pyr (as written by part 0 for the same frames), writes the other levels of cs_out / end_out.  parts = 3 is
 * silent_gray_pass_dev.  The two parts of one batch must be ordered by the caller (same stream, or an event). */
#define SILENT_GRAY_PART_PYRAMID 1u
#define SILENT_GRAY_PART_FILTER 2u
int silent_gray_pass_parts_dev(silent_ctx* ctx, const silent_pyramid_plan* plan, const float* frames, int n_frames,
                               const float* cs_kernel, const float* end_bank, int n_orient, float clip_hi, float* pyr,
                               float* cs_out, float* end_out, unsigned parts, silent_stream stream);

/* 1 when silent_gray_pass runs this plan through the single-read stream kernel (one unit-zoom level and every
 * other level resampling the same crop with a step > 1.25: classic whole-frame pyramids), 0 when it falls
 * back to region + unit-fused + filter kernels (e.g. the reference's centred-crop layout). */
int silent_pyramid_plan_is_streamable(const silent_pyramid_plan* plan);

/* 3-channel plans: the number of WALK PLANS silent_pyramid runs this pyramid with in ONE launch of the strip-walk kernel (frame rows
 * DMA'd once per crop into an LDS ring): 1 for a classic whole-frame pyramid (unit level + every other level on the same crop), one
 * per level for crop layouts such as the reference's (image_to_zoom_tensor, util/zoom/from_image.py:45-64: nested centre crops); 0
 * when the plan falls back to the unit + region kernels (more than 8 levels of a crop layout; more than 21 outputs per wave tile:
 * zoom steps below 1.6; frame width not a multiple of 4).  pixels_per_wave (may be NULL): 36 or 32. */
int silent_pyramid_plan_walk_plans(const silent_pyramid_plan* plan, int* pixels_per_wave);

/* Optional HIP-event timing of the DOMINANT kernel of the last silent_gray_pass_dev call (the fused
 * unit-level kernel): silent_set_profiling(ctx, n) with n >= 1 brackets that kernel with an event pair on every
 * n-th call (n = 0: off), on the stream the kernel runs on -- an event is a packet in the queue, so bracketing every
 * call costs a few per cent of a millisecond-long pass.  silent_profile_elapsed_ms synchronises and returns the MEAN
 * over the (up to 8 most recent) recorded pairs since the last silent_set_profiling, and the number of
 * level-0-class pixels the kernel processed per launch. */
int silent_set_profiling(silent_ctx* ctx, int enable);
int silent_profile_elapsed_ms(silent_ctx* ctx, float* ms, int64_t* pixels);

/* Kernel-selection knobs for tests and A/B timing (no reference counterpart: the reference has one code path).
 * Every context starts from the environment variables SILENT_GRAY_OPTS / SILENT_RGB_OPTS / SILENT_PYRAMID_OPTS, read
 * once in silent_create; silent_set_tuning changes a knob of one context afterwards.  Bits -- GRAY: 1 XCD-aware tile
 * order, 16 no single-read stream kernel, 32 XCD order in the stream kernel (2 and 8 selected 32-row tiles until round 5: slower in
 * every A/B, the instantiations are gone and the bits are ignored);
 * RGB: 1 dense weights, 2 no two-group form, 8 no short tiles, 16 the one-pixel-per-lane chain kernel (default: two pixels
 * per lane on packed f32, same bits), 32 no sparse keypoint tail, 64 no symmetric forms (the two-group instantiation of the pair
 * kernel, bit-identical to the one-pixel kernel), 128 16-byte stores of orient / line_end where rows are 16-byte aligned (same bits as the default
 * 12-byte form, same speed), bits 8-15 tile height / 2; PYRAMID: 1 no single-read pyramid
 * (gray stream kernel, RGB strip walk), 2 no RGB strip walk.  All variants give the same results (bit-identical, or within
 * the re-association tolerance for the RGB forms); the defaults are the fastest measured. */
#define SILENT_TUNE_GRAY 0
#define SILENT_TUNE_RGB 1
#define SILENT_TUNE_PYRAMID 2
#define SILENT_TUNE_COUNT 3
int silent_set_tuning(silent_ctx* ctx, int which, unsigned value);
int silent_get_tuning(const silent_ctx* ctx, int which, unsigned* value);

/* ---------------------------------------------------------------------------- a-6 regulator
 * Replaces regulate_tensor, slam_recognition/util/regulator/gaussian_regulator_tensor.py:10-36:
 *   y = x * (rv / pow(min(conv(x, blur), 1), root)); blur is HWIO [kh, kw, C, C]. */
int silent_regulate(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels, int n_frames,
                    int channels, const float* blur_hwio, int kh, int kw, float regulation_value,
                    float regulation_root, int flat_policy, float* out);
int silent_regulate_dev(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels, int n_frames,
                        int channels, const float* blur_hwio, int kh, int kw, float regulation_value,
                        float regulation_root, int flat_policy, float* out, silent_stream stream);

/* ---------------------------------------------------------------------------- a-7 / a-8 pointwise
 * pad_inwards            slam_recognition/util/selection/isolate_rectangle.py:19-23 (multiply by a 0/1 mask)
 * get_value_from_color   slam_recognition/util/color/get_value.py:6-12 (channel sum * float32(1/C)) */
int silent_pad_inwards(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels, int n_frames,
                       int channels, int pad_top, int pad_bottom, int pad_left, int pad_right, float* out);
int silent_pad_inwards_dev(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                           int n_frames, int channels, int pad_top, int pad_bottom, int pad_left, int pad_right,
                           float* out, silent_stream stream);
int silent_value_from_color(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                            int n_frames, int channels, float* out);
int silent_value_from_color_dev(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                                int n_frames, int channels, float* out, silent_stream stream);
/* get_bw_from_color      slam_recognition/util/color/get_bw.py:6-13: 1 where the channel sum (x . ones, summed left to
 * right in float32) is not 0 -- NaN counts as not 0, like tf.not_equal -- else 0; one output channel. */
int silent_bw_from_color(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels, int n_frames,
                         int channels, float* out);
int silent_bw_from_color_dev(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels,
                             int n_frames, int channels, float* out, silent_stream stream);

/* ---------------------------------------------------------------------------- a-9 non-max suppression */
int silent_nms3x3(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels, int n_frames,
                  int channels, int mode, float* out);
int silent_nms3x3_dev(silent_ctx* ctx, const float* in, const silent_extent* levels, int n_levels, int n_frames,
                      int channels, int mode, float* out, silent_stream stream);

/* ---------------------------------------------------------------------------- a-10 top-percent threshold
 * Replaces top_value_points, slam_recognition/util/selection/top_value_points.py:8-29: per level
 * (= per batch item of the reference) thr = (1-p)*max(value) + p*min(value), out = color * (value >= thr).
 * value: packed 1-channel map (NULL -> computed from color as get_value_from_color does). */
int silent_top_value_points(silent_ctx* ctx, const float* color, const float* value, const silent_extent* levels,
                            int n_levels, int n_frames, int channels, double top_percent, float* out);
int silent_top_value_points_dev(silent_ctx* ctx, const float* color, const float* value,
                                const silent_extent* levels, int n_levels, int n_frames, int channels,
                                double top_percent, float* out, silent_stream stream);

/* ---------------------------------------------------------------------------- a-11 keypoint indices
 * Replaces max_value_indices_region, slam_recognition/util/selection/top_value_points.py:32-45:
 * tf.where(value >= resize_nearest(max_pool(value, k=(H,W), strides=(rH,rW), SAME))) with the TF1
 * SAME / NEAREST index rules.  regions[l] = (rH, rW) of level l, any extents >= 1: up to
 * 4 x 4 windows per level (the reference uses 2 x 2) run a one-pass cell-maximum path, more windows the separable
 * prefix / suffix path (any level width: the row pass walks a row in chunks); same results either way.
 * idx: n_frames x cap_per_frame x 4 int64 rows (level, y, x, 0), ROW-MAJOR SORTED within a frame like
 * tf.where; counts[f] = number of rows frame f produced.  If any count exceeds cap_per_frame only the
 * first cap_per_frame rows of that frame are written and the host call returns SILENT_E_CAPACITY.
 * The _dev twin leaves counts on the device and cannot report capacity: check counts yourself. */
int silent_max_value_indices_region(silent_ctx* ctx, const float* value, const silent_extent* levels, int n_levels,
                                    int n_frames, const silent_extent* regions, int64_t* idx,
                                    size_t cap_per_frame, int64_t* counts);
int silent_max_value_indices_region_dev(silent_ctx* ctx, const float* value, const silent_extent* levels,
                                        int n_levels, int n_frames, const silent_extent* regions, int64_t* idx,
                                        size_t cap_per_frame, int64_t* counts, silent_stream stream);

/* ---------------------------------------------------------------------------- fused selection
 * top_value_points (util/selection/top_value_points.py:8-29) -> 3x3 NMS in product form
 * (_experimental/vision_filter.py:88-89) -> get_value_from_color (util/color/get_value.py:6-12) in one streaming
 * pass after the per-level max / min reduction: bit-identical to silent_top_value_points + silent_nms3x3 +
 * silent_value_from_color, without the two intermediate colour maps in HBM.  value may be NULL (computed from color).
 * Any of top_out [C ch], peaks_out [C ch], peak_value_out [1 ch] may be NULL, not all.  channels: 1 or 3. */
int silent_select_peaks(silent_ctx* ctx, const float* color, const float* value, const silent_extent* levels,
                        int n_levels, int n_frames, int channels, double top_percent, float* top_out,
                        float* peaks_out, float* peak_value_out);
int silent_select_peaks_dev(silent_ctx* ctx, const float* color, const float* value, const silent_extent* levels,
                            int n_levels, int n_frames, int channels, double top_percent, float* top_out,
                            float* peaks_out, float* peak_value_out, silent_stream stream);

/* SURVEY 8d config 3 in one call: silent_select_peaks followed by silent_max_value_indices_region on its peak value
 * (a-10 -> a-9 -> a-8 -> a-11), with the cell maxima of the keypoint search folded into the selection pass (one pass
 * over the value map less).  Outputs exactly as those two calls: peak_value_out [1 ch] (may be NULL: the map then lives in
 * the context workspace), idx / counts as silent_max_value_indices_region. */
int silent_select_keypoints(silent_ctx* ctx, const float* color, const float* value, const silent_extent* levels,
                            int n_levels, int n_frames, int channels, double top_percent,
                            const silent_extent* regions, float* peak_value_out, int64_t* idx, size_t cap_per_frame,
                            int64_t* counts);
int silent_select_keypoints_dev(silent_ctx* ctx, const float* color, const float* value, const silent_extent* levels,
                                int n_levels, int n_frames, int channels, double top_percent,
                                const silent_extent* regions, float* peak_value_out, int64_t* idx,
                                size_t cap_per_frame, int64_t* counts, silent_stream stream);

/* ---------------------------------------------------------------------------- centroids
 * Replaces get_centroids, slam_recognition/util/centroids.py:21-46 (with index_tensor.from_shape,
 * util/index_tensor.py:7-20, dimensions reversed: channel 0 = x, channel 1 = y): per cell of region_h x
 * region_w pixels (window = stride, TF SAME geometry) the value-weighted centroid; every pixel gets the L1
 * distance to the centroid of its (nearest-neighbour) cell.  Empty cells give 0/0 = NaN like the reference.
 * value: packed 1-channel maps.  dist_out: same geometry.  total_out: packed cell maps, level l has
 * ceil(h_l / region_h) x ceil(w_l / region_w) cells (the reference's total_pool). */
int silent_centroids(silent_ctx* ctx, const float* value, const silent_extent* levels, int n_levels, int n_frames,
                     int region_h, int region_w, float* dist_out, float* total_out);
int silent_centroids_dev(silent_ctx* ctx, const float* value, const silent_extent* levels, int n_levels,
                         int n_frames, int region_h, int region_w, float* dist_out, float* total_out,
                         silent_stream stream);

/* ---------------------------------------------------------------------------- boosting state
 * Replaces get_boosting, slam_recognition/util/energy/boosting.py:10-42 with generate_recovery,
 * util/energy/recovery.py:4-22.  The state ("exhaustion_tensor", a tf.Variable in the reference,
 * recognition_testing.py:56) is a caller-owned float buffer with the geometry of the input, updated IN PLACE:
 *     m      = input ** energy                      (float64 pow rounded to float32)
 *     fired  = (m == maxpool3x3 SAME (m)) ? 1 : 0
 *     energy = clip((energy*255 - fired*255 + recovery) / 255, -exhaustion_max, +excitation_max)
 * Saving / restoring a stream is a copy of that buffer (initial value: 8 everywhere, boosting.py:6-7).  Each frame
 * slot of the batch is its own stream; successive calls advance every stream by one step, so the frames of one
 * stream must come to the same slot, in order.
 * visualize = 0: fired_out and energy_out (may be NULL) have 1 channel (has_fired, new energy).
 * visualize = 1 (for_visualizing, boosting.py:35-40): both have 3 identical channels, fired * input and
 *                energy * 255/(exhaustion_max+excitation_max) + 255*excitation_max/(exhaustion_max+excitation_max). */
#define SILENT_RECOVERY_CONSTANT 1u
#define SILENT_RECOVERY_INPUT 2u

typedef struct silent_boosting_params {
    float exhaustion_max;      /* 1 */
    float excitation_max;      /* 1 */
    unsigned recovery_mode;    /* SILENT_RECOVERY_CONSTANT (reference default) | SILENT_RECOVERY_INPUT */
    float recovery_amount;     /* 10   (recovery.py:4) */
    float recovery_percentage; /* 0.8  (recovery.py:8) */
    int32_t visualize;
} silent_boosting_params;

int silent_boosting_step(silent_ctx* ctx, const float* input, const silent_extent* levels, int n_levels,
                         int n_frames, const silent_boosting_params* params, float* energy, float* fired_out,
                         float* energy_out);
int silent_boosting_step_dev(silent_ctx* ctx, const float* input, const silent_extent* levels, int n_levels,
                             int n_frames, const silent_boosting_params* params, float* energy, float* fired_out,
                             float* energy_out, silent_stream stream);

/* ---------------------------------------------------------------------------- display-graph glue
 * The scalar ops between the kernels of LineEndDisplayer.compile, recognition_testing.py:79-81 and :99
 * (``t / 255.0``, ``clip_by_value(t * (255 / 4.0), 1, 256) - 1``, ``255 - t * 255``, ``t * 255``):
 *     out[i] = clip(in[i] * mul / div + add, lo, hi) + post_add        (float32, one rounding per operation;
 * mul = div = 1, add = post_add = 0, lo = -INFINITY, hi = +INFINITY are the neutral values).  in == out is allowed. */
typedef struct silent_affine_params {
    float mul, div, add, lo, hi, post_add;
} silent_affine_params;

int silent_affine_clip(silent_ctx* ctx, const float* in, size_t n_values, const silent_affine_params* params,
                       float* out);
int silent_affine_clip_dev(silent_ctx* ctx, const float* in, size_t n_values, const silent_affine_params* params,
                           float* out, silent_stream stream);

/* np.asarray(frame, dtype=np.float32) (slam_recognition/recognition_testing.py:141: the camera's uint8 frame becomes the
 * float32 tensor the pyramid takes) and the colour-plane slicing of image_to_zoom_tensor (util/zoom/from_image.py:54-64,
 * one scipy zoom per plane) as ONE strided cast on the device:
 *     out[p * out_stride + out_offset + k] = (float)in[p * in_stride + in_offset + k]      p < n_pixels, k < count
 * (strides / offsets in ELEMENTS).  in_dtype: SILENT_DT_*.  Widening (strides = count = channels), cutting plane c out of
 * an interleaved image (in_stride = C, in_offset = c, count = 1, out_stride = 1) and interleaving planes back are all
 * instances.  in and out must not overlap.  The conversion is the C cast: exact for uint8 / int16 / uint16 and for integers up to
 * 2^24 in magnitude, round-to-nearest-even beyond (int32 / int64 / float64), like numpy's astype(float32).  Buffer sizes: `in`
 * holds at least (n_pixels - 1) * in_stride + in_offset + count elements, `out` (n_pixels - 1) * out_stride + out_offset + count
 * floats -- the last pixel needs no whole stride, and the host form copies exactly that much. */
#define SILENT_DT_U8 0
#define SILENT_DT_F32 1
#define SILENT_DT_F64 2
#define SILENT_DT_I32 3
#define SILENT_DT_U16 4
#define SILENT_DT_I16 5
#define SILENT_DT_I64 6
int silent_cast_interleave(silent_ctx* ctx, const void* in, int in_dtype, size_t n_pixels, int in_stride, int in_offset,
                           int count, float* out, int out_stride, int out_offset);
int silent_cast_interleave_dev(silent_ctx* ctx, const void* in, int in_dtype, size_t n_pixels, int in_stride, int in_offset,
                               int count, float* out, int out_stride, int out_offset, silent_stream stream);

/* tf.image.resize_nearest_neighbor with the TF1 defaults (align_corners = False), recognition_testing.py:82-83
 * and util/centroids.py:41: out(y, x) = in(min(floor(y * float32(in_h / out_h)), in_h - 1), ...).
 * in: packed maps with extents in_levels; out: packed maps with extents out_levels (same level count). */
int silent_resize_nearest(silent_ctx* ctx, const float* in, const silent_extent* in_levels, int n_levels,
                          int n_frames, int channels, const silent_extent* out_levels, float* out);
int silent_resize_nearest_dev(silent_ctx* ctx, const float* in, const silent_extent* in_levels, int n_levels,
                              int n_frames, int channels, const silent_extent* out_levels, float* out,
                              silent_stream stream);

/* ---------------------------------------------------------------------------- fused RGB chain
 * The reference graph recognition_testing.py:69-77 on 3-channel levels.  With a channel-uniform blur (what
 * blur_tensor generates) this is ONE fused launch that reads the pyramid once; any other blur runs the
 * stages as separate launches through workspace temporaries:
 *   orient   = regulate(relu(conv(relu(conv(relu(conv(x, rgc)), rgby)), stripe)), blur, rv, root)   (:69-71)
 *   line_end = pad_inwards(clip(relu(conv(orient, end)), 0, clip_hi), pad)                          (:73-75)
 *   value    = get_value_from_color(line_end)                                                        (:77)
 * Every kernel is HWIO [3,3,3,3] except blur [7,7,3,3].  Any of the three outputs may be NULL. */
typedef struct silent_rgb_chain_params {
    const float* rgc;    /* midget_rgc(2)              filters/rgc.py:9      */
    const float* rgby;   /* rgby_3(2)                  filters/rgby.py:9     */
    const float* stripe; /* rgb_2d_stripe_tensors()    filters/orientation.py:19 */
    const float* blur;   /* blur_tensor(2, 7)          filters/orientation.py:20 */
    const float* end;    /* rgb_2d_end_tensors()       recognition_testing.py:29 */
    float regulation_value, regulation_root; /* 1.0, 0.1   filters/orientation.py:33 */
    int32_t flat_policy;
    float clip_hi;       /* 255                        recognition_testing.py:74 */
    int32_t pad;         /* 2                          recognition_testing.py:75 */
} silent_rgb_chain_params;

/* Host-only: which structure silent_rgb_line_end finds in the weights (no GPU needed).  flags: bit 0 rgc is
 * channel-diagonal, bit 1 stripe does not depend on the input channel, bit 2 rgby / bit 3 end are "two-group" kernels
 * (every tap vector K[t][i][:] a multiple of one of two vectors), bit 4 every channel of rgc is mirror-symmetric in both axes,
 * bit 5 rgby = S (x) A around the centre + B at the centre with a profile S that is mirror-symmetric in both axes.  masks (may
 * be NULL): 6 words, the group-A tap masks (bit dy * 3 + dx) of rgby and end per input channel. */
int silent_rgb_chain_structure(const silent_rgb_chain_params* params, unsigned* flags, unsigned* masks);

/* Config 3 from the pyramid on in ONE call (recognition_testing.py:69-77 followed by a-10 -> a-9 -> a-8 -> a-11 on its result):
 * = silent_rgb_line_end, then silent_select_keypoints(color = line_end, value = its value map, channels = 3).  Same outputs,
 * bit for bit, as those two calls.  The fused chain kernel accumulates the per-level max / min of the value map that a-10's
 * threshold needs (top_value_points.py:16-27) while it writes line_end, so the separate reduction pass and -- unless value_out
 * is given -- the value map itself are never moved through memory.  orient_out / value_out may be NULL; the host form also
 * accepts NULL line_end_out; idx / cap_per_frame / counts as silent_max_value_indices_region.
 * peak_value_out may be NULL (both forms): nobody then needs the selection's value map as a MAP.  Either way the tail runs sparse --
 * the chain kernel leaves max_pool(value) per (pixel pair x 16 rows), and a-10 / a-9 / a-8 / a-11 are evaluated only around
 * the pixels that reach their level's threshold (a handful per level on natural and noise frames).  A search window
 * without a positive peak makes every non-NaN pixel mapped to it a keypoint: in a level without NaNs (the chain kernel
 * notes them) the count pass synthesises that; with NaNs, or with > 16384 candidates in a frame, the (frame, level) runs
 * the dense kernels on a map in the context workspace.  With peak_value_out given, the map is zero-filled and receives the
 * evaluated pixels' values; levels that hold NaNs run the dense pass (a NaN pixel's peak value is a NaN).  Keypoints and map are
 * identical either way (tested). */
int silent_rgb_keypoints(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels, int n_frames,
                         const silent_rgb_chain_params* params, double top_percent, const silent_extent* regions,
                         float* orient_out, float* line_end_out, float* value_out, float* peak_value_out, int64_t* idx,
                         size_t cap_per_frame, int64_t* counts);
int silent_rgb_keypoints_dev(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels, int n_frames,
                             const silent_rgb_chain_params* params, double top_percent, const silent_extent* regions,
                             float* orient_out, float* line_end_out, float* value_out, float* peak_value_out, int64_t* idx,
                             size_t cap_per_frame, int64_t* counts, silent_stream stream);

/* What the sparse tail of the LAST silent_rgb_keypoints[_dev] call of this context did (synchronises that call's stream):
 * stats[0] = 1 if it ran sparse, [1] = (frame, level) pairs, [2] = pairs it handed to the dense kernels, [3] = candidate
 * pixels (value >= threshold) it evaluated, [4] = pairs settled with a synthesised all-zero map (a search window without a
 * positive peak in a level without NaNs: every pixel mapped to it is a keypoint).  stats holds 5 values.  For tests and the
 * bench report; no reference counterpart. */
int silent_sparse_tail_stats(silent_ctx* ctx, int64_t* stats);

/* Host-only (no GPU needed): the weight STREAM silent_rgb_line_end hands to its pair kernel for these weights -- the
 * weights in the order the kernel consumes them (csrc/silent_rgb2.h), zero-padded to whole pairs of 16-float blocks.
 * knobs: SILENT_TUNE_RGB bits 0 / 1 / 6.  variant: 3 symmetric forms (rgc folded over both mirror axes, rgby as channel mix ->
 * one profile -> centre mix, the stripe bank as left / right SGPR pairs), 2 two-group, 1 basic (diagonal rgc + channel-sum
 * stripe), 0 dense; n_used = 149 / 160 / 265 / 373 (two-group and symmetric: the mirror-symmetric blur travels as 16 folded
 * weights).  stream must hold SILENT_RGB_STREAM_MAX floats.  For tests of the host logic. */
#define SILENT_RGB_STREAM_MAX 384
int silent_rgb_chain_stream(const silent_rgb_chain_params* params, unsigned knobs, float* stream, int* n_used, int* variant);
int silent_rgb_line_end(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels, int n_frames,
                        const silent_rgb_chain_params* params, float* orient_out, float* line_end_out,
                        float* value_out);
int silent_rgb_line_end_dev(silent_ctx* ctx, const float* pyr, const silent_extent* levels, int n_levels,
                            int n_frames, const silent_rgb_chain_params* params, float* orient_out,
                            float* line_end_out, float* value_out, silent_stream stream);

/* ---------------------------------------------------------------------------- one camera frame, one call
 * LineEndDisplayer.callback + run, slam_recognition/recognition_testing.py:106-144: a frame in host memory ->
 * np.asarray(frame, float32) (:141) -> zoom.from_image(frame, 3, output_size, zoom_ratio) (:142) -> the graph of compile()
 * (:60-100) -> the six tensors session.run fetches (:99-100, :132), in host memory:
 *   0 orient [L,h,w,3]   1 255 - centroids * 255 [L,h,w,1]   2 255 - centroids2 * 255 [L,h2,w2,1] (h2 = int(h / e ** .5))
 *   3 fired * 255 [L,ch,cw,C]   4 update_importances [L,ch,cw,C]   5 padded line_end [L,h,w,3]      (ch = ceil(h / region_h);
 *                                                                                      C = 3 with boosting.visualize, else 1)
 * The reference pays a feed, a session.run and six fetches per frame; the displayer replays ONE HIP graph per frame (five kernels
 * that read the pinned input buffer and write the pinned result slot themselves: no copy node) on a stream of its own and owns every buffer, a private context and the boosting state
 * (energy_values, :56; 8 everywhere at creation).  levels: the host's geometry of image_to_zoom_tensor (from_image.py:45-64),
 * every level on the same canvas extent.  Not thread-safe; frames of one camera must come to one displayer, in order. */
typedef struct silent_displayer silent_displayer;
typedef struct silent_displayer_params {
    int32_t frame_h, frame_w;        /* camera frame [frame_h, frame_w, 3] */
    int32_t frame_dtype;             /* SILENT_DT_U8 (a camera's) ... SILENT_DT_F32 */
    int32_t centroid_region_h, centroid_region_w;   /* centroid_region_shape[1:3] = 3, 3 (recognition_testing.py:38) */
    silent_rgb_chain_params chain;   /* kernels are copied at creation */
    silent_boosting_params boosting; /* get_boosting(..., 1, 1, recovery, True): visualize = 1 in the reference (:86) */
} silent_displayer_params;
int silent_displayer_create(silent_ctx* ctx, const silent_displayer_params* params, const silent_pyr_level* levels, int n_levels,
                            silent_displayer** out);
/* Lifetime: `ctx` (errors of the displayer's calls are reported there) must outlive every silent_displayer_step / _input /
 * _get_state / _set_state call; silent_displayer_destroy itself does not touch it (it may run after silent_destroy(ctx), e.g. at
 * interpreter shutdown).  destroy frees the pinned input buffer and both result slots: pointers handed out by _step / _input die
 * with it. */
void silent_displayer_destroy(silent_displayer* d);
/* shape7 = {L, h, w, ch, cw, h2, w2}; out_floats6 (may be NULL): floats of each of the six results */
int silent_displayer_shape(const silent_displayer* d, int32_t* shape7, size_t* out_floats6);
/* Synchronous.  results[0..5]: pointers INTO the displayer's pinned result slot, valid until the SECOND next step (two slots
 * alternate).  gpu_ms (may be NULL): device time of the frame from HIP events around the graph; the call then ends in a stream
 * synchronisation.  With gpu_ms == NULL it waits for the frame's own completion word instead -- a one-thread kernel behind the last
 * one stores the frame count into pinned host memory, the host polls it (bounded: after 20 ms it synchronises the stream the
 * ordinary way, where a failed launch surfaces) -- which is ~0.02 ms less wall time per frame. */
int silent_displayer_step(silent_displayer* d, const void* frame_host, const float** results, float* gpu_ms);
/* More result slots, for callers that hand the results out zero-copy and must not overwrite what somebody still holds (the reference's
 * session.run returns fresh arrays, recognition_testing.py:132): silent_displayer_add_slot allocates one more pinned slot (numbers 0 and
 * 1 exist from creation: silent_displayer_step alternates between those two), silent_displayer_step_slot is silent_displayer_step
 * into a slot of the caller's choice -- one nobody references any more (pysilent_amd._runtime.FrameDisplayer keeps the books). */
int silent_displayer_add_slot(silent_displayer* d, int* slot_index);
int silent_displayer_step_slot(silent_displayer* d, const void* frame_host, int slot, const float** results, float* gpu_ms);
/* The displayer's pinned input buffer: a capture loop that writes the camera frame THERE and passes this pointer to
 * silent_displayer_step skips the staging copy (6 MB per 1080p frame).  A frame_host that overlaps the buffer at another
 * offset is moved into place (memmove). */
int silent_displayer_input(silent_displayer* d, void** frame_buffer, size_t* bytes);
int silent_displayer_get_state(silent_displayer* d, float* energy_host);       /* [L, ch, cw] */
int silent_displayer_set_state(silent_displayer* d, const float* energy_host);

#ifdef __cplusplus
}
#endif
#endif /* SILENT_HIP_H */
