#!/usr/bin/env python3
"""Time-bounded random parity testing on the GPU box: random extents, zoom steps, level counts, banks and batch
sizes through the C ABI against the oracle (the fixed cases live in tests/test_gpu_parity.py; this looks for the
extent / tile-boundary combination nobody thought of).  usage: scripts/fuzz_gpu.py [seconds] [seed]

Every failure prints the case so that it can be replayed (FUZZ_ONLY=<case name> fuzz_gpu.py 0 <seed>); the exit code
is the number of failing cases."""
import math, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import silent_oracle as so
from conftest import assert_close, assert_regulated_close, noise_frame, structured_frame
from pysilent_amd import _lib, _runtime as rt
from pysilent_amd.util.zoom.from_image import classic_levels

RTOL = 1e-5


def make_kernels():
    from pysilent_amd.pipeline import default_constants
    k = dict(default_constants("rgb"))
    for K in (3, 4, 8):
        g = default_constants("gray", K)
        k["end%d" % K] = g["end"]
        k["cs_gray"] = g["cs"]
    return k


def frame(rng, h, w, c):
    seed = int(rng.integers(0, 1 << 30))
    kind = rng.integers(0, 3)
    if kind == 0:
        return noise_frame(seed, h, w, c)
    if kind == 1:
        return structured_frame(seed, h, w, c, int(rng.integers(1, 60)))
    f = np.floor(np.random.default_rng(seed).random((h, w, c)) * 4).astype(np.float32) * 64   # plateaus and ties
    return f


def case_gray_pass(rng, k):
    h, w = int(rng.integers(1, 260)), int(rng.integers(1, 420))
    if rng.integers(0, 8) == 0:
        w = int(rng.integers(420, 1400))          # several 224-column blocks per row
    if rng.integers(0, 2) == 0:
        w = max(4, w // 4 * 4)                    # 16-byte aligned rows
    scale = float(rng.choice([1.3, 1.4, 2 ** .5, 1.5, 1.7, 2.0, 2.5, math.e ** .5]))
    n = int(rng.integers(1, 7))
    K = int(rng.choice([3, 4, 8]))
    B = int(rng.integers(1, 4))
    # development knobs select other (bit-identical) code paths: tile order, 32-row tiles, no stream path, unit + region pyramid
    kg, kp = int(rng.choice([0, 0, 1, 2, 3, 8, 16, 24])), int(rng.choice([0, 0, 1]))
    rt.get_context().set_tuning(_lib.TUNE_GRAY, kg)
    rt.get_context().set_tuning(_lib.TUNE_PYRAMID, kp)
    desc = "gray_pass h=%d w=%d scale=%.3f n=%d K=%d B=%d knobs=%s/%s" % (h, w, scale, n, K, B, kg, kp)
    try:
        levels = classic_levels((h, w), scale, n)
    except ValueError:
        return desc + " (no such pyramid)"
    frames = np.stack([frame(rng, h, w, 1) for _ in range(B)])
    try:
        plan = rt.PyramidPlan(h, w, 1, levels)
    except ValueError as e:
        return desc + " (plan refused: %s)" % str(e)[:60]
    bank = k["end%d" % K]
    pyr, cs, end = plan.gray_pass(frames, k["cs_gray"], bank)
    pyr2 = plan.run(frames)
    cs2, end2 = rt.gray_line_end(pyr2, k["cs_gray"], bank)
    np.testing.assert_array_equal(pyr.data, pyr2.data, err_msg=desc)
    np.testing.assert_array_equal(cs.data, cs2.data, err_msg=desc)
    np.testing.assert_array_equal(end.data, end2.data, err_msg=desc)
    f = int(rng.integers(0, B))
    want = so.classic_pyramid(frames[f], scale, n)
    for l, (wcs, wend) in enumerate(so.gray_line_end_pass(want, k["cs_gray"], bank)):
        assert_close(pyr.level(l)[f:f + 1], want[l], RTOL, scale=255.0, what=desc + " pyr %d" % l)
        assert_close(cs.level(l)[f:f + 1], wcs, RTOL, scale=255.0, what=desc + " cs %d" % l)
        assert_close(end.level(l)[f:f + 1], wend, RTOL, scale=255.0, what=desc + " end %d" % l)
    return desc + (" [stream]" if plan.streamable else " [region]")


def case_rgb(rng, k):
    h, w = int(rng.integers(3, 200)), int(rng.integers(3, 330))
    if rng.integers(0, 2) == 0:
        w = max(8, w // 4 * 4)                    # eligible for the single-read RGB pyramid walk
    scale = float(rng.choice([1.5, 2.0, 2.0, 2.5, math.e ** .5]))
    n = int(rng.integers(1, 5))
    B = int(rng.integers(1, 3))
    # 0: specialised kernel, short tiles; 8: 90-row tiles; 1 / 2: dense / no two-group forms; 9, 10: combinations;
    # + 16: the one-pixel-per-lane kernel instead of the pair kernel; widths around its 112 / 224-column wave / tile boundaries
    kr = int(rng.choice([0, 0, 8, 1, 2, 9, 10, 64, 72])) | (16 if rng.integers(0, 4) == 0 else 0)   # (64: two-group instead of the symmetric forms)
    if rng.integers(0, 6) == 0:
        w = int(rng.choice([111, 112, 113, 223, 224, 225, 336, 337, 449])) + int(rng.integers(-1, 2))
    rt.get_context().set_tuning(_lib.TUNE_RGB, kr)
    desc = "rgb h=%d w=%d scale=%.3f n=%d B=%d knob=%s" % (h, w, scale, n, B, kr)
    try:
        levels = classic_levels((h, w), scale, n)
    except ValueError:
        return desc + " (no such pyramid)"
    frames = np.stack([frame(rng, h, w, 3) for _ in range(B)])
    try:
        plan = rt.PyramidPlan(h, w, 3, levels)
    except ValueError as e:
        return desc + " (plan refused: %s)" % str(e)[:60]
    pyr = plan.run(frames)
    f = int(rng.integers(0, B))
    want = so.classic_pyramid(frames[f], scale, n)
    for l in range(n):
        assert_close(pyr.level(l)[f:f + 1], want[l], RTOL, scale=255.0, what=desc + " pyr %d" % l)
    ks = {x: k[x] for x in ("rgc", "rgby", "stripe", "blur", "end")}
    policy = "zero" if rng.integers(0, 2) else "ieee"
    got = rt.rgb_line_end(pyr, ks, flat_policy=policy)
    for l in range(n):
        lev = np.ascontiguousarray(pyr.level(l)[f:f + 1])
        w_ = so.rgb_line_end_chain(lev, ks, policy)
        if policy == "ieee" and (np.isnan(w_["orient"]).any() or np.isnan(got["orient"].level(l)[f:f + 1]).any()):
            # 0 * inf of the regulator on flat regions: the deterministic three-zone rule of conftest.assert_regulated_close
            b = so.conv2d_same(w_["stripe"], ks["blur"])
            assert_regulated_close(got["orient"].level(l)[f:f + 1], w_["stripe"], b, w_["orient"], RTOL, what=desc + " orient %d" % l)
            continue
        assert_close(got["orient"].level(l)[f:f + 1], w_["orient"], RTOL, what=desc + " orient %d" % l)
        le = so.pad_inwards(so.conv2d_same(np.ascontiguousarray(got["orient"].level(l)[f:f + 1]), ks["end"], relu=True,
                                           clip_hi=255.0), [[0, 0], [2, 2], [2, 2], [0, 0]])
        assert_close(got["line_end"].level(l)[f:f + 1], le, RTOL, scale=255.0, what=desc + " line_end %d" % l)
        np.testing.assert_array_equal(got["value"].level(l)[f:f + 1],
                                      so.value_from_color(np.ascontiguousarray(got["line_end"].level(l)[f:f + 1])), err_msg=desc)
    return desc + " " + policy


def case_select(rng, k):
    from pysilent_amd.util.selection import max_value_indices_region
    nl = int(rng.integers(1, 5))
    extents = [(int(rng.integers(1, 150)), int(rng.integers(1, 330))) for _ in range(nl)]
    c = int(rng.choice([1, 3]))
    B = int(rng.integers(1, 3))
    p = float(rng.choice([0.0, 0.1, 0.37, 0.5, 1.0]))
    desc = "select extents=%s c=%d B=%d p=%g" % (extents, c, B, p)
    levels = [np.stack([frame(rng, h, w, c) for _ in range(B)]) for h, w in extents]
    packed = rt.PackedPyramid.from_levels(levels)
    got = rt.select_peaks(packed, p, None)
    for l, lev in enumerate(levels):
        v = so.value_from_color(lev)
        peaks = so.nms3x3(so.top_value_points(lev, p, v), "product")
        np.testing.assert_array_equal(got["peaks"].level(l), peaks, err_msg=desc + " peaks %d" % l)
        np.testing.assert_array_equal(got["peak_value"].level(l), so.value_from_color(peaks), err_msg=desc + " pv %d" % l)
    regions = [(max(1, h // int(rng.integers(1, 4))), max(1, w // int(rng.integers(1, 4)))) for h, w in extents]
    try:
        kp = max_value_indices_region(got["peaks"], regions, got["peak_value"])
    except ValueError as e:
        return desc + " (keypoints refused: %s)" % str(e)[:60]
    for f in range(B):
        rows = []
        for l in range(nl):
            v = np.ascontiguousarray(got["peak_value"].level(l)[f:f + 1])
            r = so.max_value_indices_region(None, (1,) + regions[l] + (c,), v)
            r[:, 0] = l
            rows.append(r)
        np.testing.assert_array_equal(kp[f], np.concatenate(rows), err_msg=desc + " keypoints frame %d" % f)
    return desc


def case_ops(rng, k):
    """Stand-alone entry points on one random NHWC tensor: conv2d_same (specialised and generic shapes), resize_nearest,
    affine_clip, nms3x3, pad_inwards, get_centroids, regulate (zero policy)."""
    from pysilent_amd.util import get_centroids
    B, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 90)), int(rng.integers(1, 200))
    kh, kw = int(rng.choice([1, 2, 3, 3, 3, 5, 7])), int(rng.choice([1, 3, 3, 3, 4, 7]))
    ci, co = int(rng.choice([1, 1, 2, 3, 3])), int(rng.choice([1, 3, 4, 5, 8]))
    desc = "ops B=%d h=%d w=%d k=%dx%dx%dx%d" % (B, h, w, kh, kw, ci, co)
    g = np.random.default_rng(int(rng.integers(0, 1 << 30)))
    x = (g.standard_normal((B, h, w, ci)) * 40).astype(np.float32)
    kern = g.standard_normal((kh, kw, ci, co))
    relu = bool(rng.integers(0, 2))
    clip = float(rng.choice([30.0, 255.0])) if relu and rng.integers(0, 2) else None
    try:
        got = rt.conv2d_same(x, kern, relu=relu, clip_hi=clip)
    except ValueError as e:
        if "exceeds" not in str(e):
            raise
        got = None                     # documented capacity of the constant-memory weight block
    if got is not None:
        # random-sign taps cancel: the float32 accumulation error scales with sum|k| * max|x|, not with the result
        assert_close(got, so.conv2d_same(x, kern, relu=relu, clip_hi=clip), RTOL,
                     scale=float(np.abs(kern).sum() * np.abs(x).max()), what=desc + " conv")
    oh, ow = int(rng.integers(1, 120)), int(rng.integers(1, 220))
    np.testing.assert_array_equal(rt.resize_nearest(x, (oh, ow)), so.resize_nearest_tf1(x, oh, ow), err_msg=desc + " resize")
    kwargs = dict(mul=float(g.standard_normal() * 10), add=float(g.standard_normal() * 5), lo=-50.0, hi=60.0, post_add=-1.0)
    np.testing.assert_array_equal(rt.affine_clip(x, **kwargs), so.affine_clip(x, **kwargs), err_msg=desc + " affine")
    q = np.floor(x / 16).astype(np.float32)     # ties
    for mode in ("product", "fired"):
        np.testing.assert_array_equal(rt.nms3x3(q, mode), so.nms3x3(q, mode), err_msg=desc + " nms " + mode)
    pads = [int(v) for v in rng.integers(0, 4, 4)]
    np.testing.assert_array_equal(rt.pad_inwards(x, *pads),
                                  so.pad_inwards(x, [[0, 0], [pads[0], pads[1]], [pads[2], pads[3]], [0, 0]]), err_msg=desc + " pad")
    np.testing.assert_array_equal(rt.bw_from_color(q), so.bw_from_color(q), err_msg=desc + " bw")
    np.testing.assert_array_equal(rt.value_from_color(x), so.value_from_color(x), err_msg=desc + " value")
    if ci == 1:
        from pysilent_amd.util.color import to_channels
        np.testing.assert_array_equal(to_channels(x, co), np.tile(x, (1, 1, 1, co)), err_msg=desc + " to_channels")
        v = (np.abs(x) * (g.random(x.shape) > 0.5)).astype(np.float32)
        region = [1, int(rng.integers(1, 6)), int(rng.integers(1, 6))]
        dist, total = get_centroids(v, region)
        wd, wt = so.get_centroids(v, region)
        assert_close(total, wt, RTOL, what=desc + " centroid totals")
        assert_close(dist, wd, RTOL, scale=float(max(h, w)), what=desc + " centroid distances")
    if ci == 3:
        xs = np.abs(x) * np.float32(10.0 ** float(rng.integers(-3, 1)))
        root = float(rng.choice([0.1, 0.5, 1.0]))
        assert_close(rt.regulate(xs, k["blur"], 1.0, root, "zero"), so.regulate(xs, k["blur"], 1.0, root, "zero"), RTOL,
                     what=desc + " regulate root %g" % root)
    return desc


def case_big(rng, k):
    """Frames of camera size against the C port of the oracle (multi-block rows, many tiles, the 90-row RGB tiles)."""
    import c_oracle as co
    h, w = int(rng.integers(300, 1300)), int(rng.integers(400, 2300))
    scale = float(rng.choice([1.7, 2.0, 2.0, math.e ** .5]))
    n = int(rng.integers(2, 8))
    gray = bool(rng.integers(0, 2))
    B = int(rng.integers(1, 3)) if gray else int(rng.integers(1, 5))
    desc = "big %s h=%d w=%d scale=%.3f n=%d B=%d" % ("gray" if gray else "rgb", h, w, scale, n, B)
    try:
        levels = classic_levels((h, w), scale, n)
    except ValueError:
        return desc + " (no such pyramid)"
    c = 1 if gray else 3
    frames = np.stack([noise_frame(int(rng.integers(0, 1 << 30)), h, w, c) for _ in range(B)])
    plan = rt.PyramidPlan(h, w, c, levels)
    f = int(rng.integers(0, B))
    want = co.classic_pyramid(frames[f], plan.extents)
    if gray:
        K = int(rng.choice([4, 8]))
        pyr, cs, end = plan.gray_pass(frames, k["cs_gray"], k["end%d" % K])
        for l in range(n):
            assert_close(pyr.level(l)[f:f + 1], want[l], RTOL, scale=255.0, what=desc + " pyr %d" % l)
            wcs, wend = co.gray_line_end_level(want[l], k["cs_gray"], k["end%d" % K])
            assert_close(cs.level(l)[f:f + 1], wcs, RTOL, scale=255.0, what=desc + " cs %d" % l)
            assert_close(end.level(l)[f:f + 1], wend, RTOL, scale=255.0, what=desc + " end %d" % l)
        return desc + (" [stream]" if plan.streamable else " [region]")
    pyr = plan.run(frames)
    ks = {x: k[x] for x in ("rgc", "rgby", "stripe", "blur", "end")}
    got = rt.rgb_line_end(pyr, ks)
    for l in range(n):
        assert_close(pyr.level(l)[f:f + 1], want[l], RTOL, scale=255.0, what=desc + " pyr %d" % l)
        x = np.ascontiguousarray(pyr.level(l)[f:f + 1])
        for name in ("rgc", "rgby", "stripe"):
            x = co.conv2d_same(x, ks[name], relu=True)
        orient = co.regulate(x, ks["blur"], 1.0, 0.1)
        assert_close(got["orient"].level(l)[f:f + 1], orient, RTOL, what=desc + " orient %d" % l)
        g_or = np.ascontiguousarray(got["orient"].level(l)[f:f + 1])
        line = co.pad_inwards(co.conv2d_same(g_or, ks["end"], relu=True, clip_hi=255.0), [[0, 0], [2, 2], [2, 2], [0, 0]])
        assert_close(got["line_end"].level(l)[f:f + 1], line, RTOL, scale=255.0, what=desc + " line_end %d" % l)
    return desc


def case_rgb_keypoints(rng, k):
    """silent_rgb_keypoints (chain + selection + keypoints in one call, extrema from the chain kernel) against
    silent_rgb_line_end followed by silent_select_keypoints on the same pyramid: every output bit for bit."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    h, w = int(rng.integers(8, 180)), int(rng.integers(8, 300))
    if rng.integers(0, 5) == 0:
        w = int(rng.choice([111, 112, 113, 223, 224, 225, 337])) + int(rng.integers(-1, 2))
    n, B = int(rng.integers(1, 5)), int(rng.integers(1, 3))
    kr = int(rng.choice([0, 0, 0, 8, 16, 1, 2]))
    policy = "zero" if rng.integers(0, 2) else "ieee"
    vm = bool(rng.integers(0, 2))
    p = float(rng.choice([0.0, 0.1, 0.1, 0.37, 1.0]))
    desc = "rgb_keypoints h=%d w=%d n=%d B=%d knob=%d %s value_map=%s p=%g" % (h, w, n, B, kr, policy, vm, p)
    try:
        consts = {x: k[x] for x in ("rgc", "rgby", "stripe", "blur", "end")}
        kw = dict(mode="rgb", n_levels=n, batch=B, selection=True, top_percent=p, flat_policy=policy, constants=consts,
                  max_keypoints_per_frame=h * w * 2)
        fused = LineEndPipeline((h, w), value_map=vm, **kw)
        plain = LineEndPipeline((h, w), **kw)
    except ValueError as e:
        return desc + " (no such pyramid)" if "pyramid" in str(e) or "level" in str(e) else desc + " (plan refused: %s)" % str(e)[:60]
    frames = np.stack([frame(rng, h, w, 3) for _ in range(B)])
    if rng.integers(0, 4) == 0:
        frames[0, int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(0, 3))] = rng.choice([np.nan, np.inf, -np.inf])
    t = torch.from_numpy(frames).cuda()
    rt.get_context().set_tuning(_lib.TUNE_RGB, kr)
    fused.step(t)
    plain.run_pyramid(t)
    plain.run_filters()
    plain.run_keypoints()
    torch.cuda.synchronize()
    rt.get_context().set_tuning(_lib.TUNE_RGB, 0)
    a, b = fused.outputs(allow_truncated=True), plain.outputs(allow_truncated=True)
    for name in ("orient", "line_end", "peak_value") + (("value",) if vm else ()):
        x, y = a[name].data.cpu().numpy(), b[name].data.cpu().numpy()
        assert np.array_equal(np.isnan(x), np.isnan(y)), desc + " NaN pattern of " + name
        np.testing.assert_array_equal(np.nan_to_num(x, nan=7.0), np.nan_to_num(y, nan=7.0), err_msg=desc + " " + name)
    np.testing.assert_array_equal(a["keypoint_counts"], b["keypoint_counts"], err_msg=desc)
    for f in range(B):
        np.testing.assert_array_equal(a["keypoints"][f], b["keypoints"][f], err_msg=desc)
    return desc


def case_sparse_tail(rng, k):
    """silent_rgb_keypoints with the sparse keypoint tail (no peak-value map) against the same call with the map (dense
    kernels): identical keypoints on random extents, level counts, tile heights, thresholds, policies and frame kinds
    (noise, line drawings, plateaus with ties, black / constant frames, NaN / inf pixels)."""
    import torch
    from pysilent_amd.pipeline import LineEndPipeline
    h, w = int(rng.integers(8, 260)), int(rng.integers(8, 420))
    if rng.integers(0, 5) == 0:
        w = int(rng.choice([111, 112, 113, 223, 224, 225, 337])) + int(rng.integers(-1, 2))
    n, B = int(rng.integers(1, 6)), int(rng.integers(1, 4))
    th = int(rng.choice([0, 0, 0, 7, 9, 16, 20, 33, 45]))
    kr = (th << 8) | int(rng.choice([0, 0, 0, 8]))
    policy = "zero" if rng.integers(0, 2) else "ieee"
    p = float(rng.choice([0.0, 0.1, 0.1, 0.1, 0.37, 0.9, 1.0]))
    desc = "sparse_tail h=%d w=%d n=%d B=%d knob=%d %s p=%g" % (h, w, n, B, kr, policy, p)
    try:
        consts = {x: k[x] for x in ("rgc", "rgby", "stripe", "blur", "end")}
        kw = dict(mode="rgb", n_levels=n, batch=B, selection=True, top_percent=p, flat_policy=policy, constants=consts,
                  max_keypoints_per_frame=h * w * 2, value_map=False)
        sparse = LineEndPipeline((h, w), peak_value_map=False, **kw)
        dense = LineEndPipeline((h, w), peak_value_map=True, **kw)
    except ValueError as e:
        return desc + " (no such pyramid)" if "pyramid" in str(e) or "level" in str(e) else desc + " (plan refused: %s)" % str(e)[:60]
    frames = np.stack([frame(rng, h, w, 3) for _ in range(B)])
    kind = rng.integers(0, 8)
    if kind == 0:
        frames[0] = 0.0
    elif kind == 1:
        frames[0] = float(rng.integers(1, 255))
    elif kind == 2:
        frames[0, int(rng.integers(0, h)), int(rng.integers(0, w)), int(rng.integers(0, 3))] = rng.choice([np.nan, np.inf, -np.inf])
    t = torch.from_numpy(frames).cuda()
    rt.get_context().set_tuning(_lib.TUNE_RGB, kr)
    sparse.step(t)
    stats = sparse.sparse_tail_stats()
    dense.step(t)
    torch.cuda.synchronize()
    rt.get_context().set_tuning(_lib.TUNE_RGB, 0)
    assert stats["ran"], desc
    a, b = sparse.outputs(allow_truncated=True), dense.outputs(allow_truncated=True)
    np.testing.assert_array_equal(a["keypoint_counts"], b["keypoint_counts"], err_msg=desc)
    for f in range(B):
        np.testing.assert_array_equal(a["keypoints"][f], b["keypoints"][f], err_msg=desc)
    x, y = a["line_end"].data.cpu().numpy(), b["line_end"].data.cpu().numpy()
    np.testing.assert_array_equal(np.nan_to_num(x, nan=7.0), np.nan_to_num(y, nan=7.0), err_msg=desc + " line_end")
    return desc + " [dense %d zero %d of %d]" % (stats["dense_pairs"], stats["zero_map_pairs"], stats["pairs"])


def case_crop_walk(rng, k):
    """3-channel pyramids in the reference's crop layout (image_to_zoom_tensor: nested centre crops) and classic ones at small
    zoom steps: the strip-walk kernel's walk plans against the unit + region kernels (bit for bit) and the oracle."""
    from pysilent_amd.util.zoom.from_image import reference_levels
    h, w = int(rng.integers(40, 400)), int(rng.integers(10, 160)) * 4 + (int(rng.integers(0, 4)) if rng.integers(0, 2) else 0)   # any width
    scale = float(rng.choice([1.2, 2 ** (1 / 3), 1.3, 2 ** .5, 1.5, 1.6, math.e ** .5, 1.7, 2.0, 2.5]))
    B = int(rng.integers(1, 4))
    if rng.integers(0, 3) == 0:
        n = int(rng.integers(2, 10))
        desc = "crop_walk classic h=%d w=%d scale=%.3f n=%d B=%d" % (h, w, scale, n, B)
        try:
            levels = classic_levels((h, w), scale, n)
        except ValueError:
            return desc + " (no such pyramid)"
        want_fn = lambda img: np.concatenate([l.reshape(-1) for l in so.classic_pyramid(img, scale, n)])
    else:
        cw, ch = int(rng.integers(4, max(5, w // 2))), int(rng.integers(4, max(5, h // 2)))
        desc = "crop_walk reference h=%d w=%d center=(%d,%d) scale=%.3f B=%d" % (h, w, cw, ch, scale, B)
        try:
            levels = reference_levels((h, w), (cw, ch), scale)
        except (ValueError, ZeroDivisionError):
            return desc + " (no such pyramid)"
        if not levels or len(levels) > 12:
            return desc + " (no such pyramid)"
        want_fn = lambda img: so.zoom_from_image(img, 3, (cw, ch), scale).reshape(-1)
    frames = np.stack([frame(rng, h, w, 3) for _ in range(B)])
    try:
        plan = rt.PyramidPlan(h, w, 3, levels)
    except ValueError as e:
        return desc + " (plan refused: %s)" % str(e)[:60]
    n_plans, px = plan.walk_plans
    got = plan.run(frames)
    rt.get_context().set_tuning(_lib.TUNE_PYRAMID, 2)
    two = plan.run(frames)
    rt.get_context().set_tuning(_lib.TUNE_PYRAMID, 0)
    np.testing.assert_array_equal(got.data, two.data, err_msg=desc)
    f = int(rng.integers(0, B))
    want = want_fn(frames[f])
    per = got.data.reshape(B, -1)
    assert_close(per[f], want, RTOL, scale=255.0, what=desc)
    return desc + (" [walk %d x %d]" % (n_plans, px) if n_plans else " [region]")


CASES = {"gray_pass": case_gray_pass, "rgb": case_rgb, "select": case_select, "ops": case_ops, "rgb_keypoints": case_rgb_keypoints,
         "sparse_tail": case_sparse_tail, "crop_walk": case_crop_walk}
BIG = {"big": case_big}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    only = os.environ.get("FUZZ_ONLY")
    k = make_kernels()
    rng = np.random.default_rng(seed)
    t0, n, fails = time.time(), 0, 0
    tally = {}
    names = [only] if only else list(CASES)
    last = t0
    replay = os.environ.get("FUZZ_SUB")
    while True:
        name = names[n % len(names)]
        sub = int(replay) if replay else int(rng.integers(0, 1 << 31))
        try:
            desc = {**CASES, **BIG}[name](np.random.default_rng(sub), k)
            kind = name + " ok"
            for mark in ("big gray", "big rgb", "(no such pyramid)", "(plan refused", "(keypoints refused", "[stream]", "[region]", "[walk"):
                if mark in desc:
                    kind = name + " " + mark
            tally[kind] = tally.get(kind, 0) + 1
        except Exception:
            fails += 1
            print("FAIL case=%s sub_seed=%d" % (name, sub))
            traceback.print_exc(limit=3)
            sys.stdout.flush()
        n += 1
        if time.time() - last > 30:
            last = time.time()
            print("[%4.0f s] %d cases, %d failures; last: %s" % (last - t0, n, fails, desc))
            sys.stdout.flush()
        if time.time() - t0 > budget:
            break
    for kind in sorted(tally):
        print("  %-60s %d" % (kind, tally[kind]))
    print("seed %d: %d cases in %.0f s, %d failures" % (seed, n, time.time() - t0, fails))
    return fails


if __name__ == "__main__":
    sys.exit(min(main(), 100))
