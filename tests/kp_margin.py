"""End-to-end keypoint parity helpers (test infrastructure; uses the oracle).

``oracle_keypoints(frame, ...)``: frame -> zoom pyramid -> rgc > rgby > stripe > regulate > end > clip > pad (the reference
graph, recognition_testing.py:69-77) -> top 10 % (a-10) -> 3x3 NMS (a-9) -> value (a-8) -> per-region indices (a-11), all
through the oracle; returns the int64 [K, 4] rows a LineEndPipeline(selection=True) step must produce for that frame and,
per level, the MARGINS of every float comparison an index depends on (hard part 4):

  thr    min |value - threshold| over the pixels of the level (a-10: value >= thr), as a fraction of the value range
  nms    min over threshold passers, channels: |t_c - max of the 8 neighbours' t_c| where that decides t_c == maxpool(t)
         (pairs of values that are both below ``tiny`` are not decisions: they move the peak value by < tiny^2)
  peak   min over search windows and pixels mapped to them: window maximum - peak value, for every pixel that is NOT the
         (unique) maximum itself.  A window whose maximum is not > 0 makes every non-NaN pixel mapped to it a keypoint:
         exact on both sides when the level's peak-value map holds no NaN (a non-passer's peak value is exactly 0), margin 0
         otherwise (the NaN pattern of the 'ieee' regulator would have to be reproduced pixel for pixel)

A frame is a MARGIN FIXTURE when all three exceed ``margin`` (1e-4 of the range) on every level: the GPU's float32
evaluation order moves a response by ~1e-6 of the range, so its keypoints must then be IDENTICAL to the oracle's.
"""
import numpy as np

import c_oracle as co
import silent_oracle as so

F32 = np.float32


def chain_maps(level, kernels, flat_policy="ieee", pad=2, chain=None):
    """line_end (padded) of one pyramid level [1, h, w, 3] through the C oracle (bit-identical to the NumPy one); ``chain``:
    optional dict that receives every intermediate map (the keys of silent_oracle.rgb_line_end_chain)."""
    chain = {} if chain is None else chain
    x = level
    for name in ("rgc", "rgby", "stripe"):
        x = chain[name] = co.conv2d_same(x, kernels[name], relu=True)
    orient = chain["orient"] = co.regulate(x, kernels["blur"], 1.0, 0.1, flat_policy)
    line = chain["line_end"] = co.conv2d_same(orient, kernels["end"], relu=True, clip_hi=255.0)
    chain["padded"] = co.pad_inwards(line, [[0, 0], [pad, pad], [pad, pad], [0, 0]])
    return chain["padded"]


def selection_of(line, top_percent=0.1):
    """(value, thr, top, peak value) of one level's line-end map, oracle semantics."""
    value = co.value_from_color(line)
    mx, mn = so.level_max_min(value)
    thr = F32(F32(F32(1.0 - top_percent) * mx[0]) + F32(F32(top_percent) * mn[0]))
    top = co.top_value_points(line, top_percent, value)
    pv = co.value_from_color(co.nms3x3(top, "product"))
    return value, thr, top, pv


def level_margins(line, top_percent=0.1, tiny=1e-3):
    value, thr, top, pv = selection_of(line, top_percent)
    v = value[0, :, :, 0]
    fin = np.isfinite(v)
    rng = float(v[fin].max() - min(v[fin].min(), 0.0)) if fin.any() else 1.0
    rng = rng if rng > 0 else 1.0
    out = {"range": rng, "passers": int((v[fin] >= thr).sum())}
    out["thr"] = float(np.abs(v[fin].astype(np.float64) - float(thr)).min() / rng) if fin.any() else 0.0
    # NMS decisions of the passers
    t = top[0]
    h, w, _ = t.shape
    ys, xs = np.nonzero(fin & (v >= thr))
    nms = np.inf
    for y, x in zip(ys, xs):
        y0, y1, x0, x1 = max(y - 1, 0), min(y + 2, h), max(x - 1, 0), min(x + 2, w)
        for c in range(3):
            nb = t[y0:y1, x0:x1, c].astype(np.float64).copy()
            nb[y - y0, x - x0] = -np.inf
            nb = nb[~np.isnan(nb)]
            m = float(nb.max()) if nb.size else -np.inf
            a = float(t[y, x, c])
            if max(a, m) < tiny * rng:
                continue
            nms = min(nms, abs(a - m) / rng)
    out["nms"] = float(nms)
    # window maxima against everything else mapped to the window (cell by cell of the nearest-neighbour resize)
    p = pv[0, :, :, 0].astype(np.float64)
    thr_map = so.region_threshold(pv, max(h // 2, 1), max(w // 2, 1))[0, :, :, 0].astype(np.float64)
    prange = float(np.nanmax(p)) if np.isfinite(p).any() and np.nanmax(p) > 0 else 1.0
    _, _, src_y = so._region_pool_geometry(h, max(h // 2, 1))
    _, _, src_x = so._region_pool_geometry(w, max(w // 2, 1))
    peak = np.inf
    for j in np.unique(src_y):
        for i in np.unique(src_x):
            cell = np.ix_(src_y == j, src_x == i)
            pc, tc = p[cell], thr_map[cell]
            t0 = float(tc.flat[0])
            if not t0 > 0:            # no positive peak in the window: every non-NaN pixel of the cell is a keypoint
                if np.isnan(p).any():
                    peak = 0.0
                continue
            with np.errstate(invalid="ignore"):
                hit = pc >= t0
            if hit.sum() > 1:         # a tie at the maximum: a rounding difference drops one of the keypoints
                peak = 0.0
            rest = ~hit & np.isfinite(pc)
            if rest.any():
                peak = min(peak, float((t0 - pc[rest]).min() / prange))
    out["peak"] = float(peak)
    return out, pv


def oracle_keypoints(frame, n_levels, kernels, flat_policy="ieee", top_percent=0.1, scale=2.0, keep=None, center=None):
    """keep: optional list that receives (pyramid level, dict of the oracle's chain maps) of every level.
    center = (w, h): the reference's own layout instead of the classic one -- zoom.from_image(frame, 3, center, scale)
    (util/zoom/from_image.py:10-69: nested centre crops resampled to one fixed size; n_levels is ignored, the level count follows
    from the frame and the centre size like in the reference)."""
    h, w, _ = frame.shape
    if center is not None:
        z = so.zoom_from_image(frame, 3, center, scale)
        pyr = [np.ascontiguousarray(z[l:l + 1], dtype=F32) for l in range(z.shape[0])]
        extents = [(int(center[1]), int(center[0]))] * len(pyr)
    else:
        extents = so.classic_extents(h, w, scale, n_levels)
        pyr = co.classic_pyramid(frame, extents)
    rows, margins = [], []
    for l, lev in enumerate(pyr):
        chain = {}
        line = chain_maps(lev, kernels, flat_policy, chain=chain)
        if keep is not None:
            keep.append((lev, chain))
        m, pv = level_margins(line, top_percent)
        lh, lw = extents[l]
        r = co.max_value_indices_region(None, (1, max(lh // 2, 1), max(lw // 2, 1), 3), pv)
        r[:, 0] = l
        rows.append(r)
        margins.append(m)
    return np.concatenate(rows), margins


def is_margin_fixture(margins, margin=1e-4):
    return all(min(m["thr"], m["nms"], m["peak"]) > margin for m in margins)
