import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE = os.path.join(ROOT, "oracle")
if ORACLE not in sys.path:
    sys.path.insert(0, ORACLE)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(os.path.join(GOLDEN, "kernels.npz")))


@pytest.fixture(scope="session")
def golden_pyramid():
    """Outputs of the reference's own image_to_zoom_tensor (tests/golden/make_golden_pyramid.py): name -> (image f32,
    pyramid f64 [L, h, w, C], (center_w, center_h, scale))."""
    z = np.load(os.path.join(GOLDEN, "pyramid.npz"))
    names = sorted(k[:-3] for k in z.files if k.endswith("_in"))
    d = {n: (z[n + "_in"], z[n + "_out"], z[n + "_par"]) for n in names}
    d["__images__"] = {k[:-7]: z[k] for k in z.files if k.endswith("_images")}
    return d


@pytest.fixture(scope="session")
def kernels(golden):
    """The constant kernels of the chain, from the PRODUCT generators (checked against the goldens in
    test_generators.py), HWIO float64."""
    from pysilent_amd import constant_convolutions as cc
    from pysilent_amd.util.normalize import normalize_tensor_positive_negative
    return dict(
        rgc=cc.midget_rgc(2), rgby=cc.rgby_3(2), stripe=cc.rgb_2d_stripe_tensors(), blur=cc.blur_tensor(2, 7),
        end=cc.rgb_2d_end_tensors(),
        cs_gray=normalize_tensor_positive_negative(cc.center_surround_tensor(2, [1], [1], [1], [-1])),
        end4=cc.end_bank(4), end8=cc.end_bank(8), end3=cc.end_bank(3),
    )


def noise_frame(seed, h, w, c):
    """SURVEY.md section 8d synthetic input: uint8-range noise cast to float32."""
    return np.random.default_rng(seed).integers(0, 256, (h, w, c)).astype(np.float32)


def structured_frame(seed, h, w, c, n_lines=200):
    """Black background + random 1-3 px wide bright line segments: real line ends and flat regions."""
    rng = np.random.default_rng(1000 + seed)
    img = np.zeros((h, w, c), np.float32)
    for _ in range(n_lines):
        x0, y0 = rng.integers(0, w), rng.integers(0, h)
        ang = rng.uniform(0, 2 * np.pi)
        length = rng.integers(5, max(6, min(h, w) // 3))
        width = rng.integers(1, 4)
        col = rng.integers(128, 256, c).astype(np.float32)
        for t in range(length):
            x, y = int(round(x0 + t * np.cos(ang))), int(round(y0 + t * np.sin(ang)))
            img[max(0, y):min(h, y + width), max(0, x):min(w, x + width)] = col
    return img


def margin_frame(seed, h, w, c=3, n_weak=10, central=True):
    """Black background, weak clutter (1-px segments, amplitude < 110) and a few strong oblique segments: frames on which
    the keypoint decisions have room (tests/kp_margin.py).  central=True puts the strong segments into the central box that
    belongs to all four search windows of max_value_indices_region at region = extent / 2."""
    rng = np.random.default_rng(5000 + seed)
    img = np.zeros((h, w, c), np.float32)

    def seg(x0, y0, ang, length, col, width=1):
        for t in range(int(length)):
            x, y = int(round(x0 + t * np.cos(ang))), int(round(y0 + t * np.sin(ang)))
            img[max(0, y):min(h, y + width), max(0, x):min(w, x + width)] = col

    for _ in range(n_weak):
        seg(rng.integers(0, w), rng.integers(0, h), rng.uniform(0, 2 * np.pi), rng.integers(8, 40),
            rng.integers(30, 110, c).astype(np.float32))
    for k in range(3):
        L = rng.integers(max(h // 8, 12), max(h // 5, 16))
        ang = rng.uniform(0.2, np.pi / 2 - 0.2) + rng.integers(0, 4) * np.pi / 2
        col = ((255.0 - 40.0 * k) * rng.uniform(0.3, 1.0, c)).astype(np.float32)
        col[rng.integers(0, c)] = 255.0 - 40.0 * k
        lo, hi = (0.32, 0.68) if central else (0.05, 0.95)
        seg(rng.uniform(lo, hi) * w, rng.uniform(lo, hi) * h, ang, L, col, int(rng.integers(1, 3)))
    return img


def ref_margin_frame(seed, h, w, center=(288, 192), scale=float(np.e) ** .5):
    """Margin frame for the reference's nested-crop layout (zoom.from_image(frame, 3, center, scale)): level s looks at the
    centre crop of size center * scale^s, so every level gets weak clutter and a few strong oblique segments INSIDE ITS OWN
    ring (between the previous level's crop and its own), scaled with the level so that they survive the resampling."""
    rng = np.random.default_rng(9000 + seed)
    img = np.zeros((h, w, 3), np.float32)

    def seg(x0, y0, ang, length, col, width=1):
        for t in range(int(length)):
            x, y = int(round(x0 + t * np.cos(ang))), int(round(y0 + t * np.sin(ang)))
            img[max(0, y):min(h, y + width), max(0, x):min(w, x + width)] = col

    cw, ch = center
    n_levels = int(np.ceil(max(np.log(h / ch) / np.log(scale), np.log(w / cw) / np.log(scale))))
    for s in range(max(n_levels, 1)):
        f = scale ** s
        bw, bh = min(cw * f, w), min(ch * f, h)                       # this level's crop
        x0, y0 = (w - bw) / 2, (h - bh) / 2
        thick = max(1, int(round(f)))
        for _ in range(6):                                            # weak clutter
            seg(x0 + rng.uniform(0.05, 0.95) * bw, y0 + rng.uniform(0.05, 0.95) * bh, rng.uniform(0, 2 * np.pi),
                rng.integers(8, 30) * f, rng.integers(30, 100, 3).astype(np.float32), thick)
        for k in range(2):                                            # strong segments, in the central box of the crop
            col = ((255.0 - 35.0 * k - 10.0 * s) * rng.uniform(0.3, 1.0, 3)).astype(np.float32)
            col[rng.integers(0, 3)] = 255.0 - 35.0 * k - 10.0 * s
            ang = rng.uniform(0.2, np.pi / 2 - 0.2) + rng.integers(0, 4) * np.pi / 2
            seg(x0 + rng.uniform(0.35, 0.65) * bw, y0 + rng.uniform(0.35, 0.65) * bh, ang, rng.uniform(0.12, 0.2) * bh, col,
                thick * int(rng.integers(1, 3)))
    return img


WORST_REL = {}   # what -> (worst element-wise relative error at the asserted floor, same at the 1e-3 floor, share of the elements at
                 # the 1e-3 floor whose relative error exceeds rtol, number of elements at that floor)


WORST_BOUND = {}   # what -> (worst |error| / bound, share of elements without a finite bound)


def assert_close(got, want, rtol=1e-5, scale=None, what="", rel_floor=0.1, bound=None):
    """Response-map tolerance of BASELINE.json: 1e-5 relative, stated twice.
    (1) Everywhere: |a-b| <= rtol * (|b| + range), range = dynamic range of the expected map, so values that cancel
        to ~0 are judged against the magnitude of the terms that produced them, not against 0.
    (2) Element-wise, with ``bound`` (tests/err_bound.py): |a-b| <= bound for EVERY element with a finite bound -- the
        rounding-error bound of any float32 evaluation of the same sums, c * 2^-24 * sum |w| |x| propagated through the
        chain.  A float32 stencil whose taps cancel cannot be reproduced to 1e-5 of the RESULT by anyone (the reference's
        own TF kernels included); it can, and must, be reproduced to a few ulps of the TERMS.  Elements without a finite
        bound (the regulator's residue zone, NaN / inf reach) stay under (1); their share is reported.
    (3) Element-wise relative, with or without ``bound``: |a-b| <= rtol * |b| wherever |b| >= rel_floor * range
        (rel_floor=None: not asserted -- callers that compare two float32 evaluation orders with a tolerance of their own)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert np.array_equal(nan_g, nan_w), "%s: NaN pattern differs (%d vs %d)" % (what, nan_g.sum(), nan_w.sum())
    inf = np.isinf(got) | np.isinf(want)
    assert np.array_equal(got[inf], want[inf]), "%s: infinities differ" % what
    fin = ~nan_w & ~inf
    if scale is None:
        scale = float(np.max(np.abs(want[fin]))) if fin.any() else 1.0
    w64 = np.abs(want[fin].astype(np.float64))
    err = np.abs(got[fin].astype(np.float64) - want[fin].astype(np.float64))
    tol = rtol * (w64 + scale)
    bad = err > tol
    assert not bad.any(), "%s: %d / %d elements off; max err %.3e (tol %.3e)" % (
        what, bad.sum(), bad.size, err.max(), tol[np.argmax(err)])
    if bound is not None:
        bound = np.broadcast_to(np.asarray(bound, np.float64), want.shape)[fin]
        has = bound < 1e29
        # (1e-38: two exact zeros / denormal residues)
        over = has & (err > bound + 1e-38)
        if over.any():
            i = int(np.argmax(np.where(over, err / np.maximum(bound, 1e-300), 0)))
            raise AssertionError("%s: %d / %d elements beyond their rounding bound; worst |err| %.3e on a value of %.6g, bound %.3e"
                                 % (what, over.sum(), over.size, err[i], w64[i], bound[i]))
        # (the 1e-38 of the assertion above comes off: a float32 denormal against an exact 0 is not a multiple of a bound of 0)
        ratio = float((np.maximum(err[has] - 1e-38, 0.0) / np.maximum(bound[has], 1e-300)).max()) if has.any() else 0.0
        old = WORST_BOUND.get(what, (0.0, 0.0))
        WORST_BOUND[what] = (max(old[0], ratio), max(old[1], 1.0 - float(has.mean()) if has.size else 0.0))
        # (no return: a bound that has grown loose through the chain must not replace BASELINE's "1e-5 relative" -- the
        # element-wise relative rule on the significant values below is asserted as well)
    worst = [0.0, 0.0]
    n_over = n_sig = 0
    for k, floor in enumerate((rel_floor, 1e-3)):
        if floor is None:                  # caller compares two float32 evaluation orders with a tolerance of its own
            continue
        sig = w64 >= floor * scale
        if sig.any() and scale > 0:
            rel = err[sig] / w64[sig]
            worst[k] = float(rel.max())
            if k == 1:
                n_over, n_sig = int((rel > rtol).sum()), int(rel.size)
            if k == 0:
                assert worst[0] <= rtol, "%s: element-wise relative error %.3e > %.1e on a value of %.4g (floor %.3g)" % (
                    what, worst[0], rtol, w64[sig][np.argmax(rel)], floor * scale)
    old = WORST_REL.get(what, (0.0, 0.0, 0, 0))
    WORST_REL[what] = (max(old[0], worst[0]), max(old[1], worst[1]), old[2] + n_over, old[3] + n_sig)


def assert_regulated_close(got, stripe, blur, want, rtol=1e-5, what=""):
    """The regulator's output y = x * rv / pow(min(b, 1), root) under flat_policy "ieee" (0 * inf = NaN where the stripe
    value AND its whole 7 x 7 x 3 blur window are 0), compared by a DETERMINISTIC rule in three zones of the ORACLE's blur b
    (float64 accumulation), tau = 1e-5 * max(1, max stripe response of the level):
      b == 0   every stripe value of the window is exactly 0 (black input: every product is exactly 0 in any summation
               order): the GPU must show NaN exactly where the oracle does, and be close elsewhere;
      b > tau  ordinary response: finite on both sides, within the response tolerance;
      0 < b <= tau   rounding residue of cancelling taps on flat, non-black regions -- whether it is an exact 0 depends on
               the summation order (TF's own CPU and GPU kernels would disagree there): the GPU value must be NaN or a
               residue itself, 0 <= y <= 2 * tau ** 0.9 (x <= b because the blur's centre tap is 1, so y = x / b ** 0.1)."""
    got, stripe, blur, want = (np.asarray(a) for a in (got, stripe, blur, want))
    tau = 1e-5 * max(1.0, float(np.nanmax(stripe)))
    zero, resid = blur == 0, (blur > 0) & (blur <= tau)
    normal = ~zero & ~resid
    gn, wn = np.isnan(got), np.isnan(want)
    assert np.array_equal(gn[zero], wn[zero]), "%s: NaN pattern differs on exactly-zero windows (%d vs %d)" % (
        what, gn[zero].sum(), wn[zero].sum())
    assert not gn[normal].any() and not wn[normal].any(), "%s: NaN on an ordinary response" % what
    firm = (zero & ~wn) | normal
    if firm.any():
        assert_close(got[firm], want[firm], rtol, scale=float(np.nanmax(np.abs(want[firm]))) or 1.0, what=what + " (firm zones)")
    r = got[resid]
    ok = np.isnan(r) | ((r >= 0) & (r <= 2.0 * tau ** 0.9))
    assert ok.all(), "%s: %d residue-band values are neither NaN nor residue (max %.3g, bound %.3g)" % (
        what, (~ok).sum(), np.nanmax(r), 2.0 * tau ** 0.9)
    return int(resid.sum()), int((gn != wn)[resid].sum())


def pytest_terminal_summary(terminalreporter):
    if WORST_REL:
        worst = sorted(WORST_REL.items(), key=lambda kv: -kv[1][1])[:10]
        terminalreporter.write_line("worst element-wise relative error per map, |want| >= 0.1 range (asserted <= 1e-5) / "
                                    ">= 1e-3 range (reported) / share of the elements >= 1e-3 range whose relative error exceeds 1e-5: " +
                                    ", ".join("%s %.1e/%.1e/%.2e (%d of %d)" % (k, v[0], v[1], v[2] / max(v[3], 1), v[2], v[3]) for k, v in worst))
        tot_over, tot_sig = sum(v[2] for v in WORST_REL.values()), sum(v[3] for v in WORST_REL.values())
        terminalreporter.write_line("all maps together: %d of %d elements >= 1e-3 range (%.2e) are off by more than 1e-5 of THEMSELVES; every "
                                    "one of them is inside its rounding bound (next line) and inside 1e-5 of (|value| + range)"
                                    % (tot_over, tot_sig, tot_over / max(tot_sig, 1)))
    # the gray maps of BASELINE configs 1, 2, 4, 5 (the headline path) on a line of their own, whatever their rank among the RGB
    # maps: per config and map kind over all levels -- worst relative error at the 0.1 and 1e-3 floors, share of the elements
    # >= 1e-3 range beyond 1e-5 of themselves, worst |err| / rounding bound
    gray = {}
    for what, v in WORST_REL.items():
        parts = what.split()
        if len(parts) == 3 and parts[0] in ("pyramid", "cs", "end") and parts[1].startswith("config"):
            g = gray.setdefault((parts[1], parts[0]), [0.0, 0.0, 0, 0, 0.0])
            g[0], g[1], g[2], g[3] = max(g[0], v[0]), max(g[1], v[1]), g[2] + v[2], g[3] + v[3]
            g[4] = max(g[4], WORST_BOUND.get(what, (0.0, 0.0))[0])
    if gray:
        terminalreporter.write_line("gray maps of the headline path (BASELINE configs 1 / 2 / 4 / 5, all levels): worst relative error at "
                                    "|want| >= 0.1 range / >= 1e-3 range / share of elements >= 1e-3 range beyond 1e-5 of themselves / "
                                    "worst |err| / rounding bound: " +
                                    ", ".join("%s %s %.1e/%.1e/%.2e (%d of %d)/%.2f" % (c, m, g[0], g[1], g[2] / max(g[3], 1), g[2], g[3], g[4])
                                              for (c, m), g in sorted(gray.items())))
    if WORST_BOUND:
        worst = sorted(WORST_BOUND.items(), key=lambda kv: -kv[1][0])[:12]
        terminalreporter.write_line("worst |gpu - oracle| / rounding bound per map (asserted <= 1; bound = (taps + 4) * 2^-24 * "
                                    "sum |w||x|, propagated) / share of elements without a finite bound: " +
                                    ", ".join("%s %.2f/%.1e" % (k, v[0], v[1]) for k, v in worst))
