"""Element-wise rounding-error bounds for the response maps (test infrastructure).

BASELINE.json states the response tolerance as "1e-5 relative".  A float32 stencil cannot hold that on values that are the
residue of cancelling taps: ANY float32 evaluation of y = sum_t w_t x_t -- the reference's own TF kernels included -- carries
an absolute error of up to (n + 1) * 2^-24 * S with S = sum_t |w_t| |x_t| (Higham, Accuracy and Stability of Numerical
Algorithms, eq. 3.5; fma or not, any order), however small |y| is.  Round 2 asserted the element-wise 1e-5 only above a
hand-picked floor (|y| >= 0.1 * range); this module computes S for every element instead and PROPAGATES the bound through
the chain, so that the tests assert, for every element of every map,

        |gpu - oracle| <= E,      E = c * 2^-24 * S + (the input's bound pushed through |w|),      c = taps + 4

(c: n products and n - 1 additions of the chain, the oracle's own final rounding, the two-group re-association of the
structured RGB kernels: one more product and sum).  ReLU, clip and the border mask are 1-Lipschitz: they pass a bound on.
The regulator y = x * rv / min(b, 1)^root is linearised around the oracle's blur b, valid where b is well above its own
bound; elsewhere (b within 4 E_b of 0: the residue zone of the 'ieee' policy, conftest.assert_regulated_close) the
element is UNBOUNDED (np.inf) and so is everything its error can reach -- such elements stay under the range-relative rule only.

All arithmetic here is float64 NumPy on the oracle's float32 maps.
"""
import numpy as np

U = 2.0 ** -24
BIG = 1e30            # stands for "unbounded" inside the matrix products (inf * 0 would be NaN)


def _corr_same(x, k):
    """float64 SAME cross-correlation, NHWC x HWIO (the oracle's conv2d_same without the float32 roundings)."""
    x = np.asarray(x, np.float64)
    k = np.asarray(k, np.float64)
    n, h, w, ci = x.shape
    kh, kw, _, co = k.shape
    ph, pw = (kh - 1) // 2, (kw - 1) // 2
    xp = np.zeros((n, h + kh - 1, w + kw - 1, ci))
    xp[:, ph:ph + h, pw:pw + w, :] = x
    acc = np.zeros((n, h, w, co))
    for dy in range(kh):
        for dx in range(kw):
            acc += xp[:, dy:dy + h, dx:dx + w, :] @ k[dy, dx]
    return acc


def _finite(e):
    return np.minimum(np.nan_to_num(np.asarray(e, np.float64), nan=BIG, posinf=BIG), BIG)


def unbounded(e):
    return ~(np.asarray(e) < BIG / 2)


def conv(x, k, e_x=None, extra=4):
    """Bound of any float32 evaluation of conv2d_same(x, k) (+ relu / clip) against the oracle's."""
    k32 = np.asarray(k, np.float32).astype(np.float64)
    taps = k32.shape[0] * k32.shape[1] * k32.shape[2]
    xa = np.abs(np.nan_to_num(np.asarray(x, np.float64), nan=0.0, posinf=0.0, neginf=0.0))
    e = (taps + extra) * U * _corr_same(xa, np.abs(k32))
    if e_x is not None:
        e = e + _corr_same(_finite(e_x), np.abs(k32))
    # a NaN / inf input makes the outputs it reaches NaN / inf on both sides: compared by pattern, not by bound
    bad = ~np.isfinite(np.asarray(x, np.float64))
    if bad.any():
        reach = _corr_same(bad.astype(np.float64), np.ones_like(k32)) > 0
        e = np.where(reach, BIG, e)
    return np.minimum(e, BIG)


def regulate(x, blur, rv, root, e_x=None, flat_policy="ieee"):
    """Bound of y = x * rv / pow(min(conv(x, blur), 1), root) around the oracle's float64 blur."""
    x64 = np.nan_to_num(np.asarray(x, np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    k = np.asarray(blur, np.float32).astype(np.float64)
    e_x = np.zeros_like(x64) if e_x is None else _finite(e_x)
    b = _corr_same(x64, k)
    e_b = conv(x, blur, e_x)
    bc = np.minimum(np.maximum(b, 1e-300), 1.0)
    r = rv * bc ** (-root)
    with np.errstate(over="ignore"):
        dr = np.where(b - e_b < 1.0, root * rv * bc ** (-root - 1.0), 0.0)
    y = x64 * r
    # log2 / exp2 route of the fused kernels, powf elsewhere: a few ulp of the factor, 24 covers both
    with np.errstate(over="ignore", invalid="ignore"):
        e = r * e_x + np.abs(x64) * dr * e_b + 24 * U * np.abs(y)
    e = np.where((b <= 4.0 * e_b) | unbounded(e_b) | unbounded(e_x) | ~np.isfinite(e), BIG, np.minimum(e, BIG))
    if flat_policy == "zero":      # y = 0 wherever x == 0: exact on both sides when x is an exact 0 for everybody
        e = np.where((x64 == 0) & (e_x == 0), 0.0, e)
    return e


def value(x, e_x):
    """(x0 + x1 + x2) * float32(1/3): the mean of the channel bounds plus its own three roundings."""
    xa = np.abs(np.nan_to_num(np.asarray(x, np.float64), nan=0.0, posinf=0.0, neginf=0.0))
    c = xa.shape[-1]
    return np.minimum((_finite(e_x).sum(-1, keepdims=True) + 4 * U * xa.sum(-1, keepdims=True)) / c, BIG)


def pad(e, pad_px):
    out = np.zeros_like(e)
    p = int(pad_px)
    if p == 0:
        return e.copy()
    if e.shape[1] > 2 * p and e.shape[2] > 2 * p:
        out[:, p:-p, p:-p] = e[:, p:-p, p:-p]
    return out


def rgb_chain(x, kernels, chain, flat_policy="ieee", blur_root=0.1, blur_rv=1.0, pad_px=2, e_x=None):
    """Bounds for the maps of silent_oracle.rgb_line_end_chain(x, kernels, ...) = ``chain`` (same keys)."""
    e = {}
    e["rgc"] = conv(x, kernels["rgc"], e_x)
    e["rgby"] = conv(chain["rgc"], kernels["rgby"], e["rgc"])
    e["stripe"] = conv(chain["rgby"], kernels["stripe"], e["rgby"])
    e["orient"] = regulate(chain["stripe"], kernels["blur"], blur_rv, blur_root, e["stripe"], flat_policy)
    if flat_policy == "ieee":      # 0 * inf: NaN on both sides where the oracle's blur is exactly 0 (compared by pattern)
        e["orient"] = np.where(np.isnan(np.asarray(chain["orient"])), BIG, e["orient"])
    e["line_end"] = conv(chain["orient"], kernels["end"], e["orient"])
    e["padded"] = pad(e["line_end"], pad_px)
    e["value"] = value(chain["padded"], e["padded"])
    return e


def gray_chain(level, cs_kernel, end_bank, cs_map, e_x=None):
    """Bounds (E_cs, E_end) for silent_oracle.gray_line_end_pass on one level; ``cs_map`` is the oracle's CS map."""
    e_cs = conv(level, cs_kernel, e_x)
    return e_cs, conv(cs_map, end_bank, e_cs)


def zoom(level, extra=4):
    """Bound for a pyramid level: the quintic B-spline's 36 weights are >= 0, so S = sum w |x| is the zoom of |x| -- for the
    non-negative frames of this path the level itself; 6 + 6 separable taps and the float32 weights: c = 12 + extra."""
    # (no absolute slack: at zoom 1 the kernels evaluate scipy's sixth tap too -- its weight, 2^-53, is what a black pixel three
    # left of / above a bright one holds in the reference)
    return np.minimum((12 + extra) * U * np.abs(np.nan_to_num(np.asarray(level, np.float64), nan=BIG, posinf=BIG, neginf=BIG)), BIG)
