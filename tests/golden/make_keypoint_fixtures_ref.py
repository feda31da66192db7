#!/usr/bin/env python3
"""Generates tests/golden/keypoints_ref.npz: margin fixtures like make_keypoint_fixtures.py, but on THE REFERENCE'S OWN LAYOUT --
LineEndDisplayer's defaults, pyramid_displayer.py:22: zoom.from_image(frame, 3, (288, 192), e ** .5) (util/zoom/from_image.py:
nested centre crops, each resampled to 192 x 288) -- for 480p (2 levels) and 1080p (4 levels) frames, both flat policies:
frame -> LineEndPipeline(center_dimensions=(288, 192), scale=e ** .5, selection=True).step -> keypoints must be IDENTICAL to
oracle(zoom_from_image -> chain -> top 10 % -> NMS -> value -> per-region indices) wherever every float comparison an index
depends on has more than 1e-4 of the range to spare in the oracle (tests/kp_margin.py).  Frames are regenerated from the seed.

    python tests/golden/make_keypoint_fixtures_ref.py        (CPU only: the oracle; a few minutes)
"""
import hashlib
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import kp_margin as km                                          # noqa: E402
from conftest import ref_margin_frame                           # noqa: E402
from pysilent_amd.pipeline import default_constants            # noqa: E402

MARGIN = 1e-4
CENTER, SCALE = (288, 192), math.e ** .5
SHAPES = {"ref480": (480, 640), "ref1080": (1080, 1920)}
WANT = {("ref480", "zero"): 3, ("ref480", "ieee"): 2, ("ref1080", "zero"): 3, ("ref1080", "ieee"): 2}


def main():
    K = default_constants("rgb")
    out, meta = {}, []
    for (layout, policy), want in WANT.items():
        h, w = SHAPES[layout]
        found, seed = 0, 0
        while found < want and seed < 300:
            frame = ref_margin_frame(seed, h, w, CENTER)
            kp, margins = km.oracle_keypoints(frame, 0, K, policy, scale=SCALE, center=CENTER)
            worst = min(min(m["thr"], m["nms"], m["peak"]) for m in margins)
            if worst > MARGIN:
                name = "%s_%s_%d" % (layout, policy, seed)
                if len(kp) <= 4096:
                    out[name + "_kp"] = kp
                else:
                    out[name + "_n"] = np.array([len(kp)])
                    out[name + "_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(kp).tobytes()).digest(), np.uint8)
                out[name + "_margins"] = np.array([[m["thr"], min(m["nms"], 9.0), min(m["peak"], 9.0), m["passers"]] for m in margins])
                meta.append(name)
                found += 1
                print("%-22s levels %d keypoints %7d  worst margin %.2e  passers/level %s"
                      % (name, len(margins), len(kp), worst, [m["passers"] for m in margins]), flush=True)
            seed += 1
        assert found == want, (layout, policy, found)
    out["names"] = np.array(meta)
    np.savez_compressed(os.path.join(HERE, "keypoints_ref.npz"), **out)
    print("wrote", os.path.join(HERE, "keypoints_ref.npz"), os.path.getsize(os.path.join(HERE, "keypoints_ref.npz")), "bytes")


if __name__ == "__main__":
    main()
