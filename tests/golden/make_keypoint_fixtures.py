#!/usr/bin/env python3
"""Generates tests/golden/keypoints.npz: MARGIN FIXTURES for end-to-end keypoint identity (hard part 4;
VERDICT r2 item 2).  A fixture is a seeded synthetic frame on which every float comparison a keypoint index depends on
(value >= threshold, t == maxpool(t), peak value >= window maximum) has a margin of more than 1e-4 of the range in the
ORACLE's evaluation (tests/kp_margin.py): the GPU's float32 evaluation order moves a response by ~1e-6 of the range, so
frame -> LineEndPipeline.step -> keypoints must be IDENTICAL to oracle(frame) -> keypoints.  The file holds, per fixture,
the generator arguments (frames are regenerated from the seed, not stored), the oracle's keypoints (lists of more than 4096
rows -- a search window without a positive peak makes every pixel mapped to it a keypoint -- as row count + SHA-256 of the
int64 bytes + the rows that are not part of such a block) and the margins found.

    python tests/golden/make_keypoint_fixtures.py        (CPU only: the oracle; ~1 minute)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import kp_margin as km                                          # noqa: E402
from conftest import margin_frame, structured_frame            # noqa: E402
from pysilent_amd.pipeline import default_constants            # noqa: E402

MARGIN = 1e-4
SHAPE, LEVELS = (272, 480), 4
WANT = {("margin", "ieee"): 4, ("margin", "zero"): 3, ("lines", "zero"): 5}


def make_frame(kind, seed):
    if kind == "margin":
        return margin_frame(seed, SHAPE[0], SHAPE[1])
    return structured_frame(seed, SHAPE[0], SHAPE[1], 3, n_lines=60)


def main():
    K = default_constants("rgb")
    out, meta = {}, []
    for (kind, policy), want in WANT.items():
        found, seed = 0, 0
        while found < want and seed < 400:
            kp, margins = km.oracle_keypoints(make_frame(kind, seed), LEVELS, K, policy)
            worst = min(min(m["thr"], m["nms"], m["peak"]) for m in margins)
            if worst > MARGIN:
                name = "%s_%s_%d" % (kind, policy, seed)
                if len(kp) <= 4096:
                    out[name + "_kp"] = kp
                else:
                    import hashlib
                    out[name + "_n"] = np.array([len(kp)])
                    out[name + "_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(kp).tobytes()).digest(), np.uint8)
                out[name + "_margins"] = np.array([[m["thr"], min(m["nms"], 9.0), min(m["peak"], 9.0), m["passers"]] for m in margins])
                meta.append(name)
                found += 1
                print("%-20s keypoints %6d  worst margin %.2e  passers/level %s" % (name, len(kp), worst, [m["passers"] for m in margins]))
            seed += 1
        assert found == want, (kind, policy, found)
    out["names"] = np.array(meta)
    out["shape"] = np.array(SHAPE + (LEVELS,))
    np.savez_compressed(os.path.join(HERE, "keypoints.npz"), **out)
    print("wrote", os.path.join(HERE, "keypoints.npz"), os.path.getsize(os.path.join(HERE, "keypoints.npz")), "bytes")


if __name__ == "__main__":
    main()
