#!/usr/bin/env python3
"""Build libsilent_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python pysilent_amd/csrc/build.py [--force] [--verbose]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "lib", "libsilent_hip.so")
SOURCES = ["silent_api.hip"]
DEPS = SOURCES + ["silent_common.h", "silent_conv.h", "silent_peaks.h", "silent_pyramid.h", "silent_rgb.h", "silent_rgb2.h", "silent_walk_rgb.h",
                  os.path.join("..", "..", "include", "silent_hip.h")]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not stale():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wextra", "-Wno-unused-parameter",
           "-o", OUT] + [os.path.join(HERE, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
