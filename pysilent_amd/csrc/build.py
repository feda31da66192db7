#!/usr/bin/env python3
"""Build libsilent_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python pysilent_amd/csrc/build.py [--force] [--verbose]
    python pysilent_amd/csrc/build.py --host-asan     # CPU container only: lib/libsilent_hostonly_asan.so, the HOST side of
                                                      # silent_api.hip under -fsanitize=address,undefined (silent_host_shim.h)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "lib", "libsilent_hip.so")
SOURCES = ["silent_api.hip"]
HOST_ASAN_OUT = os.path.join(PKG, "lib", "libsilent_hostonly_asan.so")
DEPS = SOURCES + ["silent_host_shim.h", "silent_common.h", "silent_conv.h", "silent_peaks.h", "silent_pyramid.h", "silent_rgb.h", "silent_rgb2.h", "silent_walk_rgb.h",
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


def build_host_asan(force=False):
    """The host code of the library without a GPU behind it (kernel launches compiled out, device memory = host memory), with
    AddressSanitizer + UndefinedBehaviorSanitizer.  Never loaded by the product; tests/test_sanitizers.py runs it in a python
    started with LD_PRELOAD=<clang's asan runtime>."""
    if not force and os.path.exists(HOST_ASAN_OUT) and \
            all(os.path.getmtime(os.path.join(HERE, d)) <= os.path.getmtime(HOST_ASAN_OUT) for d in DEPS):
        return HOST_ASAN_OUT
    os.makedirs(os.path.dirname(HOST_ASAN_OUT), exist_ok=True)
    cmd = [hipcc(), "--offload-host-only", "-cuid=silenthost", "-DSILENT_HOST_ONLY", "-O1", "-g", "-fno-omit-frame-pointer", "-std=c++17", "-fPIC",
           "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-shared-libsan", "-Wl,-Bsymbolic", "-Wno-unused-parameter", "-Wno-unused-variable",
           "-Wno-unused-but-set-variable", "-o", HOST_ASAN_OUT] + [os.path.join(HERE, s) for s in SOURCES]
    subprocess.check_call(cmd)
    return HOST_ASAN_OUT


def asan_runtime():
    """Path of the sanitizer runtime the host-only build needs preloaded (clang's, from the ROCm LLVM)."""
    out = subprocess.check_output([hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


if __name__ == "__main__":
    if "--host-asan" in sys.argv:
        print(build_host_asan(force="--force" in sys.argv))
        sys.exit(0)
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
