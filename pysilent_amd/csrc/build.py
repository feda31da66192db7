#!/usr/bin/env python3
"""Build libsilent_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python pysilent_amd/csrc/build.py [--force] [--verbose] [--out PATH] [-D...]
    python pysilent_amd/csrc/build.py --host-asan     # CPU container only: lib/libsilent_hostonly_asan.so, the HOST side of
                                                      # the library under -fsanitize=address,undefined (silent_host_shim.h)

The library is one translation unit per kernel family (UNITS); they compile side by side, and a unit is recompiled only when
one of the files its last compilation read (hipcc -MMD) is newer than its object -- an A/B build of one kernel header costs one
unit.  Objects and dependency files live in csrc/build/ (git-ignored); `--out` + `-D` flags build a variant library from
objects of its own (build/<name of the library>/).
"""
import concurrent.futures
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "lib", "libsilent_hip.so")
UNITS = ["silent_core", "silent_conv_api", "silent_gray_api", "silent_peaks_api", "silent_rgb_api", "silent_pyramid_api", "silent_displayer_api"]
HOST_ASAN_OUT = os.path.join(PKG, "lib", "libsilent_hostonly_asan.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "-fno-slp-vectorize",
         "-Wall", "-Wextra", "-Wno-unused-parameter"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _deps(depfile):
    """The prerequisites a `-MMD` dependency file lists."""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    return [p for p in text.split(":", 1)[1].split() if p]


def _stale(obj, depfile, stamp):
    if not os.path.exists(obj):
        return True
    deps = _deps(depfile)
    if deps is None:
        return True
    t = os.path.getmtime(obj)
    try:
        return any(os.path.getmtime(d) > t for d in deps) or open(stamp).read() != _stamp_text
    except OSError:
        return True


_stamp_text = ""


def build(force=False, verbose=False, out=OUT, defines=()):
    """Compile the stale units (in parallel) and link.  Returns the path of the library."""
    global _stamp_text
    variant = "" if out == OUT else os.path.splitext(os.path.basename(out))[0]
    objdir = os.path.join(HERE, "build", variant)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    flags = FLAGS + list(defines) + (["-Rpass-analysis=kernel-resource-usage"] if verbose else [])
    _stamp_text = " ".join(flags)          # a change of flags recompiles everything
    jobs = []
    for u in UNITS:
        obj, dep, stamp = (os.path.join(objdir, u + ext) for ext in (".o", ".d", ".flags"))
        if force or _stale(obj, dep, stamp):
            jobs.append((u, [hipcc()] + flags + ["-MMD", "-MF", dep, "-c", os.path.join(HERE, u + ".hip"), "-o", obj], stamp))

    def run(job):
        u, cmd, stamp = job
        t0 = time.time()
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=not verbose, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s.hip:\n%s" % (u, (r.stderr or "") + (r.stdout or "")))
        if r.stderr and not verbose:
            sys.stderr.write(r.stderr)
        with open(stamp, "w") as f:
            f.write(_stamp_text)
        return u, time.time() - t0

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            for u, dt in pool.map(run, jobs):
                if verbose or os.environ.get("SILENT_BUILD_TIMES"):
                    print("  %-20s %5.1f s" % (u, dt), flush=True)
    objs = [os.path.join(objdir, u + ".o") for u in UNITS]
    if jobs or not os.path.exists(out) or any(os.path.getmtime(o) > os.path.getmtime(out) for o in objs):
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_host_asan(force=False):
    """The host code of the library without a GPU behind it (kernel launches compiled out, device memory = host memory), with
    AddressSanitizer + UndefinedBehaviorSanitizer, as ONE translation unit (silent_unity.hip includes the six).  Never loaded by
    the product; tests/test_sanitizers.py runs it in a python started with LD_PRELOAD=<clang's asan runtime>."""
    dep = os.path.join(HERE, "build", "hostonly_asan.d")
    os.makedirs(os.path.dirname(dep), exist_ok=True)
    if not force and os.path.exists(HOST_ASAN_OUT):
        deps = _deps(dep)
        if deps is not None and all(os.path.exists(d) and os.path.getmtime(d) <= os.path.getmtime(HOST_ASAN_OUT) for d in deps):
            return HOST_ASAN_OUT
    os.makedirs(os.path.dirname(HOST_ASAN_OUT), exist_ok=True)
    cmd = [hipcc(), "--offload-host-only", "-cuid=silenthost", "-DSILENT_HOST_ONLY", "-O1", "-g", "-fno-omit-frame-pointer", "-std=c++17", "-fPIC",
           "-shared", "-fvisibility=hidden", "-ffp-contract=off", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-shared-libsan", "-Wl,-Bsymbolic", "-Wno-unused-parameter", "-Wno-unused-variable",
           "-Wno-unused-but-set-variable", "-MMD", "-MF", dep, "-o", HOST_ASAN_OUT, os.path.join(HERE, "silent_unity.hip")]
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
    out = OUT
    if "--out" in sys.argv:
        out = os.path.abspath(sys.argv[sys.argv.index("--out") + 1])
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv, out=out,
                defines=[a for a in sys.argv[1:] if a.startswith("-D")]))
