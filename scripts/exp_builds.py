#!/usr/bin/env python3
"""Variant builds of libsilent_hip.so for A/B timing on one GPU box (scratch copies of csrc/, never product code).

    python scripts/exp_builds.py <spec.py> [name ...]

<spec.py> defines VARIANTS = {name: [(file under pysilent_amd/csrc, old text, new text), ...]}; every variant is the current
tree with those replacements, built to gpurun_exp/lib_<name>.so (gpurun_exp/ travels to the GPU box, is not committed).
Select a build with SILENT_LIB_PATH=gpurun_exp/lib_<name>.so (pysilent_amd/_lib.py)."""
import os
import runpy
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "gpurun_exp")


def build(name, edits):
    d = os.path.join(EXP, "src_" + name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(os.path.join(d, "pysilent_amd", "csrc"))
    os.makedirs(os.path.join(d, "include"))
    src = os.path.join(ROOT, "pysilent_amd", "csrc")
    for f in os.listdir(src):
        if f.endswith((".h", ".hip")):
            shutil.copy(os.path.join(src, f), os.path.join(d, "pysilent_amd", "csrc", f))
    shutil.copy(os.path.join(ROOT, "include", "silent_hip.h"), os.path.join(d, "include"))
    for f, old, new in edits:
        p = os.path.join(d, "pysilent_amd", "csrc", f)
        s = open(p).read()
        assert old in s, "%s: %r not found in %s" % (name, old[:60], f)
        open(p, "w").write(s.replace(old, new))
    out = os.path.join(EXP, "lib_%s.so" % name)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-ffp-contract=off", "-fno-slp-vectorize", "-o", out, os.path.join(d, "pysilent_amd", "csrc", "silent_api.hip")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    shutil.rmtree(d, ignore_errors=True)
    return name, r.returncode, r.stdout[-2000:]


def main():
    spec = runpy.run_path(sys.argv[1])["VARIANTS"]
    names = sys.argv[2:] or list(spec)
    os.makedirs(EXP, exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        for name, rc, out in ex.map(lambda n: build(n, spec[n]), names):
            print("%-24s %s" % (name, "ok" if rc == 0 else "FAILED\n" + out))


if __name__ == "__main__":
    main()
