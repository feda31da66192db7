#!/usr/bin/env python3
"""Register / LDS / occupancy table of the library's kernels from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python scripts/kernel_resources.py [substring ...]      # rebuilds with build.py --force --verbose, prints the kernels that match
    python scripts/kernel_resources.py --log FILE [substring ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = [("VGPRs", "VGPRs"), ("TotalSGPRs", "SGPRs"), ("Occupancy [waves/SIMD]", "waves/SIMD"), ("SGPRs Spill", "sgpr spill"),
        ("VGPRs Spill", "vgpr spill"), ("ScratchSize [bytes/lane]", "scratch"), ("LDS Size [bytes/block]", "LDS")]


def parse(text):
    rows, cur = {}, None
    for line in text.split("\n"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        for key, _ in KEYS:
            m = re.search(r"remark:\s+" + re.escape(key) + r": (\d+)", line)
            if m and cur:
                rows[cur][key] = int(m.group(1))
    return rows


def demangle(names):
    filt = "c++filt"
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o).replace("void silent::", "") for o in out]


def main():
    args = sys.argv[1:]
    if "--log" in args:
        i = args.index("--log")
        text = open(args[i + 1]).read()
        args = args[:i] + args[i + 2:]
    else:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "pysilent_amd", "csrc", "build.py"), "--force", "--verbose"],
                           capture_output=True, text=True)
        text = r.stdout + r.stderr
        if r.returncode:
            sys.exit(text[-3000:])
    rows = parse(text)
    names = list(rows)
    print("%-72s %s" % ("kernel", "  ".join(k for _, k in KEYS)))
    for mangled, name in zip(names, demangle(names)):
        if args and not any(a in name for a in args):
            continue
        r = rows[mangled]
        print("%-72s %s" % (name[:72], "  ".join(str(r.get(k, "-")).rjust(len(lbl)) for k, lbl in KEYS)))


if __name__ == "__main__":
    main()
