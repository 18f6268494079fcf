#!/usr/bin/env python3
"""Lists `s_waitcnt vmcnt(0)` instructions that sit INSIDE a loop which also issues stores or
atomics — the compiler-placed drains profiles/NOTEBOOK.md A9 keeps finding: no-return atomics and stores stay
counted in vmcnt until the memory side acknowledges them (1-3 k cycles), so such a wait stalls every
trip for all of them, although it was only meant for a load issued before the loop.
Usage: python tools/isa_drains.py [file.hip ...]   (default: every kernel source; compiles with
-save-temps into a scratch directory, no GPU needed)."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "volsurfs_amd", "csrc")


def demangle(n):
    try:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except OSError:
        d = n
    d = re.sub(r"^void ", "", d).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", d)[:80]


def scan(path):
    out = []
    s = open(path).read()
    for m in re.finditer(r"^(_Z[\w$.]+):[^\n]*\n(.*?)\ts_endpgm", s, re.S | re.M):
        name, lines = m.group(1), m.group(2).split("\n")
        labels = {}
        for i, l in enumerate(lines):
            mm = re.match(r"^(\.LBB\d+_\d+):", l)
            if mm:
                labels[mm.group(1)] = i
        for i, l in enumerate(lines):
            mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if not (mm and mm.group(1) in labels and labels[mm.group(1)] < i):
                continue
            a, b = labels[mm.group(1)], i
            body = lines[a:b + 1]
            waits = [a + j for j, x in enumerate(body) if "s_waitcnt vmcnt(0)" in x]
            stores = sum(1 for x in body if re.search(r"(global|buffer|flat)_(store|atomic)", x))
            loads = sum(1 for x in body if re.search(r"(global|buffer|flat)_load", x))
            if waits and stores:
                out.append((demangle(name), a, b, len(waits), stores, loads))
    return out


def main():
    srcs = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    with tempfile.TemporaryDirectory() as tmp:
        for src in map(os.path.abspath, srcs):
            base = os.path.splitext(os.path.basename(src))[0]
            subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
                            "-I" + CSRC, "-x", "hip", "-c", src, "-save-temps", "-o", os.path.join(tmp, base + ".o")],
                           cwd=tmp, stderr=subprocess.DEVNULL)
            for f in glob.glob(os.path.join(tmp, base + "-hip-amdgcn*.s")):
                for name, a, b, nw, st, ld in scan(f):
                    print(f"{base}: {name}\n    loop at lines {a}-{b}: vmcnt(0) x{nw}, stores/atomics {st}, loads {ld}")


if __name__ == "__main__":
    main()
