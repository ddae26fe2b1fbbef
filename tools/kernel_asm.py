#!/usr/bin/env python3
"""Developer tool: compile one .hip file of madronalib_amd/csrc with -save-temps and print, for the
kernels whose mangled name contains PATTERN, the instruction histogram, register counts and the
hot-loop body.   usage: tools/kernel_asm.py chains.hip 'Li2ELi18ELi48EEEELb0' [--dump]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "madronalib_amd", "csrc")


def main():
    src, pat = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    tmp = tempfile.mkdtemp(prefix="kasm")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                           "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-save-temps", "-c", os.path.join(CSRC, src),
                           "-o", os.path.join(tmp, "x.o")], cwd=tmp, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    lines = open(os.path.join(tmp, asm)).read().split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\S+):", lines[i])
        if m and pat in m.group(1):
            name = m.group(1)
            j = i + 1
            body = []
            while not lines[j].startswith(".Lfunc_end"):
                body.append(lines[j])
                j += 1
            ins = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().endswith(":")]
            hist = collections.Counter(x.split()[0] for x in ins)
            meta = "\n".join(lines[j:j + 60])
            print("==", name)
            for key in ("NumVgprs", "NumSgprs", "Occupancy", "ScratchSize", "codeLenInByte"):
                mm = re.search(r"; %s: (\S+)" % key, meta)
                print(f"   {key}: {mm.group(1) if mm else '?'}")
            print("   total instructions:", len(ins))
            print("   ", ", ".join(f"{k}:{v}" for k, v in hist.most_common(30)))
            if dump:
                print("\n".join(body))
            i = j
        i += 1


if __name__ == "__main__":
    main()
