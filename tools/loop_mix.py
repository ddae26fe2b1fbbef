#!/usr/bin/env python3
"""Static instruction mix per loop of a gfx950 assembly file (hipcc -S): every backward branch closes a loop [label, branch];
for each, the number of VALU / SALU / branch / memory instructions inside (nested loops included in their parents).
    hipcc --offload-arch=gfx950 ... --cuda-device-only -S kernel.hip -o kernel.s;  tools/loop_mix.py kernel.s"""
import collections
import re
import sys


def classify(m):
    if m.startswith("v_"):
        return "valu"
    if m.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if m.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if m.startswith("ds_"):
        return "lds"
    if m.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if m.startswith(("s_waitcnt", "s_nop", "s_setprio", "s_sleep")):
        return "wait"
    if m.startswith("s_"):
        return "salu"
    return "other"


def main():
    lines = open(sys.argv[1]).read().split("\n")
    ins, labels = [], {}
    for l in lines:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        ins.append(t.split(";")[0].strip())
    loops = []
    for i, t in enumerate(ins):
        p = t.split()
        if p[0].startswith(("s_cbranch", "s_branch")) and p[-1] in labels and labels[p[-1]] <= i:
            loops.append((labels[p[-1]], i, p[-1]))
    for a, b, lab in sorted(loops, key=lambda x: x[0]):
        c = collections.Counter(classify(t.split()[0]) for t in ins[a:b + 1])
        detail = collections.Counter(t.split()[0] for t in ins[a:b + 1] if t.startswith("v_"))
        print(f"{lab}: {b - a + 1} instr  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items())))
        if "-v" in sys.argv:
            print("    " + ", ".join(f"{k}:{v}" for k, v in detail.most_common(40)))


if __name__ == "__main__":
    main()
