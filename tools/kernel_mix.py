#!/usr/bin/env python3
"""Instruction mix of a kernel AS SHIPPED: pulls the gfx950 code objects out of madronalib_amd/csrc/libmlgpu.so (in a scratch
directory), disassembles them and counts, for every kernel whose demangled name contains PATTERN, the VALU instructions by issue
class - in particular the share of packed FP32 (v_pk_*_f32) among the FP32 add / mul / fma instructions, which the SQ_INSTS_VALU_*
counters count once each although a packed instruction occupies the SIMD as long as a "slow"-class one.

    tools/kernel_mix.py 'cascade_lanes_kernel<16, 8, 1, 8, 2, true>'      -> one JSON line per matching kernel

The mix is static (every instruction of the kernel once); for the long unrolled loops of the voice-bank kernels that is the
dynamic mix to within a few per cent. Fused graph kernels are compiled at run time and are not in the library: tools/emit_stats.py."""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
F32_PLAIN = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad|fmaak|fmamk)_f32")
F32_PACKED = re.compile(r"^v_pk_(add|mul|fma)_f32")


def code_objects(lib):
    tmp = tempfile.mkdtemp(prefix="kmix")
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, capture_output=True, check=True)
    return tmp, sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "gfx950" in f)


def kernels(lib):
    """{demangled kernel name: Counter(mnemonic)} over every gfx950 code object in the library."""
    tmp, objs = code_objects(lib)
    out = {}
    try:
        for obj in objs:
            asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "-C", obj], capture_output=True, text=True, check=True).stdout
            cur = None
            for line in asm.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
                if m:
                    cur = out.setdefault(m.group(1), collections.Counter())
                    continue
                if cur is not None and line.startswith("\t"):
                    cur[line.split()[0]] += 1
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def mix(hist):
    plain = sum(v for k, v in hist.items() if F32_PLAIN.match(k))
    packed = sum(v for k, v in hist.items() if F32_PACKED.match(k))
    valu = sum(v for k, v in hist.items() if k.startswith("v_"))
    return {"valu": valu, "f32_plain": plain, "f32_packed": packed,
            "packed_f32_share": (packed / float(plain + packed)) if plain + packed else 0.0}


def packed_share(kernel_name, lib=None):
    """Share of packed instructions among the FP32 add / mul / fma instructions of the library kernel whose demangled name matches
    (the profiler's kernel name: exact, or one containing the other); None when the kernel is not in the library."""
    lib = lib or os.path.join(ROOT, "madronalib_amd", "csrc", "libmlgpu.so")
    def bare(n):
        return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
    short = bare(kernel_name)
    for name, hist in kernels(lib).items():
        n = bare(name)
        if n == short or (min(len(short), len(n)) > 12 and (short in n or n in short)):
            return mix(hist)["packed_f32_share"]
    return None


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, hist in kernels(os.path.join(ROOT, "madronalib_amd", "csrc", "libmlgpu.so")).items():
        if pat in name and sum(hist.values()) > 8:
            print(json.dumps({"kernel": name[:120], **mix(hist)}))
