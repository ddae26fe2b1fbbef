#!/usr/bin/env python3
"""Developer tool (no GPU needed): generate a patch's fused graph kernel with hiprtc under the current MLGPU_GRAPH_* knobs and
print registers, scratch, code size and the instruction mix.   usage: tools/emit_stats.py cfg5|cfg5full|synth|synthfused [outprefix]"""
import collections
import os
import re
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd import patches  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
out = sys.argv[2] if len(sys.argv) > 2 else "/tmp/emit_" + which
kw = dict(cfg5={}, cfg5full=dict(full=True), synth=dict(pitch_input=True), synthfused=dict(event_rows=True))[which]
d, o = patches.synth16(**kw)
g = ml.Graph(ml.OfflineEngine(), 262144, d, o)
src, code = g.emit()
open(out + ".hip", "w").write(src)
open(out + ".co", "wb").write(code)
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", out + ".co"], capture_output=True, text=True).stdout
asm = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", out + ".co"], capture_output=True, text=True).stdout
open(out + ".s", "w").write(asm)
vals = {k: re.search(r"\.%s:\s+(\d+)" % k, notes).group(1) for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "vgpr_spill_count")}
ins = [l.split()[0] for l in asm.split("\n") if l.startswith("\t")]
hist = collections.Counter(ins)
print(which, vals, "instructions:", len(ins), "scratch ops:", sum(v for k, v in hist.items() if k.startswith("scratch_")))
