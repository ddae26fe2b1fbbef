"""Register / LDS / scratch use of a graph's generated kernel, without a GPU.

    python tools/graph_stats.py allpass4 [--windows] [--dump DIR]
    python tools/graph_stats.py synth16 [--vpl 2]

Builds the graph offline (mlgpu_graph_emit), reads the code object's metadata with llvm-readelf.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import madronalib_amd as ml  # noqa: E402
from madronalib_amd import patches  # noqa: E402
from madronalib_amd.constants import Proc  # noqa: E402

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def description(name):
    if name == "synth16":
        return patches.synth16()
    if name == "synth16full":
        return patches.synth16(full=True)
    if name == "synthvoice":      # the instrument bank's voice: pitch streamed in (bench.py --workload synth)
        return patches.synth16(pitch_input=True)
    if name == "synthfused":      # the same with the pitch and gate rows computed inside (bench.py --workload synthfused)
        return patches.synth16(pitch_input=True, event_rows=True)
    if name == "allpass4":
        desc = [dict(name="x", type="input"), dict(name="dl", type="param")]
        src = "x"
        for j in range(4):
            sub, src = patches.allpass(f"ap{j}_", src, Proc.PITCHBENDABLE_DELAY, 4096.0 - 64.0, "dl")
            desc += sub
        return desc, [src]
    if name == "fdn4":
        sub, outs = patches.fdn(4, "x", 512.0)
        return [dict(name="x", type="input")] + sub, outs
    raise SystemExit(f"unknown graph {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("graph")
    ap.add_argument("--windows", action="store_true")
    ap.add_argument("--vpl", type=int, default=0)
    ap.add_argument("--dump", default=None)
    ap.add_argument("--voices", type=int, default=1024, help="bank size (>= 65536: the register budget of a bank that fills the chip)")
    ap.add_argument("--group-sum", type=int, default=0, help="output 0 = the in-order sum of groups of that many voices")
    a = ap.parse_args()
    desc, outs = description(a.graph)
    g = ml.Graph(ml.OfflineEngine(), a.voices, desc, outs, voices_per_lane=a.vpl, delay_windows=a.windows,
                 output_groups={0: a.group_sum} if a.group_sum else None)
    src, code = g.emit()
    d = a.dump or tempfile.mkdtemp()
    os.makedirs(d, exist_ok=True)
    base = os.path.join(d, a.graph + ("_windows" if a.windows else ""))
    open(base + ".hip", "w").write(src)
    open(base + ".co", "wb").write(code)
    notes = subprocess.run([READELF, "--notes", base + ".co"], capture_output=True, text=True).stdout
    for key in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count"):
        m = re.search(r"\." + key + r":\s+(\d+)", notes)
        print(f"{key:28s} {m.group(1) if m else '-'}")
    print("files:", base + ".hip", base + ".co")


if __name__ == "__main__":
    main()
