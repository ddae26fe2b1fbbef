#!/usr/bin/env python3
"""One table of every workload's bench line under one tag, regenerated from the files tools/gpu_profile_all.sh wrote - so that
DESIGN.md's summary and profiles/README.md are produced, not hand-edited.

    tools/summarize_profiles.py <dir> <tag> [> profiles/<tag>_summary.md]      (also writes <dir>/<tag>_summary.json)

Per `<tag>_<w>_bench.json`: units/s, the dominant kernel, its live launch time, the HBM fraction, the clock the chip held
(GRBM_GUI_ACTIVE / 8 / launch time), VALU instructions per unit, PMC traffic against algorithmic bytes, and rocprofv3's average
for the same kernel from `<tag>_<w>_kernel_stats.csv` next to the live figure."""
import csv
import glob
import json
import os
import sys


def stats_avg_us(path, kernel):
    try:
        rows = list(csv.DictReader(l for l in open(path) if not l.startswith("#")))
    except OSError:
        return None
    for r in rows:
        if kernel and kernel.split("<")[0] in r.get("Name", "") and (kernel[:40] in r["Name"] or "graph_kernel" in kernel):
            return float(r["AverageNs"]) * 1e-3
    return None


def main():
    d, tag = sys.argv[1], sys.argv[2]
    rows = []
    for f in sorted(glob.glob(os.path.join(d, f"{tag}*_bench.json"))):
        try:
            b = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception:
            continue
        r = b["roofline"]
        name = os.path.basename(f)[:-len("_bench.json")]
        rec = {"file": os.path.basename(f), "workload": b["config"]["workload"], "value": b["value"], "unit": b["unit"], "kernel": r["kernel"],
               "kernel_ms": r["kernel_ms"], "hbm_frac": r["frac"], "bound": r["bound"], "algorithmic_bytes": r["algorithmic_bytes_per_launch"],
               "traffic": r.get("traffic"), "pmc_stale": r.get("pmc_stale", False),
               "valu_insts_per_unit": (r.get("valu") or {}).get("insts_per_unit"), "valu_frac": (r.get("valu") or {}).get("frac"),
               "clock_ghz": (r.get("clock") or {}).get("ghz_live"), "cycles_per_launch": (r.get("clock") or {}).get("cycles_per_launch"),
               "valu_busy_measured": (r.get("valu") or {}).get("busy_measured"), "valu_busy_model": (r.get("valu") or {}).get("busy_frac"),
               "valu_frac_at_live_clock": (r.get("valu") or {}).get("frac_at_live_clock"), "scalar_insts_per_unit": (r.get("valu") or {}).get("scalar_insts_per_unit"),
               "frac_of_ceiling": r.get("frac_of_ceiling"),
               "rocprof_avg_us": stats_avg_us(os.path.join(d, name + "_kernel_stats.csv"), r["kernel"])}
        rows.append(rec)
    json.dump({"tag": tag, "rows": rows}, open(os.path.join(d, f"{tag}_summary.json"), "w"), indent=1)
    print(f"| bench file | units/s | kernel | ms / launch (live) | rocprofv3 avg ms | of 8 TB/s | clock GHz | VALU instr / unit | VALU busy measured (model) | PMC traffic / algorithmic |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        tr = "stale" if r["pmc_stale"] else (f"{r['traffic'] / 1e9:.3f} / {r['algorithmic_bytes'] / 1e9:.3f} GB" if r["traffic"] else f"- / {r['algorithmic_bytes'] / 1e9:.3f} GB")
        avg = f"{r['rocprof_avg_us'] / 1e3:.3f}" if r["rocprof_avg_us"] else "-"
        clk = f"{r['clock_ghz']:.2f}" if r["clock_ghz"] else "-"
        ipu = f"{r['valu_insts_per_unit']:.1f}" if r["valu_insts_per_unit"] else "-"
        bm = f"{r['valu_busy_measured']:.2f}" if r.get("valu_busy_measured") else "-"
        if r.get("valu_busy_model"):
            bm += f" ({r['valu_busy_model'][0]:.2f}-{r['valu_busy_model'][1]:.2f})"
        print(f"| `{r['file']}` | {r['value']:.3g} | `{r['kernel'][:44]}` | {r['kernel_ms']:.3f} | {avg} | {100 * r['hbm_frac']:.1f} % | {clk} | {ipu} | {bm} | {tr} |")


if __name__ == "__main__":
    main()
