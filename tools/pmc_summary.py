#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into the small files committed under profiles/.

    tools/pmc_summary.py stats  <dir-with-*_kernel_stats.csv>   -> prints the --stats table
    tools/pmc_summary.py pmc    <pmc_dir> [...]                 -> per-kernel mean counter values
    tools/pmc_summary.py traffic <fetch_dir> <write_dir> <out.json>
        HBM bytes per launch per kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM
        prescribes: FETCH_SIZE and WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts a wide
        coalesced 16 B/lane stream at exactly 1/2 of its bytes, so the read side is doubled.
        WRITE_SIZE is taken as is (the guide calls it uncalibrated: our own calibration against the
        known 4 B/sample output stream of the voice-bank kernel is printed next to it).
"""
import collections
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def counters(d):
    """{kernel: {counter: [values per dispatch]}} from *_counter_collection.csv"""
    path = find(d, "_counter_collection.csv")
    out = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(path)):
        out[r["Kernel_Name"]][r["Counter_Name"]].setdefault(r["Dispatch_Id"], 0.0)
        out[r["Kernel_Name"]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: {c: list(v.values()) for c, v in cs.items()} for k, cs in out.items()}


def durations(d):
    path = find(d, "_kernel_trace.csv")
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        out[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    return out


def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")


def main():
    mode = sys.argv[1]
    if mode == "stats":
        path = find(sys.argv[2], "_kernel_stats.csv")
        print(open(path).read())
    elif mode == "pmc":
        for d in sys.argv[2:]:
            cs = counters(d)
            du = durations(d)
            for k, c in cs.items():
                n = len(next(iter(c.values())))
                if n < 3:
                    continue
                print(f"== {short(k)}  ({n} dispatches, mean duration {sum(du[k]) / len(du[k]):.1f} us under PMC)")
                for name, vals in sorted(c.items()):
                    print(f"   {name:24s} mean {sum(vals) / len(vals):16.1f}   min {min(vals):16.1f}   max {max(vals):16.1f}")
    elif mode == "traffic":
        fetch, write, outp = counters(sys.argv[2]), counters(sys.argv[3]), sys.argv[4]
        res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 2 --warmup 1`",
               "correction": "bytes = KiB*1024; gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md §HBM); WRITE_SIZE as reported",
               "kernels": {}}
        for k in fetch:
            if k not in write or "FETCH_SIZE" not in fetch[k] or "WRITE_SIZE" not in write[k]:
                continue
            f = fetch[k]["FETCH_SIZE"]
            w = write[k]["WRITE_SIZE"]
            if len(f) < 3:
                continue
            fb = 2.0 * 1024.0 * sum(f) / len(f)
            wb = 1024.0 * sum(w) / len(w)
            res["kernels"][short(k)] = {"launches": len(f), "fetch_bytes_per_launch_corrected": fb,
                                        "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
            print(f"{short(k)}: read {fb / 1e6:.1f} MB + write {wb / 1e6:.1f} MB = {(fb + wb) / 1e6:.1f} MB per launch")
        json.dump(res, open(outp, "w"), indent=1)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
