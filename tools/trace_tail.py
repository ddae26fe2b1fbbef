#!/usr/bin/env python3
"""Tail of a rocprofv3 --kernel-trace CSV as a timeline: kernel, start, end (us from the first row shown), duration, gap to the
previous kernel's end, stream. Usage: trace_tail.py kernel_trace.csv [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -40:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:8.1f}"
    print(f'{r["Kernel_Name"][:44]:44s} {s / 1e3:10.1f} {e / 1e3:10.1f} {(e - s) / 1e3:8.1f} {gap:>8s}  {r.get("Stream_Id", "")}')
    prev_end = e if prev_end is None else max(prev_end, e)
