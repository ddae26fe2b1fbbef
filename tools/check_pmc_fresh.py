#!/usr/bin/env python3
"""Which records of profiles/pmc_workloads.json were measured on the device code in this tree? (bench.py attaches a record to
its line only when the fingerprints agree; a partial re-profile must not leave the headline's record behind.)
Exit status 1 when a BASELINE config's record is stale."""
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import madronalib_amd as ml  # noqa: E402

now = ml.device_source_hash()
recs = json.load(open(os.path.join(root, "profiles", "pmc_workloads.json")))["workloads"]
need = ("cfg2:65536x1", "cfg2:4194304x1", "cfg3:262144x30", "cfg4:131072x32", "cfg5:262144x16", "cfg5full:262144x16")
bad = 0
for case, r in recs.items():
    fresh = r.get("device_source_hash") == now
    print(f"{'fresh' if fresh else 'STALE':6s} {case:40s} {r.get('files')}")
    bad += (case in need) and not fresh
for case in need:
    if case not in recs:
        print("MISSING", case)
        bad += 1
sys.exit(1 if bad else 0)
