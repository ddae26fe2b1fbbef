#!/usr/bin/env python3
"""Writes tests/golden/context.npz: host sessions / controller scripts and what the reference's own AudioContext and EventsToSignals
(oracle/_ref/libdropin_ref.so, compiled from /root/reference by oracle/Makefile) make of them.   python tests/golden/make_golden_context.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import ctypes  # noqa: E402

import test_oracle_context as t  # noqa: E402


def main():
    Lr = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libdropin_ref.so"))
    d = {}
    sessions = [(7, 48000.0), (8, 44100.0), (9, 96000.0), (10, 48000.0)]
    for k, (seed, sr) in enumerate(sessions):
        rows = t.host_session(seed, sr=sr)
        d[f"session{k}"] = rows
        d[f"phase{k}"], d[f"since{k}"] = t.run_transport(Lr, "transport_ref_run", rows)
    d["n_sessions"] = len(sessions)
    scripts = [(3, 48000.0), (4, 44100.0), (5, 8000.0), (6, 192000.0)]
    for k, (seed, sr) in enumerate(scripts):
        values, awake_from, events = t.controller_script(seed, 80)
        d[f"values{k}"], d[f"awake{k}"], d[f"sr{k}"] = values, awake_from, sr
        d[f"ctl{k}"] = t.ref_controller(Lr, events, 80, sr)
    d["n_scripts"] = len(scripts)
    np.savez_compressed(os.path.join(HERE, "context.npz"), **d)
    print("wrote context.npz:", sorted(d))


if __name__ == "__main__":
    main()
