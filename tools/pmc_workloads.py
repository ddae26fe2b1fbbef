#!/usr/bin/env python3
"""Build profiles/pmc_workloads.json: the PMC facts bench.py attaches to a bench line, keyed by WORKLOAD CASE
(bench.workload_key: workload, variant switches, voices x vectors) — not by kernel name: `mlgpu_graph_kernel` is the name
of every fused graph, and a strings launch moves 4x the bytes of a config-5 launch.

    tools/pmc_workloads.py <out.json> <case>@<prefix> [...]

<prefix>_traffic.json (tools/pmc_summary.py traffic) and <prefix>_pmc.txt (tools/pmc_summary.py pmc) are the per-run
summaries tools/gpu_profile_all.sh writes; <case> is e.g. cfg5:262144x16. The dominant kernel of a case is the one with
the most HBM bytes per launch among kernels launched at least 8 times."""
import json
import re
import sys


def parse_pmc_txt(path):
    out, cur = {}, None
    try:
        for line in open(path):
            m = re.match(r"== (.*?)\s+\((\d+) dispatches, mean duration ([\d.]+) us", line)
            if m:
                cur = out.setdefault(m.group(1), {"dispatches": int(m.group(2)), "mean_us_under_pmc": float(m.group(3))})
                continue
            m = re.match(r"\s+(\w+)\s+mean\s+([\d.]+)", line)
            if m and cur is not None:
                cur[m.group(1)] = float(m.group(2))
    except OSError:
        pass
    return out


def main():
    outp = sys.argv[1]
    res = {"source": "rocprofv3 --pmc passes (SQ set, FETCH_SIZE, WRITE_SIZE; --kernel-trace only) of `bench.py --workload <w> --steps 2 --warmup 1`, "
                     "tools/gpu_profile_all.sh; bytes corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x2 on gfx950)",
           "workloads": {}}
    try:
        res["workloads"] = json.load(open(outp)).get("workloads", {})
    except Exception:
        pass
    # the fingerprint of the device code these counters were measured on (bench.py drops records of another build)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import madronalib_amd as ml
    device_hash = ml.device_source_hash()
    for arg in sys.argv[2:]:
        case, prefix = arg.rsplit("@", 1)
        try:
            traffic = json.load(open(prefix + "_traffic.json")).get("kernels", {})
        except OSError:
            print("skip", case, "(no traffic file)")
            continue
        cands = {k: v for k, v in traffic.items() if v["launches"] >= 8 and "fill" not in k}
        if not cands:
            continue
        kernel = max(cands, key=lambda k: cands[k]["hbm_bytes_per_launch"])
        rec = {"kernel": kernel, "hbm_bytes_per_launch": cands[kernel]["hbm_bytes_per_launch"],
               "fetch_bytes_per_launch_raw": cands[kernel]["fetch_bytes_per_launch_corrected"] / 2.0,   # FETCH_SIZE x 1024, as counted
               "fetch_bytes_per_launch_corrected": cands[kernel]["fetch_bytes_per_launch_corrected"],   # x 2: right when every read is a coalesced stream
               "write_bytes_per_launch": cands[kernel]["write_bytes_per_launch"], "files": prefix.split("/")[-1] + "_{traffic.json,pmc.txt}"}
        rec["device_source_hash"] = device_hash
        pmc = parse_pmc_txt(prefix + "_pmc.txt")
        for k, v in pmc.items():
            if k == kernel or k in kernel or kernel in k:
                rec["mean_us_under_pmc"] = v.get("mean_us_under_pmc")
                if "SQ_INSTS_VALU" in v:
                    rec["valu_wave_insts_per_launch"] = v["SQ_INSTS_VALU"]
                for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE",
                          # the instruction classes behind roofline.valu.busy_frac (bench.py) and the scalar side
                          "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT",
                          "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                          "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM",
                          # round 5: what the vector unit was measured to do (roofline.valu.busy_measured)
                          "SQ_ACTIVE_INST_VALU2", "SQ_BUSY_CU_CYCLES", "SQ_ACTIVE_INST_SCA", "SQ_INST_CYCLES_SALU", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS"):
                    if c in v:
                        rec[c] = v[c]
                break
        # packed FP32 is counted once by SQ_INSTS_VALU_*_F32 but issues at the slow class's rate: its static share in the shipped kernel
        # (tools/kernel_mix.py; None for kernels compiled at run time - the fused graph kernels hold no packed instructions)
        try:
            import kernel_mix
            rec["packed_f32_share"] = kernel_mix.packed_share(kernel)
        except Exception as e:  # noqa: BLE001  (no llvm-objdump: the record simply lacks the field)
            print("packed share not computed:", e)
        res["workloads"][case] = rec
        print(case, kernel[:60], f"{rec['hbm_bytes_per_launch'] / 1e9:.3f} GB", rec.get("valu_wave_insts_per_launch"))
    json.dump(res, open(outp, "w"), indent=1)


if __name__ == "__main__":
    main()
