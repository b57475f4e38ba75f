#!/usr/bin/env python3
"""Per-kernel PMC summary from rocprofv3 --pmc CSV output (counter_collection.csv): mean counter values per
launch of each kernel, plus derived figures (effective clock, MFMA pipe utilisation, HBM bytes with the gfx950
FETCH_SIZE x2 correction of MI355X_MICROARCH.md section HBM).
Usage: summarize_pmc.py <dir with pmc_*/...csv> [min kernel duration in ms, default 1.0]"""
import collections
import csv
import glob
import json
import sys


def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    names, dur = {}, {}
    for r in csv.DictReader(open(path)):
        did = r["Dispatch_Id"]
        names[did] = r["Kernel_Name"]
        dur[did] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[did][r["Counter_Name"]] += float(r["Counter_Value"])
    return agg, names, dur


def main(root, min_ms=1.0):
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(root + "/pmc_*/*counter_collection.csv")):
        agg, names, dur = load(f)
        for did, c in agg.items():
            if dur[did] < min_ms * 1e6:
                continue
            # kernel name without its argument list; kernels of an anonymous namespace demangle to "void (anonymous namespace)::k<...>(...)"
            key = names[did].replace("(anonymous namespace)::", "").split("(")[0]
            for cn, v in c.items():
                per_kernel[key][cn].append(v)
            per_kernel[key]["duration_ns"].append(dur[did])
    out = {}
    for k, c in per_kernel.items():
        m = {cn: sum(v) / len(v) for cn, v in c.items()}
        d = {"launches_sampled": len(c["duration_ns"]), "avg_duration_ms": m["duration_ns"] / 1e6}
        if "GRBM_GUI_ACTIVE" in m:
            cyc = m["GRBM_GUI_ACTIVE"] / 8.0                    # summed over the 8 XCDs
            d["effective_clock_ghz"] = cyc / m["duration_ns"]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                d["mfma_pipe_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)   # 256 CUs x 4 SIMDs
            if "SQ_INSTS_VALU_MFMA_MOPS_F64" in m:
                d["mfma_f64_flops_executed"] = m["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512
        if "FETCH_SIZE" in m:
            d["hbm_read_bytes"] = m["FETCH_SIZE"] * 1024 * 2    # KB units; x2: gfx950 counts 128-B requests as 64 B
        if "WRITE_SIZE" in m:
            d["hbm_write_bytes"] = m["WRITE_SIZE"] * 1024
        if "hbm_read_bytes" in d:
            d["hbm_bytes"] = d["hbm_read_bytes"] + d.get("hbm_write_bytes", 0.0)
        d["raw_mean_counters"] = {cn: v for cn, v in m.items() if cn != "duration_ns"}
        out[k] = d
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0)
