#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the PMC summaries of one round (scripts/pmc_workload.sh <as|pod|kle> <tag> writes
gpurun_out/<tag>_pmc_<w>_summary.json):

    python profiles/make_pmc_traffic.py r02j [directory with the summaries, default gpurun_out]

bench.py reports `roofline.traffic` from this file, and only if its `build_tag` equals hfmi_build_tag() of the library
it is running -- a record measured on other kernel sources is never used."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = {"as": (51200, 74, 200000), "pod": (2048, 138, 500000), "kle": (100000, 84, 100000)}


def main(tag, src):
    from hippyflow_amd import _build
    out = {"build_tag": _build.source_tag(),
           "source": "profiles/%s_pmc_{as,pod,kle}_summary.json: rocprofv3 --kernel-trace --pmc passes (one counter group per run: "
                     "SQ_VALU_MFMA_BUSY_CYCLES+GRBM_GUI_ACTIVE+SQ_BUSY_CYCLES, FETCH_SIZE, WRITE_SIZE) of `bench.py --workload <w> --steps 1 "
                     "--warmup 1` (scripts/pmc_workload.sh); hbm bytes = FETCH_SIZE*1024*2 (gfx950 correction, MI355X_MICROARCH.md "
                     "section HBM) + WRITE_SIZE*1024, mean per launch of the launches longer than the script's threshold" % tag,
           "kernels": {}}
    for w, (m, k, N) in SHAPES.items():
        path = os.path.join(src, "%s_pmc_%s_summary.json" % (tag, w))
        if not os.path.exists(path):
            continue
        summ = json.load(open(path))
        for kind in ("tn", "nn"):
            cands = [(v["avg_duration_ms"] * v["launches_sampled"], name, v) for name, v in summ.items()
                     if (("k_tsgemm_%sI" % kind) in name or ("k_tsgemm_%s<" % kind) in name) and "hbm_bytes" in v]
            if not cands:
                continue
            _, name, v = max(cands)
            out["kernels"]["k_tsgemm_%s m=%d k=%d N=%d" % (kind, m, k, N)] = {
                "hbm_bytes_per_launch": v["hbm_bytes"], "mfma_pipe_util": v.get("mfma_pipe_util"),
                "effective_clock_ghz": v.get("effective_clock_ghz"), "avg_duration_ms_under_pmc": v["avg_duration_ms"],
                "launches_sampled": v["launches_sampled"], "kernel_symbol": name}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out"))
