"""Print the rows of DESIGN.md section 5 from a round's files:  python scripts/design_numbers.py r04g [dir, default profiles]"""
import json, os, sys
tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "profiles"


def last(name):
    return json.loads(open(os.path.join(src, "%s_%s.json" % (tag, name))).read().strip().splitlines()[-1])


pmc = json.load(open(os.path.join("profiles", "pmc_traffic.json")))
print("build tag", pmc["build_tag"])
for w, label in (("as", "config 4"), ("pod", "config 3"), ("kle", "config 2")):
    d = last("bench_" + w)
    r = d["roofline"]
    cb = d["cpu_baseline"]
    print("%s: %.2f ms  %.4f GDoF*rank/s  dominant %s: %.2f ms, %.1f TF, %.3f; traffic %s / %.2f GB; eig err %.1e; literal %s; cpu blas3 %.1e / %.1e ref-style %.1e / %.1e"
          % (label, d["ms_per_step"], d["value"], r["kernel"], r["avg_launch_ms"], r["achieved"], r["frac"],
             ("%.2f" % (r["traffic"] / 1e9)) if r.get("traffic") else "null", r["algorithmic_bytes_per_launch"] / 1e9,
             d["parity"]["eig_rel_err_vs_oracle"], d.get("literal_T_ms_per_step"),
             cb["blas3"]["threads_all"]["value"], cb["blas3"]["threads_1"]["value"],
             cb["reference_style"]["threads_all"]["value"], cb["reference_style"]["threads_1"]["value"]))
    print("    phases", {k: round(v, 2) for k, v in d["phases_ms_per_step"].items() if v}, " peaks", {k: round(v, 1) for k, v in d["device_peaks_measured"].items()})
for name in ("bench_as_shard64", "bench_as_shard64_dist1"):
    d = last(name)
    print("%s: %.2f ms (%.3f) roof %.3f literal %.2f" % (name, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["literal_T_ms_per_step"]),
          {k: round(v, 3) for k, v in d["phases_ms_per_step"].items() if "allreduce" in k})
for k, v in pmc["kernels"].items():
    print("pmc", k, "%.2f GB busy %.3f clock %.3f" % (v["hbm_bytes_per_launch"] / 1e9, v["mfma_pipe_util"], v["effective_clock_ghz"]))
rows = json.load(open(os.path.join(src, tag + "_kernel_point.json")))
for r in rows:
    print("kp n=%d %s %.3f ms hbm %.3f mfma %.3f %s" % (r["n"], r["orientation"], r["ms"], r["hbm_frac_of_8TBs"], r["mfma_frac_of_78.6"], r["binding_roof"]))
kp = json.load(open(os.path.join(src, tag + "_pmc_kernel_point_summary.json")))
for k, v in kp.items():
    if "randn" in k or "tsgemm_ss" in k:
        print("kp-pmc %s: %.3f ms busy %.2f clock %.2f GB %.2f (n %d)" % (k[:40], v["avg_duration_ms"], v.get("mfma_pipe_util", 0), v.get("effective_clock_ghz", 0), v.get("hbm_bytes", 0) / 1e9, v["launches_sampled"]))
for line in open(os.path.join(src, tag + "_kernel_point_kernel_stats.csv")):
    if "randn" in line:
        f = line.split(",")
        print("kp-trace k_randn: %s launches, %.3f ms total" % (f[1], int(f[2]) / 1e6))
