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
    legs = lambda leg: " / ".join("%s %.1e" % (k.replace("threads_", ""), v["value"]) for k, v in cb[leg].items() if isinstance(v, dict) and "value" in v)
    print("%s: %.2f ms  %.4f GDoF*rank/s  dominant %s: %.2f ms, %.1f TF, %.3f; traffic %s / %.2f GB; eig err %.1e; literal %s; cpu blas3 [%s] best %s; ref-style [%s] best composed %.1e"
          % (label, d["ms_per_step"], d["value"], r["kernel"], r["avg_launch_ms"], r["achieved"], r["frac"],
             ("%.2f" % (r["traffic"] / 1e9)) if r.get("traffic") else "null", r["algorithmic_bytes_per_launch"] / 1e9,
             d["parity"]["eig_rel_err_vs_oracle"], d.get("literal_T_ms_per_step"),
             legs("blas3"), cb.get("best_setting"), legs("reference_style"), cb["reference_style"].get("value", float("nan"))))
    print("    host: %s threads, %s physical cores, %s sockets; reference-style components (best setting each): %s"
          % (cb.get("threads"), cb.get("physical_cores"), cb.get("sockets"), cb["reference_style"].get("best_per_component_seconds")))
    print("    phases", {k: round(v, 2) for k, v in d["phases_ms_per_step"].items() if v}, " peaks", {k: round(v, 1) for k, v in d["device_peaks_measured"].items()})
for name in ("bench_as_shard64", "bench_as_shard64_dist1"):
    d = last(name)
    print("%s: %.2f ms (%.3f) roof %.3f literal %.2f" % (name, d["ms_per_step"], d["value"], d["roofline"]["frac"], d["literal_T_ms_per_step"]),
          {k: round(v, 3) for k, v in d["phases_ms_per_step"].items() if "allreduce" in k})
for k, v in pmc["kernels"].items():
    print("pmc", k, "%.2f GB busy %s clock %s" % (v["hbm_bytes_per_launch"] / 1e9, v.get("mfma_pipe_util"), v.get("effective_clock_ghz")))
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
d = last("bench_default")
print("default line: %.2f ms, extras:" % d["ms_per_step"], d.get("extra_keys"), {k: round(v, 1) for k, v in d.get("extras_wall_seconds", {}).items()})
for key in ("config3", "config2", "shard64", "shard64_rccl_1rank"):
    e = d.get(key, {})
    print("   %s: %s ms, roofline %s, parity %s" % (key, e.get("ms_per_step"), (e.get("roofline") or {}).get("frac"), (e.get("parity") or {}).get("eig_rel_err_vs_oracle")))
print("   kernel point:", [(r["n"], round(r["ms"], 3), r["bound"], round(r["frac"], 3)) for r in d.get("kernel_point", {}).get("rows", [])])
r8 = last("bench_as_8ranks_one_gpu")
print("8 ranks on one GPU: %.2f ms, parity %.1e, communicator %s" % (r8["ms_per_step"], r8["parity"]["eig_rel_err_vs_oracle"], {k: r8["communicator"].get(k) for k in ("ranks", "transport", "p2p_sync", "p2p_probe_generations")}))
