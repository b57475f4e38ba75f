"""One-line digest of bench.py JSON lines:  python scripts/bench_line.py <file> [...]"""
import json
import sys

for f in sys.argv[1:]:
    lines = [ln for ln in open(f) if ln.startswith("{")]
    if not lines:
        print(f, "no JSON line")
        continue
    j = json.loads(lines[-1])
    sysfs = j.get("gpu_sysfs") or {}
    clk = [v.get("sclk") for v in (sysfs.get("after_timed_region") or {}).values()]
    print("%-44s %8.3f ms/step (median %8.3f, min/max %s)  value %.4g %s  frac %.3f  traffic %s  build %s  sclk %s"
          % (f.split("/")[-1], j["ms_per_step"], j.get("median_ms_per_step", 0.0),
             ["%.2f" % x for x in j.get("step_ms_min_max", [])], j["value"], j["unit"], j["roofline"]["frac"],
             j["roofline"].get("traffic"), j.get("build_tag"), clk))
