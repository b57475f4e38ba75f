#!/bin/bash
# Arbitrary PMC groups over one bench workload (one rocprofv3 run per quoted group), mean per launch of kernels > MIN_MS:
#   MIN_MS=2 bash scripts/pmc_custom.sh as "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
w=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out; rm -rf /tmp/pmcc; mkdir -p /tmp/pmcc
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmc_run
  ( cd $R && rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pmc_run -- python3 bench.py --workload $w --steps 1 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2>&1 )
  mkdir -p /tmp/pmcc/pmc_$i
  f=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f /tmp/pmcc/pmc_$i/counter_collection.csv
done
python3 $R/profiles/summarize_pmc.py /tmp/pmcc ${MIN_MS:-2.0} > $out/pmc_custom_$w.json
python3 - <<PY
import json
d = json.load(open("$out/pmc_custom_$w.json"))
for k, v in d.items():
    if "tsgemm" not in k: continue
    print(k, "n=%d avg %.3f ms" % (v["launches_sampled"], v["avg_duration_ms"]))
    for c, x in sorted(v["raw_mean_counters"].items()): print("    %-32s %.4g" % (c, x))
PY
