#!/bin/bash
# PMC passes (MFMA busy + clock, FETCH_SIZE, WRITE_SIZE; one group per run) over one bench workload:
#   bash scripts/pmc_workload.sh pod r01e
w=${1:-as}; tag=${2:-rXX}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out/${tag}_pmc_$w
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/pmc_$name
  ( cd $R && rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pmc_$name -- python3 bench.py --workload $w --headline-only --steps 1 --warmup 1 --no-cpu-baseline --no-check > /dev/null 2>&1 )
  mkdir -p $out/${tag}_pmc_$w/pmc_$name
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $out/${tag}_pmc_$w/pmc_$name/counter_collection.csv
done
python3 $R/profiles/summarize_pmc.py $out/${tag}_pmc_$w ${MIN_MS:-2.0} > $out/${tag}_pmc_${w}_summary.json
python3 - <<PY
import json
d = json.load(open("$out/${tag}_pmc_${w}_summary.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1]["avg_duration_ms"] * kv[1]["launches_sampled"]):
    print("%-44s n=%3d avg %.3f ms  clock %.3f GHz  mfma util %.3f  hbm %.3f GB" % (k[:44], v["launches_sampled"], v["avg_duration_ms"], v.get("effective_clock_ghz", 0), v.get("mfma_pipe_util", 0), v.get("hbm_bytes", 0) / 1e9))
PY
rm -rf $out/${tag}_pmc_$w
