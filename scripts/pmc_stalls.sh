#!/bin/bash
# Where the matrix pipe's idle time goes: SQ wait / activity / queue counters of the two big contractions of one bench workload
# (one counter group per pass, raw means per launch):   bash scripts/pmc_stalls.sh <tag> [bench args]   -> gpurun_out/<tag>_pmc_stalls.json
tag=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out; rm -rf /tmp/stalls; mkdir -p /tmp/stalls
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  ( cd $R && rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/stalls/run_$i -- python3 bench.py "$@" --headline-only --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-literal > /dev/null 2>/tmp/stalls/err_$i )
  mkdir -p /tmp/stalls/pmc_$i
  f=$(find /tmp/stalls/run_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f /tmp/stalls/pmc_$i/counter_collection.csv || tail -3 /tmp/stalls/err_$i
done
python3 $R/profiles/summarize_pmc.py /tmp/stalls ${MIN_MS:-1.0} > $out/${tag}_pmc_stalls.json
python3 - <<PY
import json
d = json.load(open("$out/${tag}_pmc_stalls.json"))
for k, v in sorted(d.items(), key=lambda kv: -kv[1]["avg_duration_ms"]):
    if "tsgemm" not in k: continue
    c = v["raw_mean_counters"]
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(k[:60], "avg %.3f ms" % v["avg_duration_ms"], "clock %.3f" % v.get("effective_clock_ghz", 0), "mfma util %.3f" % v.get("mfma_pipe_util", 0))
    for name in sorted(c):
        print("    %-32s %.4g   (/wave-cycles %.4f)" % (name, c[name], c[name] / wc))
PY
