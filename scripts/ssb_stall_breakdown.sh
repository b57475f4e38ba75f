#!/bin/bash
# Round 6, VERDICT r5 item 5: NAME what stalls k_tsgemm_ssb at the ridge (G = X^T Omega, N = 1e6, k = 138, n = 48 / 64 / 96 / 138): issue-side
# counters of the product kernel, one counter group per rocprofv3 run (the program directly after --), on scripts/ss_ridge_probe.py.
#   bash scripts/ssb_stall_breakdown.sh <tag>     -> gpurun_out/<tag>_ssb_stall_breakdown.json + a table on stdout
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ssb_stall; mkdir -p /tmp/ssb_stall
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT"; do
  i=$((i+1))
  ( cd $R && rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/ssb_stall/run_$i -- python3 scripts/ss_ridge_probe.py > /dev/null 2>/tmp/ssb_stall/err_$i )
  mkdir -p /tmp/ssb_stall/pmc_$i
  f=$(find /tmp/ssb_stall/run_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f /tmp/ssb_stall/pmc_$i/counter_collection.csv || { echo "group $i ($grp) failed:"; tail -3 /tmp/ssb_stall/err_$i; }
done
python3 $R/profiles/summarize_pmc.py /tmp/ssb_stall 0.1 > $out/${tag}_ssb_stall_breakdown.json 2>$out/${tag}_ssb_stall_breakdown.err
python3 - <<PY
import json
d = json.load(open("$out/${tag}_ssb_stall_breakdown.json"))
for k, v in sorted(d.items(), key=lambda kv: kv[1]["avg_duration_ms"]):
    if "ssb" not in k: continue
    c = v["raw_mean_counters"]
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(k[:70], "avg %.3f ms" % v["avg_duration_ms"], "clock %.3f" % v.get("effective_clock_ghz", 0), "mfma util %.3f" % v.get("mfma_pipe_util", 0))
    for name in sorted(c):
        print("    %-34s %.5g   (/wave-cycles %.4f)" % (name, c[name], c[name] / wc))
PY
