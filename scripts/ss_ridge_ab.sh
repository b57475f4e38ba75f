#!/bin/bash
# Round-5 bounded experiment on the ridge shapes of k_tsgemm_ssb:  bash scripts/ss_ridge_ab.sh <tag>
#   1. the product library against the two timing probes (libhfmi_ssp1.so: second operand not written to LDS; libhfmi_ssp3.so: nor
#      read from it), interleaved, three rounds                               -> <tag>_ssb_ridge_ab.txt
#   2. LDS / MFMA / clock counters of the product kernel on the same script   -> <tag>_ssb_ridge_pmc.json
# The probe libraries are built beforehand with  bash scripts/build_variant.sh ssp1 hfmi_skinny.hip -DSS_PROBE=1  (and ssp3 / =3).
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
cd $R
: > $out/${tag}_ssb_ridge_ab.txt
for round in 1 2 3; do
  for lib in libhfmi.so build/libhfmi_ssp1.so build/libhfmi_ssp3.so; do
    HFMI_LIB=$R/hippyflow_amd/$lib timeout 300 python scripts/ss_ridge_probe.py 2>/dev/null >> $out/${tag}_ssb_ridge_ab.txt
  done
done
cat $out/${tag}_ssb_ridge_ab.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ssb_pmc; mkdir -p /tmp/ssb_pmc
i=0
for grp in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  ( cd $R && rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/ssb_pmc/run_$i -- python3 scripts/ss_ridge_probe.py > /dev/null 2>/tmp/ssb_pmc/err_$i )
  mkdir -p /tmp/ssb_pmc/pmc_$i
  f=$(find /tmp/ssb_pmc/run_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f /tmp/ssb_pmc/pmc_$i/counter_collection.csv || tail -3 /tmp/ssb_pmc/err_$i
done
python3 $R/profiles/summarize_pmc.py /tmp/ssb_pmc 0.1 > $out/${tag}_ssb_ridge_pmc.json 2>$out/${tag}_ssb_ridge_pmc.err
python3 - <<PY
import json
d = json.load(open("$out/${tag}_ssb_ridge_pmc.json"))
for k, v in sorted(d.items(), key=lambda kv: kv[1]["avg_duration_ms"]):
    if "ssb" not in k: continue
    c = v["raw_mean_counters"]
    print(k[:70], "avg %.3f ms" % v["avg_duration_ms"], "clock %.3f" % v.get("effective_clock_ghz", 0), "mfma util %.3f" % v.get("mfma_pipe_util", 0))
    for name in sorted(c):
        print("    %-28s %.5g" % (name, c[name]))
PY
