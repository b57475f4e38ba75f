#!/bin/bash
# Per-dispatch MFMA-pipe utilisation and effective clock of the tsgemm_tn probe modes (scripts/tn_probe.py):
# is the time the streamed operand costs lost to a lower clock (power) or to an idle matrix pipe?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmc_probe
( cd $R && rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/pmc_probe -- python3 scripts/tn_probe.py > /tmp/pmc_probe.log 2>&1 )
f=$(find /tmp/pmc_probe -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    d = agg.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "dur": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for did, d in sorted(agg.items()):
    if "tsgemm_tn" not in d["name"] or d["dur"] < 1e6: continue
    cyc = d["GRBM_GUI_ACTIVE"] / 8.0
    print("%5d %-40s %8.3f ms  clock %.3f GHz  mfma busy %.3f" % (did, d["name"][:40], d["dur"] / 1e6, cyc / d["dur"], d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)))
PY
