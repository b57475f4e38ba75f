#!/bin/bash
# MFMA-busy / clock PMC pass of scripts/tn_probe.py under a given library build:  bash scripts/pmc_variant.sh <lib.so> <tag>
lib=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export HFMI_LIB=$R/$lib
out=$R/gpurun_out; mkdir -p $out/${tag}_pmcv/pmc_SQ
rm -rf /tmp/pmcv_$tag
( cd $R && rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d /tmp/pmcv_$tag -- python3 scripts/tn_probe.py > /dev/null 2>&1 )
f=$(find /tmp/pmcv_$tag -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && cp $f $out/${tag}_pmcv/pmc_SQ/counter_collection.csv
python3 $R/profiles/summarize_pmc.py $out/${tag}_pmcv 2.0 | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in sorted(d.items()):
    if 'tsgemm_tn' in k: print('%-44s n=%3d avg %.3f ms clock %.3f GHz mfma util %.3f' % (k[:44], v['launches_sampled'], v['avg_duration_ms'], v.get('effective_clock_ghz',0), v.get('mfma_pipe_util',0)))
"
rm -rf $out/${tag}_pmcv
