#!/bin/bash
# kernel-trace summary of one bench workload:  bash scripts/trace_workload.sh kle [extra bench args]
w=${1:-as}; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
rm -rf /tmp/prof_tw
( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_tw -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-check "$@" > $out/tw_${w}_bench.json 2> $out/tw_${w}.err )
db=$(find /tmp/prof_tw -name "*.db" | head -1)
python3 $R/profiles/summarize_rocpd.py $db > $out/tw_${w}_kernel_stats.csv
cut -c1-120 $out/tw_${w}_kernel_stats.csv | head -${LINES_OUT:-25}
