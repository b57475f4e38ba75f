#!/bin/bash
# kernel trace of the config-2 bench (2 steps): per-kernel totals via profiles/summarize_rocpd.py
tag=${1:-tmp}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/prof_kle
( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_kle -- python3 bench.py --workload kle --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-literal > $R/gpurun_out/${tag}_kle_bench.json 2> $R/gpurun_out/${tag}_kle_prof.err )
db=$(find /tmp/prof_kle -name "*.db" | head -1)
[ -n "$db" ] && python3 $R/profiles/summarize_rocpd.py $db > $R/gpurun_out/${tag}_kle_kernel_stats.csv
head -30 $R/gpurun_out/${tag}_kle_kernel_stats.csv
