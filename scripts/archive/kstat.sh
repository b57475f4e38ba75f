#!/bin/bash
# Per-kernel average durations of one bench workload, filtered:  bash scripts/kstat.sh as jacobi [bench args...]
w=$1; pat=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_ks
( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-check "$@" > /dev/null 2>&1 )
db=$(find /tmp/prof_ks -name "*.db" | head -1)
python3 $R/profiles/summarize_rocpd.py $db | grep -i "$pat" | cut -c1-60,60-200 | awk -F, '{printf "%-60s calls %s avg %.1f us\n", substr($1,1,60), $2, $4/1000}'
