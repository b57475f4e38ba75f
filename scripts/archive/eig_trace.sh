#!/bin/bash
# Kernel trace of scripts/eig_ab.py: per-kernel averages of the eigensolver kernels.  bash scripts/eig_trace.sh <tag> [k ...]
tag=${1:-rXX}; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
rm -rf /tmp/prof_eig
( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_eig -- python3 scripts/eig_ab.py "$@" > $out/${tag}_eig_ab.txt 2> $out/${tag}_eig_ab.err )
db=$(find /tmp/prof_eig -name "*.db" | head -1)
[ -n "$db" ] && python3 $R/profiles/summarize_rocpd.py $db > $out/${tag}_eig_kernel_stats.csv
grep -i "tridiag\|k_dc\|jacobi" $out/${tag}_eig_kernel_stats.csv | cut -c1-200
