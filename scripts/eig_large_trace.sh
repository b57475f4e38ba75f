#!/bin/bash
# bash scripts/eig_large_trace.sh <tag> [n ...]: kernel trace of scripts/eig_large_time.py, one CSV per n
tag=${1:-rXX}; shift
ns=${@:-"512 1024 2048 4096"}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
for n in $ns; do
  rm -rf /tmp/prof_el
  ( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_el -- python3 scripts/eig_large_time.py $n --no-host --blas-threads=8 > $out/${tag}_eig_large_n$n.txt 2> $out/${tag}_eig_large_n$n.err )
  db=$(find /tmp/prof_el -name "*.db" | head -1)
  [ -n "$db" ] && python3 $R/profiles/summarize_rocpd.py $db > $out/${tag}_eig_large_n${n}_kernel_stats.csv
  echo "== n=$n"; cat $out/${tag}_eig_large_n$n.txt
  head -14 $out/${tag}_eig_large_n${n}_kernel_stats.csv | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-150
done
