#!/bin/bash
# bash scripts/small_k_trace.sh <tag> [k ...]: kernel trace of scripts/small_k_point.py, one CSV per k
tag=${1:-rXX}; shift
ks=${@:-"138 148 200 256"}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
for k in $ks; do
  rm -rf /tmp/prof_sk
  ( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_sk -- python3 scripts/small_k_point.py $k > $out/${tag}_small_k$k.txt 2> $out/${tag}_small_k$k.err )
  db=$(find /tmp/prof_sk -name "*.db" | head -1)
  [ -n "$db" ] && python3 $R/profiles/summarize_rocpd.py $db > $out/${tag}_small_k${k}_kernel_stats.csv
  echo "== k=$k"; cat $out/${tag}_small_k$k.txt
  grep -i "chol\|tridiag\|k_dc\|jacobi" $out/${tag}_small_k${k}_kernel_stats.csv | cut -d, -f1,2,4 | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-90
done
