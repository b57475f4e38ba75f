#!/bin/bash
# bash scripts/eig_corner_trace.sh <tag> [k ...]: kernel trace of scripts/eig_corner.py, one CSV per k
tag=${1:-rXX}; shift
ks=${@:-"160 192 224 256"}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
for k in $ks; do
  rm -rf /tmp/prof_ec
  ( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_ec -- python3 scripts/eig_corner.py $k > $out/${tag}_eig_k$k.txt 2> $out/${tag}_eig_k$k.err )
  db=$(find /tmp/prof_ec -name "*.db" | head -1)
  [ -n "$db" ] && python3 $R/profiles/summarize_rocpd.py $db > $out/${tag}_eig_k${k}_kernel_stats.csv
  echo "== k=$k"; cat $out/${tag}_eig_k$k.txt
  grep -i "tridiag\|k_dc" $out/${tag}_eig_k${k}_kernel_stats.csv | cut -d, -f1,2,4 | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-90
done
