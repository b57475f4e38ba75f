#!/bin/bash
# Round 6, VERDICT r5 item 2: ONE bounded experiment on the headline's dominant kernel aimed at joules per flop, not at stalls -- a TALLER
# workgroup tile of k_tsgemm_tn (knob "waves": 8 = two waves per SIMD, 16 accumulator tiles each [default]; 4 = one wave per SIMD with up to
# 32 tiles: the shared operand W is then fetched from L2 by half as many row blocks; 44 = two 4-wave workgroups per CU), same box, interleaved,
# three rounds of un-profiled config-4 lines, then one PMC pass each (clock + MFMA busy + HBM bytes).   bash scripts/tn_tile_ab.sh <tag>
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
cd $R
: > $out/${tag}_tn_tile_ab.txt
for round in 1 2 3; do
  for wv in 8 4 44; do
    HFMI_TUNE=waves=$wv timeout 300 python bench.py --headline-only --steps 6 --warmup 2 --no-cpu-baseline --no-check --no-literal 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
r = d['roofline']
print('waves=$wv round $round  ms/step %.3f  tn avg launch %.3f ms  %.2f TF  frac %.4f  kernels %s' % (d['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac'], [(k['kernel'], round(k['avg_launch_ms'], 3)) for k in d.get('kernels', [])[:3]]))
" >> $out/${tag}_tn_tile_ab.txt
  done
done
for wv in 8 4 44; do
  HFMI_TUNE=waves=$wv MIN_MS=2.0 bash scripts/pmc_workload.sh as ${tag}_waves$wv >> $out/${tag}_tn_tile_ab.txt 2>&1
done
cat $out/${tag}_tn_tile_ab.txt
