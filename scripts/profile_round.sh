#!/bin/bash
# Per-round rocprofv3 evidence (run on the GPU box from the repo root):  bash scripts/profile_round.sh r01d
# kernel traces of the three workloads + the kernel-point script, then PMC passes (own runs) on config 4 and on
# the kernel point.  Summaries land in gpurun_out/<tag>_*; copy the ones to keep into profiles/.
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=/root/repo
out=$R/gpurun_out
mkdir -p $out
for w in as pod kle; do
  rm -rf /tmp/prof_$w
  ( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_$w -- python3 bench.py --workload $w --headline-only --steps 3 --warmup 1 --no-cpu-baseline --no-check > $out/${tag}_${w}_bench.json 2> $out/${tag}_${w}_prof.err )
  db=$(find /tmp/prof_$w -name "*.db" | head -1)
  [ -n "$db" ] && python3 $R/profiles/summarize_rocpd.py $db > $out/${tag}_${w}_kernel_stats.csv
done
rm -rf /tmp/prof_kp
( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_kp -- python3 scripts/kernel_point.py > $out/${tag}_kernel_point.log 2>&1 )
db=$(find /tmp/prof_kp -name "*.db" | head -1)
[ -n "$db" ] && python3 $R/profiles/summarize_rocpd.py $db > $out/${tag}_kernel_point_kernel_stats.csv
# PMC passes: one counter group per run, kernel-trace only
for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/pmc_kp_$name
  ( cd $R && rocprofv3 --kernel-trace --output-format csv --pmc $grp -d /tmp/pmc_kp_$name -- python3 scripts/kernel_point.py > /dev/null 2>&1 )
  mkdir -p $out/${tag}_pmc_kp/pmc_$name
  f=$(find /tmp/pmc_kp_$name -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $out/${tag}_pmc_kp/pmc_$name/counter_collection.csv
done
python3 $R/profiles/summarize_pmc.py $out/${tag}_pmc_kp 0.1 > $out/${tag}_pmc_kernel_point_summary.json 2>$out/${tag}_pmc_kp.err
ls -la $out | tail -20
