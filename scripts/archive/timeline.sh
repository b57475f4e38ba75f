#!/bin/bash
# Kernel timeline of the last step of a bench workload (name, start, duration, gap):  bash scripts/timeline.sh pod
w=${1:-pod}; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
rm -rf /tmp/prof_tl
( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_tl -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-check "$@" > $out/tl_${w}_bench.json 2> $out/tl_${w}.err )
db=$(find /tmp/prof_tl -name "*.db" | head -1)
python3 - <<PY
import sqlite3
db = sqlite3.connect("$db"); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
rows = [r for r in rows if "k_bench" not in r[0]]
big = [i for i, r in enumerate(rows) if ("k_tsgemm_tn" in r[0] or "k_tsgemm_nn" in r[0]) and (r[2] - r[1]) > 2e6]
# the last step starts at the third-from-last big launch
start = big[-3]
t0 = rows[start][1]; prev = t0
tot_gap = 0
for r in rows[start:]:
    if (r[1] - t0) / 1e3 > 60000: break
    gap = (r[1] - prev) / 1e3
    tot_gap += max(gap, 0)
    print("%-58s %9.1f %8.1f %7.1f" % (r[0].split("(")[0][:58], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap))
    prev = r[2]
print("total gaps us", tot_gap)
PY
