#!/bin/bash
# Kernel timeline of one step.  Default: the per-GPU share of the 8-GPU config-4 run (64 of the 512 samples on one GPU): what
# does NOT scale.  bash scripts/shard_profile.sh <name> <bench.py arguments> traces another workload (e.g. pod --workload pod).
name=${1:-shard}; [ $# -gt 0 ] && shift
args=${@:-"--samples-total 64"}
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out; mkdir -p $out
rm -rf /tmp/prof_shard
( cd $R && rocprofv3 --kernel-trace --stats -d /tmp/prof_shard -- python3 bench.py $args --headline-only --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-literal > $out/${name}_bench.json 2> $out/${name}_prof.err )
db=$(find /tmp/prof_shard -name "*.db" | head -1)
python3 $R/profiles/summarize_rocpd.py $db > $out/${name}_kernel_stats.csv
python3 - <<PY
import sqlite3, json
db = sqlite3.connect("$db"); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = [r for r in cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start") if "k_bench" not in r[0]]
# The last TIMED step: bench.py runs its timed steps with events on the big contractions only and then up to three extra steps with an
# event pair per phase (each pair costs ~5 us of idle GPU between two dependent kernels) -- the timeline of one of THOSE, which this
# script printed up to round 4, overstates the gaps of the steps that are timed.  A step = 3 big launches (tn, nn, tn).
big = [i for i, r in enumerate(rows) if ("k_tsgemm_tn" in r[0] or "k_tsgemm_nn" in r[0]) and (r[2] - r[1]) > 2e6]
extra = 3                                   # min(3, --steps) per-phase steps follow the timed region
start, stop = big[-3 * (extra + 1)], big[-3 * extra]
t0 = rows[start][1]
prev_end = t0
gaps = small = bigt = 0.0
print("timeline of the last TIMED step (us from its first big launch): name, start, duration, gap before")
in_step = True
for r in rows[start:stop]:
    name = r[0].split("(")[0][:60]
    dur, gap = (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3
    print("%-60s %10.1f %9.1f %8.1f" % (name, (r[1] - t0) / 1e3, dur, gap))
    side = "copyBuffer" in name or "fillBuffer" in name     # copies of the auxiliary stream / the next step's preparation: beside the solve
    if not side and gap > 500.0:
        in_step = False                     # the host is between two steps
    if in_step and not side:
        gaps += max(gap, 0.0)
        if dur > 2000: bigt += dur
        else: small += dur
        prev_end = max(prev_end, r[2])
print("sum over the kernels of the solve's stream: three contractions %.1f us, other kernels %.1f us, gaps between them %.1f us" % (bigt, small, gaps))
PY
