#!/bin/bash
# kernel-level durations of the small eigensolver over scripts/jacobi_ab.py (first two Jacobi kernels per matrix size)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pj
rocprofv3 --kernel-trace --stats -d /tmp/pj -- python3 $R/scripts/jacobi_ab.py > /tmp/pj.log 2>&1
db=$(find /tmp/pj -name "*.db" | head -1)
python3 - $db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
seq = [(r[0].split("(")[0][:32], round((r[2] - r[1]) / 1e3, 1)) for r in rows if "jacobi" in r[0]]
for i in range(0, len(seq), 22): print(seq[i:i + 2])
PY
