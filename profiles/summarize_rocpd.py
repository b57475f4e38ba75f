#!/usr/bin/env python3
"""Per-kernel summary (count, total, average, min, max, share) of a rocprofv3 --kernel-trace --stats run stored in
rocpd (SQLite) format -- the default output of rocprofv3 on ROCm 7.2.  Usage: summarize_rocpd.py results.db > out.csv"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size) "
         f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPR,AGPR,SGPR,LDS_bytes,Scratch_bytes")
    for r in rows:
        print('"%s",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s,%s' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, r[6], r[7], r[8], r[9], r[10]))


if __name__ == "__main__":
    main(sys.argv[1])
