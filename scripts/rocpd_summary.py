#!/usr/bin/env python
"""Summarise the kernel dispatches of a rocprofv3 run (rocpd sqlite output) as CSV:
name, calls, total_ns, avg_ns, min_ns, max_ns, percent -- the same columns as
`rocprofv3 --stats`' kernel_stats.csv.  Usage: rocpd_summary.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = cur.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
        f"max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name "
        f"order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n")
    for r in rows:
        out.write('"%s",%d,%d,%.1f,%d,%d,%.2f\n' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))


if __name__ == "__main__":
    main()
