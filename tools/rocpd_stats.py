#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace into the --stats table: per kernel
calls / total / avg / min / max / % of GPU kernel time.  Usage: rocpd_stats.py results.db"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    rows = list(cur.execute(
        f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
        f"max(d.end-d.start), max(d.grid_size_x), max(d.workgroup_size_x), max(d.group_segment_size) "
        f"from {disp} d join {sym} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"))
    total = sum(r[2] for r in rows)
    print("| kernel | calls | total ms | avg us | min us | max us | % | grid | wg | LDS B |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx, grid, wg, lds in rows:
        print(f"| `{name[:90]}` | {n} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | "
              f"{100*tot/total:.1f} | {grid} | {wg} | {lds} |")


if __name__ == "__main__":
    main(sys.argv[1])
