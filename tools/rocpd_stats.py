#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel count / total / mean / min / max (us).
Usage: python tools/rocpd_stats.py <results.db> [--grid]   (prints a markdown table)"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "kernel_name" if "kernel_name" in scols else "display_name"
    q = ("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
         "max(d.grid_size_x), max(d.workgroup_size_x) from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc"
         % (name_col, kd, ks, name_col))
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | %% | mean us | min us | max us | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|")
    for n, cnt, s, a, mn, mx, gx, wx in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"void ", "", n)
        if len(n) > 110:
            n = n[:107] + "..."
        print("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f | %s | %s |" % (n, cnt, s / 1e6, 100.0 * s / tot, a / 1e3, mn / 1e3, mx / 1e3, gx, wx))


if __name__ == "__main__":
    main()
