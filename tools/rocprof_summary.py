#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 run (rocpd .db from `--kernel-trace`, or a *_kernel_stats.csv) as a small
markdown table for profiles/.

    rocprof_summary.py <db-or-csv> [title]            one row per kernel
    rocprof_summary.py <db> [title] --by-grid         one row per (kernel, grid size): the shapes of one object
"""
import csv
import re
import sqlite3
import sys


def short_name(n):
    """kernel name with template arguments, without return type / namespaces / parameter list"""
    m = re.search(r"((?:\w+::)*)(\w+)(<[^()]*>)?\s*\(", n)
    if m:
        return (m.group(2) + (m.group(3) or ""))[:60]
    return n.split("(")[0][-60:]


def rows_from_db(path, by_grid):
    cur = sqlite3.connect(path).cursor()
    if by_grid:
        q = ("select name, grid_x, workgroup_x, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels "
             "group by name, grid_x, workgroup_x order by 5 desc")
        rows = list(cur.execute(q))
        tot = sum(r[4] for r in rows) or 1.0
        return [("%s  [%d x %d]" % (short_name(n), gx // max(wx, 1), wx), c, t, a, 100.0 * t / tot) for n, gx, wx, c, t, a in rows]
    q = "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by name order by 3 desc"
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows) or 1.0
    return [(short_name(n), c, t, a, 100.0 * t / tot) for n, c, t, a in rows]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((short_name(r["Name"]), int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                    float(r["Percentage"])))
    return out


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    by_grid = "--by-grid" in sys.argv
    path = args[0]
    title = args[1] if len(args) > 1 else path
    rows = rows_from_db(path, by_grid) if path.endswith(".db") else rows_from_csv(path)
    print("# %s\n" % title)
    print("| kernel%s | calls | total us | avg us | %% |\n|---|---|---|---|---|" % ("  [workgroups x threads]" if by_grid else ""))
    for n, c, t, a, p in rows:
        if p < 0.05:
            continue
        print("| %s | %d | %.1f | %.2f | %.1f |" % (n, c, t, a, p))


if __name__ == "__main__":
    main()
