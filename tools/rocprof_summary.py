#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats) of a rocpd .db or *_kernel_stats.csv
as a small markdown table for profiles/.   usage: rocprof_summary.py <db-or-csv> [title] > profiles/x.md"""
import csv
import sqlite3
import sys


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    return [(n, c, t, a, p) for n, c, t, a, p in
            cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels")]


def rows_from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3,
                    float(r["Percentage"])))
    return out


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else path
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    print("# %s\n" % title)
    print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
    for n, c, t, a, p in rows:
        short = n.split("(")[0].replace("(anonymous namespace)::", "")
        if not short:
            short = n[:60]
        print("| %s | %d | %.1f | %.2f | %.1f |" % (short[-70:], c, t, a, p))


if __name__ == "__main__":
    main()
