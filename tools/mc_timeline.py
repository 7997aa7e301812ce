#!/usr/bin/env python3
"""Per-launch timeline of ONE marching-cubes call from a rocprofv3 kernel trace (rocpd .db) of tools/bench_mc.py: every device
operation of the last call (the zero-fill of the workspace, classify, scan, vertices, faces), its start relative to the first one,
its duration and the idle gap in front of it -- where the time between "grid in HBM" and "mesh in HBM" goes.

    rocprofv3 --kernel-trace --memory-copy-trace? no: --kernel-trace only -- python tools/bench_mc.py --field blob --iters 5
    python tools/mc_timeline.py <db> [title]
"""
import sqlite3
import sys

import re


def short_name(n):
    m = re.search(r"(mc_\w+(?:<[^>]*>)?)", n)
    if m:
        return m.group(1)
    m = re.search(r"(__amd_rocclr_\w+)", n)
    return m.group(1) if m else n.split("(")[0][-50:]


def main():
    db, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
    names = [short_name(r[0]) for r in rows]
    idx = [i for i, n in enumerate(names) if n.startswith("mc_classify")]
    if not idx:
        raise SystemExit("no mc_classify kernel in the trace")
    # the second-to-last call: from the zero-fill in front of its classify kernel to the last mc_* kernel before the next call's
    if len(idx) < 2:
        raise SystemExit("need at least two calls in the trace")
    def call_start(c):
        f = c
        while f > 0 and ("fillBuffer" in names[f - 1] or "memset" in names[f - 1].lower()) and rows[c][1] - rows[f - 1][1] < 200000:
            f -= 1
        return f
    c = idx[-2]
    first, nxt = call_start(c), call_start(idx[-1])
    last = max(i for i in range(first, nxt) if names[i].startswith("mc_"))
    t0 = rows[first][1]
    print("# %s\n" % title)
    print("| operation [workgroups x threads] | start, us | duration, us | idle gap in front, us |\n|---|---|---|---|")
    prev_end, busy = None, 0.0
    for i in range(first, last + 1):
        n, s, e, gx, wx = rows[i]
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        busy += (e - s) / 1e3
        print("| %s [%d x %d] | %.1f | %.1f | %.1f |" % (names[i], gx // max(wx, 1), wx, (s - t0) / 1e3, (e - s) / 1e3, gap))
        prev_end = e
    span = (rows[last][2] - t0) / 1e3
    print("\nfirst start -> last end: %.1f us; kernels busy %.1f us; idle between them %.1f us" % (span, busy, span - busy))
    # distribution of the classify kernel over all calls of the run
    d = sorted((rows[i][2] - rows[i][1]) / 1e3 for i in idx)
    print("mc_classify over the run's %d calls: median %.1f us, min %.1f us" % (len(d), d[len(d) // 2], d[0]))


if __name__ == "__main__":
    main()
