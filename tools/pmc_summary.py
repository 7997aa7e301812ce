#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv: per kernel (name up to the parameter list) the number of
dispatches and the average counter value per dispatch.   usage: pmc_summary.py <dir-or-csv> [regex]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    rx = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if rx and not rx.search(name):
                continue
            m = re.search(r"(\w+)(<[^()]*>)?\s*\(", name)
            short = (m.group(1) + (m.group(2) or "")) if m else name[:60]
            k = (short, r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    print("| kernel | counter | dispatches | average per dispatch | total |\n|---|---|---|---|---|")
    tot = defaultdict(lambda: [0, 0.0])
    for (short, cn), (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %s | %d | %.1f | %.4g |" % (short, cn, n, v / n, v))
        fam = short.split("<")[0]
        tot[(fam, cn)][0] += n
        tot[(fam, cn)][1] += v
    print()
    for (fam, cn), (n, v) in sorted(tot.items()):
        print("family %s %s: %d dispatches, average %.1f, total %.4g" % (fam, cn, n, v / n, v))


if __name__ == "__main__":
    main()
