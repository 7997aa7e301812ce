#!/usr/bin/env python3
"""Instruction counts of the attention kernels per variant (round 6): rocprofv3 --pmc SQ_INSTS_VALU
SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS (csv output), one directory per run; prints VALU / SALU / LDS instructions per MFMA
for every (kernel, grid) of every run side by side.

    python tools/attn_insts_table.py <dir of variant 0> <dir of variant 1> ...
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short_name(n):
    m = re.search(r"((?:\w+::)*)(\w+)(<[^()]*>)?\s*\(", n)
    return (m.group(2) + (m.group(3) or "")) if m else n.split("(")[0][-60:]


print("# attention kernels: instructions per MFMA by the SQ counters (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS)\n")
print("| run | kernel [workgroups] | launches | VALU / MFMA | SALU / MFMA | LDS / MFMA | MFMA instructions |")
print("|---|---|---|---|---|---|---|")
for d in sys.argv[1:]:
    tot = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            key = (short_name(row["Kernel_Name"]), int(row["Grid_Size"]) // max(1, int(row["Workgroup_Size"])))
            tot[key][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "SQ_INSTS_MFMA":
                cnt[key] += 1
    for key in sorted(tot, key=lambda k: -tot[k].get("SQ_INSTS_MFMA", 0)):
        t = tot[key]
        m = t.get("SQ_INSTS_MFMA", 0)
        if not m:
            continue
        print("| %s | %s [%d] | %d | %.2f | %.2f | %.2f | %.4g |" % (os.path.basename(d.rstrip("/")), key[0], key[1], cnt[key],
              t.get("SQ_INSTS_VALU", 0) / m, t.get("SQ_INSTS_SALU", 0) / m, t.get("SQ_INSTS_LDS", 0) / m, m))
