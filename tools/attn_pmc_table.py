#!/usr/bin/env python3
"""per (kernel, grid) sums of the counters of rocprofv3 --pmc passes (csv output), and the ratios that say what a kernel waits for"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short_name(n):
    m = re.search(r"((?:\w+::)*)(\w+)(<[^()]*>)?\s*\(", n)
    return (m.group(2) + (m.group(3) or "")) if m else n.split("(")[0][-60:]


tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            key = (short_name(row["Kernel_Name"]), int(row["Grid_Size"]) // max(1, int(row["Workgroup_Size"])))
            tot[key][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[key][row["Counter_Name"]] += 1
print("# round 4: SQ counters of the attention kernels (rocprofv3 --pmc, three passes of 8 counters; sums over the launches of a 2-step run of 4 objects, per pass)\n")
names = sorted({c for k in tot for c in tot[k]})
for key in sorted(tot, key=lambda k: -tot[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    n = min(cnt[key].values())          # launches of one pass (a counter that sits in several passes was summed over all of them)
    t = {c: v * n / cnt[key][c] for c, v in tot[key].items()}
    print("## %s  [%d workgroups], %d launches per pass\n" % (key[0], key[1], n))
    for c in names:
        if c in t:
            print("* %s = %.4g" % (c, t[c]))
    wc = t.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print()
        if "SQ_VALU_MFMA_BUSY_CYCLES" in t and "GRBM_GUI_ACTIVE" in t:
            print("* MFMA pipe busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024) / (GRBM_GUI_ACTIVE / 8) = %.1f %%" % (100 * (t["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (t["GRBM_GUI_ACTIVE"] / 8)))
        for c, label in (("SQ_ACTIVE_INST_VALU", "VALU instructions in flight"), ("SQ_ACTIVE_INST_ANY", "any instruction in flight"),
                         ("SQ_WAIT_INST_ANY", "issue stalls"), ("SQ_WAIT_ANY", "parked in s_waitcnt / barrier"), ("SQ_ACTIVE_INST_LDS", "LDS instructions in flight"),
                         ("SQ_WAIT_INST_LDS", "LDS issue stalls")):
            if c in t:
                print("* %s / SQ_WAVE_CYCLES = %.1f %%  (%s)" % (c, 100 * t[c] / wc, label))
        if "SQ_INSTS_VALU" in t and "SQ_INSTS_MFMA" in t and t["SQ_INSTS_MFMA"]:
            print("* VALU instructions per MFMA = %.2f" % (t["SQ_INSTS_VALU"] / t["SQ_INSTS_MFMA"]))
        if "SQ_INSTS_VALU_TRANS_F32" in t and "SQ_INSTS_MFMA" in t and t["SQ_INSTS_MFMA"]:
            print("* transcendental instructions per MFMA = %.2f" % (t["SQ_INSTS_VALU_TRANS_F32"] / t["SQ_INSTS_MFMA"]))
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in t and "SQ_VALU_MFMA_BUSY_CYCLES" in t:
            print("* cycles with VALU and MFMA executing together / MFMA busy cycles = %.1f %%" % (100 * t["SQ_VALU_MFMA_COEXEC_CYCLES"] / t["SQ_VALU_MFMA_BUSY_CYCLES"]))
        if "SQ_INSTS_VALU" in t and "GRBM_GUI_ACTIVE" in t:
            tr = t.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            # a wave-64 VALU instruction occupies its SIMD for 2 cycles (4 for the quarter-rate transcendentals' 16 lanes/clk: 8)
            print("* VALU pipe busy, estimated = ((SQ_INSTS_VALU - trans) x 2 + trans x 8 cycles) / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8) = %.1f %%"
                  % (100 * ((t["SQ_INSTS_VALU"] - tr) * 2 + tr * 8) / 1024 / (t["GRBM_GUI_ACTIVE"] / 8)))
    print()
