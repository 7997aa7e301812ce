#!/usr/bin/env python3
"""MFMA-pipe utilisation per kernel family from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
GRBM_GUI_ACTIVE), weighted with the launch counts and durations of a full 50-step object from a --kernel-trace run.

    mfma_util.py --pmc <pmc dir> --trace <rocpd .db> [--objects-per-launch B]

gfx950's derived-metric XML is missing in ROCm 7.2 (MI355X_MICROARCH.md, "rocprofv3 PMC slots"), so the utilisation is formed
from the raw counters:  SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs (256 CUs x 4; 16 cycles per
16x16x32 bf16 MFMA, 32 per 32x32x16), GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs:

    mfma_busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024) / (GRBM_GUI_ACTIVE / 8)

Key = (kernel name with template arguments, workgroups), as in tools/traffic_json.py: the short PMC run (2 denoising
steps) and the full trace are matched shape by shape.  Profiled passes run at a lower clock than un-profiled ones; the
ratio of two counters of the same pass does not depend on it."""
import argparse
import csv
import glob
import os
import re
import sqlite3
from collections import defaultdict


def short_name(n):
    m = re.search(r"((?:\w+::)*)(\w+)(<[^()]*>)?\s*\(", n)
    return (m.group(2) + (m.group(3) or "")) if m else n.split("(")[0][-60:]


def family(k):
    return ("gemm" if "gemm" in k else "attention" if "attn" in k else "layernorm" if "layernorm" in k or "ln_dot" in k
            else "marching cubes" if k.startswith("mc_") else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", required=True)
    ap.add_argument("--trace", required=True)
    ap.add_argument("--objects-per-launch", type=int, default=1)
    a = ap.parse_args()
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(a.pmc, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            c = acc[(short_name(r["Kernel_Name"]), wg)][r["Counter_Name"]]
            c[0] += 1
            c[1] += float(r["Counter_Value"])
    cur = sqlite3.connect(a.trace).cursor()
    rows = list(cur.execute("select name, grid_x, workgroup_x, count(*), sum(end-start)/1e3 from kernels group by name, grid_x, workgroup_x"))
    fam = defaultdict(lambda: [0, 0.0, 0.0, 0.0])     # launches, us, busy-weighted us, us without counters
    table = []
    for n, gx, wx, cnt, us in rows:
        k = (short_name(n), gx // max(1, wx))
        fm = family(k[0])
        if fm is None:
            continue
        c = acc.get(k)
        need = ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
        if not c or any(x not in c for x in need):
            fam[fm][0] += cnt
            fam[fm][1] += us
            fam[fm][3] += us
            continue
        mf = c["SQ_VALU_MFMA_BUSY_CYCLES"][1] / c["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        gr = c["GRBM_GUI_ACTIVE"][1] / c["GRBM_GUI_ACTIVE"][0]
        sb = c["SQ_BUSY_CYCLES"][1] / c["SQ_BUSY_CYCLES"][0] if "SQ_BUSY_CYCLES" in c else float("nan")
        util = (mf / 1024.0) / (gr / 8.0) if gr > 0 else 0.0
        fam[fm][0] += cnt
        fam[fm][1] += us
        fam[fm][2] += us * util
        table.append((us, "| %s  [%d] | %d | %.1f | %.3g | %.3g | %.3g | %.1f %% |" % (
            k[0], k[1], cnt, us / cnt, mf, sb, gr, 100.0 * util)))
    B = max(1, a.objects_per_launch)
    print("| kernel  [workgroups] | launches (trace, %d object(s)) | avg us | SQ_VALU_MFMA_BUSY_CYCLES | SQ_BUSY_CYCLES | "
          "GRBM_GUI_ACTIVE | MFMA pipe busy |\n|---|---|---|---|---|---|---|" % B)
    for _, line in sorted(table, reverse=True)[:28]:
        print(line)
    print()
    tot_us = tot_busy = 0.0
    for f, (n, us, bus, miss) in sorted(fam.items()):
        cov = us - miss
        print("family %s: %d launches, %.1f ms per object, MFMA pipe busy %.1f %% of the family's time (counters cover %.0f %% of it)"
              % (f, n, us / 1e3 / B, 100.0 * bus / cov if cov > 0 else 0.0, 100.0 * cov / us if us > 0 else 0.0))
        tot_us += us
        tot_busy += bus
    if tot_us > 0:
        print("\nall listed families: MFMA pipe busy %.1f %% of %.1f ms per object" % (100.0 * tot_busy / tot_us, tot_us / 1e3 / B))


if __name__ == "__main__":
    main()
