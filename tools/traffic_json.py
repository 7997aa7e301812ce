#!/usr/bin/env python3
"""HBM-side traffic per GEMM launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs), weighted with
the launch counts of a full 50-step object from a --kernel-trace run.  Writes profiles/traffic.json (read by bench.py
for `roofline.traffic`) and prints the markdown table for profiles/.

    traffic_json.py --fetch <pmc dir> --write <pmc dir> --trace <rocpd .db> --commit <sha> [--out profiles/traffic.json]

Key = (kernel name with template arguments, workgroups): the grid identifies the problem shape, so the short PMC run (few
denoising steps) and the full run are matched shape by shape.  Units: KB as rocprofv3 reports them.  Correction
(MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts a 128-byte request as 64 B for the wide (16 B per lane) loads of the
LDS-DMA kernels -> doubled for them; WRITE_SIZE as reported."""
import argparse
import csv
import glob
import json
import os
import re
import sqlite3
import time
from collections import defaultdict

WIDE = re.compile(r"gemm|attn")     # kernels whose loads are 16 bytes per lane


def short_name(n):
    m = re.search(r"((?:\w+::)*)(\w+)(<[^()]*>)?\s*\(", n)
    return (m.group(2) + (m.group(3) or "")) if m else n.split("(")[0][-60:]


def pmc(dirname, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            k = (short_name(r["Kernel_Name"]), wg)
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--trace", required=True)
    ap.add_argument("--commit", default="")
    ap.add_argument("--objects-per-launch", type=int, default=1, help="objects in the traced launch group")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"))
    a = ap.parse_args()
    fetch, write = pmc(a.fetch, "FETCH_SIZE"), pmc(a.write, "WRITE_SIZE")
    cur = sqlite3.connect(a.trace).cursor()
    rows = list(cur.execute("select name, grid_x, workgroup_x, count(*), avg(end-start)/1e3 from kernels group by name, grid_x, workgroup_x"))
    fam = defaultdict(lambda: [0, 0.0, 0])   # launches, bytes, launches without counters
    print("| kernel  [workgroups] | launches / launch group of %d object(s) | avg us |" % a.objects_per_launch + " FETCH_SIZE avg KB | WRITE_SIZE avg KB | bytes / launch (corrected) |\n|---|---|---|---|---|---|")
    table = []
    for n, gx, wx, c, avg in rows:
        k = (short_name(n), gx // max(1, wx))
        family = "gemm" if "gemm" in k[0] else "attention" if "attn" in k[0] else "layernorm" if "layernorm" in k[0] or "ln_dot" in k[0] else \
                 "mc_classify" if "mc_classify" in k[0] else None
        if family is None:
            continue
        if k not in fetch or k not in write:
            fam[family][2] += c
            continue
        b = ((2.0 if WIDE.search(k[0]) else 1.0) * fetch[k] + write[k]) * 1024.0
        fam[family][0] += c
        fam[family][1] += c * b
        table.append((c * b, "| %s  [%d] | %d | %.1f | %.0f | %.0f | %.1f MB |" % (k[0], k[1], c, avg, fetch[k], write[k], b / 1e6)))
    for _, line in sorted(table, reverse=True)[:24]:
        print(line)
    out = {}
    for f, (n, b, miss) in fam.items():
        if n:
            out[f] = {"bytes_per_launch": b / n, "launches": n, "objects_per_launch": a.objects_per_launch, "launches_without_counters": miss,
                      "measured": time.strftime("%Y-%m-%d"), "commit": a.commit,
                      "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 for 16-byte-per-lane loads, "
                                "weighted by the launch counts of one 50-step object (tools/traffic_json.py)"}
            print("\nfamily %s: %d launches (%d without counters), %.1f MB per launch, %.3f TB per object" % (
                f, n, miss, b / n / 1e6, b / 1e12 / max(1, a.objects_per_launch)))
    # the build the counters were measured on: bench.py marks the record stale when the library it loads has another digest
    try:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-re-gen_amd", "libr3g.digest")) as f:
            out["library_digest"] = f.read().strip()
    except OSError:
        out["library_digest"] = None
    with open(a.out, "w") as fo:
        json.dump(out, fo, indent=1)


if __name__ == "__main__":
    main()
