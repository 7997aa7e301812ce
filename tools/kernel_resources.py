#!/usr/bin/env python3
"""Register / scratch / LDS / occupancy table of every gfx950 kernel in libr3g.so, from hipcc's own resource remarks
(-Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU).

    python tools/kernel_resources.py [--min-vgprs 0] > profiles/rNN_kernel_resources.md
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
import build  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("r3g::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
            for o in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-vgprs", type=int, default=0)
    a = ap.parse_args()
    rows = []
    for src in build.sources():
        if not src.endswith(".hip"):
            continue
        cmd = ["hipcc"] + build.COMMON + build.PER_FILE.get(src, []) + ["-Rpass-analysis=kernel-resource-usage", "-c",
                                                                       os.path.join(build.CSRC, src), "-o", os.devnull]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        for line in err.split("\n"):
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = {"file": src, "name": m.group(1)}
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    names = demangle([r["name"] for r in rows])
    print("| file | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spills | LDS B/workgroup (static) | waves/SIMD |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r, n in zip(rows, names):
        if int(r.get("VGPRs", 0)) < a.min_vgprs:
            continue
        n = re.sub(r"\(.*$", "", n)[:90]
        print("| %s | `%s` | %s | %s | %s | %s | %s | %s | %s |" % (
            r["file"], n, r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"),
            r.get("VGPRs Spill"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))


if __name__ == "__main__":
    main()
