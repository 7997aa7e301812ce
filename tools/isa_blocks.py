#!/usr/bin/env python3
"""Basic blocks of one kernel in a `hipcc -S` listing: instruction count and class mix per block, in layout order.

    python tools/isa_blocks.py gemm.s 'gemm8p_kernelILi0ELb1ELi0E'      (a substring of the mangled name)
"""
import collections
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    s = open(path).read().split("\n")
    starts = [n for n, l in enumerate(s) if re.match(r"^_Z\S*" + re.escape(key) + r"\S*:", l)]
    if not starts:
        sys.exit("no kernel matches " + key)
    blocks, cur = [], ["entry", collections.Counter(), 0, []]
    regs = {}
    for l in s[starts[0] + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), collections.Counter(), 0, []]
            continue
        t = l.strip()
        m = re.match(r"^([a-z_0-9]+)", t)
        if not m or t.startswith(";"):
            continue
        op = m.group(1)
        cls = ("mfma" if "mfma" in op else "valu" if op.startswith("v_") else "ds" if op.startswith("ds_") else
               "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu" if op.startswith("s_") else "other")
        cur[1][cls] += 1
        cur[2] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            cur[3].append(t.split()[-1])
    blocks.append(cur)
    tot = collections.Counter()
    for name, c, n, br in blocks:
        print("%-12s %5d  %s  -> %s" % (name, n, " ".join("%s %d" % kv for kv in sorted(c.items())), ",".join(br)))
        tot.update(c)
    print("total", sum(tot.values()), dict(tot))
    for l in s[starts[0]:]:
        m = re.match(r"^\s*[;.]\s*(\.?(?:vgpr_count|sgpr_count|NumVgprs|NumAgprs|ScratchSize|Occupancy|TotalNumVgprs)\S*)\s*[:=]?\s*(\d+)", l)
        if m:
            regs[m.group(1)] = m.group(2)
        if l.startswith("\t.end_amdhsa_kernel") or len(regs) > 8:
            break
    print(regs)


if __name__ == "__main__":
    main()
