#!/usr/bin/env python3
"""Copy the tables of a round's evidence run (tools/profile_evidence.sh) from gpurun_out/ into profiles/ with their explanatory headers.
    python tools/profile_headers.py <commit> <library digest prefix> [file prefix, default r05]"""
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
commit = sys.argv[1] if len(sys.argv) > 1 else "?"
digest = sys.argv[2] if len(sys.argv) > 2 else "?"
P = sys.argv[3] if len(sys.argv) > 3 else "r05"
RN = P[1:].lstrip("0")
common = ("Round %s evidence run (`tools/r05_gpu.sh evidence` = `tools/profile_evidence.sh`), commit %s, one MI355X box, "
          "library digest %s..., 4 objects per launch.\n" % (RN, commit, digest))
HDR = {
    P + "_kernel_stats.md": ("round " + RN + ": rocprofv3 --kernel-trace of one launch group of the bench (4 objects, 50 steps, 257^3 grid + marching cubes)",
        "`rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 0 --objects-per-launch 4 --no-cpu-baseline --no-roofline`; the\n"
        "group is the process's first, so it also BUILDS the geo decoder's query-side cache (the 129 fourier_grid / query_proj / ln_1 / c_q launches\n"
        "that later groups do not run) and includes torch's weight-synthesis kernels.  Totals are for the 4 objects together: divide by 4 per object."),
    P + "_kernel_stats_by_grid.md": ("round " + RN + ": the same trace per (kernel, grid) = per problem shape",
        "One row per problem shape of the launch group (4 objects).  DiT: `attn2_kernel [3840]` = the ragged CFG attention of 8 entries,\n"
        "`gemm8p_kernel<1> [236]` = MLP-in + GELU(tanh) on the persistent phased kernel, `gemm8_kernel<8> [472]` = the fp16-residual GEMMs (attention\n"
        "projection / MLP-out / linear2), `gemm8_kernel<5> [1416]` = the fused QKV projections.  Geo decoder: `attn3_kernel [8192]`, `gemm8p_kernel<2> [256]`\n"
        "(c_fc + GELU(erf)), `gemm8_kernel<6> [2048]` (bf16 residual GEMMs)."),
    P + "_pmc_traffic.md": ("round " + RN + ": L2 <-> fabric traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)",
        "`python bench.py --steps 4 --warmup 0 --objects-per-launch 4 --inference-steps 2` under each counter, matched shape by shape with the launch\n"
        "counts of the full 50-step trace (tools/traffic_json.py -> profiles/traffic.json, which records the library digest it belongs to).\n"
        "FETCH_SIZE doubled for the 16-byte-per-lane loads of the GEMM / attention kernels (MI355X_MICROARCH.md, HBM section).  These counters sit\n"
        "between the 8 XCD-private L2s and the fabric: Infinity-Cache hits are counted, and an operand panel that several XCDs need is fetched once per\n"
        "XCD.  The family's ratio to the algorithmic bytes is printed at the end of the table (round 3: 1.58; the DiT's residual stream is fp16\n"
        "since round 4, so its read-modify-write launches move half of round 3's bytes)."),
    P + "_mfma_util.md": ("round " + RN + ": MFMA pipe utilisation per kernel and shape (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE)",
        "One counter pass over a 2-step run of a 4-object launch group, weighted with the durations of the full 50-step trace (tools/mfma_util.py).\n"
        "MFMA pipe busy = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs): a ratio of two counters of one pass, independent of the\n"
        "clock the box happens to run at.  north_star's target is >= 40 % for the path; see the last line of the table."),
}
for name, (title, text) in HDR.items():
    src = os.path.join(ROOT, "gpurun_out", name)
    s = open(src).read()
    s = re.sub(r"^# x\n", "", s)
    open(os.path.join(ROOT, "profiles", name), "w").write("# " + title + "\n\n" + common + text + "\n\n" + s.lstrip())
for name in (P + "_bench.json",):
    if os.path.exists(os.path.join(ROOT, "gpurun_out", name)):
        shutil.copy(os.path.join(ROOT, "gpurun_out", name), os.path.join(ROOT, "profiles", name))
shutil.copy(os.path.join(ROOT, "gpurun_out", "traffic.json"), os.path.join(ROOT, "profiles", "traffic.json"))
print("ok")
