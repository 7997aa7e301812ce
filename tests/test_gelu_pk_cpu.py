"""The packed-fp16 GELU of csrc/gemm_common.h through its instruction-exact emulation (tools/fit_gelu_pk.py; the kernel is checked
against the same emulation bit for bit on the GPU, tests/test_ops_gpu.py): coefficients in the header = coefficients of the tool,
accuracy against both GELU flavours, exact ends."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fit_gelu_pk as G  # noqa: E402


def test_header_coefficients_are_the_tools():
    src = open(os.path.join(ROOT, "3d-re-gen_amd", "csrc", "gemm_common.h")).read()
    body = src[src.index("gelu_pk_s(float x0, float x1)"):]
    for k in range(7):
        m = re.search(r"constexpr float c%d = \(ERF \? (-?[0-9.e-]+)f : (-?[0-9.e-]+)f\) \* 0.25f;" % k, body)
        assert m, k
        assert float(m.group(1)) == G.COEF[True][k] and float(m.group(2)) == G.COEF[False][k]
        assert float(np.float16(G.COEF[True][k])) == G.COEF[True][k] and float(np.float16(G.COEF[False][k])) == G.COEF[False][k]


def test_accuracy_and_exact_ends():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(400000).astype(np.float32) * 1.5, np.linspace(-8, 8, 100001).astype(np.float32)])
    for erf, S in ((False, G.S_tanh), (True, G.S_erf)):
        s = G.emulate_s(x, erf).astype(np.float64)
        assert np.abs(s - S(x.astype(np.float64))).max() <= 7.5e-4     # the ONE bound: csrc/gemm_common.h, include/r3g.h
        assert (s[x <= -4] == 0).all() and (s[x >= 4] == 1).all() and ((s >= 0) & (s <= 1)).all()
        ref = x.astype(np.float64) * S(x.astype(np.float64))
        got = G.emulate(x, erf).astype(np.float64)
        exact = G.bf16_rne(ref.astype(np.float32)).astype(np.float64)
        n = 400000
        rl = lambda a: np.sqrt(((a[:n] - ref[:n]) ** 2).sum() / (ref[:n] ** 2).sum())
        assert rl(got) <= 1.02 * rl(exact)                       # + 1.2 % behind the bf16 rounding of the output
        assert abs(((got[:n] - ref[:n]) * np.sign(ref[:n])).sum() / np.abs(ref[:n]).sum()) <= 5e-5      # no bias
    sp = G.emulate(np.array([np.inf, -np.inf, 70000.0, -70000.0, 0.0, np.nan], np.float32), False)
    assert sp[0] == np.inf and np.isnan(sp[1]) and sp[2] == G.bf16_rne(np.float32(70000.0)) and sp[3] == 0 and sp[4] == 0 and np.isnan(sp[5])
