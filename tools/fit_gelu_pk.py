#!/usr/bin/env python3
"""tools/fit_gelu_pk.py -- the packed-fp16 GELU of csrc/gemm_common.h (round 5): fit, fp16 coefficients, and an instruction-exact
emulation of the kernel's sequence in numpy (fp16 round-to-nearest-even per instruction, fused multiply-adds rounded once).

    gelu(x) = x S(x),   S = clamp(0.5 + t P(z) / 4, 0, 1),   t = fp16(x),   z = 2 clamp(t (t / 16), 0, 1) - 1

P = 1/2 + (z - 1) Q(z): the constraint P(1) = 1/2 makes S = 0.5 +- 0.5 exactly at |x| = 4 (and beyond, through the clamps).
Q is fitted in the Chebyshev basis of z on [-1, 1] by iteratively re-weighted least squares (minimax of the error of S), then
converted to powers of z, rounded to fp16, and the constant term is nudged so that fp16 Horner gives P(1) = 1/2 exactly.
Prints the coefficients (paste into gemm_common.h: gelu_pk_s) and the error statistics quoted there.
`emulate(x, erf)` is imported by tests/test_gelu_pk_cpu.py and tests/test_ops_gpu.py (bit-exact comparison with the kernel)."""
import numpy as np

H = np.float16
DEG = 6
# fp16 coefficients of P in powers of z (constant term first), as printed by main() and used by the kernel (x 1/4 there)
COEF = {False: [0.7041015625, -0.3388671875, 0.22216796875, -0.13525390625, 0.08990478515625, -0.07281494140625, 0.0306396484375],
        True: [0.7041015625, -0.33837890625, 0.2225341796875, -0.1378173828125, 0.0916748046875, -0.07086181640625, 0.0289154052734375]}


def S_tanh(x):
    return 1 / (1 + np.exp(-2 * np.sqrt(2 / np.pi) * (x + 0.044715 * x ** 3)))


def S_erf(x):
    from scipy.special import ndtr
    return ndtr(x)


def _fma16(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(H)


def _clamp01(v):
    v = v.astype(np.float64)
    return np.where(np.isnan(v), 0.0, np.clip(v, 0.0, 1.0)).astype(H)


def bf16_rne(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def emulate_s(x, erf):
    """S as the kernel computes it: x float32 array -> fp16 array"""
    c = [H(v * 0.25) for v in COEF[bool(erf)]]                     # exact scalings
    with np.errstate(over="ignore", invalid="ignore"):
        t = np.asarray(x, np.float32).astype(H)                     # v_cvt_pk_f16_f32
        q = (t.astype(np.float64) * 0.0625).astype(H)               # v_pk_mul_f16
        u = _clamp01((t.astype(np.float64) * q.astype(np.float64)).astype(H))      # v_pk_mul_f16 clamp
        z = _fma16(u, np.full_like(u, H(2)), np.full_like(u, H(-1)))
        acc = np.full_like(z, c[6])
        for k in range(5, -1, -1):
            acc = _fma16(acc, z, np.full_like(z, c[k]))
        return _clamp01(_fma16(t, acc, np.full_like(z, H(0.5))))


def emulate(x, erf):
    """the bf16 value the GELU epilogue stores for the fp32 value x (v_fma_mix_f32: fma(x, S, +0) in fp32 -- a product of -0
    comes out as +0 --, then round to bf16)"""
    x = np.asarray(x, np.float32)
    with np.errstate(invalid="ignore"):
        g = (x.astype(np.float64) * emulate_s(x, erf).astype(np.float64) + 0.0).astype(np.float32)    # one fp32 rounding of the exact product
    return bf16_rne(g)


def fit(S, deg, iters=400, c=4.0):
    from numpy.polynomial import chebyshev as C
    n = 8000
    tp = np.maximum(np.sqrt(np.cos(np.pi * (np.arange(n) + 0.5) / n) * 0.5 + 0.5), 1e-6)
    z = 2 * tp * tp - 1
    F = (S(c * tp) - 0.5) / tp
    V = C.chebvander(z, deg - 1) * (z - 1)[:, None]
    rhs = F - 0.5
    wt, best = np.ones(n), None
    for _ in range(iters):
        q, *_ = np.linalg.lstsq(V * (tp * wt)[:, None], rhs * tp * wt, rcond=None)
        err = (V @ q - rhs) * tp
        m = np.abs(err).max()
        if best is None or m < best[1]:
            best = (q.copy(), m)
        wt *= 1 + 0.5 * np.abs(err) / m
        wt /= wt.mean()
    qp = C.cheb2poly(best[0])
    p = np.zeros(deg + 1)
    p[0] = 0.5
    for k in range(len(qp)):
        p[k] -= qp[k]
        p[k + 1] += qp[k]
    p16 = p.astype(H)
    one = np.array([1.0], H)
    for _ in range(4):       # fp16 Horner at z = 1 must give exactly 1/2
        acc = np.full_like(one, p16[-1])
        for k in range(deg - 1, -1, -1):
            acc = _fma16(acc, one, np.full_like(one, p16[k]))
        if acc[0] == H(0.5):
            break
        p16[0] = H(np.float64(p16[0]) + (0.5 - np.float64(acc[0])))
    return p16.astype(np.float64), best[1]


def main():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.standard_normal(2000000).astype(np.float32) * 1.5, np.linspace(-8, 8, 400001).astype(np.float32)])
    for erf, S in ((False, S_tanh), (True, S_erf)):
        p16, e = fit(S, DEG)
        print("erf" if erf else "tanh", "fit error of S %.2e; fp16 coefficients:" % e, ", ".join(repr(float(v)) for v in p16))
        assert np.allclose(p16, COEF[erf], rtol=0, atol=1e-9), "COEF in this file is stale"
        s = emulate_s(xs, erf).astype(np.float64)
        ref_s = S(xs.astype(np.float64))
        ref = xs.astype(np.float64) * ref_s
        got = emulate(xs, erf).astype(np.float64)
        exact = bf16_rne(ref.astype(np.float32)).astype(np.float64)
        n = 2000000
        rl = lambda a: np.sqrt(((a[:n] - ref[:n]) ** 2).sum() / (ref[:n] ** 2).sum())
        print("  |dS| max %.2e rms %.2e; stored bf16 values: rel-L2 %.4e (exactly rounded GELU: %.4e); bias %.1e; S(x <= -4) max %g, "
              "S(x >= 4) min %g" % (np.abs(s - ref_s).max(), np.sqrt(((s - ref_s) ** 2).mean()), rl(got), rl(exact),
                                    ((got[:n] - ref[:n]) * np.sign(ref[:n])).sum() / np.abs(ref[:n]).sum(),
                                    s[xs <= -4].max(), s[xs >= 4].min()))


if __name__ == "__main__":
    main()
