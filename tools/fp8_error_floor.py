#!/usr/bin/env python3
"""What OCP e4m3 operands cost a GEMM, whatever the scaling scheme (CPU, torch float8_e4m3fn; no GPU needed).

    python tools/fp8_error_floor.py

rel-L2 error of A.W^T (M 512, K 1024, N 512) against fp64 for: bf16 operands | e4m3 with one scale per row (what gemm8f_kernel
+ the row quantiser do today) | e4m3 with MX block scales (one power-of-two E8M0 scale per 32 elements along K, what
v_mfma_scale_f32_16x16x128_f8f6f4 applies in hardware) | only one side in e4m3.  e4m3 keeps 3 mantissa bits: every
element carries a relative rounding error of ~2^-4 / sqrt(3), and in a sum of K products with random signs the errors add
like the terms do -- the relative error of the dot product stays at that level, independent of K and of the scale
granularity.  Block scales widen the dynamic range; they do not add mantissa bits."""
import torch


def q_e4m3(x, block=None):
    if block is None:
        s = x.abs().amax(-1, keepdim=True) / 448.0
        return (x / s).to(torch.float8_e4m3fn).float() * s
    xb = x.reshape(x.shape[0], -1, block)
    s = torch.exp2(torch.ceil(torch.log2(xb.abs().amax(-1, keepdim=True) / 448.0)))
    return ((xb / s).to(torch.float8_e4m3fn).float() * s).reshape(x.shape)


def main():
    torch.manual_seed(0)
    M, K, N = 512, 1024, 512
    cases = (("gaussian activations (a LayerNorm output) x gaussian weights", lambda: (torch.randn(M, K), torch.randn(N, K) / 32)),
             ("activations with 1 % outliers (x20) x gaussian weights",
              lambda: (torch.randn(M, K) * (1 + 19 * (torch.rand(M, K) < 0.01)), torch.randn(N, K) / 32)),
             ("GELU hidden x gaussian weights", lambda: (torch.nn.functional.gelu(torch.randn(M, K) * 1.5), torch.randn(N, K) / 32)))
    print("| operands | bf16 x bf16 | e4m3 x e4m3, row scales | e4m3 x e4m3, MX block scales (32) | e4m3 (MX) x bf16 | bf16 x e4m3 (MX) |")
    print("|---|---|---|---|---|---|")
    for name, gen in cases:
        a, w = gen()
        a16, w16 = a.bfloat16().float(), w.bfloat16().float()
        ref = a.double() @ w.double().t()

        def err(x, y):
            return float(torch.linalg.norm((x @ y.t()).double() - ref) / torch.linalg.norm(ref))
        print("| %s | %.4f | %.4f | %.4f | %.4f | %.4f |" % (name, err(a16, w16), err(q_e4m3(a), q_e4m3(w)), err(q_e4m3(a, 32), q_e4m3(w, 32)),
                                                        err(q_e4m3(a, 32), w16), err(a16, q_e4m3(w, 32))))


if __name__ == "__main__":
    main()
