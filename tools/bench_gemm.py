#!/usr/bin/env python3
"""GEMM kernel variants at the hot path's shapes: bit-equality screen against the two-stage kernel, then interleaved
timing rounds (HIP events on the launch stream; median and min per variant), with torch.matmul (hipBLASLt) beside.

    python tools/bench_gemm.py [--variants 8,9,11] [--rounds 5] [--iters 10] [--screen 6] [--shapes dit,geo,edge]

Variant = value of r3g_set_option("gemm_waves"): 4 / 8 = 128x128 tile, 9 = 256x256 two-stage, 11 = 256x256 phased, 12 = its
persistent form.
All variants accumulate every output element over k in the same order on the same MFMA shape, so their results must be
bit-identical: the screen runs each variant several times per shape (races in the LDS pipeline show up as rare diffs).
"""
import argparse
import ctypes
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
import torch  # noqa: E402
from r3g import ffi  # noqa: E402

SHAPES = {
    # (M, N, K, epilogue): in-situ shapes of one DiT forward under CFG de-duplication (7552 rows) ...
    "dit": [(7552, 3072, 1024, 0), (7552, 4096, 1024, 1), (7552, 1024, 5120, 3), (7552, 1024, 4096, 3),
            (7552, 1024, 1024, 3), (7552, 7168, 1024, 0)],
    # ... and of one 131072-point pass of the geo decoder
    "geo": [(131072, 1024, 1024, 0), (131072, 1024, 1024, 3), (131072, 4096, 1024, 2), (131072, 1024, 4096, 3)],
    # the geo decoder's residual GEMMs with the 16-bit residual stream
    "geo16": [(131072, 1024, 1024, 6), (131072, 1024, 4096, 6), (131072, 1024, 1024, 3), (131072, 1024, 4096, 3)],
    # K sweep at the geo decoder's M: intercept = fixed cost per tile round (prologue + epilogue), slope = cost per k
    # the DiT's under-filled deep-K residual GEMMs (120 tiles of 256x256): split-K candidates
    "split": [(7552, 1024, 5120, 3), (7552, 1024, 4096, 3), (7552, 1024, 2048, 3), (7552, 1024, 1024, 3)],
    "ksweep": [(131072, 1024, k, e) for e in (0, 3) for k in (128, 256, 512, 1024, 2048)],
    # bf16 outputs, K >= 1024 -- four objects' MLP-in and QKV-shaped
    # projections of the DiT, the geo decoder's c_fc, and ragged / single-tile / deeper-K cases for the screen
    "stream": [(30080, 4096, 1024, 1), (30080, 3072, 1024, 0), (131072, 4096, 1024, 2), (131072, 1024, 1024, 0),
               (5000, 512, 1024, 0), (300, 256, 2048, 1), (77, 512, 1152, 2), (1371, 1024, 1024, 1), (7552, 4096, 1024, 1),
               (256, 256, 1024, 0), (4442, 768, 1536, 0)],
    # round 5: the two GELU launches of the path (four objects' MLP-in, the geo decoder's c_fc) and a plain one beside them
    "gelu": [(30080, 4096, 1024, 1), (131072, 4096, 1024, 2), (30080, 3072, 1024, 0)],
    # round 5: the 3 x 3 convolutions of the texture UNets as GEMMs over im2col rows (fp32 outputs), six views and one sample
    "tex6": [(24576, 320, 2880, 4), (24576, 320, 5760, 4), (24576, 640, 5760, 4), (6144, 640, 5760, 4), (6144, 640, 11520, 4),
             (6144, 1280, 11520, 4), (1536, 1280, 11520, 4), (1536, 1280, 23040, 4), (384, 1280, 11520, 4), (384, 1280, 23040, 4),
             (24576, 2560, 320, 0), (6144, 5120, 640, 0), (1536, 10240, 1280, 0), (24576, 320, 1280, 3), (6144, 640, 2560, 3)],
    "tex1": [(4096, 320, 2880, 4), (4096, 320, 5760, 4), (1024, 640, 5760, 4), (1024, 640, 11520, 4), (256, 1280, 11520, 4),
             (256, 1280, 23040, 4), (64, 1280, 11520, 4), (64, 1280, 23040, 4), (4096, 2560, 320, 0), (1024, 5120, 640, 0)],
    # ragged edges for the screen
    "edge": [(300, 256, 128, 0), (77, 512, 256, 3), (1371, 1024, 1024, 1), (515, 768, 1024, 3), (4442, 1024, 1536, 4),
             (256, 256, 128, 2), (256, 384, 256, 0), (1000, 448, 384, 1)],
}


def make(M, N, K, epi, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    gate = torch.randn(N, device="cuda", generator=g)
    c0 = torch.randn(M, N, device="cuda", generator=g) if epi in (3, 6) else None
    if epi == 6:
        c0 = c0.to(torch.bfloat16)
    return a, w, bias, gate, c0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="8,9,11")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--screen", type=int, default=6)
    ap.add_argument("--shapes", default="edge,dit,geo")
    ap.add_argument("--no-lt", action="store_true")
    ap.add_argument("--gelu-pk", default="1", help="comma list of r3g_set_option(\"gelu_pk\") values to time side by side (1: packed "
                                                  "fp16 GELU epilogue, round 5 | 0: the fp32 forms); variants are then keyed 'v/pk'")
    ap.add_argument("--splitk", action="store_true", help="time r3g_op_gemm_splitk (the texture models' convolution launch) against "
                                                          "the ordinary launch and hipBLASLt on the fp32-output shapes of the groups")
    a_ = ap.parse_args()
    pks = [int(v) for v in a_.gelu_pk.split(",")]
    variants = [int(v) * 10 + pk for v in a_.variants.split(",") for pk in pks]     # variant * 10 + gelu_pk
    ffi.context(0)
    L = ffi.lib()
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(v, a, w, bias, gate, c, M, N, K, epi):
        ffi.check(L.r3g_set_option(b"gemm_waves", v // 10))
        ffi.check(L.r3g_set_option(b"gelu_pk", v % 10))
        ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N,
                                gate.data_ptr() if epi in (3, 6) else None, M, N, K, epi, 1, s))

    if a_.splitk:
        ws = torch.empty(512 * 128 * 128, device="cuda", dtype=torch.float32)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for group in a_.shapes.split(","):
            for (M, N, K, epi) in SHAPES[group]:
                if epi not in (3, 4):
                    continue
                a, w, bias, gate, c0 = make(M, N, K, epi, M + N + K + epi)
                c = c0.clone() if epi == 3 else torch.empty(M, N, device="cuda")
                got = ctypes.c_int(0)
                fns = {
                    "plain": lambda: ffi.check(L.r3g_op_gemm(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N,
                                                              gate.data_ptr() if epi == 3 else None, M, N, K, epi, 1, s)),
                    "splitk": lambda: ffi.check(L.r3g_op_gemm_splitk(a.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), c.data_ptr(), N,
                                                                      gate.data_ptr() if epi == 3 else None, M, N, K, epi, ws.data_ptr(),
                                                                      ws.numel(), ctypes.byref(got), s)),
                    "lt": lambda: torch.matmul(a, w.t()),
                }
                times = {k: [] for k in fns}
                for rnd in range(a_.rounds + 1):
                    for k, fn in fns.items():
                        fn()
                        ev[0].record()
                        for _ in range(a_.iters):
                            fn()
                        ev[1].record()
                        torch.cuda.synchronize()
                        if rnd > 0:
                            times[k].append(ev[0].elapsed_time(ev[1]) / a_.iters)
                fl = 2.0 * M * N * K
                print(json.dumps(dict(op="splitk", M=M, N=N, K=K, epi=epi, slices=got.value,
                                      us={k: round(1e3 * statistics.median(v), 1) for k, v in times.items()},
                                      tflops={k: round(fl / statistics.median(v) / 1e9) for k, v in times.items()})), flush=True)
                del a, w, c
        return
    for group in a_.shapes.split(","):
        for (M, N, K, epi) in SHAPES[group]:
            a, w, bias, gate, c0 = make(M, N, K, epi, M + N + K + epi)
            dt = torch.float32 if epi in (3, 4) else torch.bfloat16

            def fresh():
                return c0.clone() if epi in (3, 6) else torch.full((M, N), float("nan"), device="cuda", dtype=dt)
            refs = {}
            for pk in pks:
                refs[pk] = fresh()
                run(80 + pk, a, w, bias, gate, refs[pk], M, N, K, epi)
            ref = refs[pks[0]]
            lin = a.float() @ w.float().t() + bias
            if epi == 1:
                want = torch.nn.functional.gelu(lin, approximate="tanh")
            elif epi == 2:
                want = torch.nn.functional.gelu(lin)
            elif epi in (3, 6):
                want = c0.float() + gate * lin
            else:
                want = lin
            rel = float(torch.linalg.norm(ref.float() - want) / torch.linalg.norm(want))
            del lin, want
            bad = {}
            for v in variants:
                for it in range(a_.screen):
                    c = fresh()
                    run(v, a, w, bias, gate, c, M, N, K, epi)
                    if not torch.equal(c, refs[v % 10]):
                        d = (c.float() - refs[v % 10].float()).abs()
                        bad.setdefault(v, []).append((it, int((d > 0).sum()), float(d.max())))
            rec = dict(op="screen", M=M, N=N, K=K, epi=epi, rel_l2_vs_fp32=rel, mismatches={str(k): v for k, v in bad.items()})
            print(json.dumps(rec), flush=True)
            if group == "edge" or (group == "stream" and M * N < 4000000):
                continue
            # timing: interleaved rounds
            c = fresh()
            times = {v: [] for v in variants}
            if not a_.no_lt:
                times["lt"] = []
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            for rnd in range(a_.rounds + 1):
                for v in list(times):
                    if v == "lt":
                        fn = lambda: torch.matmul(a, w.t())   # noqa: E731
                    else:
                        fn = lambda v=v: run(v, a, w, bias, gate, c, M, N, K, epi)   # noqa: E731
                    fn()
                    ev[0].record()
                    for _ in range(a_.iters):
                        fn()
                    ev[1].record()
                    torch.cuda.synchronize()
                    if rnd > 0:
                        times[v].append(ev[0].elapsed_time(ev[1]) / a_.iters)
            fl = 2.0 * M * N * K
            for v, ts in times.items():
                med, mn = statistics.median(ts), min(ts)
                print(json.dumps(dict(op="gemm", M=M, N=N, K=K, epi=epi, variant=v if v == "lt" else "%d/pk%d" % (v // 10, v % 10), us_med=1e3 * med, us_min=1e3 * mn,
                                      tflops_med=fl / med / 1e9, tflops_best=fl / mn / 1e9)), flush=True)
            del a, w, c, ref, refs
            torch.cuda.empty_cache()
    ffi.check(L.r3g_set_option(b"gemm_waves", 0))
    ffi.check(L.r3g_set_option(b"gelu_pk", 1))


if __name__ == "__main__":
    main()
