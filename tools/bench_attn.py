#!/usr/bin/env python3
"""Attention kernel at the hot path's shapes: interleaved timing rounds (HIP events), optional ablation masks
(r3g_set_option("attn_ablate", mask): 1 no exp2, 2 no row max, 4 no PV, 8 no QK^T, 16 no LDS-DMA, 32 no wait/barrier,
64 no V^T reads -- timing only).

    python tools/bench_attn.py [--ablate 0,1,3,4,8,12,16,48,64,68,15,63,127] [--rounds 3] [--iters 5]
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

SHAPES = [(2, 16, 4442, 4442, 0), (1, 16, 131072, 3072, 1), (1, 16, 3072, 3072, 0), (1, 24, 1370, 1370, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ablate", default="0")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--shapes", default="0,1")
    ap.add_argument("--gens", default="", help="comma list of attn_generation values to time instead of ablation masks")
    ap.add_argument("--gen", type=int, default=2, help="kernel generation the ablation masks apply to")
    ap.add_argument("--zero", action="store_true", help="zero-filled operands (clock / power sensitivity)")
    a = ap.parse_args()
    # --gens entries: "G" or "GvV" = attn_generation G with attn_variant V (round 6), e.g. 2v0,2v1,2v3,6v0,6v1
    def parse(m):
        if "v" in m:
            g_, v_ = m.split("v")
            return int(g_) * 10 + int(v_) + 1000
        return int(m)
    masks = [parse(m) for m in (a.gens or a.ablate).split(",")]
    ffi.context(0)
    L = ffi.lib()
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for si in [int(i) for i in a.shapes.split(",")]:
        B, H, Lq, Lk, shared = SHAPES[si]
        lqp, lkp = (Lq + 127) // 128 * 128, (Lk + 63) // 64 * 64
        g = torch.Generator(device="cuda").manual_seed(si)
        Q = torch.randn(B, H, lqp, 64, device="cuda", generator=g).to(torch.bfloat16)
        K = torch.randn(1 if shared else B, H, lkp, 64, device="cuda", generator=g).to(torch.bfloat16)
        Vt = torch.randn(1 if shared else B, H, 64, lkp, device="cuda", generator=g).to(torch.bfloat16)
        o = torch.zeros(B, Lq, H * 64, device="cuda", dtype=torch.bfloat16)
        if a.zero:
            Q.zero_(); K.zero_(); Vt.zero_()
        if not a.gens:
            ffi.check(L.r3g_set_option(b"attn_generation", a.gen))

        def run(mask):
            if a.gens:
                if mask >= 1000:
                    ffi.check(L.r3g_set_option(b"attn_generation", (mask - 1000) // 10))
                    ffi.check(L.r3g_set_option(b"attn_variant", (mask - 1000) % 10))
                else:
                    ffi.check(L.r3g_set_option(b"attn_generation", mask))
            else:
                ffi.check(L.r3g_set_option(b"attn_ablate", mask))
            ffi.check(L.r3g_op_attention(Q.data_ptr(), K.data_ptr(), Vt.data_ptr(), o.data_ptr(), B, H, Lq, lqp, Lk, lkp,
                                         shared, 1, s))
        if a.gens:   # bit-comparison of every generation's output with the first one's
            outs = {}
            for m in masks:
                o.zero_()
                run(m)
                torch.cuda.synchronize()
                outs[m] = o.clone()
            for m in masks[1:]:
                d = (outs[m].float() - outs[masks[0]].float()).abs()
                print(json.dumps(dict(op="attn_compare", shape=[B, H, Lq, Lk], gen=m, against=masks[0],
                                      differing=int((outs[m] != outs[masks[0]]).sum()), max_abs=float(d.max()))), flush=True)
        times = {m: [] for m in masks}
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for rnd in range(a.rounds + 1):
            for m in masks:
                run(m)
                ev[0].record()
                for _ in range(a.iters):
                    run(m)
                ev[1].record()
                torch.cuda.synchronize()
                if rnd:
                    times[m].append(ev[0].elapsed_time(ev[1]) / a.iters)
        fl = 4.0 * B * H * Lq * Lk * 64
        for m, ts in times.items():
            med = statistics.median(ts)
            print(json.dumps(dict(op="attn", B=B, H=H, Lq=Lq, Lk=Lk, zero=a.zero, **({"gen": m} if a.gens else {"gen": a.gen, "ablate": m}), us_med=1e3 * med, us_min=1e3 * min(ts),
                                  tflops_med=fl / med / 1e9)), flush=True)
    ffi.check(L.r3g_set_option(b"attn_ablate", 0))
    ffi.check(L.r3g_set_option(b"attn_generation", 7))
    ffi.check(L.r3g_set_option(b"attn_variant", 1))


if __name__ == "__main__":
    main()
