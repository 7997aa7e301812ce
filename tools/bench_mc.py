#!/usr/bin/env python3
"""Micro-benchmark of the marching-cubes kernels on one grid (HIP events on the launch stream).

    python tools/bench_mc.py [--n 257] [--iters 20]
Prints one JSON line: per-phase times, algorithmic bytes (4*n^3 + 12 V + 12 F) and GB/s.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def field(n, seed=5):
    rng = np.random.default_rng(seed)
    lo = torch.from_numpy(rng.standard_normal((n // 8 + 1,) * 3).astype(np.float32))[None, None]
    return torch.nn.functional.interpolate(lo, size=(n, n, n), mode="trilinear", align_corners=True)[0, 0].contiguous()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=257)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from r3g import mc
    g = field(a.n).cuda()
    for _ in range(3):
        v, f = mc.extract_mesh(g)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.iters + 1)]
    ev[0].record()
    for i in range(a.iters):
        v, f = mc.extract_mesh(g)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(a.iters))
    med = ms[len(ms) // 2]
    V, F = v.shape[0], f.shape[0]
    bytes_alg = 4 * a.n ** 3 + 12 * V + 12 * F
    print(json.dumps({"n": a.n, "V": V, "F": F, "ms_median": med, "ms_min": ms[0], "alg_bytes": bytes_alg,
                      "GBps_median": bytes_alg / med / 1e6, "frac_of_8TBps": bytes_alg / med / 1e6 / 8000}))


if __name__ == "__main__":
    main()
