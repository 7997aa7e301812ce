#!/usr/bin/env python3
"""Times the GPU mesh cleaners (HIP events around each C-ABI call, host sync included) on marching-cubes meshes,
next to the numpy/scipy restatement on the host cores.

    python tools/bench_mesh.py [--field blob|noise] [--n 257] [--iters 5] [--no-cpu]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--field", default="blob", choices=["blob", "noise"])
    ap.add_argument("--n", type=int, default=257)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from bench_mc import blob, field
    from r3g import mc, meshops
    g = (blob(a.n) if a.field == "blob" else field(a.n)).cuda()
    v, f = mc.extract_mesh(g)
    out = {"field": a.field, "n": a.n, "V": int(v.shape[0]), "F": int(f.shape[0])}

    def timed(fn):
        fn()
        ts = []
        for _ in range(a.iters):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2] * 1e3, r

    ms, (v1, f1) = timed(lambda: meshops.remove_floaters(v, f))
    out["gpu_floaters_ms"] = ms
    ms, (v2, f2) = timed(lambda: meshops.remove_degenerate(v1, f1))
    out["gpu_degenerate_ms"] = ms
    ms, (v3, f3) = timed(lambda: meshops.reduce_faces(v2, f2, 40000))
    out["gpu_reduce_ms"] = ms
    out["after"] = {"V": int(v3.shape[0]), "F": int(f3.shape[0])}
    if not a.no_cpu:
        from oracle import mesh_clean
        hv, hf = v.cpu().numpy(), f.cpu().numpy()
        t = time.perf_counter(); w = mesh_clean.remove_floaters(hv, hf); out["cpu_floaters_ms"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter(); w = mesh_clean.remove_degenerate(*w); out["cpu_degenerate_ms"] = (time.perf_counter() - t) * 1e3
        t = time.perf_counter(); w = mesh_clean.reduce_faces(*w, 40000); out["cpu_reduce_ms"] = (time.perf_counter() - t) * 1e3
        out["equal"] = bool((w[1] == f3.cpu().numpy()).all() and (w[0] == v3.cpu().numpy()).all())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
