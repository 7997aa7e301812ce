#!/usr/bin/env python3
"""Times the dense grid query (r3g_grid_query: ShapeVAE geo decoder) for several query-chunk sizes.

    python tools/bench_geo.py [--points 4194304] [--chunks 16384,32768,65536,131072]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=4 * 1024 * 1024)
    ap.add_argument("--chunks", default="16384,32768,65536,131072")
    a = ap.parse_args()
    from hy3dgen.shapegen.pipelines import builtin_config
    from r3g import model as M, weights as W
    cfg = builtin_config("full")
    sd = W.synthetic_state_dict(cfg, 0, device="cuda")
    lat = torch.randn(cfg["vae"]["num_latents"], cfg["vae"]["embed_dim"], device="cuda")
    for ch in [int(c) for c in a.chunks.split(",")]:
        m = M.ShapeModel(cfg, sd, 0, grid_chunk=ch)
        m.vae_decode(lat)
        out = torch.empty((257, 257, 257), dtype=torch.float32, device="cuda")
        m.grid_query(1.01, 256, out, 0, a.points)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m.grid_query(1.01, 256, out, 0, a.points)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print(json.dumps({"chunk": ch, "points": a.points, "ms": ms, "ms_full_257": ms * 257 ** 3 / a.points,
                          "tflops": 3.366e7 * a.points / ms / 1e9}), flush=True)
        del m


if __name__ == "__main__":
    main()
