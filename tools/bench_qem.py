#!/usr/bin/env python3
"""FaceReducer (QEM edge collapse) wall time on smooth object-like surfaces of growing size: marching cubes of the blob field
at n^3, decimated to 40 000 faces.   python tools/bench_qem.py [n ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from bench_mc import blob  # noqa: E402
from r3g import mc, meshops  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [257, 385, 513]:
    v, f = mc.marching_cubes(blob(n).cuda(), 0.0)
    torch.cuda.synchronize()
    meshops.reduce_faces(v, f, 40000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ov, of = meshops.reduce_faces(v, f, 40000)
    torch.cuda.synchronize()
    print(json.dumps({"grid": n, "faces_in": int(f.shape[0]), "faces_out": int(of.shape[0]),
                      "ms": round(1e3 * (time.perf_counter() - t0), 1)}), flush=True)
