#!/usr/bin/env python3
"""Where the stage's cleaner time goes on the bench's objects: wall seconds of each cleaner on the 257^3 mesh of synthetic
crop 0 (noise-like: 1.2 M faces) against the kernel time of the same calls (run under rocprofv3 --kernel-trace): the
difference is launch / read-back latency that a second stream could hide, the kernel time is not.
    python tools/r04_cleaner_probe.py [repeats]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))

import torch  # noqa: E402

from bench import synthetic_crop  # noqa: E402
from hy3dgen.shapegen import (DegenerateFaceRemover, FaceReducer, FloaterRemover,  # noqa: E402
                              Hunyuan3DDiTFlowMatchingPipeline)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
pipe = Hunyuan3DDiTFlowMatchingPipeline.from_pretrained("synthetic:full:0", device="cuda:0")
mesh = pipe(image=synthetic_crop(0), num_inference_steps=50, octree_resolution=256, num_chunks=16000,
            generator=torch.manual_seed(1234567), output_type="trimesh")[0]
torch.cuda.synchronize()
out = {"input": {"V": int(mesh.n_vertices), "F": int(mesh.n_faces)}, "repeats": reps, "seconds": {}}
for name, c in (("FloaterRemover", FloaterRemover()), ("DegenerateFaceRemover", DegenerateFaceRemover()), ("FaceReducer", FaceReducer())):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m2 = c(mesh)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out["seconds"][name] = [round(t, 4) for t in ts]
    out.setdefault("faces_after", {})[name] = int(m2.n_faces)
    mesh = m2
print(json.dumps(out))
