#!/usr/bin/env python3
"""Times the drop-in stage script on N synthetic crops (full dims, 50 steps, 257^3 grid): per-object wall time of the
whole stage loop (preprocess, shapegen, marching cubes, the three cleaners, GLB export), model set-up excluded."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
import yaml  # noqa: E402
from bench import synthetic_crop  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    with tempfile.TemporaryDirectory() as tmp:
        inp, out = os.path.join(tmp, "in"), os.path.join(tmp, "out")
        os.makedirs(inp)
        for i in range(n):
            synthetic_crop(i).save(os.path.join(inp, "obj__(%d, %d).png" % (i, i)))
        cfg = {"mini": False, "num_inf_steps_hy": 50, "octree_resolution_hy": 256, "num_chunks_hy": 16000, "seed": 1234567,
               "remesh": False, "input_folder_hy": inp, "output_folder_hy": out, "use_banana": False,
               "prepped_for_hunyuan": os.path.join(tmp, "unused"), "jobs_per_gpu": 1, "use_all_available_cuda": False,
               "r3g_weights": "synthetic:{model}"}
        cp = os.path.join(tmp, "config.yaml")
        yaml.safe_dump(cfg, open(cp, "w"))
        t = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "3d-re-gen_amd", "stage", "run.py"), "--config", cp],
                           capture_output=True, text=True)
        wall = time.time() - t
        lines = [l for l in r.stdout.splitlines() if "mesh has" in l or "seconds" in l.lower() or "took" in l.lower()]
        glbs = sum(len(f) for _, _, f in os.walk(out))
        print(json.dumps({"objects": n, "glbs": glbs, "returncode": r.returncode, "wall_s_incl_setup": round(wall, 2),
                          "log": lines[-8:]}))
        if r.returncode:
            print(r.stdout[-2000:], r.stderr[-2000:])


if __name__ == "__main__":
    main()
