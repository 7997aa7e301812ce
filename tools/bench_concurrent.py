#!/usr/bin/env python3
"""Objects in flight per GPU: J pipelines (each with its own r3g_ctx and HIP stream) fed from one list of crops by J host
threads -- the reference's `jobs_per_gpu`.  Prints objects/s for J = 1, 2 (, 3).   python tools/bench_concurrent.py [--jobs 1,2]"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-re-gen_amd"))
import torch  # noqa: E402
from bench import synthetic_crop  # noqa: E402
from hy3dgen.shapegen import Hunyuan3DDiTFlowMatchingPipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", default="1,2")
    ap.add_argument("--objects", type=int, default=6)
    a = ap.parse_args()
    crops = [synthetic_crop(j) for j in range(a.objects + 3)]
    pipes = []
    for jobs in [int(x) for x in a.jobs.split(",")]:
        while len(pipes) < jobs:
            pipes.append(Hunyuan3DDiTFlowMatchingPipeline.from_pretrained("synthetic:full:0", device="cuda:0", private_ctx=True))
        streams = [torch.cuda.Stream() for _ in range(jobs)]
        lock = threading.Lock()
        nxt = [0]
        done = []

        def worker(k, todo):
            with torch.cuda.stream(streams[k]):
                while True:
                    with lock:
                        i = nxt[0]
                        nxt[0] += 1
                    if i >= len(todo):
                        break
                    m = pipes[k](image=todo[i], num_inference_steps=50, octree_resolution=256, num_chunks=16000,
                                 generator=torch.Generator().manual_seed(1234567), output_type="raw")[0]   # (not the global generator: threads)
                    streams[k].synchronize()
                    done.append((i, int(m[0].shape[0])))

        def run(todo):
            nxt[0] = 0
            del done[:]
            th = [threading.Thread(target=worker, args=(k, todo)) for k in range(jobs)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
        run(crops[:jobs])                      # warm-up: every pipeline once
        t0 = time.perf_counter()
        run(crops[3:3 + a.objects])
        dt = time.perf_counter() - t0
        print(json.dumps({"jobs_per_gpu": jobs, "objects": a.objects, "seconds": dt, "objects_per_sec": a.objects / dt,
                          "vertices": sorted(done)}), flush=True)


if __name__ == "__main__":
    main()
