#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 400 python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r04_bench_prefetch.json 2> gpurun_out/r04_bench_prefetch.err
timeout 400 python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline --no-prefetch > gpurun_out/r04_bench_noprefetch.json 2>> gpurun_out/r04_bench_prefetch.err
timeout 400 python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r04_bench_prefetch2.json 2>> gpurun_out/r04_bench_prefetch.err
timeout 400 python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline --no-prefetch > gpurun_out/r04_bench_noprefetch2.json 2>> gpurun_out/r04_bench_prefetch.err
tail -3 gpurun_out/r04_bench_prefetch.err
python - <<'PY'
import json
for f in ("r04_bench_prefetch", "r04_bench_noprefetch", "r04_bench_prefetch2", "r04_bench_noprefetch2"):
    try:
        r = json.load(open("gpurun_out/%s.json" % f))
        fam = r["roofline"]["families_ms_per_object"]
        print(f, "%.4f obj/s  %.1f ms/object; families sum %.1f; host prep %.1f ms/object" % (r["value"], r["ms_per_step"], sum(fam.values()), r.get("host_prepare_ms_per_object", -1)))
    except Exception as e:
        print(f, "failed", e)
PY
