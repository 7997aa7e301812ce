#!/bin/bash
# round 4, fourth GPU run: host prefetch A/B on the driver's invocation (shorter), the 8-rank functional run of bench.py
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_multirank_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r04_multirank.log
cat gpurun_out/r04_multirank.log
timeout 400 python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r04_bench_prefetch.json 2> gpurun_out/r04_bench_prefetch.err
timeout 400 python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline --no-prefetch > gpurun_out/r04_bench_noprefetch.json 2>> gpurun_out/r04_bench_prefetch.err
timeout 400 python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r04_bench_prefetch2.json 2>> gpurun_out/r04_bench_prefetch.err
tail -3 gpurun_out/r04_bench_prefetch.err
python - <<'PY'
import json
for f in ("r04_bench_prefetch", "r04_bench_noprefetch", "r04_bench_prefetch2"):
    try:
        r = json.load(open("gpurun_out/%s.json" % f))
        fam = r["roofline"]["families_ms_per_object"]
        print(f, "%.4f obj/s  %.1f ms/object; families sum %.1f  (gemm %.1f attn %.1f ln %.1f)" % (r["value"], r["ms_per_step"], sum(fam.values()), fam["gemm"], fam["attention"], fam["layernorm"]))
    except Exception as e:
        print(f, "failed", e)
PY
