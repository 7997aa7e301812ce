#!/bin/bash
# round 4, second GPU run: polynomial GELU epilogues -- op tests, screen, timing (phased / persistent / stream kernels, hipBLASLt)
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r04_ops_tests.log
tail -15 gpurun_out/r04_ops_tests.log
timeout 420 python tools/bench_gemm.py --variants 11,12,14 --shapes stream --screen 3 --rounds 3 --iters 5 > gpurun_out/r04_gemm_stream2.jsonl 2> gpurun_out/r04_gemm_stream2.err
tail -5 gpurun_out/r04_gemm_stream2.err
python - <<'PY'
import json
for l in open("gpurun_out/r04_gemm_stream2.jsonl"):
    r = json.loads(l)
    if r["op"] == "screen":
        print("screen", r["M"], r["N"], r["K"], r["epi"], "rel %.2e" % r["rel_l2_vs_fp32"], r["mismatches"])
    else:
        print("time  ", r["M"], r["N"], r["K"], r["epi"], r["variant"], "%.1f us  %.0f TF/s (best %.0f)" % (r["us_med"], r["tflops_med"], r["tflops_best"]))
PY
