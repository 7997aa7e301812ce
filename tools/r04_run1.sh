#!/bin/bash
# round 4, first GPU run: the 4-wave stream GEMM (gemm4.hip) -- bit-equality screen against the 128x128 kernel and timing
# beside the phased / persistent kernels and hipBLASLt at its shapes
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 420 python tools/bench_gemm.py --variants 11,12,14 --shapes stream --screen 4 --rounds 3 --iters 5 > gpurun_out/r04_gemm_stream.jsonl 2> gpurun_out/r04_gemm_stream.err
tail -5 gpurun_out/r04_gemm_stream.err
python - <<'PY'
import json
for l in open("gpurun_out/r04_gemm_stream.jsonl"):
    r = json.loads(l)
    if r["op"] == "screen":
        print("screen", r["M"], r["N"], r["K"], r["epi"], "rel %.2e" % r["rel_l2_vs_fp32"], r["mismatches"])
    else:
        print("time  ", r["M"], r["N"], r["K"], r["epi"], r["variant"], "%.1f us  %.0f TF/s (best %.0f)" % (r["us_med"], r["tflops_med"], r["tflops_best"]))
PY
