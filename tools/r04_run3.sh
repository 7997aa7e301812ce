#!/bin/bash
# round 4, third GPU run: configs[1] at full depth against the fp32 golden; the quintic-sigmoid exact GELU (ops tests, timing)
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cfg1_golden_gpu.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r04_cfg1_golden.log
cat gpurun_out/r04_cfg1_golden.log
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x 2>&1 | tail -5
timeout 300 python tools/bench_gemm.py --variants 11,12 --shapes geo --screen 2 --rounds 3 --iters 5 --no-lt 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l)
    if r['op'] == 'gemm': print('time', r['M'], r['N'], r['K'], r['epi'], r['variant'], '%.1f us %.0f TF/s' % (r['us_med'], r['tflops_med']))
    else: print('screen', r['M'], r['N'], r['K'], r['epi'], '%.2e' % r['rel_l2_vs_fp32'], r['mismatches'])
"
