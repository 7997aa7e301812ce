#!/bin/bash
# round 4, sixth GPU run: persistent QKV + early wait -- parity tests of the touched kernels, then A/B on the bench
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "not fifty_steps_mini and not full_depth" 2>&1 | tail -6 > gpurun_out/r04_tests6.log
cat gpurun_out/r04_tests6.log
for opt in "" "gemm_persistent_qkv=0" "gemm_early_wait=0" "gemm_persistent_qkv=0,gemm_early_wait=0" ""; do
  R3G_OPTIONS="$opt" timeout 400 python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
fam = r['roofline']['families_ms_per_object']
print('OPT [$opt]  %.4f obj/s  %.1f ms/object  gemm %.1f attn %.1f' % (r['value'], r['ms_per_step'], fam['gemm'], fam['attention']))
"
done
