#!/bin/bash
# round 4, eighth GPU run: the fp16 residual stream of the DiT -- parity at full depth, then A/B/A/B on the bench
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_cfg1_golden_gpu.py -m gpu -q -x 2>&1 | tail -16
for opt in "dit_resid_f16=1" "" "dit_resid_f16=1" ""; do
  R3G_OPTIONS="$opt" timeout 400 python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
fam = r['roofline']['families_ms_per_object']
print('OPT [$opt]  %.4f obj/s  %.1f ms/object  gemm %.1f attn %.1f ln %.1f' % (r['value'], r['ms_per_step'], fam['gemm'], fam['attention'], fam['layernorm']))
"
done
