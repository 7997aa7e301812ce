#!/bin/bash
# LayerNorm: 4 rows per wave from 16 384 rows (a launch group's 30 060 DiT rows) against the round-3 rule (65 536), same box
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "layernorm or dedup or objects" 2>&1 | tail -5
for v in a1 b1 a2 b2; do
  if [ ${v:0:1} = a ]; then export R3G_OPTIONS=ln_rows4_min=65536; else export R3G_OPTIONS=; fi
  timeout 600 python bench.py --gpus 1 --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r04_ln_rows_$v.json 2>gpurun_out/r04_ln_rows_$v.err
  python - <<PY
import json
r = json.loads(open('gpurun_out/r04_ln_rows_$v.json').read().strip().splitlines()[-1])
print('$v', r['value'], r['ms_per_step'], r['roofline']['families_ms_per_object'])
PY
done
