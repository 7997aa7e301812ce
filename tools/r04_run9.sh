#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_cfg1_golden_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_stage_gpu.py -m gpu -q -x -k "not fifty_steps_mini" 2>&1 | tail -60 > gpurun_out/r04_tests9.log
grep -E "passed|failed|full depth|dedup|flow_sample|literal" gpurun_out/r04_tests9.log
