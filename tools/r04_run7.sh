#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_stage_gpu.py tests/test_mvpaint_gpu.py tests/test_texgen_gpu.py tests/test_tex_gpu.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r04_tests7.log
cat gpurun_out/r04_tests7.log
