#!/bin/bash
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -120 > gpurun_out/r04_gpu_tests.txt
tail -5 gpurun_out/r04_gpu_tests.txt
