#!/bin/bash
# after a kernel edit late in the round: the parity tests of the touched kernels, the counter evidence of the new build, the bench lines
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_unet_gpu.py tests/test_fp8_gpu.py -m gpu -q -k "not fifty_steps_mini and not full_depth and not sd21" 2>&1 | tail -8 > gpurun_out/r03_tests_short.log
tail -3 gpurun_out/r03_tests_short.log
bash tools/r03_profile.sh "$1" 4
cp gpurun_out/traffic.json profiles/traffic.json
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
cut -c1-400 gpurun_out/r03_bench.json; tail -2 gpurun_out/r03_bench.err
R3G_OPTIONS=geo_q_cache=0 timeout 300 python bench.py --steps 8 --warmup 2 --objects-per-launch 1 --no-cpu-baseline > gpurun_out/r03_bench_one_object_per_launch.json 2>> gpurun_out/r03_bench.err
cut -c1-300 gpurun_out/r03_bench_one_object_per_launch.json
