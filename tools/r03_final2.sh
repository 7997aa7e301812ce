#!/bin/bash
# round 3, after the texture-stage models were added to the library (the hot path's kernels are unchanged, the library digest is not):
# the parity tests of the hot path and of what the new code touched, the counter evidence of THIS build (-> profiles/traffic.json),
# the driver's bench invocation, and the same-box reference line (one object per launch, no query cache)
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 420 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_texgen_gpu.py tests/test_tex_gpu.py tests/test_aekl_gpu.py -m gpu -q --durations=6 -k "not fifty_steps_mini and not full_depth and not upstream_sizes and not sd_dims" 2>&1 | tail -30 > gpurun_out/r03_tests_short.log
tail -3 gpurun_out/r03_tests_short.log
bash tools/r03_profile.sh "$1" 4
cp gpurun_out/traffic.json profiles/traffic.json
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
cut -c1-400 gpurun_out/r03_bench.json; tail -2 gpurun_out/r03_bench.err
R3G_OPTIONS=geo_q_cache=0 timeout 300 python bench.py --steps 8 --warmup 2 --objects-per-launch 1 --no-cpu-baseline > gpurun_out/r03_bench_one_object_per_launch.json 2>> gpurun_out/r03_bench.err
cut -c1-300 gpurun_out/r03_bench_one_object_per_launch.json
