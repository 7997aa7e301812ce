#!/bin/bash
# round 3 evidence run on one MI355X box, at the commit given as $1: full GPU suite, the counter evidence (which also writes the
# traffic record of THIS build), the driver's bench invocation, a same-box reference line (one object per launch, no query cache =
# round 2's schedule on this round's kernels)
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -260 > gpurun_out/r03_final_tests.log
tail -4 gpurun_out/r03_final_tests.log
bash tools/r03_profile.sh "$1" 4
cp gpurun_out/traffic.json profiles/traffic.json
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
cut -c1-600 gpurun_out/r03_bench.json; tail -2 gpurun_out/r03_bench.err
R3G_OPTIONS=geo_q_cache=0 timeout 300 python bench.py --steps 8 --warmup 2 --objects-per-launch 1 --no-cpu-baseline > gpurun_out/r03_bench_one_object_per_launch.json 2>> gpurun_out/r03_bench.err
cut -c1-300 gpurun_out/r03_bench_one_object_per_launch.json
