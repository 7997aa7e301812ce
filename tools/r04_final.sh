#!/bin/bash
# round 4 evidence run on one MI355X box, at the commit given as $1: full GPU suite, the counter evidence (which also writes the
# traffic record of THIS build), the driver's bench invocation, and a same-box reference line with this round's two schedule
# changes off (fp32 residual stream, host preparation in front of every group = round 3's behaviour on this round's library)
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -300 > gpurun_out/r04_final_tests.log
tail -4 gpurun_out/r04_final_tests.log
bash tools/r04_profile.sh "$1" 4
cp gpurun_out/traffic.json profiles/traffic.json
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
cut -c1-600 gpurun_out/r04_bench.json; tail -2 gpurun_out/r04_bench.err
R3G_OPTIONS=dit_resid_f16=0 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-prefetch --no-cpu-baseline > gpurun_out/r04_bench_round3_schedule.json 2>> gpurun_out/r04_bench.err
cut -c1-300 gpurun_out/r04_bench_round3_schedule.json
