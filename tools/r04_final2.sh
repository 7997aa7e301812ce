#!/bin/bash
# round 4, evidence of the FINAL library (after the texture GroupNorm batching and the LayerNorm edit changed its digest), at the
# commit given as $1, most important first: counter evidence + traffic record of this build, the driver's bench invocation, the
# N = 2 path at full size with two ranks on the one GPU (gloo), the texture stage's kernel trace, then the full GPU suite
set -x
cd /root/repo
mkdir -p gpurun_out
bash tools/r04_profile.sh "$1" 4
cp gpurun_out/traffic.json profiles/traffic.json
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
cut -c1-600 gpurun_out/r04_bench.json; tail -2 gpurun_out/r04_bench.err
R3G_BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 4 --warmup 4 --no-roofline > gpurun_out/r04_bench_2ranks_shared.json 2> gpurun_out/r04_bench_2ranks_shared.err
cut -c1-400 gpurun_out/r04_bench_2ranks_shared.json; tail -2 gpurun_out/r04_bench_2ranks_shared.err
bash tools/r04_tex_profile.sh
timeout 720 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -300 > gpurun_out/r04_final_tests.log
tail -4 gpurun_out/r04_final_tests.log
