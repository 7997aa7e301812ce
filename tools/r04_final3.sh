#!/bin/bash
# round 4: the FaceReducer edit changed the library digest again -> counter evidence + traffic record + the driver's bench line once more
set -x
cd /root/repo
mkdir -p gpurun_out
bash tools/r04_profile.sh "$1" 4
cp gpurun_out/traffic.json profiles/traffic.json
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err
cut -c1-300 gpurun_out/r04_bench.json; tail -2 gpurun_out/r04_bench.err
