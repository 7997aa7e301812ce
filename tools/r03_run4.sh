# round 3, fourth GPU call: the texture pipeline on the chart-based unwrap, the stage GLBs, smoke, the two-rank path after the teardown change
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_texgen_gpu.py tests/test_tex_gpu.py tests/test_stage_gpu.py -m gpu -q --durations=5 -k "not config1_mini and not 512" 2>&1 | tail -30 > gpurun_out/r03_tests4.log
tail -22 gpurun_out/r03_tests4.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -3
R3G_BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 4 --warmup 4 --no-roofline > gpurun_out/r03_bench_2ranks_shared.json 2> gpurun_out/r03_bench_2ranks_shared.err
cut -c1-400 gpurun_out/r03_bench_2ranks_shared.json; tail -2 gpurun_out/r03_bench_2ranks_shared.err
python tools/time_stage.py 2>&1 | tail -6
