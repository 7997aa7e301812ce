# round 3, third GPU call: the whole-UNet forward, the texture-stage tests after the seeding change, the two-rank bench path
# on one device (gloo: RCCL refuses two ranks on one GPU), one default bench line
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_tex_gpu.py tests/test_texgen_gpu.py tests/test_stage_gpu.py -m gpu -q --durations=8 -k "unet or tex or stage_script" 2>&1 | tail -60 > gpurun_out/r03_tests3.log
tail -45 gpurun_out/r03_tests3.log
R3G_BENCH_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 4 --no-roofline > gpurun_out/r03_bench_2ranks_shared.json 2> gpurun_out/r03_bench_2ranks_shared.err
tail -c 1500 gpurun_out/r03_bench_2ranks_shared.json; tail -5 gpurun_out/r03_bench_2ranks_shared.err
timeout 400 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
cut -c1-1500 gpurun_out/r03_bench_default.json; tail -3 gpurun_out/r03_bench_default.err
