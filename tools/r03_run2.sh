# round 3, second GPU call: new tests (UNet blocks, edge-joined floaters, query cache, batching) with durations, A/B of the
# query cache and of the attention grid rule, then the profile script
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_mesh_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_stage_gpu.py -m gpu -q --durations=25 -k "unet or mesh or floaters or gemm or cache or sharing or attention_options or skipping or list_of_images or fifty or configs1 or soup or cluster" 2>&1 | tail -120 > gpurun_out/r03_tests2.log
tail -40 gpurun_out/r03_tests2.log
for cfg in "geo_q_cache=1" "geo_q_cache=0" "geo_q_cache=1,attn_wide_min=1900" "geo_q_cache=1"; do
  R3G_OPTIONS=$cfg timeout 300 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r03_ab2_${cfg}_$(date +%s).json 2>> gpurun_out/r03_ab2.err
  tail -c 3000 gpurun_out/r03_ab2_${cfg}_*.json | tail -1 | cut -c1-160
done
bash tools/r03_profile.sh "$1" 4
