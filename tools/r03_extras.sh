# round 3, extras: rasterisation-group A/B on one box, then the not-the-headline lines (fp8 geo decoder; configs[3]-shaped 513^3)
set -x
cd /root/repo
mkdir -p gpurun_out
for cfg in "gemm_raster=-1" "gemm_raster=8" "gemm_raster=2" "gemm_raster=-1" "gemm_raster=8"; do
  R3G_OPTIONS=$cfg timeout 300 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r03_ab3_${cfg}_$(date +%s).json 2>> gpurun_out/r03_ab3.err
  tail -c 4000 gpurun_out/r03_ab3_${cfg}_*.json | tail -1 | cut -c1-150
done
timeout 300 python bench.py --fp8-geo --steps 8 --warmup 4 --no-cpu-baseline > gpurun_out/r03_bench_fp8geo.json 2>> gpurun_out/r03_ab3.err
cut -c1-200 gpurun_out/r03_bench_fp8geo.json
timeout 400 python bench.py --fp8-geo --octree-resolution 512 --steps 4 --warmup 4 --no-cpu-baseline --no-roofline > gpurun_out/r03_bench_cfg4_513.json 2>> gpurun_out/r03_ab3.err
cut -c1-200 gpurun_out/r03_bench_cfg4_513.json
