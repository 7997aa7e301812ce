# round 3, first GPU call: the full GPU suite, then objects-per-launch / tile-rule A/B on one box
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -150 > gpurun_out/r03_tests1.log
tail -5 gpurun_out/r03_tests1.log
for cfg in "1 gemm_auto_rule=2" "2 gemm_auto_rule=2" "2 gemm_auto_rule=1" "4 gemm_auto_rule=2" "1 gemm_auto_rule=2" "2 gemm_auto_rule=2"; do
  set -- $cfg
  R3G_OPTIONS=$2 timeout 300 python bench.py --steps 8 --warmup 4 --objects-per-launch $1 --no-cpu-baseline > gpurun_out/r03_ab_$1_$2_$(date +%s).json 2> gpurun_out/r03_ab.err
  tail -c 1500 gpurun_out/r03_ab_$1_$2_*.json | tail -1 | cut -c1-300
done
