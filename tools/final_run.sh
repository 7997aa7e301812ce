set -x
cd /root/repo
timeout 400 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/final_tests.log
timeout 300 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cd /tmp && export TMPDIR=/tmp
R3G_OPTIONS=overlap_mlp=0 timeout 200 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_final -o b -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /root/repo/gpurun_out/prof_final.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-include-regex "gemm_kernel|attn_kernel|layernorm_kernel|mc_classify" --output-format csv -d /root/repo/gpurun_out/pmc_$c -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --inference-steps 2 > /root/repo/gpurun_out/pmc_$c.log 2>&1
  python /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmc_$c > /root/repo/gpurun_out/pmc_$c.md 2>&1
done
cd /root/repo
python tools/rocprof_summary.py gpurun_out/prof_final/b_results.db "x" > gpurun_out/prof_final_kernels.md
python tools/rocprof_summary.py gpurun_out/prof_final/b_results.db "x" --by-grid > gpurun_out/prof_final_grid.md
cat gpurun_out/final_tests.log; cut -c1-300 gpurun_out/bench_final.json; tail -3 gpurun_out/pmc_FETCH_SIZE.md
