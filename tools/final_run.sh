# round-end evidence run on the MI355X box: full GPU suite, bench, kernel trace, two PMC passes, traffic.json
set -x
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/final_tests.log
timeout 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
# not the headline: the fp8 mode of the geo decoder, and a configs[3]-shaped run (513^3 grid, fp8 geo decoder)
timeout 300 python bench.py --fp8-geo --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_fp8geo.json 2> gpurun_out/bench_fp8geo.err
[ -n "$R3G_FINAL_SHORT" ] || timeout 400 python bench.py --fp8-geo --octree-resolution 512 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_final -o b -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > /root/repo/gpurun_out/prof_final.log 2>&1
# R3G_FINAL_SHORT=1: no 513^3 run and no PMC passes (profiles/traffic.json keeps the commit it was measured at)
for c in $([ -n "$R3G_FINAL_SHORT" ] || echo FETCH_SIZE WRITE_SIZE); do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "gemm|attn|layernorm|ln_dot|mc_classify" --output-format csv -d /root/repo/gpurun_out/pmc_$c -o p -- python /root/repo/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --inference-steps 2 > /root/repo/gpurun_out/pmc_$c.log 2>&1
done
cd /root/repo
DB=$(ls gpurun_out/prof_final/*/*_results.db gpurun_out/prof_final/*_results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB "x" > gpurun_out/prof_final_kernels.md
python tools/rocprof_summary.py $DB "x" --by-grid > gpurun_out/prof_final_grid.md
[ -n "$R3G_FINAL_SHORT" ] || python tools/traffic_json.py --fetch gpurun_out/pmc_FETCH_SIZE --write gpurun_out/pmc_WRITE_SIZE --trace $DB --commit "$1" --out gpurun_out/traffic.json > gpurun_out/traffic.md 2>&1
tail -5 gpurun_out/final_tests.log; cut -c1-400 gpurun_out/bench_final.json; tail -8 gpurun_out/traffic.md
