#!/bin/bash
# Round 4 evidence, ONE script: kernel trace of one launch group of the bench, the two traffic passes
# (FETCH_SIZE, WRITE_SIZE: separate runs), the MFMA-utilisation pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
# GRBM_GUI_ACTIVE), and the tables made from them.  Counter passes carry no trace domain besides the kernel dispatches.
#   usage: tools/r04_profile.sh <commit> [objects per launch]      -> gpurun_out/r04_*.md, gpurun_out/traffic.json
set -x
COMMIT=${1:-unknown}; B=${2:-2}
mkdir -p gpurun_out
R=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_trace -o b -- python $R/bench.py --steps $B --warmup 0 --objects-per-launch $B --no-cpu-baseline --no-roofline > $R/gpurun_out/r04_trace.log 2>&1
INC="gemm|attn|layernorm|ln_dot|mc_"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "$INC" --output-format csv -d $R/gpurun_out/r04_pmc_$c -o p -- python $R/bench.py --steps $B --warmup 0 --objects-per-launch $B --no-cpu-baseline --no-roofline --inference-steps 2 > $R/gpurun_out/r04_pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "$INC" --output-format csv -d $R/gpurun_out/r04_pmc_mfma -o p -- python $R/bench.py --steps $B --warmup 0 --objects-per-launch $B --no-cpu-baseline --no-roofline --inference-steps 2 > $R/gpurun_out/r04_pmc_mfma.log 2>&1
cd $R
DB=$(ls gpurun_out/r04_trace/*/*_results.db gpurun_out/r04_trace/*_results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB "x" > gpurun_out/r04_kernel_stats.md
python tools/rocprof_summary.py $DB "x" --by-grid > gpurun_out/r04_kernel_stats_by_grid.md
python tools/traffic_json.py --fetch gpurun_out/r04_pmc_FETCH_SIZE --write gpurun_out/r04_pmc_WRITE_SIZE --trace $DB --commit "$COMMIT" --objects-per-launch $B --out gpurun_out/traffic.json > gpurun_out/r04_pmc_traffic.md 2>&1
python tools/mfma_util.py --pmc gpurun_out/r04_pmc_mfma --trace $DB --objects-per-launch $B > gpurun_out/r04_mfma_util.md 2>&1
tail -12 gpurun_out/r04_pmc_traffic.md; tail -8 gpurun_out/r04_mfma_util.md
# the raw counter / trace directories are scratch: keep the tables
rm -rf gpurun_out/r04_pmc_FETCH_SIZE gpurun_out/r04_pmc_WRITE_SIZE gpurun_out/r04_pmc_mfma
