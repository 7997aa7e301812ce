#!/bin/bash
# Counter evidence of a round, ONE script (rounds 3-5): kernel trace of one launch group of the bench, the two traffic passes
# (FETCH_SIZE, WRITE_SIZE: separate runs), the MFMA-utilisation pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
# GRBM_GUI_ACTIVE), and the tables made from them.  Counter passes carry no trace domain besides the kernel dispatches.
#   usage: tools/profile_evidence.sh <commit> [objects per launch] [file prefix, default r05]
#          -> gpurun_out/<prefix>_*.md, gpurun_out/traffic.json     (tools/r05_gpu.sh evidence calls it)
set -x
COMMIT=${1:-unknown}; B=${2:-2}; P=${3:-r05}
mkdir -p gpurun_out
R=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${P}_trace -o b -- python $R/bench.py --steps $B --warmup 0 --objects-per-launch $B --no-cpu-baseline --no-roofline > $R/gpurun_out/${P}_trace.log 2>&1
INC="gemm|attn|layernorm|ln_dot|mc_"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-include-regex "$INC" --output-format csv -d $R/gpurun_out/${P}_pmc_$c -o p -- python $R/bench.py --steps $B --warmup 0 --objects-per-launch $B --no-cpu-baseline --no-roofline --inference-steps 2 > $R/gpurun_out/${P}_pmc_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-include-regex "$INC" --output-format csv -d $R/gpurun_out/${P}_pmc_mfma -o p -- python $R/bench.py --steps $B --warmup 0 --objects-per-launch $B --no-cpu-baseline --no-roofline --inference-steps 2 > $R/gpurun_out/${P}_pmc_mfma.log 2>&1
cd $R
DB=$(ls gpurun_out/${P}_trace/*/*_results.db gpurun_out/${P}_trace/*_results.db 2>/dev/null | head -1)
python tools/rocprof_summary.py $DB "x" > gpurun_out/${P}_kernel_stats.md
python tools/rocprof_summary.py $DB "x" --by-grid > gpurun_out/${P}_kernel_stats_by_grid.md
python tools/traffic_json.py --fetch gpurun_out/${P}_pmc_FETCH_SIZE --write gpurun_out/${P}_pmc_WRITE_SIZE --trace $DB --commit "$COMMIT" --objects-per-launch $B --out gpurun_out/traffic.json > gpurun_out/${P}_pmc_traffic.md 2>&1
python tools/mfma_util.py --pmc gpurun_out/${P}_pmc_mfma --trace $DB --objects-per-launch $B > gpurun_out/${P}_mfma_util.md 2>&1
tail -12 gpurun_out/${P}_pmc_traffic.md; tail -8 gpurun_out/${P}_mfma_util.md
# the raw counter / trace directories are scratch: keep the tables
rm -rf gpurun_out/${P}_pmc_FETCH_SIZE gpurun_out/${P}_pmc_WRITE_SIZE gpurun_out/${P}_pmc_mfma
