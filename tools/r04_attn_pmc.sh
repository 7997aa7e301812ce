#!/bin/bash
# round 4: what bounds the attention kernels -- one PMC pass with the SQ instruction / busy counters next to the MFMA-busy counter
set -x
R=/root/repo
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > $R/gpurun_out/r04_sq_counters.txt
wc -l $R/gpurun_out/r04_sq_counters.txt
grep -E "VALU|MFMA|TRANS|WAVE_CYCLES|WAIT_INST|ACTIVE_INST|BUSY_CY|INSTS_LDS|INSTS_SALU" $R/gpurun_out/r04_sq_counters.txt | tr '\n' ' '
B=4
run() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "attn" --output-format csv -d $R/gpurun_out/r04_pmc_attn_$name -o p -- python $R/bench.py --steps $B --warmup 0 --objects-per-launch $B --no-cpu-baseline --no-roofline --inference-steps 2 > $R/gpurun_out/r04_pmc_attn_$name.log 2>&1
  tail -2 $R/gpurun_out/r04_pmc_attn_$name.log
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
run b SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run c SQ_WAVE_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32
cd $R
python tools/attn_pmc_table.py gpurun_out/r04_pmc_attn_a gpurun_out/r04_pmc_attn_b gpurun_out/r04_pmc_attn_c > gpurun_out/r04_attention_counters.md 2>&1
cat gpurun_out/r04_attention_counters.md
rm -rf gpurun_out/r04_pmc_attn_a gpurun_out/r04_pmc_attn_b gpurun_out/r04_pmc_attn_c
