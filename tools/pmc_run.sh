cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 60 rocprofv3 --pmc $set --kernel-include-regex "gemm_kernel" --output-format csv -d /root/repo/gpurun_out/pmcg_$i -o p -- python /root/repo/tools/pmc_gemm.py gemm > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmcg_$i 2>&1 | grep "^| gemm" 
done
