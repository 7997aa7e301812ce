cd /tmp && export TMPDIR=/tmp
i=10
for set in "TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 45 rocprofv3 --pmc $set --kernel-include-regex "gemm_kernel" --output-format csv -d /root/repo/gpurun_out/pmcg_$i -o p -- python /root/repo/tools/pmc_gemm.py gemm > /dev/null 2>&1
  python /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/pmcg_$i 2>&1 | grep "^| gemm"
done
