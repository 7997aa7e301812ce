#!/bin/bash
# SQ / TCC / TCP counters of the phased 256x256 GEMM kernel (rocprofv3 --pmc, one counter set per pass).
# usage: tools/pmc_gemm8.sh "<M> <N> <K> <epi>" <out prefix>
cd /tmp && export TMPDIR=/tmp
shape=${1:-"131072 1024 4096 3"}; out=${2:-pmc8}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $set --kernel-include-regex "gemm8_kernel" --output-format csv -d /root/repo/gpurun_out/${out}_$i -o p -- \
      python /root/repo/tools/pmc_gemm.py gemm8 $shape > /dev/null 2>/root/repo/gpurun_out/${out}_$i.err
  python /root/repo/tools/pmc_summary.py /root/repo/gpurun_out/${out}_$i 2>&1 | grep "^family"
done
