#!/bin/bash
# PMC passes on one GEMM shape: tools/pmc_gemm.sh M N K akm bkm   (tiles 128x128 and 256x256) -> gpurun_out/pmc_gemm.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
PASSES=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS"
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES"
 "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
 "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_TA_BUSY_sum"
)
i=0
for p in "${PASSES[@]}"; do
  for t in ${TILES:-128 256}; do
    rocprofv3 --pmc $p --kernel-trace -d $OUT/p${i}_$t -o r -- python3 tools/gemm_one.py "$@" ${t%x*} ${t#*x} 6 > $OUT/log_${i}_$t.txt 2>&1
  done
  i=$((i+1))
done
python3 tools/rocpd_counters.py $(find $OUT -name "*.db") --match gemm_kernel > gpurun_out/pmc_gemm.txt 2>&1
rm -rf $OUT      # (the counter databases are large: gpurun copies at most 64 MiB back)
cat gpurun_out/pmc_gemm.txt
