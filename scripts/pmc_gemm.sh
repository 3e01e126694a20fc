#!/bin/bash
# rocprofv3 PMC passes over one case of scripts/gemm_bench.hip (the program after `--` is the bench binary itself).
#   scripts/pmc_gemm.sh <binary> <outdir>      env: CASE (default "post proj_1"), MODE (default fwd)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ONLY="${CASE:-post proj_1}" MODE="${MODE:-fwd}" ITER=20
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $2/set$i -- $1 > $2.set$i.log 2>&1 || exit 1
done
