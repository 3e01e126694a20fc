#!/bin/bash
# rocprofv3 PMC passes over 3 eager C2 training steps (scripts/dev_prof.py); one counter set per run, --kernel-trace only.
#   scripts/pmc_step.sh <outdir>        (run on the GPU box; summarise with scripts/pmc_summarise.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/$1
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -- python3 $GRAFT_REPO_ROOT/scripts/dev_prof.py 3 > $OUT/set$i.log 2>&1 || { echo "set $i failed"; tail -3 $OUT/set$i.log; }
  echo "set $i done" 
done
