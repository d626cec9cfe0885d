#!/bin/bash
# PMC passes over the 512-token prefill (4 layers of 7B width)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/pmc
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  FLM_PF_ONLY=1 timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc/p$i -o p$i --output-format csv -- python tools/prefill_bench.py 4 512 > gpurun_out/pmc/p$i.log 2>&1
  tail -2 gpurun_out/pmc/p$i.log
done
find gpurun_out/pmc -name "*counter_collection.csv" | head
