#!/bin/bash
# Developer script (GPU box): SQ counters of the two dK/dV kernels on the encoder shape (each --pmc pass is its own run, kernel trace only)
cd /tmp && export TMPDIR=/tmp ONE=1
R=$GRAFT_REPO_ROOT
i=0
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/dkpmc/p$i -o x -- python3 $R/tools/dev/dkdv4w_one.py 32 > $R/gpurun_out/dkpmc_p$i.log 2>&1 || tail -3 $R/gpurun_out/dkpmc_p$i.log
done
cd $R && python3 tools/dev/pmc_sum.py gpurun_out/dkpmc | grep -A40 "attn_bwd_dkdv"
