#!/bin/bash
# SQ counter passes over one attention launch shape (default: causal GQA 26112 tokens).  usage: tools/pmc_attn.sh [S] [tag]
S=${1:-26112}; TAG=${2:-a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" \
         "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM"; do
  n=$(echo $C | cut -c1-14 | tr " " _)
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc_${TAG}_$n -o g -- python tools/run_one_attn.py $S > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$n/g_results.db k_attn 2>&1 | tail -8 | cut -c30-140
done
