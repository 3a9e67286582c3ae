#!/bin/bash
# SQ counter passes over the ViT attention shape (B frames x 577 tokens, 16 heads x 64).  usage: tools/pmc_attn_vit.sh [tag]   (SC_ATTN_RES=0|1 selects the kernel)
TAG=${1:-v}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo $C | cut -c1-14 | tr " " _)
  rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_${TAG}_$n -o g -- python $R/tools/run_one_attn.py 577 vit512 > /dev/null 2>&1
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$n/g_results.db k_attn 2>&1 | tail -8 | cut -c30-140
done
