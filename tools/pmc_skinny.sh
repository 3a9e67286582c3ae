#!/bin/bash
# TA / TCP counters of one skinny-GEMM shape, register-operand kernel (SC_SKINNY_LDS=0) vs LDS-ring kernel (=1).  usage: tools/pmc_skinny.sh M N K
M=${1:-26}; N=${2:-3584}; K=${3:-18944}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in 0 1; do
  echo "== SC_SKINNY_LDS=$L  (M=$M N=$N K=$K)"
  for C in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $C | cut -c1-12 | tr " " _)
    SC_SKINNY_LDS=$L timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmcs_${L}_$n -o g -- python tools/run_one_skinny.py $M $N $K r > /dev/null 2>&1
    python tools/pmc_summary.py gpurun_out/pmcs_${L}_$n/g_results.db k_gemm_skinny 2>&1 | tail -6 | cut -c17-130
    rm -rf gpurun_out/pmcs_${L}_$n
  done
done
