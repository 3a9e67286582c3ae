#!/bin/bash
# SQ counter passes over one GEMM shape (default: the 49k-token LLM down projection).  usage: tools/pmc_gemm.sh [M N K] [tag]
M=${1:-48994}; N=${2:-3584}; K=${3:-18944}; TAG=${4:-g}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for C in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS"; do
  n=$(echo $C | cut -c1-14 | tr " " _)
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmcg_${TAG}_$n -o g -- python tools/run_one_gemm.py $M $N $K > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/pmcg_${TAG}_$n/g_results.db k_gemm 2>&1 | tail -8 | cut -c30-140
  python - gpurun_out/pmcg_${TAG}_$n/g_results.db <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
for r in db.execute(f"select s.kernel_name, count(*), avg(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id where s.kernel_name like '%k_gemm%' group by s.kernel_name"): print("   duration ns", r[0][:50], r[1], round(r[2]))
PY
done
