#!/bin/bash
# Round profile of the headline command on the GPU box: kernel-trace stats + the two PMC passes for HBM / fabric traffic.
#   bash tools/run_profile.sh r03     -> gpurun_out/<tag>_prof/{bench_kernel_stats.md, pmc_traffic.json}   (copy into profiles/)
TAG=${1:-rXX}; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG}_prof; mkdir -p $O
ARGS="--no-cpu-baseline --decode-tokens 0 --with-captions 0"
timeout 900 rocprofv3 --kernel-trace -d $O/kt -o bench -- python bench.py --steps 3 --warmup 1 $ARGS > $O/kt.log 2>&1
{ echo "# rocprofv3 --kernel-trace -- python bench.py --steps 3 --warmup 1 $ARGS   ($TAG; 4 passes of the C3 step incl. warm-up; init-time torch kernels included)"; echo;
  python tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1); } > $O/bench_kernel_stats.md 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o p -- python bench.py --steps 2 --warmup 1 $ARGS > $O/pmc_$C.log 2>&1
done
python tools/pmc_traffic.py $(find $O/pmc_FETCH_SIZE -name "*.db" | head -1) $(find $O/pmc_WRITE_SIZE -name "*.db" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
head -30 $O/bench_kernel_stats.md; cat $O/pmc_traffic.log | head -40
