cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05d; mkdir -p $O
timeout 900 python tools/bench_overlap.py --tokens 256 --prefill 16384 --reps 4 --dec-cus 64 96 128 > $O/overlap.jsonl 2> $O/overlap.err
timeout 900 python -m pytest tests/test_gpu_llm.py tests/test_gpu_bench_product.py tests/test_abi.py tests/test_gpu_kmeans.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python tools/bench_batched_decode.py > $O/bd.log 2>&1
cat $O/overlap.jsonl; tail -4 $O/pytest.log; grep "^{" $O/bd.log
