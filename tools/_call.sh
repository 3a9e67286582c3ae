cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemv_spec.py tests/test_gpu_dense.py tests/test_gpu_llm.py tests/test_gpu_text.py tests/test_gpu_session.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for l in 1 0; do SC_SKINNY_LDS=$l timeout 300 python tools/bench_gemm_m32.py 26 2>&1 | grep "^{" | sed "s/^{/{\"lds\": $l, /" >> $O/m32.jsonl; done
SC_SKINNY_LDS=1 timeout 300 python tools/bench_gemm_m32.py 8 2>&1 | grep "^{" | sed "s/^{/{\"lds\": 1, /" >> $O/m32.jsonl
SC_SKINNY_LDS=0 timeout 300 python tools/bench_gemm_m32.py 8 2>&1 | grep "^{" | sed "s/^{/{\"lds\": 0, /" >> $O/m32.jsonl
cat $O/m32.jsonl
timeout 300 python tools/bench_batched_decode.py > $O/bd.log 2>&1; grep "^{" $O/bd.log
timeout 300 python tools/bench_decode.py 49152 64 > $O/dec.log 2>&1; grep "^{" $O/dec.log
SC_KV_ROW_PAD=0 timeout 300 python tools/bench_decode.py 49152 64 > $O/dec_nopad.log 2>&1; grep "^{" $O/dec_nopad.log
