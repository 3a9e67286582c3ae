cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemv_spec.py tests/test_gpu_llm.py tests/test_gpu_rope_fused.py tests/test_gpu_dense.py -x -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for l in 1 0; do SC_SKINNY_LX=$l timeout 300 python tools/bench_gemm_m32.py 26 2>&1 | grep "^{" | grep "gate_up\|lm_head" | sed "s/^{/{\"lx\": $l, /" >> $O/m32.jsonl; done
for l in 1 0; do SC_SKINNY_LX=$l timeout 300 python tools/bench_gemm_m32.py 8 2>&1 | grep "^{" | grep "gate_up\|lm_head" | sed "s/^{/{\"lx\": $l, /" >> $O/m32.jsonl; done
cat $O/m32.jsonl
timeout 300 python tools/bench_batched_decode.py > $O/bd.log 2>&1; grep "^{" $O/bd.log
SC_SKINNY_LX=0 timeout 300 python tools/bench_batched_decode.py > $O/bd0.log 2>&1; grep "^{" $O/bd0.log
