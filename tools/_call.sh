cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05b; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_composed_ref.py tests/test_gpu_kmeans.py tests/test_gpu_vision.py tests/test_gpu_gemv_spec.py tests/test_gpu_llm.py tests/test_gpu_dense.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for w in 4 8 16; do SC_SKINNY_WAVES=$w timeout 300 python tools/bench_gemm_m32.py 26 2>&1 | grep "^{" >> $O/m32.jsonl; done
timeout 300 python tools/bench_gemm_m32.py 26 2>&1 | grep "^{" >> $O/m32.jsonl
SC_LIB=$PWD/tools/bin/lib_dectrace.so timeout 300 python tools/trace_decode_attn.py 49152 64 > $O/dectrace_64.json 2>&1
SC_LIB=$PWD/tools/bin/lib_dectrace.so timeout 300 python tools/trace_decode_attn.py 49152 128 > $O/dectrace_128.json 2>&1
SC_LIB=$PWD/tools/bin/lib_dectrace.so SC_DEC_PD=2 timeout 300 python tools/trace_decode_attn.py 49152 64 > $O/dectrace_64_pd2.json 2>&1
timeout 300 python tools/bench_attn_decode_layout.py 49152 64 128 > $O/layout_default.log 2>&1
SC_LIB=$PWD/tools/bin/lib_decabl1.so timeout 300 python tools/bench_attn_decode_layout.py 49152 64 128 > $O/layout_abl1.log 2>&1
SC_LIB=$PWD/tools/bin/lib_decabl1.so SC_DEC_PD=2 timeout 300 python tools/bench_attn_decode_layout.py 49152 64 128 > $O/layout_abl1_pd2.log 2>&1
timeout 300 python tools/bench_batched_decode.py > $O/bd.log 2>&1
tail -3 $O/pytest.log; cat $O/m32.jsonl; cat $O/dectrace_64.json; grep "^{" $O/layout_*.log $O/bd.log
