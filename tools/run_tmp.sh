cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r03l; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_llm.py tests/test_gpu_dense.py tests/test_gpu_rope_fused.py tests/test_gpu_entrypoint.py -q 2>&1 | tail -6) > $O/pytest.log 2>&1
for r in 1 2; do timeout 300 python tools/bench_decode.py 49152 64 2>&1 | grep "^{"; done > $O/decode.log
timeout 400 rocprofv3 --kernel-trace -d $O/prof_dec -o dec -- python tools/bench_decode.py 49152 64 > $O/prof_dec.log 2>&1
python tools/rocpd_stats.py $(find $O/prof_dec -name "*.db" | head -1) 2>&1 | head -12 > $O/decode_stats.md; rm -rf $O/prof_dec
tail -3 $O/pytest.log; cat $O/decode.log; cut -c1-150 $O/decode_stats.md
