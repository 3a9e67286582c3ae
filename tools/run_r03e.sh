mkdir -p gpurun_out/r03e; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r03e
(timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -60) > $O/pytest.log 2>&1
(timeout 600 python bench.py --no-cpu-baseline 2> $O/bench.err | tail -1) > $O/bench.json
timeout 400 python tools/bench_decode.py 49152 32 64 128 2>&1 | grep "^{" > $O/decode.log
timeout 300 python tools/bench_gemm.py llm49k.q vit512.qkv+b 2>&1 | grep "^{" > $O/gemm.log
tail -25 $O/pytest.log; cut -c1-600 $O/bench.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03e/bench.json'))
print(d['value'], d['ms_per_step'], d['encode_ms_per_step'], d.get('decode_tokens_per_s'))
print({k:(v['achieved'],v['frac']) for k,v in d['roofline_stages'].items()})
print({k:v['ms_per_step'] for k,v in d['stages'].items()})
PY
cat $O/decode.log $O/gemm.log
