mkdir -p gpurun_out/r03h
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r03h/pytest.log
python bench.py > gpurun_out/r03h/bench.json 2> gpurun_out/r03h/bench.err
tail -3 gpurun_out/r03h/pytest.log
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03h/bench.json'))
print(d['value'], d['ms_per_step'], d.get('encode_ms_per_step'), d.get('decode_tokens_per_s'))
print({k: (round(v['achieved'],1), round(v['frac'],4)) for k, v in d.get('roofline_stages', {}).items()})
print(d.get('retrieval_crc32'), d.get('first_token'))
PY
