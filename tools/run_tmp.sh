cd $GRAFT_REPO_ROOT; O=gpurun_out/r03n; mkdir -p $O
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $O/pytest.log 2>&1
( time timeout 900 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json ) 2> $O/bench.time
bash tools/run_profile.sh r03 > $O/profile.log 2>&1
tail -4 $O/pytest.log; cat $O/bench.time; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r03n/bench.json'))
print(d['value'], d['ms_per_step'], d['encode_frames_per_s'], d.get('decode_tokens_per_s'))
print(d['roofline'])
print({k:(v['achieved'],v['frac']) for k,v in d['roofline_stages'].items()})
print(json.dumps(d['cpu_baseline'])[:900])
print(d['power'])
PY
