cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05h; mkdir -p $O
timeout 1200 python bench.py --no-cpu-baseline --session-decode-cus 96 112 128 144 > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05h/bench.json").read().strip().splitlines()[-1])
print(d["value"], d.get("decode_tokens_per_s"), d.get("c3_with_decode_frames_per_s")); print(json.dumps(d.get("session"), indent=1)); print(json.dumps(d.get("product"))[:1200])
PY
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
