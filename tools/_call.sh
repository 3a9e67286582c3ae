cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05k; mkdir -p $O
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shaped.json 2> $O/bench.err; echo "bench wall ${SECONDS}s"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05k/bench_driver_shaped.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value","ms_per_step","steps","decode_tokens_per_s","c3_with_decode_frames_per_s")}); print(d["roofline"]["frac"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
print({k: d["session"].get(k) for k in ("serial_frames_per_s","overlapped_frames_per_s","speedup","identical_to_serial","error")}); print(d["product"].get("product_frames_per_s"), d["product"].get("caption_decode",{}).get("frac"))
PY
timeout 600 python -m pytest tests/test_gpu_session.py tests/test_gpu_llm.py -x -q 2>&1 | tail -3
