#!/bin/bash
# The reference's entry point on a synthetic stream with full-size random-init models (the 7B model also writes the chunk captions), serial vs --overlap:
#   bash tools/bench_entry_overlap.sh [videos] [breakpoints] [max_new_tokens]   -> wall seconds of each run (model build excluded: printed by the script)
V=${1:-1}; B=${2:-4}; T=${3:-256}; D=$(mktemp -d)
for mode in serial overlap; do
  extra="--overlap 0"; [ $mode = overlap ] && extra="--overlap 128"
  python - <<PY
import json, time, torch, sys
sys.path.insert(0, ".")
import inference_streaming_longva_v2 as E
args = E.parse_args(["--video_dir", "none", "--model_name", "none", "--memory_basic_dir", "$D/$mode/mem", "--save_file", "$D/$mode/out.json", "--annotations", "none",
                     "--language", "en", "--conv-mode", "qwen_1_5", "--synthetic", "$V", "--synthetic_breakpoints", "$B", "--chunk_size", "20", "--max_new_tokens", "$T",
                     "--multi_modal_memory", "--batch_captions", "--temperature", "0"] + "$extra".split())
import os; os.makedirs("$D/$mode", exist_ok=True)
real = E.build_models
box = {}
def timed_build(a):
    t0 = time.perf_counter(); r = real(a); torch.cuda.synchronize(); box["build"] = time.perf_counter() - t0; return r
E.build_models = timed_build
t0 = time.perf_counter(); E.run_inference(args); torch.cuda.synchronize(); dt = time.perf_counter() - t0
out = json.load(open("$D/$mode/out.json"))
print(json.dumps(dict(mode="$mode", videos=$V, breakpoints=$B, max_new_tokens=$T, run_s=round(dt - box["build"], 2), build_s=round(box["build"], 1), answers=len(out),
                      answer_chars=[len(r["predict"]) for r in out])))
PY
done
rm -rf $D
