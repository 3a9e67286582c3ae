mkdir -p gpurun_out/r03g
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_edges.py tests/test_gpu_vision.py tests/test_gpu_rope_fused.py -x -q 2>&1 | tail -5
for v in trace trace3 trace trace3; do
  echo "== $v"; SC_LIB=$PWD/tools/bin/lib_$v.so timeout 300 python tools/trace_fat.py vit512.qkv+b vit512.o+res vit512.fc1+gelu vit512.fc2+res llm49k.q llm49k.o+res 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['name'], 'ms', d['ms'], 'loop', d['loop_us'][0], 'epi', d['epilogue_us'], 'tile', d['tile_us'])"
done > gpurun_out/r03g/abl.log 2>&1
cat gpurun_out/r03g/abl.log
