cd $GRAFT_REPO_ROOT; O=gpurun_out/r03m; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_dense.py tests/test_gpu_rope_fused.py tests/test_gpu_vision.py tests/test_gpu_llm.py tests/test_gpu_sharded.py -q -k "not processes and not weak and not rccl and not bare" 2>&1 | tail -6) > $O/pytest.log 2>&1
for r in 1 2; do for lib in prev new; do
  if [ $lib = new ]; then unset SC_LIB; else export SC_LIB=$PWD/tools/bin/lib_prev.so; fi
  echo -n "$lib: "; timeout 300 python tools/bench_gemm.py vit512.fc1+gelu vit512.fc2+res vit512.qkv+b vit512.o+res llm49k.q llm49k.down llm.gateup+swiglu 2>&1 | grep "^{" | python -c "
import sys, json
print(' '.join(f\"{json.loads(l)['name']}={json.loads(l)['TFLOPs']}\" for l in sys.stdin))"; done; done > $O/gemm_ab.log 2>&1
for r in 1 2; do for lib in prev new; do
  if [ $lib = new ]; then unset SC_LIB; else export SC_LIB=$PWD/tools/bin/lib_prev.so; fi
  echo -n "$lib: "; timeout 400 python bench.py --no-cpu-baseline --steps 3 --decode-tokens 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['encode_ms_per_step'], {k:v['ms_per_step'] for k,v in d['stages'].items() if 'k_' in k})"; done; done > $O/bench_ab.log 2>&1
unset SC_LIB
tail -3 $O/pytest.log; cat $O/gemm_ab.log $O/bench_ab.log
