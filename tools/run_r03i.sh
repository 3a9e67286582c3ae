mkdir -p gpurun_out/r03i
( for r in 1 2 3; do for v in pre_slab default; do
  if [ $v = default ]; then unset SC_LIB; else export SC_LIB=$PWD/tools/bin/lib_$v.so; fi
  echo -n "$v: "; python bench.py --no-cpu-baseline --steps 3 --decode-tokens 0 2>&1 | grep '^{' | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(round(v['frac'],4)) for k,v in d['roofline_stages'].items()}, d['config'].get('retrieval_crc32'))"
done; done ) > gpurun_out/r03i/ab.log 2>&1
cat gpurun_out/r03i/ab.log
unset SC_LIB
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r03i/pytest.log
