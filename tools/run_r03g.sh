cd $GRAFT_REPO_ROOT; O=gpurun_out/r03g; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['encode_ms_per_step'], {k:v['ms_per_step'] for k,v in d['stages'].items() if 'k_' in k}, d.get('decode_tokens_per_s'))"; }
for r in 1 2 3; do
  (cd tools/bin/r02_final && timeout 400 python bench.py --no-cpu-baseline --steps 3 --decode-tokens 128 2>/dev/null | tail -1 | line r02_final)
  (timeout 400 python bench.py --no-cpu-baseline --steps 3 --decode-tokens 128 2>/dev/null | tail -1 | line r03_head)
done > $O/ab_same_box.log 2>&1
cat $O/ab_same_box.log
