#!/bin/bash
# interleaved A/B of an environment switch on the full bench (same box): ./tools/ab_bench.sh VAR valA valB
for r in 1 2; do for v in $2 $3; do echo -n "$1=$v: "; env $1=$v python bench.py --no-cpu-baseline --steps 2 --decode-tokens 0 2>&1 | grep -v amdgpu | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], {k:v['ms_per_step'] for k,v in d['stages'].items()})"; done; done
