#!/bin/bash
# compile attention.hip to ISA and print VGPR / spill counts per k_attn instantiation
cd "$(dirname "$0")/../streamchat_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -S --cuda-device-only attention.hip -o /tmp/attn2.s 2>/dev/null
python3 - <<'PY'
import re
t=open('/tmp/attn2.s').read()
for m in re.finditer(r'\.name:\s+(\S*k_attnI\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', t):
    n=re.search(r'k_attnILi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELb(\d)',m.group(1)).groups()
    print('DH=%s QB=%s causal=%s CH=%s PRE=%s  sgpr=%s vgpr=%s spill=%s'%(n+m.groups()[1:]))
PY
