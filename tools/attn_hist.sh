#!/bin/bash
# per-basic-block instruction histogram of the main loop of k_attn<128,2,causal> (run tools/attn_regs.sh first)
awk "/^${FN:-_ZN12_GLOBAL__N_16k_attnILi128ELi2ELb1E}/,/^\\.Lfunc_end/" /tmp/attn2.s > /tmp/b.s
python3 - <<'PY'
import re, collections
L=[l.rstrip() for l in open('/tmp/b.s')]
blocks=[]; cur=None
for i,l in enumerate(L):
    m=re.match(r'^(\.LBB\d+_\d+):(.*)',l)
    if m:
        cur=[m.group(1), 'Loop' in m.group(2), collections.Counter(), i]; blocks.append(cur); continue
    if cur is None or not l.startswith('\t') or l.strip().startswith(';') or l.strip().startswith('.'): continue
    cur[2][l.split()[0]]+=1
for name,inloop,c,i in blocks:
    if not inloop: continue
    n=sum(c.values())
    if n<8: continue
    keys=['v_mfma_f32_16x16x32_f16','v_exp_f32_e32','v_mov_b64_e32','v_mov_b32_e32','v_pk_mul_f32','v_pk_fma_f32','v_pk_add_f32','v_max3_f32','v_max_f32_e32','v_cvt_pk_f16_f32','v_cndmask_b32_e32','ds_read_b128','ds_read_b64_tr_b16','buffer_load_dwordx4','s_waitcnt','s_nop','s_barrier','scratch_load_dword']
    print(name, 'line',i,'n=',n, ' '.join('%s=%d'%(k.replace('v_','').replace('_e32','')[:12],c[k]) for k in keys if c[k]), ' other=',n-sum(c[k] for k in keys))
PY
