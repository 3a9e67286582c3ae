#!/usr/bin/env python3
"""Package power (amdgpu hwmon, the sampler of bench.py) of (a) the idle chip, (b) the single-stream decode at a 49 k context running alone - the
two numbers the session's energy model needs: under a 1400 W cap, two workloads run side by side cannot finish sooner than
(their dynamic energies) / (cap - idle power)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from streamchat_amd import llm as LM

def sample(fn, secs):
    with bench.PowerSampler(0) as ps:
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            n += fn()
        torch.cuda.synchronize()
        dt = time.time() - t0
    return ps.summary(), n, dt

idle, _, _ = sample(lambda: (time.sleep(0.25), 0)[1], 6.0)
ctx = 48994
cfg = LM.Qwen2ConfigLite(**LM.QWEN2_7B)
lm = LM.Qwen2Model(LM.random_qwen2_state_dict(cfg, seed=0), cfg, max_seq=ctx + 4200, consume=True)
lm.reset_cache()
for l in range(cfg.layers):
    lm.cache[l][:ctx].normal_(0, 0.5)
lm.cache_len = ctx
g = LM.DecodeGraph(lm, max_new_tokens=4096)
g.start(1); g.capture(); torch.cuda.synchronize(); g.run(32); torch.cuda.synchronize()
def step():
    g.run(64); torch.cuda.synchronize(); return 64
dec, n, dt = sample(step, 10.0)
print(json.dumps(dict(idle=idle, decode_alone=dec, decode_tok_per_s=round(n / dt, 1))))
