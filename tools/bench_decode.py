#!/usr/bin/env python3
"""Single-stream decode rate at a given context (Qwen2-7B shape, random-init): tokens/s for several split-KV factors."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import llm as LM

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 49152
cfg = LM.Qwen2ConfigLite(**LM.QWEN2_7B)
lm = LM.Qwen2Model(LM.random_qwen2_state_dict(cfg, seed=0), cfg, max_seq=ctx + 1100, consume=True)
lm.reset_cache()
for l in range(cfg.layers):
    lm.cache[l][:ctx].normal_(0, 0.5)
for ns in [int(x) for x in (sys.argv[2:] or ["64", "128", "256"])]:
    lm.cache_len = ctx
    g = LM.DecodeGraph(lm, max_new_tokens=256, nsplit=ns)
    g.start(1); g.capture(); torch.cuda.synchronize()
    g.run(16); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.run(128); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps(dict(context=ctx, nsplit=ns, tok_per_s=round(128 / dt, 1), ms_per_token=round(dt / 128 * 1e3, 3))))
