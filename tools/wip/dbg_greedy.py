import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from streamchat_amd import llm as LM
G = "./tests/golden"
d = np.load(os.path.join(G, "qwen2_tiny.npz"))
sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("lm.")}
cfg = LM.Qwen2ConfigLite(hidden=256, layers=2, heads=4, kv_heads=2, intermediate=512, vocab=512, rope_theta=1e6)
lm = LM.Qwen2Model(sd, cfg, max_seq=64)
emb = torch.from_numpy(d["inputs_embeds"]).cuda().half()
logits = lm.forward(emb)
toks = []
table = sd["model.embed_tokens.weight"].cuda().half()
for i in range(8):
    top = torch.topk(logits.flatten(), 3)
    toks.append(int(top.indices[0]))
    print(i, top.indices.tolist(), [round(float(v), 4) for v in top.values], "golden", int(d["greedy"][i]))
    logits = lm.forward(table[toks[-1]][None])
print(toks, d["greedy"].tolist())
