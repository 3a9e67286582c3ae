"""GPU: HIP text encoders vs the HF golden vectors (tiny BertModel) and the fp32 PyTorch restatement at
all-MiniLM-L6 / BERT-large widths.  Tolerance: fp16 storage, fp32 accumulate -> tests/_tol.py (max 4e-3 of max magnitude, rms 3e-3, per-row cosine 0.9999); the
retrieval INDICES derived from the embeddings must equal the fp32 ones exactly."""
import os

import numpy as np
import pytest
import torch

from tests._tol import assert_close_fp16

import oracle
from oracle import torch_ref as R
from streamchat_amd import ops, text as T

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_bert_tiny_vs_hf_golden():
    d = np.load(os.path.join(G, "bert_tiny.npz"))
    sd = {k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("bert.")}
    cfg = T.BertConfigLite(hidden=128, layers=2, heads=4, intermediate=256, vocab=500, max_pos=64)
    enc = T.BertEncoder(sd, cfg)
    ids, mask = torch.from_numpy(d["input_ids"]), torch.from_numpy(d["attention_mask"])
    out = enc(input_ids=ids.cuda(), attention_mask=mask.cuda()).last_hidden_state
    ref = torch.from_numpy(d["last_hidden_state"]).cuda()
    m = mask.bool().cuda()
    assert_close_fp16(out.float()[m], ref[m], what="tiny BERT last_hidden_state vs HF golden")


@pytest.mark.parametrize("name,cfgd,layers", [("minilm", T.MINILM_L6, 6), ("bert-large", T.BERT_LARGE, 3)])
def test_sentence_and_cls_embeddings_vs_torch_fp32(name, cfgd, layers):
    cfgd = dict(cfgd, layers=layers)
    cfg = T.BertConfigLite(**cfgd)
    sd = T.random_bert_state_dict(cfg, seed=3, std=0.05)
    enc = T.BertEncoder(sd, cfg)
    tok = T.HashTokenizer()
    texts = [f"clip {i}: " + " ".join(["kitchen", "table", "dog", "river", "phone"][(i + j) % 5] for j in range(3 + 2 * i)) for i in range(9)]
    b = tok(texts)
    ids, mask = b["input_ids"].cuda(), b["attention_mask"].cuda()
    sdf = {k: v.float() for k, v in sd.items()}
    if name == "minilm":
        emb = T.SentenceEmbedder(enc).embed(ids, mask.sum(1))
        ref = R.sentence_embedding(sdf, ids, mask, heads=cfg.heads, layers=layers)
    else:
        emb = enc.embed_cls(ids, mask.sum(1))
        ref = R.bert_last_hidden(sdf, ids, mask, heads=cfg.heads, layers=layers)[:, 0]
    assert emb.dtype == torch.float32
    assert_close_fp16(emb, ref, what=f"{name} embeddings vs fp32 torch_ref")
    # retrieval built on top: identical top-k indices from the fp16-pipeline embeddings and the fp32 ones
    q = ref[4] + 0.05 * ref[2]
    metric = "l2" if name == "minilm" else "cos"
    idx, _ = ops.sim_topk(q, emb, 3, metric)
    ridx, _ = oracle.topk(q.cpu().numpy(), ref.cpu().numpy(), 3, metric)
    assert idx.cpu().tolist() == list(ridx)
