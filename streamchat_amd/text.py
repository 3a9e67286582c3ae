"""Text encoders of the retrieval path on hand-written gfx950 kernels.

* `BertEncoder` — HF `BertModel` arithmetic (post-LN encoder, GELU-erf FFN).  The reference loads
  mxbai-colbert-large-v1 as a plain `AutoModel` (inference_streaming_longva_v2.py:703-705) and CLS-pools it for the
  memory-tree search (utiles.py:687-708,725-729: "ColBERT" is NOT late interaction — SURVEY.md §0 item 10).
* `SentenceEmbedder` — all-MiniLM-L6-v2 as sentence-transformers runs it (BERT 6x384 -> masked mean pool -> L2
  normalise), the `HuggingFaceEmbeddings` behind the dialogue memory (memory_bank/memory_retrieval/local_doc_qa.py:193).
Weights use the transformers state-dict names; texts are batched (the reference encodes one caption at a time with a
host sync each, utiles.py:696,721-732)."""
import types

import torch

from . import ops


class BertConfigLite:
    def __init__(self, hidden=1024, layers=24, heads=16, intermediate=4096, vocab=30522, max_pos=512, eps=1e-12):
        self.hidden, self.layers, self.heads, self.intermediate = hidden, layers, heads, intermediate
        self.vocab, self.max_pos, self.eps = vocab, max_pos, eps


BERT_LARGE = dict(hidden=1024, layers=24, heads=16, intermediate=4096, vocab=30522, max_pos=512)
MINILM_L6 = dict(hidden=384, layers=6, heads=12, intermediate=1536, vocab=30522, max_pos=512)


def random_bert_state_dict(cfg: BertConfigLite, seed=0, device="cuda", dtype=torch.float16, std=0.02):
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s: (torch.randn(*s, device=device, generator=g) * std).to(dtype)
    H, I = cfg.hidden, cfg.intermediate
    sd = {"embeddings.word_embeddings.weight": rn(cfg.vocab, H), "embeddings.position_embeddings.weight": rn(cfg.max_pos, H),
          "embeddings.token_type_embeddings.weight": rn(2, H), "embeddings.LayerNorm.weight": 1 + rn(H), "embeddings.LayerNorm.bias": rn(H)}
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            sd[p + f"attention.self.{n}.weight"] = rn(H, H); sd[p + f"attention.self.{n}.bias"] = rn(H)
        sd[p + "attention.output.dense.weight"] = rn(H, H); sd[p + "attention.output.dense.bias"] = rn(H)
        sd[p + "attention.output.LayerNorm.weight"] = 1 + rn(H); sd[p + "attention.output.LayerNorm.bias"] = rn(H)
        sd[p + "intermediate.dense.weight"] = rn(I, H); sd[p + "intermediate.dense.bias"] = rn(I)
        sd[p + "output.dense.weight"] = rn(H, I); sd[p + "output.dense.bias"] = rn(H)
        sd[p + "output.LayerNorm.weight"] = 1 + rn(H); sd[p + "output.LayerNorm.bias"] = rn(H)
    return sd


def _h(t, device):
    return t.detach().to(device=device, dtype=torch.float16).contiguous()


class BertEncoder:
    def __init__(self, state_dict, cfg: BertConfigLite, device="cuda", prefix=""):
        self.cfg, self.device = cfg, torch.device(device)
        H = cfg.hidden
        self.dh = H // cfg.heads
        if H % 128 or cfg.intermediate % 128 or self.dh not in (32, 64, 128):
            raise ValueError("HIP BERT path needs hidden/intermediate multiples of 128 and head_dim in {32, 64, 128}")
        sd, p = state_dict, prefix
        q = lambda n: _h(sd[p + n], device)
        self.word, self.pos = q("embeddings.word_embeddings.weight"), q("embeddings.position_embeddings.weight")
        self.type0 = q("embeddings.token_type_embeddings.weight")[0].contiguous()
        self.eg, self.eb = q("embeddings.LayerNorm.weight"), q("embeddings.LayerNorm.bias")
        self.L = []
        for i in range(cfg.layers):
            lp = f"encoder.layer.{i}."
            self.L.append(dict(
                wqkv=torch.cat([q(lp + "attention.self.query.weight"), q(lp + "attention.self.key.weight"), q(lp + "attention.self.value.weight")]).contiguous(),
                bqkv=torch.cat([q(lp + "attention.self.query.bias"), q(lp + "attention.self.key.bias"), q(lp + "attention.self.value.bias")]).contiguous(),
                wo=q(lp + "attention.output.dense.weight"), bo=q(lp + "attention.output.dense.bias"),
                ln1=(q(lp + "attention.output.LayerNorm.weight"), q(lp + "attention.output.LayerNorm.bias")),
                w1=q(lp + "intermediate.dense.weight"), b1=q(lp + "intermediate.dense.bias"),
                w2=q(lp + "output.dense.weight"), b2=q(lp + "output.dense.bias"),
                ln2=(q(lp + "output.LayerNorm.weight"), q(lp + "output.LayerNorm.bias"))))

    def forward(self, input_ids, lengths=None):
        """input_ids [B, L] (right-padded), lengths [B] valid tokens -> last_hidden_state [B, L, H] fp16."""
        c = self.cfg
        B, Ls = input_ids.shape
        H = c.hidden
        ids = input_ids.to(self.device)
        x = ops.bert_embed_ln(ids, self.word, self.pos, self.type0, self.eg, self.eb, c.eps)
        kv = None if lengths is None else torch.as_tensor(lengths, dtype=torch.int32, device=self.device)
        for L in self.L:
            qkv = ops.gemm(x, L["wqkv"], L["bqkv"]).view(B, Ls, 3 * H)
            att = ops.attention(qkv[..., :H], qkv[..., H:2 * H], qkv[..., 2 * H:], c.heads, c.heads, self.dh, self.dh ** -0.5,
                                kv_len=kv).view(B * Ls, H)
            y = ops.gemm(att, L["wo"], L["bo"], residual=x)
            x = ops.layernorm(y, L["ln1"][0], L["ln1"][1], c.eps)
            f = ops.gemm(x, L["w1"], L["b1"], epilogue="gelu")
            y = ops.gemm(f, L["w2"], L["b2"], residual=x)
            x = ops.layernorm(y, L["ln2"][0], L["ln2"][1], c.eps)
        return x.view(B, Ls, H)

    # transformers-style call so the reference's `model(**ids).last_hidden_state` keeps working (utiles.py:707)
    def __call__(self, input_ids=None, attention_mask=None, **kw):
        lengths = None if attention_mask is None else attention_mask.sum(dim=1)
        return types.SimpleNamespace(last_hidden_state=self.forward(input_ids, lengths))

    def embed_cls(self, input_ids, lengths=None):
        return ops.pool(self.forward(input_ids, lengths), lengths, "cls", normalize=False)


class SentenceEmbedder:
    """sentence-transformers pipeline of all-MiniLM-L6-v2: encoder -> masked mean pooling -> L2 normalise -> fp32 [B, 384]."""

    def __init__(self, encoder: BertEncoder, max_seq_len=256):
        self.encoder, self.max_seq_len = encoder, max_seq_len

    def embed(self, input_ids, lengths):
        input_ids = input_ids[:, : self.max_seq_len]
        lengths = torch.as_tensor(lengths).clamp(max=self.max_seq_len)
        return ops.pool(self.encoder.forward(input_ids, lengths), lengths, "mean", normalize=True)


class HashTokenizer:
    """Deterministic whitespace tokenizer for synthetic runs (no vocabulary files are reachable offline): word ->
    crc32 % (vocab - 1000) + 1000, [CLS]=101 ... [SEP]=102, right padding with 0.  Call surface of a HF tokenizer."""

    def __init__(self, vocab=30522, max_len=512):
        self.vocab, self.max_len = vocab, max_len

    def encode(self, text):
        import zlib
        return [101] + [zlib.crc32(w.encode()) % (self.vocab - 1000) + 1000 for w in text.split()][: self.max_len - 2] + [102]

    def __call__(self, text, padding=True, return_tensors="pt", **kw):
        texts = [text] if isinstance(text, str) else list(text)
        enc = [self.encode(t) for t in texts]
        L = max(len(e) for e in enc)
        ids = torch.zeros((len(enc), L), dtype=torch.long)
        mask = torch.zeros((len(enc), L), dtype=torch.long)
        for i, e in enumerate(enc):
            ids[i, : len(e)] = torch.tensor(e)
            mask[i, : len(e)] = 1
        return {"input_ids": ids, "attention_mask": mask}
