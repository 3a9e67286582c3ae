"""Shared machinery of the composed-parity tests (tests/test_gpu_composed_c2.py: 88 frames, chunk 8; tests/test_gpu_composed_shipped.py:
440 frames at the SHIPPED memory parameters, chunk 40 / K 5 / interval 10, reference inference_streamchat_v0.3.sh:13-18).

The SAME seeded uint8 frames and the same host RNG draws go through

  HIP   fused preprocess -> ViT-L/14-336 (23 layers) -> mlp2x_gelu (fp16) -> streaming.updating_memory_buffer (ONE merge of the first ten
        depth-0 nodes = whole-frame k-means over T = 10 chunks) -> BERT-large-CLS tree search (HIP encoder, sc_sim_topk)
  CPU   oracle/torch_ref fp32 encode -> THE SAME host policy functions (updating_memory_buffer / fast_building_memory_tree_summarize_token
        / fast_search_tree_multi_modal_with_embedding: reference inference_streaming_longva_v2.py:319-358, utiles.py:567-620,715-748)
        with the k-means, top-k and text-encoder providers swapped for oracle.kmeans_fit (C, fp32 features), oracle.topk and torch_ref's
        fp32 BERT

The chunk captioner is a stand-in that names chunks by POSITION (the LLM captioner is outside C2 and would see different bits on the
two sides); the merge summary is derived from the captions it summarises, as upstream."""
import os
import types
import zlib

import numpy as np
import torch

QUESTION = "where did I leave the red cup and what was on the kitchen table"


def crossfade_stream(n, seed=1234, period=16, h=336, w=336, noise=6):
    """uint8 [n, h, w, 3]: frame i cross-fades scene floor(i / period) into the next one (+ small per-frame noise), so the stream walks a
    continuous path through feature space and the k-means boundaries are decided by real distance comparisons.  (synthetic.frame_stream's
    hard scene cuts are trivially separable: relative label margin 0.998 on the CPU side; this stream at 88 frames: 0.099 after 9 Lloyd
    iterations that each move boundary frames - measured with oracle/torch_ref fp32 features when the test was written.)"""
    scenes = {}

    def scene(s):
        if s not in scenes:
            scenes[s] = np.random.default_rng([seed, 0, s]).integers(0, 256, (h, w, 3), dtype=np.uint8).astype(np.float32)
        return scenes[s]
    out = np.empty((n, h, w, 3), np.uint8)
    for i in range(n):
        s, a = divmod(i, period)
        a = a / period
        d = np.random.default_rng([seed, 1, i]).integers(-noise, noise + 1, (h, w, 3)).astype(np.float32)
        out[i] = np.clip((1 - a) * scene(s) + a * scene(s + 1) + d, 0, 255).astype(np.uint8)
    return out


class PositionCaptioner:
    """chunk n (in call order) -> synthetic.caption(n); a summary -> a caption derived from the prompt ids (i.e. from the captions merged)"""
    config = types.SimpleNamespace(mm_use_im_start_end=False)

    def __init__(self, device):
        self.device, self.n = device, 0

    def generate_with_image_embedding(self, ids, image_embeddings=None, **kw):
        if image_embeddings is not None:
            self.n += 1
            return torch.tensor([[self.n - 1]])
        key = torch.as_tensor(ids).reshape(-1).to("cpu", torch.int64).numpy().tobytes()
        return torch.tensor([[1000 + zlib.crc32(key) % 1000]])


def describe(nodes):
    def one(n):
        return dict(depth=n.depth, rows=int(n.centroids.shape[0]), text=n.text, children=[one(c) for c in n.children])
    return [one(n) for n in nodes]


def frame_index(t, bank0, row_elems):
    return (t.storage_offset() - bank0.storage_offset()) // row_elems


def run_policy(feats, colbert, tok, record, mem):
    """the host policy on one feature bank (device or CPU): returns what was decided"""
    import random
    from streamchat_amd import streaming as S, synthetic, utiles as U
    bank = [feats[i:i + 1] for i in range(feats.shape[0])]
    cap, stok = PositionCaptioner(feats.device), synthetic.SyntheticTokenizer()
    torch.manual_seed(0)                                         # init_idx = CPU randperm(T)[:K] inside weighted_kmeans_feature
    random.seed(0)
    tree, short = S.updating_memory_buffer(bank, None, cap, stok, True, rng=np.random.RandomState(0), **mem)
    row = feats[0].numel()
    short_idx = [int(frame_index(t, feats, row)) for t in short]
    path, texts = U.fast_search_tree_multi_modal_with_embedding(tree, QUESTION, feats, colbert, tok, cache=U.CaptionEmbeddingCache())
    retrieved = []
    for t in path:                                               # a retrieved node is a run of whole frames of the bank (depth-0 chunk)
        assert t.shape[0] == mem["chunk_size"]
        f0 = int(frame_index(t, feats, row))
        retrieved.append(list(range(f0, f0 + t.shape[0])))
    return dict(tree=describe(tree), short=short_idx, texts=list(texts), retrieved=retrieved, **record)


def build(n_frames, mem, period, micro_batch, cpu_workers=1, cpu_batch=8):
    """both sides of the chain on one seeded cross-fade stream -> dict(hip=..., cpu=..., feats (device fp16), ref (host fp32), gaps, dev)"""
    import oracle
    from oracle import torch_ref as R
    from streamchat_amd import ops, text as T, utiles as U, vision as V
    dev = torch.device("cuda:0")
    cfg = V.CLIPVisionConfigLite(**V.VIT_L_336)
    sd_vit = V.random_clip_state_dict(cfg, seed=0, device=dev)
    sd_proj = V.random_projector_state_dict(1024, 3584, seed=1, device=dev)
    enc = V.FrameEncoder(V.CLIPVisionTower(sd_vit, cfg, device=dev), V.MMProjector(sd_proj, device=dev), micro_batch=micro_batch)
    u8 = crossfade_stream(n_frames, period=period)
    bl = T.BertConfigLite(**T.BERT_LARGE)
    sd_bert = T.random_bert_state_dict(bl, seed=2, device=dev)
    tok = T.HashTokenizer()

    import time
    t_start = time.time()
    # ---- HIP path (the product functions as they are) ----
    feats = enc.encode_frames_u8(torch.from_numpy(u8).to(dev))                     # [n, 576, 3584] fp16
    rec_hip = {}
    real_km = U.weighted_kmeans_feature

    def km_hip(x, k, *a, **kw):
        red, labels, info = real_km(x, k, *a, return_info=True, **kw)
        rec_hip.update(labels=labels.cpu().numpy(), exit_iter=int(info["info"][0]), T=int(x.shape[0]))
        return red, labels
    U.weighted_kmeans_feature = km_hip
    try:
        hip = run_policy(feats, T.BertEncoder(sd_bert, bl, device=dev), tok, rec_hip, mem)
    finally:
        U.weighted_kmeans_feature = real_km

    t_hip = time.time()
    # ---- CPU path: fp32 encode (oracle/torch_ref), same policy functions, oracle providers ----
    cores = os.cpu_count() or 1
    usable = R.host_cpu_budget()[1]                                 # cores the process may really burn (cgroup quota: 16 on the pool's 256-core hosts)
    workers, threads = (1, min(32, usable)) if cpu_workers <= 1 else R.parallel_plan(n_frames, batch=cpu_batch)
    torch.set_num_threads(min(32, usable))
    ref = R.encode_frames_u8_parallel(sd_vit, sd_proj, u8, workers=workers, threads=threads, batch=cpu_batch)
    t_enc = time.time()
    rec_cpu = {}

    def km_cpu(img_feature, K, weights=None, *, init_idx=None, reseed_idx=None, max_iter=10, **kw):
        import random
        Tn, P, D = img_feature.shape
        if init_idx is None:
            init_idx = torch.randperm(Tn)[:K]
        if reseed_idx is None:
            reseed_idx = [random.randint(0, Tn - 1) for _ in range(max_iter * K)]
        X = img_feature.reshape(Tn, -1).numpy()
        o = oracle.kmeans_fit(X, K, np.asarray(init_idx, np.int32), np.asarray(reseed_idx, np.int32), max_iter=max_iter, trace=True)
        d2 = np.sort(oracle.kmeans_dist2(X, o["centroids"]), axis=1)
        rec_cpu.update(labels=o["labels"], exit_iter=o["iters"], T=Tn, margin=(d2[:, 1] - d2[:, 0]) / d2[:, 1], trace=o["trace"])
        return torch.from_numpy(o["centroids"]).view(K, P, D), torch.from_numpy(o["labels"])

    gaps = []

    def topk_cpu(q, docs, k=1, metric="cos"):
        idx, sc = oracle.topk(q.numpy(), docs.numpy(), k, metric)
        s = np.sort(torch.nn.functional.cosine_similarity(q[None], docs).numpy())[::-1]
        gaps.append(float(s[0] - s[1]) if len(s) > 1 else float("inf"))
        return torch.from_numpy(idx), torch.from_numpy(sc)

    sdc = {k: v.float().cpu() for k, v in sd_bert.items()}

    class RefBert:
        def __call__(self, input_ids=None, attention_mask=None, **kw):
            with torch.no_grad():
                return types.SimpleNamespace(last_hidden_state=R.bert_last_hidden(sdc, input_ids.cpu(), attention_mask.cpu(), heads=16, layers=24))
    saved = (U.weighted_kmeans_feature, ops.sim_topk)
    U.weighted_kmeans_feature, ops.sim_topk = km_cpu, topk_cpu
    try:
        cpu = run_policy(ref, RefBert(), tok, rec_cpu, mem)
    finally:
        U.weighted_kmeans_feature, ops.sim_topk = saved
    print(f"\n[composed] {n_frames} frames: HIP side {t_hip - t_start:.1f} s, fp32 host encode {t_enc - t_hip:.1f} s ({workers} workers x "
          f"{threads} threads), host policy with oracle providers {time.time() - t_enc:.1f} s")
    return dict(hip=hip, cpu=cpu, feats=feats, ref=ref, gaps=gaps, dev=dev)


def prefill_both(c2, layers=2, vocab=8192, row_chunk=None):
    """The LAST stage of the chain on what the stages above produced: the [short | retrieved] frame tokens spliced into the reference's
    answer prompt and prefilled through `layers` Qwen2 layers at the 7B widths - HIP on the fp16 HIP features vs oracle/torch_ref fp32 on the
    CPU path's fp32 features.  Returns (hip last-position logits, fp32 logits, context length)."""
    from oracle import torch_ref as R
    from streamchat_amd import llm as LM, streaming as S, synthetic
    from streamchat_amd.conversation import conv_templates
    from streamchat_amd.mm_utils import tokenizer_image_token
    dev = c2["dev"]
    assert c2["hip"]["retrieved"] == c2["cpu"]["retrieved"] and c2["hip"]["short"] == c2["cpu"]["short"]
    frames = c2["hip"]["short"] + [f for r in c2["hip"]["retrieved"] for f in r]
    cfg = LM.Qwen2ConfigLite(**dict(LM.QWEN2_7B, layers=layers, vocab=vocab))
    sd = LM.random_qwen2_state_dict(cfg, seed=4, device=dev)
    qs = S.build_answer_prompt(QUESTION, c2["hip"]["texts"][-1], None)
    conv = conv_templates["qwen_1_5"].copy()
    conv.append_message(conv.roles[0], qs)
    conv.append_message(conv.roles[1], None)
    ids = tokenizer_image_token(conv.get_prompt(), synthetic.SyntheticTokenizer(), -200, return_tensors="pt")
    ids = torch.where(ids >= 0, ids % cfg.vocab, ids)                       # synthetic ids into the small test vocabulary (the sentinel stays -200)
    assert int((ids == -200).sum()) == 1
    # ---- HIP: pieces spliced without a cat, prefill ----
    model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, cfg, device=dev, max_seq=len(frames) * 576 + ids.numel() + 8))
    pieces = [c2["feats"][f].reshape(-1, 3584) for f in frames]
    _, _, _, _, embeds, _ = model.prepare_inputs_embeddings_for_multimodal(ids.unsqueeze(0), None, None, None, None, [pieces], ["video"])
    assert embeds.shape[1] == len(frames) * 576 + ids.numel() - 1
    model.lm.reset_cache()
    logits = model.lm.forward(embeds[0]).float().cpu()
    # ---- CPU fp32: the same prompt rows around the fp32 features of the same frames ----
    table = sd["model.embed_tokens.weight"].float().cpu()
    p = int((ids == -200).nonzero()[0])
    img = torch.cat([c2["ref"][f].reshape(-1, 3584) for f in frames])
    emb32 = torch.cat([table[ids[:p]], img, table[ids[p + 1:]]])
    sd32 = {k: v.float().cpu() for k, v in sd.items()}
    torch.set_num_threads(min(32, R.host_cpu_budget()[1]))
    with torch.no_grad():
        ref = R.qwen2_logits(sd32, emb32, heads=cfg.heads, kv_heads=cfg.kv_heads, layers=layers, head_dim=cfg.head_dim, theta=cfg.rope_theta, eps=cfg.eps,
                             last_only=True, head_chunk=None if row_chunk else 4, row_chunk=row_chunk)
    return logits, ref, emb32.shape[0]
