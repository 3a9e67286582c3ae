"""CPU: oracle/torch_ref.py (fp32 restatement of the transformers arithmetic) against golden vectors produced by the
REAL HF modules (tools/make_golden_hf.py) — this is what pins the third-party part of the oracle."""
import json
import os

import numpy as np
import torch

from oracle import torch_ref as R

G = os.path.join(os.path.dirname(__file__), "golden")


def test_clip_tower_and_projector_vs_hf():
    d = np.load(os.path.join(G, "clip_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("vit.")}
    sp = {k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("proj.")}
    px = torch.from_numpy(d["pixel_values"])
    h = R.clip_vision_hidden(sd, px, heads=2, patch=14, layers_run=2)           # hidden_states[-2] of a 3-layer tower
    torch.testing.assert_close(h[:, 1:], torch.from_numpy(d["features"]), rtol=1e-4, atol=1e-5)
    y = R.encode_images(sd, sp, px, heads=2, patch=14, num_layers=3)
    torch.testing.assert_close(y, torch.from_numpy(d["projected"]), rtol=1e-4, atol=1e-5)


def test_bert_vs_hf():
    d = np.load(os.path.join(G, "bert_tiny.npz"))
    sd = {k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("bert.")}
    ids, mask = torch.from_numpy(d["input_ids"]), torch.from_numpy(d["attention_mask"])
    h = R.bert_last_hidden(sd, ids, mask, heads=4, layers=2)
    m = mask.bool()
    torch.testing.assert_close(h[m], torch.from_numpy(d["last_hidden_state"])[m], rtol=1e-4, atol=1e-5)
    e = R.sentence_embedding(sd, ids, mask, heads=4, layers=2)
    torch.testing.assert_close(e.norm(dim=1), torch.ones(3), rtol=1e-5, atol=1e-5)


def test_tokenizer_image_token_vs_reference():
    from streamchat_amd.mm_utils import tokenizer_image_token
    import types

    class Tok:
        def __init__(s, bos):
            s.bos_token_id = bos

        def __call__(s, t):
            ids = [ord(c) % 50 + 2 for c in t]
            return types.SimpleNamespace(input_ids=([s.bos_token_id] + ids) if s.bos_token_id is not None else ids)
    for c in json.load(open(os.path.join(G, "tokenizer_image_token.json"))):
        assert tokenizer_image_token(c["prompt"], Tok(c["bos"]), -200) == c["ids"]


def test_qwen2_vs_hf():
    d = np.load(os.path.join(G, "qwen2_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("lm.")}
    lg = R.qwen2_logits(sd, torch.from_numpy(d["inputs_embeds"]), heads=4, kv_heads=2, layers=2, head_dim=64)
    torch.testing.assert_close(lg, torch.from_numpy(d["logits"]), rtol=1e-4, atol=1e-4)
    # greedy continuation with the oracle reproduces HF generate()
    emb = torch.from_numpy(d["inputs_embeds"])
    toks = []
    for _ in range(8):
        t = int(R.qwen2_logits(sd, emb, heads=4, kv_heads=2, layers=2, head_dim=64)[-1].argmax())
        toks.append(t)
        emb = torch.cat([emb, sd["model.embed_tokens.weight"][t][None]])
    assert toks == d["greedy"].tolist()


def test_preprocess_restatement_matches_hf_processor_golden():
    """G1: oracle.torch_ref.preprocess_u8 == HF CLIPImageProcessor output (fixture made by tools/make_golden_hf.py)."""
    import zlib
    d = np.load(os.path.join(G, "preprocess.npz"))
    u8 = np.random.default_rng(int(d["seed"])).integers(0, 256, tuple(d["shape"]), dtype=np.uint8)
    assert zlib.crc32(u8[0].tobytes()) == int(d["first_frame_crc"])
    h = R.preprocess_u8(u8).astype(np.float16)
    assert np.array_equal(h[:, :, ::8, ::8].view(np.uint16), d["sub"].view(np.uint16))
    assert h.astype(np.float64).sum() == float(d["sum64"])


def test_parallel_host_encode_equals_the_in_process_loop_and_plans_inside_the_cpu_budget():
    """oracle/torch_ref.encode_frames_u8_parallel (the all-cores fp32 reference encode of bench.py's cpu_baseline and of the composed-parity tests):
    spawned single-threaded workers give the in-process result bit for bit (a batch is the same frames wherever it runs), and parallel_plan never
    asks for more than host_cpu_budget() allows (cgroup quota / affinity: the pool's GPU boxes show 256 cores and grant 16)."""
    import numpy as np
    import torch
    from oracle import torch_ref as R
    from streamchat_amd import vision as V
    present, usable = R.host_cpu_budget()
    assert 1 <= usable <= present
    for n in (1, 7, 64, 440):
        w, t = R.parallel_plan(n, batch=4)
        assert 1 <= w <= max(1, min(usable, (n + 3) // 4)) and 1 <= w * t <= max(usable, 1)
    cfg = V.CLIPVisionConfigLite(hidden=64, layers=3, heads=2, intermediate=128, patch=14, image_size=56)
    sd = V.random_clip_state_dict(cfg, seed=0, device="cpu", std=0.08)
    sp = V.random_projector_state_dict(64, 96, seed=1, device="cpu", std=0.08)
    u8 = np.random.default_rng(0).integers(0, 256, (10, 56, 56, 3), dtype=np.uint8)
    one = R.encode_frames_u8_parallel(sd, sp, u8, workers=1, threads=1, batch=4, heads=2, patch=14, num_layers=3)
    two = R.encode_frames_u8_parallel(sd, sp, u8, workers=2, threads=1, batch=4, heads=2, patch=14, num_layers=3)
    assert one.shape == (10, 16, 96) and torch.equal(one, two)
    with torch.no_grad():
        ref = torch.cat([R.encode_images({k: v.float() for k, v in sd.items()}, {k: v.float() for k, v in sp.items()}, torch.from_numpy(R.preprocess_u8(u8[i:i + 4])),
                                         heads=2, patch=14, num_layers=3) for i in range(0, 10, 4)])
    assert torch.equal(one, ref)
