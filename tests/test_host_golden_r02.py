"""CPU: host-side mirrors against golden vectors produced by the REFERENCE's own functions (tools/make_golden_r02.py):
  a2   frame-index arithmetic of video_reader_thread_with_embedding (inference_streaming_longva_v2.py:454-531) incl. the
       `> 900 -> 200` clamp, the `<= chunk_size` branch and a decoder failure
  a10  the prompt branches of longva_inference_with_embedding_multi_modal (:164-264, Q18) rendered by the reference's
       conversation templates + tokenizer_image_token: prompt text, ids, the [short | long] block, the generate kwargs
  a8   search_tree (utiles.py:909-935)"""
import json
import os
import types
import zlib

import numpy as np
import pytest
import torch

from streamchat_amd import streaming as S, utiles as U
from streamchat_amd.conversation import conv_templates

G = os.path.join(os.path.dirname(__file__), "golden")


# ---------------------------------------------------------------------------------------------------------
# a2
# ---------------------------------------------------------------------------------------------------------
class Cap:
    def __init__(self, fail_after=None):
        self.calls, self.fail_after = [], fail_after

    def read_rgb(self, n):
        self.calls.append(int(n))
        if self.fail_after is not None and len(self.calls) > self.fail_after:
            return None
        return np.full((2, 2, 3), n % 256, np.uint8)


def test_frame_indices_match_reference_seeks():
    cases = json.load(open(os.path.join(G, "frame_indices.json")))
    assert len(cases) >= 12
    model = types.SimpleNamespace(encode_frames_u8=lambda u8: u8.float().mean(dim=(1, 2, 3)).view(-1, 1, 1))
    for c in cases:
        idx = S.sample_frame_indices(c["total_frames"], c["frame_rate"], c["start"], c["end"], c["sample_rate"], c["chunk_size"])
        if c["fail_after"] is None:
            assert idx == c["seeks"], c
        else:
            assert idx[:len(c["seeks"])] == c["seeks"]
        cap = Cap(c["fail_after"])
        bank = S.video_reader_thread_with_embedding(cap, c["total_frames"], c["frame_rate"], None, model, c["start"], c["end"], "cpu",
                                                    c["sample_rate"], chunk_size=c["chunk_size"])
        assert cap.calls == c["seeks"] and len(bank) == c["bank_len"]
        assert [float(t.flatten()[0]) for t in bank] == c["bank_first_value"]
    clamped = [c for c in cases if c["bank_len"] == 200]
    assert len(clamped) >= 3                                                    # the > 900 -> 200 guard is exercised
    # opting out of the clamp (micro-batched encoder): every sampled frame
    assert len(S.sample_frame_indices(9000, 30, 0, 300, 0.2, 40, clamp=None)) == 1800


# ---------------------------------------------------------------------------------------------------------
# a10
# ---------------------------------------------------------------------------------------------------------
class WordTok:
    bos_token_id = None

    def __call__(self, text, **kw):
        return types.SimpleNamespace(input_ids=[zlib.crc32(w.encode()) % 50000 + 10 for w in text.split()])

    def batch_decode(self, ids, skip_special_tokens=True):
        return ["  an answer  "]


def test_answer_prompt_branches_match_reference(monkeypatch):
    fx = json.load(open(os.path.join(G, "answer_prompts.json")))
    captured = {}

    class Model:
        config = types.SimpleNamespace(mm_use_im_start_end=False)

        def generate_with_image_embedding(self, input_ids, image_embeddings=None, **kw):
            e = image_embeddings[0]                  # one image block: a tensor, or its [short | retrieved ...] pieces in order (spliced without a cat)
            captured.update(input_ids=input_ids[0].tolist(), emb=torch.cat(list(e)) if isinstance(e, (list, tuple)) else e.clone(), kw=kw)
            return torch.tensor([[1, 2, 3]])

    def fake_search(tree, question, short, emb_model, emb_tok, **kw):
        return [torch.full((2, 3, 8), 7.0), torch.full((4, 3, 8), 9.0)], ["coarse summary of ten clips", "clip 17: a red cup on the kitchen table"]
    monkeypatch.setattr(U, "fast_search_tree_multi_modal_with_embedding", fake_search)
    short = [torch.full((1, 3, 8), float(i)) for i in range(5)]
    seen = set()
    for c in fx["cases"]:
        captured.clear()
        out, t_proc, t_gen = S.longva_inference_with_embedding_multi_modal(
            fx["question"], 8, c["conv_mode"], Model(), None, WordTok(), None, None, short, ["tree"] if c["has_tree"] else None,
            history_prompt=c["history_prompt"], temperature=0.2, top_p=None, num_beams=1)
        assert out == c["output"]
        assert captured["input_ids"] == c["input_ids"] and captured["input_ids"].count(-200) == c["n_sentinels"]
        assert list(captured["emb"].shape) == c["emb_shape"] and captured["emb"][:, 0].tolist() == c["emb_first_col"]
        kw = {k: v for k, v in captured["kw"].items() if k != "modalities"}
        assert kw == c["gen_kwargs"] and captured["kw"]["modalities"] == ["video"]
        # the rendered prompt itself (conversation template + branch text)
        qs = S.build_answer_prompt(fx["question"], "clip 17: a red cup on the kitchen table" if c["has_tree"] else None, c["history_prompt"])
        conv = conv_templates[c["conv_mode"]].copy()
        conv.append_message(conv.roles[0], qs)
        conv.append_message(conv.roles[1], None)
        assert conv.get_prompt() == c["prompt"]
        seen.add((c["history_prompt"] is not None, c["has_tree"], c["n_sentinels"]))
    # the three live branches of Q18 (+ the no-history / no-tree start of a video)
    assert {(True, True, 1), (False, True, 1), (True, False, 0), (False, False, 1)} <= seen


def test_conversation_templates_match_reference():
    fx = json.load(open(os.path.join(G, "conv_templates.json")))
    assert {"qwen_1_5", "qwen_1_5_ego", "qwen_1_5_summarize"} <= set(fx)
    for name, want in fx.items():
        c = conv_templates[name].copy()
        assert list(c.roles) == want["roles"]
        c.append_message(c.roles[0], "<image>\nwhat do you see?")
        c.append_message(c.roles[1], None)
        assert c.get_prompt() == want["one_turn"]
        c = conv_templates[name].copy()
        for r, m in ((0, "first question"), (1, "first answer"), (0, "second question"), (1, None)):
            c.append_message(c.roles[r], m)
        assert c.get_prompt() == want["two_turns"]


# ---------------------------------------------------------------------------------------------------------
# a8: search_tree
# ---------------------------------------------------------------------------------------------------------
def test_search_tree_matches_reference():
    cases = json.load(open(os.path.join(G, "search_tree.json")))
    depths = set()
    for c in cases:
        def build(d):
            n = U.TreeNode(torch.tensor(d["centroids"]))
            n.children = [build(x) for x in d["children"]]
            return n
        root = build(c["tree"])
        path = U.search_tree(root, torch.tensor(c["query"]))
        assert len(path) == len(c["path"]) and all(torch.equal(p, torch.tensor(w)) for p, w in zip(path, c["path"]))
        depths.add(len(path))
    assert {1, 2, 3, 4} <= depths
