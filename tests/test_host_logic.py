"""CPU: host-side mirror (streamchat_amd/utiles.py) against golden traces from the reference's functions.
The k-means provider is swapped for the ORACLE here (test infrastructure) so that the tree policy can be
exercised without a GPU; tests/test_gpu_kmeans.py runs the same traces through the HIP path."""
import json
import os
import types

import numpy as np
import pytest
import torch

import oracle
from streamchat_amd import utiles as U

G = os.path.join(os.path.dirname(__file__), "golden")


def test_forgetting_probabilities_and_selection():
    d = np.load(os.path.join(G, "forgetting.npz"))
    for key in d.files:
        if key.startswith("p_"):
            _, L, tau = key.split("_")
            np.testing.assert_array_equal(U.calculate_forgetting_probabilities(int(L), tau=int(tau)), d[key])
    for row in d["select_20_5_5"]:
        seed, want = int(row[0]), [int(x) for x in row[1:]]
        np.random.seed(seed)
        p = U.calculate_forgetting_probabilities(20, tau=5)
        assert U.select_data_without_replacement(list(range(100, 120)), p, 5) == want
        # explicit rng path: same stream when seeded identically
        assert U.select_data_without_replacement(list(range(100, 120)), p, 5, rng=np.random.RandomState(seed)) == want


class FakeTok:
    def __call__(self, text, **kw):
        return types.SimpleNamespace(input_ids=[1, 2, 3])

    def batch_decode(self, ids, skip_special_tokens=True):
        return [f" caption#{int(ids[0][0])} "]


class FakeSummarizer:
    device = "cpu"

    def __init__(self):
        self.n = 0

    def generate_with_image_embedding(self, ids, image_embeddings=None, **kw):
        self.n += 1
        return torch.tensor([[self.n]])


def oracle_kmeans_feature(img_feature, K, weights=None, **kw):
    T, P, D = img_feature.shape
    init = torch.randperm(T)[:K]
    r = oracle.kmeans_fit(img_feature.reshape(T, -1).numpy(), K, init.numpy().astype(np.int32), np.zeros(10 * K, np.int32))
    return torch.from_numpy(r["centroids"]).view(K, P, D), torch.from_numpy(r["labels"])


def describe(nodes):
    def one(n):
        return dict(depth=n.depth, shape=list(n.centroids.shape), text=n.text, children=[one(c) for c in n.children])
    return [one(n) for n in nodes]


def test_tree_policy_trace(monkeypatch):
    """Q9 / G5: highest eligible depth merges first, one merge per update (reference utiles.py:525-536,574-614)."""
    monkeypatch.setattr(U, "weighted_kmeans_feature", oracle_kmeans_feature)
    cases = json.load(open(os.path.join(G, "tree_trace.json")))
    for c in cases:
        chunk, K, interval, P, D = c["chunk"], c["K"], c["interval"], c["P"], c["D"]
        torch.manual_seed(100 + chunk)
        summ, tok = FakeSummarizer(), FakeTok()
        tree, gframe = None, 0
        for upd in c["trace"]:
            buf = []
            for _ in range(c["frames_per_update"]):
                buf.append(torch.full((1, P, D), float(gframe)) + 0.01 * torch.randn(1, P, D)); gframe += 1
            chunked = [buf[i:i + chunk] for i in range(0, len(buf), chunk)]
            km = [torch.cat(x) for x in chunked]
            tree = U.fast_building_memory_tree_summarize_token(km, K, interval, summ, torch.zeros(1, 3, dtype=torch.long), tok,
                                                               chunked, tree, conv_templates=None)
            assert describe(tree) == upd["top"]
            cnt = U.count_nodes_by_depth(tree)
            assert {int(k): int(v) for k, v in cnt.items()} == {int(k): v for k, v in upd["count"].items()}


def test_batched_captions_build_the_same_tree(monkeypatch):
    """batch_captions=True routes all chunks of one update through ONE generate_batch_with_image_embedding call; with a
    deterministic summarizer the tree (texts, shapes, merge) is the one the chunk-by-chunk path builds."""
    monkeypatch.setattr(U, "weighted_kmeans_feature", oracle_kmeans_feature)

    class BatchSummarizer(FakeSummarizer):
        batch_calls = 0

        def generate_batch_with_image_embedding(self, ids_list, feats_list, **kw):
            self.batch_calls += 1
            assert len(ids_list) == len(feats_list) and all(f[0].dim() == 2 for f in feats_list)
            return [self.generate_with_image_embedding(i, image_embeddings=f) for i, f in zip(ids_list, feats_list)]

    def build(batch):
        torch.manual_seed(3)
        summ, tok, tree = BatchSummarizer(), FakeTok(), None
        for upd in range(3):
            buf = [torch.full((1, 2, 4), float(upd * 8 + i)) for i in range(8)]
            chunked = [buf[i:i + 4] for i in range(0, 8, 4)]
            tree = U.fast_building_memory_tree_summarize_token([torch.cat(x) for x in chunked], 2, 3, summ, torch.zeros(1, 3, dtype=torch.long), tok,
                                                               chunked, tree, conv_templates=None, batch_captions=batch)
        return describe(tree), summ.batch_calls
    seq, n0 = build(False)
    bat, n1 = build(True)
    assert seq == bat and n0 == 0 and n1 == 3


def test_get_summarize_depth():
    N = U.MultimodalTreeNode
    nodes = [N(None, "", depth=1)] * 3 + [N(None, "", depth=0)] * 3
    assert U.get_summarize_depth(nodes, 3)[0] == 1
    nodes = [N(None, "", depth=2)] + [N(None, "", depth=0)] * 4
    assert U.get_summarize_depth(nodes, 3)[0] == 0        # nothing eligible -> 0


def test_kmeans_small_T_returns_three_tuple():
    d = np.load(os.path.join(G, "kmeans_small_T.npz"))
    X = torch.from_numpy(d["X"])
    out = U.weighted_kmeans_feature(X, int(d["K"]))          # T <= K: no device work (Q2)
    assert len(out) == 3
    assert torch.equal(out[0], X) and torch.equal(out[1], torch.from_numpy(d["out1"]))
    assert np.array_equal(np.asarray(out[2], np.int64), d["out2"])


def test_conversation_templates_chatml():
    from streamchat_amd.conversation import conv_templates
    conv = conv_templates["qwen_1_5"].copy()
    conv.append_message(conv.roles[0], "<image>\nwhat?")
    conv.append_message(conv.roles[1], None)
    assert conv.get_prompt() == ("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n<image>\nwhat?<|im_end|>\n"
                                 "<|im_start|>assistant\n")
    assert conv_templates["qwen_1_5"].messages == []       # copy() does not alias
