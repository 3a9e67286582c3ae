"""GPU: preprocess and similarity top-k through the C ABI against numpy / the oracle."""
import json
import os

import numpy as np
import pytest
import torch

import oracle
from streamchat_amd import ops, utiles as U

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def ref_preprocess(u8):
    # transformers image_transforms.rescale (float64 product -> float32) + normalize (float32)
    x = (u8.astype(np.float64) * (1 / 255)).astype(np.float32)
    x = (x - np.asarray(ops.CLIP_MEAN, np.float32)) / np.asarray(ops.CLIP_STD, np.float32)
    return x.transpose(0, 3, 1, 2).astype(np.float16)


@pytest.mark.parametrize("n,h,w", [(3, 336, 336), (2, 28, 42), (1, 14, 14)])
def test_preprocess_bit_exact(n, h, w):
    u8 = np.random.default_rng(1234).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    out = ops.preprocess_u8(torch.from_numpy(u8).cuda())
    assert np.array_equal(out.cpu().numpy().view(np.uint16), ref_preprocess(u8).view(np.uint16))


def test_preprocess_matches_hf_processor_golden():
    """G1: bit-exact against the output of HF CLIPImageProcessor (what reference utiles.py:71-87 calls) on 4 seeded frames."""
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "preprocess.npz"))
    u8 = np.random.default_rng(int(d["seed"])).integers(0, 256, tuple(d["shape"]), dtype=np.uint8)
    import zlib
    assert zlib.crc32(u8[0].tobytes()) == int(d["first_frame_crc"])          # the seeded generator reproduced the fixture's input
    out = ops.preprocess_u8(torch.from_numpy(u8).cuda()).cpu().numpy()
    assert np.array_equal(out[:, :, ::8, ::8].view(np.uint16), d["sub"].view(np.uint16))
    assert out.astype(np.float64).sum() == float(d["sum64"])


def test_preprocess_patchify_matches_unfold():
    n, h, w, P, ld = 2, 56, 42, 14, 640
    u8 = np.random.default_rng(7).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    out = ops.preprocess_patchify_u8(torch.from_numpy(u8).cuda(), P, ld).cpu()
    chw = torch.from_numpy(ref_preprocess(u8))                       # [n,3,h,w]
    ref = chw.unfold(2, P, P).unfold(3, P, P)                        # [n,3,gh,gw,P,P]
    ref = ref.permute(0, 2, 3, 1, 4, 5).reshape(n * (h // P) * (w // P), 3 * P * P)
    assert torch.equal(out[:, : 3 * P * P], ref)
    assert torch.count_nonzero(out[:, 3 * P * P:]) == 0


@pytest.mark.parametrize("metric", ["cos", "l2"])
@pytest.mark.parametrize("M,d,k", [(26, 1024, 1), (205, 1024, 8), (32, 384, 1), (1, 16, 1), (300, 130, 64)])
def test_sim_topk_indices_match_oracle(metric, M, d, k):
    rng = np.random.default_rng(M * 7 + d)
    docs = rng.standard_normal((M, d)).astype(np.float32)
    q = rng.standard_normal(d).astype(np.float32)
    k = min(k, M)
    idx, sc = ops.sim_topk(torch.from_numpy(q).cuda(), torch.from_numpy(docs).cuda(), k, metric)
    ridx, rsc = oracle.topk(q, docs, k, metric)
    assert np.array_equal(idx.cpu().numpy(), ridx)                   # identical retrieved indices
    np.testing.assert_allclose(sc.cpu().numpy(), rsc, rtol=1e-5, atol=1e-6)


def test_sim_topk_ties_lowest_index():
    docs = torch.randn(10, 64)
    docs[7] = docs[3]
    idx, _ = ops.sim_topk(docs[3].cuda(), docs.cuda(), 2, "cos")
    assert idx.cpu().tolist() == [3, 7]


class _Tok:
    def __call__(self, text, padding=True, return_tensors="pt"):
        return {"text": text}


class _Model:
    def __init__(self, table):
        self.table = table

    def __call__(self, text):
        import types
        v = torch.tensor(self.table[text], dtype=torch.float32, device="cuda")
        return types.SimpleNamespace(last_hidden_state=torch.stack([v, -v])[None])


def test_tree_search_matches_reference_golden():
    """G6: fast_search_tree_multi_modal_with_embedding (reference utiles.py:685-788) incl. the all-negative case."""
    N = U.MultimodalTreeNode
    for case in json.load(open(os.path.join(G, "search.json"))):
        texts = case["texts"]

        def leaf(name, val):
            return N(torch.full((3, 2, 4), float(val), device="cuda"), name, depth=0)
        if case.get("all_negative"):
            nodes = [leaf(f"neg{i}", i) for i in range(3)]
        else:
            l = [leaf(f"leaf{i}", i) for i in range(9)]
            a = N(torch.full((2, 2, 4), 100.0, device="cuda"), "A", depth=1); a.children = l[0:3]
            b = N(torch.full((2, 2, 4), 101.0, device="cuda"), "B", depth=1); b.children = l[3:6]
            root = N(torch.full((2, 2, 4), 200.0, device="cuda"), "R", depth=2); root.children = [a, b]
            c = N(torch.full((2, 2, 4), 102.0, device="cuda"), "C", depth=1); c.children = l[6:9]
            nodes = [root, c, leaf("red1", 50), leaf("red2", 51)]
        feats, txt = U.fast_search_tree_multi_modal_with_embedding(nodes, "QUERY", torch.zeros(1, device="cuda"), _Model(texts), _Tok(),
                                                                   cache=U.CaptionEmbeddingCache(), batch_captions=False)
        assert txt == case["path_text"]
        assert [float(f.flatten()[0]) for f in feats] == case["path_first_value"]
        assert [list(f.shape) for f in feats] == case["path_shapes"]


@pytest.mark.parametrize("B,P,D,r", [(3, 24, 3584, 2), (2, 24, 64, 5), (1, 6, 16, 6), (2, 4, 8, 1)])
def test_compress_spatial_features_matches_avg_pool2d(B, P, D, r):
    """reference utiles.py:264-289: F.avg_pool2d over the P x P token grid (floor mode), fp32 accumulation, fp16 out."""
    x = torch.randn(B, P * P, D, device="cuda").half()
    out = U.compress_spatial_features(list(torch.split(x, 1)), r)
    ref = torch.nn.functional.avg_pool2d(x.float().reshape(B, P, P, D).permute(0, 3, 1, 2), (r, r)).permute(0, 2, 3, 1).reshape(B, -1, D)
    assert len(out) == B and out[0].shape == (1, (P // r) ** 2, D)
    torch.testing.assert_close(torch.cat(out).float(), ref, rtol=1e-3, atol=1e-3)
