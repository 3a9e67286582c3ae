"""GPU: ragged / degenerate inputs through the C ABI (the reference has no tests of its own; these mirror the edge cases its
code paths can hit: single-row segments, T = K+1, one-token prompts, fully padded rows, loud argument errors)."""
import numpy as np
import pytest
import torch

import oracle
from streamchat_amd import ops, utiles as U
from streamchat_amd._lib import StreamChatHipError

pytestmark = pytest.mark.gpu


def test_kmeans_T_equals_K_plus_one_and_K1():
    X = torch.randn(6, 520).cuda().half()
    C, labels, wsum, info = ops.kmeans_fit(X, 5, [0, 1, 2, 3, 4], [0] * 50)
    ref = oracle.kmeans_fit(X.cpu().numpy(), 5, np.arange(5, dtype=np.int32), np.zeros(50, np.int32))
    assert np.array_equal(labels.cpu().numpy(), ref["labels"]) and np.array_equal(C.cpu().numpy(), ref["centroids"])
    C, labels, wsum, info = ops.kmeans_fit(X, 1, [3], None)                         # K = 1: the mean of everything
    assert labels.cpu().tolist() == [0] * 6 and float(wsum[0]) == 6.0
    torch.testing.assert_close(C[0], X.float().mean(0), rtol=1e-5, atol=1e-5)


def test_kmeans_identical_rows_all_tie():
    X = torch.ones(10, 512).cuda()
    C, labels, wsum, info = ops.kmeans_fit(X, 3, [0, 1, 2], [5, 6, 7] * 10)
    assert labels.cpu().tolist() == [0] * 10                                      # every distance ties -> first centroid
    assert int(info[1]) == 0


def test_kmeans_bad_arguments_fail_loudly():
    X = torch.randn(8, 64).cuda()
    with pytest.raises(StreamChatHipError):
        ops.kmeans_fit(X, 3, [0, 1])                                              # wrong init length
    with pytest.raises(StreamChatHipError):
        ops.kmeans_fit(X.to(torch.float64), 2, [0, 1])                            # unsupported dtype


def test_attention_single_query_and_fully_padded_row():
    q, k, v = torch.randn(2, 1, 128).cuda().half(), torch.randn(2, 70, 128).cuda().half(), torch.randn(2, 70, 128).cuda().half()
    kv_len = torch.tensor([70, 0], device="cuda", dtype=torch.int32)               # second batch row: no valid key at all
    out = ops.attention(q, k, v, 2, 2, 64, 0.125, False, kv_len)
    assert torch.isfinite(out).all() and torch.count_nonzero(out[1]) == 0         # defined as zeros, never NaN
    s = (q[0, :, :64].float() @ k[0, :, :64].float().T) * 0.125
    ref = torch.softmax(s, -1) @ v[0, :, :64].float()
    torch.testing.assert_close(out[0, :, :64].float(), ref, rtol=2e-3, atol=2e-3)


def test_gemm_single_row_and_tail_rows():
    for M in (1, 127, 129, 1025):
        a, w = torch.randn(M, 128).cuda().half(), (torch.randn(256, 128) / 11).cuda().half()
        torch.testing.assert_close(ops.gemm(a, w).float(), a.float() @ w.float().T, rtol=2e-3, atol=2e-3)


def test_topk_k_equals_M_and_errors():
    docs = torch.randn(5, 32).cuda()
    idx, sc = ops.sim_topk(docs[2], docs, 5, "l2")
    assert idx[0].item() == 2 and sorted(idx.cpu().tolist()) == [0, 1, 2, 3, 4] and float(sc[0]) == 0.0
    with pytest.raises(StreamChatHipError):
        ops.sim_topk(docs[0], docs, 6, "cos")                                     # k > M


def test_tree_search_empty_and_single_node():
    class Tok:
        def __call__(self, t, padding=True, return_tensors="pt"):
            return {"text": t}

    class Model:
        def __call__(self, text):
            import types
            g = torch.Generator().manual_seed(abs(hash(text)) % 1000)
            v = torch.randn(16, generator=g).cuda()
            return types.SimpleNamespace(last_hidden_state=torch.stack([v, v])[None])
    feats, txt = U.fast_search_tree_multi_modal_with_embedding([], "q", torch.zeros(1, device="cuda"), Model(), Tok(), batch_captions=False)
    assert feats == [] and txt == []
    n = U.MultimodalTreeNode(torch.ones(2, 2, 4, device="cuda"), "only", depth=0)
    feats, txt = U.fast_search_tree_multi_modal_with_embedding([n], "q", torch.zeros(1, device="cuda"), Model(), Tok(), batch_captions=False)
    assert txt == ["only"] and feats[0] is n.centroids


def test_preprocess_rejects_bad_input():
    with pytest.raises(StreamChatHipError):
        ops.preprocess_u8(torch.zeros(1, 4, 4, 3).cuda())                          # not uint8
    with pytest.raises(StreamChatHipError):
        ops.preprocess_patchify_u8(torch.zeros(1, 30, 28, 3, dtype=torch.uint8).cuda(), 14, 640)   # 30 % 14 != 0


def test_attention_ignores_garbage_beyond_kv_len():
    """Rows of a KV cache past kv_len hold allocator garbage (possibly NaN / Inf): they must not reach the output."""
    torch.manual_seed(3)
    q = torch.randn(1, 5, 2 * 128).cuda().half()
    kv = torch.randn(1, 200, 2 * 128).cuda().half()
    k, v = kv.clone(), kv.flip(1).contiguous()
    kv_len = torch.tensor([70], device="cuda", dtype=torch.int32)
    ref = ops.attention(q, k[0, :70].clone()[None], v[0, :70].clone()[None], 2, 2, 128, 0.09, False)
    k[:, 70:] = float("nan"); v[:, 70:] = float("inf"); v[:, 100:] = float("nan")
    out = ops.attention(q, k, v, 2, 2, 128, 0.09, False, kv_len)
    assert torch.isfinite(out).all()
    torch.testing.assert_close(out.float(), ref.float(), rtol=1e-3, atol=1e-3)
