"""CPU: the oracle (oracle/kmeans_oracle.c) against the golden vectors produced by the REFERENCE's own
functions (tools/make_golden.py, reference utiles.py:291-330).  This is what pins the oracle."""
import glob
import os

import numpy as np
import pytest

import oracle

CASES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "kmeans_0*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_kmeans_oracle_matches_reference(path):
    d = np.load(path)
    X = d["X"]
    T, K = X.shape[0], int(d["K"])
    w = d["weights"] if "weights" in d.files else None
    r = oracle.kmeans_fit(X.reshape(T, -1), K, d["init_idx"], d["reseed_idx"], weights=w, trace=True)
    assert np.array_equal(r["labels"], d["labels"])                      # bit-exact assignments
    assert np.array_equal(r["trace"], d["trace"])                        # ... at every Lloyd iteration
    assert r["iters"] == int(d["exit_iter"])
    np.testing.assert_allclose(r["centroids"], d["centroids"].reshape(K, -1), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r["wsum"], d["wsum"], rtol=1e-6)


def test_kmeans_oracle_dtypes_agree_on_labels():
    d = np.load(CASES[0])
    X = d["X"].reshape(d["X"].shape[0], -1)
    K = int(d["K"])
    r32 = oracle.kmeans_fit(X, K, d["init_idx"], d["reseed_idx"])
    r16 = oracle.kmeans_fit(X.astype(np.float16), K, d["init_idx"], d["reseed_idx"])
    assert np.array_equal(r32["labels"], r16["labels"])
    # the fp16 run must equal an fp32 run on the fp16-rounded data exactly (conversion is exact)
    r16b = oracle.kmeans_fit(X.astype(np.float16).astype(np.float32), K, d["init_idx"], d["reseed_idx"])
    assert np.array_equal(r16["centroids"], r16b["centroids"])


def test_kmeans_oracle_max_iter_exhaustion_q3():
    """Q3: on exhaustion the centroids are one update ahead of the labels (utiles.py:297-318)."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((60, 64)).astype(np.float32)
    init = np.arange(4, dtype=np.int32)
    r1 = oracle.kmeans_fit(X, 4, init, max_iter=1, trace=True)
    assert r1["iters"] == 0
    # labels are the assignment against X[init]; centroids are the means of those labels
    for k in range(4):
        m = r1["labels"] == k
        if m.any():
            np.testing.assert_allclose(r1["centroids"][k], X[m].mean(0), rtol=1e-5, atol=1e-6)


def test_topk_oracle():
    rng = np.random.default_rng(0)
    docs = rng.standard_normal((37, 48)).astype(np.float32)
    q = rng.standard_normal(48).astype(np.float32)
    idx, sc = oracle.topk(q, docs, 5, "cos")
    cos = docs @ q / (np.linalg.norm(docs, axis=1) * np.linalg.norm(q))
    assert list(idx) == list(np.argsort(-cos, kind="stable")[:5])
    idx, sc = oracle.topk(q, docs, 3, "l2")
    l2 = ((docs - q) ** 2).sum(1)
    assert list(idx) == list(np.argsort(l2, kind="stable")[:3])
    docs[5] = docs[2]
    q = docs[2].copy()
    idx, _ = oracle.topk(q, docs, 2, "cos")
    assert list(idx) == [2, 5]           # tie -> lowest index


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_reference_formula_restatement_matches_golden(path):
    """oracle/torch_ref.weighted_kmeans_reference_formula (the [T,K,D] broadcast form the reference runs) vs the golden vectors."""
    import torch
    from oracle import torch_ref as R
    d = np.load(path)
    X = torch.from_numpy(d["X"]).reshape(d["X"].shape[0], -1)
    w = torch.from_numpy(d["weights"]) if "weights" in d.files else None
    C, labels, wsum, it = R.weighted_kmeans_reference_formula(X, int(d["K"]), d["init_idx"], d["reseed_idx"], w)
    assert np.array_equal(labels.numpy(), d["labels"]) and it == int(d["exit_iter"])
    np.testing.assert_allclose(C.numpy(), d["centroids"].reshape(int(d["K"]), -1), rtol=1e-5, atol=1e-6)
