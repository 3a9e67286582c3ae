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


def test_near_tie_pins_the_argmin_footnote():
    """DESIGN section 2, arg-min footnote, pinned by data (fixture: tools/make_golden_r05.py running the reference's own
    weighted_kmeans_feature).  One row has two fp32 squared distances ONE ulp apart whose fp32 square roots are equal: the reference's
    argmin over `.sqrt()` (utiles.py:299-302) takes the first index, the oracle's argmin over the squared distances takes the strictly
    smaller one.  Everything else of the run is identical, and the restated reference formula reproduces the reference exactly."""
    import torch
    from oracle import torch_ref as R
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "kmeans_near_tie.npz"))
    X, K, tie = d["X"], int(d["K"]), int(d["tie_row"])
    T = X.shape[0]
    sA, sB = np.float32(d["sq_dist_first"]), np.float32(d["sq_dist_second"])
    assert sB < sA and sA.view(np.int32) - sB.view(np.int32) == 1 and np.sqrt(sA) == np.sqrt(sB)      # the premise
    # the premise holds on the fixture's rows: fp32 squared distances of the tie row to the two initial centroids
    x, cA, cB = X[tie, 0], X[int(d["init_idx"][0]), 0], X[int(d["init_idx"][1]), 0]
    f = lambda c: np.float32(np.float32((x[0] - c[0]) ** 2) + np.float32((x[1] - c[1]) ** 2))
    assert f(cA) == sA and f(cB) == sB
    r = oracle.kmeans_fit(X.reshape(T, -1), K, d["init_idx"], d["reseed_idx"], trace=True)
    diff0 = np.nonzero(r["trace"][0] != d["trace"][0])[0]
    assert list(diff0) == [tie] and d["trace"][0][tie] == 0 and r["trace"][0][tie] == 1               # the ONE divergent decision
    assert int(d["labels"][tie]) == 0 and int(r["labels"][tie]) == 1                                  # both are Lloyd fixed points
    assert np.array_equal(np.delete(r["labels"], tie), np.delete(d["labels"], tie)) and r["iters"] == int(d["exit_iter"])
    # distances agree to the last bit: the oracle's fp64 totals ARE the fp32 values on this D = 2 fixture
    d2 = oracle.kmeans_dist2(X.reshape(T, -1), X.reshape(T, -1)[d["init_idx"]])
    assert d2[tie, 0] == float(sA) and d2[tie, 1] == float(sB)
    C, labels, wsum, it = R.weighted_kmeans_reference_formula(torch.from_numpy(X).reshape(T, -1), K, d["init_idx"], d["reseed_idx"], None)
    assert np.array_equal(labels.numpy(), d["labels"]) and it == int(d["exit_iter"])
