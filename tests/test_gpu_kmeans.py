"""GPU: the HIP k-means (through the C ABI) against the oracle and the reference golden vectors."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import oracle
from streamchat_amd import ops, utiles as U

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(glob.glob(os.path.join(G, "kmeans_0*.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p) for p in CASES])
def test_kmeans_matches_reference_golden(path):
    d = np.load(path)
    X = torch.from_numpy(d["X"]).cuda()
    T, K = X.shape[0], int(d["K"])
    w = torch.from_numpy(d["weights"]).cuda() if "weights" in d.files else None
    C, labels, wsum, info = ops.kmeans_fit(X.reshape(T, -1), K, d["init_idx"], d["reseed_idx"], weights=w)
    assert np.array_equal(labels.cpu().numpy(), d["labels"])                     # bit-exact assignments
    assert int(info[0]) == int(d["exit_iter"]) and int(info[1]) == 0
    np.testing.assert_allclose(C.cpu().numpy(), d["centroids"].reshape(K, -1), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(wsum.cpu().numpy(), d["wsum"], rtol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("T,D,K", [(40, 512 * 3, 5), (64, 1024 + 8, 8), (37, 520, 3), (50, 100, 4), (300, 2048, 16), (70, 4096, 20)])
def test_kmeans_bit_exact_vs_oracle(dtype, T, D, K):
    """labels AND centroids bit-identical to the oracle (shared SC-KM2 reduction spec), incl. unaligned D (100)
    and K > 16 (two centroid tiles).  (40, 1536, 5) and (64, 1032, 8): the lane-mapped kernels; fp16 + D % 64 == 0 + 2 <= K <= 8 takes km2_pass.)"""
    g = torch.Generator().manual_seed(T * 1000 + D + K)
    centres = torch.randn(max(K - 1, 2), D, generator=g)
    X = (centres[torch.randint(0, centres.shape[0], (T,), generator=g)] + 0.3 * torch.randn(T, D, generator=g)).to(dtype)
    init = torch.randperm(T, generator=g)[:K].to(torch.int32)
    reseed = torch.randint(0, T, (10 * K,), generator=g).to(torch.int32)
    Xn = (X.view(torch.int16).numpy().view(np.uint16), "bfloat16") if dtype == torch.bfloat16 else X.numpy()
    ref = oracle.kmeans_fit(Xn, K, init.numpy(), reseed.numpy())
    C, labels, wsum, info = ops.kmeans_fit(X.cuda(), K, init, reseed)
    assert np.array_equal(labels.cpu().numpy(), ref["labels"])
    assert int(info[0]) == ref["iters"]
    assert np.array_equal(C.cpu().numpy(), ref["centroids"])                      # bit-exact, not just close
    assert np.array_equal(wsum.cpu().numpy(), ref["wsum"])


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("T,D,K", [(23, 64 * 9, 8), (64, 2048 * 3, 8), (90, 2048 + 64 * 5, 5), (128, 4096, 5), (200, 64 * 70, 8),
                                   (256, 2048 * 2 + 64, 5), (333, 64 * 41, 5), (400, 2048 * 5, 5), (448, 64 * 33, 8),
                                   (150, 64 * 21, 2), (230, 2048 + 128, 3), (300, 64 * 50, 4), (180, 64 * 37, 6), (390, 2048 * 2, 7)])
def test_one_read_pass_bit_exact_vs_oracle(T, D, K, weighted):
    """km2_pass (fp16, D % 64 == 0, 2 <= K <= 8, T + 7 K <= 448; both slab sizes, short last groups, weights, a forced empty cluster; the small
    shapes take the two-pass kernels) against the oracle: labels, centroids, weights and exit iteration bit-identical."""
    g = torch.Generator().manual_seed(T * 7 + D + K)
    centres = torch.randn(K + 1, D, generator=g)
    X = (centres[torch.randint(0, K + 1, (T,), generator=g)] + 0.5 * torch.randn(T, D, generator=g)).half()
    init = torch.randperm(T, generator=g)[:K].to(torch.int32)
    if T % 2 == 0:
        X[init[1]] = X[init[0]]                        # two identical initial rows: cluster 1 is empty after the first assign -> reseed path
    reseed = torch.randint(0, T, (10 * K,), generator=g).to(torch.int32)
    w = (0.5 + torch.rand(T, generator=g)) if weighted else None
    ref = oracle.kmeans_fit(X.numpy(), K, init.numpy(), reseed.numpy(), weights=None if w is None else w.numpy(), max_iter=6)
    C, labels, wsum, info = ops.kmeans_fit(X.cuda(), K, init, reseed, weights=None if w is None else w.cuda(), max_iter=6)
    assert np.array_equal(labels.cpu().numpy(), ref["labels"])
    assert int(info[0]) == ref["iters"]
    assert np.array_equal(C.cpu().numpy(), ref["centroids"])
    assert np.array_equal(wsum.cpu().numpy(), ref["wsum"])
    lab2, d2 = ops.kmeans_assign(X.cuda(), C, return_dist2=True)
    assert np.array_equal(d2.cpu().numpy(), oracle.kmeans_dist2(X.numpy(), ref["centroids"]))


def test_kmeans_assign_dist2_bit_exact():
    g = torch.Generator().manual_seed(5)
    X = torch.randn(45, 512 * 5 + 64, generator=g).half()
    C = torch.randn(6, X.shape[1], generator=g)
    labels, d2 = ops.kmeans_assign(X.cuda(), C.cuda(), return_dist2=True)
    ref = oracle.kmeans_dist2(X.numpy(), C.numpy())
    assert np.array_equal(d2.cpu().numpy(), ref)                                  # fp64 totals identical
    assert np.array_equal(labels.cpu().numpy(), ref.argmin(1))


def test_kmeans_exact_ties_pick_first():
    X = torch.zeros(16, 512)
    X[8:] = 1.0
    C, labels, wsum, info = ops.kmeans_fit(X.cuda(), 3, [0, 1, 8], [3, 4, 5] * 10)   # centroids 0 and 1 identical
    lab = labels.cpu().numpy()
    assert set(lab[:8]) == {0} and set(lab[8:]) == {2}                            # first minimum wins; cluster 1 empty
    assert int(info[2]) >= 1                                                      # a reseed was consumed


def test_near_tie_fixture_hip_equals_oracle():
    """tests/golden/kmeans_near_tie.npz (the reference's own function on a constructed one-ulp near-tie, tools/make_golden_r05.py): the HIP
    kernel takes the oracle's decision bit for bit - the strictly smaller squared distance - where the reference's argmin over fp32 `.sqrt()`
    collapses the two and takes the first index (utiles.py:299-302; DESIGN section 2, arg-min footnote)."""
    d = np.load(os.path.join(G, "kmeans_near_tie.npz"))
    X, K, tie = d["X"], int(d["K"]), int(d["tie_row"])
    T = X.shape[0]
    ref = oracle.kmeans_fit(X.reshape(T, -1), K, d["init_idx"], d["reseed_idx"])
    C, labels, wsum, info = ops.kmeans_fit(torch.from_numpy(X).cuda().reshape(T, -1), K, d["init_idx"], d["reseed_idx"])
    lab = labels.cpu().numpy()
    assert np.array_equal(lab, ref["labels"]) and np.array_equal(C.cpu().numpy(), ref["centroids"]) and int(info[0]) == ref["iters"]
    assert lab[tie] == 1 and int(d["labels"][tie]) == 0 and np.array_equal(np.delete(lab, tie), np.delete(d["labels"], tie))
    _, d2 = ops.kmeans_assign(torch.from_numpy(X).cuda().reshape(T, -1), torch.from_numpy(X.reshape(T, -1)[d["init_idx"]]).cuda(), return_dist2=True)
    assert d2[tie, 0].item() == float(d["sq_dist_first"]) and d2[tie, 1].item() == float(d["sq_dist_second"])


def test_weighted_kmeans_feature_dropin_shapes():
    d = np.load(CASES[0])
    X = torch.from_numpy(d["X"]).cuda().half()
    red, labels = U.weighted_kmeans_feature(X, int(d["K"]), init_idx=d["init_idx"], reseed_idx=d["reseed_idx"])
    assert red.shape == (int(d["K"]),) + X.shape[1:] and red.dtype == torch.float16
    assert labels.dtype == torch.int64 and np.array_equal(labels.cpu().numpy(), d["labels"])
    small = U.weighted_kmeans_feature(X[:3], 5)
    assert len(small) == 3


def test_kmeans_full_width_properties():
    """BASELINE C1-sized width (D = 576*3584) at a reduced row count: size-independent properties —
    planted clusters are recovered, labels are invariant to a permutation of the columns' chunk order
    is NOT assumed; instead: idempotence (re-fitting from the returned centroids converges at iter 0)."""
    D, T, K = 576 * 3584, 24, 4
    g = torch.Generator(device="cuda").manual_seed(1)
    centres = torch.randn(K, D, device="cuda", generator=g, dtype=torch.float16)
    which = torch.arange(T, device="cuda") % K
    X = centres[which] + 0.05 * torch.randn(T, D, device="cuda", generator=g, dtype=torch.float16)
    C, labels, wsum, info = ops.kmeans_fit(X, K, [0, 1, 2, 3], None)
    assert torch.equal(labels, which)                                             # init rows 0..3 are one per cluster
    assert torch.equal(wsum, torch.full((K,), T / K, device="cuda"))
    lab2 = ops.kmeans_assign(X, C)
    assert torch.equal(lab2, labels)
    # centroid = mean of members (fp32 sequential sum): compare with torch at fp tolerance
    for k in range(K):
        ref = X[which == k].float().mean(0)
        torch.testing.assert_close(C[k], ref, rtol=1e-5, atol=1e-5)


def test_tree_trace_on_gpu():
    """G5 trace through the real HIP k-means (device tensors)."""
    from tests.test_host_logic import FakeSummarizer, FakeTok, describe
    cases = json.load(open(os.path.join(G, "tree_trace.json")))
    c = cases[0]
    chunk, K, interval, P, D = c["chunk"], c["K"], c["interval"], c["P"], c["D"]
    torch.manual_seed(100 + chunk)
    summ, tok = FakeSummarizer(), FakeTok()
    tree, gframe = None, 0
    for upd in c["trace"]:
        buf = []
        for _ in range(c["frames_per_update"]):
            buf.append((torch.full((1, P, D), float(gframe)) + 0.01 * torch.randn(1, P, D)).cuda()); gframe += 1
        chunked = [buf[i:i + chunk] for i in range(0, len(buf), chunk)]
        km = [torch.cat(x) for x in chunked]
        tree = U.fast_building_memory_tree_summarize_token(km, K, interval, summ, torch.zeros(1, 3, dtype=torch.long), tok, chunked, tree)
        got = describe(tree)
        assert [(n["depth"], n["shape"], len(n["children"])) for n in got] == [(n["depth"], n["shape"], len(n["children"])) for n in upd["top"]]


@pytest.mark.parametrize("T,K,centres_n,max_iter", [(64, 8, 6, 10), (400, 5, 5, 3)], ids=["C1_T64_K8", "merge_T400_K5"])
def test_kmeans_full_size_bit_exact_vs_oracle(T, K, centres_n, max_iter):
    """FULL-size k-means (D = 576*3584, fp16 storage): BASELINE C1 (T = 64 frames, K = 8, 0.26 GB) and the shipped memory-merge
    shape (T = 400, K = 5, 1.65 GB; max_iter = 3 keeps the scalar oracle at a few seconds).  Labels, centroids, weights and the
    iteration count are bit-identical to the C oracle."""
    D = 576 * 3584
    g = torch.Generator(device="cuda").manual_seed(T)
    centres = torch.randn(centres_n, D, device="cuda", generator=g)
    which = torch.randint(0, centres_n, (T,), device="cuda", generator=g)
    X = torch.empty(T, D, device="cuda", dtype=torch.float16)
    for i in range(0, T, 50):                                                     # build in slabs: no 3 GB fp32 temporary
        X[i:i + 50] = (centres[which[i:i + 50]] + 0.5 * torch.randn(min(50, T - i), D, device="cuda", generator=g)).half()
    init = torch.randperm(T, generator=torch.Generator().manual_seed(0))[:K].to(torch.int32)
    reseed = torch.arange(max_iter * K, dtype=torch.int32) % T
    C, labels, wsum, info = ops.kmeans_fit(X, K, init, reseed, max_iter=max_iter)
    ref = oracle.kmeans_fit(X.cpu().numpy(), K, init.numpy(), reseed.numpy(), max_iter=max_iter)
    assert np.array_equal(labels.cpu().numpy(), ref["labels"])
    assert int(info[0]) == ref["iters"]
    assert np.array_equal(wsum.cpu().numpy(), ref["wsum"])
    assert np.array_equal(C.cpu().numpy(), ref["centroids"])


@pytest.mark.parametrize("T,K,D,dup", [(400, 5, 2048 * 24, False), (64, 8, 2048 * 9 + 64, False), (90, 5, 2048 * 8, True), (200, 3, 2048 * 5, False)])
def test_unit_weights_equal_no_weights(T, K, D, dup):
    """`weighted_kmeans_feature` without weights means weights of one (reference utiles.py:292-293).  The kernels' unweighted path must be that
    arithmetic bit for bit (1 * x = x, a sum of ones is the count, `W > 0` is `count > 0`): centroids, labels, cluster weights, exit iteration
    and reseed count of `weights=ones` and `weights=None` are identical - which is what lets the host skip the weight vector for the default call."""
    from streamchat_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    centres = torch.randn(6, D, device="cuda", generator=g)
    X = (centres[torch.randint(0, 6, (T,), device="cuda", generator=g)] + 0.6 * torch.randn(T, D, device="cuda", generator=g)).half()
    init = list(range(0, T, T // K))[:K]
    if dup:
        X[init[1]] = X[init[0]]
    rs = [7, 3, 11, 5] * 10
    a = ops.kmeans_fit(X, K, init, rs, weights=None, max_iter=10, tol=1e-4)
    b = ops.kmeans_fit(X, K, init, rs, weights=torch.ones(T, device="cuda"), max_iter=10, tol=1e-4)
    assert torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    assert torch.equal(a[0].view(torch.int32), b[0].view(torch.int32)) and torch.equal(a[2].view(torch.int32), b[2].view(torch.int32))
    if dup:
        assert int(a[3][2]) > 0
