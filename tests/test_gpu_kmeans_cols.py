"""sc_kmeans_fit_cols (ABI 8): the Lloyd loop data-parallel over COLUMNS.  Each "rank" holds the columns of whole SC-KM2 segments; per iteration
only the two fp64 segment tables (distances [32, T K], shifts [32, K]) are exchanged.  Claim under test: labels, cluster weights, exit iteration,
reseed count and every rank's columns of the centroids are BIT-IDENTICAL to sc_kmeans_fit on the whole matrix, at any rank count - here the ranks
are threads of one process on one GPU (each its own stream and workspace; the exchange is a barrier + row copies between the ranks' tables), the
multi-process form over torch.distributed is tests/test_gpu_sharded.py::test_dp_lloyd_*."""
import threading

import pytest
import torch

from streamchat_amd import ops

pytestmark = pytest.mark.gpu


def _fit_cols_threads(X, K, init, reseed, w, world, max_iter=10):
    T, D = X.shape
    seg_groups, slabs = ops.kmeans_column_slabs(D, world)
    barrier = threading.Barrier(world, timeout=120)
    tables, results, errors = {}, [None] * world, []

    def exchange_of(r):
        def exchange(what, table):
            torch.cuda.current_stream().synchronize()                   # this rank's rows are written
            tables[(r, what)] = table
            barrier.wait()
            for q, (s0, c, _, _) in enumerate(slabs):
                if q != r:
                    table[s0:s0 + c].copy_(tables[(q, what)][s0:s0 + c])
            torch.cuda.current_stream().synchronize()
            barrier.wait()                                              # nobody overwrites its rows before everybody has read them
        return exchange

    def worker(r):
        try:
            s0, c, lo, hi = slabs[r]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                results[r] = ops.kmeans_fit_cols(X[:, lo:hi].contiguous(), K, init, reseed, seg_groups, s0, c, exchange_of(r), weights=w, max_iter=max_iter)
                st.synchronize()
        except BaseException as e:      # noqa: BLE001
            errors.append(e)
            barrier.abort()

    X.record_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise errors[0]
    return slabs, results


CASES = {
    # name: (T, K, D, dtype, weighted, reseed, world)
    "merge_full_size_w8": (400, 5, 576 * 3584, torch.float16, False, False, 8),
    "merge_full_size_w3": (400, 5, 576 * 3584, torch.float16, True, False, 3),
    "c1_shape_w2": (64, 8, 2048 * 70, torch.float16, False, False, 2),
    "ragged_tail_f32_w4": (150, 6, 2048 * 67 + 520, torch.float32, True, False, 4),
    "bf16_w5": (130, 4, 2048 * 40, torch.bfloat16, False, False, 5),
    "empty_cluster_w4": (90, 5, 2048 * 33, torch.float16, False, True, 4),
}


@pytest.mark.parametrize("name", list(CASES))
def test_column_sharded_fit_is_bit_identical_to_the_one_gpu_fit(name):
    T, K, D, dtype, weighted, reseed, world = CASES[name]
    g = torch.Generator(device="cuda").manual_seed(11)
    centres = torch.randn(6, D, device="cuda", generator=g)
    X = (centres[torch.randint(0, 6, (T,), device="cuda", generator=g)] + 0.6 * torch.randn(T, D, device="cuda", generator=g)).to(dtype)
    del centres
    init = list(range(0, T, T // K))[:K]
    if reseed:
        X[init[1]] = X[init[0]]
    w = (0.5 + torch.rand(T, device="cuda", generator=g)) if weighted else None
    rs = [7, 3, 11, 5] * 10
    C, labels, wsum, info = ops.kmeans_fit(X, K, init, rs, weights=w, max_iter=10, tol=1e-4)
    slabs, res = _fit_cols_threads(X, K, init, rs, w, world)
    if reseed:
        assert int(info[2]) > 0, "the reseed case did not consume a reseed row"
    for r, ((s0, c, lo, hi), (Cr, lr, wr, ir)) in enumerate(zip(slabs, res)):
        assert torch.equal(lr, labels), f"rank {r}: labels"
        assert torch.equal(wr.view(torch.int32), wsum.view(torch.int32)), f"rank {r}: cluster weights"
        assert torch.equal(ir, info), f"rank {r}: info {ir.tolist()} != {info.tolist()}"
        assert torch.equal(Cr.view(torch.int32), C[:, lo:hi].contiguous().view(torch.int32)), f"rank {r}: centroid columns [{lo}, {hi})"


def test_column_slabs_and_argument_checks():
    sg, slabs = ops.kmeans_column_slabs(576 * 3584, 8)
    assert sg == 32 and slabs[0] == (0, 4, 0, 262144) and slabs[-1] == (28, 4, 1835008, 2064384)
    assert sum(c for _, c, _, _ in slabs) == 32 and all(a[3] == b[2] for a, b in zip(slabs, slabs[1:]))
    assert ops.kmeans_column_slabs(8192, 8) is None                     # 4 groups cannot feed 8 ranks
    X = torch.randn(16, 2048 * 4, device="cuda").half()
    with pytest.raises(ops.StreamChatHipError):                         # a middle slab must hold seg_count * seg_groups whole groups
        ops.kmeans_fit_cols(X, 2, [0, 1], None, 3, 0, 2, lambda what, t: None)
    with pytest.raises(RuntimeError, match="boom"):                     # an exception in the exchange surfaces, it does not unwind through C
        def bad(what, t):
            raise RuntimeError("boom")
        ops.kmeans_fit_cols(X, 2, [0, 1], None, 1, 28, 4, bad)
