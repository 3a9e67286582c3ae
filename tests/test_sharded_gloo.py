"""CPU, gloo, world size 2, 3, 4 and 8: the WHOLE sharded memory-update / retrieval control flow (streamchat_amd/sharded.py) against
the single-stream functions (`streaming.updating_memory_buffer` + `utiles.fast_search_tree_multi_modal_with_embedding`) on the
same global stream: tree shape and texts, merge centroids, short-memory frames, retrieved ("wanted") frames and the fetched
feature rows must not depend on the number of ranks (reference policy utiles.py:525-536,567-620,
chunking inference_streaming_longva_v2.py:346-358).

The k-means / top-k providers are swapped for the ORACLE (test infrastructure) in BOTH runs so that the control flow can run
without a GPU; tests/test_gpu_sharded.py runs the HIP path through the same code at world size 1."""
import os
import socket
import types
import zlib

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

P_, D_ = 3, 8
MEM = dict(chunk_size=4, num_clusters=2, interval=3, short_window=6, remember_window=3, tau=5)
SEGMENTS = [26, 3, 13, 24, 9]            # frames per update: straddling merges, a segment smaller than the world, a depth-1 merge
QUESTION = "where is the red cup"


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


DUPLICATES = False                       # scenario switch (set per worker): frames that are exact copies -> empty clusters -> reseeds
DP_LLOYD = False                         # scenario switch (set per worker): ShardedMemory(dp_lloyd=True) - the merge k-means data-parallel over columns


def _segment(seg, n):
    g = torch.Generator().manual_seed(1000 + seg)
    if DUPLICATES:      # two distinct frame values per segment, the rare one on every 5th frame: two init rows are usually the SAME
                        # value, the second cluster comes up empty and is re-seeded from `reseed_idx` (utiles.py:312-313)
        v = torch.where(torch.arange(n) % 5 == 4, 7.0, 0.0).view(n, 1, 1) + 100.0 * seg
        return (v + torch.zeros(n, P_, D_)).contiguous()
    base = torch.arange(n, dtype=torch.float32).view(n, 1, 1) // MEM["chunk_size"] * 3.0 + 100.0 * seg
    return (base + 0.05 * torch.randn(n, P_, D_, generator=g)).contiguous()


class Tok:                                   # summarizer-side tokenizer: ids = word hashes, decode = text of the "generated" hash
    bos_token_id = None

    def __call__(self, text, **kw):
        return types.SimpleNamespace(input_ids=[zlib.crc32(w.encode()) % 30000 for w in text.split()])

    def batch_decode(self, ids, skip_special_tokens=True):
        return [f" caption#{int(ids[0][0])} "]


class Summarizer:                            # deterministic in its INPUT (content hash), like synthetic.SyntheticCaptioner
    device = "cpu"
    config = types.SimpleNamespace(mm_use_im_start_end=False)

    def generate_with_image_embedding(self, ids, image_embeddings=None, **kw):
        key = (image_embeddings[0].reshape(-1)[:8] if image_embeddings is not None else torch.as_tensor(ids).reshape(-1).float()).numpy().tobytes()
        return torch.tensor([[zlib.crc32(key) % 100003]])


class EmbTok:
    def __call__(self, text, padding=True, return_tensors="pt"):
        texts = [text] if isinstance(text, str) else list(text)
        rows = [[zlib.crc32(t.encode()) % 9973 + 1] for t in texts]
        return {"input_ids": torch.tensor(rows)}


class EmbModel:                               # CLS embedding = seeded pseudo-random vector of the text hash
    def __call__(self, input_ids=None, **kw):
        out = torch.stack([torch.randn(1, 16, generator=torch.Generator().manual_seed(int(i))) for i in input_ids[:, 0]])
        return types.SimpleNamespace(last_hidden_state=out)


RESEEDS = []                             # per k-means call of this process: number of passes with an empty cluster


def _patch_providers():
    import oracle
    from streamchat_amd import ops, utiles as U

    def kmeans_feature(img_feature, K, weights=None, *, init_idx=None, reseed_idx=None, max_iter=10, **kw):
        import random
        T, P, D = img_feature.shape
        if init_idx is None:
            init_idx = torch.randperm(T)[:K]
        if reseed_idx is None:              # the product's default (utiles.weighted_kmeans_feature): Python's GLOBAL `random`, advanced
            reseed_idx = [random.randint(0, T - 1) for _ in range(max_iter * K)]
        r = oracle.kmeans_fit(img_feature.reshape(T, -1).numpy(), K, np.asarray(init_idx, np.int32), np.asarray(reseed_idx, np.int32),
                              max_iter=max_iter, trace=True)
        RESEEDS.append(int(sum(len(set(t.tolist())) < K for t in r["trace"])))      # assignment passes that left a cluster empty
        return torch.from_numpy(r["centroids"]).view(K, P, D), torch.from_numpy(r["labels"])

    def sim_topk(q, docs, k=1, metric="cos"):
        idx, sc = oracle.topk(q.numpy(), docs.numpy(), k, metric)
        return torch.from_numpy(idx), torch.from_numpy(sc)
    def kmeans_fit_cols(X, K, init_idx, reseed_idx, seg_groups, seg_first, seg_count, exchange, weights=None, max_iter=10, tol=1e-4):
        """CPU stand-in for the column-sharded fit: (1) drives the caller's exchange once per table with recognisable rows and checks that every
        rank's window arrived; (2) rebuilds the whole matrix from the ranks' slabs and returns this rank's columns of the ORACLE's centroids."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        sg, cols = ops.kmeans_column_slabs(P_ * D_, world)
        assert (sg, cols[rank][0], cols[rank][1], cols[rank][3] - cols[rank][2]) == (seg_groups, seg_first, seg_count, X.shape[1])
        T = X.shape[0]
        for what, width in ((0, T * K), (1, K)):
            table = torch.zeros(ops.KM_SEGMENTS, width, dtype=torch.float64)
            table[seg_first:seg_first + seg_count] = 1000.0 * (rank + 1) + torch.arange(seg_first, seg_first + seg_count, dtype=torch.float64).view(-1, 1)
            exchange(what, table)
            for q, (s0, c, _, _) in enumerate(cols):
                want = 1000.0 * (q + 1) + torch.arange(s0, s0 + c, dtype=torch.float64).view(-1, 1)
                assert torch.equal(table[s0:s0 + c], want.expand(c, width)), f"segment rows of rank {q} did not arrive on rank {rank}"
        parts = [None] * world
        dist.all_gather_object(parts, X.numpy())
        full = np.concatenate(parts, axis=1)
        r = oracle.kmeans_fit(full, K, np.asarray(init_idx, np.int32), np.asarray(reseed_idx, np.int32), max_iter=max_iter, trace=True)
        RESEEDS.append(int(sum(len(set(t.tolist())) < K for t in r["trace"])))
        lo, hi = cols[rank][2], cols[rank][3]
        return torch.from_numpy(np.ascontiguousarray(r["centroids"][:, lo:hi])), torch.from_numpy(r["labels"]), None, None
    U.weighted_kmeans_feature = kmeans_feature
    ops.sim_topk = sim_topk
    ops.kmeans_fit_cols = kmeans_fit_cols
    if DP_LLOYD:
        ops.KM_GROUP = 1                 # one column per "group": the 24-column test rows split into 24 non-empty segments


def _describe(nodes):
    def one(n):
        return dict(depth=n.depth, rows=int(n.centroids.shape[0]), text=n.text, children=[one(c) for c in n.children])
    return [one(n) for n in nodes]


def _single_stream():
    """the reference policy through the single-GPU functions: per update (tree description, short rows, path rows, path texts)"""
    import random
    from streamchat_amd import streaming as S, utiles as U
    torch.manual_seed(7)
    random.seed(5)
    rng = np.random.RandomState(11)
    tree, out = None, []
    for seg, n in enumerate(SEGMENTS):
        feats = _segment(seg, n)
        bank = [feats[i:i + 1] for i in range(n)]
        tree, short = S.updating_memory_buffer(bank, tree, Summarizer(), Tok(), True, rng=rng, **MEM)
        path, texts = U.fast_search_tree_multi_modal_with_embedding(tree, QUESTION, feats[0], EmbModel(), EmbTok(), cache=U.CaptionEmbeddingCache())
        merged = [n.centroids.clone() for n in tree if n.depth > 0]
        out.append(dict(tree=_describe(tree), short=torch.cat(short), path=torch.cat(path), texts=texts, merged=merged))
    return out


def _sharded(ctx):
    import random
    from streamchat_amd import sharded as SH, utiles as U
    torch.manual_seed(7)
    random.seed(5)
    rng = np.random.RandomState(11)
    mem = SH.ShardedMemory(ctx, dp_lloyd=DP_LLOYD, **MEM)
    out = []
    for seg, n in enumerate(SEGMENTS):
        a, b = mem.partition(n)[ctx.rank]
        local = _segment(seg, n)[a:b].contiguous()
        tree, short = mem.update(local, n, Summarizer(), Tok(), rng=rng)
        wanted, texts = None, None
        if ctx.is_root:                                   # retrieval runs on the root only; its decision is broadcast as metadata
            path, texts = U.fast_search_tree_multi_modal_with_embedding(tree, QUESTION, local, EmbModel(), EmbTok(), cache=U.CaptionEmbeddingCache())
            wanted = list(short) + list(path)
        wanted = mem.broadcast_refs(wanted)
        n_short = len(short)
        sel = mem.fetch(wanted, dst=0, mode="allgather")
        merged = [mem.fetch([nd.centroids], dst=0, mode="p2p") for nd in tree if nd.depth > 0]
        srows = sum(r.rows for r in wanted[:n_short])
        out.append(dict(tree=_describe(tree), short=None if sel is None else sel[:srows], path=None if sel is None else sel[srows:],
                        texts=texts, merged=merged, wanted=[mem.frames_of(r) for r in wanted]))
        if seg == len(SEGMENTS) - 1 and ctx.world > 1:
            # a fetched block must stay valid when the next fetch reuses the persistent receive buffer (single-piece results
            # used to be views of it)
            leafs = [nd.centroids for nd in tree if nd.depth == 0][-2:]
            x = mem.fetch([leafs[0]], dst=0, mode="allgather")
            keep = None if x is None else x.clone()
            mem.fetch([leafs[1]], dst=0, mode="allgather")
            assert x is None or torch.equal(x, keep), "an earlier fetch result was overwritten by the next fetch"
    if DP_LLOYD and ctx.world > 1:
        assert mem.traffic["dp_lloyd_fits"] > 0, "no merge took the data-parallel path"
    return out


def _worker(rank, world, port, q, duplicates=False, dp_lloyd=False):
    try:
        global DUPLICATES, DP_LLOYD
        DUPLICATES, DP_LLOYD = duplicates, dp_lloyd
        os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        from streamchat_amd import dist as D
        _patch_providers()
        ctx = D.init_from_env("cpu")
        import random
        ref = _single_stream()
        n_reseed_ref = sum(RESEEDS)
        probe_ref = (random.random(), float(torch.rand(1)))
        got = _sharded(ctx)
        probe = (random.random(), float(torch.rand(1)))
        ok, why = True, ""
        if probe != probe_ref:        # Python's `random` and the CPU torch generator end where the single-stream run leaves them, on EVERY rank
            ok, why = False, f"host RNG streams left the single-stream position: {probe} != {probe_ref}"
        if duplicates and n_reseed_ref == 0:
            ok, why = False, "the duplicate-frame stream met no empty cluster: the scenario does not test the reseed path"
        for u, (r, g) in enumerate(zip(ref, got)):
            if r["tree"] != g["tree"]:
                ok, why = False, f"update {u}: tree differs"
            if ctx.is_root:
                if not (torch.equal(r["short"], g["short"]) and torch.equal(r["path"], g["path"]) and r["texts"] == g["texts"]):
                    ok, why = False, f"update {u}: selected rows / texts differ"
                if len(r["merged"]) != len(g["merged"]) or not all(torch.equal(x, y) for x, y in zip(r["merged"], g["merged"])):
                    ok, why = False, f"update {u}: merge centroids differ"
        q.put((rank, ok, why, [g["wanted"] for g in got]))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:                                  # surface the failure instead of a queue timeout
        import traceback
        q.put((rank, False, traceback.format_exc(), None))
        raise


def _run_world(world, duplicates=False, dp_lloyd=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q, duplicates, dp_lloyd)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=300) for _ in ps)
    [p.join(timeout=60) for p in ps]
    for rank, ok, why, _ in res:
        assert ok, f"world {world} rank {rank}: {why}"
    assert all(r[3] == res[0][3] for r in res)               # every rank holds the same retrieval decision
    return res[0][3]


@pytest.mark.timeout(600)
def test_sharded_equals_single_stream_world2_and_world4():
    w2 = _run_world(2)
    w4 = _run_world(4)
    assert w2 == w4                                          # identical retrieved (segment, frame) indices for P = 2 and P = 4


@pytest.mark.timeout(900)
def test_sharded_equals_single_stream_world8():
    """World 8 - the size BASELINE.json's C4 / C5 are defined at.  Chunk 4 over 8 ranks: the 3-frame segment is ONE chunk
    (seven ranks own nothing of it), the 26-frame segment is 7 chunks (one rank idle, every other rank ONE chunk), so each merge group of
    `interval` = 3 sibling chunks straddles three ranks and the merged node's rows travel point to point from two of them; the 13- and
    9-frame segments leave four / five ranks without frames.  Same assertions as at world 2 / 4 (tree, texts, merge centroids, short and
    retrieved rows equal the single-stream functions on every update; host RNG streams in lockstep), and the retrieved (segment, frame)
    indices equal the world-4 run's."""
    from streamchat_amd import dist as D
    assert [b - a for a, b in D.partition_chunks(3, MEM["chunk_size"], 8)].count(0) == 7
    assert [b - a for a, b in D.partition_chunks(26, MEM["chunk_size"], 8)].count(0) == 1
    w8 = _run_world(8)
    w4 = _run_world(4)
    assert w8 == w4


@pytest.mark.timeout(600)
def test_sharded_empty_cluster_reseeds_stay_in_lockstep_with_single_stream():
    """ADVICE r02 (sharded.py): a stream of duplicate frames makes merge k-means calls come up with an empty cluster, whose reseed
    rows come from Python's global `random` (utiles.py:312-313).  Every rank must draw them (the executor changes from merge to
    merge), otherwise the ranks drift apart and the merge centroids differ from the single-stream run after the first merge."""
    _run_world(3, duplicates=True)


@pytest.mark.timeout(900)
def test_sharded_dp_lloyd_equals_single_stream_world2_3_and_8():
    """`ShardedMemory(dp_lloyd=True)`: the merge k-means data-parallel over COLUMNS (sharded._dp_lloyd).  Here the control flow on CPU - the
    transposing exchange of the group's rows into column slabs (uneven at world 3), the all-gather of the segment-table row windows, the
    centroid slabs back to the executor - with the oracle behind the fit; the kernels' bit-identity is tests/test_gpu_kmeans_cols.py and
    tests/test_gpu_sharded.py.  Same assertions as everywhere in this file: tree, texts, merge centroids, retrieved rows, host RNG streams."""
    w2 = _run_world(2, dp_lloyd=True)
    w3 = _run_world(3, dp_lloyd=True)
    assert w2 == w3 == _run_world(2)
    assert _run_world(8, dp_lloyd=True) == w2                # 24 one-column segments over 8 ranks: 3 each, the last rank also the empty tail
    _run_world(3, duplicates=True, dp_lloyd=True)


def test_world1_sharded_memory_is_views_and_equal():
    """world size 1: the sharded code path IS the single-stream path (no collective, fetch returns views of the bank)."""
    from streamchat_amd import dist as D, sharded as SH
    import streamchat_amd.utiles as U
    import streamchat_amd.ops as ops
    saved = (U.weighted_kmeans_feature, ops.sim_topk, ops.kmeans_fit_cols)
    try:
        _patch_providers()
        ref = _single_stream()
        got = _sharded(D.DistContext(0, 1, "cpu"))
        for r, g in zip(ref, got):
            assert r["tree"] == g["tree"] and r["texts"] == g["texts"]
            assert torch.equal(r["short"], g["short"]) and torch.equal(r["path"], g["path"])
            assert all(torch.equal(x, y) for x, y in zip(r["merged"], g["merged"]))
        mem = SH.ShardedMemory(D.DistContext(0, 1, "cpu"), **MEM)
        feats = _segment(0, 26)
        tree, short = mem.update(feats, 26, Summarizer(), Tok(), rng=np.random.RandomState(0))
        leaf = next(n for n in tree if n.depth == 0)
        assert mem.fetch([leaf.centroids]).data_ptr() == feats[mem.frames_of(leaf.centroids)[0][2]].data_ptr()
    finally:
        U.weighted_kmeans_feature, ops.sim_topk, ops.kmeans_fit_cols = saved


def test_ref_algebra_and_partition():
    from streamchat_amd import dist as D, sharded as SH
    mem = SH.ShardedMemory(D.DistContext(1, 3, "cpu"), chunk_size=4)
    mem.seg_parts.append(mem.partition(26))                # 7 chunks over 3 ranks: [0,8) [8,16) [16,26)
    assert mem.seg_parts[0] == [(0, 8), (8, 16), (16, 26)]
    r = mem.frame_ref(0, 6, 18)
    assert r.pieces == ((0, 0, 0, 6, 8), (1, 0, 0, 0, 8), (2, 0, 0, 0, 2)) and r.rows == 12 and r.shape == (12,)
    assert SH.Ref.concat([mem.frame_ref(0, 0, 2), mem.frame_ref(0, 20, 21)]).pieces == ((0, 0, 0, 0, 2), (2, 0, 0, 4, 5))
    assert mem.frames_of(mem.frame_ref(0, 15, 17)) == [("frame", 0, 15), ("frame", 0, 16)]
