"""Sharded memory update with SINGLE-STREAM semantics.

One process per GPU.  The frames of a segment are dealt to the ranks by whole chunks (`dist.partition_chunks`); every rank
encodes and captions its own chunks.  Everything the reference decides on the GLOBAL stream is decided here on the global stream
too, from metadata that is identical on every rank:

  * short-term memory: forgetting-curve sampling over the last `short_window` frames of the segment
    (/root/reference/inference_streaming_longva_v2.py:319-337) -> global frame numbers, same RNG draw on every rank;
  * long-term memory: ALL new depth-0 nodes are appended to the ONE tree, then at most ONE merge of the first `interval` siblings
    of the highest eligible depth (/root/reference/utiles.py:525-536,567-620 via `utiles.plan_merge`) — not one merge per rank;
  * the merge-group k-means runs ONCE, on the rank that owns the group's first row (the rows another rank holds are sent to it
    point to point), with the init rows every rank drew from the same CPU generator: labels / centroids are those of the 1-GPU run.
    With `dp_lloyd` (SC_DP_LLOYD=1, bench.py --dp-lloyd) it runs DATA-PARALLEL over columns instead (`_dp_lloyd`): every rank clusters the
    columns of whole SC-KM2 segments and only two small fp64 segment tables are all-gathered per iteration - the same bits at any N.

A tree node's `.centroids` is a `Ref` — a list of (owner rank, store key, row range) — instead of a tensor; tensors stay where they
were produced until `fetch` moves exactly the selected rows (short-memory frames + retrieved nodes) to the consumer: ONE
`all_gather_into_tensor` of a right-sized buffer (every rank derives the per-rank row counts from the same metadata, so there is
no size exchange and no capacity guess), after which rank 0 owns the [short | long] block for the single-GPU 7B prefill.
Frame ranges are never exchanged: they follow from `partition_chunks`.  With world size 1 every `fetch` is a view.

The retrieved-frame indices, tree shape and texts therefore do not depend on the number of GPUs
(tests/test_sharded_gloo.py: world 2 and 4 == the single-stream `updating_memory_buffer` on the same stream)."""
import os
import random

import numpy as np
import torch

from . import ops
from . import utiles as U
from .dist import all_gather_row_windows, all_gather_rows, broadcast_object, broadcast_tensor, exchange, gather_objects, partition_chunks

BANK, MERGE = 0, 1          # store key kinds: ("bank", segment) frame features / ("merge", node id) k-means centroids


class Ref:
    """Rows of a logical [rows, P, D] tensor scattered over ranks: pieces = ((owner, kind, id, lo, hi), ...) in row order."""
    __slots__ = ("pieces",)

    def __init__(self, pieces):
        self.pieces = tuple(tuple(int(x) for x in p) for p in pieces if p[4] > p[3])

    @property
    def rows(self):
        return sum(p[4] - p[3] for p in self.pieces)

    @property
    def shape(self):                        # the builder's `combined_centroids.shape[0] > num_clusters` test (utiles.py:586)
        return (self.rows,)

    @staticmethod
    def concat(refs):
        return Ref([p for r in refs for p in r.pieces])

    def __eq__(self, o):
        return isinstance(o, Ref) and self.pieces == o.pieces

    def __repr__(self):
        return f"Ref{self.pieces}"


class ShardedMemory:
    def __init__(self, ctx, chunk_size=30, num_clusters=5, interval=10, short_window=20, remember_window=5, tau=5, always_collective=False,
                 dp_lloyd=None):
        self.ctx = ctx
        # dp_lloyd: run the merge group's k-means DATA-PARALLEL over columns (`_dp_lloyd` below) instead of shipping the group's rows to
        # one rank.  Same labels and centroids, bit for bit (sc_kmeans_fit_cols).  Default: SC_DP_LLOYD=1 switches it on - off until the
        # transport has been seen on a multi-GPU box (the owner-rank path is the one the full-size 8-process jobs have run since round 4)
        self.dp_lloyd = (os.environ.get("SC_DP_LLOYD", "0") == "1") if dp_lloyd is None else bool(dp_lloyd)
        # always_collective: take the collective code paths even where a shortcut exists (world size 1, rows already on the consumer) —
        # lets a single-rank RCCL process group exercise exactly the calls an N-rank run makes (tests/test_gpu_sharded.py)
        self.always_collective = always_collective
        self.chunk_size, self.num_clusters, self.interval = chunk_size, num_clusters, interval
        self.short_window, self.remember_window, self.tau = short_window, remember_window, tau
        self.store = {}             # (kind, id) -> local tensor [rows, P, D]
        self.seg_parts = []         # per segment: [(start, end)] per rank (segment-local frame numbers)
        self.tree = None
        self.next_node = 0
        self.row_shape = None       # (P, D), dtype, device of a feature row
        self._send = self._recv = None
        self.traffic = dict(fetches=0, bytes_moved=0, dp_lloyd_fits=0, dp_lloyd_bytes=0)   # row bytes that crossed ranks in fetch() / _dp_lloyd() (computed from the Refs: the same number on every rank)
        self.kmeans_max_iter = 10   # weighted_kmeans_feature's default (utiles.py:291): fixes how many reseed rows a merge draws

    # ---------------------------------------------------------------------------------------------
    def partition(self, n_frames):
        return partition_chunks(n_frames, self.chunk_size, self.ctx.world)

    def frame_ref(self, seg, a, b):
        """frames [a, b) of segment `seg` (segment-local numbering) -> Ref (split at rank boundaries)."""
        out = []
        for r, (x, y) in enumerate(self.seg_parts[seg]):
            lo, hi = max(a, x), min(b, y)
            if hi > lo:
                out.append((r, BANK, seg, lo - x, hi - x))
        return Ref(out)

    def frames_of(self, ref):
        """global description of a Ref's frame rows: [(segment, frame)] for BANK pieces, (MERGE, node, row) otherwise (tests / logs)."""
        out = []
        for (r, kind, i, lo, hi) in ref.pieces:
            base = self.seg_parts[i][r][0] if kind == BANK else 0
            out += [("frame", i, base + f) if kind == BANK else ("centroid", i, f) for f in range(lo, hi)]
        return out

    # ---------------------------------------------------------------------------------------------
    def update(self, local_feats, n_frames, summarizer_model, summarizer_tokenzier, rng=None):
        """One memory update over a new segment of `n_frames` frames (the sharded `updating_memory_buffer`,
        inference_streaming_longva_v2.py:267-378).  `local_feats` [n_local, P, D]: this rank's frames, i.e. frames
        `self.partition(n_frames)[rank]` of the segment.  Returns (tree, short) with Refs in place of tensors."""
        from .streaming import _captioning_ids
        ctx, cs = self.ctx, self.chunk_size
        parts = self.partition(n_frames)
        a, b = parts[ctx.rank]
        if local_feats.shape[0] != b - a:
            raise ValueError(f"rank {ctx.rank} owns frames [{a}, {b}) of this segment but was given {local_feats.shape[0]} rows")
        seg = len(self.seg_parts)
        self.seg_parts.append(parts)
        self.store[(BANK, seg)] = local_feats
        if self.row_shape is None:
            self.row_shape = (tuple(local_feats.shape[1:]), local_feats.dtype, local_feats.device)

        # ---- short-term memory: the same draw on every rank, over global frame numbers (:319-337) ----
        window = min(self.short_window, n_frames)
        fifo = list(range(n_frames - window, n_frames))
        probs = U.calculate_forgetting_probabilities(window, tau=self.tau)
        short = [self.frame_ref(seg, f, f + 1) for f in
                 U.select_data_without_replacement(fifo, probs, min(self.remember_window, window), rng=rng)]

        # ---- captions of the local chunks, exchanged as text (once per update) ----
        ids = _captioning_ids(summarizer_model, summarizer_tokenzier)
        mine = [U.caption_chunk(summarizer_model, summarizer_tokenzier, ids, local_feats[s:s + cs]) for s in range(0, b - a, cs)]
        captions = [c for part in gather_objects(ctx, mine) for c in part]
        n_chunks = (n_frames + cs - 1) // cs
        assert len(captions) == n_chunks, (len(captions), n_chunks)
        nodes = [U.MultimodalTreeNode(self.frame_ref(seg, c * cs, min((c + 1) * cs, n_frames)), captions[c], depth=0) for c in range(n_chunks)]
        if self.tree:
            nodes = self.tree + nodes

        # ---- at most ONE merge on the global node list (utiles.py:567-620) ----
        start = U.plan_merge(nodes, self.interval)
        if start is not None:
            group = nodes[start:start + self.interval]
            combined = Ref.concat([n.centroids for n in group])
            executor = combined.pieces[0][0]
            if combined.rows > self.num_clusters:
                # every rank draws the init rows AND the empty-cluster reseed rows (keeps the CPU torch generator and Python's global
                # `random` state in lockstep with the 1-GPU run and with each other — the executor changes from merge to merge, so a
                # draw taken only there would let the ranks' states drift apart); only the executor clusters
                init_idx = torch.randperm(combined.rows)[:self.num_clusters]
                reseed_idx = [random.randint(0, combined.rows - 1) for _ in range(self.kmeans_max_iter * self.num_clusters)]
                node_id = self.next_node
                self.next_node += 1
                (P, D), _, _ = self.row_shape
                slabs = ops.kmeans_column_slabs(P * D, ctx.world) if (self.dp_lloyd and (ctx.world > 1 or self.always_collective)) else None
                if slabs is not None:
                    new_centroids = self._dp_lloyd(combined, executor, init_idx, reseed_idx, slabs)
                    if ctx.rank == executor:
                        self.store[(MERGE, node_id)] = new_centroids
                else:
                    X = self.fetch([combined], dst=executor, mode="p2p")
                    if ctx.rank == executor:
                        new_centroids, _ = U.weighted_kmeans_feature(X, self.num_clusters, init_idx=init_idx, reseed_idx=reseed_idx,
                                                                      max_iter=self.kmeans_max_iter)
                        self.store[(MERGE, node_id)] = new_centroids
                new_ref = Ref([(executor, MERGE, node_id, 0, self.num_clusters)])
            else:
                new_ref = combined
            text = None
            if ctx.rank == executor:
                text = U.summarize_captions(summarizer_model, summarizer_tokenzier, [n.text for n in group])
            text = broadcast_object(ctx, text, src=executor)
            new_node = U.MultimodalTreeNode(new_ref, text, depth=group[0].depth + 1)
            new_node.children.extend(group)
            nodes[start:start + self.interval] = [new_node]
        self.tree = nodes
        return nodes, short

    # ---------------------------------------------------------------------------------------------
    def _dp_lloyd(self, combined, executor, init_idx, reseed_idx, slabs):
        """The merge group's k-means data-parallel over COLUMNS (`north_star`: "k-means data-parallel"; reference utiles.py:294-318 on one
        device).  Rank q takes the columns of whole SC-KM2 segments (`ops.kmeans_column_slabs`); three steps:
          1. transpose: every owner of rows of the group sends rank q the columns of q's slab (one batch of point-to-point pieces - the
             bytes of the gather-to-executor, but spread over all links instead of into one rank's);
          2. `ops.kmeans_fit_cols` on the [T, D_q] slab: per iteration the ranks exchange their rows of two fp64 segment tables
             (all-gather, 32 x T K and 32 x K values); arg-min / ordering / convergence run replicated on identical inputs;
          3. the K centroid rows (cast to the feature dtype per slab, as the 1-GPU path casts the whole rows) go to the executor.
        Returns the [K, P, D] centroids on the executor (None elsewhere) - the bits `weighted_kmeans_feature` returns on one GPU."""
        ctx, K = self.ctx, self.num_clusters
        (P, D), dtype, dev = self.row_shape
        seg_groups, cols = slabs
        T = combined.rows
        s0, cnt, lo, hi = cols[ctx.rank]
        esz = torch.empty((), dtype=dtype).element_size()
        X = torch.empty((T, hi - lo), dtype=dtype, device=dev)
        sends, recvs, off, moved = [], [], 0, 0
        for p in combined.pieces:
            n = p[4] - p[3]
            if p[0] == ctx.rank:
                rows = self._local(p).reshape(n, P * D)
                X[off:off + n].copy_(rows[:, lo:hi])
                sends += [(rows[:, cols[q][2]:cols[q][3]], q) for q in range(ctx.world) if q != ctx.rank]       # (exchange() makes them contiguous)
            else:
                recvs.append((X[off:off + n], p[0]))
            moved += n * (P * D - (cols[p[0]][3] - cols[p[0]][2])) * esz
            off += n
        exchange(ctx, sends, recvs)
        windows = [(c[0], c[1]) for c in cols]
        C, _, _, _ = ops.kmeans_fit_cols(X, K, init_idx, reseed_idx, seg_groups, s0, cnt,
                                         lambda what, table: all_gather_row_windows(ctx, table, windows),
                                         weights=None, max_iter=self.kmeans_max_iter, tol=1e-4)      # (unit weights, as weighted_kmeans_feature passes them on one GPU)
        del X
        mine = C.to(dtype)
        sends, recvs, out, tmp = [], [], None, {}
        if ctx.rank == executor:
            out = torch.empty((K, P * D), dtype=dtype, device=dev)
            out[:, lo:hi].copy_(mine)
            for q in range(ctx.world):
                if q != executor:
                    tmp[q] = torch.empty((K, cols[q][3] - cols[q][2]), dtype=dtype, device=dev)
                    recvs.append((tmp[q], q))
        else:
            sends.append((mine, executor))
        exchange(ctx, sends, recvs)
        self.traffic["dp_lloyd_fits"] += 1
        # bytes received, summed over the ranks: the transposed rows, the centroid slabs, max_iter all-gathers of the two segment tables
        self.traffic["dp_lloyd_bytes"] += moved + K * (P * D - (cols[executor][3] - cols[executor][2])) * esz \
            + self.kmeans_max_iter * (ctx.world - 1) * ops.KM_SEGMENTS * (T * K + K) * 8
        if ctx.rank != executor:
            return None
        for q, t in tmp.items():
            out[:, cols[q][2]:cols[q][3]].copy_(t)
        return out.view(K, P, D)

    def _local(self, piece):
        _, kind, i, lo, hi = piece
        return self.store[(kind, i)][lo:hi]

    def fetch(self, refs, dst=0, mode="gather"):
        """Rows of `refs` (in order) as ONE [rows, P, D] tensor on rank `dst` (None elsewhere).  `refs` must be identical on every
        rank.
        mode "gather" (default; "p2p" is the same path): gather-to-root - every owner sends exactly its pieces straight into their final
        position of dst's [rows] buffer, as ONE batch of point-to-point transfers (RCCL: one group, each peer over its own xGMI link).
        Only dst allocates, and only what it consumes: bytes moved = the rows dst lacks (reference inference_streaming_longva_v2.py:
        696-699 hands the selected rows to ONE generate call).  Round 4 used the all-gather for the selected rows of a question: every
        rank then held world x max-rows-per-rank rows although only rank 0 reads them (C4: 8 x 85 frames x 4.13 MB = 2.8 GB of buffers
        and 7 x the traffic for 351 MB of payload).
        mode "allgather": one all_gather_into_tensor of max-rows-per-rank slots - kept for a consumer on EVERY rank (none today)."""
        ctx = self.ctx
        pieces = [p for r in refs for p in r.pieces]
        if ctx.world == 1 and not self.always_collective:
            return U.cat_frames([self._local(p) for p in pieces])                    # adjacent bank rows: a view, no copy
        (P, D), dtype, dev = self.row_shape
        total = sum(p[4] - p[3] for p in pieces)
        if mode == "gather" and ctx.world == 1:
            mode = "allgather"            # (always_collective at world 1, the 1-rank RCCL test: there is no peer to send to; make the collective call)
        if all(p[0] == dst for p in pieces) and not (self.always_collective and mode == "allgather"):   # nothing to move (e.g. C4's first ten chunks)
            return U.cat_frames([self._local(p) for p in pieces]) if ctx.rank == dst else None
        row_bytes = P * D * torch.empty((), dtype=dtype).element_size()
        self.traffic["fetches"] += 1
        if mode in ("gather", "p2p"):
            self.traffic["bytes_moved"] += row_bytes * sum(p[4] - p[3] for p in pieces if p[0] != dst)      # (identical on every rank: from the Refs)
            sends, recvs, out, off = [], [], None, 0
            if ctx.rank == dst:
                out = torch.empty((total, P, D), dtype=dtype, device=dev)
            for p in pieces:
                n = p[4] - p[3]
                if ctx.rank == dst:
                    if p[0] == dst:
                        out[off:off + n].copy_(self._local(p))
                    else:
                        recvs.append((out[off:off + n], p[0]))
                elif p[0] == ctx.rank:
                    sends.append((self._local(p), dst))
                off += n
            exchange(ctx, sends, recvs)
            return out
        # ---- all-gather: rank r packs its pieces (in order) into slot r of a [world, cap] buffer ----
        counts = [0] * ctx.world
        for p in pieces:
            counts[p[0]] += p[4] - p[3]
        cap = max(counts)
        self.traffic["bytes_moved"] += row_bytes * cap * ctx.world * (ctx.world - 1)            # every rank receives the other ranks' slots
        if self._send is None or self._send.shape[0] < cap:
            self._send = torch.empty((cap, P, D), dtype=dtype, device=dev)
            self._recv = torch.empty((ctx.world * cap, P, D), dtype=dtype, device=dev)
        send, recv = self._send[:cap], self._recv[:ctx.world * cap]
        k = 0
        for p in pieces:
            if p[0] == ctx.rank:
                n = p[4] - p[3]
                send[k:k + n].copy_(self._local(p))
                k += n
        all_gather_rows(ctx, recv, send)                         # RCCL: every peer pushes its slot over its own xGMI link
        if ctx.rank != dst:
            return None
        slot, views = [0] * ctx.world, []
        for p in pieces:
            n = p[4] - p[3]
            views.append(recv[p[0] * cap + slot[p[0]]: p[0] * cap + slot[p[0]] + n])
            slot[p[0]] += n
        # always a fresh tensor: `recv` is reused by the next fetch, so an alias of it (one piece -> one view) would be overwritten
        # under a caller that holds two results
        return views[0].clone() if len(views) == 1 else torch.cat(views, dim=0)

    # ---------------------------------------------------------------------------------------------
    def broadcast_refs(self, refs, src=0, capacity=None):
        """The root's retrieval decision (a list of Refs) to every rank as ONE small int64 tensor broadcast (no pickling):
        rows = (ref index, owner, kind, id, lo, hi); preceded by a one-element broadcast of the row count."""
        ctx = self.ctx
        if ctx.world == 1 and not self.always_collective:
            return refs
        dev = self.row_shape[2]
        # the piece count travels first, so the table is sized by what the root actually selected (no capacity guess) and a bad
        # selection raises on EVERY rank after the collective instead of leaving the others waiting inside it
        rows = [(i,) + p for i, r in enumerate(refs) for p in r.pieces] if ctx.rank == src else []
        n = torch.tensor([len(rows)], dtype=torch.int64).to(dev)
        broadcast_tensor(ctx, n, src=src)
        n = int(n.item())
        if capacity is not None and n > capacity:
            raise ValueError(f"{n} pieces exceed the broadcast capacity {capacity}")
        if n == 0:
            return []
        buf = torch.tensor(rows, dtype=torch.int64).reshape(n, 6) if ctx.rank == src else torch.empty((n, 6), dtype=torch.int64)
        buf = buf.to(dev)
        broadcast_tensor(ctx, buf, src=src)
        rows = buf.cpu().tolist()
        out = {}
        for i, *p in rows:
            if i >= 0:
                out.setdefault(i, []).append(tuple(p))
        return [Ref(out[i]) for i in sorted(out)]
