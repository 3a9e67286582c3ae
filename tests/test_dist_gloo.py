"""CPU, world_size = 2, gloo: the multi-GPU primitives (chunk partition, text collectives, the two row-movement modes of
sharded.ShardedMemory.fetch with provenance-carrying features).  The whole sharded control flow is in test_sharded_gloo.py."""
import os
import socket

import torch
import torch.multiprocessing as mp

from streamchat_amd import dist as D


def test_partition_chunks_covers_stream_in_order():
    for n, c, w in [(1024, 40, 1), (1024, 40, 2), (4096, 40, 8), (8192, 40, 8), (50, 40, 4), (400, 40, 3)]:
        parts = D.partition_chunks(n, c, w)
        assert parts[0][0] == 0 and parts[-1][1] == n
        for (a, b), (a2, b2) in zip(parts, parts[1:]):
            assert b == a2 and a <= b
        assert all(a % c == 0 for a, _ in parts)                      # whole chunks only
        assert all(D.owner_of(f, parts) == next(i for i, (a, b) in enumerate(parts) if a <= f < b) for f in (0, n // 2, n - 1))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from streamchat_amd import sharded as SH
    ctx = D.init_from_env("cpu")
    n, chunk, P, Dm = 200, 40, 3, 8
    mem = SH.ShardedMemory(ctx, chunk_size=chunk)
    parts = mem.partition(n)
    a, b = parts[rank]
    # feature of global frame f = f everywhere, so the consumer can check provenance
    bank = torch.arange(a, b, dtype=torch.float32).view(-1, 1, 1).expand(b - a, P, Dm).contiguous()
    mem.seg_parts.append(parts)
    mem.store[(SH.BANK, 0)] = bank
    mem.row_shape = ((P, Dm), bank.dtype, bank.device)
    metas = D.gather_objects(ctx, dict(rank=rank, captions=[f"clip {c}" for c in range(a // chunk, (b + chunk - 1) // chunk)]))
    ok = [m["rank"] for m in metas] == list(range(world)) and [c for m in metas for c in m["captions"]] == [f"clip {c}" for c in range(5)]
    wanted_frames = [199, 3, 120, 121, 40, 0]
    refs = [mem.frame_ref(0, f, f + 1) for f in wanted_frames] + [mem.frame_ref(0, 70, 130)]        # the last one straddles both ranks
    refs = mem.broadcast_refs(refs if rank == 0 else None)
    expect = [float(f) for f in wanted_frames] + [float(f) for f in range(70, 130)]
    row_bytes, moved = P * Dm * 4, {}
    for mode in ("allgather", "p2p", "gather"):
        for dst in (0, 1):
            before = mem.traffic["bytes_moved"]
            got = mem.fetch(refs, dst=dst, mode=mode)
            moved[(mode, dst)] = mem.traffic["bytes_moved"] - before
            if rank == dst:
                ok = ok and got.shape == (len(expect), P, Dm) and got[:, 0, 0].tolist() == expect and bool((got == got[:, :1, :1]).all())
            else:
                ok = ok and got is None
    # gather-to-root moves exactly the rows dst lacks (for dst 0: the frames rank 1 owns, i.e. >= parts[1][0], incl. the tail of the
    # straddling ref); the all-gather moves world x cap slots to every rank
    lacks0 = sum(1 for f in expect if f >= parts[1][0])
    ok = ok and moved[("gather", 0)] == lacks0 * row_bytes and moved[("gather", 1)] == (len(expect) - lacks0) * row_bytes
    ok = ok and moved[("p2p", 0)] == moved[("gather", 0)] and moved[("allgather", 0)] == max(lacks0, len(expect) - lacks0) * 2 * row_bytes
    only0 = mem.fetch([mem.frame_ref(0, 0, 80)], dst=0)                    # rows already on dst: no collective, a view of the bank
    ok = ok and ((only0.data_ptr() == bank.data_ptr()) if rank == 0 else only0 is None)
    ok = ok and D.broadcast_object(ctx, "summary" if rank == 1 else None, src=1) == "summary"
    q.put((rank, ok))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_fetch_modes_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert [r[1] for r in res] == [True, True]
