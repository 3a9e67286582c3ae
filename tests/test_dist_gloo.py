"""CPU, world_size = 2, gloo: the multi-GPU layer (chunk partition, metadata gather, fixed-capacity feature all-gather)."""
import os
import socket

import pytest
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
    ctx = D.init_from_env("cpu")
    n, chunk, P, Dm = 200, 40, 3, 8
    parts = D.partition_chunks(n, chunk, world)
    a, b = parts[rank]
    # feature of global frame f = f everywhere, so the consumer can check provenance
    bank = torch.arange(a, b, dtype=torch.float32).view(-1, 1, 1).expand(b - a, P, Dm).contiguous()
    metas = D.gather_objects(ctx, dict(rank=rank, frames=(a, b), captions=[f"clip {c}" for c in range(a // chunk, (b + chunk - 1) // chunk)]))
    wanted = D.broadcast_object(ctx, [199, 3, 120, 121, 40, 0] if rank == 0 else None)
    got = D.gather_selected_frames(ctx, bank, (a, b), wanted, capacity=8)
    ok = got.shape == (len(wanted), P, Dm) and got[:, 0, 0].tolist() == [float(f) for f in wanted]
    ok = ok and [m["rank"] for m in metas] == list(range(world)) and metas[0]["frames"][0] == 0 and metas[-1]["frames"][1] == n
    try:
        D.gather_selected_frames(ctx, bank, (a, b), list(range(0, 9)), capacity=8)
        over = False
    except ValueError:
        over = True
    q.put((rank, ok, over))
    torch.distributed.destroy_process_group()


def test_gather_selected_frames_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert [r[1] for r in res] == [True, True]
    assert res[0][2] is True and res[1][2] is True                    # over-capacity request fails on EVERY rank, before the collective


def test_world1_is_identity():
    ctx = D.DistContext(0, 1, "cpu")
    bank = torch.arange(10, dtype=torch.float32).view(10, 1, 1).expand(10, 2, 4).contiguous()
    got = D.gather_selected_frames(ctx, bank, (0, 10), [7, 2], capacity=4)
    assert got[:, 0, 0].tolist() == [7.0, 2.0]
