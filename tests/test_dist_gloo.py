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


# ---- round 6: explicit time-out + communicator warm-up ----
def _warm_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ctx = D.init_from_env("cpu", timeout_s=60)
    q.put((rank, D.warm_up(ctx)))
    torch.distributed.destroy_process_group()


def test_warm_up_creates_every_pair_at_world_4():
    world, port = 4, _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    ps = [mpc.Process(target=_warm_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    got = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert all(got[r] == dict(world=4, all_gather=True, p2p_peers=3, backend="gloo") for r in range(world))


def _lonely_worker(rank, world, port, q):
    import time
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    t0 = time.time()
    try:
        D.init_from_env("cpu", timeout_s=4)            # rank `world - 1` never starts
        q.put((rank, "joined", time.time() - t0))
    except Exception as e:                              # noqa: BLE001
        q.put((rank, type(e).__name__, time.time() - t0))


def test_a_rank_that_never_arrives_raises_on_the_others_within_the_timeout():
    world, port = 3, _free_port()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    ps = [mpc.Process(target=_lonely_worker, args=(r, world, port, q)) for r in range(world - 1)]      # the last rank is missing
    [p.start() for p in ps]
    got = [q.get(timeout=90) for _ in range(world - 1)]
    [p.join(30) for p in ps]
    for rank, what, secs in got:
        assert what != "joined", f"rank {rank} believes a 3-rank group formed with 2 ranks"
        assert secs < 60, f"rank {rank} waited {secs:.0f} s for a 4 s time-out"


def test_bench_preflight_reports_one_json_error_line_without_a_gpu():
    """`bench.py --preflight` on a box where the check fails (no GPU here): ONE JSON line, exit code 3 - what a launcher script can act on."""
    import json, subprocess, sys
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the failing leg needs a box without a GPU; the passing legs are in test_gpu_sharded.py")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--preflight"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 3
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["preflight"] == "error" and "GPU" in rec["error"]
