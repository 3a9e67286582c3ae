"""Multi-GPU primitives: the frame stream shards by WHOLE CHUNKS across the GPUs of one node (one process per GPU,
`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).

What crosses the fabric: frame encode and chunk captioning are rank-local (data parallel, no data-path
collective); every decision the reference takes on the whole stream (short-memory draw, tree policy, the ONE merge-group
k-means, tree search, dialogue memory) is taken on the whole stream here too, from metadata that is identical on every
rank — see sharded.py, which owns the data movement (point-to-point rows of a merge group that straddles ranks, ONE
right-sized `all_gather_into_tensor` of the selected features before the single-GPU 7B prefill).  This module holds the
process-group set-up, the chunk partition (frame ranges follow from it: they are never exchanged) and the two small
object collectives used for TEXT (chunk captions once per update, the summary text of a merge)."""
import os

import torch
import torch.distributed as dist


class DistContext:
    def __init__(self, rank=0, world=1, device="cuda", backend=None):
        self.rank, self.world, self.device, self.backend = rank, world, torch.device(device), backend

    @property
    def is_root(self):
        return self.rank == 0


DEFAULT_TIMEOUT_S = 120        # rendezvous and every collective: a rank that never arrives raises on the others instead of hanging for RCCL's 10 minutes


def init_process_group(backend, rank, world, device=None, timeout_s=None):
    """`dist.init_process_group` with an explicit time-out (SC_DIST_TIMEOUT_S, default 120 s) and, on RCCL, the rank's device (`device_id`:
    the communicator is created eagerly on that device instead of lazily inside the first collective)."""
    import datetime
    t = float(os.environ.get("SC_DIST_TIMEOUT_S", DEFAULT_TIMEOUT_S) if timeout_s is None else timeout_s)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = dict(device_id=device) if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=t), **kw)


def init_from_env(device_type="cuda", timeout_s=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run) and initialises the process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type == "cuda":
        torch.cuda.set_device(local)
        device = torch.device(f"cuda:{local}")
        backend = "nccl"
    else:
        device, backend = torch.device("cpu"), "gloo"
    if world > 1 and not dist.is_initialized():
        init_process_group(backend, rank, world, device if backend == "nccl" else None, timeout_s)
    return DistContext(rank, world, device, backend)


def warm_up(ctx):
    """Create every communicator the sharded step uses BEFORE the first timed (or user-visible) step: one 1-element all-gather (the
    collective communicator) and one 1-element `exchange` with every peer (RCCL builds point-to-point channels lazily per pair, inside the
    first send / recv - a first 8-GPU step would otherwise pay, or hang in, eight of them).  Returns what was done, for the record."""
    if ctx.world == 1 or not dist.is_initialized():
        return dict(world=ctx.world, all_gather=False, p2p_peers=0)
    dev = ctx.device if ctx.backend == "nccl" else torch.device("cpu")
    one = torch.full((1,), float(ctx.rank), device=dev)
    got = torch.empty(ctx.world, device=dev)
    dist.all_gather_into_tensor(got, one)
    if got.tolist() != [float(r) for r in range(ctx.world)]:
        raise RuntimeError(f"dist.warm_up: the all-gather returned {got.tolist()} on rank {ctx.rank}")
    peers = 0
    for shift in range(1, ctx.world):              # round `shift`: send to rank + shift, receive from rank - shift (every pair once per direction)
        dst, src = (ctx.rank + shift) % ctx.world, (ctx.rank - shift) % ctx.world
        box = torch.empty(1, device=dev)
        exchange(ctx, [(one, dst)], [(box, src)])
        if float(box.item()) != float(src):
            raise RuntimeError(f"dist.warm_up: rank {ctx.rank} received {box.item()} from rank {src}")
        peers += 1
    dist.barrier()
    return dict(world=ctx.world, all_gather=True, p2p_peers=peers, backend=ctx.backend)


def partition_chunks(n_frames, chunk_size, world):
    """Contiguous, balanced dealing of whole chunks: returns [(frame_start, frame_end)] per rank; chunk boundaries and the
    global frame order are those of the single-process stream (inference_streaming_longva_v2.py:346)."""
    n_chunks = (n_frames + chunk_size - 1) // chunk_size
    out = []
    for r in range(world):
        c0, c1 = (n_chunks * r) // world, (n_chunks * (r + 1)) // world
        out.append((min(c0 * chunk_size, n_frames), min(c1 * chunk_size, n_frames)))
    return out


def owner_of(frame, parts):
    for r, (a, b) in enumerate(parts):
        if a <= frame < b:
            return r
    raise IndexError(frame)


def gather_objects(ctx, obj):
    """all-gather of small Python metadata (node captions, depths, frame ranges)."""
    if ctx.world == 1 and not (getattr(ctx, "always_collective", False) and dist.is_initialized()):
        return [obj]
    out = [None] * ctx.world
    dist.all_gather_object(out, obj)
    return out


def broadcast_object(ctx, obj, src=0):
    if ctx.world == 1 and not (getattr(ctx, "always_collective", False) and dist.is_initialized()):
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


# ---- tensor collectives.  On RCCL ("nccl") they run on the device tensors as they are.  The gloo backend has no device transport
# for these, so device tensors are staged through the host: that is what lets the whole N > 1 code path (real HIP kernels, one
# process per rank) run on a box with a single GPU in the tests; the CPU tests call the same functions with host tensors. ----
def _staged(ctx, t):
    return ctx.backend == "gloo" and t.is_cuda


def all_gather_rows(ctx, recv, send):
    """recv[world * n] <- every rank's send[n] (one all_gather_into_tensor of equal, right-sized slots)."""
    if _staged(ctx, send):
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(r, send.cpu())
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send)


def all_gather_row_windows(ctx, table, windows):
    """table [rows, n] (the same shape on every rank): rank q owns rows windows[q] = (first, count); afterwards every rank holds every
    window.  One all_gather_into_tensor of equal slots (the widest window; 32 fp64 rows x T K of the data-parallel k-means: 512 KB)."""
    first, count = windows[ctx.rank]
    cmax = max(c for _, c in windows)
    send = table.new_zeros((cmax,) + tuple(table.shape[1:]))
    send[:count].copy_(table[first:first + count])
    recv = table.new_empty((ctx.world * cmax,) + tuple(table.shape[1:]))
    all_gather_rows(ctx, recv, send)
    for q, (f, c) in enumerate(windows):
        if q != ctx.rank:
            table[f:f + c].copy_(recv[q * cmax:q * cmax + c])


def broadcast_tensor(ctx, t, src=0):
    if _staged(ctx, t):
        h = t.cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)


def exchange(ctx, sends, recvs):
    """Point-to-point pieces in one batch: sends = [(tensor, dst)], recvs = [(tensor view to fill, src)]; matching order on both ends."""
    stage = any(_staged(ctx, t) for t, _ in sends + recvs)
    keep, ops = [], []
    for t, dst in sends:
        t = t.contiguous()
        t = t.cpu() if stage else t
        keep.append(t)
        ops.append(dist.P2POp(dist.isend, t, dst))
    hosts = []
    for t, src in recvs:
        h = torch.empty(t.shape, dtype=t.dtype) if stage else t
        hosts.append((t, h))
        ops.append(dist.P2POp(dist.irecv, h, src))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if stage:
        for t, h in hosts:
            t.copy_(h)
