"""Multi-GPU layer: the frame stream shards by WHOLE CHUNKS across the GPUs of one node (one process per GPU,
`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).

What crosses the fabric: frame encode, chunk captioning and the chunk-group k-means are rank-local
(data parallel, no data-path collective); the reference's stages that need a global view are text-only (tree search over
captions, dialogue memory) and run on rank 0 from all-gathered node METADATA.  Only the SELECTED frame features (short-term
memory frames + the retrieved nodes) move: one fixed-capacity `all_gather` (each peer pushes its slice over its own xGMI
link), after which rank 0 owns the [short | long] token block for the single-GPU 7B prefill.  Nothing here touches label /
centroid arithmetic, so k-means results do not depend on the number of GPUs."""
import os

import torch
import torch.distributed as dist


class DistContext:
    def __init__(self, rank=0, world=1, device="cuda", backend=None):
        self.rank, self.world, self.device, self.backend = rank, world, torch.device(device), backend

    @property
    def is_root(self):
        return self.rank == 0


def init_from_env(device_type="cuda"):
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
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = dict(device_id=device) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return DistContext(rank, world, device, backend)


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
    if ctx.world == 1:
        return [obj]
    out = [None] * ctx.world
    dist.all_gather_object(out, obj)
    return out


def broadcast_object(ctx, obj, src=0):
    if ctx.world == 1:
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def gather_selected_frames(ctx, local_bank, local_range, wanted, capacity):
    """All-gather of selected frame features.
      local_bank  [n_local, P, D] features of frames local_range = (start, end) (global indices)
      wanted      list of global frame indices, identical on every rank (decided by rank 0 and broadcast)
      capacity    max frames any single rank contributes (fixed-size collective, identical on every rank)
    Returns [len(wanted), P, D] in `wanted` order on every rank (rank 0 is the consumer)."""
    a, b = local_range
    ranges = gather_objects(ctx, (a, b))
    # slot of every wanted frame: (owner rank, k-th frame that owner contributes) — computed identically on every rank, so an
    # over-capacity request fails everywhere BEFORE the collective (no rank is left waiting inside all_gather)
    counters, slots = [0] * ctx.world, []
    for f in wanted:
        r = next((i for i, (x, y) in enumerate(ranges) if x <= f < y), None)
        if r is None:
            raise IndexError(f"frame {f} is owned by no rank")
        slots.append((r, counters[r]))
        counters[r] += 1
    if max(counters, default=0) > capacity:
        raise ValueError(f"a rank would send {max(counters)} frames > capacity {capacity}")
    mine = [f for f in wanted if a <= f < b]
    P, D = local_bank.shape[1], local_bank.shape[2]
    buf = torch.zeros((capacity, P, D), dtype=local_bank.dtype, device=local_bank.device)
    if mine:
        idx = torch.tensor([f - a for f in mine], device=local_bank.device)
        buf[: len(mine)] = local_bank.index_select(0, idx)
    if ctx.world == 1:
        gathered = [buf]
    else:
        gathered = [torch.empty_like(buf) for _ in range(ctx.world)]
        dist.all_gather(gathered, buf)                      # NCCL/RCCL: each peer pushes its slice over its own xGMI link
    return torch.stack([gathered[r][k] for r, k in slots]) if slots else buf[:0]
