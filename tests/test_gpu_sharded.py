"""GPU: the sharded (N > 1) code path of bench.py run at world size 1 (`--force-sharded`) must equal the single-GPU step through the
reference-seam functions on the same stream — same tree, same retrieved frames, bit-identical [short | long] feature block
(VERDICT r01 item 1).  The multi-rank control flow itself is covered on CPU by tests/test_sharded_gloo.py (gloo, world 2 and 4)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _describe(nodes):
    def one(n):
        return dict(depth=n.depth, rows=int(n.centroids.shape[0]), text=n.text, children=[one(c) for c in n.children])
    return [one(n) for n in nodes]


def test_force_sharded_world1_equals_single_gpu_step():
    import bench
    dev = torch.device("cuda:0")
    pipe = bench.Pipeline(dev, 440, with_llm=False)                 # 11 chunks of 40: ten of them merge (k-means T=400, K=5, full D)
    a = pipe.step()
    want = torch.cat([t.reshape(-1, t.shape[-1]) for t in list(a["short"]) + list(a["path_feats"])]).clone()
    tree_a, text_a = _describe(a["tree"]), list(a["path_text"])
    merged_a = [n.centroids.clone() for n in a["tree"] if n.depth > 0]
    b = pipe.step_sharded()
    assert _describe(b["tree"]) == tree_a and list(b["path_text"]) == text_a
    assert torch.equal(b["image_embeddings"], want)
    merged_b = [b["mem"].fetch([n.centroids]) for n in b["tree"] if n.depth > 0]
    assert len(merged_a) == 1 and all(torch.equal(x, y) for x, y in zip(merged_a, merged_b))
    # retrieved frame indices: 5 short-memory frames of the last 20, then two whole chunks
    frames = [f for r in b["wanted"] for f in r]
    assert len(frames) == 5 + 40 + 40 and all(k == "frame" and 420 <= f < 440 for k, _, f in frames[:5])


def test_encode_is_independent_of_how_the_stream_is_batched():
    """What makes the sharded encode P-independent: a frame's features do not depend on which other frames share its
    micro-batch (every GEMM / attention row is computed with a fixed reduction order)."""
    import bench
    dev = torch.device("cuda:0")
    pipe = bench.Pipeline(dev, 120, with_llm=False)
    pipe.encode()
    whole = pipe.feats.clone()
    for lo, hi in ((0, 40), (40, 47), (47, 119), (119, 120)):                   # a rank owning frames [lo, hi) encodes them on their own
        part = pipe.enc.encode_frames_u8(pipe.frames[lo:hi])
        assert torch.equal(part, whole[lo:hi])


def test_c5_multi_round_session_equals_single_stream_functions():
    """C5's control flow (bench.Pipeline.session: a persistent tree grown over several question rounds) at world size 1 vs the
    reference-seam functions called round by round on the same frames: same tree, same retrieved frames, same feature rows."""
    import numpy as np
    import bench
    from streamchat_amd import streaming as S, synthetic, utiles as U
    dev = torch.device("cuda:0")
    pipe = bench.Pipeline(dev, 0, with_llm=False)
    pipe.prepare_rounds(3, 440)                                     # 11 chunks per round: a depth-0 merge in every round
    got = pipe.session(decode_tokens=0)
    # ---- the single-stream functions, round by round ----
    cap, tok = synthetic.SyntheticCaptioner(dev), synthetic.SyntheticTokenizer()
    torch.manual_seed(0)
    rng = np.random.RandomState(0)
    tree, cache = None, U.CaptionEmbeddingCache()
    for r in range(3):
        feats = pipe.round_feats[r]                                 # (encode is bit-reproducible: the session left the same rows here)
        bank = [feats[i:i + 1] for i in range(feats.shape[0])]
        tree, short = S.updating_memory_buffer(bank, tree, cap, tok, True, rng=rng, **bench.MEM)
        path, texts = U.fast_search_tree_multi_modal_with_embedding(tree, pipe.questions[r], feats, pipe.colbert, pipe.tok, cache=cache)
        rec = got["rounds"][r]
        assert rec["path_text"] == texts
        assert rec["top"] == [(n.depth, int(n.centroids.shape[0])) for n in tree]
        assert sum(len(x) for x in rec["wanted"]) == sum(int(t.shape[0]) for t in list(short) + list(path))
    # final tree: identical structure and bit-identical merge centroids
    assert _describe(got["tree"]) == _describe(tree)
    mine = [got["mem"].fetch([n.centroids]) for n in got["tree"] if n.depth > 0]
    ref = [n.centroids for n in tree if n.depth > 0]
    assert len(mine) == 3 and all(torch.equal(a, b) for a, b in zip(mine, ref))


def test_rccl_single_rank_exercises_every_collective_of_the_sharded_step():
    """A 1-rank RCCL ("nccl") process group on the GPU box: the sharded step with `always_collective` makes exactly the calls an
    N-rank run makes (all_gather_object of captions, broadcast_object_list of the summary, the int64 Ref broadcast,
    all_gather_into_tensor of the selected rows) and must still equal the single-GPU step."""
    import socket
    import torch.distributed as dist
    import bench
    from streamchat_amd import dist as DD, sharded as SH
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ctx = DD.DistContext(0, 1, dev, "nccl")
        ctx.always_collective = True
        pipe = bench.Pipeline(dev, 440, with_llm=False, ctx=ctx)
        a = pipe.step()
        want = torch.cat([t.reshape(-1, t.shape[-1]) for t in list(a["short"]) + list(a["path_feats"])]).clone()
        orig = SH.ShardedMemory.__init__

        def forced(self, *args, **kw):
            orig(self, *args, always_collective=True, **kw)
        SH.ShardedMemory.__init__ = forced
        try:
            b = pipe.step_sharded()
        finally:
            SH.ShardedMemory.__init__ = orig
        assert torch.equal(b["image_embeddings"], want) and list(b["path_text"]) == list(a["path_text"])
        assert _describe(b["tree"]) == _describe(a["tree"])
    finally:
        dist.destroy_process_group()


def _bench_json(args, env, nproc=1):
    import json
    import subprocess
    cmd = [sys.executable]
    if nproc > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", "29533"]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, env={**os.environ, **env}, cwd=ROOT, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("nproc", [2, 4])
def test_processes_through_torchrun_retrieve_and_prefill_like_one(nproc):
    """The whole `bench.py --gpus N` job (N = 2, 4) as the driver launches it (torch.distributed.run, one process per rank, real HIP kernels in both,
    7B prefill on rank 0), on this 1-GPU box: both ranks share device 0 (SC_ALL_RANKS_ON_GPU0) and the collectives go through gloo with
    host staging (SC_DIST_BACKEND=gloo; RCCL refuses two ranks on one device).  The 880-frame stream straddles the ranks inside its
    merge group (frames 0..399 over ranks owning [0, 440) and [440, 880) at N = 2; three ranks at N = 4), so the P2P fetch, the all-gather of the selected rows, the
    Ref broadcast and the caption exchange all carry data.  Retrieved frames, path text and the first generated token must equal the
    1-process run of the same stream."""
    common = ["--config", "C4", "--frames", "880", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--decode-tokens", "0"]
    one = _bench_json(common + ["--force-sharded"], {})
    two = _bench_json(common + ["--gpus", str(nproc)], {"SC_ALL_RANKS_ON_GPU0": "1", "SC_DIST_BACKEND": "gloo"}, nproc=nproc)
    assert two["n_gpus"] == nproc and one["n_gpus"] == 1
    for k in ("retrieval_crc32", "first_token", "context_tokens"):
        assert one["config"][k] == two["config"][k], (k, one["config"][k], two["config"][k])
