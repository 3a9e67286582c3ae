"""GPU: the sharded (N > 1) code path of bench.py run at world size 1 (`--force-sharded`) must equal the single-GPU step through the
reference-seam functions on the same stream — same tree, same retrieved frames, bit-identical [short | long] feature block
(VERDICT r01 item 1).  The multi-rank control flow itself is covered on CPU by tests/test_sharded_gloo.py (gloo, world 2, 3, 4 and 8); the full-size C4 / C5 jobs run here as 8 processes on the one GPU."""
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


def test_c5_session_decodes_inside_every_round():
    """C5 as configured (BASELINE.json configs[4]: "multi-round 7B decode"): `session(decode_tokens=8)` prefills and then decodes 8 tokens
    with the replayed hipGraph in EVERY round (the graph is captured once and must survive the next round's prefill, which rewinds and
    re-fills the same KV cache).  Checked against the same model driven round by round through `generate_with_image_embedding`
    (prefill + the eager token loop) on the feature block the session retrieved.  A 2-layer Qwen2-7B-width model keeps it short."""
    import bench
    from streamchat_amd import llm as LM
    dev = torch.device("cuda:0")
    pipe = bench.Pipeline(dev, 0, with_llm=False)
    qc = LM.Qwen2ConfigLite(**dict(LM.QWEN2_7B, layers=2))
    pipe.model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(LM.random_qwen2_state_dict(qc, seed=4, device=dev), qc, device=dev, max_seq=4 * 23040 + 8192), pipe.enc)
    pipe.prepare_rounds(2, 440)
    got = pipe.session(decode_tokens=8)
    assert [len(r["tokens"]) for r in got["rounds"]] == [8, 8]
    contexts = [r["context"] for r in got["rounds"]]
    assert contexts[0] > 5 * 576 + 2 * 40 * 576 - 1 and contexts[1] >= contexts[0]
    # the same rounds again: prefill through the public generate call on the rows the session's Refs describe, then the EAGER one-token
    # forward teacher-forced with the session's tokens: every token the graph emitted must be the eager arg-max (up to an fp16-level
    # tie: the two loops split the KV range differently, so their logits differ in the last bits)
    from streamchat_amd import sharded as SH
    mem = got["mem"]
    for r, rec in enumerate(got["rounds"]):
        rows = []
        for piece in rec["wanted"]:
            for kind, i, f in piece:
                rows.append(pipe.round_feats[i][f] if kind == "frame" else mem.store[(SH.MERGE, i)][f])
        emb = torch.stack(rows).reshape(-1, 3584)
        pipe.question = pipe.questions[r]
        pipe.prefill(emb, rec["path_text"][-1])
        assert pipe.last["context"] == rec["context"]
        lm = pipe.model.lm
        tok = int(pipe.last["first_token"][0, 0])
        for t in rec["tokens"]:
            logits = lm.forward(lm.embed_tokens(torch.tensor([tok], device=dev))).float()
            assert float(logits[t]) >= float(logits.max()) - 2e-3 * float(logits.abs().max()), (r, t, int(logits.argmax()))
            tok = t


def test_rccl_single_rank_exercises_every_collective_of_the_sharded_step():
    """A 1-rank RCCL ("nccl") process group on the GPU box: the sharded step with `always_collective` makes exactly the calls an
    N-rank run makes (all_gather_object of captions, broadcast_object_list of the summary, the int64 Ref broadcast,
    all_gather_into_tensor of the selected rows) and must still equal the single-GPU step; then the same with the data-parallel Lloyd."""
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

        # ... and with the merge k-means DATA-PARALLEL over columns (sharded._dp_lloyd): at one rank the slab is the whole matrix, but the calls
        # are the N-rank ones - sc_kmeans_fit_cols calling back into Python per Lloyd iteration, the fp64 segment tables through RCCL's
        # all_gather_into_tensor on the stream the kernels run on - and the merged node's centroids are retrieved rows of this stream
        def forced_dp(self, *args, **kw):
            orig(self, *args, always_collective=True, dp_lloyd=True, **kw)
        SH.ShardedMemory.__init__ = forced_dp
        try:
            c = pipe.step_sharded()
        finally:
            SH.ShardedMemory.__init__ = orig
        assert pipe.last["mem"].traffic["dp_lloyd_fits"] >= 1
        assert torch.equal(c["image_embeddings"], want) and list(c["path_text"]) == list(a["path_text"])
        assert _describe(c["tree"]) == _describe(a["tree"])
    finally:
        dist.destroy_process_group()


def _bench_json(args, env, nproc=1, launcher="torchrun", timeout=900):
    """bench.py as a subprocess: under torch.distributed.run like the driver launches N > 1 (launcher="torchrun"), or as the bare command
    `python bench.py --gpus N ...` (launcher="bare": bench.py must start its ranks itself)."""
    import json
    import subprocess
    cmd = [sys.executable]
    if nproc > 1 and launcher == "torchrun":
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1", "--master-port", "29533"]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, env={**base, **env}, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-4000:]
    return json.loads(lines[0])


C4_SMALL = ["--config", "C4", "--frames", "880", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--decode-tokens", "0"]
_one_process = {}


def _c4_small_one_process():
    if not _one_process:
        _one_process.update(_bench_json(C4_SMALL + ["--force-sharded"], {}))
    return _one_process


@pytest.mark.parametrize("nproc,launcher", [(2, "bare"), (4, "torchrun")])
def test_processes_retrieve_and_prefill_like_one(nproc, launcher):
    """The whole `bench.py --gpus N` job (7B prefill on rank 0, real HIP kernels in every rank) on this 1-GPU box: all ranks share device 0
    (SC_ALL_RANKS_ON_GPU0) and the collectives go through gloo with host staging (SC_DIST_BACKEND=gloo; RCCL refuses two ranks on one
    device).  N = 2 is started as the BARE command `python bench.py --gpus 2 ...` — no launcher around it: bench.py has to start its
    ranks itself — N = 4 through torch.distributed.run as the task brief's driver does.  The 880-frame stream
    straddles the ranks inside its merge group (frames 0..399 over ranks owning [0, 440) and [440, 880) at N = 2; three ranks at N = 4),
    so the P2P fetch, the all-gather of the selected rows, the Ref broadcast and the caption exchange all carry data.  Retrieved frames,
    path text and the first generated token must equal the 1-process run of the same stream, and the record must say what it is."""
    one = _c4_small_one_process()
    many = _bench_json(C4_SMALL + ["--gpus", str(nproc)], {"SC_ALL_RANKS_ON_GPU0": "1", "SC_DIST_BACKEND": "gloo"}, nproc=nproc, launcher=launcher)
    assert many["n_gpus"] == nproc and one["n_gpus"] == 1
    for k in ("retrieval_crc32", "first_token", "context_tokens"):
        assert one["config"][k] == many["config"][k], (k, one["config"][k], many["config"][k])
    assert many["config"]["launcher"] == ("self (bare command)" if launcher == "bare" else "torch.distributed.run")
    assert "880-frame" in many["metric"] and many["config"]["frames_total"] == 880
    assert many["encode_frames_per_s"] > 0 and many["encode_frames_per_s_1gpu_same_job"] > 0 and "SERIAL on rank 0" in many["scaling_note"]
    keys = list(many)
    assert keys.index("encode_frames_per_s") < keys.index("config")       # the encode rate leads the record
    assert many["dist_warm_up"]["all_gather"] and many["dist_warm_up"]["p2p_peers"] == nproc - 1 and many["dist_warm_up"]["timeout_s"] == 120.0


@pytest.mark.parametrize("nproc", [2, 3])
def test_dp_lloyd_processes_retrieve_and_prefill_like_one(nproc):
    """`bench.py --gpus N --dp-lloyd`: the merge group's k-means DATA-PARALLEL over columns (sharded._dp_lloyd -> sc_kmeans_fit_cols) in a real
    multi-process job - N processes on this box's one GPU, gloo with host staging: the 400 rows of the merge group are transposed into column
    slabs of whole SC-KM2 segments (16 + 16 at N = 2, 10 + 11 + 11 at N = 3), every Lloyd iteration all-gathers the ranks' rows of the two fp64
    segment tables, the centroid slabs go to the executor.  The selected frames, the path text, the context and the first generated token
    must equal the 1-process run - the merged node's centroids are retrieved rows of this stream - and the record must show the fit."""
    one = _c4_small_one_process()
    many = _bench_json(C4_SMALL + ["--gpus", str(nproc), "--dp-lloyd"], {"SC_ALL_RANKS_ON_GPU0": "1", "SC_DIST_BACKEND": "gloo"}, nproc=nproc, launcher="bare")
    assert many["n_gpus"] == nproc
    for k in ("retrieval_crc32", "first_token", "context_tokens"):
        assert one["config"][k] == many["config"][k], (k, one["config"][k], many["config"][k])
    dp = many["collective"]["dp_lloyd"]
    assert dp["fits_last_step"] >= 1 and dp["bytes_moved_last_step"] > 400 * 576 * 3584 * 2 * (nproc - 1) // nproc


_full_size_one = {}          # tag -> the 1-process record of a full-size job (shared by the tests that compare against it)


def _one_process_record(tag, cfg_args):
    if tag not in _full_size_one:
        _full_size_one[tag] = _bench_json(cfg_args + ["--force-sharded"], {}, timeout=1100)
    return _full_size_one[tag]


def _full_size_pair(cfg_args, tag):
    """the SAME full-size job as ONE process and as EIGHT processes (bare command: bench.py starts its ranks; all on device 0, gloo with host
    staging); both JSON lines are left under gpurun_out/ when that directory exists (copied to profiles/ by hand)"""
    import json
    one = _one_process_record(tag, cfg_args)
    eight = _bench_json(cfg_args + ["--gpus", "8"], {"SC_ALL_RANKS_ON_GPU0": "1", "SC_DIST_BACKEND": "gloo"}, nproc=8, launcher="bare", timeout=1100)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, f"r04_{tag}_1proc_vs_8proc_one_gpu.jsonl"), "w") as f:
            f.write(json.dumps(one) + "\n" + json.dumps(eight) + "\n")
    assert one["n_gpus"] == 1 and eight["n_gpus"] == 8 and eight["config"]["launcher"] == "self (bare command)"
    return one, eight


@pytest.mark.timeout(1500)
def test_c4_full_size_4096_frames_in_8_processes_equals_one_process():
    """BASELINE.json configs[3] AT ITS SIZE: ONE 4096-frame stream = 103 chunks dealt to EIGHT ranks by whole chunks
    (12-13 chunks = 480-520 frames each), rank-local encode + captions, the single-stream tree policy (ten merges' worth of depth-0 nodes,
    ONE merge k-means T = 400 on the rank owning the group), Ref broadcast, all-gather of the selected rows, 49 k-token 7B prefill on rank 0.
    Eight processes share this box's one GPU; the collectives are gloo with host staging (RCCL refuses two ranks per device, so RCCL's own
    transport stays unmeasured).  Retrieved frames + path text, context length and the first generated token equal the 1-process run."""
    args = ["--config", "C4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--decode-tokens", "0"]
    one, eight = _full_size_pair(args, "c4_4096")
    assert eight["config"]["frames_total"] == 4096 and "4096-frame" in eight["metric"]
    for k in ("retrieval_crc32", "first_token", "context_tokens"):
        assert one["config"][k] == eight["config"][k], (k, one["config"][k], eight["config"][k])
    assert eight["config"]["frames_rank0"] in (480, 520)


@pytest.mark.timeout(1500)
def test_c4_full_size_in_8_processes_with_the_data_parallel_lloyd_equals_one_process():
    """The same full-size C4 job with `--dp-lloyd`: the T = 400 x D = 2 064 384 merge k-means runs on EIGHT ranks at once, each on the columns of
    four SC-KM2 segments (sharded._dp_lloyd -> sc_kmeans_fit_cols; rows transposed into column slabs point to point, the fp64 segment tables
    all-gathered per Lloyd iteration, centroid slabs to the executor) - and the job still retrieves the same frames and generates the same first
    token as ONE process, because the fit is bit-identical by construction of the reduction spec."""
    args = ["--config", "C4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--decode-tokens", "0"]
    one = _one_process_record("c4_4096", args)
    eight = _bench_json(args + ["--gpus", "8", "--dp-lloyd"], {"SC_ALL_RANKS_ON_GPU0": "1", "SC_DIST_BACKEND": "gloo"}, nproc=8, launcher="bare", timeout=1100)
    assert eight["n_gpus"] == 8 and eight["config"]["frames_total"] == 4096
    for k in ("retrieval_crc32", "first_token", "context_tokens"):
        assert one["config"][k] == eight["config"][k], (k, one["config"][k], eight["config"][k])
    dp = eight["collective"]["dp_lloyd"]
    assert dp["fits_last_step"] == 1 and dp["bytes_moved_last_step"] > 400 * 576 * 3584 * 2 * 7 // 8


@pytest.mark.timeout(2400)
def test_c5_full_size_8192_frames_8_rounds_in_8_processes_equals_one_process():
    """BASELINE.json configs[4] AT ITS SIZE: ONE 8192-frame ego stream in 8 question rounds of 1024 new frames, every round sharded over
    EIGHT ranks (3-4 chunks each), the ONE persistent short / long memory tree grown over the rounds (a merge per round, deeper merges as
    depth-1 nodes accumulate), BERT-large-CLS tree search + MiniLM dialogue memory, 7B prefill of a context that grows with the tree and a
    64-token graph decode on rank 0 in EVERY round.  Per round: retrieved frames + path text, context length and the 64 decoded tokens
    equal the 1-process session."""
    args = ["--config", "C5", "--rounds", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    one, eight = _full_size_pair(args, "c5_8192")
    assert eight["config"]["frames_total"] == 8192 and len(eight["config"]["rounds"]) == 8
    for r, (a, b) in enumerate(zip(one["config"]["rounds"], eight["config"]["rounds"])):
        assert a == b, (r, a, b)
        assert a["tokens_crc32"] is not None and a["context"] > 5 * 576 + 2 * 23040
    assert one["config"]["context_tokens"] == eight["config"]["context_tokens"] > 200000          # the last round's context: ~210 k tokens


def test_weak_scaling_record_names_the_total_frames():
    """Default workload (C3 per GPU) at N = 2: the metric string must name the frames the job processed in total, not "1024-frame"."""
    rec = _bench_json(["--gpus", "2", "--frames", "440", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--decode-tokens", "0", "--no-llm"],
                      {"SC_ALL_RANKS_ON_GPU0": "1", "SC_DIST_BACKEND": "gloo"}, nproc=2, launcher="bare")
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["frames_total"] == 880 and rec["config"]["frames_per_gpu"] == 440
    assert "880-frame stream in total = 440 frames per GPU x 2 GPUs" in rec["metric"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: RCCL refuses two ranks on one device")
def test_two_ranks_on_two_gpus_over_rccl_equal_one_process():
    """On the first box with more than one GPU: the same 880-frame job on REAL RCCL (batch_isend_irecv of the straddling merge group,
    all_gather_into_tensor of the selected rows, the Ref broadcast over xGMI), started as the bare command."""
    one = _c4_small_one_process()
    two = _bench_json(C4_SMALL + ["--gpus", "2"], {}, nproc=2, launcher="bare")
    assert two["n_gpus"] == 2
    for k in ("retrieval_crc32", "first_token", "context_tokens"):
        assert one["config"][k] == two["config"][k], (k, one["config"][k], two["config"][k])


def test_bare_command_refuses_more_ranks_than_gpus():
    """`--gpus 64` on a box without 64 GPUs (and without the shared-device test switch) must fail loudly before anything is launched."""
    import subprocess
    base = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SC_ALL_RANKS_ON_GPU0")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], env=base, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "exposes" in r.stderr


@pytest.mark.parametrize("nproc", [1, 4])
def test_preflight_passes_on_a_healthy_job_and_says_what_it_checked(nproc):
    """`bench.py --gpus N --preflight` (round 6): one JSON line, exit code 0; at N = 4 (torch.distributed.run, every rank on device 0, gloo) the record carries the
    communicator warm-up (one all-gather + a point-to-point round with each of the 3 peers) and every rank's free HBM; `dist_warm_up` is also in a real N > 1 record."""
    env = {"SC_ALL_RANKS_ON_GPU0": "1", "SC_DIST_BACKEND": "gloo"} if nproc > 1 else {}
    rec = _bench_json(["--config", "C4", "--preflight"] + (["--gpus", str(nproc)] if nproc > 1 else []), env, nproc=nproc, launcher="torchrun")
    assert rec["preflight"] == "ok" and rec["n_gpus"] == nproc and rec["visible_gpus"] >= 1
    assert sum(rec["partition_frames_per_rank"]) == 4096 and rec["free_hbm_gb"] > rec["need_hbm_gb"]
    if nproc > 1:
        assert rec["collectives"] == dict(world=4, all_gather=True, p2p_peers=3, backend="gloo") and len(rec["free_hbm_gb_per_rank"]) == 4
