#!/usr/bin/env python3
"""bench.py — headline benchmark of the streaming hot path on MI355X (driver contract in the task brief).

  python bench.py --gpus N --steps K --warmup W [--config C1|C2|C3|C4|C5]
  (N > 1: one rank per GPU under torch.distributed.run, as the driver launches it; typed as a bare command it starts the N ranks itself)

A "step" is one pass of the hot path over one synthetic 336x336 frame stream:
  encode   uint8 frames -> fused preprocess/patchify -> ViT-L/14-336 (23 layers) -> mlp2x_gelu projector   [MFMA]
  select   forgetting-curve short memory + chunking + memory-tree update incl. ONE whole-frame k-means
           (T = 400 frames, K = 5, D = 576*3584)                                                            [HBM]
  retrieve caption / dialogue embeddings -> cosine / flat-L2 top-k -> tree search                           [latency]
  answer   [short | retrieved] frame tokens + prompt -> Qwen2-7B prefill + first token                      [MFMA]
Workloads (BASELINE.json configs):  C3 (default, the configuration the metric is quoted on): 1024 frames per GPU, full path;
C2 = C3 without the LLM; C1 = 64 frames, encode + k-means(k=8); C4 = ONE 4096-frame stream sharded over the N ranks
(strong scaling), all-gather of the selected features, 7B prefill on rank 0; C5 = ONE 8192-frame ego stream in 8 question rounds of
1024 new frames: every round shards its segment over the ranks, grows the ONE persistent memory tree (short / long memory),
retrieves (BERT-large CLS tree search + MiniLM dialogue memory), prefills and decodes 64 tokens on rank 0 (a "step" = the session).  With N > 1 the stream is ONE global stream dealt to
the ranks by whole chunks and the memory update has single-stream semantics (streamchat_amd/sharded.py): the retrieved frames do
not depend on N.  Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # the pool's host driver only supports dmabuf IPC: RCCL across processes needs it (task brief)

import numpy as np      # noqa: E402
import torch            # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from streamchat_amd import dist as DD, llm as LM, ops, sharded as SH, streaming as S, synthetic, text as T, utiles as U, vision as V   # noqa: E402
from streamchat_amd.memory_bank.memory_retrieval import local_doc_qa as Q   # noqa: E402
from streamchat_amd.mm_utils import tokenizer_image_token   # noqa: E402

FRAMES = 1024
MICRO_BATCH = int(os.environ.get("SC_MICRO_BATCH", "512"))     # frames per ViT pass: 512 x 577 rows = 1154 whole 256-row GEMM tiles, 6.6 GB of activations
MEM = dict(chunk_size=40, num_clusters=5, interval=10, short_window=20, remember_window=5, tau=5)   # inference_streamchat_v0.3.sh:12-19
GFLOP_PER_FRAME = 385.1            # patch 0.69 + 23 x 15.88 + projector 19.03
MFMA_PEAK_TF = 2500.0              # dense fp16/bf16 MFMA, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
TRAFFIC_PROFILE = "profiles/r06_pmc_traffic.json"       # rocprofv3 --pmc passes of this command (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["C1", "C2", "C3", "C4", "C5"], default=None)
    ap.add_argument("--rounds", type=int, default=8, help="C5: question rounds per session (1024 new frames each)")
    ap.add_argument("--frames", type=int, default=None, help="frames per GPU (C2/C3) or in total (C1/C4); default: the config's size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-llm", action="store_true", help="C2 workload: stop after retrieval (no 7B prefill)")
    ap.add_argument("--decode-tokens", type=int, default=512, help="decode tokens measured AFTER the timed region (reported separately)")
    ap.add_argument("--cpu-frames", type=int, default=64, help="frames of the CPU-baseline slice (SURVEY 8(d): 64 = C1's encode)")
    ap.add_argument("--force-sharded", action="store_true", help="exercise the N>1 code path with world size 1 (testing)")
    ap.add_argument("--dp-lloyd", action="store_true", help="N > 1: the merge group's k-means data-parallel over columns (sharded._dp_lloyd; "
                    "same bits as the owner-rank path, which stays the default until a multi-GPU box has run it); = SC_DP_LLOYD=1")
    ap.add_argument("--preflight", action="store_true",
                    help="check the job instead of running it: visible GPUs, chunk partition, free HBM per rank, one 1-element all-gather and one "
                         "point-to-point round per peer; ONE JSON line, exit code 0 / 3")
    ap.add_argument("--session", type=int, default=4, metavar="SEGMENTS",
                    help="AFTER the headline measurement (C3, one GPU): a stream of SEGMENTS segments with a --decode-tokens answer each, serial vs "
                         "reader/updater || QA-decode overlapped on two CU partitions (streamchat_amd/session.py); reported as `session`; 0 = off")
    ap.add_argument("--session-decode-cus", type=int, nargs="*", default=[128], help="CUs of the decode partition(s) to try in --session (multiples of 32)")
    ap.add_argument("--with-captions", type=int, nargs="?", const=2, default=1, metavar="STEPS",
                    help="AFTER the headline measurement: STEPS more steps (default 1; 0 = off) in which the chunk captioner is the HIP 7B model itself "
                         "(one 23 k-token prefill + 128 new tokens per 40-frame chunk through llm.BatchDecoder, reference utiles.py:539-559), "
                         "reported as a separate `product` object; `value` is untouched (the metric names encode+select+retrieve+prefill)")
    return ap.parse_args()


class TimedCaptioner:
    """The HIP LongVA model as the chunk captioner / merge summariser of the memory tree (what the reference does: utiles.py:539-559 one generate
    per 40-frame chunk - 23 040 image tokens + the caption prompt, 128 new tokens, temperature 0.1 - and :591-607 one text-only generate per
    merge), through the batched path of SURVEY 8(f).1: every chunk of the update is prefilled into its own slice of ONE per-layer KV cache, then
    all of them decode together (llm.BatchDecoder: the 15 GB of weights stream once per step for all chunks).  Times the two phases."""

    def __init__(self, model, max_new_tokens=128):
        self.m, self.device, self.config, self.max_new = model, model.device, model.config, max_new_tokens
        self.rec = dict(chunks=0, prompt_tokens=0, prefill_s=0.0, decode_s=0.0, decode_steps=0, new_tokens=0, summary_s=0.0, prefill_flop=0.0, decode_bytes=0.0)

    def generate_with_image_embedding(self, *a, **kw):                   # the merge summary (text-only prompt) and single-chunk updates
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = self.m.generate_with_image_embedding(*a, **kw)
        torch.cuda.synchronize(); self.rec["summary_s"] += time.perf_counter() - t0
        return out

    def generate_batch_with_image_embedding(self, inputs_list, image_embeddings_list, modalities=["image"], max_new_tokens=128, **kw):
        # the PRODUCT method itself runs (llm.LlavaQwenForCausalLM.generate_batch_with_image_embedding); it times its two phases and reports
        # the decode steps it really executed (EOS is checked every 16 steps: the graph replays up to 15 steps past the last sequence's end,
        # and decode_s contains them - round 4 counted max(len) - 1 steps and re-implemented the method here)
        m, r = self.m, self.rec
        m.collect_batch_stats = True
        try:
            out = m.generate_batch_with_image_embedding(inputs_list, image_embeddings_list, modalities, max_new_tokens=max_new_tokens, **kw)
        finally:
            m.collect_batch_stats = False
        st = m.batch_stats
        n, steps = st["prompt_tokens"], st["steps_run"]
        r["chunks"] += len(n); r["prompt_tokens"] += sum(n); r["prefill_s"] += st["prefill_s"]; r["decode_s"] += st["decode_s"]
        r["decode_steps"] += steps; r["new_tokens"] += st["new_tokens"]; r["nsplit"] = st["nsplit"]
        r["prefill_flop"] += sum(2 * k * 6.53e9 + 2 * k * k * 3584 * 28 for k in n)                      # SURVEY 8(d) flop model (causal)
        r["decode_bytes"] += steps * (14.1e9 + sum(2 * 28 * 4 * 128 * (k + steps / 2) * 2 for k in n))     # per step: weights once + every sequence's K/V
        return out


def measure_product(pipe, k, n_total):
    """`k` steps of the C3 step with the HIP 7B model as chunk captioner + merge summariser (TimedCaptioner), after one warm-up step"""
    pipe.captioner = TimedCaptioner(pipe.model)
    pipe.step()                                                  # warm-up: BatchDecoder's cache allocation, graph capture
    pipe.captioner = cp = TimedCaptioner(pipe.model)
    torch.cuda.synchronize(); t0c = time.perf_counter()
    for _ in range(k):
        pipe.step()
    torch.cuda.synchronize(); dtc = time.perf_counter() - t0c
    r = cp.rec
    return dict(
        product_frames_per_s=round(n_total * k / dtc, 2), ms_per_step=round(dtc / k * 1e3, 1), steps=k,
        what="the C3 step with the HIP LongVA-7B-shape model as chunk captioner + merge summariser (reference utiles.py:539-559,591-607; "
             "batched: llm.BatchDecoder, temperature 0.1, 128 new tokens): encode + caption + select + retrieve + answer prefill",
        chunks_per_step=r["chunks"] // k, prompt_tokens_per_chunk=r["prompt_tokens"] // max(r["chunks"], 1),
        caption_prefill_s_per_step=round(r["prefill_s"] / k, 3), caption_decode_s_per_step=round(r["decode_s"] / k, 3), summary_s_per_step=round(r["summary_s"] / k, 3),
        caption_prefill=dict(bound="mfma", achieved=round(r["prefill_flop"] / max(r["prefill_s"], 1e-9) / 1e12, 1), peak=MFMA_PEAK_TF, unit="TFLOP/s",
                             frac=round(r["prefill_flop"] / max(r["prefill_s"], 1e-9) / 1e12 / MFMA_PEAK_TF, 4)),
        caption_decode=dict(bound="hbm", achieved=round(r["decode_bytes"] / max(r["decode_s"], 1e-9) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=round(r["decode_bytes"] / max(r["decode_s"], 1e-9) / 1e9 / HBM_PEAK_GBS, 4),
                            tokens_per_s_aggregate=round(r["new_tokens"] / max(r["decode_s"], 1e-9), 1), ms_per_decode_step=round(r["decode_s"] / max(r["decode_steps"], 1) * 1e3, 2),
                            decode_steps_per_step=r["decode_steps"] // k, split_kv=r.get("nsplit")))


def measure_session(pipe, n_frames, segments, tokens, decode_cus_list):
    """BASELINE configs[2] as a STREAM of segments (reader / updater / QA): `segments` segments of `n_frames` frames, one question
    and one `tokens`-token greedy answer each - the same jobs once one after the other on the whole chip (the reference's batch entry point,
    inference_streaming_longva_v2.py:845-905) and once with the answer decode of segment i overlapped with the encode / update / prefill of
    segment i + 1 on two CU partitions and two host threads (streamchat_amd/session.py).  Frames per second over the whole session, and
    whether the two runs retrieved the same frames and produced the same tokens."""
    from streamchat_amd import session as SS
    dev = pipe.device
    segs = [torch.from_numpy(synthetic.frame_stream(n_frames, seed=1234, start=i * n_frames)).to(dev) for i in range(segments)]
    qs = [f"segment {i}: where did I leave the {synthetic.VOCAB[(7 * i) % len(synthetic.VOCAB)]} and what was on the kitchen table" for i in range(segments)]

    def run(overlap, dc):
        s = SS.StreamingSession(pipe.model, pipe.enc, pipe.colbert, pipe.tok, pipe.llm_tok, MEM, synthetic.SyntheticCaptioner(dev), synthetic.SyntheticTokenizer(),
                                overlap=overlap, decode_cus=dc, max_new_tokens=tokens, max_context=pipe.model.lm.max_seq)
        s.submit(segs[0], qs[0]); s.results()                    # warm-up segment (allocations), then the same session object starts over
        s.tree, s.search_cache, s.records, s.banks, s.n = None, U.CaptionEmbeddingCache(), [], [], 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f, q in zip(segs, qs):
            s.submit(f, q, new_video=True)                       # every segment is a C3 stream of its own (context ~49 k), as configs[2] describes it
        rec = s.results()
        dt = time.perf_counter() - t0
        s.close()
        keep = [{k: r[k] for k in ("short", "path_text", "retrieved_rows", "retrieved_crc", "context", "first_token", "tokens")} for r in rec]
        co = SS.StreamingSession.co_running(rec)
        del s
        torch.cuda.empty_cache()
        return dt, keep, co
    t_serial, ref, co_serial = run(False, 0)
    out = dict(what="C3 with its 512-token answer as a stream of segments: serial (one stream, whole chip) vs reader/updater || QA-decode on two CU partitions and two host threads",
               segments=segments, frames_per_segment=n_frames, answer_tokens=tokens, context=[r["context"] for r in ref],
               serial_s=round(t_serial, 3), serial_frames_per_s=round(segments * n_frames / t_serial, 2), overlapped=[])
    for dc in decode_cus_list:
        t, rec, co = run(True, dc)
        out["overlapped"].append(dict(decode_cus=dc, s=round(t, 3), frames_per_s=round(segments * n_frames / t, 2), speedup=round(t_serial / t, 3), identical_to_serial=rec == ref,
                                      gpu_timeline=co))
    best = max(out["overlapped"], key=lambda r: r["frames_per_s"])
    out["overlapped_frames_per_s"], out["best_decode_cus"], out["speedup"] = best["frames_per_s"], best["decode_cus"], best["speedup"]
    out["identical_to_serial"] = all(r["identical_to_serial"] for r in out["overlapped"])
    out["co_running_fraction"] = (best.get("gpu_timeline") or {}).get("co_running_fraction")
    if co_serial:                                                # what the pipeline can reach: fill + (segments - 1) co-running periods + drain
        m, d = co_serial["mfma_busy_ms"] / segments, co_serial["decode_busy_ms"] / segments
        out["serial_gpu_ms_per_segment"] = dict(mfma_side=round(m, 1), decode=round(d, 1))
        if best.get("gpu_timeline"):
            period = (best["s"] * 1e3 - m - d) / max(segments - 1, 1)
            out["steady_state"] = dict(period_ms=round(period, 1), speedup_many_segments=round((m + d) / period, 3),
                                       note="overlapped time = first segment's MFMA side alone + (segments - 1) co-running periods + last decode alone; a longer stream approaches "
                                            "speedup_many_segments")
    return out


class Pipeline:
    """`n_total` frames of ONE global stream; this rank holds (and encodes) the frames `parts[rank]` of it."""

    def __init__(self, device, n_total, seed=1234, with_llm=True, ctx=None, micro_batch=MICRO_BATCH, kmeans_k=None, max_seq=53248):
        self.ctx = ctx or DD.DistContext(0, 1, device)
        cfg = V.CLIPVisionConfigLite(**V.VIT_L_336)
        self.cfg = cfg
        self.sd_vit = V.random_clip_state_dict(cfg, seed=0, device=device)
        self.sd_proj = V.random_projector_state_dict(1024, 3584, seed=1, device=device)
        self.enc = V.FrameEncoder(V.CLIPVisionTower(self.sd_vit, cfg, device=device), V.MMProjector(self.sd_proj, device=device), micro_batch=micro_batch)
        self.n_total = n_total
        self.parts = DD.partition_chunks(n_total, MEM["chunk_size"], self.ctx.world)
        a, b = self.parts[self.ctx.rank]
        self.range, self.n = (a, b), b - a
        self.frames = torch.from_numpy(synthetic.frame_stream(b - a, seed=seed, start=a)).to(device)        # resident in HBM
        self.feats = torch.empty((b - a, cfg.num_patches, 3584), dtype=torch.float16, device=device)
        self.device = device
        self.kmeans_k = kmeans_k
        self.enc_events = []
        # retrieval models: BERT-large (mxbai-colbert as the reference loads it: plain encoder, CLS) for the caption tree,
        # all-MiniLM-L6 (mean-pool + L2-normalise) for the dialogue memory; random-init, hash tokenizer (offline)
        bl, ml = T.BertConfigLite(**T.BERT_LARGE), T.BertConfigLite(**T.MINILM_L6)
        self.colbert = T.BertEncoder(T.random_bert_state_dict(bl, seed=2, device=device), bl, device=device)
        self.tok = T.HashTokenizer()
        self.sent = Q.HipSentenceEmbeddings(T.SentenceEmbedder(T.BertEncoder(T.random_bert_state_dict(ml, seed=3, device=device), ml, device=device)), self.tok)
        self.docs = [Q.Document(f"Conversation content on 2024-05-{1 + i // 8:02d}:[|User|]: {synthetic.caption(100 + i, words=8)}; "
                                f"[|AI|]: {synthetic.caption(200 + i, words=10)}", {"source": f"2024-05-{1 + i // 8:02d}"}) for i in range(32)]
        self.question = "where did I leave the red cup and what was on the kitchen table"
        self.model = None
        if with_llm:        # LongVA-7B language side: Qwen2-7B shape, random-init fp16 (15 GB), KV cache for 52k tokens
            qc = LM.Qwen2ConfigLite(**LM.QWEN2_7B)
            sd = LM.random_qwen2_state_dict(qc, seed=4, device=device)
            self.model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, qc, device=device, max_seq=max_seq, consume=True), self.enc)
            del sd
            torch.cuda.empty_cache()
        self.llm_tok = synthetic.SyntheticTokenizer()
        self.last = {}

    # ---- stages ----
    def _tag(self, t):
        if ops.KernelTimer.active is not None:
            ops.KernelTimer.active.tag = t

    def encode(self):
        self._tag("encode")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if self.n:
            self.enc.encode_frames_u8(self.frames, out=self.feats)
        e1.record()
        self.enc_events.append((e0, e1))

    def dialogue_search(self):
        # dialogue memory (MiniLM, flat L2, k=1, rebuilt per round like memory_utils.py:76-83)
        lm = Q.LocalMemoryRetrieval()
        lm.init_cfg("minilm-l6", top_k=1, language="en", embedder=self.sent)
        store = Q.FlatL2VectorStore(self.docs, lm._embed_docs(self.docs), self.sent.embed_query)
        return lm.search_memory(self.question, store)

    def prefill(self, image_embeddings, caption):
        self._tag("prefill")
        qs = S.build_answer_prompt(self.question, caption, None)
        conv = S.conv_templates["qwen_1_5"].copy()
        conv.append_message(conv.roles[0], qs)
        conv.append_message(conv.roles[1], None)
        ids = tokenizer_image_token(conv.get_prompt(), self.llm_tok, -200, return_tensors="pt").unsqueeze(0)
        out = self.model.generate_with_image_embedding(ids, image_embeddings=[image_embeddings], modalities=["video"], do_sample=False,
                                                       max_new_tokens=1)
        rows = sum(int(t.shape[0]) for t in image_embeddings) if isinstance(image_embeddings, (list, tuple)) else int(image_embeddings.shape[0])
        self.last.update(first_token=out, context=rows + ids.shape[1] - 1)

    def step(self):
        """single GPU, through the reference-seam functions (streaming.updating_memory_buffer, utiles.fast_search_tree_...)"""
        self.encode()
        self._tag("select")
        bank = [self.feats[i:i + 1] for i in range(self.n)]
        cap, tok = synthetic.SyntheticCaptioner(self.device), synthetic.SyntheticTokenizer()
        if getattr(self, "captioner", None) is not None:      # --with-captions: the 7B model writes the chunk captions (batched) and the merge summary
            cap = self.captioner
        torch.manual_seed(0)                                  # init_idx = CPU randperm(T)[:K]  
        if self.kmeans_k:                                     # C1: ONE weighted_kmeans_feature(X[64,576,3584], 8) over all frames
            red, labels = U.weighted_kmeans_feature(self.feats, self.kmeans_k)
            self.last = dict(labels=labels, reduced=red)
            return self.last
        tree, short = S.updating_memory_buffer(bank, None, cap, tok, True, rng=np.random.RandomState(0), batch_captions=cap is getattr(self, "captioner", 0), **MEM)
        self._tag("retrieve")
        related, dates = self.dialogue_search()
        # caption-tree search (BERT-large CLS + cosine, strict > 0 rule, utiles.py:685-788)
        short_emb = U.cat_frames(short).view(-1, short[0].shape[-1])
        path_feats, path_text = U.fast_search_tree_multi_modal_with_embedding(tree, self.question, short_emb, self.colbert, self.tok,
                                                                              cache=U.CaptionEmbeddingCache())
        self.last = dict(tree=tree, short=short, related=related, path_text=path_text, path_feats=path_feats)
        if self.model is not None:       # [short | retrieved] pieces go straight into the spliced prompt (no torch.cat of the 350 MB block)
            self.prefill([short_emb] + [t.reshape(-1, t.shape[-1]) for t in path_feats], path_text[-1])
        return self.last

    def step_sharded(self):
        """N >= 1 ranks: encode and chunk captions are rank-local; the tree policy, the ONE merge-group k-means, the retrieval and
        the prefill are those of the single stream (sharded.ShardedMemory); only the selected rows cross xGMI (one all-gather)."""
        ctx = self.ctx
        self.encode()
        self._tag("select")
        cap, tok = synthetic.SyntheticCaptioner(self.device), synthetic.SyntheticTokenizer()
        torch.manual_seed(0)
        mem = SH.ShardedMemory(ctx, **MEM)
        tree, short = mem.update(self.feats, self.n_total, cap, tok, rng=np.random.RandomState(0))
        self._tag("retrieve")
        wanted, path_text, related = None, None, None
        if ctx.is_root:
            related, dates = self.dialogue_search()
            path, path_text = U.fast_search_tree_multi_modal_with_embedding(tree, self.question, self.feats, self.colbert, self.tok,
                                                                            cache=U.CaptionEmbeddingCache())
            wanted = list(short) + list(path)
        wanted = mem.broadcast_refs(wanted)
        sel = mem.fetch(wanted, dst=0)
        self.last = dict(tree=tree, wanted=[mem.frames_of(r) for r in wanted], path_text=path_text, related=related, mem=mem)
        if ctx.is_root:
            self.last["image_embeddings"] = image_embeddings = sel.reshape(-1, sel.shape[-1])
            if self.model is not None:
                self.prefill(image_embeddings, path_text[-1])
        return self.last

    # ---- C5: multi-round session over ONE growing stream ----
    def prepare_rounds(self, rounds, per_round, seed=1234):
        """round r = frames [r*per_round, (r+1)*per_round) of the global stream, dealt to the ranks by whole chunks; every round keeps
        its own feature bank (tree nodes of earlier rounds keep referring to it, as the reference's node tensors do, utiles.py:561)."""
        self.round_parts = DD.partition_chunks(per_round, MEM["chunk_size"], self.ctx.world)
        a, b = self.round_parts[self.ctx.rank]
        self.per_round = per_round
        self.round_frames = [torch.from_numpy(synthetic.frame_stream(b - a, seed=seed, start=r * per_round + a)).to(self.device) for r in range(rounds)]
        self.round_feats = [torch.empty((b - a, self.cfg.num_patches, 3584), dtype=torch.float16, device=self.device) for _ in range(rounds)]
        self.questions = [f"round {r}: where did I leave the {synthetic.VOCAB[(7 * r) % len(synthetic.VOCAB)]} and what was next to the "
                          f"{synthetic.VOCAB[(11 * r + 3) % len(synthetic.VOCAB)]}" for r in range(rounds)]

    def session(self, decode_tokens=64):
        ctx = self.ctx
        cap, tok = synthetic.SyntheticCaptioner(self.device), synthetic.SyntheticTokenizer()
        torch.manual_seed(0)
        rng = np.random.RandomState(0)
        mem = SH.ShardedMemory(ctx, **MEM)
        cache = U.CaptionEmbeddingCache()                    # caption embeddings persist across the rounds of a session
        log = []
        for r, (frames, feats) in enumerate(zip(self.round_frames, self.round_feats)):
            self._tag("encode")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if frames.shape[0]:
                self.enc.encode_frames_u8(frames, out=feats)
            e1.record()
            self.enc_events.append((e0, e1))
            self._tag("select")
            tree, short = mem.update(feats, self.per_round, cap, tok, rng=rng)
            self._tag("retrieve")
            self.question = self.questions[r]
            wanted, path_text = None, None
            if ctx.is_root:
                self.dialogue_search()
                path, path_text = U.fast_search_tree_multi_modal_with_embedding(tree, self.question, feats, self.colbert, self.tok, cache=cache)
                wanted = list(short) + list(path)
            wanted = mem.broadcast_refs(wanted)
            sel = mem.fetch(wanted, dst=0)
            rec = dict(wanted=[mem.frames_of(x) for x in wanted], path_text=path_text, top=[(n.depth, n.centroids.rows) for n in tree])
            if ctx.is_root and self.model is not None:
                self.prefill(sel.reshape(-1, sel.shape[-1]), path_text[-1])
                rec["context"] = self.last["context"]
                if decode_tokens > 0:
                    self._tag("decode")
                    lm = self.model.lm
                    if getattr(self, "_dg", None) is None or not self._dg.valid():
                        self._dg = LM.DecodeGraph(lm, max_new_tokens=max(decode_tokens, 16))
                    self._dg.start(int(self.last["first_token"][0, 0]))
                    rec["tokens"] = self._dg.run(decode_tokens)
            log.append(rec)
        self.last.update(rounds=log, mem=mem, tree=mem.tree)
        return self.last

    def decode_rate(self, n_tokens):
        """greedy decode tokens/s on the context left in the KV cache by the last step (reported separately from the metric)"""
        lm = self.model.lm
        g = LM.DecodeGraph(lm, max_new_tokens=max(n_tokens, 16))
        g.start(int(self.last["first_token"][0, 0]))
        g.capture()                                       # one-time graph capture, not part of the rate
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.run(n_tokens)
        torch.cuda.synchronize()
        return n_tokens / (time.perf_counter() - t0)


def cpu_baseline(pipe, n_cpu_frames, km_iters_gpu):
    """SURVEY §8(d), to the letter: the CPU restatement of the reference path (oracle/torch_ref.py: plain fp32 PyTorch — the arithmetic
    the reference's CPU path runs) timed on THIS box's host cores next to the GPU number, on a bounded sample:
      * thread sweep on a BATCH of 4 frames (32 / 64 / 128 / all cores, ascending, stopped once a count is > 30 % slower than the best:
        oversubscribed runs take minutes); the best count is used for everything below;
      * ViT-L + projector on a 64-frame slice of the stream (`n_cpu_frames`, = C1's whole encode);
      * C1 in full, once: that encode + the reference's [T,K,D]-broadcast k-means (utiles.py:294-318 formula) on its 64 x 2 064 384
        features, K = 8, run to its own exit;
      * the C2/C3 merge: ONE real iteration of the same formula at T = 400, K = 5, D = 2 064 384 (the [T,K,D] intermediate is 16.5 GB
        of host memory; skipped and scaled from the C1 run if the box has less than 96 GB free) x the Lloyd passes the GPU run took;
      * a REAL fp32 prefill of 2048 tokens through 2 Qwen2-7B-shape layers, extrapolated with SURVEY's flop model.
    C2/C3 frames/s = 1024 / (1024 x s-per-frame + merge k-means + prefill): linear scaling of the per-frame encode, stated in `sample`."""
    from oracle import torch_ref as R
    cores = os.cpu_count() or 1
    sd = {k: v.float().cpu() for k, v in pipe.sd_vit.items()}
    sp = {k: v.float().cpu() for k, v in pipe.sd_proj.items()}
    n_cpu_frames = min(n_cpu_frames, pipe.n)
    u8 = pipe.frames[:n_cpu_frames].cpu().numpy()
    x = torch.from_numpy(R.preprocess_u8(u8))
    t_all = time.time()
    enc = lambda xs: torch.cat([R.encode_images(sd, sp, xs[i:i + 16], heads=16, patch=14, num_layers=24) for i in range(0, xs.shape[0], 16)])
    sweep = {}
    with torch.no_grad():
        torch.set_num_threads(min(32, cores, R.host_cpu_budget()[1]))
        R.encode_images(sd, sp, x[:2], heads=16, patch=14, num_layers=24)                  # warm-up (thread pool, allocator, page-in)
        for th in sorted({t for t in (R.host_cpu_budget()[1], 32, 64, 128, cores) if t <= cores} or {cores}):     # (the cgroup quota first) ascending; stops once more threads clearly lose
            torch.set_num_threads(th)
            t0 = time.time()
            enc(x[:4])
            sweep[th] = round((time.time() - t0) / min(4, n_cpu_frames), 4)
            if sweep[th] > 1.3 * min(sweep.values()):
                break
        threads = min(sweep, key=sweep.get)
        torch.set_num_threads(threads)
        # ALL cores the process may use: the pool's hosts show 256 cores but run the container under a cgroup CPU quota of 16
        # (reported as `cpu_quota_cores`; tools/probe_host.py: 15.5 cores busy whatever is asked for), so the slice is dealt in batches of 4 frames to
        # one single-threaded worker process per usable core (oracle/torch_ref.parallel_plan / encode_frames_u8_parallel: same arithmetic per frame;
        # 4.7 core-seconds per frame against 8-9 for multi-threaded forwards whose waiting threads burn the quota; worker start-up timed separately)
        workers, wthreads = R.parallel_plan(n_cpu_frames, batch=4)
        t0 = time.time()
        tm = {}
        feats = R.encode_frames_u8_parallel(sd, sp, u8, workers=workers, threads=wthreads, batch=4, timing=tm)      # the 64-frame slice = C1's encode
        t_enc_wall = time.time() - t0
        # steady state (first worker up -> last worker done): the start-up of the worker processes (spawn, `import torch`, sharing the fp32 weights) is a
        # fixed cost that must not be multiplied by 1024 / 64 in the C2 / C3 extrapolation; both are reported
        t_enc = tm.get("steady_s", t_enc_wall)
        quota = R.host_cpu_budget()[1]
    t_frame = t_enc / n_cpu_frames
    # ---- C1 in full: k-means(k=8) over the 64 encoded frames with the reference's broadcast formula, to its own exit ----
    Xs = feats.reshape(n_cpu_frames, -1)
    torch.manual_seed(0)
    init8 = torch.randperm(n_cpu_frames)[:8].tolist()
    t0 = time.time()
    _, _, _, it8 = R.weighted_kmeans_reference_formula(Xs, 8, init8, [0] * 80, max_iter=10)
    t_km_c1 = time.time() - t0
    t_iter_c1 = t_km_c1 / (it8 + 1)
    c1 = n_cpu_frames / (t_enc + t_km_c1)
    # ---- the T = 400, K = 5 merge: one REAL iteration (features of the GPU run's merge group, fp32 on the host) ----
    free_gb = None
    try:
        import psutil
        free_gb = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        pass
    km_note = ""
    if pipe.n >= 400 and free_gb is not None and free_gb > 96:
        Xm = pipe.feats[:400].reshape(400, -1).float().cpu()
        torch.manual_seed(0)
        t0 = time.time()
        R.weighted_kmeans_reference_formula(Xm, 5, torch.randperm(400)[:5].tolist(), [0] * 50, max_iter=1)
        t_iter_merge = time.time() - t0
        del Xm
        km_note = f"ONE real iteration at T=400,K=5,D=2064384: {t_iter_merge:.2f} s"
    else:
        t_iter_merge = t_iter_c1 * (400 * 5) / (n_cpu_frames * 8)
        km_note = (f"T=400,K=5 iteration scaled from the C1 run by T*K ({t_iter_merge:.2f} s; the 16.5 GB [T,K,D] intermediate was not "
                   f"attempted: {free_gb and round(free_gb)} GB of host memory free)")
    km = km_iters_gpu * t_iter_merge
    del Xs, feats
    # answer: a REAL fp32 prefill of 2048 tokens through 2 Qwen2-7B-shape layers (+ final norm / lm_head), extrapolated with the
    # flop model 2*N*6.53e9 + 2*N^2*3584*28 at the rate measured on that run
    t_prefill, note = 0.0, ""
    if pipe.model is not None and pipe.last.get("context"):
        qc = LM.Qwen2ConfigLite(**dict(LM.QWEN2_7B, layers=2))
        sdq = {k: v.float().cpu() for k, v in LM.random_qwen2_state_dict(qc, seed=4, device=pipe.device).items() if "embed_tokens" not in k and "lm_head" not in k}
        sdq["lm_head.weight"] = torch.zeros(8, qc.hidden)       # 8-row head: the timed work is the decoder layers
        n0 = 2048
        emb = torch.randn(n0, qc.hidden) * 0.02
        with torch.no_grad():
            t0 = time.time()
            R.qwen2_logits(sdq, emb, heads=qc.heads, kv_heads=qc.kv_heads, layers=2, head_dim=qc.head_dim)
            dt = time.time() - t0
        per_layer = (6.53e9 - 152064 * 3584) / 28
        flops0 = 2 * (2 * n0 * per_layer + 2 * n0 * n0 * 3584)          # the reference's HF attention computes the full N x N scores
        rate = flops0 / dt
        n = pipe.last["context"]
        flops = 2 * n * 6.53e9 + 2 * n * n * 3584 * 28 * 0.5 * 2
        t_prefill = flops / rate
        note = (f"; 7B prefill: 2 Qwen2-7B-shape layers x {n0} tokens fp32 measured ({dt:.1f} s = {rate / 1e12:.2f} TFLOP/s), "
                f"{n} tokens x 28 layers extrapolated with the flop model: {t_prefill:.0f} s")
    n_frames = FRAMES
    host_s = time.time() - t_all
    sys.stderr.write(f"[cpu_baseline] {host_s:.1f} s of host work\n")
    total = n_frames * t_frame + km + t_prefill
    return dict(value=round(n_frames / total, 5), unit="frames/s", cores=min(wthreads * workers, cores), cores_present=cores, cpu_quota_cores=quota, kind="port",
                c1_frames_per_s=round(c1, 4), thread_sweep_s_per_frame=sweep, encode_workers=workers, threads_per_worker=wthreads, host_seconds=round(host_s, 1),
                sample=f"oracle/torch_ref ViT-L+projector fp32 on a {n_cpu_frames}-frame slice: {workers} worker processes x {wthreads} threads on the {cores} cores present "
                       f"(cgroup CPU quota of the container: {quota} cores), batches of 4 ({t_frame:.3f} s/frame for the whole host in steady state - first worker up to last worker done; {t_enc_wall:.1f} s incl. worker start-up; one process: "
                       f"{min(sweep.values()):.3f} s/frame at {threads} threads = best of the sweep {sweep} on a 4-frame batch) x{n_frames}; k-means and prefill legs: one process, "
                       f"{threads} threads; C1 RUN IN FULL: those {n_cpu_frames} frames + "
                       f"reference-formula k-means K=8 on their features to its exit ({it8 + 1} iterations, {t_km_c1:.1f} s) = {c1:.3f} frames/s; merge k-means: "
                       f"{km_note} x {km_iters_gpu} Lloyd passes (the GPU run's count)" + note + "; retrieval negligible")


class PowerSampler:
    """Package power and shader clock of this rank's GPU over the timed region, read from the amdgpu hwmon files (power1_input in uW,
    freq1_input = sclk in Hz, power1_cap) every 0.25 s by a host thread: every dominant kernel of the step runs on the board's power
    cap with the clock throttled below its 2.4 GHz boost, which is what `roofline.frac` against the nominal
    peak cannot show.  No GPU work, no subprocess; None if the files are not there."""

    def __init__(self, dev_index):
        import glob
        self.dir, self.samples, self._stop, self._th = None, [], False, None
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = "%04x:%02x:%02x." % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            want = None
        cands = []
        for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            if os.path.exists(os.path.join(h, "power1_input")) and os.path.exists(os.path.join(h, "freq1_input")):
                cands.append((os.path.basename(os.path.realpath(os.path.join(h, "..", ".."))), h))
        for pci, h in cands:
            if want and pci.startswith(want):
                self.dir = h
        if self.dir is None and len(cands) == 1:
            self.dir = cands[0][1]
        self._cands = [h for _, h in cands]

    @staticmethod
    def _read(path):
        with open(path) as f:
            return float(f.read().strip())

    def _loop(self):
        while not self._stop:
            try:
                if self.dir is None:        # PCI ids not exposed: the busy board is the one drawing the most power
                    self.dir = max(self._cands, key=lambda h: self._read(os.path.join(h, "power1_input")))
                self.samples.append((self._read(os.path.join(self.dir, "power1_input")) / 1e6, self._read(os.path.join(self.dir, "freq1_input")) / 1e6))
            except Exception:
                pass
            time.sleep(0.25)

    def __enter__(self):
        if self.dir is not None or self._cands:
            import threading
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._th is not None:
            self._th.join()

    def summary(self):
        if not self.samples:
            return None
        w = [x[0] for x in self.samples]
        c = [x[1] for x in self.samples]
        cap = None
        try:
            cap = self._read(os.path.join(self.dir, "power1_cap")) / 1e6
        except Exception:
            pass
        return dict(package_w_avg=round(sum(w) / len(w), 1), package_w_max=round(max(w), 1), cap_w=cap, sclk_mhz_avg=round(sum(c) / len(c)),
                    sclk_mhz_min=round(min(c)), boost_mhz=2400, samples=len(w), source="amdgpu hwmon power1_input / freq1_input, 0.25 s period, timed region only")


def relaunch_one_rank_per_gpu(n):
    """`python bench.py --gpus N` typed as a bare command (no torch.distributed.run around it): start the N ranks ourselves, exactly as
    the task brief's launcher would (one process per GPU, rendezvous on 127.0.0.1, a free port), pass the arguments through and hand
    back the job's exit code.  Rank 0's JSON line goes to this process's stdout unchanged."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and os.environ.get("SC_ALL_RANKS_ON_GPU0") != "1":
        sys.exit(f"bench.py --gpus {n}: this box exposes {have} GPU(s).  (Testing the N > 1 path on one GPU: SC_ALL_RANKS_ON_GPU0=1 "
                 f"SC_DIST_BACKEND=gloo; RCCL refuses two ranks on one device.)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write(f"[bench] --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd[1:9])} bench.py ...\n")
    return subprocess.call(cmd, env=dict(os.environ, SC_BENCH_SELF_LAUNCHED="1"))


def preflight(a, rank, world, local):
    """`bench.py --gpus N --preflight`: everything a first N-GPU run can trip over, checked in seconds, one JSON line from rank 0 (every rank
    exits non-zero on an error; a rank that never arrives fails the others after SC_DIST_TIMEOUT_S)."""
    import json as _json
    rec, err = dict(preflight="ok", n_gpus=world, rank=rank), None
    try:
        have = torch.cuda.device_count()
        rec["visible_gpus"] = have
        if have <= local:
            raise RuntimeError(f"rank {rank}: LOCAL_RANK {local} but only {have} GPU(s) visible")
        dev = torch.device(f"cuda:{local}")
        torch.cuda.set_device(dev)
        config = a.config or ("C2" if a.no_llm else "C3")
        n_total = {"C1": 64, "C4": 4096}.get(config, (a.frames or FRAMES) * (a.rounds if config == "C5" else world))
        parts = DD.partition_chunks(n_total if config != "C5" else n_total // a.rounds, 40, world)
        rec["partition_frames_per_rank"] = [b - a_ for a_, b in parts]
        free, total = torch.cuda.mem_get_info(dev)
        # rank 0 holds the 7B model, its KV cache and the retrieved rows next to its shard's feature bank; the others bank + encoder only
        need = (40 if rank == 0 else 12) * 2 ** 30 + (parts[rank][1] - parts[rank][0]) * 576 * 3584 * 2 * 2
        rec["free_hbm_gb"], rec["need_hbm_gb"] = round(free / 2 ** 30, 1), round(need / 2 ** 30, 1)
        if free < need:
            raise RuntimeError(f"rank {rank}: {free / 2 ** 30:.1f} GB of HBM free, the step needs about {need / 2 ** 30:.1f} GB")
        if world > 1:
            backend = os.environ.get("SC_DIST_BACKEND", "nccl")
            DD.init_process_group(backend, rank, world, dev)
            rec["collectives"] = DD.warm_up(DD.DistContext(rank, world, dev, backend))
            import torch.distributed as dist
            frees = [None] * world
            dist.all_gather_object(frees, rec["free_hbm_gb"])
            rec["free_hbm_gb_per_rank"] = frees
            dist.destroy_process_group()
    except Exception as e:                                    # noqa: BLE001 - reported as ONE line, exit code 3
        err = f"{type(e).__name__}: {e}"
    if err is not None:
        print(_json.dumps(dict(preflight="error", n_gpus=world, rank=rank, error=err)), flush=True)
        return 3
    if rank == 0:
        print(_json.dumps(rec), flush=True)
    return 0


def main():
    a = parse()
    if a.dp_lloyd:
        os.environ["SC_DP_LLOYD"] = "1"                   # (read by sharded.ShardedMemory; the argument reaches self-started ranks through sys.argv)
    if a.gpus > 1 and "LOCAL_RANK" not in os.environ and int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        sys.exit(relaunch_one_rank_per_gpu(a.gpus))       # not under a launcher (no LOCAL_RANK) and no N-rank job around us: start the ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("SC_ALL_RANKS_ON_GPU0") == "1":      # testing on a 1-GPU box: every rank of the job shares device 0
        local = 0
    if a.preflight:
        sys.exit(preflight(a, rank, world, local))
    if world > 1:
        torch.cuda.set_device(local)
        backend = os.environ.get("SC_DIST_BACKEND", "nccl")     # "gloo": host-staged collectives, for N > 1 runs on a 1-GPU box (tests)
        DD.init_process_group(backend, rank, world, torch.device(f"cuda:{local}"))       # explicit time-out (SC_DIST_TIMEOUT_S, 120 s) + device_id
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    ctx = DD.DistContext(rank, world, dev, os.environ.get("SC_DIST_BACKEND", "nccl") if world > 1 else "nccl")
    dist_warm = DD.warm_up(ctx)                               # every communicator (collective + one point-to-point pair per peer) exists before step 1
    config = a.config or ("C2" if a.no_llm else "C3")
    full = config in ("C3", "C4", "C5") and not a.no_llm
    if config == "C1":
        n_total, scaling = a.frames or 64, "strong"
    elif config == "C5":
        n_total, scaling = (a.frames or FRAMES) * a.rounds, "strong"   # ONE ego stream: `rounds` segments of 1024 frames over the N ranks
    elif config == "C4":
        n_total, scaling = a.frames or 4096, "strong"          # ONE 4096-frame stream over the N ranks
    else:
        n_total, scaling = (a.frames or FRAMES) * world, "weak"   # 1024 frames per GPU
    sharded = world > 1 or a.force_sharded
    if config == "C1" and sharded:
        sys.exit("C1 (64 frames, one k-means over all of them) is a single-GPU configuration")
    if config == "C5":
        # every top-level merged node adds one retrieved chunk (23 040 tokens) to the context (utiles.py:715-748): ~210 k tokens by round 8
        pipe = Pipeline(dev, 0, with_llm=full and rank == 0, ctx=ctx, max_seq=(a.rounds + 2) * 23040 + 8192)
        pipe.prepare_rounds(a.rounds, n_total // a.rounds)
        run_step = pipe.session
    else:
        pipe = Pipeline(dev, n_total, with_llm=full and rank == 0, ctx=ctx, kmeans_k=8 if config == "C1" else None)
        run_step = pipe.step_sharded if sharded else pipe.step

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        run_step()
    barrier()
    # the 1-GPU encode rate inside the same job (N > 1 only): rank 0 encodes its shard alone while the other ranks wait at the barrier,
    # so that 1 -> N encode scaling (north_star: ">= 6x frame-encode scaling at 8 GPUs") can be read off ONE record
    enc_solo = None
    if world > 1:
        solo_frames = pipe.round_frames[0] if config == "C5" else pipe.frames
        if rank == 0 and solo_frames.shape[0]:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            pipe.enc.encode_frames_u8(solo_frames, out=pipe.round_feats[0] if config == "C5" else pipe.feats)
            e1.record()
            torch.cuda.synchronize()
            enc_solo = solo_frames.shape[0] / (e0.elapsed_time(e1) / 1e3)
        barrier()
    pipe.enc_events.clear()
    t0 = time.perf_counter()
    with PowerSampler(local) as power, ops.KernelTimer() as kt:
        for _ in range(a.steps):
            run_step()
        barrier()
        dt = time.perf_counter() - t0
        prof = kt.summary()
        km_infos = [i.cpu().tolist() for i in kt.aux.get("kmeans_info", [])]
    t_enc = sum(e0.elapsed_time(e1) for e0, e1 in pipe.enc_events) / 1e3
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt, t_enc], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt, t_enc = float(tmax[0].item()), float(tmax[1].item())
        dist.barrier()
        dist.destroy_process_group()                      # the last collective: ranks > 0 leave now, rank 0 goes on to format the record alone
    if rank != 0:
        return
    ms_step = dt / a.steps * 1e3
    value = n_total * a.steps / dt

    def fam(suffix, tags=None):
        """(launches, ms, work) summed over the records '<tag>/<suffix>' (all tags, or the given ones)"""
        n = ms = w = 0
        for k, v in prof.items():
            t, _, s = k.rpartition("/")
            if s == suffix and (tags is None or t in tags):
                n, ms, w = n + v[0], ms + v[1], w + v[2]
        return n, ms, w

    def mfma_roof(name, rec):
        n, ms, w = rec
        return dict(bound="mfma", kernel=name, launches=n, avg_ms=round(ms / max(n, 1), 5), achieved=round(w / max(ms, 1e-9) / 1e9, 1),
                    peak=MFMA_PEAK_TF, unit="TFLOP/s", frac=round(w / max(ms, 1e-9) / 1e9 / MFMA_PEAK_TF, 4))

    n, ms, work = fam("k_gemm")
    traffic = None
    try:        # HBM bytes per launch from the committed PMC passes of this same command (separate --pmc runs; see the file's `correction` note)
        traffic = round(json.load(open(os.path.join(ROOT, TRAFFIC_PROFILE)))["kernels"]["k_gemm_all"]["hbm_bytes_per_launch"])
    except Exception:
        pass
    roof = mfma_roof("k_gemm_fat (+k_gemm256/k_gemm128/k_gemm_skinny for K%128, small-M and fp32-out shapes)", (n, ms, work))
    roof.update(traffic=traffic, traffic_source=TRAFFIC_PROFILE if traffic else None, flops_per_launch=round(work / max(n, 1)))
    # (context for `frac`, measured once with tools/probes/probe_mfma_energy.hip: a register-resident loop of v_mfma_f32_16x16x32_f16 on random operands - no LDS,
    #  no memory - issues one MFMA per 16.1 clocks and still delivers 0.78 of the nominal peak, because it alone puts the package on its 1400 W cap at 1.89 GHz)
    roof.update(power_cap_note="register-resident MFMA loop, random fp16 operands: 0.78 of peak at the 1400 W cap (1.89 GHz); with k_gemm_fat's LDS fragment reads 0.70 - "
                               "profiles/r05_run_aa_mfma_energy_probe.jsonl")
    # per-stage rooflines: the ViT and LLM halves of the two MFMA kernel families, the HBM-bound k-means (1x = SURVEY 8(d)'s algorithmic
    # bytes: one read of X per Lloyd iteration; 2x = what the two-pass kernel moves), decode below
    stages = dict(vit_gemm=mfma_roof("k_gemm* (ViT-L + projector)", fam("k_gemm", ("encode",))),
                  vit_attention=mfma_roof("k_attn<64> S=577", fam("k_attn", ("encode",))))
    if full:
        stages.update(llm_gemm=mfma_roof("k_gemm* (Qwen2-7B prefill)", fam("k_gemm", ("prefill",))),
                      llm_attention=mfma_roof("k_attn<128> causal GQA", fam("k_attn", ("prefill",))))
    kn, kms, kw = fam("kmeans_fit")
    if kn:
        iters = sum(i[0] + 1 for i in km_infos) or kn           # assign(+update) passes actually run (device-side exit iteration + 1)
        gbs = kw / kn * iters / max(kms, 1e-9) / 1e6
        stages["kmeans"] = dict(bound="hbm", kernel="km2_pass (one read of X per Lloyd iteration) + first assign / last update", launches=kn, lloyd_passes=iters, avg_ms=round(kms / kn, 4),
                                achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4),
                                reads_of_X_per_lloyd_iteration=round((iters / kn + 1) / (iters / kn), 3),      # n + 1 reads for n iterations (rounds 1-5: 2 n)
                                frac_counting_every_read_of_X=round((iters / kn + 1) / (iters / kn) * gbs / HBM_PEAK_GBS, 4))
    if kn and world == 1 and pipe.feats.shape[0] >= 400:
        # the same merge k-means (T = 400, K = 5, D = 2 064 384, the step's own features) forced through all 10 Lloyd iterations (tol < 0): a
        # fixed amount of work whatever the stream does.  (The synthetic stream cuts its scenes at the chunk size, so the timed step's own run
        # converges after two passes.  Moving the cuts - tried in round 6 with 37-frame scenes - changes what the question retrieves: 39 778
        # context tokens instead of 48 994, i.e. another workload than BASELINE's C3 and rounds 1-5; the stream stays, THIS is the k-means figure,
        # and ten real Lloyd iterations at full size on a non-converging stream are in tests/test_gpu_composed_shipped.py)
        Xk = pipe.feats[:400].reshape(400, -1)
        init10 = list(range(0, 400, 80))
        ops.kmeans_fit(Xk, 5, init10, None, max_iter=10, tol=-1.0); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _, _, _, info10 = ops.kmeans_fit(Xk, 5, init10, None, max_iter=10, tol=-1.0); e1.record(); torch.cuda.synchronize()
        ms10 = e0.elapsed_time(e1)
        b1 = 400 * Xk.shape[1] * 2 + 2 * 5 * Xk.shape[1] * 4
        stages["kmeans_10_passes"] = dict(bound="hbm", kernel="km2_pass x 9 + first assign + last km_update", lloyd_passes=int(info10[0]) + 1, ms_total=round(ms10, 3), ms_per_pass=round(ms10 / 10, 4),
                                          achieved=round(b1 / (ms10 / 10) / 1e6, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(b1 / (ms10 / 10) / 1e6 / HBM_PEAK_GBS, 4),
                                          note="1x basis (T D 2 + 2 K D 4 bytes per Lloyd iteration); round 6: SC-KM2 + the one-read pass km2_pass (11 reads of X for 10 iterations instead of 20): profiles/r06_run_a_kmeans_one_read_pass.md; "
                                               "SC_KM_FUSED=0 runs the two-pass kernels (same bits)")
    per = {k: dict(launches=v[0], ms_per_step=round(v[1] / a.steps, 3)) for k, v in sorted(prof.items())}
    enc_fps = n_total * a.steps / max(t_enc, 1e-9)
    names = dict(C1="C1: 64-frame stream, ViT-L encode + ONE weighted_kmeans_feature(k=8) over all frames (no LLM)",
                 C2="C2: 1024-frame 336x336 stream per GPU, ViT-L/14-336(23 layers)+mlp2x_gelu encode, memory update (chunk 40, K 5, interval 10: one "
                    "k-means T=400), MiniLM flat-L2 + BERT-large-CLS tree retrieval",
                 C3="C3: 1024-frame 336x336 stream per GPU, ViT-L/14-336(23 layers)+mlp2x_gelu encode, memory update (chunk 40, K 5, interval 10: one "
                    "k-means T=400), MiniLM flat-L2 + BERT-large-CLS tree retrieval, LongVA-7B (Qwen2-7B shape) prefill of the retrieved context + first token",
                 C4="C4: ONE 4096-frame stream sharded over the ranks by whole chunks, encode + chunk captions rank-local, single-stream tree policy "
                    "(one k-means T=400), all-gather of the selected features, LongVA-7B prefill on rank 0",
                 C5=f"C5: ONE {n_total}-frame ego stream in {a.rounds} rounds of {n_total // a.rounds} frames: per round sharded encode, persistent short/long "
                    "memory tree (one merge k-means per round), BERT-large-CLS tree search + MiniLM dialogue memory, LongVA-7B prefill + 64-token decode on rank 0")
    per_gpu = n_total // world
    if config == "C5":
        metric = (f"frames/sec over a multi-round session (per round: encode+select+retrieve{'+7B prefill+64-token decode' if full else ''}), "
                  f"{n_total}-frame ego stream" + (f" sharded over {world} GPUs" if world > 1 else ""))
    elif config == "C4" and full:
        metric = f"frames/sec end-to-end (encode+select+retrieve+7B prefill), {n_total}-frame stream sharded over the ranks"
    elif full and world == 1:
        metric = "frames/sec end-to-end (encode+select+retrieve+7B prefill), 1024-frame stream"           # BASELINE.json's metric, verbatim
    elif full:          # the same workload per GPU; the line says how many frames the job processed in total
        metric = (f"frames/sec end-to-end (encode+select+retrieve+7B prefill), {n_total}-frame stream in total = {per_gpu} frames per GPU x "
                  f"{world} GPUs (ONE stream sharded by whole chunks)")
    else:
        metric = ("frames/sec end-to-end (encode+select" + ("" if config == "C1" else "+retrieve") + f"), {n_total}-frame stream"
                  + (f" in total = {per_gpu} frames per GPU x {world} GPUs" if world > 1 and scaling == "weak" else ""))
    out = dict(metric=metric, value=round(value, 2), unit="frames/s",
               n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling=scaling,
               vs_baseline=None, dtype="f16", data="synthetic",
               # frame-encode throughput of the whole job (max over ranks of the HIP-event time around each rank's encode stage): the
               # figure the ">= 6x frame-encode scaling at 8 GPUs" target is about; `value` additionally contains the stages below
               encode_frames_per_s=round(enc_fps, 1), encode_ms_per_step=round(t_enc / a.steps * 1e3, 2))
    if world > 1:
        out["encode_frames_per_s_1gpu_same_job"] = None if enc_solo is None else round(enc_solo, 1)
        out["scaling_note"] = (
            f"encode (ViT-L + projector) and chunk captions are rank-local: {n_total} frames over {world} GPUs, no data-path collective; "
            "select / retrieve run on the whole stream from metadata identical on every rank (one merge k-means on the rank owning the "
            "group, rows it lacks arrive point to point); the selected rows are GATHERED TO RANK 0 (one batch of point-to-point sends into "
            "their final position: only rank 0 allocates, only the rows it lacks move - `collective`); the 7B prefill"
            + (" + decode" if config == "C5" else "") + " is SERIAL on rank 0 (single-GPU, no TP) and its context (~49 k tokens) does not "
            "grow with the stream, so `value` scales with the frames per step while ms_per_step stays that of the serial tail + one "
            "rank's encode; read encode scaling from encode_frames_per_s against encode_frames_per_s_1gpu_same_job (rank 0 encoding its "
            "shard alone, other ranks idle at a barrier) or against the N = 1 record")
    if world > 1:
        out["dist_warm_up"] = dict(dist_warm, timeout_s=float(os.environ.get("SC_DIST_TIMEOUT_S", DD.DEFAULT_TIMEOUT_S)),
                                   note="communicators (one all-gather, one point-to-point round per peer) created before the warm-up steps")
    if world > 1 and pipe.last.get("mem") is not None:         # row bytes that crossed ranks in the LAST step (merge-group pieces + selected rows), from the Refs
        tr = pipe.last["mem"].traffic
        out["collective"] = dict(kind="gather-to-root (batch_isend_irecv)", fetches_with_traffic=tr["fetches"], bytes_moved_last_step=tr["bytes_moved"],
                                 note="round 4's all-gather of the selected rows moved world x max-rows-per-rank slots to EVERY rank")
        if pipe.last["mem"].dp_lloyd:
            out["collective"]["dp_lloyd"] = dict(fits_last_step=tr["dp_lloyd_fits"], bytes_moved_last_step=tr["dp_lloyd_bytes"],
                                                 what="merge-group k-means data-parallel over columns (sc_kmeans_fit_cols): rows transposed into column slabs "
                                                      "point to point, two fp64 segment tables all-gathered per Lloyd iteration, centroid slabs to the executor")
    out.update(config=dict(workload=names[config], context_tokens=pipe.last.get("context"), frames_total=n_total, frames_rank0=pipe.n,
                           frames_per_gpu=per_gpu, micro_batch=MICRO_BATCH, parallelism=f"dp{world}" + (" (sharded path)" if sharded else ""),
                           weights="random-init", launcher="self (bare command)" if os.environ.get("SC_BENCH_SELF_LAUNCHED") == "1" else
                           ("torch.distributed.run" if world > 1 else "single process")),
               roofline=roof, roofline_stages=stages, stages=per, power=power.summary(), build=ops.build_info())
    # rounds are compared on boxes whose chips hold 1.74 - 1.93 GHz under the same 1400 W cap (+-4 % on `value`): the MFMA-bound share of the
    # step (GEMM + attention launches of the encode and the prefill) rescaled to a 1900 MHz shader clock, the rest left as measured
    pw = out.get("power") or {}
    if pw.get("sclk_mhz_avg"):
        mfma_ms = sum(v["ms_per_step"] for k, v in per.items() if k.split("/")[0] in ("encode", "prefill") and k.split("/")[1] in ("k_gemm", "k_attn"))
        at1900 = mfma_ms * pw["sclk_mhz_avg"] / 1900.0 + (ms_step - mfma_ms)
        out["value_at_1900MHz"] = dict(value=round(n_total / at1900 * 1e3, 2), ms_per_step=round(at1900, 3), mfma_bound_ms_per_step=round(mfma_ms, 3), sclk_mhz_avg=pw["sclk_mhz_avg"],
                                       note="value with the MFMA-bound launches rescaled by sclk_avg / 1900 MHz: the figure to compare between boxes and rounds; `value` is what was measured")
    if pipe.last.get("path_text") is not None:      # what the question retrieved (and, on the sharded path, which global frames): equal for every GPU count
        import zlib
        sig = json.dumps(dict(path_text=list(pipe.last["path_text"]), wanted=pipe.last.get("wanted")))
        out["config"]["retrieval_crc32"] = zlib.crc32(sig.encode())
        if pipe.last.get("first_token") is not None:
            out["config"]["first_token"] = int(pipe.last["first_token"][0, 0])
    if config == "C5":      # per round: what was retrieved (digest of path text + global frame ids) and what was decoded - equal for every GPU count
        import zlib
        out["config"]["rounds"] = [dict(context=r.get("context"), top_level_nodes=len(r["top"]), frames_retrieved=sum(len(x) for x in r["wanted"]),
                                        retrieval_crc32=zlib.crc32(json.dumps(dict(path_text=list(r["path_text"]), wanted=r["wanted"])).encode()),
                                        tokens_crc32=None if r.get("tokens") is None else zlib.crc32(json.dumps([int(t) for t in r["tokens"]]).encode()))
                                   for r in pipe.last["rounds"]]
    if full and a.decode_tokens > 0 and world == 1 and config != "C5":
        rate = pipe.decode_rate(a.decode_tokens)         # greedy, batch 1, after the timed region (512 tokens)
        ctxlen = pipe.last["context"] + a.decode_tokens / 2
        gb_tok = 14.1 + 2 * 28 * 4 * 128 * ctxlen * 2 / 1e9       # fp16 weights incl. lm_head + KV bytes per token
        out["decode_tokens_per_s"] = round(rate, 2)
        out["decode_tokens"] = a.decode_tokens
        # BASELINE.json configs[2] in full: the step of the headline metric PLUS the 512-token answer, one after the other (the overlapped stream of
        # segments is the `session` object below)
        out["c3_with_decode_frames_per_s"] = round(n_total / (ms_step / 1e3 + a.decode_tokens / rate), 2)
        out["c3_with_decode_ms_per_step"] = round(ms_step + a.decode_tokens / rate * 1e3, 1)
        out["roofline_stages"]["decode"] = dict(bound="hbm", kernel="k_gemv / k_decode_qkv + k_attn_decode (hipGraph)", achieved=round(rate * gb_tok, 1), peak=HBM_PEAK_GBS,
                                                unit="GB/s", frac=round(rate * gb_tok / HBM_PEAK_GBS, 4), gb_per_token=round(gb_tok, 2))
        try:        # HBM bytes per token from the committed FETCH_SIZE pass over an eager token loop at the same context (rocprofv3's counters do not survive graph replays)
            tr = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_pmc_decode_traffic.json")))
            out["roofline_stages"]["decode"].update(traffic_gb_per_token=tr["fetched_GB_per_token"], traffic_source="profiles/r06_pmc_decode_traffic.json")
        except Exception:
            pass
    # (the contract's cpu_baseline before the optional side measurements: whatever happens in them, the line carries it)
    if not a.no_cpu_baseline and world == 1 and config in ("C2", "C3"):
        out["cpu_baseline"] = cpu_baseline(pipe, a.cpu_frames, max((i[0] + 1 for i in km_infos[-1:]), default=3))
    if a.session > 0 and full and world == 1 and config == "C3" and a.decode_tokens > 0:
        # a side measurement must never take the headline line with it: an exception is recorded, and a run that does not come back (two host
        # threads, two streams) is cut by a watchdog that prints the line as it stands and ends the process
        import threading

        def give_up():
            out["session"] = dict(error="timeout: the session measurement did not finish within 240 s; headline fields above are complete")
            print(json.dumps(out), flush=True)
            os._exit(0)
        dog = threading.Timer(240.0, give_up)
        dog.daemon = True
        dog.start()
        try:
            out["session"] = measure_session(pipe, n_total, a.session, a.decode_tokens, a.session_decode_cus)
        except Exception as e:
            out["session"] = dict(error=f"{type(e).__name__}: {e}"[:300])
        finally:
            dog.cancel()
            torch.cuda.empty_cache()
    if a.with_captions and full and world == 1 and config == "C3":
        # what the metric leaves out (SURVEY 8(f).1 "the true wall-clock dominator"), measured once on this code: the same step with the HIP
        # 7B model as the chunk captioner.  Separate object; `value` above is the headline metric and does not contain it.
        try:
            out["product"] = measure_product(pipe, a.with_captions, n_total)
        except Exception as e:                                   # the side measurement must never take the headline line with it
            out["product"] = dict(error=f"{type(e).__name__}: {e}"[:300])
        finally:
            pipe.captioner = None
            torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
