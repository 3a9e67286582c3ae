#!/usr/bin/env python3
"""bench.py — headline benchmark of the streaming hot path on MI355X (driver contract in the task brief).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched under torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one synthetic 1024-frame 336x336 stream per GPU:
  encode   uint8 frames -> fused preprocess/patchify -> ViT-L/14-336 (23 layers) -> mlp2x_gelu projector   [MFMA]
  select   forgetting-curve short memory + chunking + memory-tree update incl. ONE whole-frame k-means
           (T = 400 frames, K = 5, D = 576*3584)                                                            [HBM]
  retrieve caption / dialogue embeddings -> cosine / flat-L2 top-k -> tree search                           [latency]
Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from streamchat_amd import dist as DD, llm as LM, ops, streaming as S, synthetic, text as T, utiles as U, vision as V   # noqa: E402
from streamchat_amd.memory_bank.memory_retrieval import local_doc_qa as Q   # noqa: E402

FRAMES = 1024
MICRO_BATCH = int(os.environ.get("SC_MICRO_BATCH", "512"))     # frames per ViT pass: 512 x 577 rows = 1154 whole 256-row GEMM tiles, 6.6 GB of activations
MEM = dict(chunk_size=40, num_clusters=5, interval=10, short_window=20, remember_window=5, tau=5)   # inference_streamchat_v0.3.sh:12-19
GFLOP_PER_FRAME = 385.1            # patch 0.69 + 23 x 15.88 + projector 19.03
MFMA_PEAK_TF = 2500.0              # dense fp16/bf16 MFMA, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-llm", action="store_true", help="C2 workload: stop after retrieval (no 7B prefill)")
    ap.add_argument("--decode-tokens", type=int, default=512, help="decode tokens measured AFTER the timed region (reported separately)")
    ap.add_argument("--cpu-frames", type=int, default=4)
    ap.add_argument("--force-sharded", action="store_true", help="exercise the N>1 code path with world size 1 (testing)")
    return ap.parse_args()


class Pipeline:
    def __init__(self, device, n_frames, seed, with_llm=True):
        cfg = V.CLIPVisionConfigLite(**V.VIT_L_336)
        self.cfg = cfg
        self.sd_vit = V.random_clip_state_dict(cfg, seed=0, device=device)
        self.sd_proj = V.random_projector_state_dict(1024, 3584, seed=1, device=device)
        self.enc = V.FrameEncoder(V.CLIPVisionTower(self.sd_vit, cfg, device=device), V.MMProjector(self.sd_proj, device=device), micro_batch=MICRO_BATCH)
        self.frames = torch.from_numpy(synthetic.frame_stream(n_frames, seed=seed)).to(device)        # resident in HBM
        self.feats = torch.empty((n_frames, cfg.num_patches, 3584), dtype=torch.float16, device=device)
        self.device = device
        self.n = n_frames
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
        if with_llm:        # LongVA-7B language side: Qwen2-7B shape, random-init fp16 (15 GB), KV cache for 64k tokens
            qc = LM.Qwen2ConfigLite(**LM.QWEN2_7B)
            sd = LM.random_qwen2_state_dict(qc, seed=4, device=device)
            self.model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(sd, qc, device=device, max_seq=53248, consume=True), self.enc)
            del sd
            torch.cuda.empty_cache()
        self.llm_tok = synthetic.SyntheticTokenizer()

    def step(self):
        # ---- encode ----
        self.enc.encode_frames_u8(self.frames, out=self.feats)
        bank = [self.feats[i:i + 1] for i in range(self.n)]
        # ---- select (memory update; captions come from the synthetic captioner, untimed-equivalent host work) ----
        cap, tok = synthetic.SyntheticCaptioner(self.device), synthetic.SyntheticTokenizer()
        torch.manual_seed(0)                                  # init_idx = CPU randperm(T)[:K]  
        tree, short = S.updating_memory_buffer(bank, None, cap, tok, True, rng=np.random.RandomState(0), **MEM)
        # ---- retrieve: dialogue memory (MiniLM, flat L2, k=1, rebuilt per round like memory_utils.py:76-83) ----
        lm = Q.LocalMemoryRetrieval()
        lm.init_cfg("minilm-l6", top_k=1, language="en", embedder=self.sent)
        store = Q.FlatL2VectorStore(self.docs, lm._embed_docs(self.docs), self.sent.embed_query)
        related, dates = lm.search_memory(self.question, store)
        # ---- retrieve: caption-tree search (BERT-large CLS + cosine, strict > 0 rule, utiles.py:685-788) ----
        short_emb = U.cat_frames(short).view(-1, short[0].shape[-1])
        path_feats, path_text = U.fast_search_tree_multi_modal_with_embedding(tree, self.question, short_emb, self.colbert, self.tok,
                                                                              cache=U.CaptionEmbeddingCache())
        self.last = dict(tree=tree, short=short, related=related, path_text=path_text, path_feats=path_feats)
        # ---- answer-time 7B prefill over [short | long] frame tokens + prompt (first answer token) ----
        if self.model is not None:
            long_emb = torch.cat([t.reshape(-1, t.shape[-1]) for t in path_feats], dim=0)
            image_embeddings = torch.cat([short_emb, long_emb], dim=0)
            qs = S.build_answer_prompt(self.question, path_text[-1], None)
            conv = S.conv_templates["qwen_1_5"].copy()
            conv.append_message(conv.roles[0], qs)
            conv.append_message(conv.roles[1], None)
            from streamchat_amd.mm_utils import tokenizer_image_token
            ids = tokenizer_image_token(conv.get_prompt(), self.llm_tok, -200, return_tensors="pt").unsqueeze(0)
            out = self.model.generate_with_image_embedding(ids, image_embeddings=[image_embeddings], modalities=["video"], do_sample=False,
                                                           max_new_tokens=1)
            self.last.update(first_token=out, context=int(image_embeddings.shape[0]) + ids.shape[1] - 1)
        return self.last

    def step_sharded(self, ctx):
        """N > 1: the stream of world*frames frames is dealt to the ranks by whole chunks.  Encode, chunk captions and the chunk-group
        k-means are rank-local; rank 0 searches the all-gathered node METADATA; only the selected frames (short memory + retrieved
        chunks) cross xGMI in one fixed-capacity all_gather; rank 0 prefills the 7B model."""
        parts = DD.partition_chunks(self.n * ctx.world, MEM["chunk_size"], ctx.world)
        a, b = parts[ctx.rank]
        self.enc.encode_frames_u8(self.frames[: b - a], out=self.feats[: b - a])
        bank = [self.feats[i:i + 1] for i in range(b - a)]
        cap, tok = synthetic.SyntheticCaptioner(self.device), synthetic.SyntheticTokenizer()
        cap.n = a // MEM["chunk_size"]                                     # global chunk numbering of the synthetic captions
        torch.manual_seed(0)
        tree, short = S.updating_memory_buffer(bank, None, cap, tok, True, rng=np.random.RandomState(0), **MEM)
        # ---- metadata to rank 0: per top-level node (depth, text, children (text, frame range)), + short-memory frame ids ----
        base = self.feats.data_ptr()
        fsz = self.feats[0].numel() * self.feats.element_size()
        gidx = lambda t: a + (t.data_ptr() - base) // fsz                     # global index of a bank view's first frame
        def desc(n):
            rng = (int(gidx(n.centroids)), int(gidx(n.centroids)) + n.centroids.shape[0]) if n.depth == 0 else None
            return dict(depth=n.depth, text=n.text, frames=rng, children=[desc(c) for c in n.children])
        meta = DD.gather_objects(ctx, dict(nodes=[desc(n) for n in tree], short=[int(gidx(t)) for t in short]))
        wanted = None
        if ctx.is_root:
            def proxy(d):
                n = U.MultimodalTreeNode(d["frames"], d["text"], depth=d["depth"])
                n.children = [proxy(c) for c in d["children"]]
                return n
            nodes = [proxy(d) for m in meta for d in m["nodes"]]
            lm = Q.LocalMemoryRetrieval()
            lm.init_cfg("minilm-l6", top_k=1, language="en", embedder=self.sent)
            lm.search_memory(self.question, Q.FlatL2VectorStore(self.docs, lm._embed_docs(self.docs), self.sent.embed_query))
            path, path_text = U.fast_search_tree_multi_modal_with_embedding(nodes, self.question, self.feats[0], self.colbert, self.tok,
                                                                            cache=U.CaptionEmbeddingCache())
            wanted = list(meta[-1]["short"]) + [f for rng in path for f in range(rng[0], rng[1])]
            self.last = dict(path_text=path_text)
        wanted = DD.broadcast_object(ctx, wanted)
        sel = DD.gather_selected_frames(ctx, self.feats[: b - a], (a, b), wanted, capacity=2 * MEM["chunk_size"] + MEM["remember_window"])
        if ctx.is_root and self.model is not None:
            image_embeddings = sel.reshape(-1, sel.shape[-1])
            qs = S.build_answer_prompt(self.question, self.last["path_text"][-1], None)
            conv = S.conv_templates["qwen_1_5"].copy()
            conv.append_message(conv.roles[0], qs)
            conv.append_message(conv.roles[1], None)
            from streamchat_amd.mm_utils import tokenizer_image_token
            ids = tokenizer_image_token(conv.get_prompt(), self.llm_tok, -200, return_tensors="pt").unsqueeze(0)
            out = self.model.generate_with_image_embedding(ids, image_embeddings=[image_embeddings], modalities=["video"], do_sample=False,
                                                           max_new_tokens=1)
            self.last.update(first_token=out, context=int(image_embeddings.shape[0]) + ids.shape[1] - 1)
        return self.last if ctx.is_root else None

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


def cpu_baseline(pipe, n_cpu_frames):
    """The CPU restatement of the reference path (oracle/torch_ref.py: plain fp32 PyTorch, the arithmetic the reference's CPU path
    runs) timed on this box's host cores on a bounded sample and scaled to the full workload (the scaling rule is in `sample`)."""
    from oracle import torch_ref as R
    cores = os.cpu_count() or 1
    threads = min(cores, 32)                      # PyTorch CPU GEMMs stop scaling (and oversubscribe) far below 256 threads
    torch.set_num_threads(threads)
    sd = {k: v.float().cpu() for k, v in pipe.sd_vit.items()}
    sp = {k: v.float().cpu() for k, v in pipe.sd_proj.items()}
    u8 = pipe.frames[:n_cpu_frames].cpu().numpy()
    x = (u8.astype(np.float64) * (1 / 255)).astype(np.float32)
    x = torch.from_numpy(np.ascontiguousarray(((x - np.asarray(ops.CLIP_MEAN, np.float32)) / np.asarray(ops.CLIP_STD, np.float32)).transpose(0, 3, 1, 2)))
    with torch.no_grad():
        R.encode_images(sd, sp, x[:1], heads=16, patch=14, num_layers=24)          # warm-up (thread pool, allocator)
        t0 = time.time()
        R.encode_images(sd, sp, x, heads=16, patch=14, num_layers=24)
    t_frame = (time.time() - t0) / n_cpu_frames
    # select: the reference's [T,K,D] broadcast k-means on 40 of the 400 merge-group frames, full width (cost is linear in T)
    Tsub = 40
    Xs = pipe.feats[:400:10].reshape(Tsub, -1).float().cpu()
    t0 = time.time()
    _, _, _, it = R.weighted_kmeans_reference_formula(Xs, 5, list(range(0, Tsub, Tsub // 5))[:5], [0] * 50, max_iter=2)
    t_km_iter = (time.time() - t0) / (it + 1) * (400 / Tsub)
    km_iters = 3
    # 7B prefill: flop model 2*N*6.53e9 + 2*N^2*3584*28 at the CPU's measured ViT GEMM rate
    t_prefill = 0.0
    note = ""
    if pipe.model is not None and pipe.last.get("context"):
        n = pipe.last["context"]
        flops = 2 * n * 6.53e9 + 2 * n * n * 3584 * 28 * 0.5 * 2
        cpu_rate = 385.1e9 / t_frame
        t_prefill = flops / cpu_rate
        note = f"; 7B prefill of {n} tokens extrapolated with the flop model at the measured CPU rate ({cpu_rate / 1e12:.2f} TFLOP/s): {t_prefill:.0f} s"
    total = pipe.n * t_frame + km_iters * t_km_iter + t_prefill
    return dict(value=round(pipe.n / total, 5), unit="frames/s", cores=threads, kind="port",
                sample=f"oracle/torch_ref ViT-L+projector fp32 on {n_cpu_frames} frames ({t_frame:.2f} s/frame, {threads} threads of {cores} cores) x{pipe.n}; "
                       f"reference-formula k-means T={Tsub} of 400, K=5, D=2064384 scaled x10 ({t_km_iter:.2f} s/iter x {km_iters} iters)" + note
                       + "; retrieval negligible")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    elif a.gpus > 1:
        sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    ctx = DD.DistContext(rank, world, dev, "nccl")
    # weak scaling: world * frames frames in total; the 7B model lives on rank 0 only (single-GPU LLM stage)
    pipe = Pipeline(dev, a.frames + MEM["chunk_size"], seed=1234 + rank, with_llm=(not a.no_llm) and rank == 0) if world > 1 else \
        Pipeline(dev, a.frames, seed=1234 + rank, with_llm=not a.no_llm)
    if world > 1:
        pipe.n = a.frames
    if a.force_sharded and world == 1:
        pipe = Pipeline(dev, a.frames + MEM["chunk_size"], seed=1234, with_llm=not a.no_llm)
        pipe.n = a.frames
    run_step = (lambda: pipe.step_sharded(ctx)) if (world > 1 or a.force_sharded) else pipe.step

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        run_step()
    barrier()
    t0 = time.perf_counter()
    with ops.KernelTimer() as kt:
        for _ in range(a.steps):
            run_step()
        barrier()
        dt = time.perf_counter() - t0
        prof = kt.summary()
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank != 0:
        return
    ms_step = dt / a.steps * 1e3
    value = a.frames * world * a.steps / dt
    n, ms, work = prof.get("k_gemm", (0, 0.0, 0.0))
    traffic = None
    try:        # HBM bytes per launch from the committed PMC passes of this same command (profiles/, see its `correction` note)
        traffic = round(json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"]["k_gemm_all"]["hbm_bytes_per_launch"])
    except Exception:
        pass
    roof = dict(bound="mfma", kernel="k_gemm_fat (+k_gemm256/k_gemm128/k_gemm_skinny for K%128, small-M and fp32-out shapes)", launches=n, avg_ms=round(ms / max(n, 1), 5),
                achieved=round(work / max(ms, 1e-9) / 1e9, 1), peak=MFMA_PEAK_TF, unit="TFLOP/s",
                frac=round(work / max(ms, 1e-9) / 1e9 / MFMA_PEAK_TF, 4), traffic=traffic,
                flops_per_launch=round(work / max(n, 1)))
    stages = {k: dict(launches=v[0], ms_per_step=round(v[1] / a.steps, 3)) for k, v in prof.items()}
    if "kmeans_fit" in prof:
        kn, kms, kw = prof["kmeans_fit"]
        stages["kmeans_fit"]["note"] = "whole Lloyd fit (assign+update per iteration)"
    full = not a.no_llm
    out = dict(metric="frames/sec end-to-end (encode+select+retrieve+7B prefill), 1024-frame stream" if full else
               "frames/sec end-to-end (encode+select+retrieve), 1024-frame stream", value=round(value, 2), unit="frames/s",
               n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="f16", data="synthetic",
               config=dict(workload=("C3" if full else "C2") + ": 1024-frame 336x336 stream per GPU, ViT-L/14-336(23 layers)+mlp2x_gelu encode, "
                                    "memory update (chunk 40, K 5, interval 10: one k-means T=400), MiniLM flat-L2 + BERT-large-CLS tree retrieval"
                                    + (", LongVA-7B (Qwen2-7B shape) prefill of the retrieved context + first token" if full else ""),
                           context_tokens=pipe.last.get("context"),
                           frames_per_gpu=a.frames, micro_batch=MICRO_BATCH, parallelism=f"dp{world}", weights="random-init"),
               roofline=roof, stages=stages)
    if full and a.decode_tokens > 0 and world == 1:
        out["decode_tokens_per_s"] = round(pipe.decode_rate(a.decode_tokens), 2)         # greedy, batch 1, after the timed region (512 tokens)
        out["decode_tokens"] = a.decode_tokens
    if not a.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(pipe, a.cpu_frames)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
