#!/usr/bin/env python3
"""Streaming video-QA entry point — MI355X-native drop-in for the reference's `inference_streaming_longva_v2.py`.

Same command line (reference :48-87) and the same per-video / per-question control flow (`run_inference`, reference :680-933):
  read + encode the segment -> update short/long memory -> retrieve dialogue + visual memory -> answer -> persist.
The functions it drives keep the reference's names (streamchat_amd.streaming / utiles / memory_bank / llm); everything under
them runs on hand-written gfx950 kernels.  Differences: ONE model replica serves both answering and chunk captioning (the
reference loads two 7B copies on cuda:0 / cuda:1, :697-700); generation uses a KV cache.

Two modes:
  * real run (default): `--model_name` is a LongVA-7B checkpoint directory in HF format (config.json, sharded safetensors / .bin,
    tokenizer), `--embedding_model_id` the mxbai-colbert-large-v1 directory (reference :703-705), `--sentence_model_id` the
    all-MiniLM-L6-v2 directory behind the dialogue memory (local_doc_qa.py:193), `--vision_tower` the CLIP directory when the
    checkpoint does not carry the tower.  They are loaded by streamchat_amd/checkpoint.py (what load_pretrained_model,
    longva/model/builder.py:27-285, does upstream); a missing or unreadable checkpoint is an ERROR, never a silent random init.
    Frames of any resolution are resized + centre-cropped on the host like CLIPImageProcessor does (utiles.py:71-87).
  * `--synthetic N`: N synthetic videos (seeded scene-structured frames, random-init weights of the real architectures, hash
    tokenizers) — there are no checkpoints or datasets in this environment."""
import argparse
import json
import os
import time

import numpy as np
import torch

from streamchat_amd import llm as LM, streaming as S, synthetic, text as T, utiles as U, vision as V
from streamchat_amd.memory_bank.memory_retrieval.local_doc_qa import HipSentenceEmbeddings, LocalMemoryRetrieval
from streamchat_amd.memory_bank.memory_utils import enter_name, save_local_memory
from streamchat_amd.memory_bank.prompt_utils import only_related_prompt_dict_ego


def parse_args(argv=None):
    """Command-line arguments of the reference (:48-87) + `--synthetic` / `--embedding_model_id` / `--tiny`."""
    p = argparse.ArgumentParser()
    p.add_argument("--video_dir", help="Directory containing video files.", required=True)
    p.add_argument("--model_name", type=str, required=True)
    p.add_argument("--conv-mode", type=str, required=False, default="video-chatgpt_v1")
    p.add_argument("--mode", type=str, required=False, default="off_line")
    p.add_argument("--chunk_size", type=int, default=20)
    p.add_argument("--num_clusters", type=int, default=5)
    p.add_argument("--interval", type=int, default=10)
    p.add_argument("--short_window", type=int, default=20)
    p.add_argument("--remember_window", type=int, default=5)
    p.add_argument("--tau", type=int, default=5)
    p.add_argument("--compress_rate", type=int, default=1)
    p.add_argument("--num_chunks", type=int, default=1)
    p.add_argument("--chunk_idx", type=int, default=0)
    p.add_argument("--num_frames", type=int, default=4)
    p.add_argument("--device", type=str, required=False, default="cuda:0")
    p.add_argument("--model-base", type=str, default=None)
    p.add_argument("--num_beams", type=int, default=1)
    p.add_argument("--temperature", type=float, default=0.2)
    p.add_argument("--sample_rate", type=float, default=0.5)
    p.add_argument("--top_p", type=float, default=None)
    p.add_argument("--memory_basic_dir", type=str, required=True, default="/Ours/memory_bank/memories")
    p.add_argument("--memory_file", type=str, required=False, default="updata_memories_for_streaming.json")
    p.add_argument("--save_file", type=str, required=True, default="result_for_streaming.json")
    p.add_argument("--annotations", type=str, required=True, default="result_for_streaming.json")
    p.add_argument("--language", type=str, required=True, default="en")
    p.add_argument("--memory_search_top_k", type=int, default=1)
    p.add_argument("--ppl", action="store_true", help="weather to calculate ppl")
    p.add_argument("--multi_modal_memory", action="store_true", help="weather to open multi-modal memory")
    # additions
    p.add_argument("--synthetic", type=int, default=0, help="run N synthetic videos instead of --annotations / --video_dir")
    p.add_argument("--embedding_model_id", type=str, default=None, help="mxbai-colbert-large-v1 checkpoint directory (reference :703)")
    p.add_argument("--sentence_model_id", type=str, default=None, help="all-MiniLM-L6-v2 directory (dialogue-memory embedder, local_doc_qa.py:193)")
    p.add_argument("--vision_tower", type=str, default=None, help="CLIP ViT-L/14-336 directory if the LongVA checkpoint carries no tower weights")
    p.add_argument("--tiny", action="store_true", help="tiny random-init architectures (smoke runs)")
    p.add_argument("--synthetic_breakpoints", type=int, default=2, help="questions per synthetic video (one every 60 s)")
    p.add_argument("--max_new_tokens", type=int, default=256)
    p.add_argument("--batch_captions", type=int, nargs="?", const=1, default=1, metavar="0|1",
                   help="caption all chunks of an update with one batched generate (SURVEY 8(f).1; default since round 6: a sequence samples the same "
                        "tokens batched or alone - per-sequence seeds - so the batch is only faster); 0 = one generate per chunk as the reference does")
    p.add_argument("--overlap", type=int, nargs="?", const=128, default=128, metavar="DECODE_CUS",
                   help="reader / updater of the NEXT segment on a second host thread and CU partition while the answer is decoded on DECODE_CUS CUs "
                        "(multiples of 32; the reference's three-thread design, previous_version/streaming_demo_llava_next_3.py:967-991; SURVEY 8(f).3).  "
                        "ON by default since round 6 (`--overlap 0`: one stage after the other on the whole chip, as the reference's batch entry point). "
                        "Same outputs as the serial run, greedy AND sampled: every sequence samples from its own seed (answers and captions take their "
                        "seeds from two separate generators seeded from torch.manual_seed), nothing is drawn from a generator shared by the two threads")
    p.add_argument("--memory_tree_dir", type=str, default=None,
                   help="persist the visual memory tree of every video here after each question (safetensors + JSON manifest; SURVEY 8(f).4)")
    args = p.parse_args(argv)
    if args.num_beams != 1 and (args.temperature or 0) > 0:
        # the reference forwards num_beams and do_sample = (temperature > 0) to HF generate (:252-256): with both that is HF's beam SAMPLING, which
        # this build does not have (deterministic beam search it has: streamchat_amd/beam.py).  Said HERE, before a model is loaded and a
        # video is read, not by a NotImplementedError out of the first answer (the shipped script runs num_beams 1, inference_streamchat_v0.3.sh:20)
        p.error(f"--num_beams {args.num_beams} with --temperature {args.temperature}: beam sampling is not implemented in streamchat_amd; "
                f"use --temperature 0 (deterministic beam search, HF semantics) or --num_beams 1")
    return args


class SyntheticCapture:
    """cv2.VideoCapture stand-in: `read_rgb(frame_number)` over a seeded scene-structured stream."""
    def __init__(self, n_frames, fps=2, size=336, seed=0, device="cpu"):
        self.frames = torch.from_numpy(synthetic.frame_stream(n_frames, seed=seed, h=size, w=size)).to(device)
        self.n, self.fps = n_frames, fps

    def read_rgb(self, i):
        return self.frames[i] if 0 <= i < self.n else None


class Cv2Capture:
    """cv2.VideoCapture adapter: seek + decode + BGR->RGB (reference :503-511), then the resize / centre-crop half of
    CLIPImageProcessor.preprocess on the host (utiles.py:71-87); rescale + normalise are fused into the GPU patch gather."""
    def __init__(self, path, image_size=336):
        import cv2                                           # host decode stays on the CPU (out of the GPU hot path)
        self.cv2, self.cap, self.image_size = cv2, cv2.VideoCapture(path), image_size
        self.n = int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT))
        self.fps = int(self.cap.get(cv2.CAP_PROP_FPS))

    def read_rgb(self, i):
        from streamchat_amd.mm_utils import resize_center_crop_u8
        self.cap.set(self.cv2.CAP_PROP_POS_FRAMES, i)
        ret, frame = self.cap.read()
        return resize_center_crop_u8(self.cv2.cvtColor(frame, self.cv2.COLOR_BGR2RGB), self.image_size) if ret else None


def build_models(args):
    """Real checkpoints (default) or, with --synthetic, random-init models of the real (or --tiny) architectures."""
    dev = args.device
    if not args.synthetic:
        from streamchat_amd import checkpoint as CK
        missing = [f for f, v in (("--embedding_model_id", args.embedding_model_id), ("--sentence_model_id", args.sentence_model_id)) if not v]
        if missing:
            raise SystemExit(f"a real run needs {', '.join(missing)} (checkpoint directories); use --synthetic N for random-init smoke runs")
        model, tokenizer, vc = CK.load_longva(args.model_name, device=dev, vision_tower_path=args.vision_tower)
        embedding_model, embedding_tokenizer = CK.load_bert(args.embedding_model_id, device=dev)
        minilm, minilm_tok = CK.load_bert(args.sentence_model_id, device=dev)
        sentence = HipSentenceEmbeddings(T.SentenceEmbedder(minilm), minilm_tok)
        return model, tokenizer, embedding_model, embedding_tokenizer, sentence, vc
    if args.tiny:
        vc = V.CLIPVisionConfigLite(hidden=128, layers=3, heads=2, intermediate=256, patch=14, image_size=56)
        qc = LM.Qwen2ConfigLite(hidden=256, layers=2, heads=4, kv_heads=2, intermediate=512, vocab=32768)
        bc = T.BertConfigLite(hidden=128, layers=2, heads=2, intermediate=256)
        mc = T.BertConfigLite(hidden=128, layers=2, heads=4, intermediate=256)
    else:
        vc, qc = V.CLIPVisionConfigLite(**V.VIT_L_336), LM.Qwen2ConfigLite(**LM.QWEN2_7B)
        bc, mc = T.BertConfigLite(**T.BERT_LARGE), T.BertConfigLite(**T.MINILM_L6)
    enc = V.FrameEncoder(V.CLIPVisionTower(V.random_clip_state_dict(vc, device=dev), vc, device=dev),
                         V.MMProjector(V.random_projector_state_dict(vc.hidden, qc.hidden, device=dev), device=dev))
    model = LM.LlavaQwenForCausalLM(LM.Qwen2Model(LM.random_qwen2_state_dict(qc, device=dev), qc, device=dev, max_seq=65536, consume=True), enc)
    embedding_model = T.BertEncoder(T.random_bert_state_dict(bc, seed=2, device=dev), bc, device=dev)
    sentence = HipSentenceEmbeddings(T.SentenceEmbedder(T.BertEncoder(T.random_bert_state_dict(mc, seed=3, device=dev), mc, device=dev)), T.HashTokenizer())
    return model, synthetic.SyntheticTokenizer(), embedding_model, T.HashTokenizer(), sentence, vc


def inference_thread_with_memory_and_dialogue_retrival_test(long_memory_tree_cache, short_memory_buffer_cache, fps, model, embedding_model, tokenizer,
                                                            embedding_tokenizer, time_line, num_frames, conv_mode, chat, memory_config, args,
                                                            save_file, question, labels, qa_class, time, output_loss=False, **generate_kwargs):
    """reference :588-677: dialogue-memory prompt -> multi-modal answer -> append to the results JSON."""
    with open(save_file, "r", encoding="utf-8") as f:
        existing_data = json.load(f)
    with torch.no_grad():
        searched_history = U.build_prompt_with_search_memory_only_related(
            question, memory_config["user_name"], memory_config["user_memory_index"], memory_config["local_memory_qa"],
            memory_config["only_related_prompt"], memory_config["user_keyword"], memory_config["ai_keyword"], memory_config["boot_actual_name"])
        output, process_time, generate_time = S.longva_inference_with_embedding_multi_modal(
            question, num_frames, conv_mode, model, embedding_model, tokenizer, embedding_tokenizer, chat, short_memory_buffer_cache,
            long_memory_tree_cache, searched_history, temperature=args.temperature, top_p=args.top_p, num_beams=args.num_beams,
            max_new_tokens=args.max_new_tokens, **generate_kwargs)
    existing_data.append({"time": time, "question": question, "label": labels, "predict": output, "class": qa_class, "process_time": process_time})
    with open(save_file, "w", encoding="utf-8") as f:
        json.dump(existing_data, f, ensure_ascii=False, indent=4)
    return output


class Lookahead:
    """--overlap: the reader + updater of the NEXT segment on a second host thread and the MFMA partition of the chip, started when the
    current answer's prefill is done, while the answer's token loop runs on the decode partition (streamchat_amd/session.py explains the
    partitioning).  The updater captions with its own view of the model (same weights, own KV cache and buffers: the reference keeps a
    second replica for that, :697-700); a segment's reader / updater do not depend on the previous answer, only its prompt does."""

    def __init__(self, model, decode_cus, device):
        import threading
        from streamchat_amd import ops
        ncu = ops.device_info()["cu_count"]
        dc = max(32, min(ncu - 32, int(decode_cus) // 32 * 32))
        self.s_hbm, self.s_mfma = ops.masked_stream(0, dc, device), ops.masked_stream(dc, ncu - dc, device)
        self.captioner = LM.LlavaQwenForCausalLM(model.lm.shared_view(), model.frame_encoder, model.eos_token_id)
        self.captioner.generation_config, self.captioner.config = model.generation_config, model.config
        self.s_full = torch.cuda.Stream(device)                  # the reader / updater take the whole chip once the answer beside them is out
        self.device, self.threading = device, threading
        self.thread, self.go, self.box, self.decoding, self.busy = None, None, None, False, False

    def arm(self, fn):
        """`fn()` -> (feature_bank, tree, short) of the next segment; runs once `fire()` has been called"""
        self.go, self.box = self.threading.Event(), {}

        self.busy = True

        def work():
            self.go.wait()
            from streamchat_amd import ops
            try:
                with torch.no_grad(), torch.cuda.device(self.device), torch.cuda.stream(self.s_mfma):
                    ops.move_to_stream_when(lambda: not self.decoding, self.s_full)
                    try:
                        self.box["out"] = fn()
                    finally:
                        ops.move_to_stream_when(None, None)
                    torch.cuda.current_stream().synchronize()
            except BaseException as e:          # noqa: BLE001 - re-raised by take()
                self.box["exc"] = e
            finally:
                self.busy = False                # (the answer's token loop moves back to the whole chip: see run_inference)
        self.thread = self.threading.Thread(target=work, name="streamchat-reader-updater", daemon=True)
        self.thread.start()

    def fire(self):
        if self.go is not None:
            self.decoding = True
            self.go.set()

    def answer_done(self):
        self.decoding = False

    def take(self):
        """the armed segment's result (None if nothing was armed)"""
        if self.thread is None:
            return None
        self.go.set()                            # (an answer that ended at its first token never fired)
        self.thread.join()
        self.thread = None
        if "exc" in self.box:
            raise self.box["exc"]
        return self.box["out"]


def run_inference(args):
    """reference :680-933"""
    main_device = args.device
    model, tokenizer, embedding_model, embedding_tokenizer, sentence, vc = build_models(args)
    conv_mode = args.conv_mode if args.conv_mode in S.conv_templates else "qwen_1_5"
    if args.synthetic:
        all_annotations = [{"info": {"video_path": f"synthetic_{i}", "class_1": "synthetic"},
                            "breakpoint": [{"time": 60 * (j + 1), "question": f"what happened around {synthetic.caption(j, words=3)}?", "answer": "n/a",
                                            "class": "synthetic"} for j in range(args.synthetic_breakpoints)]} for i in range(args.synthetic)]
    else:
        all_annotations = json.load(open(args.annotations, "r"))
    inference_count = 0
    look = Lookahead(model, args.overlap, main_device) if args.overlap else None
    # Sampling seeds by ROLE (round 6): the answers draw their per-sequence seeds from one CPU generator, the chunk captions / merge summaries
    # from another (`llm.draw_seeds`; the n-th token of a sequence is a pure function of its seed and n).  Caption calls are sequential whether
    # they run on this thread or on the look-ahead thread, answer calls likewise, so a seeded run samples the same tokens serially and with
    # --overlap, with and without --batch_captions - the two threads share no generator state.  `captioner` is a view of the
    # model for that role: same weights, same KV cache as `model` (the serial path captions with the model itself, as the reference's second
    # replica does, :697-700); the look-ahead thread's view has a cache of its own and shares the role's seed generator.
    base_seed = torch.initial_seed()
    model.seed_generator = torch.Generator().manual_seed((base_seed ^ 0x5DEECE66D) & (2 ** 62 - 1))
    captioner = LM.LlavaQwenForCausalLM(model.lm, model.frame_encoder, model.eos_token_id)
    captioner.generation_config, captioner.config = model.generation_config, model.config
    captioner.seed_generator = torch.Generator().manual_seed((base_seed ^ 0x2545F4914F6CDD1D) & (2 ** 62 - 1))
    if look is not None:
        import copy
        look.tokenizer = copy.deepcopy(tokenizer)           # a HF fast tokenizer is not re-entrant: the look-ahead thread encodes with a copy of its own
        look.captioner.seed_generator = captioner.seed_generator
    for anno in all_annotations:
        os.makedirs(args.memory_basic_dir, exist_ok=True)
        args.memory_file = "memory_{}.json".format(inference_count)
        memory_dir = os.path.join(args.memory_basic_dir, args.memory_file)
        save_file = args.save_file
        if not os.path.exists(memory_dir):
            json.dump({}, open(memory_dir, "w", encoding="utf-8"))
        if not os.path.exists(save_file):
            json.dump([], open(save_file, "w", encoding="utf-8"))
        language = args.language
        local_memory_qa = LocalMemoryRetrieval()
        local_memory_qa.init_cfg(embedding_model="minilm-l6", embedding_device=main_device, top_k=args.memory_search_top_k, language=language,
                                 embedder=sentence)
        only_related_prompt = only_related_prompt_dict_ego()[language]
        memory = json.loads(open(memory_dir, "r", encoding="utf-8").read())
        user_name = "User"
        hello_msg, user_memory, memory, user_name, user_memory_index = enter_name(user_name, memory, local_memory_qa, args)
        memory_config = dict(user_memory=user_memory, user_name=user_name, user_memory_index=user_memory_index, local_memory_qa=local_memory_qa,
                             only_related_prompt=only_related_prompt, user_keyword="[|User|]", ai_keyword="[|AI|]", boot_actual_name="AI",
                             language=language, memory=memory)
        question_list = anno["breakpoint"]
        time_line = [int(q["time"]) for q in question_list]
        if args.synthetic:
            cap = SyntheticCapture(int(time_line[-1] * 2), fps=2, size=vc.image_size, seed=inference_count, device=main_device)
        else:
            video_path = os.path.join(args.video_dir, anno["info"]["class_1"], anno["info"]["video_path"])
            assert os.path.exists(video_path), "{} not exist ".format(video_path)
            cap = Cv2Capture(video_path, vc.image_size)
        total_frames, frame_rate = cap.n, cap.fps
        frame_line = [0] + time_line
        long_memory_tree, short_memory_buffer = None, None
        segments = list(zip(question_list, frame_line[:-1], frame_line[1:]))

        def read_and_update(star, end, tree, short, summarizer, tok=None):
            tok = tokenizer if tok is None else tok
            feature_bank = S.video_reader_thread_with_embedding(cap, total_frames, frame_rate, None, model, star, end, main_device, args.sample_rate,
                                                                chunk_size=args.chunk_size)
            if len(feature_bank) > 0:
                tree, short = S.updating_memory_buffer(
                    feature_bank, tree, summarizer, tok, args.multi_modal_memory, short_window=args.short_window,
                    remember_window=args.remember_window, tau=args.tau, compress_rate=args.compress_rate, chunk_size=args.chunk_size,
                    num_clusters=args.num_clusters, interval=args.interval, batch_captions=args.batch_captions)
            return feature_bank, tree, short
        for si, (questions, star, end) in enumerate(segments):
            question, labels, qa_class = questions["question"], questions["answer"], questions["class"]
            ahead = look.take() if look is not None else None
            if ahead is not None:
                feature_bank, long_memory_tree, short_memory_buffer = ahead
            else:
                feature_bank, long_memory_tree, short_memory_buffer = read_and_update(star, end, long_memory_tree, short_memory_buffer, captioner)
            gen_kw = {}
            if look is not None and si + 1 < len(segments):
                # the next segment's reader / updater start when this answer's prefill is done, and the answer's token loop runs on the decode
                # partition for as long as they are busy (the last answer of a video has nothing beside it: whole chip)
                nxt = segments[si + 1]
                look.arm(lambda a=nxt[1], b=nxt[2], t=long_memory_tree, sh=short_memory_buffer: read_and_update(a, b, t, sh, look.captioner, look.tokenizer))
                gen_kw = dict(on_prefill_done=look.fire, decode_stream=look.s_hbm)
                from streamchat_amd import ops
                ops.move_to_stream_when(lambda: look.decoding and not look.busy, torch.cuda.current_stream(torch.device(main_device)))
            try:
                output = inference_thread_with_memory_and_dialogue_retrival_test(
                    long_memory_tree, short_memory_buffer, frame_rate, model, embedding_model, tokenizer, embedding_tokenizer, time_line,
                    args.num_frames, conv_mode, None, memory_config, args, save_file, question, labels, qa_class, questions["time"], **gen_kw)
            finally:
                if look is not None:
                    from streamchat_amd import ops
                    ops.move_to_stream_when(None, None)
                    look.answer_done()
            # persist the dialogue turn and refresh the retrieval index (reference :918-920)
            memory = save_local_memory(memory, [[question, output]], user_name, args)
            _, _, memory, user_name, user_memory_index = enter_name(user_name, memory, local_memory_qa, args)
            memory_config["user_memory_index"] = user_memory_index
            if args.memory_tree_dir and long_memory_tree is not None:
                from streamchat_amd.persistence import save_memory_tree
                save_memory_tree(long_memory_tree, os.path.join(args.memory_tree_dir, f"video_{inference_count}"), short_memory_buffer,
                                 extra=dict(time=questions["time"], question=question))
        inference_count += 1


if __name__ == "__main__":
    run_inference(parse_args())
