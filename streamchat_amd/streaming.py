"""Host-side mirror of the streaming entry point's hot functions
(`/root/reference/inference_streaming_longva_v2.py`): frame sampling + encode (:454-531), memory update
(:267-378) and answer-time retrieval/prompt assembly (:164-264).  Names and argument order follow the
reference; tensors live in HBM and all arithmetic runs in the HIP kernels behind streamchat_amd.ops."""
import numpy as np
import torch

from . import utiles as U
from .conversation import conv_templates

IMAGE_TOKEN_INDEX = -200            # longva/constants.py:8
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"


def sample_frame_indices(total_frames, frame_rate, start, end, sample_rate, chunk_size=30, clamp=(900, 200)):
    """Integer frame-index arithmetic of video_reader_thread_with_embedding (:470-495).
    `clamp=(900, 200)` reproduces the reference's `num_frame > 900 -> 200` guard (:484-485), which exists only
    because it encodes a whole segment as ONE batch with 25 retained hidden states;
    pass clamp=None to process every sampled frame (the encoder here is micro-batched)."""
    start_frame = max(0, int(start * frame_rate))
    end_frame = min(total_frames, int(end * frame_rate))
    total_frames_to_process = end_frame - start_frame
    num_frame = int(total_frames_to_process * sample_rate)
    if clamp is not None and num_frame > clamp[0]:
        num_frame = clamp[1]
    if total_frames_to_process <= chunk_size:
        return list(range(start_frame, end_frame))
    return [int(start_frame + i * total_frames_to_process / num_frame) for i in range(num_frame)]


def video_reader_thread_with_embedding(cap, total_frames, frame_rate, image_processor, model, start, end, device, sample_rate,
                                       chunk_size=30, clamp=(900, 200)):
    """Mirror of :454-531.  `cap` is anything with `read_rgb(frame_number) -> uint8 [H, W, 3]` (a cv2.VideoCapture
    adapter or a synthetic stream); frames are decoded on the host, shipped as uint8 and preprocessed + encoded on
    the GPU (`model.encode_frames_u8`), replacing the per-frame PIL -> numpy -> torch -> H2D float round trip of
    :503-516.  Returns the feature bank: a list of [1, 576, D] views into ONE contiguous tensor."""
    frame_indices = sample_frame_indices(total_frames, frame_rate, start, end, sample_rate, chunk_size, clamp)
    if len(frame_indices) == 0:
        return []
    first = cap.read_rgb(frame_indices[0])
    if first is None:
        return []
    if not isinstance(first, torch.Tensor) and hasattr(model, "frame_encoder") and torch.device(device).type == "cuda":
        # host decoder (cv2): decode -> pinned staging -> H2D -> encode as a bounded, overlapped pipeline (streamchat_amd/ingest.py)
        from .ingest import AsyncFrameIngest
        enc = model.frame_encoder
        tokens, d_out = enc.tower.cfg.num_patches + (0 if enc.tower.select_feature == "patch" else 1), enc.projector.d_out
        bank = torch.empty((len(frame_indices), tokens, d_out), dtype=torch.float16, device=device)

        def frames_iter():
            yield first
            for n in frame_indices[1:]:
                fr = cap.read_rgb(n)
                if fr is None:
                    return
                yield fr
        ing = AsyncFrameIngest(enc.encode_frames_u8, tuple(first.shape), micro_batch=min(64, len(frame_indices)), depth=2, device=device)
        with torch.no_grad():
            bs = ing.run(frames_iter(), bank)
        return [bank[i:i + 1] for i in range(bs)]
    frames = [first]
    for current_frame_number in frame_indices[1:]:
        fr = cap.read_rgb(current_frame_number)
        if fr is None:
            break
        frames.append(fr)
    batch = frames[0].new_empty((len(frames),) + tuple(frames[0].shape)) if isinstance(frames[0], torch.Tensor) else None
    if batch is None:
        batch = torch.from_numpy(np.stack(frames))
    else:
        torch.stack(frames, out=batch)
    with torch.no_grad():
        image_embedding = model.encode_frames_u8(batch.to(device, non_blocking=True))
    bs = image_embedding.shape[0]
    feature_bank = [image_embedding[i:i + 1] for i in range(bs)]
    assert len(feature_bank) == bs
    return feature_bank


def _captioning_ids(summarizer_model, summarizer_tokenzier):
    captioning = ("Please describe what you see in this video in as much detail as possible from a first-person perspective, "
                  "including the surrounding environment, what objects are there, etc.")
    if getattr(summarizer_model.config, "mm_use_im_start_end", False):
        qs = DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN + "\n" + captioning
    else:
        qs = DEFAULT_IMAGE_TOKEN + "\n" + captioning
    conv = conv_templates["qwen_1_5_ego"].copy()
    conv.append_message(conv.roles[0], qs)
    conv.append_message(conv.roles[1], None)
    from .mm_utils import tokenizer_image_token
    ids = tokenizer_image_token(conv.get_prompt(), summarizer_tokenzier, IMAGE_TOKEN_INDEX, return_tensors="pt")
    return ids.unsqueeze(0)


def updating_memory_buffer(buffer_cache, long_memory_tree, summarizer_model, summarizer_tokenzier, building_multi_modal_memory_tree,
                           short_window=20, remember_window=5, tau=5, compress_rate=1, chunk_size=30, num_clusters=5, interval=10,
                           rng=None, batch_captions=False):
    """Mirror of :267-378: short-term memory by forgetting-curve sampling over the last `short_window` frames,
    long-term memory by chunking the whole buffer and growing the caption tree (at most one k-means merge)."""
    captioning_input_ids = _captioning_ids(summarizer_model, summarizer_tokenzier)
    if len(buffer_cache) > short_window:
        waite_FIFO = buffer_cache[-short_window:]
    else:
        short_window = len(buffer_cache)
        waite_FIFO = buffer_cache
    remember_window_set = min(remember_window, len(waite_FIFO))
    forgetting_probs = U.calculate_forgetting_probabilities(short_window, tau=tau)
    short_memory_buffer = U.select_data_without_replacement(waite_FIFO, forgetting_probs, remember_window_set, rng=rng)

    chunk_feature_list = [buffer_cache[i:i + chunk_size] for i in range(0, len(buffer_cache), chunk_size)]
    # :347 — `len(chunk) > chunk_size` can never hold, so depth-0 nodes keep their raw frames (Q1)
    k_means_chunk_feature_list = [U.weighted_kmeans_feature(U.cat_frames(c), num_clusters)[0] if len(c) > chunk_size else U.cat_frames(c)
                                  for c in chunk_feature_list]
    long_memory_tree = U.fast_building_memory_tree_summarize_token(k_means_chunk_feature_list, num_clusters, interval, summarizer_model,
                                                                   captioning_input_ids, summarizer_tokenzier, chunk_feature_list,
                                                                   long_memory_tree, batch_captions=batch_captions)
    assert len(short_memory_buffer) > 0, "No memory ?"
    return long_memory_tree, short_memory_buffer


def build_answer_prompt(question, most_fine_grad_text, history_prompt, mm_use_im_start_end=False):
    """Prompt branches of longva_inference_with_embedding_multi_modal (:197-228, Q18): with history AND a retrieved caption the
    `<image>` sentinel sits inside the instruction sentence; with no history the caption is ignored and the prompt is
    `<image>\n{question}{notion}`; with history but no caption there is NO `<image>` (visual tokens are dropped by the splice)."""
    prm = ("In addition, the text caption memory information articles most relevant to the current problem is '{most_fine_grad_text}'. \
        The image information you currently see and recall in the {image_token} is equally important as the contextual information mentioned earlier. \
        Sometimes the contextual information does not contain a direct answer to the question. \
        You need to synthesize this information and give an answer to the following question:")
    notion = "DO NOT OUTPUT ANY EXPLANATORY TEXT THAT IS UNCERTAIN ABOUT THE CURRENT QUESTION."
    img = (DEFAULT_IM_START_TOKEN + DEFAULT_IMAGE_TOKEN + DEFAULT_IM_END_TOKEN) if mm_use_im_start_end else DEFAULT_IMAGE_TOKEN
    if history_prompt is not None:
        if most_fine_grad_text is not None:
            return history_prompt + prm.format(most_fine_grad_text=most_fine_grad_text, image_token=DEFAULT_IMAGE_TOKEN) + "\n" + question + notion
        return history_prompt + "\n" + question + notion
    return img + "\n" + question + notion


def longva_inference_with_embedding_multi_modal(question, num_frames, conv_mode, model, embedding_model, tokenizer, embedding_tokenizer, chat,
                                                short_memory_buffer_cache, long_memory_tree_cache, history_prompt=None, temperature=0.2,
                                                top_p=None, num_beams=1, max_new_tokens=256, search_cache=None, **generate_kwargs):
    """Mirror of :164-264: retrieve long-term memory for the question, concatenate [short | long] frame tokens, build the prompt,
    tokenise with the -200 sentinel, generate.  Returns (text, process_time, generate_time) like upstream; `temperature` etc. are
    explicit (upstream reads the global `args`)."""
    import time
    from .mm_utils import tokenizer_image_token
    short_memory_embedding = U.cat_frames(short_memory_buffer_cache).view(-1, short_memory_buffer_cache[0].shape[-1])
    time_0 = time.time()
    if long_memory_tree_cache is not None:
        long_memory_list, long_memory_text_list = U.fast_search_tree_multi_modal_with_embedding(
            long_memory_tree_cache, question, short_memory_embedding, embedding_model, embedding_tokenizer, cache=search_cache)
        most_fine_grad_text = long_memory_text_list[-1]
        # upstream: torch.cat([short, torch.cat(long)]) (:188-196); the pieces are handed over as they are and the <image> splice writes them
        # in this order into the prompt embeddings (llm.splice_image_embeddings) - same rows, one 350 MB copy less
        image_embeddings = [short_memory_embedding] + [t.reshape(-1, t.shape[-1]) for t in long_memory_list]
    else:
        image_embeddings, most_fine_grad_text = short_memory_embedding, None
    qs = build_answer_prompt(question, most_fine_grad_text, history_prompt, getattr(model.config, "mm_use_im_start_end", False))
    conv = conv_templates[conv_mode].copy()
    conv.append_message(conv.roles[0], qs)
    conv.append_message(conv.roles[1], None)
    input_ids = tokenizer_image_token(conv.get_prompt(), tokenizer, IMAGE_TOKEN_INDEX, return_tensors="pt").unsqueeze(0)
    time_1 = time.time()
    with torch.no_grad():        # (not inference_mode: the persistent activation buffers are reused outside)
        output_ids = model.generate_with_image_embedding(input_ids, image_embeddings=[image_embeddings], modalities=["video"],
                                                         do_sample=True if temperature > 0 else False, temperature=temperature, top_p=top_p,
                                                         num_beams=num_beams, max_new_tokens=max_new_tokens, use_cache=False, **generate_kwargs)
    outputs = tokenizer.batch_decode(output_ids, skip_special_tokens=True)[0].strip()
    time_2 = time.time()
    return outputs, time_1 - time_0, time_2 - time_1
