"""Host-side mirror of the reference's memory / selection helpers (`/root/reference/utiles.py`).

Same function names, argument order and return shapes as the reference so that
`inference_streaming_longva_v2.py` can switch imports and keep working; the arithmetic underneath
runs in hand-written gfx950 kernels (streamchat_amd.ops -> libstreamchat_hip.so).  Behavioural
quirks of the reference that callers can observe are reproduced and cited.
"""
import math
import random
from collections import defaultdict

import numpy as np
import torch

from . import ops

RED = "\033[31m"
GREEN = "\033[32m"
BLUE = "\033[34m"
RESET = "\033[0m"

VERBOSE = False   # the reference prints the whole tree after every update (utiles.py:491-503)


def _log(*a):
    if VERBOSE:
        print(*a)


class TreeNode:
    """reference utiles.py:41-46"""
    def __init__(self, centroids, labels=None, depth=0):
        self.centroids = centroids
        self.labels = labels
        self.children = []
        self.depth = depth


class MultimodalTreeNode:
    """reference utiles.py:48-56 (field names are API: .centroids .text .children .depth .labels)"""
    def __init__(self, centroids, text, text_distance=None, image_distance=None, labels=None, depth=0):
        self.centroids = centroids
        self.text = text
        self.text_distance = text_distance
        self.image_distance = image_distance
        self.labels = labels
        self.children = []
        self.depth = depth


def cat_frames(tensors):
    """torch.cat(list, dim=0) — but adjacent views of one contiguous feature bank are returned as a VIEW
    (the reference's cat copies up to 1.65 GB per merge; the feature bank here is one tensor)."""
    t0 = tensors[0]
    if all(t.is_contiguous() and t.dtype == t0.dtype and t.shape[1:] == t0.shape[1:] for t in tensors):
        row = t0[0].numel() * t0.element_size()
        ptr, ok = t0.data_ptr(), True
        for t in tensors:
            if t.data_ptr() != ptr or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr():
                ok = False
                break
            ptr += t.shape[0] * row
        if ok:
            n = sum(t.shape[0] for t in tensors)
            return torch.as_strided(t0, (n,) + tuple(t0.shape[1:]), t0.stride(), t0.storage_offset())
    return torch.cat(tensors, dim=0)


# ------------------------------------------------------------------------------------------------
# short-term memory: forgetting-curve sampling (host; reference utiles.py:251-262)
# ------------------------------------------------------------------------------------------------
def calculate_forgetting_probabilities(length, tau=10):
    """p_t = exp(-t/tau) / sum, t = 0 is the OLDEST frame of the window (Q6)  — utiles.py:251-255"""
    t = np.arange(length)
    R_t = np.exp(-t / tau)
    return R_t / R_t.sum()


def select_data_without_replacement(queue, probabilities, selection_length=10, rng=None):
    """utiles.py:257-262.  `rng` (optional numpy Generator / RandomState) replaces the global numpy RNG
    the reference consumes; with rng=None the global `np.random.choice` is used exactly as upstream."""
    indices = np.arange(len(queue))
    chooser = np.random if rng is None else rng
    selected_indices = chooser.choice(indices, size=selection_length, replace=False, p=probabilities)
    return [queue[idx] for idx in selected_indices]


def compress_spatial_features(feature_list, compress_rate):
    """utiles.py:264-289: spatial avg-pool of [1, S, D] ViT maps (S a perfect square) by compress_rate."""
    assert len(feature_list) > 0, "compressed feature must larger then 0"
    all_features = torch.cat(feature_list)
    patch_size = round(math.sqrt(all_features.shape[1]))
    B, SEQ, DIM = all_features.shape
    assert patch_size * patch_size == feature_list[0].shape[1], \
        f"For ViT feature map, {patch_size}*{patch_size}={patch_size**2} != {all_features.shape[1]}"
    # one HIP kernel (sc_avgpool_tokens_f16: fp32 mean of the r x r window, one rounding); like every other op of the package there is no
    # CPU / PyTorch path - ops.avgpool_tokens refuses anything but CUDA fp16 (round 4 still carried a torch `.float().mean()` fallback here)
    from . import ops
    if all_features.dtype != torch.float16 or DIM % 8 != 0:
        raise ops.StreamChatHipError(f"compress_spatial_features: fp16 features with a width that is a multiple of 8 only (got {all_features.dtype}, {DIM})")
    return list(torch.split(ops.avgpool_tokens(all_features, compress_rate), 1))


# ------------------------------------------------------------------------------------------------
# selective-frame k-means (device; reference utiles.py:291-330)
# ------------------------------------------------------------------------------------------------
def weighted_kmeans_feature(img_feature, video_max_frames, weights=None, *, init_idx=None, reseed_idx=None,
                            tol=1e-4, max_iter=10, return_info=False):
    """Drop-in for reference utiles.py:291-330.

    img_feature [T, P, D] (CUDA, f16/bf16/f32); returns (reduced [K, P, D] in img_feature.dtype, labels [T] int64).
    If T <= video_max_frames the reference's 3-tuple (features, weights, [[ [i] ... ]]) is returned (Q2).

    Determinism: the reference draws its initial rows with `torch.randperm(device=...)` (:295) and its
    empty-cluster reseeds with `random.randint` (:313).  Both are explicit here: `init_idx` defaults to a
    CPU `torch.randperm(T)[:K]` (global CPU generator) and `reseed_idx` to max_iter*K draws of `random.randint(0, T-1)`
    from Python's GLOBAL `random` state, which IS advanced (by all max_iter*K draws — the reference advances it by one draw per
    empty cluster actually met, which the device-side loop cannot report without a host round trip per iteration; successive
    calls therefore see fresh reseed rows like upstream, but a caller that shares `random` with other code sees a different
    stream position than upstream after a call).  Arithmetic is fp32-canonical (the reference's fp16 distances overflow to inf
    — SURVEY.md §0 item 4)."""
    unit_weights = weights is None
    if weights is None:
        weights = torch.ones(img_feature.size(0), dtype=img_feature.dtype, device=img_feature.device)
    T, P, D = img_feature.shape
    T0 = video_max_frames
    if T <= T0:
        return img_feature, weights, [[[i] for i in range(T)]]
    X = img_feature.reshape(T, -1)
    if init_idx is None:
        init_idx = torch.randperm(T)[:T0]
    if reseed_idx is None:
        reseed_idx = [random.randint(0, T - 1) for _ in range(max_iter * T0)]
    # (the reference's default weights are ones (:292-293): the kernels' unweighted path is the same arithmetic bit for bit - 1 * x = x, sums of ones are
    #  counts - without a weight load per row and the sequential W sums: tests/test_gpu_kmeans.py::test_unit_weights_equal_no_weights)
    C, labels, wsum, info = ops.kmeans_fit(X, T0, init_idx, reseed_idx, weights=None if unit_weights else weights, max_iter=max_iter, tol=tol)
    reduced_feature = C.view(T0, P, D).to(img_feature.dtype)
    if return_info:
        return reduced_feature, labels, dict(centroids_f32=C.view(T0, P, D), wsum=wsum, info=info)
    return reduced_feature, labels


def k_means_clustering(X, num_clusters, max_iter=10, *, init_idx=None):
    """utiles.py:332-345: plain (unweighted) Lloyd on [N, D] rows; returns (centroids, labels).  Upstream stops when
    `torch.allclose(centroids, new_centroids)` (rtol 1e-5, atol 1e-8) and then keeps the OLD centroids; here the exit test is the
    kernels' sum_k ||dC_k||_2 < 1e-6 (a fixed point of Lloyd's iteration moves nothing, so both stop at the same iteration on
    converging data and the old / new centroids then agree to that tolerance).  An empty cluster's mean is NaN upstream (0/0); here
    it is re-seeded with row 0 — documented divergence, upstream's NaN centroid can never be assigned a point again."""
    if init_idx is None:
        init_idx = torch.randperm(X.size(0))[:num_clusters]
    C, labels, _, _ = ops.kmeans_fit(X, num_clusters, init_idx, None, max_iter=max_iter, tol=1e-6)
    return C.to(X.dtype), labels


# ------------------------------------------------------------------------------------------------
# long-term memory tree (host policy; reference utiles.py:489-620)
# ------------------------------------------------------------------------------------------------
def make_summary_prompt(caption_list, tokenizer, conv_templates=None):
    """utiles.py:505-523 (at most 10 captions: `order` has 10 entries, so interval <= 10)."""
    order = ["first", "second", "third", "fourth", "fifth", "sixth", "seventh", "eighth", "ninth", "tenth"]
    new_caption = ["The caption of the {} video clip is:{} \n".format(order[i], c) for i, c in enumerate(caption_list)]
    qs = " ".join(new_caption)
    qs = "You need to write a summary of the following, including as many key details as possible into one sentence." + qs
    if conv_templates is None:
        from .conversation import conv_templates as _ct
        conv_templates = _ct
    conv = conv_templates["qwen_1_5_summarize"].copy()
    conv.append_message(conv.roles[0], qs)
    conv.append_message(conv.roles[1], None)
    summarize_prompt = conv.get_prompt()
    summarize_ids = torch.tensor(tokenizer(summarize_prompt).input_ids, dtype=torch.long).unsqueeze(0)
    return summarize_ids


def get_summarize_depth(nodes, interval):
    """utiles.py:525-536: highest depth whose top-level count is a positive multiple of `interval` (Q9)."""
    depth_count = defaultdict(int)
    for node in nodes:
        depth_count[node.depth] += 1
    max_depth = max(depth_count.keys())
    for depth in range(max_depth, -1, -1):
        if depth_count[depth] % interval == 0 and depth_count[depth] > 0:
            return depth, depth_count
    return 0, depth_count


def plan_merge(nodes, interval):
    """Merge decision of utiles.py:567-574 on a top-level node list: index of the first node of the `interval` siblings to merge,
    or None.  Pure metadata (depths only) — shared by the single-stream builder below and the sharded one (sharded.py), so both
    take the same decision on the same global node list."""
    if len(nodes) == 0:
        return None
    summarize_depth, _ = get_summarize_depth(nodes, interval)
    start_index = next((index for index, node in enumerate(nodes) if node.depth == summarize_depth), None)
    chunk_length = len([x for x in nodes if x.depth == summarize_depth])
    _log("summarize_depth:{}/ start_index:{}/ chunk_length:{} / len(nodes):{}".format(summarize_depth, start_index, chunk_length, len(nodes)))
    return start_index if chunk_length >= interval else None


def caption_chunk(summarizer, tokenizer, input_ids, chunk_feature):
    """One chunk caption (utiles.py:539-559): the chunk's frames as image tokens, 128 new tokens, temperature 0.1.
    `chunk_feature`: list of [1, P, D] frame tensors or one [n, P, D] tensor."""
    if isinstance(chunk_feature, (list, tuple)):
        chunk_feature = cat_frames(chunk_feature)
    chunk_feature = chunk_feature.reshape(-1, chunk_feature.shape[-1]).to(summarizer.device)
    with torch.no_grad():
        output_ids = summarizer.generate_with_image_embedding(
            input_ids.to(summarizer.device), image_embeddings=[chunk_feature], modalities=["video"],
            do_sample=True, temperature=0.1, top_p=None, max_new_tokens=128, use_cache=False)
    return tokenizer.batch_decode(output_ids, skip_special_tokens=True)[0].strip()


def summarize_captions(summarizer, tokenizer, caption_list, conv_templates=None):
    """Summary text of a merge (utiles.py:589-607): text-only prompt, 256 new tokens, temperature 0.1."""
    summarize_ids = make_summary_prompt(caption_list, tokenizer, conv_templates)
    with torch.no_grad():
        output_ids = summarizer.generate_with_image_embedding(
            summarize_ids.to(summarizer.device), image_embeddings=None, modalities=["video"],
            do_sample=True, temperature=0.1, top_p=None, max_new_tokens=256, use_cache=False)
    return tokenizer.batch_decode(output_ids, skip_special_tokens=True)[0].strip()


def fast_building_memory_tree_summarize_token(k_means_chunk_feature_list, num_clusters, interval, summarizer, input_ids,
                                              tokenizer, chunked_feature_list, existing_tree=None, conv_templates=None, batch_captions=False):
    """Drop-in for reference utiles.py:489-620: caption every new chunk with the LLM, append depth-0 nodes,
    then perform AT MOST ONE merge of `interval` sibling nodes (k-means over their concatenated frames).
    batch_captions=True: the chunks of this call are captioned by ONE batched generate (BatchDecoder: weights stream
    once per decode step for all chunks) when the summarizer offers `generate_batch_with_image_embedding`; with the reference's
    sampling settings (temperature 0.1) the texts then depend on the batch-wise RNG order, with do_sample=False they are identical."""
    if batch_captions and len(chunked_feature_list) > 1 and hasattr(summarizer, "generate_batch_with_image_embedding"):
        captions = _caption_chunks_batched(summarizer, tokenizer, input_ids, chunked_feature_list)
    else:
        captions = [caption_chunk(summarizer, tokenizer, input_ids, frames) for frames in chunked_feature_list]
    leaves = [MultimodalTreeNode(feature, text, depth=0) for feature, text in zip(k_means_chunk_feature_list, captions)]
    tree = list(existing_tree) + leaves if existing_tree else leaves
    first = plan_merge(tree, interval)
    if first is not None:
        tree[first:first + interval] = [merge_siblings(tree[first:first + interval], num_clusters, summarizer, tokenizer, conv_templates)]
    return tree


def _caption_chunks_batched(summarizer, tokenizer, input_ids, chunked_feature_list):
    """all chunk captions of one update from ONE batched generate (same prompt, sampling settings and token budget as caption_chunk)"""
    dev = summarizer.device
    rows = [[cat_frames(frames).reshape(-1, frames[0].shape[-1]).to(dev)] for frames in chunked_feature_list]
    with torch.no_grad():
        outs = summarizer.generate_batch_with_image_embedding([input_ids.to(dev)] * len(rows), rows, modalities=["video"],
                                                              do_sample=True, temperature=0.1, max_new_tokens=128)
    return [tokenizer.batch_decode(o, skip_special_tokens=True)[0].strip() for o in outs]


def merge_siblings(group, num_clusters, summarizer, tokenizer, conv_templates=None):
    """`interval` sibling nodes -> their parent (reference utiles.py:581-614): the siblings' frames concatenated and, when there are more than
    `num_clusters` of them, reduced by the weighted k-means; the parent's text is the LLM's summary of the siblings' texts; one level up."""
    frames = cat_frames([sib.centroids for sib in group])
    if frames.shape[0] > num_clusters:
        frames, _ = weighted_kmeans_feature(frames, num_clusters)
    parent = MultimodalTreeNode(frames, summarize_captions(summarizer, tokenizer, [sib.text for sib in group], conv_templates), depth=group[0].depth + 1)
    parent.children.extend(group)
    return parent


def count_nodes_by_depth(nodes):
    """utiles.py:1002-1011"""
    depth_count = defaultdict(int)
    for node in nodes:
        depth_count[node.depth] += 1
        stack = node.children.copy()
        while stack:
            current_node = stack.pop()
            depth_count[current_node.depth] += 1
            stack.extend(current_node.children)
    return depth_count


# ------------------------------------------------------------------------------------------------
# tree search (reference utiles.py:685-788)
# ------------------------------------------------------------------------------------------------
def _cls_embed(model, tokenizer, text, device):
    """CLS-pooled last hidden state of one text, fp32 [d] on `device` (reference `pooling(..., 'cls')` :687-696)."""
    ids = tokenizer(text, padding=True, return_tensors="pt")
    ids = {k: (v.to(device) if hasattr(v, "to") else v) for k, v in ids.items()}
    out = model(**ids).last_hidden_state
    return out[0, 0].detach().to(torch.float32)


class CaptionEmbeddingCache:
    """Caption -> embedding table.  The reference re-encodes every visited caption on every query, one text per
    forward with a host sync each (utiles.py:696,721-732); captions never change once a node exists, so they are
    encoded once, and all uncached captions of a sibling set go through the encoder as ONE padded batch."""
    def __init__(self):
        self.table = {}

    def get_many(self, model, tokenizer, texts, device, batch=True):
        todo = [t for t in dict.fromkeys(texts) if (id(model), t) not in self.table]
        if todo and batch and len(todo) > 1:
            ids = tokenizer(todo, padding=True, return_tensors="pt")
            ids = {k: (v.to(device) if hasattr(v, "to") else v) for k, v in ids.items()}
            out = model(**ids).last_hidden_state[:, 0].detach().to(torch.float32)
            for t, e in zip(todo, out):
                self.table[(id(model), t)] = e
        return [self.get(model, tokenizer, t, device) for t in texts]

    def get(self, model, tokenizer, text, device):
        key = (id(model), text)
        e = self.table.get(key)
        if e is None:
            e = _cls_embed(model, tokenizer, text, device)
            self.table[key] = e
        return e


_caption_cache = CaptionEmbeddingCache()


def _best_positive(query_embedding, embeddings):
    """Reference selection rule (:717-741): running `sim > best_sim` from best_sim = 0, i.e. the FIRST
    maximum if it is strictly positive, else None.  Similarities + arg-best are computed on the device."""
    docs = torch.stack(embeddings)
    idx, score = ops.sim_topk(query_embedding, docs, k=1, metric="cos")
    i, s = int(idx[0].item()), float(score[0].item())
    return (i if s > 0 else None), s


def fast_search_tree_multi_modal_with_embedding(all_nodes, query, image_embedding, model, tokenizer, top_k=1, cache=None,
                                                batch_captions=True):
    """Drop-in for reference utiles.py:685-788: for every top-level node of depth > 0 walk down the
    best-cosine child at each level (appending the CHILD's features/text — Q8); among the depth-0
    top-level ("redundant") nodes append the best one.  Returns (path_features, path_text).

    Divergence (documented, Q7): where the reference would index `children[None]` because every
    similarity is <= 0 (:738-746), child 0 is used; the redundant-node rule (:751-777) already
    defaults to index 0 upstream."""
    cache = _caption_cache if cache is None else cache
    device = image_embedding.device if isinstance(image_embedding, torch.Tensor) else torch.device("cuda")
    path_features, path_text, redundant_nodes = [], [], []
    query_embedding = _cls_embed(model, tokenizer, query, device)

    for node in all_nodes:
        current_node = node
        if current_node.depth == 0:
            redundant_nodes.append(node)
            continue
        while current_node.children:
            embs = cache.get_many(model, tokenizer, [child.text for child in current_node.children], device, batch_captions)
            best_child_index, _ = _best_positive(query_embedding, embs)
            if best_child_index is None:
                best_child_index = 0
            path_features.append(current_node.children[best_child_index].centroids)
            path_text.append(current_node.children[best_child_index].text)
            current_node = current_node.children[best_child_index]

    if len(redundant_nodes) >= 1:
        embs = cache.get_many(model, tokenizer, [n.text for n in redundant_nodes], device, batch_captions)
        best_index, _ = _best_positive(query_embedding, embs)
        if best_index is None:
            best_index = 0
        path_features.append(redundant_nodes[best_index].centroids)
        path_text.append(redundant_nodes[best_index].text)
    return path_features, path_text


def search_tree(node, query, top_k=1):
    """Drop-in for reference utiles.py:909-935 (caller: inference_streaming_longva_v2.py:124, the non-multimodal answer path):
    walk from `node` (a TreeNode) to a leaf and return the `.centroids` of every visited node, root and leaf included.

    Upstream scores each child with `(query @ child.centroids.view(-1, d).T).sum()` and keeps the smaller one, but BOTH arms of
    its `if distance < best_distance: ... else: best_child_index = i` assign the loop index, so the child it descends into is
    always the LAST one, whatever the scores are (NaN included).  The scores therefore cannot influence the result and are not
    computed here; the golden fixture tests/golden/search_tree.json (the reference function's own output) pins this."""
    path_features = []
    current_node = node
    while current_node.children:
        path_features.append(current_node.centroids)
        current_node = current_node.children[len(current_node.children) - 1]
    path_features.append(current_node.centroids)
    return path_features


# ------------------------------------------------------------------------------------------------
# dialogue-memory prompt glue (reference utiles.py:1057-1078)
# ------------------------------------------------------------------------------------------------
def build_prompt_with_search_memory_only_related(text, user_name, user_memory_index, local_memory_qa, meta_prompt,
                                                 user_keyword, ai_keyword, boot_actual_name):
    memory_search_query = text.replace(user_keyword, user_name).replace(ai_keyword, "AI")
    if user_memory_index:
        related_memos, memo_dates = local_memory_qa.search_memory(memory_search_query, user_memory_index)
        related_memos = "\n".join(related_memos)
        related_memory_content = f"\n{str(related_memos).strip()}\n"
    else:
        related_memory_content = None
    if related_memory_content is not None:
        return meta_prompt.format(related_memory_content=related_memory_content)
    return None
