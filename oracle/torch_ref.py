"""TEST INFRASTRUCTURE — plain PyTorch fp32 restatement of the third-party arithmetic the reference calls.

The ViT / projector / BERT / Qwen2 math is NOT under /root/reference: it lives in transformers==4.37.2
(requirements.txt:149) behind the call sites clip_encoder.py:41,76 (CLIPVisionModel), multimodal_projector/
builder.py:41-48 (nn.Sequential), utiles.py:707,728 (AutoModel -> BertModel) and llava_qwen.py:29,46,155
(Qwen2ForCausalLM).  These functions restate the published algorithms with explicit state-dict weights (HF
parameter names) and are pinned against the installed transformers (5.15.0) modules by the golden fixtures
tests/golden/clip_tiny.npz, bert_tiny.npz, qwen2_tiny.npz (tools/make_golden_hf.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import math

import torch
import torch.nn.functional as F


# ---- fp16 storage points (round 6, tests/test_gpu_heavy_tail.py) ----
# `with storage(torch.float16):` rounds every tensor HF's half-precision modules WRITE (every nn.Linear / LayerNorm / softmax / activation /
# residual output of transformers' CLIP, Qwen2 eager paths) to that dtype and back: the arithmetic stays fp32 per op, the roundings are where
# an fp16 HF model has them.  It answers "how far is HF's own fp16 execution from the fp32 truth on THIS input" - the yardstick for the HIP
# path on heavy-tailed activations, where a bound relative to max|ref| says nothing about the ordinary channels.  Outside the context
# `_st` is the identity and every function below is the plain fp32 restatement it always was.
import contextlib

_STORE = [None]


def _st(x):
    return x if _STORE[0] is None else x.to(_STORE[0]).float()


@contextlib.contextmanager
def storage(dtype):
    _STORE[0] = dtype
    try:
        yield
    finally:
        _STORE[0] = None


def _ln(x, w, b, eps):
    return _st(F.layer_norm(x, (x.shape[-1],), w, b, eps))


def _mha(x, sd, pre, heads, mask=None, causal=False):
    B, S, D = x.shape
    dh = D // heads
    q = _st(F.linear(x, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"])).view(B, S, heads, dh).transpose(1, 2)
    k = _st(F.linear(x, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"])).view(B, S, heads, dh).transpose(1, 2)
    v = _st(F.linear(x, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"])).view(B, S, heads, dh).transpose(1, 2)
    s = _st(_st(q @ k.transpose(-1, -2)) * dh ** -0.5)
    if mask is not None:
        s = _st(s + mask)
    a = _st(_st(torch.softmax(s, dim=-1)) @ v)
    a = a.transpose(1, 2).reshape(B, S, D)
    return _st(F.linear(a, sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"]))


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess_u8(u8_nhwc, mean=CLIP_MEAN, std=CLIP_STD):
    """reference utiles.py:71-87 -> HF CLIPImageProcessor.preprocess on frames that are already 336x336 (resize and centre-crop
    are identities): transformers image_transforms.rescale (float64 product, cast to float32) then normalize (float32), CHW.
    numpy uint8 [N,H,W,3] -> float32 [N,3,H,W].  Pinned by tests/golden/preprocess.npz (the processor's own output)."""
    import numpy as np
    x = (np.asarray(u8_nhwc).astype(np.float64) * (1 / 255)).astype(np.float32)
    x = (x - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def clip_vision_hidden(sd, pixel_values, *, heads, patch, layers_run, eps=1e-5, prefix="vision_model."):
    """hidden state after `layers_run` encoder layers (HF hidden_states[layers_run]); [N, 1+P, D] fp32.
    HF CLIPVisionTransformer: patch Conv2d(no bias) -> [cls | patches] + position embedding -> pre_layrnorm ->
    layers of (LN1, MHA, +res, LN2, fc1, quick_gelu, fc2, +res)."""
    p = prefix
    w = sd[p + "embeddings.patch_embedding.weight"]
    x = _st(F.conv2d(pixel_values, w, stride=patch))                   # [N, D, gh, gw]
    N, D = x.shape[0], x.shape[1]
    x = x.flatten(2).transpose(1, 2)                                   # [N, P, D]
    cls = sd[p + "embeddings.class_embedding"].expand(N, 1, D)
    x = _st(torch.cat([cls, x], dim=1) + sd[p + "embeddings.position_embedding.weight"][None])
    x = _ln(x, sd[p + "pre_layrnorm.weight"], sd[p + "pre_layrnorm.bias"], eps)
    for i in range(layers_run):
        lp = f"{p}encoder.layers.{i}."
        h = _ln(x, sd[lp + "layer_norm1.weight"], sd[lp + "layer_norm1.bias"], eps)
        x = _st(x + _mha(h, sd, lp + "self_attn.", heads))
        h = _ln(x, sd[lp + "layer_norm2.weight"], sd[lp + "layer_norm2.bias"], eps)
        h = _st(F.linear(h, sd[lp + "mlp.fc1.weight"], sd[lp + "mlp.fc1.bias"]))
        h = _st(h * _st(torch.sigmoid(_st(1.702 * h))))                # quick_gelu
        x = _st(x + _st(F.linear(h, sd[lp + "mlp.fc2.weight"], sd[lp + "mlp.fc2.bias"])))
    return x


def mm_projector(sd, x, prefix=""):
    """mlp2x_gelu (reference multimodal_projector/builder.py:41-48): Linear -> GELU(erf) -> Linear; keys 0.* / 2.*"""
    h = _st(F.gelu(_st(F.linear(x, sd[prefix + "0.weight"], sd[prefix + "0.bias"]))))
    return _st(F.linear(h, sd[prefix + "2.weight"], sd[prefix + "2.bias"]))


def encode_images(sd_vit, sd_proj, pixel_values, *, heads, patch, num_layers, select_layer=-2):
    """reference llava_arch.py:179-184 + clip_encoder.py:46-79: hidden_states[select_layer][:, 1:] -> projector.
    hidden_states has num_layers+1 entries, so index -2 is the output of layer num_layers-1."""
    run = num_layers + 1 + select_layer if select_layer < 0 else select_layer
    h = clip_vision_hidden(sd_vit, pixel_values, heads=heads, patch=patch, layers_run=run)
    return mm_projector(sd_proj, h[:, 1:])


# ---- the same encode on every host core: W worker processes x `threads` intra-op threads (one fp32 forward does not scale past
# ~32 threads; a 256-core host finishes a 440-frame stream W times sooner).  Same arithmetic per frame: a batch is `batch` frames
# wherever it runs.  Used by the composed-parity tests at the shipped geometry and by bench.py's cpu_baseline (all-cores leg).
def _encode_worker(rank, workers, threads, sd_vit, sd_proj, u8, out, batch, heads, patch, num_layers, stamps=None):
    import time
    torch.set_num_threads(threads)
    n = u8.shape[0]
    nb = (n + batch - 1) // batch
    if stamps is not None:
        stamps[rank, 0] = time.time()          # the worker is up (spawn + `import torch` + unpickling are behind it): steady-state work starts here
    with torch.no_grad():
        for b in range(rank, nb, workers):
            s, e = b * batch, min(n, (b + 1) * batch)
            x = torch.from_numpy(preprocess_u8(u8[s:e].numpy()))
            out[s:e] = encode_images(sd_vit, sd_proj, x, heads=heads, patch=patch, num_layers=num_layers)
    if stamps is not None:
        stamps[rank, 1] = time.time()


def host_cpu_budget():
    """(cores present, cores this process may actually burn): the second number is the cgroup CPU quota where one is set (the GPU boxes of the pool show
    256 cores and `cpu.max = 1600000 100000`, i.e. 16 cores' worth of CPU time), else the scheduling affinity."""
    import os
    present, usable = os.cpu_count() or 1, len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            usable = min(usable, max(1, int(q) // int(per)))
    except Exception:
        try:
            q, per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()), int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                usable = min(usable, max(1, q // per))
        except Exception:
            pass
    return present, usable


def parallel_plan(n_frames, batch=4):
    """(workers, threads per worker) for encode_frames_u8_parallel on this host: as many single-threaded workers as the process may burn cores
    (host_cpu_budget: the cgroup quota, 16 on the pool's 256-core hosts).  Measured there (tools/probe_host.py, profiles/r04_run4_host_probe.md): the
    quota is what binds (15.5 cores busy whatever is asked for), so the plan that wastes the least CPU time wins - 16 workers x 1 thread 4.7 core-s per
    frame (0.31 s/frame incl. worker start-up on 64 frames), 1 x 16 threads 8.1, 16 x 16 threads 9.0 (threads waiting at barriers burn the quota)."""
    _, usable = host_cpu_budget()
    workers = max(1, min(usable, (n_frames + batch - 1) // batch))
    return workers, max(1, usable // workers)


def encode_frames_u8_parallel(sd_vit, sd_proj, u8, *, workers, threads, batch=8, heads=16, patch=14, num_layers=24, timing=None):
    """uint8 [N,H,W,3] (numpy) -> fp32 [N, P, d_out]: preprocess_u8 + encode_images in batches of `batch`, the batches dealt round-robin to
    `workers` spawned processes of `threads` threads each (weights, frames and the output live in shared memory)."""
    import os
    import torch.multiprocessing as mp
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")          # (inherited by the workers) waiting threads sleep instead of burning a CPU quota
    frames = torch.from_numpy(u8).share_memory_()
    sv = {k: v.detach().float().cpu().share_memory_() for k, v in sd_vit.items()}
    sp = {k: v.detach().float().cpu().share_memory_() for k, v in sd_proj.items()}
    img = u8.shape[1]
    out = torch.empty((u8.shape[0], (img // patch) ** 2, sp["2.weight"].shape[0]), dtype=torch.float32).share_memory_()
    workers = max(1, min(workers, (u8.shape[0] + batch - 1) // batch))
    stamps = torch.zeros((workers, 2), dtype=torch.float64).share_memory_()
    if workers == 1:
        _encode_worker(0, 1, threads, sv, sp, frames, out, batch, heads, patch, num_layers, stamps)
    else:
        mp.spawn(_encode_worker, args=(workers, threads, sv, sp, frames, out, batch, heads, patch, num_layers, stamps), nprocs=workers, join=True)
    if timing is not None:                        # `timing` (a dict): steady-state seconds = first worker up -> last worker done (start-up excluded)
        timing["steady_s"] = float(stamps[:, 1].max() - stamps[:, 0].min())
        timing["workers"] = workers
    return out


# ---------------------------------------------------------------------------------------------------------
# BERT (HF BertModel; call sites reference utiles.py:707,728 and local_doc_qa.py:193 via sentence-transformers)
# ---------------------------------------------------------------------------------------------------------
def bert_last_hidden(sd, input_ids, attention_mask, *, heads, layers, eps=1e-12, prefix=""):
    p = prefix
    B, L = input_ids.shape
    x = sd[p + "embeddings.word_embeddings.weight"][input_ids] + sd[p + "embeddings.token_type_embeddings.weight"][0] \
        + sd[p + "embeddings.position_embeddings.weight"][:L][None]
    x = _ln(x, sd[p + "embeddings.LayerNorm.weight"], sd[p + "embeddings.LayerNorm.bias"], eps)
    D = x.shape[-1]
    dh = D // heads
    neg = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
    for i in range(layers):
        lp = f"{p}encoder.layer.{i}."
        lin = lambda n, t: F.linear(t, sd[lp + n + ".weight"], sd[lp + n + ".bias"])
        q = lin("attention.self.query", x).view(B, L, heads, dh).transpose(1, 2)
        k = lin("attention.self.key", x).view(B, L, heads, dh).transpose(1, 2)
        v = lin("attention.self.value", x).view(B, L, heads, dh).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + neg, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, L, D)
        x = _ln(lin("attention.output.dense", a) + x, sd[lp + "attention.output.LayerNorm.weight"], sd[lp + "attention.output.LayerNorm.bias"], eps)
        f = F.gelu(lin("intermediate.dense", x))
        x = _ln(lin("output.dense", f) + x, sd[lp + "output.LayerNorm.weight"], sd[lp + "output.LayerNorm.bias"], eps)
    return x


def sentence_embedding(sd, input_ids, attention_mask, *, heads, layers):
    """sentence-transformers all-MiniLM-L6-v2: mean pooling over the attention mask, then L2 normalise."""
    h = bert_last_hidden(sd, input_ids, attention_mask, heads=heads, layers=layers)
    m = attention_mask[..., None].to(h.dtype)
    e = (h * m).sum(1) / m.sum(1).clamp(min=1e-9)
    return F.normalize(e, p=2, dim=1)


# ---------------------------------------------------------------------------------------------------------
# Qwen2 (HF Qwen2ForCausalLM; call site reference llava_qwen.py:155)
# ---------------------------------------------------------------------------------------------------------
def _rms(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return _st(w * _st((x.float() * torch.rsqrt(v + eps)).to(x.dtype)))      # HF Qwen2RMSNorm: fp32 inside, cast to the input dtype, then the weight


def qwen2_logits(sd, embeds, *, heads, kv_heads, layers, head_dim, theta=1e6, eps=1e-6, last_only=False, head_chunk=None, row_chunk=None):
    """logits [L, vocab] for inputs_embeds [L, H] (causal, positions 0..L-1): RMSNorm -> q/k/v (bias) -> rotate-half RoPE ->
    GQA causal attention -> o_proj -> +res -> RMSNorm -> down(silu(gate) * up) -> +res; final norm; lm_head.
    last_only: logits of the last position only ([vocab]); head_chunk: attention computed for that many heads at a time (same
    arithmetic per head; bounds the [heads, L, L] score tensor for contexts of ~10 k tokens in the composed tests); row_chunk: attention
    for that many query rows at a time against the keys up to the chunk's last row (the 49 k-token context of the shipped geometry)."""
    L, H = embeds.shape
    x = embeds
    pos = torch.arange(L, device=x.device, dtype=torch.float32)
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, device=x.device, dtype=torch.float32) / head_dim))
    fr = torch.outer(pos, inv)
    cos, sin = torch.cat([fr, fr], -1).cos(), torch.cat([fr, fr], -1).sin()

    cos, sin = _st(cos), _st(sin)                 # (HF hands cos / sin to apply_rotary_pos_emb in the activations' dtype)

    def rot(t):                                   # t [h, L, d]
        t1, t2 = t[..., : head_dim // 2], t[..., head_dim // 2:]
        return _st(_st(t * cos) + _st(torch.cat([-t2, t1], -1) * sin))
    mask = None if row_chunk else torch.full((L, L), float("-inf"), device=x.device).triu(1)
    for i in range(layers):
        p = f"model.layers.{i}."
        h = _rms(x, sd[p + "input_layernorm.weight"], eps)
        if last_only and row_chunk and i == layers - 1:
            # long contexts, last-position logits only: in the LAST layer every row still gives its K / V, but the query, the attention row,
            # the output projection and the MLP are needed for the final row alone (same numbers for that row; half the attention work)
            k = rot(F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"]).view(L, kv_heads, head_dim).transpose(0, 1))
            v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"]).view(L, kv_heads, head_dim).transpose(0, 1)
            q1 = F.linear(h[-1:], sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]).view(1, heads, head_dim).transpose(0, 1)
            t1, t2 = q1[..., : head_dim // 2], q1[..., head_dim // 2:]
            q1 = q1 * cos[-1:] + torch.cat([-t2, t1], -1) * sin[-1:]
            k = k.repeat_interleave(heads // kv_heads, dim=0)
            v = v.repeat_interleave(heads // kv_heads, dim=0)
            a = torch.softmax(q1 @ k.transpose(-1, -2) / math.sqrt(head_dim), dim=-1) @ v           # [heads, 1, d]: the last row sees every key
            x = x[-1:] + F.linear(a.transpose(0, 1).reshape(1, heads * head_dim), sd[p + "self_attn.o_proj.weight"])
            h = _rms(x, sd[p + "post_attention_layernorm.weight"], eps)
            x = x + F.linear(F.silu(F.linear(h, sd[p + "mlp.gate_proj.weight"])) * F.linear(h, sd[p + "mlp.up_proj.weight"]), sd[p + "mlp.down_proj.weight"])
            break
        q = _st(F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"])).view(L, heads, head_dim).transpose(0, 1)
        k = _st(F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])).view(L, kv_heads, head_dim).transpose(0, 1)
        v = _st(F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])).view(L, kv_heads, head_dim).transpose(0, 1)
        q, k = rot(q), rot(k)
        k = k.repeat_interleave(heads // kv_heads, dim=0)
        v = v.repeat_interleave(heads // kv_heads, dim=0)
        if row_chunk:                                  # long contexts: query rows r0..r1 against keys 0..r1 only (the masked rest is exp(-inf) = 0)
            a = torch.empty_like(q)
            for r0 in range(0, L, row_chunk):
                r1 = min(L, r0 + row_chunk)
                s = q[:, r0:r1] @ k[:, :r1].transpose(-1, -2) / math.sqrt(head_dim)
                s += torch.full((r1 - r0, r1), float("-inf"), device=x.device).triu(r0 + 1)
                a[:, r0:r1] = torch.softmax(s, dim=-1) @ v[:, :r1]
        elif head_chunk:
            a = torch.cat([torch.softmax(q[j:j + head_chunk] @ k[j:j + head_chunk].transpose(-1, -2) / math.sqrt(head_dim) + mask, dim=-1) @ v[j:j + head_chunk]
                           for j in range(0, heads, head_chunk)])
        else:
            a = _st(_st(torch.softmax(_st(_st(_st(q @ k.transpose(-1, -2)) / math.sqrt(head_dim)) + mask), dim=-1)) @ v)
        x = _st(x + _st(F.linear(a.transpose(0, 1).reshape(L, heads * head_dim), sd[p + "self_attn.o_proj.weight"])))
        h = _rms(x, sd[p + "post_attention_layernorm.weight"], eps)
        m = _st(_st(F.silu(_st(F.linear(h, sd[p + "mlp.gate_proj.weight"])))) * _st(F.linear(h, sd[p + "mlp.up_proj.weight"])))
        x = _st(x + _st(F.linear(m, sd[p + "mlp.down_proj.weight"])))
    x = _rms(x[-1:] if last_only else x, sd["model.norm.weight"], eps)
    out = _st(F.linear(x, sd["lm_head.weight"] if "lm_head.weight" in sd else sd["model.embed_tokens.weight"]))
    return out[0] if last_only else out


# ---------------------------------------------------------------------------------------------------------
# k-means in the reference's own formulation (utiles.py:294-318): [T,K,D] broadcast distances, boolean-mask updates.
# Used (a) as a second, independent oracle against the golden vectors and (b) as the CPU-baseline timer of bench.py, because
# this is the arithmetic the reference actually executes on a CPU (the C oracle restates the GPU reduction tree instead).
# ---------------------------------------------------------------------------------------------------------
def weighted_kmeans_reference_formula(X, K, init_idx, reseed_idx=None, weights=None, tol=1e-4, max_iter=10):
    T = X.shape[0]
    w = torch.ones(T, dtype=X.dtype) if weights is None else weights
    C = X[torch.as_tensor(init_idx, dtype=torch.long)].clone()
    rs = list(reseed_idx) if reseed_idx is not None else []
    for i in range(max_iter):
        d = ((X.unsqueeze(1) - C.unsqueeze(0)) ** 2).sum(dim=2).sqrt()
        labels = torch.argmin(d, dim=1)
        ws = torch.zeros_like(C)
        wsum = torch.zeros(K, dtype=X.dtype)
        for j in range(K):
            m = labels == j
            ws[j] = torch.sum(w[m, None] * X[m], dim=0)
            wsum[j] = torch.sum(w[m])
        mask = wsum > 0
        newC = torch.zeros_like(ws)
        newC[mask] = ws[mask] / wsum[mask, None]
        if mask.sum() < K:
            newC[~mask] = torch.stack([X[int(rs.pop(0))] for _ in range(K - int(mask.sum()))])
        if torch.norm(C - newC, dim=1).sum() < tol:
            break
        C = newC
    return C, labels, wsum, i
