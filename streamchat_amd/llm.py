"""LongVA-7B language side on hand-written gfx950 kernels: Qwen2 decoder with a KV cache, the `<image>` embedding
splice and `generate_with_image_embedding`.

Mirrors the reference seams `LlavaQwenForCausalLM.generate_with_image_embedding` (longva/model/language_model/
llava_qwen.py:137-155) and `LlavaMetaForCausalLM.prepare_inputs_embeddings_for_multimodal` (longva/model/llava_arch.py:
208-343) for the single-sequence case the streaming path uses.  The reference calls HF `generate(..., use_cache=False)`,
re-running the full 26k-49k-token prefill for EVERY generated token (inference_streaming_longva_v2.py:257,
utiles.py:556,605); here the prompt is prefilled once into a resident KV cache and each new token is one decode step
(greedy outputs are identical with and without a cache — SURVEY.md Appendix D)."""
import types
import os
import typing

import torch

from . import ops

IMAGE_TOKEN_INDEX = -200
IGNORE_INDEX = -100


class Qwen2ConfigLite:
    def __init__(self, hidden=3584, layers=28, heads=28, kv_heads=4, intermediate=18944, vocab=152064, eps=1e-6, rope_theta=1e6,
                 head_dim=None, tokenizer_model_max_length=None):
        self.hidden, self.layers, self.heads, self.kv_heads = hidden, layers, heads, kv_heads
        self.intermediate, self.vocab, self.eps, self.rope_theta = intermediate, vocab, eps, rope_theta
        self.head_dim = head_dim or hidden // heads
        self.tokenizer_model_max_length = tokenizer_model_max_length
        self.mm_use_im_start_end = False            # LongVA config; the True branch of the prompt code is dead (Q18)


QWEN2_7B = dict(hidden=3584, layers=28, heads=28, kv_heads=4, intermediate=18944, vocab=152064, rope_theta=1e6)


def random_qwen2_state_dict(cfg: Qwen2ConfigLite, seed=0, device="cuda", dtype=torch.float16, std=0.02):
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s: (torch.randn(*s, device=device, generator=g) * std).to(dtype)
    H, I, dq, dkv = cfg.hidden, cfg.intermediate, cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim
    sd = {"model.embed_tokens.weight": rn(cfg.vocab, H), "model.norm.weight": 1 + rn(H), "lm_head.weight": rn(cfg.vocab, H)}
    for i in range(cfg.layers):
        p = f"model.layers.{i}."
        sd[p + "self_attn.q_proj.weight"] = rn(dq, H); sd[p + "self_attn.q_proj.bias"] = rn(dq)
        sd[p + "self_attn.k_proj.weight"] = rn(dkv, H); sd[p + "self_attn.k_proj.bias"] = rn(dkv)
        sd[p + "self_attn.v_proj.weight"] = rn(dkv, H); sd[p + "self_attn.v_proj.bias"] = rn(dkv)
        sd[p + "self_attn.o_proj.weight"] = rn(H, dq)
        sd[p + "mlp.gate_proj.weight"] = rn(I, H); sd[p + "mlp.up_proj.weight"] = rn(I, H); sd[p + "mlp.down_proj.weight"] = rn(H, I)
        sd[p + "input_layernorm.weight"] = 1 + rn(H); sd[p + "post_attention_layernorm.weight"] = 1 + rn(H)
    return sd


def _h(t, device):
    return t.detach().to(device=device, dtype=torch.float16).contiguous()


class Qwen2Model:
    """Decoder stack + lm_head on the HIP kernels, with a contiguous per-layer KV cache [max_seq, 2*Hkv*Dh] (K | V)."""

    def __init__(self, state_dict, cfg: Qwen2ConfigLite, device="cuda", max_seq=65536, consume=False, trim_last_layer=False):
        self.cfg, self.device, self.max_seq = cfg, torch.device(device), max_seq
        # opt-in: in a last_only prefill run the last layer's query / attention / MLP for the final row only (K/V still for every
        # row).  Same logits and same cache, 1/layers fewer flops; OFF by default so that the default path (and bench.py) does
        # exactly the work the reference's HF forward does.
        self.trim_last_layer = trim_last_layer
        H, I, Dh = cfg.hidden, cfg.intermediate, cfg.head_dim
        if H % 128 or I % 128 or (cfg.heads * Dh) % 128 or (2 * cfg.kv_heads * Dh) % 128 or cfg.vocab % 128 or Dh not in (32, 64, 128):
            raise ValueError("HIP Qwen2 path needs 128-multiple widths and head_dim in {32, 64, 128}")
        sd = state_dict
        take = (lambda k: _h(sd.pop(k), device)) if consume else (lambda k: _h(sd[k], device))
        self.embed = take("model.embed_tokens.weight")
        self.norm = take("model.norm.weight")
        self.lm_head = take("lm_head.weight") if "lm_head.weight" in sd else self.embed
        self.L = []
        for i in range(cfg.layers):
            p = f"model.layers.{i}."
            g, u = take(p + "mlp.gate_proj.weight"), take(p + "mlp.up_proj.weight")
            wgu = torch.cat([g.view(I // 2, 2, H), u.view(I // 2, 2, H)], dim=1).reshape(2 * I, H).contiguous()   # (g g u u) interleave
            del g, u
            # q | k | v projection weights in ONE buffer (round 5): wq and wkv are row views of it (every kernel that takes them separately
            # is unchanged), and the batched decode step projects all three with one launch over `wqkv`
            dq_ = cfg.heads * Dh
            wqkv = torch.cat([take(p + "self_attn.q_proj.weight"), take(p + "self_attn.k_proj.weight"), take(p + "self_attn.v_proj.weight")]).contiguous()
            bqkv = torch.cat([take(p + "self_attn.q_proj.bias"), take(p + "self_attn.k_proj.bias"), take(p + "self_attn.v_proj.bias")]).contiguous()
            self.L.append(dict(
                wqkv=wqkv, bqkv=bqkv, wq=wqkv[:dq_], bq=bqkv[:dq_], wkv=wqkv[dq_:], bkv=bqkv[dq_:],
                wo=take(p + "self_attn.o_proj.weight"), wgu=wgu, wd=take(p + "mlp.down_proj.weight"),
                ln1=take(p + "input_layernorm.weight"), ln2=take(p + "post_attention_layernorm.weight")))
        self.cache = None
        self.cache_len = 0
        self._buf_rows = 0

    def embed_tokens(self, ids):
        return ops.gather_rows(ids.to(self.device).view(-1), self.embed)

    def shared_view(self, max_seq=None):
        """A second model over the SAME weight tensors with its own KV cache, activation buffers and scratch (round 5, session.py: the answer
        whose tokens are being decoded and the next question's prefill each need a cache; the reference keeps two full replicas on two
        GPUs, inference_streaming_longva_v2.py:696-720).  Nothing is copied."""
        import copy
        v = copy.copy(self)
        v.cache, v.cache_len, v._buf_rows, v._b, v._f32_scratch, v._nsplit_prompt = None, 0, 0, None, None, None
        v.max_seq = max_seq or self.max_seq
        return v

    # ---- rotary tables (round 3): RoPE is applied to the fp32 projection sums and rounded ONCE; the query table carries the softmax
    # scale * log2 e, so q reaches sc_attention_f16 pre-scaled (SC_ATTN_Q_PRESCALED) - llm_ops.hip k_rope_table / gemm.hip rotary epilogue
    def rope_tabs(self, n_pos=None):
        c = self.cfg
        n = max(self.max_seq, n_pos or 0)
        return (ops.rope_table(n, c.head_dim, c.rope_theta, c.head_dim ** -0.5 * ops.LOG2E, self.device),
                ops.rope_table(n, c.head_dim, c.rope_theta, 1.0, self.device))

    def _qkv_rows(self, x, L, pos0, q_out, kv_out, positions=None):
        """q_out <- rope(x wq^T + bq) * scale*log2e, kv_out <- [rope(k) | v] for the rows of x at positions pos0.. (or `positions`): the rotary
        GEMM epilogue where the hand-scheduled kernel serves the shape, otherwise fp32 projections + k_rope_f32in - the same numbers."""
        c = self.cfg
        n, Dh, dq, dkv = x.shape[0], c.head_dim, c.heads * c.head_dim, c.kv_heads * c.head_dim
        tq, tk = self.rope_tabs(pos0 + n)
        fused = (positions is None and Dh == 128 and n >= 256 and ops.gemm_headed_ok(dq, x.shape[1], x, L["wq"], L["bq"], q_out)
                 and ops.gemm_headed_ok(2 * dkv, x.shape[1], x, L["wkv"], L["bkv"], kv_out))
        if fused:
            ops.gemm_headed(x, L["wq"], L["bq"], q_out, "rope", dq, tq, pos0)
            ops.gemm_headed(x, L["wkv"], L["bkv"], kv_out, "rope", dkv, tk, pos0)
            return q_out, kv_out
        # fp32 projections + k_rope_f32in (other head dims, short prompts, unaligned views): in row blocks through ONE reused fp32 scratch,
        # so that a long prefill on a non-7B shape does not materialise [n, dq] + [n, 2 dkv] fp32 per layer (0.7 GB at 49 k x 3584)
        RB = 4096
        if getattr(self, "_f32_scratch", None) is None or self._f32_scratch.shape[0] < min(n, RB) or self._f32_scratch.shape[1] < max(dq, 2 * dkv):
            self._f32_scratch = torch.empty((min(max(n, 1), RB), max(dq, 2 * dkv)), dtype=torch.float32, device=x.device)
        for r0 in range(0, n, RB):
            r1 = min(n, r0 + RB)
            pz = None if positions is None else positions[r0:r1]
            s32 = self._f32_scratch[: r1 - r0]
            ops.rope_f32in(ops.gemm(x[r0:r1], L["wq"], L["bq"], out=s32[:, :dq], out_f32=True), tq, c.heads, Dh, q_out[r0:r1], 0, pos0 + r0, pz)
            ops.rope_f32in(ops.gemm(x[r0:r1], L["wkv"], L["bkv"], out=s32[:, :2 * dkv], out_f32=True), tk, c.kv_heads, Dh, kv_out[r0:r1], dkv, pos0 + r0, pz)
        return q_out, kv_out

    def reset_cache(self, max_seq=None):
        c = self.cfg
        if max_seq is not None:
            self.max_seq = max_seq
        if self.cache is None or self.cache[0].shape[0] < self.max_seq:
            # rows of [K of all KV heads | V of all KV heads]; KV_ROW_PAD extra halves per row (a view: every kernel takes the row stride)
            w = 2 * c.kv_heads * c.head_dim
            self.cache = [torch.empty((self.max_seq, w + KV_ROW_PAD), dtype=torch.float16, device=self.device)[:, :w] for _ in range(c.layers)]
        self.cache_len = 0
        self._nsplit_prompt = None

    def _buffers(self, n):
        if self._buf_rows < n:
            c, e = self.cfg, lambda *s: torch.empty(*s, dtype=torch.float16, device=self.device)
            self._b = dict(x=e(n, c.hidden), q=e(n, c.heads * c.head_dim), att=e(n, c.heads * c.head_dim), h2=e(n, c.hidden),
                           m=e(n, c.intermediate), h=e(n, c.hidden))
            self._buf_rows = n
        return self._b

    def forward(self, embeds, last_only=True):
        """Append `embeds` [n, H] at positions cache_len.. (prefill chunk or one decode token).  Returns fp32 logits of the last
        row ([vocab]) or of every row ([n, vocab]) and advances the cache."""
        c = self.cfg
        if self.cache is None:
            self.reset_cache()
        n, pos0 = embeds.shape[0], self.cache_len
        if pos0 + n > self.max_seq:
            raise ValueError(f"sequence {pos0 + n} exceeds the KV cache ({self.max_seq})")
        if n == 1:
            return self._decode_one(embeds)
        B = self._buffers(n)
        dq, dkv, Dh = c.heads * c.head_dim, c.kv_heads * c.head_dim, c.head_dim
        h = B["h"][:n]
        h.copy_(embeds)
        h2 = B["h2"][:n]
        S = pos0 + n
        tail = None
        for l, L in enumerate(self.L):
            x = ops.rmsnorm(h, L["ln1"], c.eps, out=B["x"][:n])
            if self.trim_last_layer and last_only and l == len(self.L) - 1:
                # Only the last position's logits are wanted: in the LAST layer every row still contributes its K/V (cache), but
                # the query, attention, output projection and MLP are needed for the final row alone (1/layers of the prefill
                # flops saved; HF computes all rows and slices the logits afterwards - the returned row is the same).
                tq, tk = self.rope_tabs(S)
                kv = self.cache[l][pos0:S]
                if Dh == 128 and n >= 256 and ops.gemm_headed_ok(2 * dkv, x.shape[1], x, L["wkv"], L["bkv"], kv):
                    ops.gemm_headed(x, L["wkv"], L["bkv"], kv, "rope", dkv, tk, pos0)
                else:
                    ops.rope_f32in(ops.gemm(x, L["wkv"], L["bkv"], out_f32=True), tk, c.kv_heads, Dh, kv, dkv, pos0)
                ql = ops.rope_f32in(ops.gemm(x[n - 1:n], L["wq"], L["bq"], out_f32=True), tq, c.heads, Dh,
                                    torch.empty((1, dq), dtype=torch.float16, device=self.device), 0, S - 1)
                ck = self.cache[l][:S]
                nsplit = max(1, min(128, ((S + 63) // 64) // 2))
                al = ops.attention(ql.unsqueeze(0), ck[:, :dkv].unsqueeze(0), ck[:, dkv:].unsqueeze(0), c.heads, c.kv_heads, Dh, Dh ** -0.5,
                                   causal=True, nsplit=nsplit, q_prescaled=True).squeeze(0)
                hl = ops.gemm(al, L["wo"], None, residual=h[n - 1:n])
                ml = ops.gemm(ops.rmsnorm(hl, L["ln2"], c.eps), L["wgu"], None, epilogue="swiglu")
                tail = ops.gemm(ml, L["wd"], None, residual=hl)
                break
            q, _ = self._qkv_rows(x, L, pos0, B["q"][:n], self.cache[l][pos0:S])      # K | V land in the cache rows; q pre-scaled
            ck = self.cache[l][:S]
            att = ops.attention(q.unsqueeze(0), ck[:, :dkv].unsqueeze(0), ck[:, dkv:].unsqueeze(0), c.heads, c.kv_heads, Dh, Dh ** -0.5,
                                causal=True, out=B["att"][:n].unsqueeze(0), q_prescaled=True).squeeze(0)
            ops.gemm(att, L["wo"], None, residual=h, out=h2)
            x = ops.rmsnorm(h2, L["ln2"], c.eps, out=B["x"][:n])
            m = ops.gemm(x, L["wgu"], None, epilogue="swiglu", out=B["m"][:n])
            ops.gemm(m, L["wd"], None, residual=h2, out=h)
        self.cache_len = S
        self._nsplit_prompt = decode_nsplit(Dh, S)          # the split-KV factor of every decode step that follows this prefill
        if tail is None:
            tail = h[n - 1:n] if last_only else h
        xn = ops.rmsnorm(tail, self.norm, c.eps)
        logits = ops.gemm(xn, self.lm_head, None, out_f32=True)
        return logits[0] if last_only else logits


def splice_image_embeddings(ids, embed_table, image_features, max_len=None, labels=None):
    """llava_arch.py:208-343 for one sequence: text ids -> embedding rows, every -200 sentinel replaced by the next tensor of
    `image_features`; zero sentinels => the visual tokens are dropped (:247-254, Q18); truncated to max_len (:288-291).
    An entry of `image_features` may also be a LIST of tensors ([short | retrieved ...] pieces of one image block): the pieces are copied
    straight into the spliced sequence, which is what the reference's `torch.cat([short, long])` (inference_streaming_longva_v2.py:
    188-196) followed by this splice produces, without materialising the concatenation first."""
    dev = embed_table.device
    ids = ids.to(dev)
    pos = (ids == IMAGE_TOKEN_INDEX).nonzero().flatten().tolist()
    feats = [[p.reshape(-1, p.shape[-1]) for p in f] if isinstance(f, (list, tuple)) else [f.reshape(-1, f.shape[-1])] for f in (image_features or [])]
    n_img_rows = sum(p.shape[0] for i in range(len(pos)) for p in feats[i])
    total = ids.numel() - len(pos) + n_img_rows
    out = torch.empty((total, embed_table.shape[1]), dtype=torch.float16, device=dev)
    labels_out = torch.full((total,), IGNORE_INDEX, dtype=torch.long, device=dev)
    emb = ops.gather_rows(ids, embed_table)          # one gather for all text rows (sentinel ids < 0 give zero rows)
    src, dst = 0, 0
    for k, p in enumerate(pos + [ids.numel()]):
        seg = p - src
        out[dst:dst + seg] = emb[src:src + seg]
        if labels is not None:
            labels_out[dst:dst + seg] = labels.to(dev)[src:src + seg]
        dst += seg
        src = p + 1
        if k < len(pos):
            for f in feats[k]:
                out[dst:dst + f.shape[0]] = f          # (copy_ converts device / dtype where a caller hands over something else)
                dst += f.shape[0]
    if max_len is not None:
        out, labels_out = out[:max_len], labels_out[:max_len]
    return out, labels_out


def decode_nsplit(head_dim: int, cache_len: int) -> int:
    """split-KV factor of a batch-1 decode step.  It is fixed ONCE per prompt, from the cache length the prefill leaves (Qwen2Model.forward
    stores it; the eager token loop keeps it for every token of that generate, a DecodeGraph built after the same prefill computes the same
    value at construction), so the eager loop and a graph captured for that prompt sum the split partials in the same order and stay
    bit-identical however many tokens are generated.  A graph REUSED for a later prompt of a very different length keeps the factor it was
    captured with: same mathematics, partials merged in a different grouping (last-bit differences in the logits).  k_attn_decode (Dh = 128): one workgroup of four streaming waves per (KV head, split), ~6 chunks of 32 rows per wave: 64 splits x
    4 KV heads = one workgroup per CU at a 49 k context (measured 32 / 64 / 128 splits: 290 / 309 / 301 tok/s, profiles/r03_run5).  Other
    head dims (k_attn, one computing wave per workgroup): ~6 tiles of 64 rows per workgroup, up to 128 splits."""
    if head_dim == 128:
        return max(2, min(64, ((cache_len + 31) // 32) // 24))
    return max(1, min(128, ((cache_len + 63) // 64) // 6))


def _decode_one(self, embeds):
    """One decode step (batch 1): every projection is a GEMV that streams its weights once (HBM-bound), attention is split-KV
    over the cache with the G = Hq/Hkv query heads of a KV group packed as G query rows (so each K/V byte is read once)."""
    c = self.cfg
    pos0, S = self.cache_len, self.cache_len + 1
    dq, dkv, Dh, G = c.heads * c.head_dim, c.kv_heads * c.head_dim, c.head_dim, c.heads // c.kv_heads
    h = embeds.reshape(1, -1)
    nsplit = getattr(self, "_nsplit_prompt", None) or decode_nsplit(Dh, pos0)       # fixed by the prefill of this prompt (see decode_nsplit)
    tq, tk = self.rope_tabs(S)
    for l, L in enumerate(self.L):
        # RMSNorm fused into the projections; fp32 sums -> RoPE with the fp32 tables -> ONE rounding (q pre-scaled): the arithmetic of
        # k_decode_qkv<true> (the captured graph's single launch) in three launches
        q32 = ops.gemv(L["wq"], h, L["bq"], out_f32=True, rms_gamma=L["ln1"], rms_eps=c.eps).view(1, dq)
        kv32 = ops.gemv(L["wkv"], h, L["bkv"], out_f32=True, rms_gamma=L["ln1"], rms_eps=c.eps).view(1, 2 * dkv)
        q = ops.rope_f32in(q32, tq, c.heads, Dh, torch.empty((1, dq), dtype=torch.float16, device=self.device), 0, pos0)
        ops.rope_f32in(kv32, tk, c.kv_heads, Dh, self.cache[l][pos0:S], dkv, pos0)
        ck = self.cache[l][:S]
        # the G heads of a KV group become G query ROWS of that KV head purely by addressing (row stride Dh, head stride G*Dh)
        qv = q.as_strided((1, G, Dh), (dq, Dh, 1))
        att = ops.attention(qv, ck[:, :dkv].unsqueeze(0), ck[:, dkv:].unsqueeze(0), c.kv_heads, c.kv_heads, Dh, Dh ** -0.5, causal=False,
                            nsplit=nsplit, q_head_stride=G * Dh, o_head_stride=G * Dh, out_ld=Dh, q_prescaled=True).view(1, dq)
        h2 = ops.gemv(L["wo"], att, None, residual=h).view(1, -1)
        m = ops.gemv(L["wgu"], h2, None, epilogue="swiglu", rms_gamma=L["ln2"], rms_eps=c.eps)
        h = ops.gemv(L["wd"], m, None, residual=h2).view(1, -1)
    self.cache_len = S
    return ops.gemv(self.lm_head, h, None, out_f32=True, rms_gamma=self.norm, rms_eps=c.eps)


Qwen2Model._decode_one = _decode_one


def _lib_pick_bytes(B):
    from . import _lib
    return int(_lib.load().sc_pick_token_workspace_bytes(B))


_UNSET = object()
# KV-cache row padding in halves (SC_KV_PAD overrides; 0 = rows 2 * Hkv * Dh halves apart).  A row is [K of the 4 KV heads | V of the 4 KV heads] =
# 2048 B at Qwen2-7B's widths, so head h's 256-byte piece of consecutive rows sits 2048 B apart: every row of a head falls into the same slot of
# the memory channels' interleave, and the decode attention's in-kernel timeline showed ONE head's stream ending 2.7 us after the others - its
# addresses, not its CUs (profiles/r05_run_m_*).  128 B of padding per row (2176 B apart) walks the slots: +1.4 % tokens/s at a 49 k context in
# the real token loop, interleaved on two boxes (325.2 -> 329.8 and 325.5 -> 330.3; 64 B: +0.2 %, 192 - 512 B: +1.0 - 1.3 %), the prefill unchanged,
# results bit-identical (profiles/r06_run_k_kv_row_padding.md).  Every kernel takes the row stride, so the cache is a view of the padded rows.
import os as _os
KV_ROW_PAD = int(_os.environ.get("SC_KV_PAD", "64"))

# The default CUDA generator of a device has ONE capture state: while a host thread captures a graph (torch.cuda.graph puts the generator into
# capture mode, whether or not the graph draws from it), a draw or the replay of a sampling graph on ANOTHER host thread fails ("... during
# capture").  Since round 5 two host threads generate at the same time (session.py, the entry point's --overlap: the answer on one, the chunk
# captions on the other): captures, eager draws and sampling replays take this lock.  Uncontended it costs ~0.1 us per replay.
import threading as _threading
_GEN_LOCK = _threading.RLock()


# Round 6: SAMPLING draws nothing from a generator on the device any more.  Every sequence gets a 63-bit SEED when its generate starts and the
# uniform of its n-th sampled token is `ops.counter_uniform(seed, n)` - a pure function (sampling.hip, sc_counter_uniform_f32).  So a sequence
# samples the same tokens decoded alone or in a batch (BatchDecoder), eagerly or from a replayed hipGraph, and whatever another host thread
# is sampling at the same moment (session.py, the entry point's --overlap: ADVICE r05).  Seeds come, in call order, from `seed=` (explicit), a
# `generator` (CPU or CUDA), the model object's `seed_generator` (a CPU torch.Generator of its own: one per role in the entry point), or torch's
# default CPU generator (`torch.manual_seed` seeds it).  _GEN_LOCK is still taken around graph CAPTURES: torch puts the device's default
# generator into capture mode whether or not the graph draws from it.
def draw_seeds(B, generator=None, seed=None):
    """[B] int64 CPU tensor of per-sequence seeds"""
    if seed is not None:
        sd = torch.as_tensor(seed, dtype=torch.int64).reshape(-1)
        if sd.numel() == 1 and B > 1:
            sd = sd + torch.arange(B, dtype=torch.int64) * 0x632BE5AB            # one explicit seed for a batch: a distinct stream per sequence
        if sd.numel() != B:
            raise ValueError(f"draw_seeds: {sd.numel()} seeds for {B} sequences")
        return sd.clone()
    if generator is not None and generator.device.type != "cpu":
        with _GEN_LOCK:
            return torch.randint(0, 2 ** 62, (B,), generator=generator, device=generator.device).cpu()
    return torch.randint(0, 2 ** 62, (B,), generator=generator)


class Sampling(typing.NamedTuple):
    """What turns logits into the next token: HF's processor chain as `generate` builds it (transformers generation/utils.py
    _get_logits_processor / _get_logits_warper): repetition penalty always, temperature / top-k / top-p only when sampling."""
    temperature: float = 0.0          # 0: arg-max
    top_k: int = 0                    # 0: off
    top_p: float = 1.0                # 1: off
    repetition_penalty: float = 1.0   # 1: off

    @property
    def plain(self):
        return self.top_k == 0 and self.top_p >= 1.0 and self.repetition_penalty == 1.0


def resolve_sampling(generation_config, do_sample, temperature=_UNSET, top_p=_UNSET, top_k=_UNSET, repetition_penalty=_UNSET):
    """HF semantics (GenerationConfig.update): an argument the caller passes - even None - overrides the checkpoint's
    generation_config.json; one it does not pass falls back to that file, then to HF's defaults (temperature 1.0, top_k 50, top_p 1.0,
    repetition_penalty 1.0, do_sample False); `do_sample` follows the same rule.  The reference passes do_sample, temperature and top_p (inference_streaming_longva_v2.py:252-256, utiles.py:
    551-556) and inherits top_k and repetition_penalty.  None / 0 / 1.0 switch a stage off; the warpers only exist when sampling."""
    g = dict(temperature=1.0, top_k=50, top_p=1.0, repetition_penalty=1.0)
    g.update({k: v for k, v in (generation_config or {}).items() if k in g})
    if do_sample is _UNSET or do_sample is None:        # not passed: the checkpoint's file decides, then HF's default (greedy)
        do_sample = bool((generation_config or {}).get("do_sample", False))
    for k, v in (("temperature", temperature), ("top_p", top_p), ("top_k", top_k), ("repetition_penalty", repetition_penalty)):
        if v is not _UNSET:
            g[k] = v
    pen = float(g["repetition_penalty"] or 1.0)
    if not do_sample:
        return Sampling(0.0, 0, 1.0, pen)
    t = float(g["temperature"] if g["temperature"] is not None else 1.0)
    if t <= 0:
        raise ValueError("do_sample=True needs temperature > 0 (HF raises the same way)")
    k = int(g["top_k"] or 0)
    pp = float(g["top_p"] if g["top_p"] is not None else 1.0)
    # (top_k > 64 and a nucleus without top-k run through sampling.hip's full-vocabulary radix-select path since round 3)
    return Sampling(t, k, pp, pen)


class DecodeGraph:
    """Batch-1 decode (greedy, or temperature sampling: the reference's default, inference_streaming_longva_v2.py:252-253) captured ONCE as a hipGraph and replayed per token (the 13 launches x layers of a decode step are
    launch-bound from Python: cdna guide "capture launch-bound inner loops in hipGraphs").  Everything that changes from token to
    token lives in device memory: the input token id, the cache position (GEMV / RoPE write the KV row `pos`), the valid key count
    (attention takes it as kv_len over the full-capacity cache view), and the output token ring."""

    def __init__(self, lm, max_new_tokens=1024, nsplit=None, temperature=0.0, sampling=None):
        self.sampling = sampling if sampling is not None else Sampling(float(temperature))
        self.temperature = self.sampling.temperature    # > 0: sample; the uniform of token n is counter_uniform(seed, n): a pure function, graph-safe
        self.nsplit = nsplit if nsplit else decode_nsplit(lm.cfg.head_dim, lm.cache_len)
        self.lm = lm
        dev = lm.device
        self.tok = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.len = torch.zeros(1, dtype=torch.int32, device=dev)
        self.cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        self.hist = torch.zeros(max_new_tokens + 1, dtype=torch.int64, device=dev)     # every token generated so far: [first | ring]
        self.out = self.hist[1:]                                                       # the ring the graph appends to
        self.nprev = torch.ones(1, dtype=torch.int32, device=dev)                      # valid entries of hist (the repetition penalty's set)
        self.graph = None
        # Everything the captured graph reads or writes through a raw pointer is OWNED here or checked before every replay: a
        # private split-KV workspace and token-pick workspace (the shared grow-only scratch of ops._workspace may be reallocated by
        # any later launch — a k-means merge needs ~100 MB of it), and the KV-cache pointers recorded at capture (`valid()`).
        c = lm.cfg
        self.ws_attn = torch.empty(max(ops.attention_workspace_bytes(1, c.kv_heads, c.heads // c.kv_heads, self.nsplit, c.head_dim), 256),
                                   dtype=torch.uint8, device=dev)
        self.ws_pick = torch.empty(max(ops.sample_token_workspace_bytes(1), 256), dtype=torch.uint8, device=dev)
        self.nxt = torch.zeros(1, dtype=torch.int64, device=dev)
        self.seed = torch.zeros(1, dtype=torch.int64, device=dev)                      # the sequence's sampling seed (start())
        self.u = torch.zeros(1, dtype=torch.float32, device=dev)
        self.q_buf = torch.empty(c.heads * c.head_dim, dtype=torch.float16, device=dev)
        self.tab_q, self.tab_k = lm.rope_tabs(lm.cache[0].shape[0] if lm.cache is not None else None)      # fp32 rotary tables the graph reads by pointer
        self._captured_ptrs = None

    def _ptrs(self):
        return tuple(c.data_ptr() for c in self.lm.cache) + (self.lm.cache[0].shape[0], self.tab_q.data_ptr(), self.tab_k.data_ptr())

    def valid(self):
        """False once the KV cache was reallocated after capture (reset_cache with a larger max_seq): the graph would replay into
        freed memory — the owner must drop it and capture a new one."""
        return self.graph is None or self._captured_ptrs == self._ptrs()

    def _body(self):
        lm, c = self.lm, self.lm.cfg
        dq, dkv, Dh, G = c.heads * c.head_dim, c.kv_heads * c.head_dim, c.head_dim, c.heads // c.kv_heads
        h = ops.gather_rows(self.tok, lm.embed)
        for l, L in enumerate(lm.L):
            # q / k / v projections + RMSNorm + RoPE + KV-cache append at row `pos`: one launch (gemv.hip k_decode_qkv)
            q = ops.decode_qkv_tab(L["wq"], L["wkv"], L["bq"], L["bkv"], h, L["ln1"], c.eps, self.q_buf, lm.cache[l], self.pos, c.heads, c.kv_heads, Dh,
                                   self.tab_q, self.tab_k).view(1, dq)
            ck = lm.cache[l]
            qv = q.as_strided((1, G, Dh), (dq, Dh, 1))
            att = ops.attention(qv, ck[:, :dkv].unsqueeze(0), ck[:, dkv:].unsqueeze(0), c.kv_heads, c.kv_heads, Dh, Dh ** -0.5, causal=False,
                                kv_len=self.len, nsplit=self.nsplit, q_head_stride=G * Dh, o_head_stride=G * Dh, out_ld=Dh, ws=self.ws_attn,
                                q_prescaled=True).view(1, dq)
            h2 = ops.gemv(L["wo"], att, None, residual=h).view(1, -1)
            m = ops.gemv(L["wgu"], h2, None, epilogue="swiglu", rms_gamma=L["ln2"], rms_eps=c.eps)
            h = ops.gemv(L["wd"], m, None, residual=h2).view(1, -1)
        logits = ops.gemv(lm.lm_head, h, None, out_f32=True, rms_gamma=lm.norm, rms_eps=c.eps)
        sp = self.sampling                                                   # HIP next-token kernels over the 152 064 logits (sampling.hip)
        u = ops.counter_uniform(self.seed, self.cnt, 1, out=self.u) if sp.temperature > 0 else None      # token n = cnt + 1 (the first was picked by the caller)
        if sp.plain:
            nxt = ops.pick_token(logits, sp.temperature, u, out=self.nxt, ws=self.ws_pick)
        else:
            nxt = ops.sample_token(logits, sp.temperature, u, sp.top_k, sp.top_p, sp.repetition_penalty, prev_ids=self.hist.view(1, -1), n_prev=self.nprev,
                                   out=self.nxt, ws=self.ws_pick)
        ops.decode_advance(nxt, self.out, self.cnt, self.tok, self.pos, self.len, self.nprev)     # ring append + the five counters: one launch
        return logits

    def start(self, first_token: int, seed: int = 0):
        """position the graph right after the prefill: next input token = first_token, cache length = lm.cache_len; `seed`: the sequence's
        sampling seed (its first token was drawn with n = 0 by the caller, the graph draws n = 1, 2, ..)"""
        self.tok.fill_(int(first_token)); self.pos.fill_(self.lm.cache_len); self.len.fill_(self.lm.cache_len + 1); self.cnt.zero_()
        self.hist[0] = int(first_token); self.nprev.fill_(1); self.seed.fill_(int(seed))

    def capture(self):
        with _GEN_LOCK:                 # (the capture puts the device's default generator into capture mode: see _GEN_LOCK)
            self._capture()

    def _capture(self):
        snap = (self.tok.clone(), self.pos.clone(), self.len.clone(), self.cnt.clone(), self.nprev.clone())
        s = torch.cuda.Stream(device=self.lm.device)
        s.wait_stream(torch.cuda.current_stream(self.lm.device))
        with torch.cuda.stream(s):                          # warm-up outside capture (workspaces, function attributes)
            self._body()
        torch.cuda.current_stream(self.lm.device).wait_stream(s)
        for t, v in zip((self.tok, self.pos, self.len, self.cnt, self.nprev), snap):
            t.copy_(v)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):      # (another host thread may be driving its own stream: session.py)
            self.logits = self._body()
        self._captured_ptrs = self._ptrs()
        for t, v in zip((self.tok, self.pos, self.len, self.cnt, self.nprev), snap):
            t.copy_(v)

    def run(self, n_tokens: int, eos=(), check_every: int = 16):
        """up to n tokens; returns them as a CPU list and advances lm.cache_len.  Without `eos` there is ONE host sync at the end;
        with `eos` (ids that end the sequence, HF semantics: the EOS token is kept) the output ring is read every `check_every`
        replays and the run stops at the first EOS — the few tokens decoded past it are dropped and the cache length rewound."""
        if not self.valid():
            raise RuntimeError("DecodeGraph: the KV cache was reallocated after capture; build a new DecodeGraph")
        if n_tokens > self.out.numel() or self.lm.cache_len + n_tokens > self.lm.cache[0].shape[0]:
            raise ValueError("DecodeGraph.run: more tokens than the output ring / KV cache can hold")
        if self.graph is None:
            self.capture()
        eos = set(eos)
        done, toks = 0, []
        while done < n_tokens:
            step = n_tokens - done if not eos else min(check_every, n_tokens - done)
            for r in range(step):
                if r % 32 == 0:
                    ops.stream_ptr(self.lm.device)          # (a pending ops.move_to_stream_when of this thread takes effect between replays)
                self.graph.replay()
            done += step
            if eos:
                toks = self.out[:done].cpu().tolist()
                cut = next((i for i, t in enumerate(toks) if t in eos), None)
                if cut is not None:
                    toks, done = toks[:cut + 1], cut + 1
                    break
        if not eos:
            toks = self.out[:done].cpu().tolist()
        self.lm.cache_len += done                          # rows past an EOS stay in the cache but are never attended to again
        return toks


class BatchDecoder:
    """Decode B sequences together (the reference captions every 40-frame chunk with its own 7B generate at batch 1 and
    without a KV cache, utiles.py:539-559).  Each sequence is prefilled on its own (prefill is compute-bound, batching buys nothing)
    into slice b of ONE per-layer cache [B, cap, 2*dkv]; decode then advances all B sequences per step: the projections are GEMMs
    with M = B, so the 15 GB of weights stream once per step instead of once per sequence, the new K/V rows are scattered with one
    index_copy per layer, and attention is ONE launch per layer over the batch (per-sequence kv_len, split-KV)."""

    def __init__(self, lm, prompts, max_new_tokens, replicate=None):
        """prompts: list of [n_b, H] fp16 embeddings (image rows already spliced).  `replicate=N` (beam search): ONE prompt, prefilled once,
        its cache rows copied into N slots - the N beams of llm.LlavaQwenForCausalLM's beam search start from the same prompt."""
        c = lm.cfg
        if replicate is not None:
            if len(prompts) != 1:
                raise ValueError("BatchDecoder(replicate=N) takes exactly one prompt")
            prompts_all, prompts = [prompts[0]] * int(replicate), prompts
        else:
            prompts_all = prompts
        self.lm, self.B = lm, len(prompts_all)
        self.cap = max(int(e.shape[0]) for e in prompts_all) + max_new_tokens
        dev = lm.device
        self._rope_tabs = lm.rope_tabs(self.cap)                 # rotary tables cover every position of this batch before a step is captured; the
                                                                 # reference held here keeps them alive if the process-wide cache regrows
        # (dense rows here: padded rows - KV_ROW_PAD - measured no gain on the batched step, 10.64 / 10.68 against 10.67 / 10.73 ms: 26 streams at 26 bases)
        self.cache = [torch.empty((self.B, self.cap, 2 * c.kv_heads * c.head_dim), dtype=torch.float16, device=dev) for _ in range(c.layers)]
        self.len = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self._ws_pick = torch.empty(max(ops.sample_token_workspace_bytes(self.B), 256), dtype=torch.uint8, device=dev)      # owned: the decode step is graph-captured
        first_logits = []
        # (forward() also records the split-KV factor of "the prompt in the cache": it belongs to the saved state - ADVICE r04 - or an eager
        #  decode that continues the earlier prompt would merge its partials in another grouping than a DecodeGraph built for it)
        saved = (lm.cache, lm.cache_len, lm.max_seq, getattr(lm, "_nsplit_prompt", None))
        try:
            for b, e in enumerate(prompts):
                lm.cache, lm.cache_len, lm.max_seq = [cl[b] for cl in self.cache], 0, self.cap
                first_logits.append(lm.forward(e.to(dev)))
                self.len[b] = lm.cache_len
        finally:
            lm.cache, lm.cache_len, lm.max_seq, lm._nsplit_prompt = saved
        if replicate is not None:                      # slot 0 holds the prompt: the other beams get its K / V rows and its logits
            n0 = int(self.len[0].item())
            for cl in self.cache:
                cl[1:, :n0].copy_(cl[0:1, :n0].expand(self.B - 1, -1, -1))
            self.len.fill_(n0)
            first_logits = first_logits * self.B
        # fp32 q | k | v sums of one step: owned here (the decode step is graph-captured; the model's shared scratch may be regrown by a later call)
        self._qkv32 = torch.empty((self.B, (c.heads + 2 * c.kv_heads) * c.head_dim), dtype=torch.float32, device=dev)
        self._q16 = torch.empty((self.B, c.heads * c.head_dim), dtype=torch.float16, device=dev)
        self.logits = torch.stack(first_logits)                                  # [B, vocab] fp32

    def step(self, tokens):
        """tokens [B] int32 (device) -> logits [B, vocab] fp32 for the next position of every sequence.  Everything that changes from
        step to step (tokens, lengths, cache rows) lives in device tensors, so the step is hipGraph-capturable."""
        lm, c, B = self.lm, self.lm.cfg, self.B
        dq, dkv, Dh = c.heads * c.head_dim, c.kv_heads * c.head_dim, c.head_dim
        h = ops.gather_rows(tokens, lm.embed)
        kvlen = self.len + 1
        for l, L in enumerate(lm.L):
            x = ops.rmsnorm(h, L["ln1"], c.eps)
            # q | k | v in ONE projection launch over the fused weight (the same column sums as two launches), fp32 sums -> RoPE at each
            # sequence's position -> one rounding (the numerics of Qwen2Model._qkv_rows) and the new K | V row appended to each sequence's
            # cache, in ONE launch (round 5: was rope(q) + rope(k|v) + index_copy_)
            s32 = ops.gemm(x, L["wqkv"], L["bqkv"], out=self._qkv32, out_f32=True)
            tq, tk = self._rope_tabs
            ck = self.cache[l]
            q = ops.rope_qkv_rows(s32, tq, tk, self.len, c.heads, c.kv_heads, Dh, self._q16, ck)
            # the G query heads of a KV group as G query ROWS of that KV head (addressing only; q batch stride = dq)
            G = c.heads // c.kv_heads
            att = ops.attention(q.as_strided((B, G, Dh), (dq, Dh, 1)), ck[:, :, :dkv], ck[:, :, dkv:], c.kv_heads, c.kv_heads, Dh, Dh ** -0.5,
                                causal=False, kv_len=kvlen, nsplit=self.nsplit, q_head_stride=G * Dh, o_head_stride=G * Dh, out_ld=Dh,
                                ws=self._ws_attn, q_prescaled=True).view(B, dq)
            h2 = ops.gemm(att, L["wo"], None, residual=h)
            m = ops.gemm(ops.rmsnorm(h2, L["ln2"], c.eps), L["wgu"], None, epilogue="swiglu")
            h = ops.gemm(m, L["wd"], None, residual=h2)
        self.len += 1
        return ops.gemm(ops.rmsnorm(h, lm.norm, c.eps), lm.lm_head, None, out_f32=True)

    def prepare_steps(self, max_new_tokens):
        """what `step()` needs when it is driven from outside `generate()` (beam search): the split-KV factor and its workspace"""
        B, dev, c = self.B, self.lm.device, self.lm.cfg
        longest = int(self.len.max().item()) + max_new_tokens
        self.nsplit = getattr(self, "nsplit_override", None) or max(1, min(64, ((longest + 31) // 32) // 24, -(-10 * 256 // (B * c.kv_heads))))
        self._ws_attn = torch.empty(max(ops.attention_workspace_bytes(B, c.kv_heads, c.heads // c.kv_heads, self.nsplit, c.head_dim), 256),
                                    dtype=torch.uint8, device=dev)

    def reorder(self, beam_idx):
        """slot b continues the sequence that slot beam_idx[b] held (HF `reorder_cache`): the used rows of every layer's cache are gathered"""
        idx = torch.as_tensor(beam_idx, dtype=torch.int64, device=self.lm.device)
        if bool((idx.cpu() == torch.arange(self.B)).all()):
            return
        n = int(self.len.max().item())
        for cl in self.cache:
            cl[:, :n].copy_(cl[:, :n].index_select(0, idx))
        self.len.copy_(self.len.index_select(0, idx))

    def _pick(self, logits, sp, counter=None):
        """next token per sequence on the device (sampling.hip): HF's processor chain of `sp` over the ids each sequence generated so far
        (self.hist[b, :nprev]); the uniform of sequence b's n-th token is counter_uniform(seed[b], n), n = counter[0] (None: 0, the first token)"""
        u = ops.counter_uniform(self.seed, counter, 0, out=self._u) if sp.temperature > 0 else None
        if sp.plain:
            return ops.pick_token(logits, sp.temperature, u, ws=self._ws_pick)
        return ops.sample_token(logits, sp.temperature, u, sp.top_k, sp.top_p, sp.repetition_penalty, prev_ids=self.hist, n_prev=self.nprev, ws=self._ws_pick)

    def _graph_body(self, sp):
        logits = self.step(self.tok)
        nxt = self._pick(logits, sp, self.cnt)                 # cnt = the index of the token being drawn (1 after the first)
        self.hist.index_copy_(1, self.cnt, nxt.view(-1, 1))
        self.tok.copy_(nxt)
        self.cnt.add_(1); self.nprev.add_(1)

    def generate(self, max_new_tokens, do_sample=False, temperature=1.0, eos_token_id=None, generator=None, use_graph=True, sampling=None, seed=None):
        """Returns a list of B python lists of new token ids (each cut at its first EOS, EOS included like HF).  The decode step is
        captured once as a hipGraph and replayed (the ~340 launches of a step are launch-bound from Python).  Sampling: sequence b draws
        counter_uniform(seed[b], n) for its n-th token (`draw_seeds`: `seed`, `generator` or torch's default CPU generator, in that order) -
        the tokens a sequence samples do not depend on what it is batched with.  EOS is checked on the host every 16 steps (sequences that
        are done keep stepping; their extra tokens are dropped)."""
        B, dev = self.B, self.lm.device
        self.seed = draw_seeds(B, generator, seed).to(dev)
        self._u = torch.zeros(B, dtype=torch.float32, device=dev)
        sp = sampling if sampling is not None else Sampling(float(temperature) if do_sample and temperature > 0 else 0.0)
        eos = set(eos_token_id) if isinstance(eos_token_id, (list, tuple, set)) else (set() if eos_token_id is None else {eos_token_id})   # HF allows a list
        longest = int(self.len.max().item()) + max_new_tokens
        # split-KV factor of the batched step: ~10 workgroups per CU over the whole launch (B x KV heads x splits), at least ~24 chunks of 32
        # rows per workgroup.  Round 4 took the batch-1 rule (64 splits): 6656 workgroups of < 3 chunks per wave at 26 x 22.7 k - the launch
        # was its own prologue, merge and 64-partial combine (15.6 -> 11.5 ms per step at 24 splits, profiles/r05_*)
        c_ = self.lm.cfg
        self.nsplit = getattr(self, "nsplit_override", None) or max(1, min(64, ((longest + 31) // 32) // 24, -(-10 * 256 // (B * c_.kv_heads))))
        c = self.lm.cfg
        self._ws_attn = torch.empty(max(ops.attention_workspace_bytes(B, c.kv_heads, c.heads // c.kv_heads, self.nsplit, c.head_dim), 256),
                                    dtype=torch.uint8, device=dev)
        self._row0 = torch.arange(B, device=dev, dtype=torch.int64) * self.cap
        self.hist = torch.zeros((B, max_new_tokens), dtype=torch.int64, device=dev)      # generated ids per sequence (the repetition penalty's set)
        self.nprev = torch.zeros(B, dtype=torch.int32, device=dev)
        first = self._pick(self.logits, sp, None)
        self.hist[:, 0] = first
        self.nprev.fill_(1)
        self.tok = first.to(torch.int32).contiguous()
        self.cnt = torch.ones(1, dtype=torch.int64, device=dev)
        graph = None
        if use_graph and max_new_tokens > 2:
            state = (self.tok, self.len, self.cnt, self.hist, self.nprev)
            snap = tuple(t.clone() for t in state)
            s = torch.cuda.Stream(device=dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):              # warm-up outside capture (workspaces, function attributes); undone below
                self._graph_body(sp)
            torch.cuda.current_stream(dev).wait_stream(s)
            restore = lambda: [t.copy_(v) for t, v in zip(state, snap)]
            restore()
            graph = torch.cuda.CUDAGraph()
            with _GEN_LOCK, torch.cuda.graph(graph, capture_error_mode="thread_local"):      # (_GEN_LOCK: see its definition)
                self._graph_body(sp)
            restore()
        steps = 1
        while steps < max_new_tokens:
            if graph is not None:
                graph.replay()
            else:
                nxt = self._pick(self.step(self.tok), sp, self.cnt)
                self.hist[:, steps] = nxt
                self.tok.copy_(nxt)
                self.cnt.add_(1); self.nprev.add_(1)
            steps += 1
            if eos and (steps % 16 == 0 or steps == max_new_tokens):
                col = self.hist[:, :steps].cpu()
                if all(any(int(v) in eos for v in col[b]) for b in range(B)):
                    break
        self.steps_run = steps - 1                      # decode steps actually executed (EOS is only looked at every 16 steps: >= the longest kept sequence - 1)
        toks = self.hist[:, :steps].cpu().tolist()
        res = []
        for b in range(B):
            t = toks[b]
            cut = next((i for i, v in enumerate(t) if v in eos), None)
            if cut is not None:
                t = t[:cut + 1]
            res.append(t)
        return res


class LlavaQwenForCausalLM:
    """Mirror of the reference model object the entry point drives (longva/model/language_model/llava_qwen.py:40-155 +
    LlavaMetaForCausalLM): `.encode_images`, `.get_model().embed_tokens`, `.prepare_inputs_embeddings_for_multimodal`,
    `.generate_with_image_embedding`, `.config`, `.device`."""

    def __init__(self, lm: Qwen2Model, frame_encoder=None, eos_token_id=None):
        self.lm, self.frame_encoder = lm, frame_encoder
        self.config = lm.cfg
        self.device = lm.device
        self.eos_token_id = eos_token_id
        self.training = False
        self._dg, self._dgs = None, {}                      # decode graphs by Sampling spec
        self.generation_config = {}                         # the checkpoint's generation_config.json (checkpoint.load_longva); {} = HF's defaults

    def get_model(self):
        return types.SimpleNamespace(embed_tokens=self.lm.embed_tokens, mm_projector=getattr(self.frame_encoder, "projector", None))

    def encode_images(self, images):
        return self.frame_encoder.encode_images(images)

    def encode_frames_u8(self, frames):
        return self.frame_encoder.encode_frames_u8(frames)

    def prepare_inputs_embeddings_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels, image_features,
                                                 modalities=["image"]):
        """llava_arch.py:208-343 for batch size 1: text ids -> embed_tokens rows, every -200 sentinel replaced by the next tensor of
        `image_features`; zero sentinels => visual tokens dropped (:247-254, Q18); truncated to tokenizer_model_max_length (:288-291).
        Returns the reference's 6-tuple (None, position_ids, attention_mask, past_key_values, inputs_embeds [1, L, H], labels)."""
        ids = input_ids[0] if input_ids.dim() == 2 else input_ids
        if attention_mask is not None:
            ids = ids[attention_mask[0].bool()] if attention_mask.dim() == 2 else ids[attention_mask.bool()]
        lab = None if labels is None else (labels[0] if labels.dim() == 2 else labels)
        out, labels_out = splice_image_embeddings(ids, self.lm.embed, image_features, self.config.tokenizer_model_max_length, lab)
        L = out.shape[0]
        new_pos = None if position_ids is None else torch.arange(L, device=self.device).unsqueeze(0)
        new_mask = None if attention_mask is None else torch.ones((1, L), dtype=attention_mask.dtype, device=self.device)
        return None, new_pos, new_mask, past_key_values, out.unsqueeze(0), (None if labels is None else labels_out.unsqueeze(0))

    @torch.no_grad()
    def generate_batch_with_image_embedding(self, inputs_list, image_embeddings_list, modalities=["image"], do_sample=_UNSET, temperature=_UNSET,
                                            max_new_tokens=256, generator=None, top_p=_UNSET, top_k=_UNSET, repetition_penalty=_UNSET, **kwargs):
        """B independent prompts (one `inputs` ids tensor and one image_embeddings list each, as for generate_with_image_embedding)
        decoded together by BatchDecoder.  Returns a list of B LongTensors [1, n_b] of new token ids."""
        sp = resolve_sampling(self.generation_config, do_sample, temperature, top_p, top_k, repetition_penalty)
        prompts = []
        for ids, img in zip(inputs_list, image_embeddings_list):
            _, _, _, _, embeds, _ = self.prepare_inputs_embeddings_for_multimodal(ids, None, None, None, None, img, modalities)
            prompts.append(embeds[0])
        stats = getattr(self, "collect_batch_stats", False)          # bench.py: time the two phases of THIS method (not a copy of it)
        if stats:
            import time
            torch.cuda.synchronize(); t0 = time.perf_counter()
        dec = BatchDecoder(self.lm, prompts, max_new_tokens)
        if stats:
            torch.cuda.synchronize(); t1 = time.perf_counter()
        seeds = draw_seeds(len(prompts), generator if generator is not None else getattr(self, "seed_generator", None), kwargs.get("seed"))
        toks = dec.generate(max_new_tokens, eos_token_id=self.eos_token_id, sampling=sp, seed=seeds)
        if stats:
            torch.cuda.synchronize(); t2 = time.perf_counter()
            self.batch_stats = dict(prompt_tokens=[int(e.shape[0]) for e in prompts], prefill_s=t1 - t0, decode_s=t2 - t1, steps_run=dec.steps_run,
                                    new_tokens=sum(len(t) for t in toks), nsplit=dec.nsplit)
        return [torch.tensor([t], dtype=torch.long, device=self.device) for t in toks]

    def _next(self, logits, sp, prev, seed=0):
        """one token from the last-row logits through the processor chain of `sp`; prev = ids generated so far (python list); the uniform
        of token n = len(prev) is counter_uniform(seed, n)"""
        u = ops.counter_uniform(torch.tensor([int(seed)], dtype=torch.int64, device=self.device), None, len(prev)) if sp.temperature > 0 else None
        if sp.plain:
            return int(ops.pick_token(logits, sp.temperature, u).item())
        pv = torch.tensor([prev if prev else [0]], dtype=torch.int64, device=self.device)
        return int(ops.sample_token(logits.clone(), sp.temperature, u, sp.top_k, sp.top_p, sp.repetition_penalty, prev_ids=pv, n_prev=len(prev)).item())

    @torch.no_grad()
    def generate_with_image_embedding(self, inputs=None, image_embeddings=None, modalities=["image"], do_sample=_UNSET, temperature=_UNSET,
                                      top_p=_UNSET, num_beams=1, max_new_tokens=256, use_cache=True, generator=None, top_k=_UNSET,
                                      repetition_penalty=_UNSET, **kwargs):
        """llava_qwen.py:137-155 -> Qwen2 generate(inputs_embeds=...).  Returns the NEW token ids [1, n] like HF does for
        inputs_embeds prompts.  `use_cache` is accepted for call compatibility; a KV cache is always used.  Sampling arguments follow
        HF: what the caller passes (even None) wins over the checkpoint's generation_config.json, which wins over HF's defaults
        (`resolve_sampling`); the chain repetition penalty -> temperature -> top-k -> top-p runs in sampling.hip."""
        sp = resolve_sampling(self.generation_config, do_sample, temperature, top_p, top_k, repetition_penalty)
        _, _, _, _, embeds, _ = self.prepare_inputs_embeddings_for_multimodal(inputs, None, None, None, None, image_embeddings, modalities)
        if num_beams not in (None, 1):
            # HF `generate(num_beams=N, do_sample=False)` (round 6; beam.py restates the algorithm, this is the model side): the prompt is
            # prefilled once, its cache rows replicated over N slots of a BatchDecoder; every step feeds one token per beam after gathering the
            # caches by the beams' origins.  Deterministic beam search only: beam SAMPLING (do_sample with beams) and logits processors on
            # beams are not built and say so.
            if sp.temperature > 0:
                raise NotImplementedError("beam sampling (num_beams > 1 with do_sample / temperature > 0) is not built: use temperature 0 with num_beams > 1")
            if sp.repetition_penalty != 1.0:
                raise NotImplementedError("beam search with a repetition penalty is not built")
            from .beam import beam_search
            eos = self.eos_token_id if isinstance(self.eos_token_id, (list, tuple, set)) else ([] if self.eos_token_id is None else [self.eos_token_id])
            dec = BatchDecoder(self.lm, [embeds[0]], max_new_tokens, replicate=int(num_beams))
            dec.prepare_steps(max_new_tokens)

            def step(tokens, origin):
                dec.reorder(origin)
                return dec.step(tokens.to(device=self.device, dtype=torch.int32))
            toks, self.last_beam_score = beam_search(dec.logits, step, int(num_beams), max_new_tokens, eos)
            return torch.tensor([toks], dtype=torch.long, device=self.device)
        self.lm.reset_cache(max_seq=max(self.lm.max_seq, embeds.shape[1] + max_new_tokens))
        logits = self.lm.forward(embeds[0])
        eos = self.eos_token_id if isinstance(self.eos_token_id, (list, tuple, set)) else ([] if self.eos_token_id is None else [self.eos_token_id])
        seed = int(draw_seeds(1, generator if generator is not None else getattr(self, "seed_generator", None), kwargs.get("seed"))[0]) if sp.temperature > 0 else 0
        if max_new_tokens > 1 and generator is None and kwargs.get("decode_graph", True):
            # the token loop runs as a replayed hipGraph for greedy AND for sampling (token n draws counter_uniform(seed, n) inside the graph);
            # a user `generator` (it only seeds) or decode_graph=False select the eager loop below: same tokens
            first = self._next(logits, sp, [], seed)
            if first in eos:
                return torch.tensor([[first]], dtype=torch.long, device=self.device)
            key = tuple(round(float(v), 6) for v in sp)
            dg = self._dgs.get(key)
            if dg is None or dg.out.numel() < max_new_tokens or not dg.valid():
                dg = self._dgs[key] = DecodeGraph(self.lm, max_new_tokens=max(max_new_tokens, 256), sampling=sp)
            self._dg = dg
            dg.start(first, seed)
            # round 5 (the entry point's --overlap): the prefill is done and its first token is on the host - `on_prefill_done`
            # lets the caller start the next segment's reader / updater on another host thread, and the HBM-bound token loop moves to
            # `decode_stream` (a CU partition of its own) so that the two run beside each other.  Same kernels, same tokens.
            cb, ds = kwargs.get("on_prefill_done"), kwargs.get("decode_stream")
            if cb is not None:
                cb()
            if ds is not None:
                cur = torch.cuda.current_stream(self.device)
                ds.wait_stream(cur)
                with torch.cuda.stream(ds):
                    rest = dg.run(max_new_tokens - 1, eos=eos)
                cur.wait_stream(ds)
            else:
                rest = dg.run(max_new_tokens - 1, eos=eos)
            return torch.tensor([[first] + rest], dtype=torch.long, device=self.device)
        new = []
        for step in range(max_new_tokens):
            tok = self._next(logits, sp, new, seed)
            new.append(tok)
            if tok in eos:
                break
            if step + 1 < max_new_tokens:
                logits = self.lm.forward(self.lm.embed_tokens(torch.tensor([tok], device=self.device)))
        return torch.tensor([new], dtype=torch.long, device=self.device)
