"""Frame encoder: CLIP ViT-L/14-336 tower + mlp2x_gelu projector on hand-written gfx950 kernels.

Mirrors the reference seam `model.encode_images(images[N,3,336,336] fp16) -> [N,576,3584]`
(longva/model/llava_arch.py:179-184 -> CLIPVisionTower.forward / feature_select,
longva/model/multimodal_encoder/clip_encoder.py:46-79 -> mm_projector, multimodal_projector/builder.py:41-48
-> IdentityMap resampler, multimodal_resampler/builder.py:9-14).

Differences from the reference that do not change results:
  * only the layers that feed `hidden_states[select_layer]` run (23 of 24 for select_layer = -2; the reference
    also executes layer 24 and post_layernorm and keeps all 25 hidden states alive — SURVEY.md §0 item 7);
  * frames are encoded in bounded micro-batches with one resident set of activation buffers;
  * q/k/v projections are one fused GEMM; the CLS row is dropped by the projector GEMM's row map (no copy).
Weights use the transformers state-dict names, so a real checkpoint loads unchanged."""
import math

import torch

from . import ops


class CLIPVisionConfigLite:
    def __init__(self, hidden=1024, layers=24, heads=16, intermediate=4096, patch=14, image_size=336, eps=1e-5):
        self.hidden, self.layers, self.heads, self.intermediate = hidden, layers, heads, intermediate
        self.patch, self.image_size, self.eps = patch, image_size, eps

    @property
    def num_patches(self):
        return (self.image_size // self.patch) ** 2


VIT_L_336 = dict(hidden=1024, layers=24, heads=16, intermediate=4096, patch=14, image_size=336)


def random_clip_state_dict(cfg: CLIPVisionConfigLite, seed=0, device="cuda", dtype=torch.float16, std=0.02):
    """Random-init weights of the CLIP vision architecture under the transformers parameter names."""
    g = torch.Generator(device=device).manual_seed(seed)
    D, I, P = cfg.hidden, cfg.intermediate, cfg.patch

    def rn(*shape, s=std):
        return (torch.randn(*shape, device=device, generator=g) * s).to(dtype)
    sd = {"vision_model.embeddings.class_embedding": rn(D),
          "vision_model.embeddings.patch_embedding.weight": rn(D, 3, P, P),
          "vision_model.embeddings.position_embedding.weight": rn(cfg.num_patches + 1, D),
          "vision_model.pre_layrnorm.weight": 1 + rn(D), "vision_model.pre_layrnorm.bias": rn(D)}
    for i in range(cfg.layers):
        p = f"vision_model.encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{n}.weight"] = rn(D, D)
            sd[p + f"self_attn.{n}.bias"] = rn(D)
        sd[p + "layer_norm1.weight"] = 1 + rn(D); sd[p + "layer_norm1.bias"] = rn(D)
        sd[p + "layer_norm2.weight"] = 1 + rn(D); sd[p + "layer_norm2.bias"] = rn(D)
        sd[p + "mlp.fc1.weight"] = rn(I, D); sd[p + "mlp.fc1.bias"] = rn(I)
        sd[p + "mlp.fc2.weight"] = rn(D, I); sd[p + "mlp.fc2.bias"] = rn(D)
    return sd


def random_projector_state_dict(d_in, d_out, seed=1, device="cuda", dtype=torch.float16, std=0.02):
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s: (torch.randn(*s, device=device, generator=g) * std).to(dtype)
    return {"0.weight": rn(d_out, d_in), "0.bias": rn(d_out), "2.weight": rn(d_out, d_out), "2.bias": rn(d_out)}


def _h(t, device):
    return t.detach().to(device=device, dtype=torch.float16).contiguous()


class CLIPVisionTower:
    """HIP replacement of the reference's CLIPVisionTower (clip_encoder.py:9-79)."""

    def __init__(self, state_dict, cfg: CLIPVisionConfigLite, select_layer=-2, select_feature="patch", device="cuda",
                 prefix="vision_model."):
        self.cfg, self.device = cfg, torch.device(device)
        self.select_layer, self.select_feature = select_layer, select_feature
        self.layers_run = cfg.layers + 1 + select_layer if select_layer < 0 else select_layer
        if not (0 <= self.layers_run <= cfg.layers):
            raise ValueError(f"select_layer {select_layer} out of range")
        if select_feature not in ("patch", "cls_patch"):
            raise ValueError(f"Unexpected select feature: {select_feature}")            # clip_encoder.py:55-58
        D, P = cfg.hidden, cfg.patch
        if D % 128 or cfg.intermediate % 128 or (D // cfg.heads) != 64:
            raise ValueError("HIP ViT path needs hidden/intermediate multiples of 128 and head_dim 64")
        sd, p = state_dict, prefix
        self.kpad = (3 * P * P + 63) // 64 * 64
        w = _h(sd[p + "embeddings.patch_embedding.weight"], device).reshape(D, 3 * P * P)
        self.patch_w = torch.zeros((D, self.kpad), dtype=torch.float16, device=device)
        self.patch_w[:, : 3 * P * P] = w
        self.cls = _h(sd[p + "embeddings.class_embedding"], device)
        self.pos = _h(sd[p + "embeddings.position_embedding.weight"], device)
        self.pre_g, self.pre_b = _h(sd[p + "pre_layrnorm.weight"], device), _h(sd[p + "pre_layrnorm.bias"], device)
        self.L = []
        for i in range(self.layers_run):
            lp = f"{p}encoder.layers.{i}."
            q = lambda n: _h(sd[lp + n], device)
            self.L.append(dict(
                ln1=(q("layer_norm1.weight"), q("layer_norm1.bias")), ln2=(q("layer_norm2.weight"), q("layer_norm2.bias")),
                wqkv=torch.cat([q("self_attn.q_proj.weight"), q("self_attn.k_proj.weight"), q("self_attn.v_proj.weight")]).contiguous(),
                bqkv=torch.cat([q("self_attn.q_proj.bias"), q("self_attn.k_proj.bias"), q("self_attn.v_proj.bias")]).contiguous(),
                wo=q("self_attn.out_proj.weight"), bo=q("self_attn.out_proj.bias"),
                w1=q("mlp.fc1.weight"), b1=q("mlp.fc1.bias"), w2=q("mlp.fc2.weight"), b2=q("mlp.fc2.bias")))
        self._buf, self._buf_n = None, 0

    # activation buffers of one micro-batch, allocated once
    def _buffers(self, n):
        if self._buf is None or self._buf_n < n:
            c, dev = self.cfg, self.device
            M = n * (c.num_patches + 1)
            e = lambda *s: torch.empty(*s, dtype=torch.float16, device=dev)
            self._buf = dict(patches=e(n * c.num_patches, self.kpad), pe=e(n * c.num_patches, c.hidden), h=e(M, c.hidden), x=e(M, c.hidden),
                             qkv=e(M, 3 * c.hidden), att=e(M, c.hidden), f=e(M, c.intermediate), h2=e(M, c.hidden))
            self._buf_n = n
        return self._buf

    def hidden_from_patches(self, n):
        """runs the tower on self._buf['patches'][: n*P]; returns the [n*(P+1), D] hidden state (a buffer view)."""
        c, B = self.cfg, self._buf
        P, D, S = c.num_patches, c.hidden, c.num_patches + 1
        M = n * S
        pe = ops.gemm(B["patches"][: n * P], self.patch_w, out=B["pe"][: n * P])
        h = ops.vit_embed_ln(pe, self.cls, self.pos, self.pre_g, self.pre_b, c.eps, n, P, out=B["h"][:M])
        h2 = B["h2"][:M]
        # the q third of the fused q|k|v projection leaves the GEMM already multiplied by the softmax scale * log2 e (applied to the fp32
        # sum, one rounding - HF's CLIPAttention scales q the same way) where the hand-scheduled kernel serves the shape; attention then
        # runs without a per-score multiply (SC_ATTN_Q_PRESCALED)
        # (decided for ALL layers together: the attention mode must not change inside a frame batch; one unaligned layer weight sends every
        # layer through the plain GEMM + in-kernel scale)
        pre = bool(self.L) and all(ops.gemm_headed_ok(3 * D, D, B["x"][:M], L["wqkv"], L["bqkv"], B["qkv"][:M]) for L in self.L)
        for L in self.L:
            x = ops.layernorm(h, L["ln1"][0], L["ln1"][1], c.eps, out=B["x"][:M])
            if pre:
                qkv = ops.gemm_headed(x, L["wqkv"], L["bqkv"], B["qkv"][:M], "colscale", D, col_scale=0.125 * ops.LOG2E).view(n, S, 3 * D)
            else:
                qkv = ops.gemm(x, L["wqkv"], L["bqkv"], out=B["qkv"][:M]).view(n, S, 3 * D)
            att = ops.attention(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], c.heads, c.heads, 64, 0.125,
                                out=B["att"][:M].view(n, S, D), q_prescaled=pre).view(M, D)
            ops.gemm(att, L["wo"], L["bo"], residual=h, out=h2)
            x = ops.layernorm(h2, L["ln2"][0], L["ln2"][1], c.eps, out=B["x"][:M])
            f = ops.gemm(x, L["w1"], L["b1"], epilogue="quick_gelu", out=B["f"][:M])
            ops.gemm(f, L["w2"], L["b2"], residual=h2, out=h)
        return h


class MMProjector:
    """mlp2x_gelu (reference multimodal_projector/builder.py:41-48)."""

    def __init__(self, state_dict, device="cuda", prefix=""):
        self.w0, self.b0 = _h(state_dict[prefix + "0.weight"], device), _h(state_dict[prefix + "0.bias"], device)
        self.w2, self.b2 = _h(state_dict[prefix + "2.weight"], device), _h(state_dict[prefix + "2.bias"], device)
        self.d_out = self.w2.shape[0]
        self._mid = None

    def __call__(self, hidden, n, P, drop_cls=True, out=None, pool=1):
        """`pool` = r > 1 (reference compress_spatial_features utiles.py:264-289 / --compress_rate): the r x r spatial mean of the
        projected token map.  The mean commutes with the last Linear (pool(mid) W2^T + b2 = pool(mid W2^T + b2)), so it is applied to the
        post-GELU activations and the second GEMM runs on 1 / r^2 of the rows: the pooled features cost LESS than the unpooled ones, where an
        epilogue on the full-size GEMM would only save the store.  Rounding: one fp16 rounding of the pooled activations (fp32 mean) instead of
        one of every unpooled output - inside the encoder tolerance (tests/test_gpu_vision.py)."""
        M = n * P
        if self._mid is None or self._mid.shape[0] < M:
            self._mid = torch.empty((M, self.w0.shape[0]), dtype=torch.float16, device=hidden.device)
        # `feature_select` 'patch' (clip_encoder.py:53-58): the CLS rows are skipped by the GEMM's A row map (k_gemm256, 0.71 PF, 3.1 ms per 512
        # frames).  Round 3 tried one strided copy + the hand-scheduled kernel instead: its erf-GELU epilogue (erff: ~40 VALU instructions per
        # value, nothing to overlap them with at one wave per SIMD) ran 3.36 ms - slower; reverted (profiles/r03_bench_kernel_stats.md history).
        rows = (P, P + 1, 1) if drop_cls else None
        mid = ops.gemm(hidden, self.w0, self.b0, epilogue="gelu", out=self._mid[:M], a_rows=rows, M=M)
        if pool > 1:
            if not drop_cls:
                raise ops.StreamChatHipError("MMProjector: spatial pooling needs the patch grid alone (select_feature 'patch')")
            mid = ops.avgpool_tokens(mid.view(n, P, -1), pool).view(-1, mid.shape[1])
        return ops.gemm(mid, self.w2, self.b2, out=out)


class FrameEncoder:
    """`encode_images` of the reference (llava_arch.py:179-184): vision tower -> projector -> identity resampler."""

    def __init__(self, tower: CLIPVisionTower, projector: MMProjector, micro_batch=512, compress_rate=1):
        """compress_rate r > 1: every frame leaves the encoder as (grid // r)^2 tokens, the r x r spatial means of its projected patch map (the
        reference's `compress_spatial_features(feature_list, compress_rate)`, utiles.py:264-289, fused into the projector)."""
        self.tower, self.projector, self.micro_batch, self.compress_rate = tower, projector, micro_batch, int(compress_rate)

    def _run(self, n_total, fill_patches, out):
        c = self.tower.cfg
        P = c.num_patches
        drop = self.tower.select_feature == "patch"                      # clip_encoder.py:53-58
        tokens = P if drop else P + 1
        r = self.compress_rate
        out_tokens = tokens if r <= 1 else (int(round(math.sqrt(P))) // r) ** 2
        if out is None:
            out = torch.empty((n_total, out_tokens, self.projector.d_out), dtype=torch.float16, device=self.tower.device)
        mb = min(self.micro_batch, n_total)
        self.tower._buffers(mb)
        for s in range(0, n_total, mb):
            n = min(mb, n_total - s)
            fill_patches(s, n, self.tower._buf["patches"])
            h = self.tower.hidden_from_patches(n)
            self.projector(h, n, tokens, drop_cls=drop, out=out[s:s + n].view(n * out_tokens, -1), pool=r)
        return out

    def encode_images(self, images, out=None):
        """images: [N, 3, H, W] fp16 normalised pixel values (what the reference passes) -> [N, 576, 3584] fp16."""
        c = self.tower.cfg
        images = images.to(device=self.tower.device, dtype=torch.float16)
        if tuple(images.shape[2:]) != (c.image_size, c.image_size):
            raise ValueError(f"encode_images: pixel values are {tuple(images.shape[2:])}, the tower takes {c.image_size} x {c.image_size}")
        return self._run(images.shape[0], lambda s, n, buf: ops.patchify_f16(images[s:s + n], c.patch, self.tower.kpad,
                                                                            out=buf[: n * c.num_patches]), out)

    def encode_frames_u8(self, frames, out=None):
        """frames: uint8 [N, H, W, 3] RGB — preprocessing (reference utiles.py:71-87) fused into the patch gather."""
        c = self.tower.cfg
        if tuple(frames.shape[1:3]) != (c.image_size, c.image_size):
            raise ValueError(f"encode_frames_u8: frames are {tuple(frames.shape[1:3])}, the tower takes {c.image_size} x {c.image_size} "
                             "(resize + centre-crop first: streamchat_amd.mm_utils.resize_center_crop_u8, the CLIPImageProcessor step of utiles.py:71-87)")
        return self._run(frames.shape[0], lambda s, n, buf: ops.preprocess_patchify_u8(frames[s:s + n], c.patch, self.tower.kpad,
                                                                                      out=buf[: n * c.num_patches]), out)
