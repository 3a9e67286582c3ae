"""GPU: the HIP frame encoder (`encode_images`) against (a) the golden vectors produced by the real HF
CLIPVisionModel (tools/make_golden_hf.py) and (b) the fp32 PyTorch restatement at ViT-L/14-336 size.
Tolerance (tests/_tol.py): max|err| <= 4e-3 max|ref|, rms(err) <= 3e-3 rms(ref), per-row cosine >= 0.9999 (fp16 storage, fp32 accumulate)
(observed ~3e-3); stated per test."""
import os

import numpy as np
import pytest
import torch

from tests._tol import assert_close_fp16

from oracle import torch_ref as R
from streamchat_amd import ops, vision as V

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _tiny():
    d = np.load(os.path.join(G, "clip_tiny.npz"))
    sd = {k[4:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("vit.")}
    sp = {k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("proj.")}
    cfg = V.CLIPVisionConfigLite(hidden=128, layers=3, heads=2, intermediate=256, patch=14, image_size=56)
    return d, sd, sp, cfg


def test_encode_images_tiny_vs_hf_golden():
    d, sd, sp, cfg = _tiny()
    enc = V.FrameEncoder(V.CLIPVisionTower(sd, cfg), V.MMProjector(sp), micro_batch=2)      # 3 frames -> two micro-batches
    out = enc.encode_images(torch.from_numpy(d["pixel_values"]).cuda().half())
    ref = torch.from_numpy(d["projected"]).cuda()
    assert out.shape == ref.shape == (3, 16, 256)
    assert_close_fp16(out, ref, what="tiny CLIP + projector vs HF golden")


def test_feature_select_layer_and_cls_patch():
    d, sd, sp, cfg = _tiny()
    tower = V.CLIPVisionTower(sd, cfg, select_layer=-1, select_feature="cls_patch")
    enc = V.FrameEncoder(tower, V.MMProjector(sp))
    px = torch.from_numpy(d["pixel_values"])
    out = enc.encode_images(px.cuda().half())
    h = R.clip_vision_hidden(sd, px, heads=2, patch=14, layers_run=3)
    ref = R.mm_projector(sp, h).cuda()
    assert out.shape == (3, 17, 256)
    assert_close_fp16(out, ref, what="tiny CLIP select_layer=-1 cls_patch")
    with pytest.raises(ValueError):
        V.CLIPVisionTower(sd, cfg, select_feature="bogus")


def test_encode_frames_u8_equals_preprocess_then_encode():
    """fused u8 path == preprocess_u8 + encode_images, bit for bit (same kernels downstream)."""
    d, sd, sp, cfg = _tiny()
    enc = V.FrameEncoder(V.CLIPVisionTower(sd, cfg), V.MMProjector(sp))
    u8 = torch.from_numpy(np.random.default_rng(1234).integers(0, 256, (2, 56, 56, 3), dtype=np.uint8)).cuda()
    a = enc.encode_frames_u8(u8)
    b = enc.encode_images(ops.preprocess_u8(u8))
    assert torch.equal(a, b)


def test_encode_images_vit_l_vs_torch_fp32():
    cfg = V.CLIPVisionConfigLite(**V.VIT_L_336)
    sd = V.random_clip_state_dict(cfg, seed=0)
    sp = V.random_projector_state_dict(1024, 3584, seed=1)
    enc = V.FrameEncoder(V.CLIPVisionTower(sd, cfg), V.MMProjector(sp), micro_batch=2)
    u8 = torch.from_numpy(np.random.default_rng(1234).integers(0, 256, (3, 336, 336, 3), dtype=np.uint8)).cuda()
    out = enc.encode_frames_u8(u8)
    assert out.shape == (3, 576, 3584) and out.dtype == torch.float16
    px = ops.preprocess_u8(u8).float()
    ref = R.encode_images({k: v.float() for k, v in sd.items()}, {k: v.float() for k, v in sp.items()}, px, heads=16, patch=14, num_layers=24)
    assert_close_fp16(out, ref, what="ViT-L/14-336 + mlp2x_gelu vs fp32 torch_ref")
    # frames are independent: encoding frame 1 alone gives the same rows as inside the batch
    single = enc.encode_frames_u8(u8[1:2])
    assert torch.equal(single[0], out[1])


def test_async_ingest_equals_synchronous_encode():
    """host frames -> pinned staging -> copy stream -> encode (streamchat_amd/ingest.py) gives the same bank, bit for bit, as one
    synchronous encode_frames_u8 with the same micro-batch, including a ragged tail and more micro-batches than staging slots."""
    from streamchat_amd.ingest import AsyncFrameIngest
    d, sd, sp, cfg = _tiny()
    enc = V.FrameEncoder(V.CLIPVisionTower(sd, cfg), V.MMProjector(sp), micro_batch=4)
    frames = np.random.default_rng(5).integers(0, 256, (19, 56, 56, 3), dtype=np.uint8)
    ref = enc.encode_frames_u8(torch.from_numpy(frames).cuda())
    bank = torch.zeros(19, 16, 256, dtype=torch.float16, device="cuda")
    ing = AsyncFrameIngest(enc.encode_frames_u8, (56, 56, 3), micro_batch=4, depth=2)
    assert ing.run(iter(frames), bank) == 19 and ing.stats["micro_batches"] == 5
    torch.cuda.synchronize()
    assert torch.equal(bank, ref)


def test_video_reader_host_frames_go_through_async_ingest_and_match_device_frames():
    """reference inference_streaming_longva_v2.py:454-531 with a HOST decoder (numpy frames, as cv2 yields them): the reader uses the
    overlapped ingest pipeline; the feature bank equals the one built from the same frames already on the device."""
    from streamchat_amd import llm as LM, streaming as S
    d, sd, sp, cfg = _tiny()
    enc = V.FrameEncoder(V.CLIPVisionTower(sd, cfg), V.MMProjector(sp))
    model = LM.LlavaQwenForCausalLM.__new__(LM.LlavaQwenForCausalLM)
    model.frame_encoder, model.device = enc, torch.device("cuda")
    frames = np.random.default_rng(9).integers(0, 256, (90, 56, 56, 3), dtype=np.uint8)

    class HostCap:
        def read_rgb(self, i):
            return frames[i] if i < len(frames) else None

    class DevCap:
        def read_rgb(self, i):
            return torch.from_numpy(frames[i]).cuda() if i < len(frames) else None
    a = S.video_reader_thread_with_embedding(HostCap(), 90, 2, None, model, 0, 40, "cuda", 0.5, chunk_size=4)
    b = S.video_reader_thread_with_embedding(DevCap(), 90, 2, None, model, 0, 40, "cuda", 0.5, chunk_size=4)
    torch.cuda.synchronize()
    assert len(a) == len(b) > 0 and a[0].shape == (1, 16, 256)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert a[1].data_ptr() - a[0].data_ptr() == a[0].numel() * 2                  # views into ONE contiguous bank


@pytest.mark.parametrize("r", [2, 4])
def test_compress_rate_fused_into_the_projector(r):
    """FrameEncoder(compress_rate=r): the r x r spatial mean applied to the post-GELU activations, second projector GEMM on
    1 / r^2 of the rows == the reference's compress_spatial_features (utiles.py:264-289: F.avg_pool2d of the projected [B, D, P, P] map) on the
    unpooled features, within the encoder tolerance - against the HF golden features pooled in fp32 and against our own unpooled path + the
    drop-in utiles.compress_spatial_features."""
    d, sd, sp, cfg = _tiny()
    px = torch.from_numpy(d["pixel_values"]).cuda().half()
    pooled = V.FrameEncoder(V.CLIPVisionTower(sd, cfg), V.MMProjector(sp), compress_rate=r).encode_images(px)
    g = 4 // r
    assert pooled.shape == (3, g * g, 256)
    ref = torch.from_numpy(d["projected"])                                       # [3, 16, 256] fp32 from the HF modules
    ref = torch.nn.functional.avg_pool2d(ref.reshape(3, 4, 4, 256).permute(0, 3, 1, 2), (r, r)).permute(0, 2, 3, 1).reshape(3, g * g, 256)
    assert_close_fp16(pooled, ref, what=f"pooled projector r={r} vs avg_pool2d of the HF golden")
    from streamchat_amd import utiles as U
    full = V.FrameEncoder(V.CLIPVisionTower(sd, cfg), V.MMProjector(sp)).encode_images(px)
    two_step = torch.cat(U.compress_spatial_features([full[i:i + 1] for i in range(3)], r))
    assert_close_fp16(pooled, two_step.float(), max_rel=2e-3, rms_rel=1.5e-3, what=f"pooled projector r={r} vs encode + compress_spatial_features")
