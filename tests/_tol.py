"""Shared closeness check for fp16-storage / fp32-accumulate kernels against an fp32 reference (a bound that is
only normalised by the output's MAX magnitude lets a systematic error hide in small-magnitude channels).  Three bounds, all asserted:
  max |err|  <= max_rel * max|ref|      worst element            (observed 0.5e-3 .. 1.7e-3: tiny models to ViT-L 23 layers)
  rms(err)   <= rms_rel * rms(ref)      energy of the error over the WHOLE tensor: a per-channel bias shows up here
  cosine(out, ref) per row >= cos_min   direction of every row / every logit vector"""
import torch


def assert_close_fp16(out, ref, max_rel=4e-3, rms_rel=3e-3, cos_min=0.9999, what=""):
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    err = (out - ref).abs()
    mx, scale = err.max().item(), ref.abs().max().item()
    rms = (err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp(min=1e-30)).item()
    o2, r2 = out.reshape(-1, out.shape[-1]), ref.reshape(-1, ref.shape[-1])
    keep = r2.norm(dim=1) > 1e-6 * r2.norm(dim=1).max()
    cos = torch.nn.functional.cosine_similarity(o2[keep], r2[keep], dim=1).min().item()
    print(f"\n[tol] {what}: max|err| {mx:.3e} = {mx / scale:.2e} of max|ref|, rms rel {rms:.2e}, min row cosine {cos:.6f}")
    assert mx <= max_rel * scale, (what, mx, scale)
    assert rms <= rms_rel, (what, rms)
    assert cos >= cos_min, (what, cos)
