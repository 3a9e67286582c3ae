#!/usr/bin/env python3
"""Frame-encode throughput (GPU box): ViT-L/14-336 (23 layers) + mlp2x_gelu projector, 385.1 GFLOP/frame."""
import json, os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from streamchat_amd import vision as V
ap = argparse.ArgumentParser(); ap.add_argument("--frames", type=int, default=256); ap.add_argument("--mb", type=int, default=64)
a = ap.parse_args()
cfg = V.CLIPVisionConfigLite(**V.VIT_L_336)
enc = V.FrameEncoder(V.CLIPVisionTower(V.random_clip_state_dict(cfg), cfg), V.MMProjector(V.random_projector_state_dict(1024, 3584)), micro_batch=a.mb)
u8 = torch.from_numpy(np.random.default_rng(1234).integers(0, 256, (a.frames, 336, 336, 3), dtype=np.uint8)).cuda()
out = torch.empty((a.frames, 576, 3584), dtype=torch.float16, device="cuda")
enc.encode_frames_u8(u8[: a.mb], out=out[: a.mb]); torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); enc.encode_frames_u8(u8, out=out); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = min(ts)
print(json.dumps(dict(frames=a.frames, micro_batch=a.mb, ms=round(ms, 2), frames_per_s=round(a.frames / ms * 1e3, 1),
                      TFLOPs=round(385.1e9 * a.frames / ms / 1e9, 1))))
