#!/usr/bin/env python3
"""PCIe-inclusive encode rate: 1024 HOST uint8 frames -> AsyncFrameIngest (pinned staging, copy stream) -> ViT-L/14-336 + projector,
against the same encode with the frames already resident in HBM (what bench.py's `value` times)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from streamchat_amd import vision as V
from streamchat_amd.ingest import AsyncFrameIngest

n, mb = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = V.CLIPVisionConfigLite(**V.VIT_L_336)
enc = V.FrameEncoder(V.CLIPVisionTower(V.random_clip_state_dict(cfg, seed=0), cfg), V.MMProjector(V.random_projector_state_dict(1024, 3584, seed=1)), micro_batch=mb)
host = np.random.default_rng(0).integers(0, 256, (n, 336, 336, 3), dtype=np.uint8)
dev = torch.from_numpy(host).cuda()
bank = torch.empty((n, 576, 3584), dtype=torch.float16, device="cuda")
enc.encode_frames_u8(dev[:mb], out=bank[:mb]); torch.cuda.synchronize()
res = {}
for rep in range(2):
    t0 = time.perf_counter(); enc.encode_frames_u8(dev, out=bank); torch.cuda.synchronize(); res["resident_s"] = time.perf_counter() - t0
    ing = AsyncFrameIngest(enc.encode_frames_u8, (336, 336, 3), micro_batch=mb, depth=2)
    t0 = time.perf_counter(); ing.run(iter(host), bank); torch.cuda.synchronize(); res["from_host_s"] = time.perf_counter() - t0
print(json.dumps(dict(frames=n, micro_batch=mb, resident_frames_per_s=round(n / res["resident_s"], 1), from_host_frames_per_s=round(n / res["from_host_s"], 1),
                      producer_waits=ing.stats["producer_waits"], **{k: round(v, 4) for k, v in res.items()})))
