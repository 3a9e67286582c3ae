#!/usr/bin/env python3
"""What the GPU box's host really gives a process: cores present, scheduling affinity, cgroup CPU quota, memory; and how the fp32 ViT-L forward
scales over threads / worker processes (oracle/torch_ref, the cpu_baseline leg of bench.py and the composed tests use it)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
info = dict(cpu_count=os.cpu_count(), affinity=len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective", "/proc/loadavg"):
    try:
        info[f] = open(f).read().strip()
    except Exception as e:
        info[f] = None
print(json.dumps(info), flush=True)
if __name__ == "__main__" and len(sys.argv) > 1:
    import numpy as np, torch
    from oracle import torch_ref as R
    from streamchat_amd import vision as V
    cfg = V.CLIPVisionConfigLite(**V.VIT_L_336)
    sd = V.random_clip_state_dict(cfg, seed=0, device="cpu"); sp = V.random_projector_state_dict(1024, 3584, seed=1, device="cpu")
    u8 = np.random.default_rng(0).integers(0, 256, (64, 336, 336, 3), dtype=np.uint8)
    for workers, threads in ((1, 16), (1, 32), (1, 64), (2, 32), (4, 32), (8, 32), (4, 16), (8, 16), (16, 16)):
        n = 8 * workers * 2
        t0 = time.time()
        R.encode_frames_u8_parallel(sd, sp, u8[:n], workers=workers, threads=threads, batch=8)
        dt = time.time() - t0
        print(json.dumps(dict(workers=workers, threads=threads, frames=n, s=round(dt, 2), s_per_frame=round(dt / n, 3))), flush=True)
