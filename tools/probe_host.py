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
if __name__ == "__main__":
    print(json.dumps(info), flush=True)
if __name__ == "__main__" and len(sys.argv) > 1:
    import numpy as np, torch
    from oracle import torch_ref as R
    from streamchat_amd import vision as V
    cfg = V.CLIPVisionConfigLite(**V.VIT_L_336)
    sd = V.random_clip_state_dict(cfg, seed=0, device="cpu"); sp = V.random_projector_state_dict(1024, 3584, seed=1, device="cpu")
    u8 = np.random.default_rng(0).integers(0, 256, (64, 336, 336, 3), dtype=np.uint8)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")      # inherited by the spawned workers: waiting threads sleep instead of burning the CPU quota
    os.environ.setdefault("KMP_BLOCKTIME", "0")
    for workers, threads, batch in ((16, 1, 4), (16, 2, 4), (8, 2, 8), (8, 4, 8), (4, 4, 8), (16, 4, 4), (16, 16, 4), (32, 1, 2), (1, 16, 8), (16, 16, 4)):
        n = 64
        c0 = os.times()
        t0 = time.time()
        R.encode_frames_u8_parallel(sd, sp, u8[:n], workers=workers, threads=threads, batch=batch)
        dt = time.time() - t0
        c1 = os.times()
        cpu = (c1.children_user + c1.children_system + c1.user + c1.system) - (c0.children_user + c0.children_system + c0.user + c0.system)
        print(json.dumps(dict(workers=workers, threads=threads, batch=batch, frames=n, s=round(dt, 2), s_per_frame=round(dt / n, 3), cpu_s=round(cpu, 1),
                              cores_busy=round(cpu / dt, 1), loadavg=open("/proc/loadavg").read().split()[0])), flush=True)
