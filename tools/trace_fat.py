#!/usr/bin/env python3
"""Per-tile phase timing of k_gemm_fat.  Diagnostic build (the product library has no trace code: gemm.hip only carries three empty
FAT_STAMP hooks): `hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -include tools/diag/fat_trace.h -c streamchat_amd/csrc/gemm.hip -o
tools/bin/gemm_trace.o`, linked with the other objects as tools/bin/lib_trace.so, loaded through SC_LIB.  Each workgroup stamps the 100 MHz counter at tile start / end of the K loop / end of the epilogue; this prints, per
shape, the K-loop and epilogue durations and how far the workgroups' epilogues are apart in time (all CUs storing their 128 KB C tiles
in the same few microseconds is a burst the write path has to absorb).

    SC_LIB=$PWD/tools/bin/lib_trace.so python tools/trace_fat.py vit512.qkv+b vit512.fc1+gelu vit512.o+res vit512.fc2+res llm49k.q"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from streamchat_amd import ops, _lib

SHAPES = {"vit512.qkv+b": (512 * 577, 3072, 1024), "vit512.o+res": (512 * 577, 1024, 1024), "vit512.fc1+gelu": (512 * 577, 4096, 1024),
          "vit512.fc2+res": (512 * 577, 1024, 4096), "llm49k.q": (48994, 3584, 3584), "llm49k.down": (48994, 3584, 18944), "llm49k.o+res": (48994, 3584, 3584), "llm49k.gateup+swiglu": (48994, 37888, 3584)}
TILES = 96
lib = _lib.load()
lib.sc_fat_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.sc_fat_trace_read.restype = ctypes.c_int

for name in (sys.argv[1:] or list(SHAPES)):
    M, N, K = SHAPES[name]
    a = (torch.rand(M, K, device="cuda") * 2 - 1).half()
    w = (torch.rand(N, K, device="cuda") * 2 - 1).half()
    epi = "quick_gelu" if "+gelu" in name else ("swiglu" if "+swiglu" in name else "none")
    bias = (torch.rand(N, device="cuda") - 0.5).half() if "+" in name else None
    res = (torch.rand(M, N, device="cuda") - 0.5).half() if "+res" in name else None
    out = torch.empty(M, N // 2 if epi == "swiglu" else N, device="cuda", dtype=torch.float16)
    for _ in range(4):
        ops.gemm(a, w, bias, res, epi, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(a, w, bias, res, epi, out=out); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(256 * TILES * 3, dtype=np.uint64)
    assert lib.sc_fat_trace_read(buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes) == 0
    t = buf.reshape(256, TILES, 3).astype(np.int64)
    ntile = -(-M // 256) * -(-N // 256)
    per = min(TILES, ntile // 256)                       # full rounds every workgroup has
    t = t[:, :per]
    t0 = t[:, 0, 0].min()
    loop = (t[:, :, 1] - t[:, :, 0]) * 0.01              # us
    epi_t = (t[:, :, 2] - t[:, :, 1]) * 0.01
    gap = (t[:, 1:, 0] - t[:, :-1, 2]) * 0.01
    start = (t[:, :, 1] - t0) * 0.01                     # epilogue start times
    spread = [float(np.percentile(start[:, i], 90) - np.percentile(start[:, i], 10)) for i in (0, per // 2, per - 1)]
    print(json.dumps(dict(name=name, ms=round(e0.elapsed_time(e1), 4), tiles_per_wg=per, loop_us=[round(float(x), 2) for x in (loop.mean(), np.percentile(loop, 10), np.percentile(loop, 90))],
                          epilogue_us=[round(float(x), 2) for x in (epi_t.mean(), np.percentile(epi_t, 10), np.percentile(epi_t, 90))],
                          between_tiles_us=round(float(gap.mean()), 2), first_tile_loop_us=round(float(loop[:, 0].mean()), 2),
                          epilogue_start_spread_p10_p90_us_first_mid_last=[round(x, 2) for x in spread],
                          tile_us=round(float((t[:, -1, 2] - t[:, 0, 0]).mean()) * 0.01 / per, 2))))
