#!/usr/bin/env python3
"""Every GEMM shape of the C3 bench step at its REAL M and epilogue: our kernel next to the vendor library (torch.matmul / addmm =
hipBLASLt / rocBLAS), interleaved on the same box.  The vendor column is a REFERENCE POINT ONLY - it is never on the product path
(VERDICT r03 item 1a).  `share_ms` = launches per step x our time: what the shape contributes to the 1024-frame step.

  python tools/bench_gemm_table.py [name ...]           one JSON line per shape + a markdown table on stderr"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streamchat_amd import ops

VM = 512 * 577          # one ViT micro-batch of the bench (2 per step)
PM = 512 * 576
LM = 48994              # C3 prefill context
# (name, M, N, K, epilogue, bias, residual, launches per step, headed mode)
SHAPES = [
    ("vit.patch",      PM, 1024, 640, "none", False, False, 2, None),
    ("vit.qkv+b",      VM, 3072, 1024, "none", True, False, 46, "colscale"),
    ("vit.o+res",      VM, 1024, 1024, "none", True, True, 46, None),
    ("vit.fc1+gelu",   VM, 4096, 1024, "quick_gelu", True, False, 46, None),
    ("vit.fc2+res",    VM, 1024, 4096, "none", True, True, 46, None),
    ("proj.1+gelu",    PM, 3584, 1024, "gelu", True, False, 2, None),
    ("proj.2",         PM, 3584, 3584, "none", True, False, 2, None),
    ("llm.q+rope",     LM, 3584, 3584, "none", True, False, 28, "rope"),
    ("llm.kv+rope",    LM, 1024, 3584, "none", True, False, 28, "rope"),
    ("llm.o+res",      LM, 3584, 3584, "none", False, True, 28, None),
    ("llm.gateup+swiglu", LM, 37888, 3584, "swiglu", False, False, 28, None),
    ("llm.down+res",   LM, 3584, 18944, "none", False, True, 28, None),
    ("8192^3",         8192, 8192, 8192, "none", False, False, 0, None),
]


def timeit(fn, reps=5, inner=4):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return sorted(ts)[len(ts) // 2]


def main():
    want = sys.argv[1:]
    rows = []
    for (name, M, N, K, epi, has_b, has_r, per_step, headed) in SHAPES:
        if want and name not in want:
            continue
        a_rows = (576, 577, 1) if name == "proj.1+gelu" else None          # the projector's CLS-drop row map (k_gemm256 fallback), as vision.MMProjector calls it
        a = (torch.rand(VM if a_rows else M, K, device="cuda") * 2 - 1).half()
        w = (torch.rand(N, K, device="cuda") * 2 - 1).half()
        bias = (torch.rand(N, device="cuda") - 0.5).half() if has_b else None
        n_out = N // 2 if epi == "swiglu" else N
        res = (torch.rand(M, n_out, device="cuda") - 0.5).half() if has_r else None
        out = torch.empty(M, n_out, device="cuda", dtype=torch.float16)
        vout = torch.empty(M, N, device="cuda", dtype=torch.float16)
        if headed == "rope":
            tab = ops.rope_table(65536, 128, 1e6, 1.0, "cuda")
            lead = N if N == 3584 else 512
            ours = lambda: ops.gemm_headed(a, w, bias, out, "rope", lead, rope_tab=tab, pos0=0)
        elif headed == "colscale":
            ours = lambda: ops.gemm_headed(a, w, bias, out, "colscale", 1024, col_scale=0.18)
        else:
            ours = lambda: ops.gemm(a, w, bias, res, epi, out=out, a_rows=a_rows, M=M if a_rows else None)
        wt = w.t()
        va = a[:M]
        if has_b:
            vend = lambda: torch.addmm(bias, va, wt, out=vout)          # bias fused by the library; activation / residual NOT included
        else:
            vend = lambda: torch.matmul(va, wt, out=vout)
        # round 5: the SAME work from the library = its GEMM + the elementwise kernel(s) our epilogue fuses (residual add / quick-GELU / GELU);
        # rotary, SwiGLU and the column scale have no one-call torch equivalent and keep the GEMM-only comparison
        if has_r:
            vend_full = lambda: (vend(), vout.add_(res))
        elif epi == "quick_gelu":
            vend_full = lambda: (vend(), vout.mul_(torch.sigmoid(1.702 * vout)))
        elif epi == "gelu":
            vend_full = lambda: (vend(), torch.nn.functional.gelu(vout, approximate="none"))
        else:
            vend_full = None
        t_o, t_v, t_f = [], [], []
        for _ in range(3):              # interleave ours / vendor: both see the same thermal / power state
            t_o.append(timeit(ours))
            t_v.append(timeit(vend))
            if vend_full is not None:
                t_f.append(timeit(vend_full))
        mo, mv = sorted(t_o)[1], sorted(t_v)[1]
        mf = sorted(t_f)[1] if t_f else None
        fl = 2.0 * M * N * K
        rec = dict(name=name, M=M, N=N, K=K, ours_ms=round(mo, 4), ours_TF=round(fl / mo / 1e9, 1), vendor_ms=round(mv, 4), vendor_TF=round(fl / mv / 1e9, 1),
                   ours_over_vendor=round(mv / mo, 3), launches_per_step=per_step, share_ms=round(per_step * mo, 1),
                   vendor_note="addmm (bias only)" if has_b else "matmul (no epilogue)",
                   vendor_same_work_ms=None if mf is None else round(mf, 4), ours_over_vendor_same_work=None if mf is None else round(mf / mo, 3))
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        del a, w, out, vout, res
        torch.cuda.empty_cache()
    sys.stderr.write("| shape | M | N | K | ours ms | ours TF | vendor GEMM ms | vendor TF | ours/vendor GEMM | vendor GEMM + the fused elementwise op, ms | ours / that | x per step | ms per step |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        sys.stderr.write(f"| {r['name']} | {r['M']} | {r['N']} | {r['K']} | {r['ours_ms']} | {r['ours_TF']} | {r['vendor_ms']} | {r['vendor_TF']} | "
                         f"{r['ours_over_vendor']} | {r['vendor_same_work_ms'] or '-'} | {r['ours_over_vendor_same_work'] or '-'} | {r['launches_per_step']} | {r['share_ms']} |\n")


if __name__ == "__main__":
    main()
