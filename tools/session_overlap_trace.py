#!/usr/bin/env python3
"""How much of the session's answer decode really ran BESIDE the next segment's MFMA work?  Reads a rocprofv3 kernel trace (rocpd sqlite) of

    rocprofv3 --kernel-trace -d /tmp/kt -o s -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --with-captions 0 --decode-tokens 512 --session 3

classifies every dispatch as HBM-side (the token loop: GEMV / skinny GEMM / decode attention / merge / token selection) or MFMA-side (tile GEMMs, tile
attention, k-means, norms of the prefill and the encoder) and measures, with a sweep over the start / end timestamps, the time during which at least one
kernel of EACH side was executing.  On one stream nothing ever co-runs, so all of that time belongs to the overlapped half of the session.

    python tools/session_overlap_trace.py /tmp/kt/.../s_results.db > gpurun_out/.../session_overlap_trace.json"""
import json
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
dcols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[-1])
qcol = "queue_id" if "queue_id" in dcols else None
rows = db.execute(f"select s.{name_col}, d.start, d.end{', d.' + qcol if qcol else ''} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()

HBM = re.compile(r"k_gemv|k_decode_qkv|k_attn_decode|k_attn_merge|k_attn_combine|k_gemm_skinny|k_pick|k_sample|k_decode_advance|k_rope_qk_row|k_rope_row")
MFMA = re.compile(r"k_gemm_fat|k_gemm256|k_gemm128|k_attn<|k_attn\b|km_|k_vit_embed_ln|k_pre_|k_bert_embed_ln|k_pool|k_rope_f32in|k_patchify")      # (k_norm runs on both sides: left out)


def side(n):
    if HBM.search(n):
        return 0
    if MFMA.search(n):
        return 1
    return 2


ev = []          # (time, +1 / -1, side)
busy = [0, 0, 0]
count = [0, 0, 0]
queues = [set(), set(), set()]
for r in rows:
    s = side(r[0])
    count[s] += 1
    busy[s] += r[2] - r[1]
    if qcol:
        queues[s].add(r[3])
    if s < 2:
        ev.append((r[1], 1, s))
        ev.append((r[2], -1, s))
ev.sort(key=lambda e: (e[0], e[1]))
act = [0, 0]
last = None
co = 0
union = [0, 0]
first_co = last_co = None
for t, d, s in ev:
    if last is not None and t > last:
        dt = t - last
        for x in (0, 1):
            if act[x] > 0:
                union[x] += dt
        if act[0] > 0 and act[1] > 0:
            co += dt
            first_co = last if first_co is None else first_co
            last_co = t
    act[s] += d
    last = t
# the same unions restricted to the window in which anything co-ran (= the overlapped half of the session, minus its un-overlapped head and tail)
win = [0, 0]
if first_co is not None:
    act = [0, 0]
    last = None
    for t, d, s in ev:
        if last is not None and t > last:
            a, b = max(last, first_co), min(t, last_co)
            if b > a:
                for x in (0, 1):
                    if act[x] > 0:
                        win[x] += b - a
        act[s] += d
        last = t
ns = 1e-9
out = dict(what="rocprofv3 --kernel-trace of bench.py --session: time with at least one token-loop kernel AND one MFMA-side kernel executing",
           dispatches=dict(hbm_side=count[0], mfma_side=count[1], other=count[2]),
           busy_s=dict(hbm_side_union=round(union[0] * ns, 3), mfma_side_union=round(union[1] * ns, 3)),
           co_running_s=round(co * ns, 3),
           window_s=None if first_co is None else round((last_co - first_co) * ns, 3),
           in_window=dict(hbm_side_busy_s=round(win[0] * ns, 3), mfma_side_busy_s=round(win[1] * ns, 3),
                          co_running_frac_of_hbm_side=None if not win[0] else round(co / win[0], 3),
                          co_running_frac_of_window=None if first_co is None else round(co / (last_co - first_co), 3)),
           queues=dict(hbm_side=sorted(queues[0]), mfma_side=sorted(queues[1])) if qcol else None)
print(json.dumps(out, indent=1))
