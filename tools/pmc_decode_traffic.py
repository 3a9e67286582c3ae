#!/usr/bin/env python3
"""HBM bytes per decode token from a rocprofv3 FETCH_SIZE pass over tools/bench_decode.py (rocpd sqlite): per kernel family and in total, next to the
algorithmic bytes (14.1 GB of weights + the KV cache).  gfx950: FETCH_SIZE (KiB) x 2 for 16 B / lane coalesced reads (see tools/pmc_traffic.py).
    python tools/pmc_decode_traffic.py <fetch_db> <tokens> <context> <out.json>"""
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); tokens, ctx = int(sys.argv[2]), int(sys.argv[3])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: next(t for t in tabs if t.startswith(p))
pe, ip, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
q = (f"select s.kernel_name, d.id, sum(e.value) from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id "
     f"join {ks} s on d.kernel_id = s.id where p.name = 'FETCH_SIZE' group by s.kernel_name, d.id")
fam = {}
for name, _, v in db.execute(q):
    key = next((k for k in ("k_gemv_u", "k_gemv", "k_decode_qkv", "k_attn_decode", "k_attn_combine", "k_rope", "k_pick", "k_gather_rows", "k_decode_advance") if k in name), None)
    if key is None:
        continue
    a = fam.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += v
per_tok = {k: dict(launches_per_token=round(n / tokens, 2), GB_per_token=round(2.0 * kb * 1024 / tokens / 1e9, 3)) for k, (n, kb) in fam.items()}
total = sum(v["GB_per_token"] for v in per_tok.values())
algo = 14.1 + 2 * 28 * 4 * 128 * ctx * 2 / 1e9
out = dict(tokens=tokens, context=ctx, kernels=per_tok, fetched_GB_per_token=round(total, 2), algorithmic_GB_per_token=round(algo, 2), ratio=round(total / algo, 3),
           correction="gfx950: FETCH_SIZE (KiB) x 2")
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out, indent=1))
