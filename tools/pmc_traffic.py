#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; rocpd sqlite) -> profiles/rXX_pmc_traffic.json.
    python tools/pmc_traffic.py <fetch_db> <write_db> <out.json>
gfx950: FETCH_SIZE (KiB) reports half of the bytes of 16 B/lane coalesced reads (guide MI355X_MICROARCH.md, HBM section; calibrated on
km_update, which reads X exactly once) -> x2; WRITE_SIZE (KiB) as reported."""
import json, sqlite3, sys

def per_kernel(path, counter):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: next(t for t in tabs if t.startswith(p))
    pe, ip, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    q = (f"select s.kernel_name, d.id, sum(e.value) from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id "
         f"join {ks} s on d.kernel_id = s.id where p.name = ? group by s.kernel_name, d.id")
    out = {}
    for name, _, v in db.execute(q, (counter,)):
        a = out.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += v
    return out

FAMILIES = {"k_gemm_all": "k_gemm", "k_gemm256": "k_gemm256", "k_attn_dh128_causal": "k_attnILi128ELi", "k_attn_dh64": "k_attnILi64ELi", "km_assign": "km_assign", "km_update": "km_update", "km2_pass": "km2_pass"}
fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for fam, pat in FAMILIES.items():
    fl = [(n, v) for n, v in fetch.items() if pat in n and (fam != "k_attn_dh128_causal" or "ELb1E" in n)]
    wl = [(n, v) for n, v in write.items() if pat in n and (fam != "k_attn_dh128_causal" or "ELb1E" in n)]
    nf, f = sum(v[0] for _, v in fl), sum(v[1] for _, v in fl)
    nw, w = sum(v[0] for _, v in wl), sum(v[1] for _, v in wl)
    if nf == 0:
        continue
    res[fam] = dict(launches=nf, fetch_kb_per_launch=f / nf, write_kb_per_launch=(w / nw if nw else 0.0),
                    hbm_bytes_per_launch=(2.0 * f / nf + (w / nw if nw else 0.0)) * 1024)
json.dump(dict(command="rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --decode-tokens 0 --with-captions 0 (two separate passes; "
                       "every launch of the warm-up and the two timed steps is counted: a warm multi-step run, averaged per launch)",
               correction="gfx950: FETCH_SIZE (KiB) x2 (calibrated on km_update: 2 x 826,624 KiB = T*D*2 bytes); WRITE_SIZE (KiB) as reported", kernels=res),
          open(sys.argv[3], "w"), indent=1)
print(json.dumps(res, indent=1))
