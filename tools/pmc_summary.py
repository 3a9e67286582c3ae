#!/usr/bin/env python3
"""Average rocprofv3 PMC counters per kernel from a rocpd sqlite db:  python tools/pmc_summary.py db [kernel-substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: next(t for t in tabs if t.startswith(p))
pe, ip, kd, ks, ev = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_event")
cols = [r[1] for r in db.execute(f"pragma table_info({pe})")]
q = (f"select s.kernel_name, p.name, avg(e.value), count(*) from {pe} e join {ip} p on e.pmc_id = p.id "
     f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.kernel_name, p.name")
try:
    rows = db.execute(q).fetchall()
except Exception as ex:
    print("schema:", cols, [r[1] for r in db.execute(f"pragma table_info({ip})")], [r[1] for r in db.execute(f"pragma table_info({kd})")]); raise
for k, n, v, c in rows:
    if pat in k:
        print(f"{k[:60]:60s} {n:32s} {v:16.1f} (n={c})")
