#!/usr/bin/env python3
"""GPU idle gaps in a rocprofv3 kernel trace (rocpd sqlite): gaps between consecutive kernel executions longer than a threshold, with the
kernels on either side.   python tools/gap_report.py <results.db> [min_gap_us=100] [last_n_ms=1400]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 1e5
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
nc = "display_name" if "display_name" in scols else "kernel_name"
rows = db.execute(f"select d.start, d.end, s.{nc} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
last = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 1.4e9
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - last]                     # the last step only
short = lambda n: re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", n)[:60]
busy = sum(r[1] - r[0] for r in rows); span = rows[-1][1] - rows[0][0]
gaps = []
cur_end = rows[0][1]
for a, b in zip(rows, rows[1:]):
    cur_end = max(cur_end, a[1])
    g = b[0] - cur_end
    if g > thr: gaps.append((g, short(a[2]), short(b[2]), (b[0] - rows[0][0]) / 1e6))
print(f"window {span / 1e6:.1f} ms, kernels busy {busy / 1e6:.1f} ms, idle {(span - busy) / 1e6:.1f} ms; gaps > {thr / 1e3:.0f} us: {len(gaps)} totalling {sum(g[0] for g in gaps) / 1e6:.2f} ms")
for g, a, b, t in sorted(gaps, key=lambda x: -x[0])[:40]:
    print(f"  {g / 1e3:8.1f} us at t={t:8.1f} ms   after {a:<50s} before {b}")
