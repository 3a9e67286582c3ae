#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats table (like `--stats` CSV):
    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db > profiles/rXX_kernel_stats.md"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[-1])
rows = db.execute(f"select s.{name_col}, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                  f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
for n, c, t, mn, mx in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = n if len(n) < 90 else n[:87] + "..."
    print(f"| `{n}` | {c} | {t / 1e6:.3f} | {t / c / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * t / tot:.2f} |")
