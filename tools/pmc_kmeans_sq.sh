#!/bin/bash
# Where the cycles of km2_pass go (SQ counters, quad-cycle units): one rocprofv3 pass over tools/bench_kmeans.py (T = 400, K = 5, 10 forced iterations).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq; timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -d /tmp/pq -o k -- python $GRAFT_REPO_ROOT/tools/bench_kmeans.py --T 400 --K 5 --reps 1 > /tmp/pq.log 2>&1
python - $(find /tmp/pq -name "*.db" | head -1) <<'PY'
import sqlite3, sys, json, re
db = sqlite3.connect(sys.argv[1]); tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: next(t for t in tabs if t.startswith(p))
pe, ip, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
rows = db.execute(f"select s.kernel_name, d.id, p.name, sum(e.value), d.end - d.start from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
                  f"where s.kernel_name like '%km2_pass%' or s.kernel_name like '%km_update%' or s.kernel_name like '%km_assign%' group by s.kernel_name, d.id, p.name").fetchall()
acc = {}
for name, did, c, v, t in rows:
    if t < 20000: continue
    m = re.search(r"km2_passILi(\d+)ELi(\d+)ELi(\d+)", name)
    key = f"km2_pass<MODE={m.group(3)}>" if m else ("km_update" if "km_update" in name else "km_assign")
    a = acc.setdefault(key, {}); a[c] = a.get(c, 0.0) + v; a.setdefault("_ids", set()).add(did); a["_ns"] = a.get("_ns", 0) + (t if c == "SQ_WAVE_CYCLES" else 0)
for k, a in acc.items():
    n = len(a.pop("_ids")); ns = a.pop("_ns")
    w = a.get("SQ_WAVE_CYCLES", 1.0)
    print(json.dumps(dict(kernel=k, launches=n, us=round(ns / n / 1e3, 1), valu_insts_per_launch=round(a.get("SQ_INSTS_VALU", 0) / n),
                          of_wave_cycles=dict(active_any=round(a.get("SQ_ACTIVE_INST_ANY", 0) / w, 3), active_valu=round(a.get("SQ_ACTIVE_INST_VALU", 0) / w, 3),
                                              active_lds=round(a.get("SQ_ACTIVE_INST_LDS", 0) / w, 3), wait_any=round(a.get("SQ_WAIT_ANY", 0) / w, 3),
                                              wait_inst_any=round(a.get("SQ_WAIT_INST_ANY", 0) / w, 3)),
                          valu_busy_of_sq_busy=round(a.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(a.get("SQ_BUSY_CYCLES", 1), 1), 3))))
PY
tail -3 /tmp/pq.log | grep -i "error" 
