#!/bin/bash
# Fabric reads (FETCH_SIZE, KiB; x2 on gfx950) and writes per launch of the k-means kernels on the merge shape (T = 400, K = 5, D = 2 064 384, 10 forced
# Lloyd iterations), one-read pass vs two-pass kernels.  usage (GPU box): bash tools/pmc_kmeans.sh > gpurun_out/pmc_kmeans.jsonl
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pk_$C; SC_KM_FUSED=$f timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pk_$C -o k -- python $GRAFT_REPO_ROOT/tools/bench_kmeans.py --T 400 --K 5 --reps 1 > /dev/null 2>&1
  done
  python - $f $(find /tmp/pk_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pk_WRITE_SIZE -name "*.db" | head -1) <<'PY'
import sqlite3, sys, json, re
def q(path, counter):
    db = sqlite3.connect(path); tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: next(t for t in tabs if t.startswith(p))
    pe, ip, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    rows = db.execute(f"select s.kernel_name, d.id, sum(e.value), d.end - d.start from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
                      f"where p.name = ? and (s.kernel_name like '%km%') group by s.kernel_name, d.id order by d.id", (counter,)).fetchall()
    out = {}
    for name, _, v, t in rows:
        m = re.search(r"km2_passILi(\d+)ELi(\d+)ELi(\d+)", name)
        key = f"km2_pass<K={m.group(1)},RGW={m.group(2)},MODE={m.group(3)}>" if m else re.sub(r"\(.*", "", name).replace("void ", "")
        if t < 20000: continue                      # launches skipped by the device-side `done` flag / tiny helper launches
        a = out.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += v; a[2] += t
    return out
f, w = q(sys.argv[2], "FETCH_SIZE"), q(sys.argv[3], "WRITE_SIZE")
X = 400 * 2064384 * 2 / 1e9
for k in f:
    n = f[k][0]
    print(json.dumps(dict(fused=int(sys.argv[1]), kernel=k, launches=n, fetch_GB_x2=round(2 * f[k][1] / n * 1024 / 1e9, 3), write_GB=round(w.get(k, [1, 0, 0])[1] / max(w.get(k, [1])[0], 1) * 1024 / 1e9, 3),
                          X_GB=round(X, 3), us=round(f[k][2] / n / 1e3, 1))))
PY
done
