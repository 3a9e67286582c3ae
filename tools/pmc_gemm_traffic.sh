#!/bin/bash
# Fabric traffic (L2 misses: FETCH_SIZE x2 on gfx950, + WRITE_SIZE) of every GEMM shape of the C3 step, one shape per rocprofv3 pass pair,
# against the algorithmic bytes (A + W + C once).  usage (GPU box): bash tools/pmc_gemm_traffic.sh > gpurun_out/gemm_traffic.md
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "| shape | M | N | K | fetch GB (x2 corrected) | write GB | total GB | algorithmic GB | ratio | us |"; echo "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"
while read name M N K; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pg_$C; timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/pg_$C -o g -- python tools/run_one_gemm.py $M $N $K 6 > /dev/null 2>&1
  done
  python - "$name" $M $N $K $(find /tmp/pg_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pg_WRITE_SIZE -name "*.db" | head -1) <<'PY'
import sqlite3, sys
name, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
def q(path, counter):
    db = sqlite3.connect(path); tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: next(t for t in tabs if t.startswith(p))
    pe, ip, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    rows = db.execute(f"select d.id, sum(e.value), d.end - d.start from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
                      f"where p.name = ? and s.kernel_name like '%k_gemm%' group by d.id order by d.id", (counter,)).fetchall()
    rows = rows[2:]                       # skip the cold launches
    return sum(r[1] for r in rows) / len(rows) * 1024, sum(r[2] for r in rows) / len(rows) / 1e3
f, us = q(sys.argv[5], "FETCH_SIZE"); w, _ = q(sys.argv[6], "WRITE_SIZE")
alg = (M * K + N * K + M * N) * 2
print(f"| {name} | {M} | {N} | {K} | {2 * f / 1e9:.3f} | {w / 1e9:.3f} | {(2 * f + w) / 1e9:.3f} | {alg / 1e9:.3f} | {(2 * f + w) / alg:.2f} | {us:.0f} |")
PY
done <<'SHAPES'
vit.qkv 295424 3072 1024
vit.o 295424 1024 1024
vit.fc1 295424 4096 1024
vit.fc2 295424 1024 4096
proj.2 294912 3584 3584
llm.q 48994 3584 3584
llm.kv 48994 1024 3584
llm.gateup 48994 37888 3584
llm.down 48994 3584 18944
SHAPES
