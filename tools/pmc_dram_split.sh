#!/bin/bash
# Which share of the GEMMs' fabric reads (L2 misses) is served by HBM and which by the Infinity Cache?  
# One rocprofv3 pass per target with the three L2 -> fabric read counters TCC_EA0_RDREQ (all), _32B, _DRAM ("destined for DRAM (MC)").
# The calibration pair says what _DRAM means on this chip: a 64 MB matrix streamed 24 times (Infinity-Cache resident after pass 1) against 24
# different matrices (HBM).  usage (GPU box): bash tools/pmc_dram_split.sh > gpurun_out/dram_split.md
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "| target | kernel launches | RDREQ per launch | 32B share | DRAM share of RDREQ | bytes per launch (64B x RDREQ - 32 x RDREQ_32B) GB | us |"; echo "|---|---:|---:|---:|---:|---:|---:|"
run() {   # name, kernel pattern, command...
  name=$1; pat=$2; shift 2
  rm -rf /tmp/pd; timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum -d /tmp/pd -o g -- "$@" > /dev/null 2>&1
  python - "$name" "$pat" $(find /tmp/pd -name "*.db" | head -1) <<'PY'
import sqlite3, sys
name, pat, path = sys.argv[1:4]
db = sqlite3.connect(path); tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: next(t for t in tabs if t.startswith(p))
pe, ip, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
rows = db.execute(f"select d.id, p.name, sum(e.value), d.end - d.start from {pe} e join {ip} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
                  f"where s.kernel_name like ? group by d.id, p.name order by d.id", (f"%{pat}%",)).fetchall()
ids = sorted({r[0] for r in rows})[2:]                 # skip the cold launches
acc, us = {}, {}
for i, n, v, t in rows:
    if i in ids:
        acc[n] = acc.get(n, 0.0) + v; us[i] = t / 1e3
k = len(ids)
g = lambda s: acc.get(s + "_sum", acc.get(s, 0.0)) / max(k, 1)
rd, r32, dr = g("TCC_EA0_RDREQ"), g("TCC_EA0_RDREQ_32B"), g("TCC_EA0_RDREQ_DRAM")
print(f"| {name} | {k} | {rd:.3e} | {r32 / max(rd, 1):.3f} | {dr / max(rd, 1):.3f} | {(64 * rd - 32 * r32) / 1e9:.3f} | {sum(us.values()) / max(k, 1):.0f} |")
PY
}
run "calib: 64 MB x 24 passes (Infinity-Cache resident)" k_gemv python tools/run_stream_read.py hot
run "calib: 24 x 64 MB distinct (HBM)" k_gemv python tools/run_stream_read.py cold
run "llm.gateup 48994x37888x3584" k_gemm python tools/run_one_gemm.py 48994 37888 3584 6
run "llm.down 48994x3584x18944" k_gemm python tools/run_one_gemm.py 48994 3584 18944 6
run "vit.fc1 295424x4096x1024" k_gemm python tools/run_one_gemm.py 295424 4096 1024 6
