#!/bin/bash
# GPU box: TCC (L2 <-> fabric) counters of every zke::k_witness_loop launch of tools/placement_probe.py — ten re-allocations of the store
# inside one process, alternating between a faster and a slower region — to see what the slower placement looks like at the L2's
# external interface: write-request stalls, DRAM credit stalls, and how evenly the requests spread over the counter instances (channels).
# -> gpurun_out/tcc_probe.txt
ROOT=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for pass in "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL" "TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_BUSY"; do
  rm -rf /tmp/tcc
  timeout 600 rocprofv3 --pmc $pass --kernel-trace -d /tmp/tcc -o p -- python $ROOT/tools/placement_probe.py > /tmp/tcc.out 2> /tmp/tcc.err
  db=$(find /tmp/tcc -name "*_results.db" | head -1)
  [ -z "$db" ] && { echo "no db"; tail -3 /tmp/tcc.err; continue; }
  python - "$db" <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in db.execute("pragma table_info(pmc_events)")]
# pmc_events is a view: one row per (dispatch, counter, dimension instance)
key = "dispatch_id" if "dispatch_id" in cols else ("event_id" if "event_id" in cols else cols[0])
rows = db.execute(f"select {key}, name, counter_name, counter_value from pmc_events where name like '%k_witness_loop%'").fetchall()
by = collections.defaultdict(lambda: collections.defaultdict(list))
for d, n, c, v in rows: by[d][c].append(v)
kd = [t for t in tabs if 'kernel_dispatch' in t]
print("dispatch  " + "  ".join(sorted({c for d in by for c in by[d]})))
for d in sorted(by):
    parts = []
    for c in sorted(by[d]):
        v = by[d][c]
        parts.append(f"{c}: sum {sum(v):.4g} n {len(v)} min {min(v):.4g} max {max(v):.4g}")
    print(d, " | ".join(parts))
PY
  grep loop_ms /tmp/tcc.out
done > $ROOT/gpurun_out/tcc_probe.txt 2>&1
cat $ROOT/gpurun_out/tcc_probe.txt | cut -c1-260
