#!/bin/bash
# GPU box: start / end of the last kernels of `python bench.py "$@"` (rocprofv3 --kernel-trace), relative ms, with their queue
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tlb
timeout 600 rocprofv3 --kernel-trace -d /tmp/tlb -o kt -- python $ROOT/bench.py --no-cpu-baseline "$@" > /tmp/tlb.out 2>/tmp/tlb.err
db=$(find /tmp/tlb -name "*_results.db" | head -1)
python - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; sym = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(db.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {sym} s on d.kernel_id = s.id order by d.start"))
# the timed steps: find the k_witness_loop launches and print from the 3rd to the 5th of them
loops = [i for i, r in enumerate(rows) if 'k_witness_loopE' in r[0]]
a, b = loops[2], loops[4]
t0 = rows[a][1]
for n, s, e, q in rows[a:b + 1]:
    if (e - s) < 20000: continue
    print(f"{(s - t0) / 1e6:9.3f} -> {(e - t0) / 1e6:9.3f} ms  q{q}  {n.split('(')[0][4:44]}")
PY
