cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/tl
CONFIGS=C3k timeout 600 rocprofv3 --kernel-trace -d /tmp/tl -o kt -- python $1/tests/config_timings.py > /tmp/tl.out 2>/tmp/tl.err
db=$(find /tmp/tl -name "*_results.db" | head -1)
python - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
sym = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(db.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {sym} s on d.kernel_id = s.id order by d.start"))
rows = rows[-40:]
t0 = rows[0][1]
for n, a, b, q in rows:
    print(f"{(a - t0) / 1e6:9.3f} -> {(b - t0) / 1e6:9.3f} ms  q{q}  {n.split('(')[0][:60]}")
PY
