#!/bin/bash
# usage (on the GPU box, repo root): tools/pmc_pass.sh <tag> <counter> [<counter> ...]   -> gpurun_out/pmc_<tag>.txt
# One rocprofv3 --pmc pass (kernel-trace only, as gpurun requires) over a small bench batch; prints per-kernel means.
# PMC_CMD overrides the profiled command (default: bench.py at BATCH instances), e.g. PMC_CMD="python tools/ntt_bench.py 20 164".
set -u
tag=$1; shift
ROOT=$(pwd)
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_$tag -o p -- ${PMC_CMD:-python "$ROOT/bench.py" --batch ${BATCH:-18} --steps 2 --warmup 0 --no-cpu-baseline} < /dev/null > /tmp/pmc_$tag.log 2>&1
db=$(find /tmp/pmc_$tag -name "*_results.db" | head -1)
if [ -z "$db" ]; then echo "no db for $tag"; tail -5 /tmp/pmc_$tag.log; exit 0; fi
python - "$db" > "$ROOT/gpurun_out/pmc_$tag.txt" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for r in db.execute("select name, counter_name, count(*), avg(counter_value), max(counter_value) from pmc_events group by name, counter_name order by name"):
    print(f"{r[0].split('(')[0]:40s} {r[1]:28s} n={r[2]:3d} mean={r[3]:.4g} max={r[4]:.4g}")
PY
cat "$ROOT/gpurun_out/pmc_$tag.txt"
