#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh <tag>  -> gpurun_out/<tag>_kernel_trace.md, <tag>_pmc.txt, <tag>_bench.json
set -u
tag=$1
ROOT=$(pwd); mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$tag
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o kt -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline < /dev/null > "$ROOT/gpurun_out/${tag}_bench_under_rocprof.json" 2> /tmp/kt_$tag.err
db=$(find /tmp/kt_$tag -name "*_results.db" | head -1)
[ -n "$db" ] && python "$ROOT/profiles/summarize_rocpd.py" "$db" > "$ROOT/gpurun_out/${tag}_kernel_trace.md"
cd "$ROOT"
BATCH=18 tools/pmc_pass.sh ${tag}_fetch FETCH_SIZE > /dev/null
BATCH=18 tools/pmc_pass.sh ${tag}_write WRITE_SIZE > /dev/null
BATCH=145 tools/pmc_pass.sh ${tag}_util VALUBusy OccupancyPercent > /dev/null
cat gpurun_out/pmc_${tag}_fetch.txt gpurun_out/pmc_${tag}_write.txt gpurun_out/pmc_${tag}_util.txt > gpurun_out/${tag}_pmc.txt
timeout 400 python bench.py --steps 5 --warmup 1 < /dev/null > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout 100 python tools/hbm_ceiling.py < /dev/null 2>/dev/null | tail -1 > gpurun_out/${tag}_hbm_ceiling.json
tail -c 600 gpurun_out/${tag}_bench.json; cat gpurun_out/${tag}_kernel_trace.md | head -14; grep -i "witness_loop" gpurun_out/${tag}_pmc.txt
